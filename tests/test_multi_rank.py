"""World-size-2 tests of the N>1 path on CPU (gloo): shard math, rank-sliced deterministic noise, all-gather ordering."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lfm_amd.sampler.random_util import get_generator
        from lfm_amd.test_flow_latent_ddp import gather_images, global_indices, shard_plan

        B = 3
        total, per_rank, iters = shard_plan(10, B, world)
        # rank-sliced deterministic noise: rows rank, rank+world, ... of ONE global draw (random_util.py:58-67)
        gen = get_generator("determ", 64, 42)
        x = gen.randn(B, 2, 2)
        # every rank tags its images with their reference global index, then gathers
        idx = torch.tensor(global_indices(B, world, rank, iteration=1))
        img = idx.to(torch.uint8).reshape(B, 1, 1, 1).expand(B, 2, 2, 3).contiguous()
        allimg = gather_images(img, world)
        q.put((rank, total, per_rank, iters, x, allimg[:, 0, 0, 0].tolist(), gen.rank, gen.world_size))
    finally:
        dist.destroy_process_group()


def test_world2_sharding_noise_and_gather_order():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full = torch.randn(64, 2, 2, generator=torch.Generator().manual_seed(42))
    for rank, total, per_rank, iters, x, order, grank, gworld in res:
        assert (total, per_rank, iters) == (12, 6, 2)  # ceil(10 / 6) * 6
        assert (grank, gworld) == (rank, world)
        assert torch.equal(x, full[rank::world][:3])  # rows rank, rank+2, rank+4
        # gathered block is in the reference's file-index order: total_so_far + j*world + rank  (ddp.py:138)
        assert order == list(range(6, 12))


def test_interleave_is_a_permutation():
    from lfm_amd.test_flow_latent_ddp import interleave_ranks

    g = torch.arange(4 * 5).reshape(20, 1)  # world 4, B 5, rank-major
    out = interleave_ranks(g, 4).flatten().tolist()
    assert sorted(out) == list(range(20))
    assert out[:4] == [0, 5, 10, 15]  # image 0 of ranks 0..3 first
