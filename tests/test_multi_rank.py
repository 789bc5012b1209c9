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


# ----------------------------------------------------------------------------- the DDP driver end to end (world 2, gloo, stub networks)
class _StubModel(torch.nn.Module):
    """v(t, x) = -x: the Euler solve of 4 steps multiplies the noise by (1 - 0.25)^4... any deterministic field will do."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))

    def forward(self, t, x, y=None, **kw):
        return -x + 0.1 * torch.as_tensor(t, dtype=x.dtype).reshape(-1, *([1] * (x.dim() - 1)))[:1]


class _StubVae:
    def decode(self, z):
        img = torch.tanh(z[:, :3].repeat_interleave(2, -1).repeat_interleave(2, -2))  # [N,3,2R,2R] in (-1, 1)
        return type("O", (), {"sample": img})()


def _ddp_worker(rank, world, port, q, outdir, generator="determ"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from lfm_amd import test_flow_latent_ddp as ddp
    from lfm_amd.io_formats import to_uint8_truncating

    saved = []
    def build(a, d):
        torch.manual_seed(a.seed)  # what build_models does with --random_weights: the GLOBAL generator, un-offset seed, on every rank
        return _StubModel().eval(), _StubVae()

    hooks = dict(backend="gloo", device="cpu", build_models=build, to_uint8=to_uint8_truncating,
                 save=lambda block, start: saved.append((start, block.clone())))
    argv = ["--model_type", "DiT-S/2", "--image_size", "32", "--num_in_channels", "4", "--n_sample", "10", "--batch_size", "3", "--method", "euler",
            "--step_size", "0.25", "--generator", generator, "--seed", "7", "--compute_fid", "--save_dir", outdir]
    res = ddp.main(argv, hooks=hooks)
    q.put((rank, res["total"], res["iters"], res["written"], [(s, tuple(b.shape), b.numpy().tobytes()) for s, b in saved]))  # plain bytes: the
    # producer exits right after the put, tensors would travel as shared-memory handles that die with it


def test_world2_ddp_main_end_to_end(tmp_path):
    """Drives lfm_amd.test_flow_latent_ddp.main with stub networks on a 2-rank gloo group: loop count, gather buffer reuse across
    iterations (two alternating buffers), file indices j*world + rank + total (reference :138), rank-local seeds (seed + rank, :30) with the
    determ generator's rank slicing, and rank 0 as the only writer."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, total, iters, written, saved), (r1, total1, iters1, written1, saved1) = res
    assert (total, iters) == (12, 2) and (total1, iters1) == (12, 2)
    assert written == [(0, 6), (6, 6)] and written1 == [] and saved1 == []
    # recompute what every rank must have produced: rank r draws with seed 7 + r and takes rows r, r + 2, r + 4 of each global draw
    from lfm_amd.io_formats import to_uint8_truncating
    from lfm_amd.sampler.random_util import DeterministicGenerator

    def rank_batches(r):
        gen = DeterministicGenerator(10, 7 + r)
        gen.rank, gen.world_size = r, world
        outs = []
        for _ in range(2):
            x = gen.randn(3, 4, 4, 4)
            for k in range(4):
                t = 1.0 - 0.25 * k
                x = x + (-0.25) * (-x + 0.1 * t)
            outs.append(to_uint8_truncating(_StubVae().decode(x / 0.18215).sample))  # run_sampling divides by --scale_factor
        return outs

    want = [rank_batches(0), rank_batches(1)]
    assert len(saved) == 2
    for it, (start, shape, raw) in enumerate(saved):
        block = torch.frombuffer(bytearray(raw), dtype=torch.uint8).reshape(shape)
        assert start == it * 6 and block.shape == (6, 8, 8, 3)
        for j in range(3):
            for r in range(world):  # global index j*world + r + start  ->  position j*world + r of the gathered block
                assert torch.equal(block[j * world + r], want[r][it][j]), (it, j, r)


def test_world2_dummy_generator_draws_differ_per_rank(tmp_path):
    """--generator dummy draws from torch's GLOBAL generator: the reference seeds it with seed + rank (test_flow_latent_ddp.py:28-32), so the two
    ranks' images must differ -- they were identical when main() left the un-offset seed of build_models in place (advisor finding, round 2)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q, str(tmp_path), "dummy")) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    saved = res[0][4]
    assert len(saved) == 2
    for start, shape, raw in saved:
        block = torch.frombuffer(bytearray(raw), dtype=torch.uint8).reshape(shape)  # [6, 8, 8, 3]: position j * world + r
        r0, r1 = block[0::2], block[1::2]
        assert not torch.equal(r0, r1)
        assert (r0 != r1).float().mean() > 0.5
    # and rank 0's first batch is what seed 7 + 0 gives
    torch.manual_seed(7)
    x = torch.randn(3, 4, 4, 4)
    for k in range(4):
        x = x + (-0.25) * (-x + 0.1 * (1.0 - 0.25 * k))
    from lfm_amd.io_formats import to_uint8_truncating

    want = to_uint8_truncating(_StubVae().decode(x / 0.18215).sample)
    block = torch.frombuffer(bytearray(saved[0][2]), dtype=torch.uint8).reshape(saved[0][1])
    assert torch.equal(block[0::2], want)


def test_bench_self_launches_two_gloo_ranks():
    """`python bench.py --gpus 2` WITHOUT a torchrun environment (how the driver starts the scaling leg) must become the launcher itself:
    two ranks over 127.0.0.1, one JSON line from rank 0 with the contract's keys.  --stub swaps the HIP step for a CPU stub (gloo)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_world"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["value"] > 0 and abs(j["value"] - 2 * 4 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]
    assert [p["rank"] for p in j["per_rank"]] == [0, 1] and all(p["images_per_sec"] > 0 for p in j["per_rank"])  # every rank reports itself in the one line


def test_bench_refuses_a_mismatched_world():
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"WORLD_SIZE=1" in r.stderr


def test_bench_flop_closed_forms_match_the_oracle():
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from oracle import dit_ref, vae_ref

    for R in (8, 32, 64):
        assert bench.vae_decode_flops(R) == vae_ref.vae_decode_flops(R)
    assert abs(bench.vae_decode_flops(32) / 1e9 - 622.2) < 0.05
    for name in ("DiT-S/2", "DiT-B/2", "DiT-L/2", "DiT-XL/2", "DiT-B/4", "DiT-L/8"):
        c = dit_ref.DiTCfg.named(name, num_classes=1, label_dropout=0.0)
        assert bench.dit_flops_per_image(c.tokens, c.hidden, c.depth, c.patch * c.patch * c.in_ch) == dit_ref.dit_flops_per_image(c)
    # config 6: the constant is the hook count the reference module produced for the fixture (oracle/make_golden.py::golden_edm_full)
    import torch

    rec = torch.load(os.path.join(root, "tests", "golden", "edm_full.pt"), map_location="cpu", weights_only=False)["ffhq_adm"]
    assert bench.EDM_FFHQ_FLOP_PER_IMAGE == rec["flops_per_image"]


# ----------------------------------------------------------------------------- world 8: the shape of the round-end scaling run
def test_shard_plan_50000_images_world8_batch64():
    """The reference's FID run (test_flow_latent_ddp.py:116-126,138; n_sample 50000) on one 8-GPU node at batch 64: 50000 is not a multiple of the global
    batch 512, so 98 iterations sample 50176 images; every global file index 0 .. 50175 is produced exactly once, by the rank / iteration / local image
    the reference's `index = j * world + rank + total` names."""
    from lfm_amd.test_flow_latent_ddp import global_indices, shard_plan

    world, B = 8, 64
    total, per_rank, iters = shard_plan(50000, B, world)
    assert (total, per_rank, iters) == (50176, 6272, 98)
    seen = torch.zeros(total, dtype=torch.int32)
    for it in range(iters):
        for r in range(world):
            idx = torch.tensor(global_indices(B, world, r, it))
            assert idx[0] == it * B * world + r and idx[1] - idx[0] == world
            seen[idx] += 1
    assert bool((seen == 1).all())


def _ddp_worker8(rank, world, port, q, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from lfm_amd import test_flow_latent_ddp as ddp

    saved, calls = [], [0]

    def build(a, d):
        torch.manual_seed(a.seed)
        return _StubModel().eval(), _StubVae()

    def tag(img):  # instead of pixels: every image carries (rank, iteration, local index) so that rank 0 can check where the gather put it
        it = calls[0]
        calls[0] += 1
        n = img.shape[0]
        t = torch.zeros(n, 2, 2, 3, dtype=torch.uint8)
        t[..., 0], t[..., 1], t[..., 2] = rank, it, torch.arange(n, dtype=torch.uint8).reshape(n, 1, 1)
        return t

    hooks = dict(backend="gloo", device="cpu", build_models=build, to_uint8=tag, save=lambda block, start: saved.append((start, block[:, 0, 0].clone())))
    argv = ["--model_type", "DiT-S/2", "--image_size", "32", "--num_in_channels", "4", "--n_sample", "1000", "--batch_size", "64", "--method", "euler",
            "--step_size", "0.5", "--generator", "determ", "--seed", "3", "--compute_fid", "--save_dir", outdir]
    res = ddp.main(argv, hooks=hooks)
    q.put((rank, res["total"], res["iters"], res["written"], [(s, b.tolist()) for s, b in saved]))


def test_world8_ddp_main_gather_order(tmp_path):
    """Eight gloo ranks through main(): n_sample 1000 at batch 64 -> 1024 images in 2 iterations (non-divisible tail).  Only rank 0 writes; block i starts at
    file index i * 512 and position p of a block holds local image p // 8 of rank p % 8 of that iteration (the reference's j * world + rank)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker8, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, total, iters, written, saved in res:
        assert (total, iters) == (1024, 2)
        if rank != 0:
            assert saved == [] and written == []
            continue
        assert written == [(0, 512), (512, 512)]
        assert [s for s, _ in saved] == [0, 512]
        for i, (_, tags) in enumerate(saved):
            for p, (r, it, j) in enumerate(tags):
                assert (r, it, j) == (p % world, i, p // world), (i, p, r, it, j)
