"""Multi-GPU sampling driver (drop-in for /root/reference/test_flow_latent_ddp.py): one process per GPU under torchrun.

Sharding is the reference's (``:116-126,138``): global batch = batch_size * world, rank-local seed = seed + rank, image
``j`` of iteration ``i`` on rank ``r`` has global index ``j*world + r + i*batch*world``.  What replaces the reference's
cross-rank mechanism -- every rank writes JPEGs into a shared directory, rank 0 reads them back (``:137-150``) -- is ONE RCCL
``all_gather_into_tensor`` of the uint8 NHWC image block per batch (12.6 MB per rank at batch 64, 256x256), interleaved back
into the reference's index order.  No collective runs during the solve.
"""
import math
import os

import torch
import torch.distributed as dist

from .sampler.random_util import get_generator
from .test_flow_latent import build_models, build_parser, run_sampling, save_images_uint8

__all__ = ["shard_plan", "gather_images", "interleave_ranks", "global_indices", "main"]


def shard_plan(n_sample, batch_size, world_size):
    """(total_samples, samples_per_rank, iterations) -- reference :116-123."""
    global_batch = batch_size * world_size
    total = int(math.ceil(n_sample / global_batch) * global_batch)
    assert total % world_size == 0
    per_rank = total // world_size
    assert per_rank % batch_size == 0
    return total, per_rank, per_rank // batch_size


def global_indices(batch_size, world_size, rank, iteration):
    """Global image index of local image j (reference :138): j*world + rank + total_so_far."""
    total = iteration * batch_size * world_size
    return [j * world_size + rank + total for j in range(batch_size)]


def interleave_ranks(gathered, world_size):
    """[world*B, ...] rank-major (all_gather layout) -> [B*world, ...] in the reference's index order (j-major, rank-minor)."""
    B = gathered.shape[0] // world_size
    return gathered.reshape(world_size, B, *gathered.shape[1:]).transpose(0, 1).reshape(gathered.shape)


def gather_images(img_u8, world_size, out=None):
    """All ranks contribute [B,H,W,3] uint8; every rank receives [B*world,H,W,3] ordered like the reference's file indices."""
    if world_size == 1:
        return img_u8
    if out is None:
        out = torch.empty(world_size * img_u8.shape[0], *img_u8.shape[1:], dtype=img_u8.dtype, device=img_u8.device)
    dist.all_gather_into_tensor(out, img_u8.contiguous())
    return interleave_ranks(out, world_size)


def main(argv=None):
    from .autoencoder import images_to_uint8

    parser = build_parser()
    args = parser.parse_args(argv)
    torch.set_grad_enabled(False)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm
    rank, world = dist.get_rank(), dist.get_world_size()
    model, vae = build_models(args, device)  # un-offset seed: --random_weights must give every rank the SAME synthetic model
    args.seed = args.seed + rank  # reference :30 (rank-local noise / label stream)
    generator = get_generator(args.generator, args.n_sample, args.seed)
    total, _, iters = shard_plan(args.n_sample, args.batch_size, world)
    save_dir = args.save_dir or "./generated_samples/{}/exp{}_ep{}_m{}".format(args.dataset, args.exp, args.epoch_id, args.method)
    if rank == 0:
        print(f"Total number of images that will be sampled: {total}")
    buf = None
    for i in range(iters):
        img = run_sampling(model, vae, args, args.batch_size, generator, device)
        u8 = images_to_uint8(img)
        if buf is None:
            buf = torch.empty(world * u8.shape[0], *u8.shape[1:], dtype=u8.dtype, device=device)
        allimg = gather_images(u8, world, out=buf)
        if rank == 0 and args.compute_fid:
            save_images_uint8(allimg, save_dir, i * args.batch_size * world)
    dist.barrier()
    if rank == 0:
        print(f"sampled {total} images on {world} GPUs" + (f" -> {save_dir}" if args.compute_fid else ""))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
