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

__all__ = ["shard_plan", "gather_images", "interleave_ranks", "global_indices", "GatherPipeline", "run", "main"]


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


class GatherPipeline:
    """One ``all_gather_into_tensor`` of the uint8 image block per batch, issued on a SIDE stream so that the next batch's solve
    overlaps the collective and the device-to-host copy of the previous one.  ``depth`` batches stay pending
    (``depth + 1`` gather buffers alternate): ``submit`` returns the gathered, reference-ordered block of the batch submitted
    ``depth`` calls earlier (or None), ``flush`` the oldest pending one (None when nothing is pending).  depth = 1 is "the previous
    batch"; the drivers use depth = lanes, so that the block the host goes on to convert and write belongs to a batch that is
    no longer one of the ``lanes`` batches in flight.  On a CPU / gloo group (the world-2 rehearsal in tests/) it degrades to the
    plain blocking collective."""

    def __init__(self, world, device, depth=1):
        from collections import deque

        self.world, self.device = world, torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(self.device) if self.cuda and world > 1 else None
        self.depth = max(1, int(depth))
        self.bufs, self.k, self.pending = [None] * (self.depth + 1), 0, deque()
        self.gather_seconds = 0.0

    def _gather(self, u8):
        import time

        k = self.k
        self.k = (self.k + 1) % len(self.bufs)
        if self.world == 1:
            if self.cuda and u8.is_cuda:  # no collective, but the consumer may run on another stream than the producer (batches in flight on lanes)
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self.device))
                return u8, done
            return u8
        if self.bufs[k] is None or self.bufs[k].shape[0] != self.world * u8.shape[0]:
            self.bufs[k] = torch.empty(self.world * u8.shape[0], *u8.shape[1:], dtype=u8.dtype, device=u8.device)
        if self.side is None:
            t0 = time.perf_counter()
            dist.all_gather_into_tensor(self.bufs[k], u8.contiguous())
            self.gather_seconds += time.perf_counter() - t0
            return interleave_ranks(self.bufs[k], self.world)
        self.side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            u8.record_stream(self.side)
            dist.all_gather_into_tensor(self.bufs[k], u8.contiguous())
            out = interleave_ranks(self.bufs[k], self.world)
            done = torch.cuda.Event()
            done.record(self.side)
        return out, done

    @staticmethod
    def _wait(p):
        if isinstance(p, tuple):
            p[1].synchronize()
            return p[0]
        return p

    def submit(self, u8):
        self.pending.append(self._gather(u8))
        return self._wait(self.pending.popleft()) if len(self.pending) > self.depth else None

    def flush(self):
        return self._wait(self.pending.popleft()) if self.pending else None


def run(args, model, vae, generator, rank, world, device, to_uint8, save=None, log=print, extractor=None):
    """The reference's sampling loop (test_flow_latent_ddp.py:116-146) for one rank: ``iters`` batches, each solved and decoded
    locally, converted to uint8 on the device, all-gathered (overlapped with the next batch) and -- on rank 0 under --compute_fid --
    written out under the reference's global file indices ``j*world + rank + total`` (:138).  Returns what was done, for tests."""
    total, _, iters = shard_plan(args.n_sample, args.batch_size, world)
    if rank == 0:
        log(f"Total number of images that will be sampled: {total}")
    written = []

    def sink(block, i):
        if block is not None and rank == 0 and args.compute_fid and save is not None:
            save(block, i * args.batch_size * world)
            written.append((i * args.batch_size * world, int(block.shape[0])))

    extract = extractor
    if extract is None and args.compute_fid and getattr(args, "fid_feature_extractor", ""):
        from .fid import load_feature_extractor

        extract = load_feature_extractor(args.fid_feature_extractor, device)
    if extract is not None and args.compute_fid:
        # on-device FID statistics: every rank reduces its own batches to (n, sum x, sum x x^T); ONE all_reduce of those sums replaces the
        # per-batch all-gather of pixels and the JPEG rendezvous (lfm_amd/fid.py; reference fid_score.py:114-174, 230-251)
        from .fid import FeatureStatistics

        stats = None
        for i in range(iters):
            feats = extract(to_uint8(run_sampling(model, vae, args, args.batch_size, generator, device)))
            stats = stats or FeatureStatistics(feats.shape[1], feats.device)
            stats.update(feats)
        stats.all_reduce()
        return {"total": total, "iters": iters, "written": written, "gather_seconds": 0.0, "fid_stats": stats}
    # Batches in flight per GPU: the batches of a sampling job are independent (reference :128-146), so consecutive batches go to alternating HIP streams
    # ("lanes": the same weights, own scratch and captured solver graphs -- solvers.concurrency_twin) and overlap on the chip; results are bit-identical to one
    # lane (tests/test_gpu_cosched.py).  Noise is still drawn in batch order from the one generator, file indices and the gather order do not change.
    n_lanes = int(getattr(args, "in_flight", 0) or 0) or (2 if torch.device(device).type == "cuda" else 1)
    lanes = [(model, vae, None)]
    if n_lanes > 1 and torch.device(device).type == "cuda":
        from .test_flow_latent import make_lanes

        lanes = make_lanes(model, vae, device, n_lanes)
    # The host's share (device-to-host copy, JPEG encoding on rank 0) is taken OUTSIDE the lane's stream context, on a copy stream of its own, and for the batch
    # submitted `lanes` iterations ago: inside the context the copy queued on the lane stream behind the batch just enqueued, the host could not launch the next
    # batch before that one had finished, and with world > 1 every rank waited for rank 0 at the next all-gather (round-5 advisor finding).
    depth = len(lanes) if lanes[0][2] is not None else 1
    pipe = GatherPipeline(world, device, depth=depth)
    copy_stream = torch.cuda.Stream(device) if lanes[0][2] is not None else None

    def drain(block, i):
        if copy_stream is None:
            sink(block, i)
        else:
            with torch.cuda.stream(copy_stream):
                sink(block, i)

    for i in range(iters):
        mdl, va, st = lanes[i % len(lanes)]
        if st is None:
            prev = pipe.submit(to_uint8(run_sampling(mdl, va, args, args.batch_size, generator, device)))
        else:
            with torch.cuda.stream(st):
                prev = pipe.submit(to_uint8(run_sampling(mdl, va, args, args.batch_size, generator, device)))  # the gather's side stream waits for THIS lane's stream
        drain(prev, i - depth)
    for i in range(max(0, iters - depth), iters):
        drain(pipe.flush(), i)
    if lanes[0][2] is not None:
        for _, _, st in lanes:
            torch.cuda.current_stream(device).wait_stream(st)
    return {"total": total, "iters": iters, "written": written, "gather_seconds": pipe.gather_seconds, "lanes": len(lanes)}


def main(argv=None, hooks=None):
    """torchrun entry point.  ``hooks`` (tests only) replaces the device-bound pieces for the CPU / gloo rehearsal of the multi-rank
    control flow: {"backend", "device", "build_models", "to_uint8", "save", "extractor"}; the product path (hooks=None) is RCCL + HIP only."""
    from .autoencoder import images_to_uint8

    hooks = hooks or {}
    parser = build_parser()
    args = parser.parse_args(argv)
    torch.set_grad_enabled(False)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(hooks.get("device") or f"cuda:{local_rank}")
    if device.type == "cuda":
        torch.cuda.set_device(device)
        dist.init_process_group(hooks.get("backend", "nccl"), device_id=device)  # "nccl" is RCCL on ROCm
    else:
        dist.init_process_group(hooks["backend"])
    rank, world = dist.get_rank(), dist.get_world_size()
    model, vae = hooks.get("build_models", build_models)(args, device)  # un-offset seed: --random_weights gives every rank the SAME model
    args.seed = args.seed + rank  # reference :28-32 (rank-local noise / label stream)
    # build_models seeded the GLOBAL generators with the un-offset seed (same random weights everywhere); the reference seeds torch / cuda with
    # seed + rank, and everything that draws from the global RNG afterwards (--generator dummy, latent_dist.sample()) must differ per rank
    torch.manual_seed(args.seed)
    if device.type == "cuda":
        torch.cuda.manual_seed_all(args.seed)
    generator = get_generator(args.generator, args.n_sample, args.seed)
    save_dir = args.save_dir or "./generated_samples/{}/exp{}_ep{}_m{}".format(args.dataset, args.exp, args.epoch_id, args.method)
    save = hooks.get("save") or (lambda block, start: save_images_uint8(block, save_dir, start))
    res = run(args, model, vae, generator, rank, world, device, hooks.get("to_uint8", images_to_uint8), save, extractor=hooks.get("extractor"))
    dist.barrier()
    if rank == 0 and "fid_stats" in res:  # reference :147-153: rank 0 reports and logs the score
        from .fid import fid_against_reference_stats

        res["fid"] = fid_against_reference_stats(res["fid_stats"], args.real_img_dir)
        print("FID = {}".format(res["fid"]))
        if getattr(args, "output_log", ""):
            with open(args.output_log, "a") as f:
                f.write("Epoch = {}, FID = {}\n".format(args.epoch_id, res["fid"]))
    if rank == 0:
        print(f"sampled {res['total']} images on {world} GPUs" + (f" -> {save_dir}" if args.compute_fid and "fid_stats" not in res else ""))
    if dist.is_initialized():
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
