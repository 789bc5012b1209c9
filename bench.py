"""Headline benchmark: images/sec of the LFM sampling hot path on MI355X.

Workload (BASELINE.json configs[1]): DiT-L/2 on 4x32x32 f8 latents (256x256 images), batch 64 per GPU, 50 Euler NFE on
the torchdiffeq grid (step_size 0.02, the path the reference actually runs) + f8 VAE decode + uint8 NHWC conversion.
Synthetic data: seeded random-init weights of the real architectures (de-zeroed), seeded Gaussian latents resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full batch through the hot path.  Ranks shard batches (weak scaling, no data-path collective during the
solve); with N > 1 every batch ends with one RCCL all_gather_into_tensor of the uint8 images.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense fp16/bf16 (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    p.add_argument("--model", type=str, default="DiT-L/2")
    p.add_argument("--nfe", type=int, default=50)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    return p.parse_args()


def usable_cores():
    """Host cores this process may actually use (affinity mask and cgroup quota; os.cpu_count() over-reports in containers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def cpu_baseline(model_name, nfe, budget_s=25.0):
    """The oracle (CPU restatement of the reference path, fp32 torch) timed on the host cores on a BOUNDED sample: a few
    velocity evaluations of the DiT at N=2 plus one VAE decode of one image, scaled to `nfe` evaluations + decode per image."""
    from oracle import dit_ref, vae_ref

    threads = min(usable_cores(), 32)  # torch-CPU GEMMs of this size stop scaling (and oversubscribe) beyond ~32 threads
    torch.set_num_threads(threads)
    cfg = dit_ref.DiTCfg.named(model_name, num_classes=1, label_dropout=0.0)
    sd = dit_ref.make_dit_state(cfg, seed=0)
    x = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(42))
    t0 = time.perf_counter()
    dit_ref.dit_forward(sd, cfg, torch.tensor(0.9), x)
    first = time.perf_counter() - t0
    reps = 0
    t_eval = first / x.shape[0]
    if first < budget_s / 6:  # fast enough: discard the first (warm-up) call and average a few more
        reps = max(1, min(4, int(budget_s / 3 / first)))
        t0 = time.perf_counter()
        for i in range(reps):
            dit_ref.dit_forward(sd, cfg, torch.tensor(0.5 + 0.1 * i), x)
        t_eval = (time.perf_counter() - t0) / reps / x.shape[0]
    vsd = vae_ref.make_vae_state(seed=0)
    R = 32 if first < budget_s / 6 else 16
    z = torch.randn(1, 4, R, R, generator=torch.Generator().manual_seed(1))
    t0 = time.perf_counter()
    vae_ref.vae_decode(vsd, z)
    t_dec = (time.perf_counter() - t0) * (32 // R) ** 2
    ips = 1.0 / (nfe * t_eval + t_dec)
    return {"value": ips, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle (fp32 torch-CPU restatement of the reference path), {threads} threads: {max(reps, 1)} {model_name} velocity "
                      f"evals at N=2 ({t_eval:.3f} s/img/eval) + 1 VAE decode at {8 * R}x{8 * R} ({t_dec:.2f} s/img at 256x256), scaled to "
                      f"{nfe} NFE + decode per image"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    torch.set_grad_enabled(False)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from lfm_amd import hip
    from lfm_amd.autoencoder import AutoencoderKL, images_to_uint8
    from lfm_amd.models import DiT_models
    from lfm_amd.solvers import GraphedFixedGrid, torchdiffeq_euler_grid
    from lfm_amd.test_flow_latent import dezero_

    # ---- models (celeb256_dit.txt: --num_classes 1 --label_dropout 0.)
    torch.manual_seed(0)
    model = dezero_(DiT_models[a.model](img_resolution=32, in_channels=4, label_dropout=0.0, num_classes=1)).to(dev).eval()
    vae = AutoencoderKL.from_random(seed=0).to(dev)
    B = a.batch
    h = 1.0 / a.nfe
    ts, dts = torchdiffeq_euler_grid(h)
    assert dts.numel() == a.nfe
    solver = GraphedFixedGrid(model, B)
    solver.set_grid(ts, dts)
    x0 = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(42 + rank)).to(dev)  # resident in HBM
    gathered = torch.empty(world * B, 256, 256, 3, dtype=torch.uint8, device=dev) if world > 1 else None

    def step():
        lat = solver.run(x0)
        img = vae.decode(lat / 0.18215).sample
        u8 = images_to_uint8(img)
        if world > 1:
            dist.all_gather_into_tensor(gathered, u8)
        return u8

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(a.warmup, 0)):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    fence()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt)
    assert out.shape == (B, 256, 256, 3) and float(out[:4].float().std()) > 0
    ips = world * B * a.steps / el

    from oracle import dit_ref, vae_ref  # FLOP closed forms only

    f_img = a.nfe * dit_ref.dit_flops_per_image(dit_ref.DiTCfg.named(a.model, num_classes=1, label_dropout=0.0)) + vae_ref.vae_decode_flops(32)

    res = {
        "metric": "images/sec", "value": ips, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
        "data": "synthetic",
        "config": {"workload": f"{a.model} celeb256 f8 (4x32x32 latents), batch {B}/GPU, {a.nfe}-step Euler (torchdiffeq grid) + f8 VAE decode "
                               f"to 256x256 + uint8 NHWC" + (" + RCCL all-gather of images" if world > 1 else ""),
                   "per_gpu_batch": B, "global_batch": B * world, "nfe": a.nfe, "sharding": f"dp{world}"},
        "algorithmic_gflop_per_image": f_img / 1e9,
        "mfma_frac_whole_path": ips * f_img / (MFMA_PEAK_TFLOPS * 1e12 * world),
    }

    # split of one step (outside the timed region): solver-only and decode-only
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    lat = solver.run(x0)
    ev[1].record()
    images_to_uint8(vae.decode(lat / 0.18215).sample)
    ev[2].record()
    torch.cuda.synchronize()
    res["split_ms"] = {"solver": ev[0].elapsed_time(ev[1]), "vae_decode_u8": ev[1].elapsed_time(ev[2])}

    if rank == 0 and not a.no_roofline:
        # dominant kernel: the fc1 MFMA GEMM + GELU epilogue (gemm256q_tn_kernel<ASrcRowMajor, EpiBiasGeluF16>, 25 % of the step).
        # Timed LIVE and IN SITU: eager forwards of the real model on the solver's latents with one HIP event pair recorded
        # around every block's fc1 launch on the launching stream (lfm_profile_fc1; a captured graph cannot be bracketed).
        import ctypes as C

        D, H, M = model.hidden_size, model.mlp_hidden, B * 256
        L = hip.lib()
        tmid = torch.tensor(0.5, device=dev)
        model(tmid, lat)  # warm
        durs = []
        for _ in range(3):
            hip.check(L.lfm_profile_fc1(1), "lfm_profile_fc1")
            model(tmid, lat)
            buf = (C.c_float * 64)()
            n = L.lfm_profile_fc1_read(buf, 64)
            durs += [buf[i] for i in range(n)]
        L.lfm_profile_fc1(0)
        dur = sum(durs) / len(durs) * 1e-3
        ach = 2.0 * M * H * D / dur / 1e12
        # HBM traffic per launch from PMC (profiles/r01_e_gemm_pmc.txt, separate FETCH_SIZE / WRITE_SIZE passes): FETCH_SIZE
        # 100694.4 KB raw -> x2 (gfx950 correction of MI355X_MICROARCH.md for 16-B-per-lane streams) + WRITE_SIZE 135168 KB;
        # measured at this shape on the shipped kernel.
        traffic = (2 * 100694.4 + 135168.0) * 1024 if (M, H, D) == (16384, 4096, 1024) else None
        res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                           "traffic": traffic, "traffic_unit": "bytes/launch (HBM, rocprofv3 PMC, separate passes)",
                           "algorithmic_bytes": 2.0 * (M * D + H * D + M * H), "algorithmic_flop": 2.0 * M * H * D,
                           "kernel": "gemm256q_tn_kernel<ASrcRowMajor,EpiBiasGeluF16> (DiT fc1 + GELU)",
                           "shape": {"M": M, "N": H, "K": D}, "avg_launch_us": dur * 1e6, "launches_timed": len(durs)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(a.model, a.nfe)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
