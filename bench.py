"""Headline benchmark: images/sec of the LFM sampling hot path on MI355X.

Default workload (BASELINE.json configs[1]): DiT-L/2 on 4x32x32 f8 latents (256x256 images), batch 64 per GPU, 50 Euler NFE on
the torchdiffeq grid (step_size 0.02, the path the reference actually runs) + f8 VAE decode + uint8 NHWC conversion.
Synthetic data: seeded random-init weights of the real architectures (de-zeroed), seeded Gaussian latents; one "step" = one full batch
through the hot path, INCLUDING the host-to-device copy of its latents (64 x 16 KiB from pinned memory) -- everything else is resident.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5|6]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

--config selects another BASELINE.json configuration (same JSON contract, its own `roofline`):
  3  DiT-L/2 class-conditional, dopri5 rtol = atol = 1e-5, CFG 1.5, 64 live + 64 null rows, + decode
  4  DiT-B/2 imnet class-conditional, CFG 1.5, batch 256 (+256 null), 50-point Karras grid Heun with the reference's steps=40 quirk (88 NFE), + decode
  5  origin-ADM celeb512 (352 M parameters), batch 32, 64x64 latents, 50-step Euler + VAE decode at 512x512
  6  EDM-style ADM UNet (models/EDM.py DhariwalUNet) at the test_args/{ffhq,bed}_adm.txt size (406 M parameters), batch 64, 50-step Euler + VAE decode (round 4;
     not a BASELINE.json configuration: 3 of the reference's 11 arg files run this backbone)

Ranks shard batches (weak scaling, no data-path collective during the solve); with N > 1 every batch ends with one RCCL
all_gather_into_tensor of the uint8 images, issued on a side stream so that it overlaps the next batch's solve.
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
EDM_FFHQ_FLOP_PER_IMAGE = 73.053765632e9  # one DhariwalUNet evaluation at the ffhq_adm size: hook-counted on the reference module (tests/golden/edm_full.pt, oracle/make_golden.py::golden_edm_full)
CONFIG3_FIELD_GAIN = 100.0  # output-layer scale of config 3's synthetic field (see --field-gain)

# L2-miss traffic of the dominant GEMM of config 2 (fc1: M 16384 x N 4096 x K 1024, folded-LayerNorm + GELU epilogue) per launch comes from a TRACKED
# measurement file, profiles/fc1_traffic.json, written by tools/profile_round.sh from separate rocprofv3 PMC passes of the shipped kernel IN SITU
# (inside DiT-L/2 batch-64 forwards): FETCH_SIZE x 2 (gfx950 correction for 16-B-per-lane streams, MI355X_MICROARCH.md) + WRITE_SIZE, KiB.  These are
# fabric-side request counters: Infinity-Cache hits are INCLUDED (the whole working set of this GEMM fits the 256 MiB cache), so this is an upper bound
# on HBM bytes, not HBM bytes.  The file names the kernel symbol it was measured on; `traffic` is null unless that symbol is the one this bench times.
FC1_KERNEL_SYMBOL = "gemm256h_tn_kernel<ASrcRowMajor, EpiModGeluF16, false, 0, 1>"


def fc1_traffic():
    path = os.path.join(ROOT, "profiles", "fc1_traffic.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    if d.get("kernel") != FC1_KERNEL_SYMBOL or not os.path.exists(os.path.join(ROOT, d.get("source", "").split(" ")[0])):
        return None  # measured on another kernel generation (or its evidence file is gone): a stale constant is worse than null
    return d


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5, 6])
    p.add_argument("--batch", type=int, default=0, help="images per GPU per step (0 = the configuration's own)")
    p.add_argument("--model", type=str, default="", help="override the DiT of configs 2-4")
    p.add_argument("--nfe", type=int, default=50)
    p.add_argument("--field-gain", type=float, default=0.0,
                   help="scale of the DiT's output layer (0 = 1, the baseline-comparable field).  Config 3 with --field-gain 100 (CONFIG3_FIELD_GAIN) makes the seeded "
                        "random field stiff enough for dopri5 at 1e-5 to take >= 15 steps (92 NFE) -- the solver-loop stress variant of profiles/r03_config3_*")
    p.add_argument("--in-flight", type=int, default=0, choices=[0, 1, 2, 3, 4],
                   help="batches in flight per GPU: 2 = consecutive steps alternate between two HIP streams with their own scratch (same weights, bit-identical "
                        "per-batch results: tests/test_gpu_cosched.py); 0 = the configuration's default -- 2 for the fixed-grid configurations 2, 4, 5, 6 (a sampling "
                        "job's batches are independent, test_flow_latent_ddp.py:128-146; while one lane's workgroups sit in their HBM-bound epilogues or run a "
                        "small-map kernel that cannot fill the chip, the other lane's kernels use it), 1 for config 3 (dopri5 reads a value back per step).  Rounds 1-4 reported one batch in "
                        "flight: --in-flight 1 reproduces that line")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--stub", action="store_true", help="(tests) CPU / gloo rehearsal of the launcher + rank logic: a stub step instead of the HIP path")
    return p.parse_args()


def self_launch(a, argv=None):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: re-exec THIS script under torch.distributed.run, one rank per GPU
    (what /root/reference/bash_scripts/run_test_ddp.sh:15-34 does for the reference's _ddp script).  Rank 0's JSON line goes to our stdout."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *(sys.argv[1:] if argv is None else argv)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // a.gpus)))
    return subprocess.run(cmd, env=env).returncode


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
    """The oracle (CPU restatement of the reference path, fp32 torch) timed on the host cores on a BOUNDED sample: a few velocity
    evaluations of the DiT at N=2 plus VAE decodes of one image (1 warm-up + 3 timed), scaled to `nfe` evaluations + decode per image."""
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
    vae_ref.vae_decode(vsd, z)  # warm-up (thread pool, oneDNN primitive caches)
    warm = time.perf_counter() - t0
    vreps = 3 if warm < budget_s / 9 else 1
    t0 = time.perf_counter()
    for _ in range(vreps):
        vae_ref.vae_decode(vsd, z)
    t_dec = (time.perf_counter() - t0) / vreps * (32 // R) ** 2
    ips = 1.0 / (nfe * t_eval + t_dec)
    return {"value": ips, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle (fp32 torch-CPU restatement of the reference path), {threads} threads: {max(reps, 1)} {model_name} velocity "
                      f"evals at N=2 after a warm-up ({t_eval:.3f} s/img/eval) + {vreps} VAE decodes at {8 * R}x{8 * R} after a warm-up "
                      f"({t_dec:.2f} s/img at 256x256), scaled to {nfe} NFE + decode per image"}


# ------------------------------------------------------------------------------------------------ algorithmic FLOPs
def dit_flops_per_image(tokens, hidden, depth, patch_inputs):
    """2 x MACs of one DiT evaluation of one image: per block qkv + proj + fc1 + fc2 (12 T D^2), QK^T and PV (2 T^2 D), the adaLN rows (6 D^2);
    patch embed, final layer, timestep MLP (256 D + D^2) and final adaLN (2 D^2).  161.4 GFLOP for DiT-L/2 (checked against the oracle's
    count in tests/test_host_logic.py)."""
    T, D, L = tokens, hidden, depth
    return 2 * (L * (12 * T * D * D + 2 * T * T * D + 6 * D * D) + T * patch_inputs * D * 2 + 2 * D * D + 256 * D + D * D)


def vae_decode_flops(R):
    """2 x MACs of one kl-f8 decode at latent side R (622.2 GFLOP at R = 32): post_quant 1x1, conv_in, mid (2 resnets + 1-head attention),
    four up blocks of 3 resnets (512, 512, 512->256, 256->128 channels at R, 2R, 4R, 8R) with three upsampler convs, conv_out."""
    px = R * R
    mac = px * 4 * 4 + px * 4 * 512 * 9                       # post_quant_conv, conv_in
    mac += 2 * 2 * px * 512 * 512 * 9                         # mid resnets
    mac += 4 * px * 512 * 512 + 2 * px * px * 512             # mid attention: q, k, v, out + QK^T, PV
    side = R
    for cin, cout, up in ((512, 512, True), (512, 512, True), (512, 256, True), (256, 128, False)):
        p = side * side
        mac += p * (cin * cout * 9 + cout * cout * 9 + (cin * cout if cin != cout else 0))  # resnet 0 (+ 1x1 shortcut)
        mac += 2 * p * 2 * cout * cout * 9                                                  # resnets 1, 2
        if up:
            side *= 2
            mac += side * side * cout * cout * 9
    mac += side * side * 128 * 3 * 9                          # conv_out
    return 2 * mac


# ------------------------------------------------------------------------------------------------ workloads
def build_workload(a, dev, rank):
    """Returns dict(step_fn(x_dev) -> fp32 images, x_host (pinned), B, res, workload, flop_per_image, model, solve_fn, extra)."""
    from lfm_amd.autoencoder import AutoencoderKL
    from lfm_amd.models import DiT_models, create_network
    from lfm_amd.solvers import GraphedFixedGrid, odeint, torchdiffeq_euler_grid
    from lfm_amd.test_flow_latent import dezero_

    def dit_flops(m):
        return dit_flops_per_image(m.x_embedder.num_patches, m.hidden_size, m.depth, m.patch_size * m.patch_size * m.in_channels)

    vae = AutoencoderKL.from_random(seed=0).to(dev)
    g = torch.Generator().manual_seed(42 + rank)
    lane_solver = None  # every fixed-grid configuration defines it below (a solver with its own buffers / captured graphs on a given module); None = one lane only
    if a.config == 2:
        name = a.model or "DiT-L/2"
        B = a.batch or 64
        torch.manual_seed(0)
        model = dezero_(DiT_models[name](img_resolution=32, in_channels=4, label_dropout=0.0, num_classes=1)).to(dev).eval()
        ts, dts = torchdiffeq_euler_grid(1.0 / a.nfe)
        assert dts.numel() == a.nfe
        def lane_solver(mod):  # a fixed-grid solver (own buffers, own captured graphs) on `mod`: the model itself or a concurrency twin of it
            sv = GraphedFixedGrid(mod, B)
            sv.set_grid(ts, dts)
            return sv.run

        solve = lane_solver(model)
        f_model = a.nfe * dit_flops(model)
        wl = f"{name} celeb256 f8 (4x32x32 latents), batch {B}/GPU, {a.nfe}-step Euler (torchdiffeq grid) + f8 VAE decode to 256x256 + uint8 NHWC"
        x_shape, res, extra = (B, 4, 32, 32), 32, {"nfe": a.nfe}
    elif a.config in (3, 4):
        name = a.model or ("DiT-L/2" if a.config == 3 else "DiT-B/2")
        B = a.batch or (64 if a.config == 3 else 256)
        torch.manual_seed(0)
        model = dezero_(DiT_models[name](img_resolution=32, in_channels=4, label_dropout=0.1, num_classes=1000))
        gain = a.field_gain or 1.0  # config 3: --field-gain 100 makes the seeded random field stiff (92 NFE, rejected steps) -- opt-in since round 4, so that
        # the default line stays comparable with BASELINE.json configs[2] and with earlier rounds
        if gain != 1.0:
            model.final_layer.linear.weight.data.mul_(gain)
            model.final_layer.linear.bias.data.mul_(gain)
        model = model.to(dev).eval()
        y = torch.cat([torch.randint(0, 1000, (B,), generator=g), torch.full((B,), 1000)]).to(dev)
        per_eval = 2 * dit_flops(model)  # CFG: 2 rows per image
        extra = {}
        if a.config == 3:
            stats = {}

            def solve(x):
                stats.clear()
                xx = torch.cat([x, x], 0)
                return odeint(lambda t, v: model.forward_with_cfg(t, v, y, cfg_scale=1.5), xx, torch.tensor([1.0, 0.0], device=dev), method="dopri5",
                              rtol=1e-5, atol=1e-5, stats=stats)[-1][:B]

            extra["solver_stats"] = stats
            extra["field_gain"] = gain
            f_model = None  # NFE is data dependent: filled in after the run
            wl = f"{name} class-conditional, dopri5 rtol=atol=1e-5, CFG 1.5, batch {B}(+{B} null)/GPU + f8 VAE decode + uint8"
        else:
            from lfm_amd.sampler.karras_sample import karras_sample

            def lane_solver(mod):  # the fused Karras solver caches its buffers / graphs on the module object: a twin gets its own
                def run(x):
                    xx = torch.cat([x, x], 0)
                    return karras_sample(mod, xx, steps=50, model_kwargs=dict(y=y, cfg_scale=1.5), device=dev, clip_denoised=False, sigma_min=1e-5,
                                         sigma_max=1.0, s_tmin=0.0, s_tmax=1.0, s_churn=0.0, sampler="heun")[:B]

                return run

            solve = lane_solver(model)

            f_model = 88 * per_eval
            extra["nfe"] = 88
            wl = (f"{name} imnet class-conditional, CFG 1.5, batch {B}(+{B} null)/GPU, 50-point Karras grid Heun with the reference's steps=40 quirk "
                  "(88 NFE) + f8 VAE decode + uint8")
        extra["per_eval_flop"] = per_eval
        x_shape, res = (B, 4, 32, 32), 32
    elif a.config == 6:
        # EDM-style ADM UNet (models/EDM.py DhariwalUNet) at the size of test_args/{ffhq,bed}_adm.txt (USE_ORIGIN_ADM=false): 3 of the reference's 11 arg files run
        # this backbone.  The arg files use dopri5; the bench line keeps the 50-step Euler grid of the headline so the numbers compare across backbones.
        from argparse import Namespace

        B = a.batch or 64
        cfg = Namespace(use_origin_adm=False, layout=False, model_type="adm", image_size=256, f=8, num_in_channels=4, num_out_channels=4, nf=256,
                        num_res_blocks=2, attn_resolutions=(16, 8, 4), dropout=0.0, ch_mult=(1, 2, 3, 4), label_dim=0, label_dropout=0.0, num_classes=1)
        torch.manual_seed(0)
        model = dezero_(create_network(cfg)).to(dev).eval()
        ts, dts = torchdiffeq_euler_grid(1.0 / a.nfe)
        def lane_solver(mod):
            sv = GraphedFixedGrid(mod, B, resolution=32)
            sv.set_grid(ts, dts)
            return sv.run

        solve = lane_solver(model)
        f_model = a.nfe * EDM_FFHQ_FLOP_PER_IMAGE
        wl = (f"EDM-style ADM UNet ffhq_adm (DhariwalUNet, 406 M params), 4x32x32 latents, batch {B}/GPU, {a.nfe}-step Euler (torchdiffeq grid) + f8 VAE decode to "
              "256x256 + uint8 NHWC")
        x_shape, res, extra = (B, 4, 32, 32), 32, {"nfe": a.nfe}
    else:
        from argparse import Namespace

        B = a.batch or 32
        cfg = Namespace(use_origin_adm=True, layout=False, model_type="adm", image_size=512, f=8, num_in_channels=4, num_out_channels=4, nf=256,
                        num_res_blocks=2, attn_resolutions=(16, 8), dropout=0.0, ch_mult=(1, 2, 2, 2, 4), resamp_with_conv=True, num_classes=None,
                        num_heads=4, num_head_channels=-1, num_head_upsample=-1)
        torch.manual_seed(0)
        model = dezero_(create_network(cfg)).to(dev).eval()
        ts, dts = torchdiffeq_euler_grid(1.0 / a.nfe)
        def lane_solver(mod):
            sv = GraphedFixedGrid(mod, B, resolution=64)
            sv.set_grid(ts, dts)
            return sv.run

        solve = lane_solver(model)
        f_model = a.nfe * 189.7e9  # hook-counted on the reference module
        wl = f"origin-ADM celeb512 (352 M params), 4x64x64 latents, batch {B}/GPU, {a.nfe}-step Euler + f8 VAE decode to 512x512 + uint8 NHWC"
        x_shape, res, extra = (B, 4, 64, 64), 64, {"nfe": a.nfe}
    x_host = torch.randn(*x_shape, generator=g).pin_memory()
    return dict(model=model, vae=vae, solve=solve, lane_solver=lane_solver, B=B, res=res, x_host=x_host, workload=wl, f_model=f_model,
                f_vae=vae_decode_flops(res),
                extra=extra, name=(a.model or {2: "DiT-L/2", 3: "DiT-L/2", 4: "DiT-B/2", 5: "ADM-celeb512", 6: "EDM-ADM-ffhq"}[a.config]))


def block_time_two_lanes(model, lat, rows, dev, evals=6):
    """The north-star quantity under the DEFAULT mode (two batches in flight): wall time of `evals` eager evaluations on EACH of two streams (the model and
    a concurrency twin, launched side by side) / (2 evals depth) -- an UPPER bound on the time the chip spends per DiTBlock, since the evaluations' other
    kernels (conditioning copy, patch embedding, final layer: ~3 %) are inside it."""
    from lfm_amd.solvers import concurrency_twin

    twin = concurrency_twin(model)
    tmid = torch.tensor(0.5, device=dev)
    x = lat if lat.shape[0] == rows else torch.cat([lat, lat], 0)[:rows]
    y = torch.zeros(rows, dtype=torch.long, device=dev) if (model.num_classes and model.num_classes > 1) else None
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    best = None
    for rep in range(3):
        cur = torch.cuda.current_stream(dev)
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(cur)
        sa.wait_event(e0)
        sb.wait_event(e0)
        for _ in range(evals):  # interleaved enqueue: both streams always have work
            with torch.cuda.stream(sa):
                model(tmid, x, y)
            with torch.cuda.stream(sb):
                twin(tmid, x, y)
        e1.record(sa)
        e2.record(sb)
        torch.cuda.synchronize()
        ms = max(e0.elapsed_time(e1), e0.elapsed_time(e2))
        if rep and (best is None or ms < best):  # first repetition = warm-up (the twin's workspace)
            best = ms
    return best * 1e-3 / (2 * evals * model.depth)


def roofline_dit(model, lat, rows, dev):
    """Dominant kernel of the DiT configurations: the fc1 MFMA GEMM + GELU epilogue (gemm256h_tn_kernel<ASrcRowMajor, EpiModGeluF16> when the
    LayerNorm-modulate is folded into the GEMM epilogues -- every chip-filling batch -- else <.., EpiBiasGeluF16>).
    Timed LIVE and IN SITU: eager forwards of the real model on the solver's latents with one HIP event pair recorded around every
    block's fc1 launch on the launching stream (lfm_profile_fc1; a captured graph cannot be bracketed)."""
    import ctypes as C

    from lfm_amd import hip

    D, H, M = model.hidden_size, model.mlp_hidden, rows * 256
    L = hip.lib()
    tmid = torch.tensor(0.5, device=dev)
    x = lat if lat.shape[0] == rows else torch.cat([lat, lat], 0)[:rows]
    y = None
    if model.num_classes and model.num_classes > 1:
        y = torch.zeros(rows, dtype=torch.long, device=dev)
    model(tmid, x, y)  # warm
    durs = []
    for _ in range(3):
        hip.check(L.lfm_profile_fc1(1), "lfm_profile_fc1")
        model(tmid, x, y)
        buf = (C.c_float * 64)()
        n = L.lfm_profile_fc1_read(buf, 64)
        durs += [buf[i] for i in range(n)]
    # the whole block loop, no events between its kernels: the quantity north_star's ">= 40 % MFMA utilisation in the DiT block" is about
    blk = []
    for _ in range(3):
        hip.check(L.lfm_profile_fc1(2), "lfm_profile_fc1")
        model(tmid, x, y)
        buf = (C.c_float * 16)()
        n = L.lfm_profile_blocks_read(buf, 16)
        blk += [buf[i] for i in range(n)]
    L.lfm_profile_fc1(0)
    T = model.x_embedder.num_patches
    blk_flop = 2.0 * rows * (12.0 * T * D * D + 2.0 * T * T * D)  # qkv + proj + fc1 + fc2 + QK^T + PV of one block, all rows
    blk_s = sum(blk) / len(blk) * 1e-3 / model.depth
    block = {"bound": "mfma", "block_us": blk_s * 1e6, "algorithmic_flop": blk_flop, "achieved": blk_flop / blk_s / 1e12, "peak": MFMA_PEAK_TFLOPS,
             "unit": "TFLOP/s", "frac": blk_flop / blk_s / 1e12 / MFMA_PEAK_TFLOPS, "evaluations_timed": len(blk),
             "what": "one DiTBlock (qkv projection + attention core as ONE kernel where the folded path applies, proj GEMM, fc1 GEMM, fc2 GEMM with their fused epilogues) = the eager block loop of an evaluation "
                     "/ depth, HIP events on the launching stream, one batch in flight"}
    blk2 = block_time_two_lanes(model, lat, rows, dev)
    block["two_lanes"] = {"block_us_upper_bound": blk2 * 1e6, "achieved": blk_flop / blk2 / 1e12, "frac": blk_flop / blk2 / 1e12 / MFMA_PEAK_TFLOPS,
                          "what": "two batches in flight (the default): wall time of 6 eager evaluations on each of two streams / (12 x depth); includes the "
                                  "evaluations' non-block kernels (~3 %), so the block's own fraction is at least this"}
    dur = sum(durs) / len(durs) * 1e-3
    ach = 2.0 * M * H * D / dur / 1e12
    r = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS, "traffic": None,
         "algorithmic_bytes": 2.0 * (M * D + H * D + M * H), "algorithmic_flop": 2.0 * M * H * D,
         "kernel": FC1_KERNEL_SYMBOL + " (DiT fc1: folded-LayerNorm correction + bias + GELU epilogue; EpiBiasGeluF16 where the fold does not apply)",
         "shape": {"M": M, "N": H, "K": D}, "avg_launch_us": dur * 1e6,
         "launches_timed": len(durs)}
    tr = fc1_traffic()
    if (M, H, D) == (16384, 4096, 1024) and tr is not None:
        r["traffic"] = (2 * tr["fetch_kib"] + tr["write_kib"]) * 1024
        r["traffic_unit"] = "bytes/launch leaving the L2s (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes); Infinity-Cache hits included, so an upper bound on HBM bytes"
        r["traffic_source"] = tr["source"]
        r["traffic_kernel"] = tr["kernel"]
    return r, block


def roofline_adm(B, dev, side=64, ch=256):
    """Dominant kernel of config 5: the 3x3 implicit-GEMM convolution of the 64x64, 256-channel ResBlocks (M = B*4096 pixels, N = 256, K = 2304;
    8 of them per evaluation plus the 512-input ones of the up path).  Timed live with HIP events on the launching stream, standalone on
    random NHWC activations of the real shape."""
    from lfm_amd import hip

    N, H, W, Cin, Cout = B, side, side, ch, ch
    x = (torch.randn(N * H * W, Cin, device=dev) * 0.5).half()
    w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).half()
    b = torch.zeros(Cout, device=dev)
    out = torch.empty(N * H * W, Cout, device=dev, dtype=torch.float16)
    L = hip.lib()
    st = torch.cuda.current_stream(dev)

    def run():
        hip.check(L.lfm_conv3x3_f16(hip.ptr(x), hip.ptr(w), hip.ptr(b), None, hip.ptr(out), N, H, W, Cin, Cout, 0, hip.stream_ptr(dev)), "conv")

    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        run()
    e1.record(st)
    torch.cuda.synchronize()
    dur = e0.elapsed_time(e1) / 10 * 1e-3
    M, K = N * H * W, 9 * Cin
    ach = 2.0 * M * Cout * K / dur / 1e12
    return {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS, "traffic": None,
            "algorithmic_bytes": 2.0 * (M * Cin + Cout * K + M * Cout), "algorithmic_flop": 2.0 * M * Cout * K,
            "kernel": f"conv3x3_halo_kernel<EpiResidF16,0> (3x3 conv {ch}->{ch} at {side}x{side}, halo-tiled direct kernel)", "shape": {"M": M, "N": Cout, "K": K},
            "avg_launch_us": dur * 1e6, "launches_timed": 10}


def main_stub(a, world, rank):
    """CPU / gloo rehearsal (tests/test_multi_rank.py): the same launch contract, rank bookkeeping, gather pipeline, max-over-ranks timing and
    ONE JSON line from rank 0 as the real run, around a stub step (no model, no HIP).  Never a measurement."""
    from lfm_amd.test_flow_latent_ddp import GatherPipeline

    if world > 1:
        dist.init_process_group("gloo")
    B = a.batch or 4
    pipe = GatherPipeline(world, "cpu")
    g = torch.Generator().manual_seed(42 + rank)
    x = torch.randn(B, 4, 8, 8, generator=g)
    got = []

    def step():
        u8 = (x.clamp(-1, 1) * 127 + 128).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        prev = pipe.submit(u8) if world > 1 else None
        if prev is not None:
            got.append(prev)
        return u8

    def fence():
        last = pipe.flush()
        if last is not None:
            got.append(last)
        if world > 1:
            dist.barrier()

    for _ in range(max(a.warmup, 0)):
        step()
    fence()
    got.clear()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        el = max(float(t) for t in allt)
        assert len(got) == a.steps and all(b.shape[0] == B * world for b in got)
        per_rank = [{"rank": r, "clock_mhz_under_mfma_load": None, "images_per_sec": B * a.steps / float(t)} for r, t in enumerate(allt)]
    else:
        per_rank = [{"rank": 0, "clock_mhz_under_mfma_load": None, "images_per_sec": B * a.steps / el}]
    if rank == 0:
        print(json.dumps({"per_rank": per_rank, "metric": "images/sec", "value": world * B * a.steps / el, "unit": "images/sec", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "u8", "data": "stub", "config": {"workload": "STUB (launcher rehearsal, not a measurement)", "sharding": f"dp{world}"},
                          "rccl_world": dist.get_world_size() if world > 1 else 1}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a))  # the driver's `python bench.py --gpus N`: become the torchrun launcher
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.set_grad_enabled(False)
    if a.stub:
        return main_stub(a, world, rank)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from lfm_amd.autoencoder import images_to_uint8
    from lfm_amd.test_flow_latent_ddp import GatherPipeline

    w = build_workload(a, dev, rank)
    B, vae, solve = w["B"], w["vae"], w["solve"]
    S = 8 * w["res"]
    x_dev = torch.empty(w["x_host"].shape, device=dev)
    gather_ms = []

    lanes = None
    in_flight = a.in_flight or (2 if w["lane_solver"] is not None else 1)
    pipe = GatherPipeline(world, dev, depth=in_flight)  # a batch's gathered block is waited for `in_flight` submissions later: never behind a batch just enqueued
    if in_flight > 1:
        # two batches in flight: consecutive steps go to two HIP streams, each with its own solver buffers / captured graphs / workspaces on the SAME weights
        if w["lane_solver"] is None:
            raise SystemExit("--in-flight 2 is built for the fixed-grid configurations (2, 4, 5, 6)")
        from lfm_amd.solvers import concurrency_twin

        lanes = [(solve, vae, x_dev, torch.cuda.Stream(dev))]
        for _ in range(in_flight - 1):
            lanes.append((w["lane_solver"](concurrency_twin(w["model"])), concurrency_twin(vae), torch.empty_like(x_dev), torch.cuda.Stream(dev)))
        first = []
        for sol, va, xd, st in lanes:  # capture each lane's graphs and size its workspaces outside the counted steps
            st.wait_stream(torch.cuda.current_stream(dev))  # weights, packed operands and the per-grid tables were produced on the launching stream
            with torch.cuda.stream(st):
                xd.copy_(w["x_host"], non_blocking=True)
                first.append(images_to_uint8(va.decode(sol(xd) / 0.18215).sample))
        torch.cuda.synchronize()
        # same latents through both lanes: the SAME images, bit for bit (the co-scheduling non-determinism of rounds 4 is root-caused and fixed:
        # profiles/r05_cosched_root_cause.txt; tests/test_gpu_cosched.py)
        assert all(torch.equal(first[0], f) for f in first[1:]), "the lanes must produce identical images for the same latents"
    nstep = [0]

    def step():
        if lanes is not None:
            sol, va, xd, st = lanes[nstep[0] % len(lanes)]
            nstep[0] += 1
            with torch.cuda.stream(st):
                xd.copy_(w["x_host"], non_blocking=True)
                u8 = images_to_uint8(va.decode(sol(xd) / 0.18215).sample)
                if world > 1:
                    t0 = time.perf_counter()
                    pipe.submit(u8)  # the side stream waits for THIS lane's stream; the collective runs under the other lane's batch
                    gather_ms.append((time.perf_counter() - t0) * 1e3)
                return u8
        x_dev.copy_(w["x_host"], non_blocking=True)  # the batch's latents cross PCIe inside the timed region (1 MiB at batch 64)
        lat = solve(x_dev)
        img = vae.decode(lat / 0.18215).sample
        u8 = images_to_uint8(img)
        if world > 1:
            t0 = time.perf_counter()
            pipe.submit(u8)  # the collective runs on a side stream under the next batch's solve
            gather_ms.append((time.perf_counter() - t0) * 1e3)
        return u8

    def fence():
        while pipe.flush() is not None:
            pass
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(a.warmup, 0)):
        step()
    fence()
    gather_ms.clear()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    fence()
    el_local = time.perf_counter() - t0
    el = el_local
    rank_ips = [B * a.steps / el_local]
    if world > 1:
        tt = torch.tensor([el_local], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        el = max(float(t) for t in allt)
        rank_ips = [B * a.steps / float(t) for t in allt]
    assert out.shape == (B, S, S, 3) and float(out[:4].float().std()) > 0
    ips = world * B * a.steps / el

    f_model = w["f_model"]
    if f_model is None:  # dopri5: NFE of the last solve
        f_model = w["extra"]["solver_stats"]["nfe"] * w["extra"]["per_eval_flop"]
    f_img = f_model + w["f_vae"]
    cfg = {"workload": w["workload"] + (" + RCCL all-gather of images (side stream)" if world > 1 else ""), "baseline_config": a.config,
           "per_gpu_batch": B, "global_batch": B * world, "sharding": f"dp{world}", "h2d_of_latents": "inside the timed step"}
    cfg.update({k: v for k, v in w["extra"].items() if k in ("nfe", "field_gain")})
    if lanes is not None:
        cfg["batches_in_flight"] = len(lanes)
        cfg["workload"] += f"; {len(lanes)} batches in flight on {len(lanes)} HIP streams (same weights, own scratch)"
    if "solver_stats" in w["extra"]:
        cfg["dopri5"] = dict(w["extra"]["solver_stats"])
    res = {
        "metric": "images/sec", "value": ips, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
        "data": "synthetic", "config": cfg,
        "rccl_world": dist.get_world_size() if world > 1 else 1,
        "per_rank_images_per_sec": {"min": min(rank_ips), "max": max(rank_ips)},
        "allgather_host_ms_per_batch": (sum(gather_ms) / len(gather_ms)) if gather_ms else 0.0,
        "algorithmic_gflop_per_image": f_img / 1e9,
        "mfma_frac_whole_path": ips * f_img / (MFMA_PEAK_TFLOPS * 1e12 * world),
    }

    # split of one step (outside the timed region): solver-only and decode-only
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    lat = solve(x_dev)
    ev[1].record()
    images_to_uint8(vae.decode(lat / 0.18215).sample)
    ev[2].record()
    torch.cuda.synchronize()
    res["split_ms"] = {"solver": ev[0].elapsed_time(ev[1]), "vae_decode_u8": ev[1].elapsed_time(ev[2])}
    # what every rank ran at: the clock its GPU sustains under matrix load (boxes / GPUs differ by several percent under the power cap) and its own
    # solve / decode times -- so that a multi-GPU line explains its own spread
    from lfm_amd import hip as _hip

    mine = torch.tensor([_hip.effective_clock_mhz(dev), res["split_ms"]["solver"], res["split_ms"]["vae_decode_u8"], B * a.steps / el_local], device=dev,
                        dtype=torch.float64)
    per = [mine]
    if world > 1:
        per = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per, mine)
    res["per_rank"] = [{"rank": r, "clock_mhz_under_mfma_load": round(float(v[0]), 1), "solver_ms": round(float(v[1]), 3), "vae_decode_u8_ms": round(float(v[2]), 3),
                        "images_per_sec": round(float(v[3]), 3)} for r, v in enumerate(per)]
    res["clock_mhz_under_mfma_load"] = res["per_rank"][0]["clock_mhz_under_mfma_load"]

    if rank == 0 and not a.no_roofline:
        if a.config == 5:
            res["roofline"] = roofline_adm(B, dev)
        elif a.config == 6:  # the 32x32 level (256 channels: 10 of the evaluation's 3x3 convolutions run at this shape or its 512-input variant)
            res["roofline"] = roofline_adm(B, dev, side=32, ch=256)
        else:
            res["roofline"], res["roofline_block"] = roofline_dit(w["model"], lat, B if a.config == 2 else 2 * B, dev)
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.config == 2:
        res["cpu_baseline"] = cpu_baseline(w["name"], a.nfe)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
