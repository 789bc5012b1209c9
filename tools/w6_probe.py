"""(M) The one-wave-per-SIMD 256x256 GEMMs (csrc/gemm256w_kernel.h = kernel 6, its quadrant-phased successor gemm256x_kernel.h = kernel 7) against the eight-wave one (kernel 5): the four DiT-L/2 block shapes with their
real epilogue families, bit equality of the results, the main-loop ablations of kernel 6 (LFM_MEASURE build: no epilogue; no LDS-DMA / no fragment reads /
neither; DMA placement variant), and the whole DiT-L/2 batch-64 forward with LFM_OPT_GEMM_V6 off / on.  Interleaved medians.
usage: LFM_MEASURE=1 python -m lfm_amd._build && python tools/w6_probe.py"""
import statistics, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M = 16384
shapes = [("fc1  bias+gelu", 4096, 1024, 1), ("fc2  gate+resid", 1024, 4096, 3), ("proj gate+resid", 1024, 1024, 3), ("qkv  split", 3072, 1024, "qkv")]
for name, N, K, epi in shapes:
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    if epi == "qkv":
        run = lambda: hip.gemm_qkv_f16(A, W, b, 64, 256)
    elif epi == 3:
        X = torch.randn(M, N, device=dev); gate = torch.randn(M // 256, N, device=dev)
        run = lambda: hip.gemm_f16(A, W, b, epilogue=3, out=X, gate=gate, gate_stride=N, tokens=256)
    else:
        out = torch.zeros(M, N, device=dev, dtype=torch.float16)
        run = lambda: hip.gemm_f16(A, W, b, epilogue=1, out=out)
    # bit equality 5 vs 6 (fresh outputs)
    outs = {}
    for k in (5, 6):
        hip.gemm_select(k)
        if epi == "qkv": outs[k] = [o.clone() for o in run()]
        elif epi == 3:
            X0 = torch.ones(M, N, device=dev); outs[k] = [hip.gemm_f16(A, W, b, epilogue=3, out=X0, gate=gate, gate_stride=N, tokens=256).clone()]
        else: outs[k] = [run().clone()]
    hip.gemm_select(0)
    same = all(torch.equal(a, c) for a, c in zip(outs[5], outs[6]))
    variants = [("v5 full", 5), ("v6 full", 6), ("v5 no epilogue", 5 | (4 << 4)), ("v6 no epilogue", 6 | (4 << 4))]
    if epi == 1:
        variants += [("v6 no epi, no DMA", 6 | ((4 | (1 << 21)) << 4)), ("v6 no epi, no reads", 6 | ((4 | (2 << 21)) << 4)), ("v6 no epi, MFMA+barrier only", 6 | ((4 | (3 << 21)) << 4))]
    res = {n: [] for n, _ in variants}
    for rnd in range(5):
        for n, sel in variants:
            hip.gemm_select(sel); res[n].append(timeit(run))
    hip.gemm_select(0)
    print(f"--- {name}  M={M} N={N} K={K}   v5 == v6 bitwise: {same}", flush=True)
    for n, _ in variants:
        us = statistics.median(res[n])
        print(f"    {n:32s}: {us:7.1f} us  ({2.0 * M * N * K / us / 1e6:6.0f} TF)   min {min(res[n]):7.1f}", flush=True)

from lfm_amd.models import DiT_models
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
x = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
cfgs = [("v5", 0), ("v6", 1)]
o = {}
for name, v6 in cfgs:
    hip.set_option(hip.OPT_GEMM_V6, v6); o[name] = m(t, x).clone()
torch.cuda.synchronize()
for name, _ in cfgs[1:]:
    print(f"DiT-L/2 b64 forward, {name} vs v5: bitwise equal {torch.equal(o['v5'], o[name])}, rel-L2 {float((o['v5'] - o[name]).norm() / o['v5'].norm()):.2e}", flush=True)
res = {n: [] for n, _ in cfgs}
for rnd in range(5):
    for name, v6 in cfgs:
        hip.set_option(hip.OPT_GEMM_V6, v6); res[name].append(timeit(lambda: m(t, x), n=6, warm=2))
hip.set_option(hip.OPT_GEMM_V6, 0)
for name, _ in cfgs: print(f"forward DiT-L/2 b64 {name:28s}: median {statistics.median(res[name]) / 1e3:7.3f} ms  min {min(res[name]) / 1e3:7.3f} ms", flush=True)
