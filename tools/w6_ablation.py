"""(M) Main-loop ablations of the one-wave-per-SIMD GEMM (kernel 6, csrc/gemm256w_kernel.h) on the fc1 / fc2 shapes, epilogue off: full loop, no LDS-DMA,
no fragment reads, neither, neither and no barrier; the eight-wave kernel's full loop and bare stream beside them.  Interleaved medians.
usage: LFM_MEASURE=1 python -m lfm_amd._build && python tools/w6_ablation.py"""
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
for M, N, K in ((16384, 4096, 1024), (16384, 1024, 4096)):
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float16)
    f = lambda k, abl: k | ((4 | (abl << 21)) << 4)
    variants = [("v5 no epilogue", f(5, 0)), ("v5 MFMA + barriers only", f(5, 3)), ("v5 MFMA only", f(5, 5)),
                ("v6 no epilogue", f(6, 0)), ("v6 no DMA", f(6, 1)), ("v6 no reads", f(6, 2)), ("v6 MFMA + barrier only", f(6, 3)), ("v6 MFMA only", f(6, 4)),
                ("v6 DMA 2/group early", f(6, 8))]
    res = {n: [] for n, _ in variants}
    for rnd in range(5):
        for name, sel in variants:
            hip.gemm_select(sel); res[name].append(timeit(lambda: hip.gemm_f16(A, W, b, epilogue=1, out=out)))
    hip.gemm_select(0)
    for name, _ in variants:
        us = statistics.median(res[name])
        print(f"M={M} N={N} K={K} {name:28s}: {us:7.1f} us  ({2.0 * M * N * K / us / 1e6:6.0f} TF)", flush=True)
