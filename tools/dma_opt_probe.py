"""(M) LDS-DMA addressing / placement variants of the 256x256 GEMMs (OPT bits, csrc/gemm256h_kernel.h): bit 0 = buffer-addressed LDS-DMA, bit 1 = the first DMA of
a LOAD part ahead of its fragment reads.  fc1 / fc2 shapes, bias + GELU epilogue and no epilogue, bit equality with the default, interleaved medians.
usage: LFM_MEASURE=1 python -m lfm_amd._build && LFM_MEASURE=1 python tools/dma_opt_probe.py"""
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
for M, N, K in ((16384, 4096, 1024), (16384, 1024, 4096), (16384, 1024, 1024)):
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float16)
    f = lambda k, opt, noepi: k | (((4 if noepi else 0) | (opt << 25)) << 4)
    variants = [(f"v{k} OPT {o}{' no epilogue' if ne else ''}", f(k, o, ne)) for ne in (0, 1) for k, o in ((5, 0), (5, 1), (5, 2), (5, 3), (6, 0), (6, 1))]
    hip.gemm_select(5); ref = hip.gemm_f16(A, W, b, epilogue=1).clone()
    same = {}
    for name, sel in variants[:6]:
        hip.gemm_select(sel); same[name] = bool(torch.equal(hip.gemm_f16(A, W, b, epilogue=1), ref))
    res = {n: [] for n, _ in variants}
    for rnd in range(5):
        for name, sel in variants:
            hip.gemm_select(sel); res[name].append(timeit(lambda: hip.gemm_f16(A, W, b, epilogue=1, out=out)))
    hip.gemm_select(0)
    for name, _ in variants:
        us = statistics.median(res[name])
        print(f"M={M} N={N} K={K} {name:24s}: {us:7.1f} us  ({2.0 * M * N * K / us / 1e6:6.0f} TF)  min {min(res[name]):7.1f}" + (f"   == default: {same[name]}" if name in same else ""), flush=True)
