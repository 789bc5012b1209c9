"""Single 256x256 tile (one CU busy, idle chip => no power cap): which resource bounds the K loop?  v2 ablation flags."""
import sys, statistics, torch
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
    return s.elapsed_time(e) / n
for M, N, K in [(256, 256, 8192), (4096, 2048, 8192), (4096, 4096, 8192)]:
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half()
    b = torch.zeros(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=torch.float16)
    for name, sel in [("v2 full", 2), ("v2 no DMA (reads+MFMA)", 2 | (1 << 4)), ("v2 no MFMA (DMA+reads)", 2 | (2 << 4)), ("v2 neither (reads+barriers)", 2 | (3 << 4)),
                      ("v3 quadrant-phased", 3), ("v3 two-barrier schedule", 3 | (1 << 4)), ("v1 128x128", 1)]:
        hip.check(hip.lib().lfm_gemm_select(sel), 'select')
        ms = statistics.median(timeit(lambda: hip.gemm_f16(A, W, b, epilogue=0, out=out)) for _ in range(3))
        nkt = K // 64
        print(f"M={M} N={N} K={K} {name:30s}: {ms*1e3:8.1f} us = {ms*1e6/nkt:7.1f} ns per 64-deep K-tile", flush=True)
hip.check(hip.lib().lfm_gemm_select(0), 'select')
