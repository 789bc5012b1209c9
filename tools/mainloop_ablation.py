"""(M) Where does the 256x256 GEMM's main loop go?  fc1 / fc2 shapes with the epilogue switched off (flag 4) and the main-loop ablations of the
LFM_MEASURE build (flags n << 21: 1 = no LDS-DMA after the prologue, 2 = no fragment reads, 3 = neither, 4 = static priority, 5 = 3 without barriers, 6 = 3 with one barrier per K-tile), interleaved medians.
usage: LFM_MEASURE=1 python -m lfm_amd._build && python tools/mainloop_ablation.py"""
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
    variants = [("full kernel", 0), ("no epilogue", 4), ("no epilogue, static prio", 4 | (4 << 21)), ("no epilogue, no DMA", 4 | (1 << 21)),
                ("no epilogue, no fragment reads", 4 | (2 << 21)), ("no epilogue, MFMA + barriers only", 4 | (3 << 21)), ("no epilogue, MFMA only, no barriers", 4 | (5 << 21)), ("no epilogue, MFMA only, 1 barrier per K-tile", 4 | (6 << 21)), ("no epilogue, MFMA only, no barriers, PINNED acc", 4 | (7 << 21)),
                ("no epilogue, full loop, PINNED acc", 4 | (7 << 21) | (1 << 24)), ("full kernel, PINNED acc", (7 << 21) | (1 << 24))]
    res = {n: [] for n, _ in variants}
    for rnd in range(5):
        for name, fl in variants:
            hip.gemm_select(5 | (fl << 4)); res[name].append(timeit(lambda: hip.gemm_f16(A, W, b, epilogue=1, out=out)))
    hip.gemm_select(0)
    for name, _ in variants:
        us = statistics.median(res[name])
        print(f"M={M} N={N} K={K} {name:36s}: {us:7.1f} us  ({2.0 * M * N * K / us / 1e6:6.0f} TF)", flush=True)
