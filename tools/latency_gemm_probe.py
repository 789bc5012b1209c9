"""Where do the ~10 us of a batch-1 K = 1024 linear go?  The 64x64-tile (lfm_gemm_select 7) and all-rows x 16-columns (8) kernels alone, on ONE weight matrix
launched back to back (W stays in the L2 / Infinity Cache: the kernel's floor) and cycling through enough distinct matrices to exceed the 256-MB Infinity Cache
(every launch streams its W from HBM, as inside a DiT-L/2 evaluation with its 0.9 GB of weights).   usage: python tools/latency_gemm_probe.py"""
import sys, statistics, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
def timed(fn, n):  # n launches captured in one graph (no host gaps), replayed three times
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * n) * 1e3
for (M, N, K, epi, what) in ((64, 64, 64, 0, "one workgroup, one K-tile (the launch-to-launch floor of a captured graph)"), (256, 1024, 256, 2, "a proj K-slice set (4 K-tiles)"),
                          (256, 3072, 1024, 0, "qkv-shaped"), (256, 4096, 1024, 1, "fc1-shaped"), (256, 3072, 128, 0, "qkv columns, 2 K-tiles")):
    A = (torch.randn(M, K, device=dev) * 0.5).half()
    nW = 64  # 64 x 6-8 MB = 400-540 MB of weights
    Ws = [(torch.randn(N, K, device=dev) / 32).half() for _ in range(nW)]
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for k in (7, 8):
        hip.gemm_select(k)
        hot, cold = [], []
        for rnd in range(3):
            hot.append(timed(lambda i: hip.gemm_f16(A, Ws[0], bias, epilogue=epi, out=out), nW))
            cold.append(timed(lambda i: hip.gemm_f16(A, Ws[i % nW], bias, epilogue=epi, out=out), nW))
        print(f"{what} {M} x {N} x {K}, kernel {k}: same W back to back {statistics.median(hot):6.2f} us | W from HBM every launch {statistics.median(cold):6.2f} us "
              f"(64 launches per captured graph)", flush=True)
hip.gemm_select(0)
