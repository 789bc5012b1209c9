"""The fused QKV projection (M 16384, D 1024, head 64, 256 tokens) per GEMM kernel, interleaved A/B."""
import sys, statistics, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
M, D = 16384, 1024
A = (torch.randn(M, D, device=dev) * 0.5).half(); W = (torch.randn(3 * D, D, device=dev) * 0.03).half(); b = torch.randn(3 * D, device=dev)
out = torch.zeros(M, 3 * D, device=dev, dtype=torch.float16)
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
variants = [("v3 QKV", 3, True), ("v3 plain fp16 epilogue N=3072", 3, False), ("v3 QKV no-epi", 3 | (4 << 4), True), ("v2 QKV", 2, True)]
res = {v[0]: [] for v in variants}
for rnd in range(5):
    for name, sel, qkv in variants:
        hip.gemm_select(sel)
        if qkv: res[name].append(timeit(lambda: hip.gemm_qkv_f16(A, W, b, 64, 256)))
        else: res[name].append(timeit(lambda: hip.gemm_f16(A, W, b, epilogue=0, out=out)))
hip.gemm_select(0)
for k, v in res.items(): print(f"{k:36s}: median {statistics.median(v)*1e3:7.1f} us  min {min(v)*1e3:7.1f}")
