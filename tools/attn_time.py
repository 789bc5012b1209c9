"""Attention kernel alone at the DiT-L/2 batch-64 shape: interleaved medians of the default (4 waves x 64 queries) and the narrow variant (8 waves x 32 queries,
select flag 256).  usage: python tools/attn_time.py"""
import statistics, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
def timeit(fn, n=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
res = {0: [], 256: []}
for rnd in range(5):
    for f in (0, 256):
        hip.gemm_select(f << 4); res[f].append(timeit(lambda: hip.dit_attention(Q, K, Vt, Bh, heads, T)))
hip.gemm_select(0)
for f in (0, 256): print(f"attention 64 x 16 x 256 x 64, flags {f}: median {statistics.median(res[f]):.1f} us  min {min(res[f]):.1f}")
