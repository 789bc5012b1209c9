"""Attention kernel at 1024 tokens (DiT-x/2 on 64x64 latents: four key chunks through one LDS image), 8 images x 16 heads, hd 64 and 72.
usage: python tools/attn_1024_time.py"""
import statistics, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
def timeit(fn, n=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for hd in (64, 72):
    Bh, heads, T = 8, 16, 1024
    Q = torch.randn(Bh * T, heads * hd, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, hd, T, device=dev).half()
    ts = [timeit(lambda: hip.dit_attention(Q, K, Vt, Bh, heads, T, hd)) for _ in range(5)]
    fl = 4.0 * Bh * heads * T * T * hd
    print(f"attention {Bh} x {heads} x {T} x {hd}: median {statistics.median(ts):.1f} us  min {min(ts):.1f}  = {fl / statistics.median(ts) / 1e6:.0f} TFLOP/s")
