"""Round-2 probe 3 (next-round design input): (a) attention split into its phases -- memory only (stage K / V^T, fetch Q, store O rows; no
math) and compute only (no K / V^T fetch) -- against the full kernel: if memory + compute ~= full, the kernel is phase-serialised (every
workgroup of a round loads at the same time, then computes at the same time), not bound by either; (b) LN-modulate with 1 / 2 / 4 rows per wave."""
import sys, statistics, torch
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
    return s.elapsed_time(e) / n
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
X = torch.randn(Bh * T, 1024, device=dev); sh = torch.randn(1, 1024, device=dev); sc = torch.randn(1, 1024, device=dev)
for rnd in range(3):
    for name, fl in (("full", 0), ("memory phases only", 1 << 25), ("compute only", 2 << 25)):
        hip.gemm_select(fl << 4)
        ms = timeit(lambda: hip.dit_attention(Q, K, Vt, Bh, heads, T))
        print(f"attention {name:20s}: {ms*1e3:6.1f} us", flush=True)
    for name, fl in (("1 row / wave, DPP sums (ships)", 0), ("1 row / wave", 65536), ("2 rows / wave (round 1)", 524288), ("4 rows / wave", 262144)):
        hip.gemm_select(fl << 4)
        out = hip.ln_modulate(X, sh, sc, T, 0)
        if fl == 0: base = out
        else:  # same numbers up to the summation order: a few outputs may land on the neighbouring fp16 value
            d = (out.float() - base.float())
            rel = float(d.norm() / base.float().norm()); nflip = int((d != 0).sum())
            assert rel < 1e-4, (name, rel)
            if rnd == 0: print(f"   {name}: rel-L2 vs the shipping kernel {rel:.2e}, {nflip} of {out.numel()} outputs differ (max {float(d.abs().max()):.4f})")
        ms = timeit(lambda: hip.ln_modulate(X, sh, sc, T, 0))
        print(f"ln_modulate {name:22s}: {ms*1e3:6.1f} us  {Bh*T*1024*6/ms/1e6:5.0f} GB/s", flush=True)
hip.gemm_select(0)
