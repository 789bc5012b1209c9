"""Scratch timing on the GPU box: GEMM TFLOP/s at the DiT-L shapes and one DiT-L/2 forward."""
import sys
import time

import torch

sys.path.insert(0, ".")
from lfm_amd import hip
from lfm_amd.models import DiT_models

dev = torch.device("cuda:0")


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


M = 16384
for sel in (1, 2):
  hip.lib().lfm_gemm_select(sel)
  print("---- gemm kernel", {1: "v1 128x128", 2: "v2 256x256 ping-pong"}[sel])
  for N, K, epi in [(3072, 1024, 0), (1024, 1024, 3), (4096, 1024, 1), (1024, 4096, 3), (8192, 8192, 0)]:
    MM = 8192 if N == 8192 else M
    A = (torch.randn(MM, K, device=dev) * 0.5).half()
    W = (torch.randn(N, K, device=dev) * 0.03).half()
    b = torch.randn(N, device=dev)
    out = torch.zeros(MM, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.float16)
    gate = torch.randn(MM // 256, N, device=dev)
    ms = timeit(lambda: hip.gemm_f16(A, W, b, epilogue=epi, out=out, gate=gate, gate_stride=N, tokens=256))
    print(f"gemm M={MM} N={N} K={K} epi={epi}: {ms*1e3:.1f} us  {2*MM*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
hip.lib().lfm_gemm_select(0)

for name, bs in [("DiT-B/2", 64), ("DiT-L/2", 64)]:
    m = DiT_models[name](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
    for p in m.parameters():
        if not bool(p.any()):
            torch.nn.init.normal_(p, std=0.02)
    m = m.to(dev).eval()
    x = torch.randn(bs, 4, 32, 32, device=dev)
    t = torch.tensor(0.5, device=dev)
    ms = timeit(lambda: m(t, x), n=5, warm=2)
    fl = {"DiT-B/2": 46.0e9, "DiT-L/2": 161.4e9}[name] * bs
    print(f"{name} bs={bs} forward: {ms:.2f} ms  {fl/ms/1e9:.0f} TFLOP/s  => {bs/(50*ms/1e3):.1f} img/s at 50 NFE (no VAE)", flush=True)

# attention alone (DiT-L shape)
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
for nm, fl in (("4 waves x 64 q", 0), ("8 waves x 32 q", 256 << 4)):
    hip.lib().lfm_gemm_select(fl)
    ms = timeit(lambda: hip.dit_attention(Q, K, Vt, Bh, heads, T), n=20)
    print(f"attention [{nm}] b={Bh} h={heads} T={T}: {ms*1e3:.1f} us  {4*Bh*heads*T*T*64/ms/1e9:.0f} TFLOP/s")
hip.lib().lfm_gemm_select(0)
X = torch.randn(Bh * T, 1024, device=dev); sh = torch.randn(1, 1024, device=dev); sc = torch.randn(1, 1024, device=dev)
ms = timeit(lambda: hip.ln_modulate(X, sh, sc, T, 0), n=20)
print(f"ln_modulate M={Bh*T} D=1024: {ms*1e3:.1f} us  {Bh*T*1024*6/ms/1e6:.0f} GB/s")
