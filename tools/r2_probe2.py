"""Round-2 probe 2: (a) attention with the token-major head slices of the model (row stride 2 KiB) vs a head-contiguous layout (heads = 1:
the same kernel, every K / Q row adjacent to the next) -- is the strided 128-B access what holds the kernel at 3.3 TB/s?  (b) head_dim 72
attention (DiT-XL) and 16-token attention; (c) DiT-XL/2 and DiT-L/2 forwards at batch 64.  usage: python tools/r2_probe2.py"""
import sys, statistics, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
dev = torch.device("cuda:0")
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def att(Bh, heads, T, hd, tag):
    Q = torch.randn(Bh * T, heads * hd, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, hd, T, device=dev).half()
    ms = statistics.median([timeit(lambda: hip.dit_attention(Q, K, Vt, Bh, heads, T, head_dim=hd), n=20) for _ in range(5)])
    print(f"attention {tag:28s} b={Bh} h={heads} T={T} hd={hd}: {ms*1e3:6.1f} us  {4*Bh*heads*T*T*hd/ms/1e9:5.0f} TFLOP/s  {4*Bh*heads*T*hd*2/ms/1e6:5.0f} GB/s", flush=True)
for rnd in range(2):
    att(64, 16, 256, 64, "token-major (model layout)")
    att(1024, 1, 256, 64, "head-contiguous")
att(64, 16, 256, 72, "hd 72 (DiT-XL/2)")
att(64, 16, 64, 72, "hd 72, 64 tokens (XL/4)")
att(64, 16, 16, 64, "16 tokens (x/8)")
for name in ("DiT-L/2", "DiT-XL/2"):
    m = DiT_models[name](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
    for p in m.parameters():
        if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
    m = m.to(dev).eval()
    x = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
    ms = statistics.median([timeit(lambda: m(t, x), n=6, warm=2) for _ in range(4)])
    print(f"forward {name} b64: {ms:7.3f} ms", flush=True)
    del m
