"""Round-2 probe: interleaved A/B of GEMM debug flags on the four DiT-L/2 block GEMMs (M = 16384) with their real epilogues,
whole-forward in-situ A/B, attention and LN-modulate alone.  usage: python tools/r2_probe.py name=flag [name=flag ...]"""
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
def _sel(v):  # "flags" or "kernel:flags"
    k, _, f = v.rpartition(":")
    return (int(k) if k else 0) | (int(f) << 4)
variants = [("default", 0)] + [(a.split("=")[0], _sel(a.split("=")[1])) for a in sys.argv[1:] if "=" in a]
SKIP_FWD = "nofwd" in sys.argv
M = 16384
for name, N, K, epi in [("qkv", 3072, 1024, "qkv"), ("proj", 1024, 1024, 3), ("fc1", 4096, 1024, 1), ("fc2", 1024, 4096, 3)]:
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 3 else torch.float16); gate = torch.randn(M // 256, N, device=dev)
    if epi == "qkv":
        D = N // 3
        Q = torch.empty(M, D, device=dev, dtype=torch.float16); Kk = torch.empty_like(Q); Vt = torch.empty(M // 256, D // 64, 64, 256, device=dev, dtype=torch.float16)
        fn = lambda: hip.check(hip.lib().lfm_gemm_qkv_f16(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(Q), hip.ptr(Kk), hip.ptr(Vt), M, D, K, hip.ptr(b), 64, 256, hip.stream_ptr()), "qkv")
    else:
        fn = lambda: hip.gemm_f16(A, W, b, epilogue=epi, out=out, gate=gate, gate_stride=N, tokens=256)
    res = {n: [] for n, _ in variants}
    for rnd in range(5):
        for vn, sel in variants:
            hip.gemm_select(sel); res[vn].append(timeit(fn))
    for vn, _ in variants:
        ms = statistics.median(res[vn])
        print(f"{name:5s} N={N} K={K} {vn:14s}: median {ms*1e3:7.1f} us ({2*M*N*K/ms/1e9:5.0f} TF)  min {min(res[vn])*1e3:7.1f}", flush=True)
hip.gemm_select(0)
if SKIP_FWD: sys.exit(0)
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
x = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
res = {}
for rnd in range(4):
    for vn, sel in variants:
        hip.gemm_select(sel); res.setdefault(vn, []).append(timeit(lambda: m(t, x), n=6, warm=2))
hip.gemm_select(0)
for k, v in res.items(): print(f"forward DiT-L/2 b64 {k:14s}: median {statistics.median(v):7.3f} ms  min {min(v):7.3f} ms  => {64/(50*statistics.median(v)/1e3):.1f} img/s at 50 NFE (no VAE)", flush=True)
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
for vn, sel in variants:
    hip.gemm_select(sel)
    ms = statistics.median([timeit(lambda: hip.dit_attention(Q, K, Vt, Bh, heads, T), n=20) for _ in range(3)])
    print(f"attention b={Bh} h={heads} T={T} {vn:14s}: {ms*1e3:.1f} us  {4*Bh*heads*T*T*64/ms/1e9:.0f} TFLOP/s  {4*Bh*T*1024*2/ms/1e6:.0f} GB/s", flush=True)
hip.gemm_select(0)
X = torch.randn(Bh * T, 1024, device=dev); sh = torch.randn(1, 1024, device=dev); sc = torch.randn(1, 1024, device=dev)
for vn, sel in variants:
    hip.gemm_select(sel)
    ms = statistics.median([timeit(lambda: hip.ln_modulate(X, sh, sc, T, 0), n=20) for _ in range(3)])
    print(f"ln_modulate M={Bh*T} D=1024 {vn:14s}: {ms*1e3:.1f} us  {Bh*T*1024*6/ms/1e6:.0f} GB/s", flush=True)
hip.gemm_select(0)
