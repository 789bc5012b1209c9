"""Interleaved A/B of the GEMM kernels on the DiT shapes with their real epilogues (median of rounds)."""
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
variants = [("v3 GM=4", 3), ("v3 GM=8", 3 | (32 << 4)), ("v3 GM=2", 3 | (64 << 4))]
for M, N, K in [(16384, 4096, 1024), (512, 768, 4096), (256, 256, 64), (256, 256, 128), (256, 512, 192), (1024, 256, 256), (300, 260, 320), (4096, 1024, 576), (256, 256, 32), (512, 256, 96), (256, 512, 160)]:
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    o1 = torch.zeros(M, N, device=dev, dtype=torch.float16)
    hip.check(hip.lib().lfm_gemm_select(2), 'select'); hip.gemm_f16(A, W, b, epilogue=0, out=o1)
    for sel in ():
        worst = 0.0
        for rep in range(5):
            o2 = torch.zeros_like(o1)
            hip.gemm_select(sel); hip.gemm_f16(A, W, b, epilogue=0, out=o2)
            torch.cuda.synchronize()
            worst = max(worst, (o1.float() - o2.float()).abs().max().item())
        print("check", M, N, K, "sel", sel, "max diff vs v2", worst, flush=True)
for M, N, K, epi in [(16384, 4096, 1024, 1), (16384, 4096, 1024, 0), (16384, 1024, 4096, 3), (16384, 3072, 1024, 0), (16384, 1024, 1024, 3)]:
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.float16); gate = torch.randn(M // 256, N, device=dev)
    res = {n: [] for n, _ in variants}
    for rnd in range(5):
        for name, sel in variants:
            hip.check(hip.lib().lfm_gemm_select(sel), 'select')
            res[name].append(timeit(lambda: hip.gemm_f16(A, W, b, epilogue=epi, out=out, gate=gate, gate_stride=N, tokens=256)))
    for name, _ in variants:
        ms = statistics.median(res[name])
        print(f"M={M} N={N} K={K} epi={epi} {name:10s}: median {ms*1e3:7.1f} us ({2*M*N*K/ms/1e9:5.0f} TF)  min {min(res[name])*1e3:7.1f}  max {max(res[name])*1e3:7.1f}", flush=True)
hip.check(hip.lib().lfm_gemm_select(0), 'select')
