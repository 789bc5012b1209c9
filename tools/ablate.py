import sys, torch
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
for M, N, K in [(16384, 4096, 1024), (512, 768, 4096), (256, 256, 64), (256, 256, 128), (256, 512, 192), (1024, 256, 256), (300, 260, 320)]:
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    o1 = torch.zeros(M, N, device=dev, dtype=torch.float16); o2 = torch.zeros_like(o1)
    hip.check(hip.lib().lfm_gemm_select(2), 'select'); hip.gemm_f16(A, W, b, epilogue=0, out=o1)
    hip.check(hip.lib().lfm_gemm_select(3), 'select'); hip.gemm_f16(A, W, b, epilogue=0, out=o2)
    torch.cuda.synchronize()
    print("check", M, N, K, "max diff v2-v3", (o1.float() - o2.float()).abs().max().item(), flush=True)
for M, N, K, epi in [(16384, 4096, 1024, 1), (16384, 1024, 4096, 3), (16384, 3072, 1024, 0), (16384, 1024, 1024, 3), (8192, 8192, 8192, 0), (4096, 4096, 4096, 0)]:
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.float16); gate = torch.randn(M // 256, N, device=dev)
    for name, sel in [("v2", 2), ("v3", 3), ("v2", 2), ("v3", 3), ("v2 no-epi", 2 | (4 << 4)), ("v3 no-epi", 3 | (4 << 4))]:
        hip.check(hip.lib().lfm_gemm_select(sel), 'select')
        ms = timeit(lambda: hip.gemm_f16(A, W, b, epilogue=epi, out=out, gate=gate, gate_stride=N, tokens=256))
        print(f"M={M} N={N} K={K} epi={epi} {name:12s}: {ms*1e3:8.1f} us  ({2*M*N*K/ms/1e9:6.0f} TF-equiv)", flush=True)
hip.check(hip.lib().lfm_gemm_select(0), 'select')
