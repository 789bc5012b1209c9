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
for M, N, K, epi in [(16384, 4096, 1024, 1), (16384, 1024, 4096, 3), (16384, 3072, 1024, 0), (16384, 1024, 1024, 3), (8192, 8192, 8192, 0)][:4]:
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.float16); gate = torch.randn(M // 256, N, device=dev)
    for name, sel in [("v2 base", 2), ("dma-first", 2 | (16 << 4)), ("GM=8", 2 | (32 << 4)), ("GM=2", 2 | (64 << 4)), ("no-xcd-remap", 2 | (128 << 4)),
                      ("v2 base again", 2)]:
        hip.lib().lfm_gemm_select(sel)
        ms = timeit(lambda: hip.gemm_f16(A, W, b, epilogue=epi, out=out, gate=gate, gate_stride=N, tokens=256))
        print(f"M={M} N={N} K={K} epi={epi} {name:12s}: {ms*1e3:8.1f} us  ({2*M*N*K/ms/1e9:6.0f} TF-equiv)", flush=True)
hip.lib().lfm_gemm_select(0)
