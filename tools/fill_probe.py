"""How much of the GEMM time is data (power / DVFS) dependent?  Same kernel, different operand fills."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
fills = {
    "zeros": lambda *s: torch.zeros(*s, device=dev),
    "ones": lambda *s: torch.ones(*s, device=dev),
    "uniform(-1,1)": lambda *s: torch.rand(*s, device=dev) * 2 - 1,
    "uniform(0,1)": lambda *s: torch.rand(*s, device=dev),
    "randn*0.5 / randn*0.03": None,
    "small ints": lambda *s: torch.randint(-4, 5, s, device=dev).float(),
}
for M, N, K in [(4096, 4096, 4096), (16384, 4096, 1024)]:
    for sel in (2, 3):
        for name, f in fills.items():
            if f is None:
                A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half()
            else:
                A = f(M, K).half(); W = f(N, K).half()
            b = torch.zeros(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=torch.float16)
            res = []
            for flags in (0, 4):
                hip.check(hip.lib().lfm_gemm_select(sel | (flags << 4)), 'select')
                ms = timeit(lambda: hip.gemm_f16(A, W, b, epilogue=0, out=out))
                res.append(2 * M * N * K / ms / 1e9)
            print(f"M={M} N={N} K={K} v{sel} {name:24s}: {res[0]:6.0f} TF   no-epilogue {res[1]:6.0f} TF", flush=True)
hip.check(hip.lib().lfm_gemm_select(0), 'select')
