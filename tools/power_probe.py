"""Is the 256x256 GEMM main loop structure-bound or power-bound?  Run it on a fraction of the CUs (fewer tiles than CUs):
per-CU throughput with most of the chip idle vs with the whole chip busy."""
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
K = 8192
for fill in ("randn", "zeros"):
    for M, N in [(256, 256), (2048, 2048), (4096, 2048), (4096, 4096)]:
        if fill == "randn":
            A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half()
        else:
            A = torch.zeros(M, K, device=dev).half(); W = torch.zeros(N, K, device=dev).half()
        b = torch.zeros(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=torch.float16)
        tiles = (M // 256) * (N // 256)
        res = {2: [], 3: [], 19: []}
        for rnd in range(3):
            for sel in (2, 3, 19):
                hip.check(hip.lib().lfm_gemm_select(sel), 'select')
                res[sel].append(timeit(lambda: hip.gemm_f16(A, W, b, epilogue=0, out=out)))
        line = f"{fill:6s} tiles={tiles:5d} (M={M} N={N} K={K})"
        for sel in (2, 3, 19):
            ms = statistics.median(res[sel]); tf = 2 * M * N * K / ms / 1e9
            waves = -(-tiles // 256)
            per_cu = tf / min(tiles, 256) * (tiles / (waves * min(tiles, 256)))  # TF per busy CU
            line += f" | sel {sel:2d}: {ms*1e3:7.1f} us {tf:7.1f} TF {tf / min(tiles, 256) / (2500 / 256) * 100:5.1f}%"
        print(line, flush=True)
hip.check(hip.lib().lfm_gemm_select(0), 'select')
