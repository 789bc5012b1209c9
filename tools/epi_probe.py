"""Is the fp16-output epilogue bound per CU or chip-wide?  fc1-shaped GEMM (N 4096, K 1024, GELU epilogue) with 64 / 128 / 256 / 1024 tiles
(one tile per CU on a quarter / half / all of the chip, then four per CU), with and without the epilogue (flag 4), kernel v5 forced."""
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
N, K = 4096, 1024
for epi in (1, 3):
  for M in (1024, 2048, 4096, 16384):
    NN = N if epi == 1 else 1024
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(NN, K, device=dev) * 0.03).half(); b = torch.randn(NN, device=dev)
    out = torch.zeros(M, NN, device=dev, dtype=torch.float32 if epi == 3 else torch.float16); gate = torch.randn(max(M // 256, 1), NN, device=dev)
    fn = lambda: hip.gemm_f16(A, W, b, epilogue=epi, out=out, gate=gate, gate_stride=NN, tokens=256)
    res = {}
    for rnd in range(5):
        for name, sel in (("full", 5), ("noepi", 5 | (4 << 4))):
            hip.gemm_select(sel); res.setdefault(name, []).append(timeit(fn))
    hip.gemm_select(0)
    f, ne = statistics.median(res["full"]) * 1e3, statistics.median(res["noepi"]) * 1e3
    tiles = (M // 256) * (NN // 256)
    byt = M * NN * (2 if epi == 1 else 8)
    print(f"epi={epi} M={M:6d} N={NN} tiles={tiles:5d}: full {f:7.1f} us, no-epilogue {ne:7.1f} us, epilogue {f-ne:6.1f} us = {byt/(f-ne)/1e6:5.2f} TB/s "
          f"({byt/tiles/1024:.0f} KiB per tile, {(f-ne)/max(1,tiles//256 if tiles>=256 else 1):.1f} us per tile-wave)", flush=True)
