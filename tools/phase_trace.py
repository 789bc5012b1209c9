"""Per-phase timeline of the quadrant-phased GEMM (v3): six s_memtime stamps per phase of one workgroup (waves 0 and 4)."""
import sys, ctypes as C, statistics, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (256, 256, 4096)
A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half()
b = torch.zeros(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=torch.float16)
hip.gemm_select(3)
for _ in range(3): hip.gemm_f16(A, W, b, epilogue=0, out=out)
hip.gemm_select(3 | (2 << 4)); hip.gemm_f16(A, W, b, epilogue=0, out=out); torch.cuda.synchronize(); hip.gemm_select(0)
nk = K // 64; n = min(2048, nk * 24)
buf = (C.c_ulonglong * (2 * n))()
hip.check(hip.lib().lfm_gemm_trace_read(buf, n), "trace")
names = ["reads issued", "DMAs issued", "vmcnt wait", "barrier 1", "MFMAs issued", "flush+barrier 2"]
for g in range(2):
    t = [buf[g * n + i] for i in range(n)]
    seg = {(p, k): [] for p in range(4) for k in range(6)}
    for kt in range(2, min(nk, n // 24) - 2):
        for p in range(4):
            i = (kt * 4 + p) * 6
            for k in range(5): seg[(p, k)].append(t[i + k + 1] - t[i + k])
            seg[(p, 5)].append(t[i + 6] - t[i + 5])
    print(f"group {g}: s_memtime ticks (median over the steady-state K-tiles)")
    print("        " + "".join(f"{x:>17s}" for x in names) + "            total")
    tot = 0
    for p in range(4):
        v = [statistics.median(seg[(p, k)]) for k in range(6)]; tot += sum(v)
        print(f"   P{p + 1}   " + "".join(f"{x:17.0f}" for x in v) + f"   {sum(v):14.0f}")
    print(f"   K-tile total {tot:.0f} ticks (MFMA work alone: 2 x 1024 = 2048 cycles per SIMD)")
