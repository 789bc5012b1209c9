"""s_memtime stamps of the row-major epilogue of the 16x16x32 GEMM (fc1 + GELU shape): where do the ~7.5 us per tile go?
slots: 0 = K loop done; per 32-row block i: 1+4i scratch filled, 2+4i read back, 3+4i aux loads returned, 4+4i stores issued; 17 = done."""
import ctypes as C, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in (sys.argv[1:4] or (16384, 4096, 1024))]
A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
out = torch.zeros(M, N, device=dev, dtype=torch.float16)
hip.gemm_select(5 | (2 << 4))
for rep in range(3):
    hip.gemm_f16(A, W, b, epilogue=1, out=out)
    buf = (C.c_ulonglong * 64)()
    hip.check(hip.lib().lfm_gemm_trace_read(buf, 32), "trace")
    for g in range(2):
        t = [buf[g * 32 + i] for i in range(18)]
        d = [t[i] - t[0] for i in range(18)]
        print(f"rep {rep} group {g}: " + " | ".join(f"blk{i}: fill {d[1+4*i]-(d[4*i] if i else 0):5d} read {d[2+4*i]-d[1+4*i]:5d} aux {d[3+4*i]-d[2+4*i]:5d} gelu+store {d[4+4*i]-d[3+4*i]:5d}" for i in range(4)) + f" | total {d[17]} ticks", flush=True)
hip.gemm_select(0)
