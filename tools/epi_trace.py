"""s_memtime stamps of the epilogue of the 16x16x32 GEMM (kernel 5, flag 2): where do the ~7.5 us per tile go?
slots: 0 = K loop done; per 32-row block i: 1+4i scratch filled, 2+4i read back, 3+4i aux loads returned, 4+4i stores issued; 17 = done.
(transposed V tiles of the QKV epilogue: only 4+4i -- after each of the four 32-column x 64-row passes -- and 17.)
    python tools/epi_trace.py            # fc1+GELU, bias-only fp16, gated fp32 residual (proj / fc2 shapes), QKV (a Q tile and a V^T tile)
"""
import ctypes as C, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
M = 16384


def show(tag, swapped=False):
    buf = (C.c_ulonglong * 64)()
    hip.check(hip.lib().lfm_gemm_trace_read(buf, 32), "trace")
    for g in range(2):
        t = [buf[g * 32 + i] for i in range(18)]
        d = [t[i] - t[0] for i in range(18)]
        if swapped:
            print(f"{tag} group {g}: " + " | ".join(f"pass{i}: fill {d[1+4*i]-(d[4*i] if i else 0):5d} read {d[2+4*i]-d[1+4*i]:5d} store {d[4+4*i]-d[2+4*i]:5d}"
                                                    for i in range(4)) + f" | total {d[17]} cycles", flush=True)
        else:
            print(f"{tag} group {g}: " + " | ".join(f"blk{i}: fill {d[1+4*i]-(d[4*i] if i else 0):5d} read {d[2+4*i]-d[1+4*i]:5d} aux {d[3+4*i]-d[2+4*i]:5d} "
                                                    f"math+store {d[4+4*i]-d[3+4*i]:5d}" for i in range(4)) + f" | total {d[17]} cycles", flush=True)


def run(tag, epi, N, K, col=0, flags=0):
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    hip.gemm_select(5 | ((2 | flags | (col << 21)) << 4))
    for rep in range(2):
        if epi == "qkv":
            hip.gemm_qkv_f16(A, W, b, 64, 256)
        else:
            out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 3 else torch.float16)
            gate = torch.randn(M // 256, N, device=dev)
            hip.gemm_f16(A, W, b, epilogue=epi, out=out, gate=gate, gate_stride=N, tokens=256)
        show(f"{tag} rep {rep}", swapped=(epi == "qkv" and col >= 8))
    hip.gemm_select(0)


run("fc1+GELU  (N 4096, K 1024)", 1, 4096, 1024)
run("bias fp16 (N 4096, K 1024)", 0, 4096, 1024)
run("gate+resid fp32, proj (N 1024, K 1024)", 3, 1024, 1024)
run("gate+resid fp32, fc2  (N 1024, K 4096)", 3, 1024, 4096)
run("qkv, Q tile  (N 3072, K 1024)", "qkv", 3072, 1024, 0)
run("qkv, V^T tile (N 3072, K 1024)", "qkv", 3072, 1024, 8)
run("qkv, V^T tile, stores skipped", "qkv", 3072, 1024, 8, flags=131072)
