"""Run one GEMM shape repeatedly (for rocprofv3 --pmc passes).  usage: gemm_probe.py SEL M N K EPI [reps] [randn|zeros]"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
from lfm_amd import hip

sel, M, N, K, epi = [int(v) for v in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
hip.check(hip.lib().lfm_gemm_select(sel), 'select')
fill = sys.argv[7] if len(sys.argv) > 7 else "randn"
if fill == "zeros":
    A = torch.zeros(M, K, device=dev).half()
    W = torch.zeros(N, K, device=dev).half()
else:
    A = (torch.randn(M, K, device=dev) * 0.5).half()
    W = (torch.randn(N, K, device=dev) * 0.03).half()
b = torch.randn(N, device=dev)
out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.float16)
gate = torch.randn(max(M // 256, 1), N, device=dev)
for _ in range(reps):
    hip.gemm_f16(A, W, b, epilogue=epi, out=out, gate=gate, gate_stride=N, tokens=256)
torch.cuda.synchronize()
print("done")
