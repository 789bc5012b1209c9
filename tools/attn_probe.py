"""Attention alone at the DiT-L/2 batch-64 shape (target of rocprofv3 --pmc passes).  usage: attn_probe.py [flags] [reps]"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
hip.gemm_select(flags << 4)
import os
if os.environ.get("LFM_ATT_STREAM") is not None: hip.set_option(hip.OPT_ATTENTION_STREAM, int(os.environ["LFM_ATT_STREAM"]))
for _ in range(reps): hip.dit_attention(Q, K, Vt, Bh, heads, T)
torch.cuda.synchronize(); print("done")
