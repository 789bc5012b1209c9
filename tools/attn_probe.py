import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
for _ in range(5): hip.dit_attention(Q, K, Vt, Bh, heads, T)
torch.cuda.synchronize(); print("done")
