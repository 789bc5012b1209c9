"""Phase split of the fused QKV + attention kernel on an LFM_MEASURE build (LFM_HIP_LIBRARY=...), to be run under rocprofv3 --kernel-trace --stats: eager DiT
forwards with the kernel's measurement flags (lfm_gemm_select flags 33554432: no key loop, 67108864: one K-tile only); read the kernel's own duration from the
stats (whole forwards mislead: garbage activations are cheap operands for the kernels downstream).  usage: fused_qkv_phases.py FLAGS [model] [batch] [fused 0/1]"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
name = sys.argv[2] if len(sys.argv) > 2 else "DiT-L/2"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
fused = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
m = DiT_models[name](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
x = torch.randn(batch, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
hip.set_option(hip.OPT_FUSED_QKV_ATTENTION, fused)
hip.gemm_select(flags << 4)
for _ in range(6): m(t, x)
torch.cuda.synchronize(); print("done")
