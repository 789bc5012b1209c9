"""DiT-L/2 batch-1 forwards in a loop (target of rocprofv3 kernel stats for the latency mode).  usage: fwd_probe_b1.py [reps]"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd.models import DiT_models
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
x = torch.randn(1, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
for _ in range(reps): m(t, x)
torch.cuda.synchronize(); print("done")
