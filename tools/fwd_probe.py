"""DiT-L/2 batch-64 forwards in a loop (target of rocprofv3 passes: kernel stats, PMC).  usage: fwd_probe.py [reps] [fold 0/1]"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
x = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
if len(sys.argv) > 2: hip.set_option(hip.OPT_FOLD_LN, int(sys.argv[2]))
import os
if os.environ.get("LFM_ATT_STREAM") is not None: hip.set_option(hip.OPT_ATTENTION_STREAM, int(os.environ["LFM_ATT_STREAM"]))
for _ in range(reps): m(t, x)
torch.cuda.synchronize(); print("done")
