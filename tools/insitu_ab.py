"""In-situ A/B: whole DiT-L/2 forwards (batch 64) with GEMM debug flags toggled in auto mode, interleaved rounds."""
import sys, statistics, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
dev = torch.device("cuda:0")
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0).to(dev).eval()
x = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
def timeit(n=6):
    for _ in range(2): m(t, x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): m(t, x)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
variants = [("default", 0)] + [(a.split("=")[0], int(a.split("=")[1]) << 4) for a in sys.argv[1:]]
res = {}
for rnd in range(4):
    for name, sel in variants:
        hip.gemm_select(sel); res.setdefault(name, []).append(timeit())
hip.gemm_select(0)
for k, v in res.items(): print(f"{k:28s}: median {statistics.median(v):7.3f} ms  min {min(v):7.3f} ms per forward")
