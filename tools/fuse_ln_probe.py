"""EXPERIMENTAL option LFM_OPT_FUSE_LN (LayerNorm-modulate inside the proj / fc2 GEMM epilogues): DiT-L/2 forward at batch 64 with the option
off / on, interleaved, and the difference of the outputs.  Not yet run on hardware (round 2 ended without GPU time).  usage: python tools/fuse_ln_probe.py"""
import sys, statistics, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
dev = torch.device("cuda:0")
def timeit(fn, n=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
x = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
base = m(t, x).clone()
hip.set_option(hip.OPT_FUSE_LN, 1)
fused = m(t, x).clone(); torch.cuda.synchronize()
print(f"fused vs separate: rel-L2 {float((fused - base).norm() / base.norm()):.2e}, finite {bool(torch.isfinite(fused).all())}", flush=True)
res = {0: [], 1: []}
for rnd in range(4):
    for v in (0, 1):
        hip.set_option(hip.OPT_FUSE_LN, v); res[v].append(timeit(lambda: m(t, x)))
hip.set_option(hip.OPT_FUSE_LN, 0)
for v in (0, 1): print(f"forward DiT-L/2 b64 fuse_ln={v}: median {statistics.median(res[v]):7.3f} ms  min {min(res[v]):7.3f} ms", flush=True)
