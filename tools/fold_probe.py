"""LFM_OPT_FOLD_LN (adaLN LayerNorm-modulate folded into the GEMM epilogues, include/lfm_hip.h): DiT-L/2 forward at batch 64 with the option off / on,
interleaved, and the difference of the outputs.  usage: python tools/fold_probe.py [model] [batch]"""
import sys, statistics, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "DiT-L/2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
def timeit(fn, n=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
m = DiT_models[name](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
x = torch.randn(B, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
hip.set_option(hip.OPT_FOLD_LN, 0); base = m(t, x).clone()
hip.set_option(hip.OPT_FOLD_LN, 1); fold = m(t, x).clone(); torch.cuda.synchronize()
print(f"{name} b{B} folded vs separate LN: rel-L2 {float((fold - base).norm() / base.norm()):.2e}, max |diff| {float((fold - base).abs().max()):.2e}, finite {bool(torch.isfinite(fold).all())}", flush=True)
res = {0: [], 1: []}
for rnd in range(4):
    for v in (0, 1):
        hip.set_option(hip.OPT_FOLD_LN, v); res[v].append(timeit(lambda: m(t, x)))
hip.set_option(hip.OPT_FOLD_LN, 1)
for v in (0, 1): print(f"forward {name} b{B} fold_ln={v}: median {statistics.median(res[v]):7.3f} ms  min {min(res[v]):7.3f} ms", flush=True)
