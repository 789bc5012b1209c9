"""A/B of LFM_OPT_FUSED_QKV_ATTENTION: eager DiT forwards (batch 64 by default) timed with events, the two settings interleaved.
usage: fused_qkv_ab.py [model] [batch] [reps]"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
name = sys.argv[1] if len(sys.argv) > 1 else "DiT-L/2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # lfm_gemm_select flags for the fused runs (A/B of kernel variants)
dev = torch.device("cuda:0")
m = DiT_models[name](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
x = torch.randn(batch, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
depth = len(m.blocks) if hasattr(m, "blocks") else {"DiT-L/2": 24, "DiT-B/2": 12}[name]
outs = {}
for opt in (0, 1):
    hip.set_option(hip.OPT_FUSED_QKV_ATTENTION, opt)
    outs[opt] = m(t, x).clone()
torch.cuda.synchronize()
print("bit-identical:", bool(torch.equal(outs[0], outs[1])), " max|diff|", float((outs[0] - outs[1]).abs().max()), " finite", bool(torch.isfinite(outs[1]).all()))
res = {0: [], 1: [], 2: []}
opts = (0, 1, 2) if flags else (0, 1)
for rnd in range(5):
    for opt in opts:
        hip.set_option(hip.OPT_FUSED_QKV_ATTENTION, 1 if opt else 0)
        hip.gemm_select((flags << 4) if opt == 2 else 0)
        for _ in range(3): m(t, x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): m(t, x)
        e1.record(); torch.cuda.synchronize()
        res[opt].append(e0.elapsed_time(e1) / reps)
hip.gemm_select(0)
for opt in opts:
    v = sorted(res[opt]); med = v[len(v) // 2]
    print(f"{name} batch {batch} fused={opt}: median {med * 1e3:8.1f} us per forward  ({med * 1e3 / depth:6.1f} us per block incl. the non-block share)  all {[round(a * 1e3, 1) for a in res[opt]]}")
hip.set_option(hip.OPT_FUSED_QKV_ATTENTION, 1)
