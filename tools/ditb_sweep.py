"""DiT-B/2 at the config-4 evaluation shape (256 + 256 CFG rows = 131072 token rows): eager evaluations under kernel-selection variants, interleaved medians.
usage: python tools/ditb_sweep.py"""
import statistics, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = DiT_models["DiT-B/2"](img_resolution=32, in_channels=4, num_classes=1000, label_dropout=0.1)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
B = 256
x = torch.randn(2 * B, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
y = torch.cat([torch.randint(0, 1000, (B,)), torch.full((B,), 1000)]).to(dev)
def run(): return m.forward_with_cfg(t, x, y, cfg_scale=1.5)
def timeit(n=6):
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
variants = {"default": (0, 0, 1), "tile groups of 8 (flag 32)": (32, 0, 1), "tile groups of 2 (flag 64)": (64, 0, 1), "no XCD remap (flag 128)": (128, 0, 1),
            "one-wave-per-SIMD GEMMs (LFM_OPT_GEMM_V6)": (0, 1, 1), "per-item attention": (0, 0, 0)}
res = {k: [] for k in variants}
ref = None
for rnd in range(4):
    for k, (flags, v6, stream) in variants.items():
        hip.gemm_select(flags << 4); hip.set_option(hip.OPT_GEMM_V6, v6); hip.set_option(hip.OPT_ATTENTION_STREAM, stream)
        res[k].append(timeit())
        if rnd == 0:
            o = run(); torch.cuda.synchronize()
            if ref is None: ref = o.clone()
            print(f"  {k}: output equal to default: {bool(torch.equal(o, ref))}", flush=True)
hip.gemm_select(0); hip.set_option(hip.OPT_GEMM_V6, 0); hip.set_option(hip.OPT_ATTENTION_STREAM, 1)
for k, v in res.items(): print(f"{k:48s}: median {statistics.median(v):7.3f} ms per evaluation (min {min(v):7.3f})")
