"""Experiment: batch 64 as ONE graph vs two concurrent batch-32 graphs on two streams (memory/compute phase overlap)."""
import copy, sys, time
import torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd.models import DiT_models
from lfm_amd.solvers import GraphedFixedGrid, torchdiffeq_euler_grid
from lfm_amd.test_flow_latent import dezero_
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = dezero_(DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, label_dropout=0.0, num_classes=1)).to(dev).eval()
ts, dts = torchdiffeq_euler_grid(0.02)
def bench(fn, n=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
s64 = GraphedFixedGrid(m, 64); s64.set_grid(ts, dts)
x64 = torch.randn(64, 4, 32, 32, device=dev)
m2 = dezero_(DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, label_dropout=0.0, num_classes=1)).to(dev).eval()
t = bench(lambda: s64.run(x64)); print(f"one graph  b64: {t*1e3:.1f} ms  ({t*1e3/50:.2f} ms/step)")
ms = [m, m2]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
sol = []
for mm in ms:
    s = GraphedFixedGrid(mm, 32); s.set_grid(ts, dts); sol.append(s)
xs = [x64[:32].contiguous(), x64[32:].contiguous()]
for s, x in zip(sol, xs): s.run(x)   # capture on default stream context
torch.cuda.synchronize()
def both():
    for s, x, st in zip(sol, xs, streams):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st): s.run(x)
    for st in streams: torch.cuda.current_stream().wait_stream(st)
t2 = bench(both); print(f"two graphs b32 x2 concurrent: {t2*1e3:.1f} ms  ({t2*1e3/50:.2f} ms/step-pair)")
t1 = bench(lambda: sol[0].run(xs[0])); print(f"one graph  b32 alone: {t1*1e3:.1f} ms")
