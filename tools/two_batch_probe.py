"""Two independent 64-image DiT-L/2 evaluations in flight on two HIP streams against the same two evaluations one after the other: does phase diversity
between the CUs (one stream's epilogues under the other's main loops) buy throughput at full batch?  usage: python tools/two_batch_probe.py"""
import statistics, sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
dev = torch.device("cuda:0")
def make():
    m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
    for p in m.parameters():
        if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
    return m.to(dev).eval()
ma, mb = make(), make()
xa = torch.randn(64, 4, 32, 32, device=dev); xb = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
REPS = 10
def serial():
    for _ in range(REPS): ma(t, xa); mb(t, xb)
def two_streams():
    cur = torch.cuda.current_stream(dev)
    sa.wait_stream(cur); sb.wait_stream(cur)
    for _ in range(REPS):
        with torch.cuda.stream(sa): ma(t, xa)
        with torch.cuda.stream(sb): mb(t, xb)
    cur.wait_stream(sa); cur.wait_stream(sb)
def timeit(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e3
for f in (serial, two_streams): f()
res = {"serial": [], "two_streams": []}
for rnd in range(5):
    for f in (serial, two_streams): res[f.__name__].append(timeit(f))
for k, v in res.items(): print(f"two 64-image evaluations, {k:12s}: median {statistics.median(v):7.3f} ms per pair   min {min(v):7.3f}", flush=True)
