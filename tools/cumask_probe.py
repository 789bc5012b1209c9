"""Spatial partitioning probe: two half-batch DiT-L/2 forwards on two CU-masked streams (128 CUs each) vs one full-batch forward."""
import ctypes as C, sys, statistics, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
dev = torch.device("cuda:0"); torch.set_grad_enabled(False)
rt = C.CDLL("libamdhip64.so")
def masked_stream(words):
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = rt.hipExtStreamCreateWithCUMask(C.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)
masks = {
    "lowhigh": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
    "evenodd": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
    "xcd0123": ([0x0F0F0F0F] * 8, [0xF0F0F0F0] * 8),   # if bit i -> XCD i%8: XCDs 0-3 / 4-7
}
def mk():
    m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
    for p in m.parameters():
        if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
    return m.to(dev).eval()
mA, mB = mk(), mk()
mB.load_state_dict(mA.state_dict())
x64 = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
xa, xb = x64[:32].contiguous(), x64[32:].contiguous()
def wall(fn, n=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ref = mA(t, x64); torch.cuda.synchronize()
print(f"full batch 64, 256 CUs, default stream: {wall(lambda: mA(t, x64)):.3f} ms per forward", flush=True)
print(f"half batch 32, default stream alone   : {wall(lambda: mA(t, xa)):.3f} ms", flush=True)
for name, (wa, wb) in masks.items():
    sa, sb = masked_stream(wa), masked_stream(wb)
    def one():
        with torch.cuda.stream(sa): mA(t, xa)
    def both():
        with torch.cuda.stream(sa): ra = mA(t, xa)
        with torch.cuda.stream(sb): rb = mB(t, xb)
        return ra, rb
    def both_n(k=4):  # k forwards back to back per stream (like a solver loop), to amortise the host launches
        for _ in range(k):
            with torch.cuda.stream(sa): mA(t, xa)
            with torch.cuda.stream(sb): mB(t, xb)
    torch.cuda.synchronize()
    t1 = wall(one)
    t2 = wall(both)
    t4 = wall(both_n, n=3) / 4
    ra, rb = both(); torch.cuda.synchronize()
    err = float((torch.cat([ra, rb]) - ref).abs().max())
    print(f"mask {name:8s}: one half alone {t1:.3f} ms | two halves concurrently {t2:.3f} ms per 64 images | in a loop {t4:.3f} ms per 64 images | max diff vs full {err:.2e}", flush=True)
# graph replay on masked streams
sa, sb = masked_stream(masks["xcd0123"][0]), masked_stream(masks["xcd0123"][1])
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
oa, ob = torch.empty_like(xa), torch.empty_like(xb)
with torch.cuda.stream(sa):
    mA(t, xa)
    with torch.cuda.graph(ga, stream=sa): mA._run(t, xa, None, False, 1.0, out=oa)
with torch.cuda.stream(sb):
    mB(t, xb)
    with torch.cuda.graph(gb, stream=sb): mB._run(t, xb, None, False, 1.0, out=ob)
torch.cuda.synchronize()
def g_one():
    with torch.cuda.stream(sa): ga.replay()
def g_both():
    with torch.cuda.stream(sa): ga.replay()
    with torch.cuda.stream(sb): gb.replay()
def g_both_n(k=4):
    for _ in range(k): g_both()
print(f"graphs on masked streams: one half {wall(g_one):.3f} ms | both {wall(g_both):.3f} ms | loop {wall(g_both_n, n=3)/4:.3f} ms per 64 images; max diff {float((torch.cat([oa, ob]) - ref).abs().max()):.2e}", flush=True)
