"""s_memtime timeline of the fused QKV + attention kernel (LFM_MEASURE build, LFM_HIP_LIBRARY=...): waves 0 (wave group 0) and 4 (group 1) of workgroup 0, the
four items of the workgroup, seven stamps per item.  One s_memtime tick = 10 ns (100 MHz reference counter)."""
import ctypes as C, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
dev = torch.device("cuda:0")
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
x = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
hip.gemm_select((2 | (int(sys.argv[1]) if len(sys.argv) > 1 else 0)) << 4)
for _ in range(3): m(t, x)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
hip.check(hip.lib().lfm_attention_trace_read(buf, 64), "trace")
hip.gemm_select(0)
names = ["top", "operands landed", "K loop done", "hand-over done", "next requested", "key loop done", "stores issued"]
for grp in (0, 1):
    v = [int(buf[grp * 32 + i]) for i in range(28)]
    t0 = v[0]
    print(f"wave {4 * grp} of workgroup 0 (the last block's launch): ticks since the first stamp; delta to the previous stamp")
    prev = t0
    for k in range(4):
        for s_ in range(7):
            cur = v[7 * k + s_]
            print(f"  item {k} {names[s_]:16s} {cur - t0:8d}  +{cur - prev:6d}")
            prev = cur
