"""s_memtime timeline of the attention kernel's trace build (MODE 3: select flags 33554432 | 67108864) at the DiT-L/2 batch-64 shape: wave 0 of the
first workgroup (first round of workgroups) and of the last one (second round).  Where do the ~40k cycles of a workgroup go?  Written at the end of
round 2, not yet run on hardware.  usage: python tools/attn_trace.py"""
import ctypes as C, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
hip.gemm_select(((1 << 25) | (2 << 25)) << 4)
for _ in range(3): hip.dit_attention(Q, K, Vt, Bh, heads, T)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
hip.check(hip.lib().lfm_attention_trace_read(buf, 64), "lfm_attention_trace_read")
hip.gemm_select(0)
names = {0: "start", 1: "DMAs + Q loads issued", 2: "K, Q landed (barrier)", 19: "stores issued", 20: "stores acknowledged", 21: "  first softmax VALU done",
         22: "  V^T landed (barrier)", 23: "  first PV MFMAs issued"}
for it in range(4):
    names[3 + 4 * it] = f"it {it}: S(odd block) issued"; names[4 + 4 * it] = f"it {it}: softmax + PV (even block) done"
    names[5 + 4 * it] = f"it {it}: S(next even block) issued"; names[6 + 4 * it] = f"it {it}: softmax + PV (odd block) done"
for base, tag in ((0, "first workgroup"), (32, "last workgroup")):
    t0 = buf[base]
    print(f"--- {tag} (cycles since its start; s_memtime ticks at the constant 100 MHz reference if the values look 25x too small)")
    order = [0, 1, 2, 3, 21, 22, 23] + list(range(4, 21))
    prev = t0
    for s in order:
        v = buf[base + s]
        if v == 0: continue
        print(f"  {names.get(s, str(s)):40s} {v - t0:9d}  (+{v - prev})")
        prev = v
