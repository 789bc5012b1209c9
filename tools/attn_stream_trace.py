"""s_memtime timeline of the streamed attention kernel (measurement build, MODE 3): wave 0 of the first and the last workgroup, and per workgroup
{CU, start, first barrier passed, end}.  usage: LFM_HIP_LIBRARY=tools/ship_variants/measure/liblfm_hip.so python tools/attn_stream_trace.py"""
import ctypes as C, sys, collections, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
hip.set_option(hip.OPT_ATTENTION_STREAM, 1)
hip.gemm_select((3 << 25) << 4)
for _ in range(3): hip.dit_attention(Q, K, Vt, Bh, heads, T)
torch.cuda.synchronize()
L = hip.lib()
L.lfm_attention_trace_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]; L.lfm_attention_wg_trace_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
tb = (C.c_ulonglong * 64)(); hip.check(L.lfm_attention_trace_read(tb, 64), "trace_read")
n = 512
buf = (C.c_ulonglong * (4 * n))(); hip.check(L.lfm_attention_wg_trace_read(buf, n), "wg_trace_read")
hip.gemm_select(0)
names = {0: "start", 1: "prologue issued"}
for i in range(2):
    for j in range(4):
        names[2 + 14 * i + 3 * j] = f"item {i} stage {j}: at the wait"; names[3 + 14 * i + 3 * j] = f"item {i} stage {j}: landed (vmcnt)"; names[4 + 14 * i + 3 * j] = f"item {i} stage {j}: barrier passed"
    names[14 + 14 * i] = f"item {i}: key loop done"; names[15 + 14 * i] = f"item {i}: stores issued"
for base, who in ((0, "first"), (32, "last")):
    print(f"--- wave 0 of the {who} workgroup (s_memtime ticks since its start)")
    t0 = tb[base]; prev = t0
    for s in sorted(names):
        if s < 32 and tb[base + s]:
            print(f"  {names[s]:40s} {tb[base + s] - t0:8d}  (+{tb[base + s] - prev})"); prev = tb[base + s]
rec = []
for i in range(n):
    hw, a, b, c = buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]
    hwid, xcc = hw & 0xffffffff, (hw >> 32) & 15
    rec.append((i, xcc, (hwid >> 13) & 7, (hwid >> 12) & 1, (hwid >> 8) & 15, a, b, c))
tmin = min(r[5] for r in rec)
bycu = collections.defaultdict(list)
for r in rec: bycu[r[1:5]].append(r)
print(f"{len(bycu)} distinct CUs; workgroups per CU: {collections.Counter(len(v) for v in bycu.values())}")
for k in list(sorted(bycu))[:8]:
    v = sorted(bycu[k], key=lambda r: r[5])
    print(f"  CU {k}: " + "  ".join(f"[id {r[0]} {r[5] - tmin}..{r[6] - tmin}..{r[7] - tmin}]" for r in v))
st = sorted(r[5] - tmin for r in rec); en = sorted(r[7] - tmin for r in rec); fb = sorted(r[6] - r[5] for r in rec); du = sorted(r[7] - r[5] for r in rec)
print(f"starts: min {st[0]} median {st[n // 2]} max {st[-1]}; ends: min {en[0]} median {en[n // 2]} max {en[-1]}")
print(f"start -> first barrier passed: min {fb[0]} median {fb[n // 2]} max {fb[-1]}; workgroup duration: min {du[0]} median {du[n // 2]} max {du[-1]}")
ov = 0
for k, v in bycu.items():
    v = sorted(v, key=lambda r: r[5])
    for a, b in zip(v, v[1:]):
        if b[5] < a[7]: ov += 1
print(f"pairs of workgroups on one CU whose lifetimes overlap: {ov} of {sum(len(v) - 1 for v in bycu.values())}")
