"""Which CU does every attention workgroup run on, and when?  MODE 3 build: per workgroup {HW_ID, XCC_ID, start, loads landed, end} (s_memtime).
Prints the dispatch pattern (linear id -> xcc / se / cu), the pairs that share a CU, and a per-CU timeline summary.  usage: python tools/attn_wgmap.py"""
import ctypes as C, sys, collections, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
hip.gemm_select(((1 << 25) | (2 << 25)) << 4)
for _ in range(3): hip.dit_attention(Q, K, Vt, Bh, heads, T)
torch.cuda.synchronize()
n = Bh * heads
buf = (C.c_ulonglong * (4 * n))()
L = hip.lib(); L.lfm_attention_wg_trace_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
hip.check(L.lfm_attention_wg_trace_read(buf, n), "wg_trace_read")
hip.gemm_select(0)
rec = []
for i in range(n):
    hw, t0, t1, t2 = buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]
    hwid, xcc = hw & 0xffffffff, (hw >> 32) & 15
    cu, sh, se = (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
    rec.append((i, xcc, se, sh, cu, t0, t1, t2))
tmin = min(r[5] for r in rec)
print("first 40 workgroups: id -> (xcc, se, sh, cu) start landed end (cycles after the first start)")
for r in rec[:40]: print(f"  {r[0]:4d} -> xcc {r[1]} se {r[2]} sh {r[3]} cu {r[4]:2d}   {r[5] - tmin:7d} {r[6] - tmin:7d} {r[7] - tmin:7d}")
bycu = collections.defaultdict(list)
for r in rec: bycu[r[1:5]].append(r)
print(f"{len(bycu)} distinct CUs; workgroups per CU: {collections.Counter(len(v) for v in bycu.values())}")
for k in list(sorted(bycu))[:6]:
    v = sorted(bycu[k], key=lambda r: r[5])
    print(f"  CU {k}: " + "  ".join(f"[id {r[0]} {r[5] - tmin}..{r[6] - tmin}..{r[7] - tmin}]" for r in v))
ends = sorted(r[7] - tmin for r in rec); starts = sorted(r[5] - tmin for r in rec)
print(f"starts: min {starts[0]} median {starts[n // 2]} max {starts[-1]};  ends: min {ends[0]} median {ends[n // 2]} max {ends[-1]}  (s_memtime ticks; 100 MHz => x24 core cycles?)")
dur = sorted(r[7] - r[5] for r in rec); ld = sorted(r[6] - r[5] for r in rec)
print(f"workgroup duration: min {dur[0]} median {dur[n // 2]} max {dur[-1]}; load phase: min {ld[0]} median {ld[n // 2]} max {ld[-1]}")
# id difference of CU-sharing pairs in the first round
d = collections.Counter()
for k, v in bycu.items():
    v = sorted(v, key=lambda r: r[5])
    if len(v) >= 2: d[abs(v[0][0] - v[1][0])] += 1
print("id distance of the two earliest workgroups on a CU:", d.most_common(8))
