"""Co-scheduling non-determinism of the folded fc1 consumer: event rate of an EXPERIMENT build, and -- for the dump build
(-DLFM_EXP_DUMP) -- the operands of the epilogue's affine exactly as the lanes saw them, solo vs co-scheduled, at the elements that leave the solo run.

usage: python tools/cosched_dump.py tools/variants/NAME/liblfm_hip.so [reps] [dump]
Builds: tools/build_variant.sh NAME "-DLFM_MEASURE [-DLFM_EXP_...]".  The per-kernel checksum launches (lfm_dit_chk_arm) are armed in every run: they
raise the event rate from ~4 % to 30-50 % of the co-scheduled evaluations (profiles/r04_two_batches_in_flight.txt)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
from lfm_amd import _build, hip  # noqa: E402

libpath = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
want_dump = len(sys.argv) > 3 and sys.argv[3] == "dump"
hip.LIB_PATH = libpath
_build.build = lambda *a, **k: libpath  # the experiment library as built; no stamp check
from lfm_amd.models import DiT_models  # noqa: E402

dev = torch.device("cuda:0")


def init(m):
    for p in m.parameters():
        if not bool(p.any()):
            torch.nn.init.normal_(p, std=0.02)
    return m.to(dev).eval()


torch.manual_seed(0)
m = init(DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0))
big = init(DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0))
depth, B, T, D, H = 24, 64, 256, 1024, 4096
M = B * T
x = torch.randn(B, 4, 32, 32, device=dev)
t = torch.tensor(0.5, device=dev)
L = hip.lib()
out_ref = m(t, x).clone()
ws = m._ws[1]
L.lfm_dit_chk_arm.argtypes = [C.c_void_p]
L.lfm_dit_chk_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
hip.check(L.lfm_dit_chk_arm(ws.data_ptr()), "lfm_dit_chk_arm")
SLOT = ["Q|K|V^T after qkv", "O after attention", "X after proj", "A' (A2) after proj", "row partials after proj", "H after fc1", "X after fc2", "A' (A) after fc2"]


def read_chk():
    buf = (C.c_ulonglong * (depth * 8))()
    hip.check(L.lfm_dit_chk_read(buf, depth * 8), "lfm_dit_chk_read")
    return list(buf)


dump = None
G = H // 8
if want_dump:
    dump = torch.zeros(depth, M, G, 16, device=dev)
    L.lfm_dit_dbg_arm.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    hip.check(L.lfm_dit_dbg_arm(ws.data_ptr(), dump.data_ptr(), M * G * 16), "lfm_dit_dbg_arm")
m(t, x)
chk_ref = read_chk()
out_ref = m(t, x).clone()
assert chk_ref == read_chk(), "solo checksums must repeat"
torch.cuda.synchronize()
dump_ref = dump.clone() if want_dump else None
if want_dump:
    m(t, x)
    torch.cuda.synchronize()
    print("solo repeat: dump identical:", torch.equal(dump.view(torch.int32), dump_ref.view(torch.int32)), flush=True)
print(f"library {libpath}: checksums armed, solo runs repeat", flush=True)
FIELD = ["acc.x", "acc.y", "acc.z", "acc.w", "acc.hx", "acc.hy", "acc.hz", "acc.hw", "t.x", "t.z", "t.hx", "t.hz", "a", "b", "x.x", "x.z"]
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
found = 0
for rep in range(reps):
    cur = torch.cuda.current_stream(dev)
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    with torch.cuda.stream(sb):
        for _ in range(5):
            big(t, x)
    with torch.cuda.stream(sa):
        for _ in range(2):
            o = m(t, x)  # the last one is compared: it runs well inside the other stream's work
    torch.cuda.synchronize()
    cur_chk = read_chk()
    firsts = [i for i in range(depth * 8) if cur_chk[i] != chk_ref[i]]
    if not firsts and torch.equal(o, out_ref):
        continue
    found += 1
    print(f"rep {rep}: output equal {torch.equal(o, out_ref)}; first differing checksums: " +
          "; ".join(f"block {i // 8}: {SLOT[i % 8]}" for i in firsts[:3]) + f"  ({len(firsts)} slots differ)", flush=True)
    if want_dump and firsts:
        blk = firsts[0] // 8
        a_, r_ = dump[blk].view(torch.int32), dump_ref[blk].view(torch.int32)
        ne = a_ != r_
        print(f"    block {blk}: dump words differing per field: " + ", ".join(f"{FIELD[f]} {int(ne[..., f].sum())}" for f in range(16) if int(ne[..., f].sum())), flush=True)
        # origin elements: the accumulators (a function of the kernel's INPUTS only) are equal, something downstream is not
        acc_eq = ~ne[..., :8].any(-1)
        down = ne[..., 8:].any(-1)
        org = (acc_eq & down).nonzero()
        print(f"    (row, column group) pairs with equal accumulators and a differing downstream operand: {org.shape[0]}; with differing accumulators: {int((~acc_eq).sum())}", flush=True)
        fa, fr = dump[blk], dump_ref[blk]
        for k in range(min(org.shape[0], 24)):
            r, g = int(org[k, 0]), int(org[k, 1])
            fl = [f for f in range(16) if bool(ne[r, g, f])]
            print(f"      row {r} (mod 256 = {r % 256}, mod 8 = {r % 8}) cols {8 * g}.. (tile {8 * g // 256}, wave {(8 * g % 256) // 64}, rcol {g % 8}): " +
                  "; ".join(f"{FIELD[f]} solo {float(fr[r, g, f]):.9g} co {float(fa[r, g, f]):.9g}" for f in fl) +
                  f" | a {float(fr[r, g, 12]):.6g} b {float(fr[r, g, 13]):.6g} t.x {float(fr[r, g, 8]):.6g} t.z {float(fr[r, g, 9]):.6g}", flush=True)
        if org.shape[0]:
            # is a wrong b some OTHER row's b?  (rows of the same tile: rs[] of the workgroup)
            r, g = int(org[0, 0]), int(org[0, 1])
            bad_b = float(fa[r, g, 13])
            m0 = r // 256 * 256
            same = [(rr, float(fr[rr, g, 13])) for rr in range(m0, m0 + 256) if rr % 8 >= 6 and float(fr[rr, g, 13]) == bad_b]
            print(f"      first origin: co-scheduled b {bad_b:.9g}; rows of the tile whose solo b equals it: {same[:8]}", flush=True)
    if found >= 6:
        break
print(f"{found} of {rep + 1} co-scheduled runs differ  [{libpath}]", flush=True)
