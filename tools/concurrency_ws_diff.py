"""Where does a co-scheduled DiT evaluation first differ from a solo one?  A depth-1 DiT of DiT-L/2's width (fold path, 64 images) is run alone and then
while a full DiT-L/2 twin hammers a second stream; after every run the WHOLE workspace is compared region by region (layout = csrc/dit.hip carve()).
usage: python tools/concurrency_ws_diff.py [reps] [depth]"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
from lfm_amd.models.DiT import DiT
from lfm_amd.solvers import concurrency_twin
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
def init(m):
    for p in m.parameters():
        if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
    return m.to(dev).eval()
torch.manual_seed(0)
m = init(DiT(img_resolution=32, patch_size=2, in_channels=4, hidden_size=1024, depth=depth, num_heads=16, num_classes=1, label_dropout=0.0))
big = init(DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0))
comp = concurrency_twin(m) if len(sys.argv) > 3 and sys.argv[3] == "twin" else big  # the competitor: an independent model, or a twin on the same weights
B, T, D, H = 64, 256, 1024, 4096
M = B * T
x = torch.randn(B, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
al = lambda v: (v + 255) // 256 * 256
J = depth * 6 * D + 2 * D
regions, off = [], 0
for name, nbytes in (("X", M * D * 4), ("A (attention output O)", M * D * 2), ("QKVH (fc1 activation H)", max(3 * M * D, M * H) * 2), ("temb", B * D * 4), ("temb_h", B * D * 4),
                     ("c_half", B * D * 2), ("mod", B * J * 4), ("ones", D * 4), ("ln_part", M * 4 * 8), ("cen0", M * 4), ("cen1", M * 4), ("amod", depth * 4 * B * D * 2),
                     ("uvq", depth * 2 * B * 3 * D * 4), ("uvf", depth * 2 * B * H * 4), ("A2 (proj A')", M * D * 2)):
    regions.append((name, off, nbytes)); off += al(nbytes)
mode = sys.argv[4] if len(sys.argv) > 4 else ""
if mode == "nofold": hip.set_option(hip.OPT_FOLD_LN, 0)       # separate LayerNorm-modulate launches
if mode == "wide": hip.gemm_select(256 << 4)                  # attention: 4 waves x 64 queries
if mode == "v6": hip.set_option(hip.OPT_GEMM_V6, 1)           # the one-wave-per-SIMD GEMMs
if mode == "noxcd": hip.gemm_select(16 << 4)
print("mode:", mode or "shipped", flush=True)
out_ref = m(t, x).clone(); ws = m._ws[1]
assert off <= ws.numel(), (off, ws.numel())
torch.cuda.synchronize(); ws_ref = ws.clone()
# (M) LFM_MEASURE builds: per-kernel checksums of m's evaluations (csrc/dit.hip lfm_dit_chk_*): WHERE does a co-scheduled run leave the solo one first?
import ctypes as C
L = hip.lib()
chk_ref = None
SLOT = ["Q|K|V^T after qkv", "O after attention", "X after proj", "A' (A2) after proj", "row partials after proj", "H after fc1", "X after fc2", "A' (A) after fc2"]
def read_chk():
    buf = (C.c_ulonglong * (depth * 8))()
    L.lfm_dit_chk_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    hip.check(L.lfm_dit_chk_read(buf, depth * 8), "lfm_dit_chk_read")
    return list(buf)
if hasattr(L, "lfm_dit_chk_arm"):
    L.lfm_dit_chk_arm.argtypes = [C.c_void_p]
    hip.check(L.lfm_dit_chk_arm(ws.data_ptr()), "lfm_dit_chk_arm")
    m(t, x); chk_ref = read_chk(); m(t, x)
    assert chk_ref == read_chk(), "solo checksums must repeat"
    out_ref = m(t, x).clone(); torch.cuda.synchronize(); ws_ref = ws.clone()
    print("per-kernel checksums armed (the checksum launches sit between the block's kernels: timing is not the shipped one)", flush=True)
out2 = m(t, x); torch.cuda.synchronize()
solo_same = [torch.equal(ws[o:o + n], ws_ref[o:o + n]) for _, o, n in regions]
print("solo repeat: every region identical:", all(solo_same), [r[0] for r, s in zip(regions, solo_same) if not s], flush=True)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
found = 0
for rep in range(reps):
    cur = torch.cuda.current_stream(dev); sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sb):
        for _ in range(2 if depth <= 4 else 5): comp(t, x)
    with torch.cuda.stream(sa):
        for _ in range(3 if depth <= 4 else 2): o = m(t, x)  # the last one is compared: it runs well inside the other stream's work
    torch.cuda.synchronize()
    bad = [(name, int((ws[o_:o_ + n] != ws_ref[o_:o_ + n]).sum())) for name, o_, n in regions if not torch.equal(ws[o_:o_ + n], ws_ref[o_:o_ + n])]
    if bad or not torch.equal(o, out_ref):
        found += 1
        print(f"rep {rep}: output equal {torch.equal(o, out_ref)}; differing regions (bytes): {bad}", flush=True)
        if chk_ref is not None:
            cur_chk = read_chk()
            firsts = [i for i in range(depth * 8) if cur_chk[i] != chk_ref[i]]
            if firsts:
                print("    first differing checksums: " + "; ".join(f"block {i // 8}: {SLOT[i % 8]}" for i in firsts[:6]) + f"  ({len(firsts)} slots differ)", flush=True)
            else:
                print("    all per-kernel checksums equal (the difference is outside the block loop)", flush=True)
        reg = {n_: (o_, b_) for n_, o_, b_ in regions}
        def f32(name, ref=False):
            o_, b_ = reg[name]
            return (ws_ref if ref else ws)[o_:o_ + b_].view(torch.float32)
        o_, b_ = reg["QKVH (fc1 activation H)"]
        hh, hr = ws[o_:o_ + M * H * 2].view(torch.int16).view(M, H), ws_ref[o_:o_ + M * H * 2].view(torch.int16).view(M, H)
        nz = (hh != hr).nonzero()
        if 0 < nz.shape[0] <= 4000:  # few elements: the LAST block's fc1 output itself is where the runs part -- where are they?
            rows, cols = nz[:, 0], nz[:, 1]
            print(f"    H: {nz.shape[0]} elements differ; rows mod 256 {sorted(set((rows % 256).tolist()))[:40]}; cols mod 256 {sorted(set((cols % 256).tolist()))[:40]}; "
                  f"col tiles {sorted(set((cols // 256).tolist()))}; first pairs {nz[:12].tolist()}", flush=True)
            d16 = (hh.view(torch.float16)[rows, cols].float() - hr.view(torch.float16)[rows, cols].float()).abs()
            print(f"       max |dH| {float(d16.max()):.3e} at |H| {float(hr.view(torch.float16)[rows, cols].float().abs().max()):.3e}", flush=True)
        for nm in ("cen0", "cen1"):
            a_, r_ = f32(nm), f32(nm, True)
            d_ = (a_ - r_).abs(); idx = (d_ > 0).nonzero().flatten()
            if idx.numel():
                print(f"    {nm}: {idx.numel()} rows differ, images {sorted(set((idx // T).tolist()))}, max |d| {float(d_.max()):.3e}, max |d| / |c| {float((d_ / r_.abs().clamp_min(1e-6)).max()):.3e}", flush=True)
        a_, r_ = f32("X").view(B, -1), f32("X", True).view(B, -1)
        imgs = ((a_ != r_).sum(1) > 0).nonzero().flatten().tolist()
        dx = (a_ - r_).abs()
        print(f"    X: images {imgs}; max |dX| {float(dx.max()):.3e} (|X| max {float(r_.abs().max()):.1f}); per affected image fraction of elements differing "
              f"{[round(float((a_[i] != r_[i]).float().mean()), 2) for i in imgs[:8]]}", flush=True)
        if found >= 8: break
print(f"{found} of {rep + 1} co-scheduled runs differ", flush=True)
