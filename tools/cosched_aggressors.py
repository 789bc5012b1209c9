"""Which foreign work makes the op_sel'd packed-fp32 FMA of the folded fc1 / qkv consumers leave its solo result?  (round 6: shrinking the aggressor of
profiles/r05_cosched_root_cause.txt.)  Victim: DiT-L/2 batch-64 evaluations of an EXPERIMENT build with the round-4 vector form of the row affine
(tools/build_variant.sh packed "-DLFM_MEASURE -DLFM_EXP_AFFINE_PACKED": v_pk_fma_f32 with op_sel) with the per-kernel checksums armed; on a second stream ONE kind
of foreign work at a time.  Also runs the shipped form (op_sel_hi only) through the same aggressors as the counter-experiment.
usage: python tools/cosched_aggressors.py <lib.so> [reps]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import _build, hip  # noqa: E402

libpath = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
hip.LIB_PATH = libpath
_build.build = lambda *a, **k: libpath
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
depth = 24
x = torch.randn(64, 4, 32, 32, device=dev)
t = torch.tensor(0.5, device=dev)
L = hip.lib()
m(t, x)
ws = m._ws[1]
L.lfm_dit_chk_arm.argtypes = [C.c_void_p]
L.lfm_dit_chk_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
hip.check(L.lfm_dit_chk_arm(ws.data_ptr()), "lfm_dit_chk_arm")


def read_chk():
    buf = (C.c_ulonglong * (depth * 8))()
    hip.check(L.lfm_dit_chk_read(buf, depth * 8), "lfm_dit_chk_read")
    return list(buf)


m(t, x)
chk_ref = read_chk()
out_ref = m(t, x).clone()
assert chk_ref == read_chk(), "solo checksums must repeat"
SLOT = ["Q|K|V^T after qkv", "O after attention", "X after proj", "A' after proj", "row partials after proj", "H after fc1", "X after fc2", "A' after fc2"]
# foreign work, each sized to cover two victim evaluations (~25 ms)
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
A = torch.randn(16384, 1024, device=dev).half(); W = (torch.randn(4096, 1024, device=dev) * 0.03).half(); bias = torch.zeros(4096, device=dev)
Xr = torch.randn(16384, 1024, device=dev); sh = torch.zeros(1, 1024, device=dev); sc = torch.zeros(1, 1024, device=dev)
blob = torch.empty(256 << 20, dtype=torch.uint8, device=dev); blob2 = torch.empty_like(blob)
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
ticks = torch.zeros(ncu, dtype=torch.int64, device=dev)
torch.manual_seed(1)
small = init(DiT_models["DiT-B/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0))
small(t, x)


def mixed():  # the block's kernel SEQUENCE without the folded consumers: plain GEMM + GELU, attention, LayerNorm-modulate, alternating like a block loop
    for _ in range(48):
        hip.gemm_f16(A, W, bias, epilogue=1)
        hip.dit_attention(Q, K, Vt, Bh, heads, T)
        hip.ln_modulate(Xr, sh, sc, 256, 0)
        hip.gemm_f16(A, W, bias, epilogue=1)


AGG = {
    "none (control)": lambda: None,
    "twin DiT-L/2, fold OFF per call (separate LayerNorm launches: no folded consumer in the aggressor) (5)": lambda: [big._run(t, x, None, False, 1.0, fold_ln=1) for _ in range(5)],
    "twin DiT-L/2, folded, GEMMs on the one-wave-per-SIMD 256x256 kernel (gemm_select 6) (5)": lambda: [big._run(t, x, None, False, 1.0, gemm_select=7) for _ in range(5)],
    "twin DiT-B/2 batch 64 evaluations (folded, K = 768) (14)": lambda: [small(t, x) for _ in range(14)],
    "mixed sequence GEMM + GELU / attention / LayerNorm-modulate / GEMM (x 48)": mixed,
    "twin DiT-L/2 evaluations (5)": lambda: [big(t, x) for _ in range(5)],
    "attention kernel only (x 900)": lambda: [hip.dit_attention(Q, K, Vt, Bh, heads, T) for _ in range(900)],
    "fc1-shaped GEMM, bias + GELU epilogue (x 200)": lambda: [hip.gemm_f16(A, W, bias, epilogue=1) for _ in range(200)],
    "LayerNorm-modulate kernel, HBM streaming (x 1200)": lambda: [hip.ln_modulate(Xr, sh, sc, 256, 0) for _ in range(1200)],
    "bare MFMA stream, one workgroup per CU (lfm_clock_probe, ~25 ms)": lambda: hip.check(L.lfm_clock_probe(ncu, 10000, hip.ptr(ticks), hip.stream_ptr(dev)), "clock_probe"),
    "device-to-device copies, 256 MB (x 40)": lambda: [blob2.copy_(blob, non_blocking=True) for _ in range(40)],
}
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
print(f"library {libpath}: checksums armed, solo runs repeat; {reps} repetitions x 2 victim evaluations per aggressor", flush=True)
for name, fn in AGG.items():
    found, where = 0, {}
    for rep in range(reps):
        cur = torch.cuda.current_stream(dev)
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sb):
            fn()
        with torch.cuda.stream(sa):
            for _ in range(2):
                o = m(t, x)
        torch.cuda.synchronize()
        c = read_chk()
        firsts = [i for i in range(depth * 8) if c[i] != chk_ref[i]]
        if firsts or not torch.equal(o, out_ref):
            found += 1
            k = SLOT[firsts[0] % 8] if firsts else "output only"
            where[k] = where.get(k, 0) + 1
    print(f"  aggressor {name:70s}: {found:2d} of {reps} co-scheduled runs differ" + (f"   first differing tensor: {where}" if where else ""), flush=True)
