"""(M) Do two co-resident workgroups hide each other's epilogues when they do NOT start together?  The 256x128 GEMM (kernel 4) and the halo-tiled 3x3
convolution keep two 4-wave workgroups per CU; dispatched together they run in lockstep.  lfm_set_option(3, ticks) delays workgroups 256..511 (the
second resident of every CU in the first wave) by `ticks` s_memtime ticks; later workgroups inherit the offset.
usage: LFM_MEASURE=1 python -m lfm_amd._build && python tools/stagger_probe.py [attn]"""
import statistics, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
L = hip.lib()
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
TICKS = [0, 500, 1000, 2000, 4000, 8000, 16000]  # s_memtime runs at 100 MHz on this part if constant-rate, else ~1 GHz: the sweep covers both
M = 16384
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""  # "attn": the attention part only
for name, N, K, epi in [] if ONLY == "attn" else [("fc1  bias+gelu", 4096, 1024, 1), ("fc2  gate+resid", 1024, 4096, 3), ("qkv  split", 3072, 1024, "qkv")]:
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    if epi == "qkv":
        run = lambda: hip.gemm_qkv_f16(A, W, b, 64, 256)
    elif epi == 3:
        X = torch.randn(M, N, device=dev); gate = torch.randn(M // 256, N, device=dev)
        run = lambda: hip.gemm_f16(A, W, b, epilogue=3, out=X, gate=gate, gate_stride=N, tokens=256)
    else:
        out = torch.zeros(M, N, device=dev, dtype=torch.float16)
        run = lambda: hip.gemm_f16(A, W, b, epilogue=1, out=out)
    res = {}
    for rnd in range(3):
        for tk in TICKS:
            hip.check(L.lfm_set_option(3, tk), "opt")
            for sel, nm in ((4, "v4"), (4 | (4 << 4), "v4 no epilogue")):
                hip.gemm_select(sel); res.setdefault((nm, tk), []).append(timeit(run))
        hip.check(L.lfm_set_option(3, 0), "opt")
        hip.gemm_select(5); res.setdefault(("v5", 0), []).append(timeit(run))
    hip.gemm_select(0)
    print(f"--- {name}  M={M} N={N} K={K}:  v5 {statistics.median(res[('v5', 0)]):7.1f} us", flush=True)
    for tk in TICKS:
        print(f"    stagger {tk:6d} ticks: v4 {statistics.median(res[('v4', tk)]):7.1f} us   v4 no epilogue {statistics.median(res[('v4 no epilogue', tk)]):7.1f} us", flush=True)

for N, H, Wd, Cin, Cout in [] if ONLY == "attn" else [(64, 256, 256, 128, 128), (64, 128, 128, 256, 256), (64, 64, 64, 512, 512), (32, 32, 32, 512, 512)]:
    if N * H * Wd * max(Cin, Cout) * 2 > 6e9:
        N = max(1, int(6e9 // (H * Wd * max(Cin, Cout) * 2)))
    x = torch.randn(N * H * Wd, Cin, device=dev, dtype=torch.float16)
    w = (torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).half()
    b = torch.zeros(Cout, device=dev)
    out = torch.empty(N * H * Wd, Cout, device=dev, dtype=torch.float16)
    flop = 2.0 * N * H * Wd * Cout * 9 * Cin
    run = lambda: hip.check(L.lfm_conv3x3_f16(hip.ptr(x), hip.ptr(w), hip.ptr(b), None, hip.ptr(out), N, H, Wd, Cin, Cout, 0, hip.stream_ptr()), "conv")
    res = {}
    for rnd in range(3):
        for tk in TICKS:
            hip.check(L.lfm_set_option(3, tk), "opt")
            res.setdefault(tk, []).append(timeit(run, n=10))
    hip.check(L.lfm_set_option(3, 0), "opt")
    print(f"--- halo conv N={N} {H}x{Wd} {Cin}->{Cout}: " + "  ".join(f"{tk}: {statistics.median(v):7.1f} us ({flop / statistics.median(v) / 1e6:5.0f} TF)" for tk, v in res.items()), flush=True)

# ---- attention (two 64-KiB workgroups per CU, 1024 workgroups = two rounds): load phase of one under the compute phase of the other?
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); Kk = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
res = {}
for rnd in range(5):
    for tk in [0, 500, 1000, 2000, 3000, 4000, 6000, 8000]:
        hip.check(L.lfm_set_option(3, tk), "opt")
        res.setdefault(tk, []).append(timeit(lambda: hip.dit_attention(Q, Kk, Vt, Bh, heads, T), n=30))
hip.check(L.lfm_set_option(3, 0), "opt")
print("--- attention 64 x 16 x 256 x 64: " + "  ".join(f"{tk}: {statistics.median(v):6.1f} us" for tk, v in res.items()), flush=True)
