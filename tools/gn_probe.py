"""GroupNorm(32) + SiLU on the small / mid maps of the UNets: the fused one-launch kernel (gn_fused_kernel) against the three-kernel path (select flag 16384),
per shape, interleaved medians.  usage: python tools/gn_probe.py"""
import statistics, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
L = hip.lib()
def timeit(fn, n=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
shapes = [(64, 1024, 256), (64, 1024, 512), (64, 256, 512), (64, 256, 768), (64, 256, 1280), (64, 64, 768), (64, 64, 1536), (64, 16, 1024), (64, 16, 2048),
          (32, 1024, 512), (32, 256, 512), (32, 256, 1024), (32, 64, 1024), (32, 64, 2048)]
for N, HW, C in shapes:
    x = torch.randn(N * HW, C, device=dev).half(); y = torch.empty_like(x)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    scratch = torch.empty(max(int(L.lfm_groupnorm_scratch_bytes(N, C)), 1 << 20), dtype=torch.uint8, device=dev)
    groups = 32
    run = lambda: hip.check(L.lfm_groupnorm_f16(hip.ptr(x), hip.ptr(y), hip.ptr(g), hip.ptr(b), None, 0, hip.ptr(scratch), N, HW, C, groups, 1e-5, 1, hip.stream_ptr(dev)), "gn")
    res = {0: [], 16384: []}
    outs = {}
    for f in (0, 16384):
        hip.gemm_select(f << 4); run(); outs[f] = y.clone()
    for rnd in range(3):
        for f in (0, 16384):
            hip.gemm_select(f << 4); res[f].append(timeit(run))
    hip.gemm_select(0)
    mb = 2 * N * HW * C * 2 / 1e6
    print(f"N={N:3d} HW={HW:5d} C={C:5d} ({mb:7.1f} MB r+w): fused {statistics.median(res[0]):7.1f} us   three kernels {statistics.median(res[16384]):7.1f} us   max |diff| {float((outs[0].float() - outs[16384].float()).abs().max()):.2e}", flush=True)
