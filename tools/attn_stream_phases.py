"""Phase split of the streamed attention kernel on a measurement build (VARDIR=tools/ship_variants tools/build_variant.sh measure "-DLFM_MEASURE";
LFM_HIP_LIBRARY=tools/ship_variants/measure/liblfm_hip.so): whole kernel, memory only, compute only, compute only without stage barriers, each with two and with
one persistent workgroup per CU; the per-item kernel's modes next to it.  usage: python tools/attn_stream_phases.py"""
import os, statistics, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
def loop(fn, n=50, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
Bh, heads, T = 64, 16, 256
Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
names = {0: "whole kernel", 1: "memory only", 2: "compute only", 3: "compute only, no stage barriers / waits"}
for stream in (1, 0):
    hip.set_option(hip.OPT_ATTENTION_STREAM, stream)
    for per_cu in ((2, 1) if stream else (2,)):
        os.environ["LFM_ATS_WG_PER_CU"] = str(per_cu)
        for mode in (0, 1, 2, 3):
            if mode == 3 and not stream: continue
            res = []
            for _ in range(5):
                hip.gemm_select((mode << 25) << 4)
                res.append(loop(lambda: hip.dit_attention(Q, K, Vt, Bh, heads, T)))
            hip.gemm_select(0)
            print(f"{'streamed' if stream else 'per-item'} kernel, {per_cu} workgroup(s) per CU, {names[mode]:40s}: median {statistics.median(res):6.1f} us  min {min(res):6.1f}")
hip.set_option(hip.OPT_ATTENTION_STREAM, 1)
