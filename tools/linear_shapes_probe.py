"""The UNets' 1x1 convolutions / linears (lfm_linear_f16) per shape on each GEMM kernel: lfm_gemm_select 0 = automatic choice, 1 = 128x128 (v1), 4 = 256x128 (two
workgroups per CU), 5 = 256x256 (16x16x32 MFMAs).  usage: linear_shapes_probe.py [reps]"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
shapes = [(65536, 768, 256), (65536, 256, 256), (16384, 1152, 384), (16384, 384, 384), (16384, 384, 768), (16384, 512, 512), (16384, 512, 1024), (4096, 1536, 512), (4096, 512, 512), (4096, 512, 1024),
          (8192, 512, 512), (8192, 1024, 1024), (2048, 1024, 1024), (32768, 256, 512)]
for (M, N, K) in shapes:
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev)
    R = (torch.randn(M, N, device=dev)).half(); C = torch.empty(M, N, device=dev, dtype=torch.float16)
    line = f"M {M:6d} N {N:5d} K {K:5d} ({2e-9 * M * N * K:6.1f} GFLOP): "
    ref = None
    for sel in (0, 1, 4, 5):
        hip.gemm_select(sel)
        try:
            for _ in range(3): hip.check(hip.lib().lfm_linear_f16(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(C), N, M, N, K, hip.ptr(b), hip.ptr(R), hip.stream_ptr()), "linear")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): hip.check(hip.lib().lfm_linear_f16(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(C), N, M, N, K, hip.ptr(b), hip.ptr(R), hip.stream_ptr()), "linear")
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            if ref is None: ref = C.clone()
            same = bool(torch.equal(ref, C))
            line += f" sel {sel}: {us:7.1f} us{'' if same else ' (differs ' + format(float((ref.float() - C.float()).abs().max()), '.2e') + ')'} |"
        except Exception as ex:
            line += f" sel {sel}: n/a |"
    hip.gemm_select(0)
    print(line)
