"""A/B of the 3x3 convolution kernels through the C ABI (lfm_conv3x3_f16): the halo-tiled direct kernel (default) against the implicit GEMM (flag 8388608)
on the shapes of the VAE decoder's last level and of the ADM UNet.  Usage: python tools/conv_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lfm_amd import hip

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
EXTRA = [int(f) for f in sys.argv[2:]]  # further debug flags to A/B on the halo kernel
dev = torch.device("cuda:0")
L = hip.lib()
SHAPES = [(64, 256, 256, 128, 128), (64, 256, 256, 256, 128), (64, 128, 128, 256, 256), (32, 64, 64, 256, 256), (32, 64, 64, 128, 128),
          (32, 512, 512, 128, 128), (64, 64, 64, 512, 512), (64, 32, 32, 512, 512), (32, 32, 32, 512, 512), (32, 32, 32, 768, 512),
          (32, 16, 16, 1024, 1024), (32, 16, 16, 1536, 1024)]
for N, H, W, Cin, Cout in SHAPES:
    if N * H * W * max(Cin, Cout) * 2 > 6e9:
        N = max(1, int(6e9 // (H * W * max(Cin, Cout) * 2)))
    x = torch.randn(N * H * W, Cin, device=dev, dtype=torch.float16)
    w = (torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).half()
    b = torch.zeros(Cout, device=dev)
    out = torch.empty(N * H * W, Cout, device=dev, dtype=torch.float16)
    flop = 2.0 * N * H * W * Cout * 9 * Cin
    res = {}
    for name, flags in (("implicit", 8388608), ("halo", 0)) + tuple((f"halo+flag{f}", f) for f in EXTRA):
        hip.gemm_select(flags << 4)
        for _ in range(2):
            hip.check(L.lfm_conv3x3_f16(hip.ptr(x), hip.ptr(w), hip.ptr(b), None, hip.ptr(out), N, H, W, Cin, Cout, 0, hip.stream_ptr()), "conv")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hip.check(L.lfm_conv3x3_f16(hip.ptr(x), hip.ptr(w), hip.ptr(b), None, hip.ptr(out), N, H, W, Cin, Cout, 0, hip.stream_ptr()), "conv")
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / reps * 1e3
        hip.gemm_select(0)
    print(f"N={N:3d} {H}x{W} Cin={Cin} Cout={Cout}: " + "  ".join(f"{k} {v:8.1f} us {flop / v / 1e6:7.1f} TFLOP/s" for k, v in res.items()), flush=True)
