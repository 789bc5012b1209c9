"""Split-K slices of the UNets' deep small-map 3x3 convolutions on the 256x256 kernel (round 5, gemm_kernel.h: splitk256_slices) against the 128x128 slices
(flag 65536), through lfm_conv3x3_f16_ws with the workspace the models pass.   Usage: python tools/splitk256_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfm_amd import hip

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
L = hip.lib()
# (N, H, W, Cin, Cout): EDM ffhq_adm at batch 64 (8x8: 768 channels, 4x4: 1024), ADM celeb512 at batch 32 (16x16 / 8x8: 512, 4x4: 1024)
SHAPES = [(64, 8, 8, 768, 768), (64, 8, 8, 1536, 768), (64, 8, 8, 1280, 768), (64, 8, 8, 512, 768), (64, 4, 4, 1024, 1024), (64, 4, 4, 2048, 1024),
          (32, 16, 16, 512, 512), (32, 16, 16, 1024, 512), (32, 8, 8, 512, 512), (32, 8, 8, 1024, 512), (32, 4, 4, 1024, 1024)]
for N, H, W, Cin, Cout in SHAPES:
    x = torch.randn(N * H * W, Cin, device=dev, dtype=torch.float16)
    w = (torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).half()
    b = torch.randn(Cout, device=dev)
    wsb = int(L.lfm_conv3x3_workspace_bytes(N, H, W, Cin, Cout))
    ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
    flop = 2.0 * N * H * W * Cout * 9 * Cin
    res, outs = {}, {}
    for name, flags in (("256x256 slices", 0), ("128x128 slices", 65536)):
        out = torch.empty(N * H * W, Cout, device=dev, dtype=torch.float16)
        hip.gemm_select(flags << 4)

        def run():
            hip.check(L.lfm_conv3x3_f16_ws(hip.ptr(x), hip.ptr(w), hip.ptr(b), None, hip.ptr(out), N, H, W, Cin, Cout, 0, hip.ptr(ws), wsb, hip.stream_ptr()), "conv")

        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / reps * 1e3
        outs[name] = out.clone()
        hip.gemm_select(0)
    d = float((outs["256x256 slices"].float() - outs["128x128 slices"].float()).abs().max())
    print(f"N={N:3d} {H:2d}x{W:2d} Cin={Cin:5d} Cout={Cout:4d} ({flop / 1e9:6.1f} GFLOP, ws {wsb >> 20} MiB): "
          + "  ".join(f"{k} {v:7.1f} us {flop / v / 1e6:6.1f} TF" for k, v in res.items()) + f"  max |diff| {d:.2e}", flush=True)
