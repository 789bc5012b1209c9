"""The UNets' 3x3 convolutions on 16x16 (and smaller) maps: the split-K implicit GEMM they take today (fewer than 256 halo tiles) against the
halo-tiled direct kernel forced onto the same shapes (flag 16777216), through lfm_conv3x3_f16_ws with the workspace the models pass.
Usage: python tools/conv_small_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lfm_amd import hip

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
L = hip.lib()
# (N, H, W, Cin, Cout): EDM ffhq-64 / imagenet-64 at batch 64, ADM celeb256 / celeb512 at batch 32
SHAPES = [(64, 16, 16, 256, 256), (64, 16, 16, 512, 256), (64, 16, 16, 384, 384), (64, 16, 16, 768, 384), (64, 32, 32, 256, 256),
          (32, 16, 16, 512, 512), (32, 16, 16, 1024, 512), (32, 16, 16, 768, 512), (32, 32, 32, 256, 256), (32, 32, 32, 512, 256),
          (64, 8, 8, 256, 256), (64, 8, 8, 512, 256), (64, 8, 8, 768, 768), (32, 8, 8, 768, 768), (32, 8, 8, 1536, 768)]
for N, H, W, Cin, Cout in SHAPES:
    x = torch.randn(N * H * W, Cin, device=dev, dtype=torch.float16)
    w = (torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).half()
    b = torch.randn(Cout, device=dev)
    wsb = int(L.lfm_conv3x3_workspace_bytes(N, H, W, Cin, Cout))
    ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
    flop = 2.0 * N * H * W * Cout * 9 * Cin
    res, outs = {}, {}
    for name, flags in (("today", 0), ("implicit", 8388608), ("halo forced", 16777216)):
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
        outs[name] = out
        hip.gemm_select(0)
    d = float((outs["halo forced"].float() - outs["implicit"].float()).abs().max())
    print(f"N={N:3d} {H:2d}x{W:2d} Cin={Cin:5d} Cout={Cout:4d} ({flop / 1e9:6.1f} GFLOP, ws {wsb >> 20} MiB): "
          + "  ".join(f"{k} {v:7.1f} us {flop / v / 1e6:6.1f} TF" for k, v in res.items()) + f"  max |halo - implicit| {d:.2e}", flush=True)
