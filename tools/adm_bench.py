"""Timing of the origin-ADM UNet (BASELINE config 5: celeb512, nf 256, ch_mult 1 2 2 2 4, attn 16 8, batch 32, 64x64 latents)."""
import sys, time
from argparse import Namespace
import torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd.models import create_network
from lfm_amd.solvers import GraphedFixedGrid, torchdiffeq_euler_grid
from lfm_amd.test_flow_latent import dezero_
dev = torch.device("cuda:0")
args = Namespace(use_origin_adm=True, layout=False, model_type="adm", image_size=512, f=8, num_in_channels=4, num_out_channels=4, nf=256,
                 num_res_blocks=2, attn_resolutions=(16, 8), dropout=0.0, ch_mult=(1, 2, 2, 2, 4), resamp_with_conv=True, num_classes=None,
                 num_heads=4, num_head_channels=-1, num_head_upsample=-1)
torch.manual_seed(0)
m = dezero_(create_network(args)).to(dev).eval()
print("params M:", sum(p.numel() for p in m.parameters()) / 1e6)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(B, 4, 64, 64, device=dev)
t = torch.tensor(0.5, device=dev)
for _ in range(2): v = m(t, x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): v = m(t, x)
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 3
print(f"eager forward B={B}: {eager*1e3:.1f} ms  -> {189.7e9*B/eager/1e12:.0f} TFLOP/s", "finite" if torch.isfinite(v).all() else "NONFINITE", float(v.abs().mean()))
ts, dts = torchdiffeq_euler_grid(0.1)
s = GraphedFixedGrid(m, B, resolution=64); s.set_grid(ts, dts)
s.run(x); torch.cuda.synchronize(); t0 = time.perf_counter(); s.run(x); torch.cuda.synchronize(); g = (time.perf_counter() - t0) / 10
print(f"graph step B={B}: {g*1e3:.1f} ms/step -> {189.7e9*B/g/1e12:.0f} TFLOP/s => {B/(50*g):.1f} img/s at 50 NFE (no VAE)")
