"""origin-ADM celeb512 configuration (352 M parameters), ONE latent: the HIP model with each GEMM kernel forced vs the CPU oracle.
(The parity tests use small UNets, whose convolutions never reach the 256x256 kernels through the automatic selection.)"""
import sys, time, torch
from argparse import Namespace
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import create_network
from oracle import unet_ref
dev = torch.device("cuda:0"); torch.set_grad_enabled(False)
args = Namespace(use_origin_adm=True, layout=False, model_type="adm", image_size=512, f=8, num_in_channels=4, num_out_channels=4, nf=256, num_res_blocks=2,
                 attn_resolutions=(16, 8), dropout=0.0, ch_mult=(1, 2, 2, 2, 4), resamp_with_conv=True, num_classes=None, num_heads=4, num_head_channels=-1,
                 num_head_upsample=-1)
cfg = dict(image_size=64, in_channels=4, model_channels=256, out_channels=4, num_res_blocks=2, attention_resolutions=(16, 8), channel_mult=(1, 2, 2, 2, 4),
           num_classes=None, num_heads=4, num_head_channels=-1, num_heads_upsample=-1)
sd = unet_ref.make_unet_state(cfg, seed=3)
m = create_network(args); m.load_state_dict(sd, strict=True); m = m.to(dev).eval()
g = torch.Generator().manual_seed(5)
x0 = torch.randn(1, 4, 64, 64, generator=g); t = torch.tensor([0.7])
t0 = time.time(); ref = unet_ref.unet_forward(sd, cfg, t, x0); print(f"oracle {time.time()-t0:.1f} s, |ref| mean {ref.abs().mean():.4f} max {ref.abs().max():.3f}", flush=True)
def rel(a, b): return float((a.float().cpu() - b).norm() / b.norm())
for name, sel in [("auto", 0), ("v1", 1), ("v2", 2), ("v3", 3)]:
    hip.gemm_select(sel)
    out = m(t.to(dev), x0.to(dev)); torch.cuda.synchronize()
    print(f"{name:5s}: finite={bool(torch.isfinite(out).all())} rel-L2 vs oracle {rel(out, ref):.3e}", flush=True)
hip.gemm_select(0)
x = torch.randn(4, 4, 64, 64, generator=g).to(dev)
for k in range(5):
    v = m(torch.full((4,), 1.0 - 0.2 * k, device=dev), x); x = x - 0.2 * v
    print(f"euler step {k}: |v| max {float(v.abs().max()):.3f} finite={bool(torch.isfinite(v).all())}", flush=True)
