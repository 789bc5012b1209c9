"""Graphed fixed-grid solve on the host-sequenced UNet: repeated calls vs the eager loop (config-5 style usage)."""
import sys, torch
from argparse import Namespace
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd.models import create_network
from lfm_amd.test_flow_latent import dezero_, sample_from_model
dev = torch.device("cuda:0"); torch.set_grad_enabled(False)
big = len(sys.argv) > 1 and sys.argv[1] == "big"
if len(sys.argv) > 2 and sys.argv[2] == "vae":  # config_bench order: a VAE (and its weights / workspace) exists before the UNet
    from lfm_amd.autoencoder import AutoencoderKL
    vae = AutoencoderKL.from_random(seed=0).to(dev)
if big:
    a = Namespace(use_origin_adm=True, layout=False, model_type="adm", image_size=512, f=8, num_in_channels=4, num_out_channels=4, nf=256, num_res_blocks=2,
                  attn_resolutions=(16, 8), dropout=0.0, ch_mult=(1, 2, 2, 2, 4), resamp_with_conv=True, num_classes=None, num_heads=4, num_head_channels=-1, num_head_upsample=-1)
    B, R = 32, 64
else:
    a = Namespace(use_origin_adm=True, layout=False, model_type="adm", image_size=128, f=8, num_in_channels=4, num_out_channels=4, nf=128, num_res_blocks=1,
                  attn_resolutions=(4, 2), dropout=0.0, ch_mult=(1, 2, 2), resamp_with_conv=True, num_classes=None, num_heads=4, num_head_channels=-1, num_head_upsample=-1)
    B, R = 8, 16
torch.manual_seed(0); m = dezero_(create_network(a)).to(dev).eval()
x = torch.randn(B, 4, R, R, device=dev)
sa = Namespace(method="euler", step_size=0.02, perturb=False, compute_nfe=False, cfg_scale=1.0, atol=1e-5, rtol=1e-5)
outs = [sample_from_model(m, x, {}, sa)[-1].clone() for _ in range(3)]
raw = sample_from_model(m, x, {}, sa)[-1]
print('un-cloned result finite:', bool(torch.isfinite(raw).all()), flush=True)
sa.fused = False
eager = sample_from_model(m, x, {}, sa)[-1]
def rel(a, b): return float((a - b).norm() / b.norm())
for i, o in enumerate(outs):
    print(f"graphed call {i}: finite={bool(torch.isfinite(o).all())} |x| max {float(o.abs().max()):.3e} rel vs eager {rel(o, eager):.3e}", flush=True)
print(f"eager: finite={bool(torch.isfinite(eager).all())} |x| max {float(eager.abs().max()):.3e}")
