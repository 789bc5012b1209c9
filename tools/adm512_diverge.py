"""config 5 with the bench's synthetic weights (default init + dezero): where do the latents stop being finite?"""
import sys, torch
from argparse import Namespace
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd.models import create_network
from lfm_amd.test_flow_latent import dezero_
dev = torch.device("cuda:0"); torch.set_grad_enabled(False)
a = Namespace(use_origin_adm=True, layout=False, model_type="adm", image_size=512, f=8, num_in_channels=4, num_out_channels=4, nf=256, num_res_blocks=2,
              attn_resolutions=(16, 8), dropout=0.0, ch_mult=(1, 2, 2, 2, 4), resamp_with_conv=True, num_classes=None, num_heads=4, num_head_channels=-1, num_head_upsample=-1)
torch.manual_seed(0); m = dezero_(create_network(a)).to(dev).eval()
x = torch.randn(4, 4, 64, 64, device=dev)
for k in range(50):
    t = torch.full((4,), 1.0 - 0.02 * k, device=dev)
    v = m(t, x); x = x - 0.02 * v
    if k < 3 or k % 10 == 9 or not torch.isfinite(v).all():
        print(f"step {k}: |v| max {float(v.abs().max()):.3e} |x| max {float(x.abs().max()):.3e} finite v={bool(torch.isfinite(v).all())}", flush=True)
    if not torch.isfinite(v).all(): break
