"""Mimic tools/config_bench.py's T(): first graphed solve dropped, synchronize, second solve checked."""
import sys, gc, torch
from argparse import Namespace
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd.models import create_network
from lfm_amd.test_flow_latent import dezero_, sample_from_model
dev = torch.device("cuda:0"); torch.set_grad_enabled(False)
def cfg(big):
    if big:
        return Namespace(use_origin_adm=True, layout=False, model_type="adm", image_size=512, f=8, num_in_channels=4, num_out_channels=4, nf=256, num_res_blocks=2,
                         attn_resolutions=(16, 8), dropout=0.0, ch_mult=(1, 2, 2, 2, 4), resamp_with_conv=True, num_classes=None, num_heads=4, num_head_channels=-1, num_head_upsample=-1), 32, 64
    return Namespace(use_origin_adm=True, layout=False, model_type="adm", image_size=128, f=8, num_in_channels=4, num_out_channels=4, nf=128, num_res_blocks=1,
                     attn_resolutions=(4, 2), dropout=0.0, ch_mult=(1, 2, 2), resamp_with_conv=True, num_classes=None, num_heads=4, num_head_channels=-1, num_head_upsample=-1), 8, 16
for big in ([False, True] if len(sys.argv) > 1 else [False]):
    a, B, R = cfg(big)
    torch.manual_seed(0); m = dezero_(create_network(a)).to(dev).eval()
    x = torch.randn(B, 4, R, R, device=dev)
    sa = Namespace(method="euler", step_size=0.02, perturb=False, compute_nfe=False, cfg_scale=1.0, atol=1e-5, rtol=1e-5)
    def solve(): return sample_from_model(m, x, {}, sa)[-1]
    solve(); torch.cuda.synchronize()
    out = solve(); torch.cuda.synchronize()
    print(f"big={big}: drop-first + sync: second result finite={bool(torch.isfinite(out).all())} nonfinite={int((~torch.isfinite(out)).sum())}", flush=True)
    out3 = solve()
    print(f"          third (no sync before): finite={bool(torch.isfinite(out3).all())}", flush=True)
    sa.fused = False
    eager = sample_from_model(m, x, {}, sa)[-1]
    print(f"          eager finite={bool(torch.isfinite(eager).all())}; rel third vs eager {float((out3 - eager).norm() / eager.norm()):.2e}", flush=True)
    del m; gc.collect()
