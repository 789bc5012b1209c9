import sys, gc, torch
from argparse import Namespace
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import solvers
from lfm_amd.models import create_network
from lfm_amd.test_flow_latent import dezero_, sample_from_model
dev = torch.device("cuda:0"); torch.set_grad_enabled(False)
a = Namespace(use_origin_adm=True, layout=False, model_type="adm", image_size=128, f=8, num_in_channels=4, num_out_channels=4, nf=128, num_res_blocks=1,
              attn_resolutions=(4, 2), dropout=0.0, ch_mult=(1, 2, 2), resamp_with_conv=True, num_classes=None, num_heads=4, num_head_channels=-1, num_head_upsample=-1)
def fin(t): return bool(torch.isfinite(t).all())
for mode in ("V1 expr;sync", "V2 hold;sync;del", "V3 expr;nosync", "V4 hold;del;sync", "V1 again", "V6 expr;sync;x-clone-input"):
    gc.collect()  # the captured solvers live on the model object and die with it
    sa = Namespace(method="euler", step_size=0.02, perturb=False, compute_nfe=False, cfg_scale=1.0, atol=1e-5, rtol=1e-5)
    torch.manual_seed(0); m = dezero_(create_network(a)).to(dev).eval()
    x = torch.randn(8, 4, 16, 16, device=dev)
    def solve(): return sample_from_model(m, x, {}, sa)[-1]
    if mode.startswith("V1"):
        solve(); torch.cuda.synchronize()
    elif mode.startswith("V2"):
        r = solve(); torch.cuda.synchronize(); del r
    elif mode.startswith("V3"):
        solve()
    elif mode.startswith("V4"):
        r = solve(); del r; torch.cuda.synchronize()
    else:
        sample_from_model(m, x.clone(), {}, sa)[-1]; torch.cuda.synchronize()
    r2 = solve(); torch.cuda.synchronize()
    fg = list(m.__dict__['_fused_solvers'].values())[0]
    print(f"{mode:28s}: second finite={fin(r2)}  fg.x finite={fin(fg.x)} d tcur={float(fg.tcur):.3f} step={int(fg.step)}", flush=True)
    del m, fg
