import sys, gc, torch
from argparse import Namespace
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import solvers
from lfm_amd.models import create_network
from lfm_amd.test_flow_latent import dezero_, sample_from_model
dev = torch.device("cuda:0"); torch.set_grad_enabled(False)
a = Namespace(use_origin_adm=True, layout=False, model_type="adm", image_size=128, f=8, num_in_channels=4, num_out_channels=4, nf=128, num_res_blocks=1,
              attn_resolutions=(4, 2), dropout=0.0, ch_mult=(1, 2, 2), resamp_with_conv=True, num_classes=None, num_heads=4, num_head_channels=-1, num_head_upsample=-1)
sa = Namespace(method="euler", step_size=0.02, perturb=False, compute_nfe=False, cfg_scale=1.0, atol=1e-5, rtol=1e-5)
def fin(t): return bool(torch.isfinite(t).all())
for mode in ("drop", "drop+occupy", "keep"):
    solvers._FUSED_CACHE.clear()
    torch.manual_seed(0); m = dezero_(create_network(a)).to(dev).eval()
    x = torch.randn(8, 4, 16, 16, device=dev)
    r1 = sample_from_model(m, x, {}, sa)[-1]
    fg = list(solvers._FUSED_CACHE.values())[0]
    g1, gen1 = id(fg.graphs.get("euler")), (m._gen, fg._graph_gen)
    f1 = fin(r1)
    if mode != "keep":
        del r1
    torch.cuda.synchronize()
    if mode == "drop+occupy":
        dummy = [torch.empty(2, 8, 4, 16, 16, device=dev) for _ in range(4)]
    r2 = sample_from_model(m, x, {}, sa)[-1]
    torch.cuda.synchronize()
    print(f"{mode:12s}: first finite={f1} second finite={fin(r2)} recaptured={id(fg.graphs.get('euler')) != g1} gen before {gen1} after {(m._gen, fg._graph_gen)} "
          f"fg.x finite={fin(fg.x)} tcur={float(fg.tcur):.4f} dt={float(fg.dt):.4f} step={int(fg.step)}", flush=True)
    del m, fg; gc.collect()
