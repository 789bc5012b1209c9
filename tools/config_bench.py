"""Timings of BASELINE.json configs 3, 4, 5 on one MI355X (synthetic weights); config 2 is bench.py."""
import sys, time
from argparse import Namespace
import torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd.autoencoder import AutoencoderKL, images_to_uint8
from lfm_amd.models import create_network
from lfm_amd.sampler.karras_sample import karras_sample
from lfm_amd.solvers import odeint
from lfm_amd.test_flow_latent import dezero_, sample_from_model
dev = torch.device("cuda:0"); torch.set_grad_enabled(False)
which = sys.argv[1:] or ["3", "4", "5"]
def T(fn, n=1):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, out
vae = AutoencoderKL.from_random(seed=0).to(dev)
if "3" in which:  # DiT-L/2, dopri5 rtol=atol=1e-5, CFG on (class-conditional variant so the guidance is real), 64 live + 64 null
    a = Namespace(use_origin_adm=False, model_type="DiT-L/2", image_size=256, f=8, num_in_channels=4, label_dropout=0.1, num_classes=1000)
    torch.manual_seed(0); m = dezero_(create_network(a)).to(dev).eval()
    B = 64; x = torch.randn(B, 4, 32, 32, device=dev); x = torch.cat([x, x]); y = torch.cat([torch.randint(0, 1000, (B,), device=dev), torch.full((B,), 1000, device=dev)])
    st = {}
    def run():
        st.clear()
        lat = odeint(lambda t, xx: m.forward_with_cfg(t, xx, y, cfg_scale=1.5), x, torch.tensor([1.0, 0.0], device=dev), method="dopri5", rtol=1e-5, atol=1e-5, stats=st)[-1][:B]
        return images_to_uint8(vae.decode(lat / 0.18215).sample)
    dt, _ = T(run)
    print(f"config3 DiT-L/2 dopri5(1e-5) CFG1.5 B=64(+64 null): {dt:.2f} s/batch, NFE={st['nfe']} steps={st['steps']} accepted={st['accepted']} -> {B/dt:.2f} img/s", flush=True)
    del m
if "4" in which:  # DiT-B/2 imnet class-conditional, batch 256 (+256 null under CFG 1.5), 50-grid Heun (reference quirk: 88 NFE)
    a = Namespace(use_origin_adm=False, model_type="DiT-B/2", image_size=256, f=8, num_in_channels=4, label_dropout=0.1, num_classes=1000)
    torch.manual_seed(0); m = dezero_(create_network(a)).to(dev).eval()
    B = 256; x = torch.randn(B, 4, 32, 32, device=dev); x = torch.cat([x, x]); y = torch.cat([torch.randint(0, 1000, (B,), device=dev), torch.full((B,), 1000, device=dev)])
    def run():
        lat = karras_sample(m, x, steps=50, model_kwargs=dict(y=y, cfg_scale=1.5), device=dev, clip_denoised=False, sigma_min=1e-5, sigma_max=1.0, s_tmin=0.0, s_tmax=1.0, s_churn=0.0, sampler="heun")[:B]
        return images_to_uint8(vae.decode(lat / 0.18215).sample)
    dt, _ = T(run)
    print(f"config4 DiT-B/2 cls-cond CFG1.5 Heun(50 grid => 88 NFE) B=256(+256 null): {dt:.2f} s/batch -> {B/dt:.1f} img/s", flush=True)
    del m
if "5" in which:  # origin-ADM celeb512, batch 32, 50-step Euler + VAE at 512x512
    a = Namespace(use_origin_adm=True, layout=False, model_type="adm", image_size=512, f=8, num_in_channels=4, num_out_channels=4, nf=256, num_res_blocks=2,
                  attn_resolutions=(16, 8), dropout=0.0, ch_mult=(1, 2, 2, 2, 4), resamp_with_conv=True, num_classes=None, num_heads=4, num_head_channels=-1, num_head_upsample=-1)
    torch.manual_seed(0); m = dezero_(create_network(a)).to(dev).eval()
    B = 32; x = torch.randn(B, 4, 64, 64, device=dev)
    sa = Namespace(method="euler", step_size=0.02, perturb=False, compute_nfe=False, cfg_scale=1.0, atol=1e-5, rtol=1e-5)
    def solve(): return sample_from_model(m, x, {}, sa)[-1]
    ds, lat = T(solve)
    print('   latents right after the solve: finite =', bool(torch.isfinite(lat).all()), 'nonfinite count', int((~torch.isfinite(lat)).sum()), 'shape', tuple(lat.shape), flush=True)
    def dec(): return images_to_uint8(vae.decode(lat / 0.18215).sample)
    dd, img = T(dec)
    print(f"config5 ADM celeb512 B=32 50-step Euler: solver {ds:.2f} s + VAE512 decode {dd*1e3:.0f} ms -> {B/(ds+dd):.2f} img/s; img {tuple(img.shape)} finite={bool(torch.isfinite(lat).all())}", flush=True)
