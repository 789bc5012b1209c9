"""Errors of the DiT-L/2 forward (batch 8, scalar t: the t-path, softmax, GELU and LayerNorm statistics all in play) and of the VAE decode against the fp32 CPU
oracle, for the -ffast-math A/B (tools/fastmath_ab.sh)."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from oracle import dit_ref, vae_ref
from lfm_amd.models import DiT_models
from lfm_amd.autoencoder import AutoencoderKL
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double().cpu() - b.double()).norm() / b.double().norm())
kw = dict(num_classes=1, label_dropout=0.0)
cfg = dit_ref.DiTCfg.named("DiT-L/2", **kw); sd = dit_ref.make_dit_state(cfg, seed=1)
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, **kw); m.load_state_dict(sd, strict=True); m = m.to(dev).eval()
x = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(1))
for t in (0.999, 0.5, 0.013):
    tt = torch.tensor(t)
    print(f"DiT-L/2 t={t}: rel-L2 vs oracle {rel(m(tt.to(dev), x.to(dev)), dit_ref.dit_forward(sd, cfg, tt, x)):.3e}")
sdv = vae_ref.make_vae_state(seed=3); vae = AutoencoderKL(); vae.load_state_dict(sdv, strict=True); vae = vae.to(dev)
z = torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(2)) * 1.5
print(f"VAE decode 256x256: rel-L2 vs oracle {rel(vae.decode(z.to(dev)).sample, vae_ref.vae_decode(sdv, z)):.3e}")
