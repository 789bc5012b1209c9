"""VAE decode at 512x512 (R = 64): does it leave its input alone, and do different chunk sizes agree?"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd.autoencoder import AutoencoderKL
dev = torch.device("cuda:0"); torch.set_grad_enabled(False)
vae = AutoencoderKL.from_random(seed=0).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn(32, 4, 64, 64, device=dev, generator=g)
guard = torch.randn(1 << 20, device=dev, generator=g); guard0 = guard.clone(); lat0 = lat.clone()
print("decode_chunk default:", getattr(vae, "decode_chunk", None), flush=True)
img = vae.decode(lat / 0.18215).sample
torch.cuda.synchronize()
print(f"after decode: lat unchanged={bool(torch.equal(lat, lat0))} guard unchanged={bool(torch.equal(guard, guard0))} img finite={bool(torch.isfinite(img).all())} |img| max {float(img.abs().max()):.3f}", flush=True)
for ch in (1, 4):
    vae.decode_chunk = ch
    im2 = vae.decode(lat[:8] / 0.18215).sample
    print(f"chunk {ch}: rel diff vs default on the first 8 images {float((im2 - img[:8]).norm() / img[:8].norm()):.3e}", flush=True)
