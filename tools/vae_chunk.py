import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd.autoencoder import AutoencoderKL, images_to_uint8
dev = torch.device("cuda:0"); torch.set_grad_enabled(False)
z = torch.randn(64, 4, 32, 32, device=dev)
for chunk in (8, 16, 32, 64):
    vae = AutoencoderKL.from_random(seed=0, decode_chunk=chunk).to(dev)
    for _ in range(2): images_to_uint8(vae.decode(z).sample)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): images_to_uint8(vae.decode(z).sample)
    torch.cuda.synchronize(); print(f"chunk {chunk}: {(time.perf_counter()-t0)/3*1e3:.1f} ms / 64 images", flush=True)
    del vae; torch.cuda.empty_cache()
