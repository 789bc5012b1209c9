"""VAE decode of 64 latents (32x32 -> 256x256) + uint8 conversion as a function of decode_chunk (images per pass through the decoder): smaller chunks keep the
full-resolution activations (chunk x 65536 pixels x 128 channels x 2 B = 16.8 MB per image) inside the 256 MB Infinity Cache between the kernels that
write and read them.  Interleaved medians.  usage: python tools/vae_chunk_probe.py [R]"""
import statistics, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd.autoencoder import AutoencoderKL, images_to_uint8
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 64 if R == 32 else 32
z = torch.randn(N, 4, R, R, device=dev)
chunks = [c for c in (2, 4, 8, 16, 32, 64) if c <= N]
vaes = {}
for c in chunks:
    v = AutoencoderKL.from_random(seed=0).to(dev); v.decode_chunk = c; vaes[c] = v
def timeit(fn, n=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
res = {c: [] for c in chunks}
ref = vaes[chunks[-1]].decode(z).sample
for c in chunks:
    print(f"chunk {c}: output identical to chunk {chunks[-1]}: {bool(torch.equal(vaes[c].decode(z).sample, ref))}", flush=True)
for rnd in range(3):
    for c in chunks: res[c].append(timeit(lambda: images_to_uint8(vaes[c].decode(z).sample)))
for c in chunks: print(f"R={R} N={N} decode_chunk {c:3d}: median {statistics.median(res[c]):7.2f} ms  min {min(res[c]):7.2f} ms", flush=True)
