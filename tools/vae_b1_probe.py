"""Batch-1 VAE decode (what --measure_time's run_sampling(1, ...) ends with, reference test_flow_latent.py:223-246): loop target for rocprofv3 + event timing.
usage: python tools/vae_b1_probe.py [reps]"""
import statistics, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd.autoencoder import AutoencoderKL
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
vae = AutoencoderKL.from_random(seed=0).to(dev)
z = torch.randn(1, 4, 32, 32, device=dev)
for _ in range(3): vae.decode(z / 0.18215).sample
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); vae.decode(z / 0.18215).sample; e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
print(f"batch-1 VAE decode 4x32x32 -> 3x256x256: median {statistics.median(ts) * 1e3:.0f} us (min {min(ts) * 1e3:.0f}) = {622.2 / statistics.median(ts):.0f} TFLOP/s of 622.2 GFLOP")
