"""Is every kernel's result independent of what else the GPU is running?  A DiT-L/2 evaluation (64 images, folded LayerNorm path) and a VAE decode are run
alone and then while a twin of the other / the same kind runs on a second stream; outputs must be bit-identical.  A difference = a kernel whose workgroups
depend on each other's timing (an intra-launch race), which a solo run never shows.  usage: python tools/concurrency_check.py [reps]"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.autoencoder import AutoencoderKL
from lfm_amd.models import DiT_models
from lfm_amd.solvers import concurrency_twin
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in m.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
m = m.to(dev).eval()
vae = AutoencoderKL.from_random(seed=0).to(dev)
x = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev); z = torch.randn(16, 4, 32, 32, device=dev)
ref_v = m(t, x).clone(); ref_i = vae.decode(z).sample.clone()
assert torch.equal(ref_v, m(t, x)) and torch.equal(ref_i, vae.decode(z).sample)
m2, vae2 = concurrency_twin(m), concurrency_twin(vae)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
torch.cuda.synchronize()
def trial(fa, fb, ref):
    bad = 0
    for _ in range(reps):
        cur = torch.cuda.current_stream(dev); sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sb):
            for _ in range(3): fb()
        with torch.cuda.stream(sa):
            out = fa()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            d = (out.float() - ref.float()).abs()
            print(f"      mismatch: {int((d > 0).sum())} elements differ, max |diff| {float(d.max()):.3e}, first index {int((d.flatten() > 0).nonzero()[0])}", flush=True)
    return bad
print("DiT evaluation under a second DiT evaluation :", trial(lambda: m(t, x), lambda: m2(t, x), ref_v), "of", reps, "differ", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "dit":  # narrow the DiT case down: which path is timing-sensitive?
    for name, setup in (("separate LayerNorm launches (fold off)", lambda: hip.set_option(hip.OPT_FOLD_LN, 0)), ("fold on, kernel 6", lambda: (hip.set_option(hip.OPT_FOLD_LN, 1), hip.set_option(hip.OPT_GEMM_V6, 1))),
                        ("fold on, kernel 5 again", lambda: hip.set_option(hip.OPT_GEMM_V6, 0))):
        setup()
        r = m(t, x).clone()
        assert torch.equal(r, m(t, x))
        print(f"  {name:42s}:", trial(lambda: m(t, x), lambda: m2(t, x), r), "of", reps, "differ", flush=True)
    hip.set_option(hip.OPT_FOLD_LN, 1); hip.set_option(hip.OPT_GEMM_V6, 0)
    # a smaller companion load: does ANY concurrent kernel do it?
    y = torch.randn(4096, 4096, device=dev)
    print("  DiT evaluation under torch matmuls          :", trial(lambda: m(t, x), lambda: (y @ y).sum(), ref_v), "of", reps, "differ", flush=True)
    sys.exit(0)
print("DiT evaluation under a VAE decode            :", trial(lambda: m(t, x), lambda: vae2.decode(z).sample, ref_v), "of", reps, "differ", flush=True)
print("VAE decode under a DiT evaluation            :", trial(lambda: vae.decode(z).sample, lambda: m2(t, x), ref_i), "of", reps, "differ", flush=True)
print("VAE decode under a second VAE decode         :", trial(lambda: vae.decode(z).sample, lambda: vae2.decode(z).sample, ref_i), "of", reps, "differ", flush=True)
