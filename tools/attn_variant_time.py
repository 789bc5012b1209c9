"""One library build (LFM_HIP_LIBRARY), attention alone at 64 x 16 x 256 x 64: median of back-to-back loops + bit equality with the per-item kernel.
usage: LFM_HIP_LIBRARY=... python tools/attn_variant_time.py [tag]"""
import statistics, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else "?"
def loop(fn, n=50, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
out = []
for Bh, heads in ((64, 16), (512, 12)):
    T = 256
    g = torch.Generator(device=dev).manual_seed(1)
    Q = torch.randn(Bh * T, heads * 64, device=dev, generator=g).half(); K = torch.randn(Bh * T, heads * 64, device=dev, generator=g).half()
    Vt = torch.randn(Bh, heads, 64, T, device=dev, generator=g).half()
    hip.set_option(hip.OPT_ATTENTION_STREAM, 0); ref = hip.dit_attention(Q, K, Vt, Bh, heads, T)
    hip.set_option(hip.OPT_ATTENTION_STREAM, 1); got = hip.dit_attention(Q, K, Vt, Bh, heads, T)
    eq = bool(torch.equal(ref, got))
    res = [loop(lambda: hip.dit_attention(Q, K, Vt, Bh, heads, T)) for _ in range(7)]
    out.append(f"{Bh}x{heads}: {statistics.median(res):6.1f} us (min {min(res):6.1f}) equal={eq}")
print(f"{tag:8s} " + "   ".join(out))
