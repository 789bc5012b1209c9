"""Attention alone, 256 tokens x hd 64: the persistent streamed kernel (LFM_OPT_ATTENTION_STREAM, csrc/attention_stream_kernel.h) against one workgroup per item
(csrc/attention_kernel.h), interleaved medians per shape.  The operands are re-written between the timed loops by a 300 MB copy so that neither variant finds
them in the Infinity Cache.  usage: python tools/attn_stream_ab.py [rounds]"""
import statistics, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
flush_a = torch.empty(300 << 20, dtype=torch.uint8, device=dev); flush_b = torch.empty_like(flush_a)
def timeit(fn, n=20, warm=2, flush=False):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        if flush: flush_b.copy_(flush_a)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return statistics.median(ts)
def loop(fn, n=50, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for Bh, heads in ((64, 16), (32, 16), (512, 12), (8, 16), (171, 6)):
    T = 256
    Q = torch.randn(Bh * T, heads * 64, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
    res = {0: [], 1: []}; resf = {0: [], 1: []}
    for rnd in range(rounds):
        for opt in (0, 1):
            hip.set_option(hip.OPT_ATTENTION_STREAM, opt)
            res[opt].append(loop(lambda: hip.dit_attention(Q, K, Vt, Bh, heads, T)))
            resf[opt].append(timeit(lambda: hip.dit_attention(Q, K, Vt, Bh, heads, T), flush=True))
    hip.set_option(hip.OPT_ATTENTION_STREAM, 1)
    mb = 4 * Bh * T * heads * 64 * 2 / 1e6
    for opt in (0, 1):
        m, f = statistics.median(res[opt]), statistics.median(resf[opt])
        print(f"attention {Bh} x {heads} x 256 x 64 ({Bh * heads} items, {mb:.0f} MB), stream {opt}: back-to-back median {m:.1f} us (min {min(res[opt]):.1f}) = {mb / m / 1e6 * 1e6 / 1e3:.2f} TB/s;"
              f"  single launch after a cache flush (event pair, incl. ~launch overhead) {f:.1f} us")
