"""Single operators under a co-scheduled load: attention (shipped 8-wave shape and the wide one), the QKV GEMM with its V^T split, the gated-residual and
GELU GEMMs, LN-modulate -- each repeated many times while a DiT-L/2 forward runs on a second stream; every result must equal the solo one bit for bit.
usage: python tools/concurrency_ops.py [reps]"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from lfm_amd import hip
from lfm_amd.models import DiT_models
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
big = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
for p in big.parameters():
    if not bool(p.any()): torch.nn.init.normal_(p, std=0.02)
big = big.to(dev).eval()
xb = torch.randn(64, 4, 32, 32, device=dev); t = torch.tensor(0.5, device=dev)
Bh, heads, T = 64, 16, 256
M = Bh * T
Q = torch.randn(M, 1024, device=dev).half(); K = torch.randn_like(Q); Vt = torch.randn(Bh, heads, 64, T, device=dev).half()
A = (torch.randn(M, 1024, device=dev) * 0.5).half(); A4 = (torch.randn(M, 4096, device=dev) * 0.5).half()
Wq = (torch.randn(3072, 1024, device=dev) * 0.03).half(); bq = torch.randn(3072, device=dev)
W1 = (torch.randn(4096, 1024, device=dev) * 0.03).half(); b1 = torch.randn(4096, device=dev)
W2 = (torch.randn(1024, 4096, device=dev) * 0.03).half(); b2 = torch.randn(1024, device=dev)
X0 = torch.randn(M, 1024, device=dev); gate = torch.randn(Bh, 1024, device=dev); sh = torch.randn(1, 1024, device=dev); sc = torch.randn(1, 1024, device=dev)
def resid():
    X = X0.clone()
    return hip.gemm_f16(A4, W2, b2, epilogue=3, out=X, gate=gate, gate_stride=1024, tokens=256)
ops = {"attention (8 waves x 32 queries)": lambda: hip.dit_attention(Q, K, Vt, Bh, heads, T),
       "qkv GEMM + V^T split": lambda: torch.cat([o.reshape(-1) for o in hip.gemm_qkv_f16(A, Wq, bq, 64, 256)]),
       "fc1 GEMM + GELU": lambda: hip.gemm_f16(A, W1, b1, epilogue=1),
       "fc2 GEMM + gated residual": resid,
       "LN-modulate": lambda: hip.ln_modulate(X0, sh, sc, T, 0)}
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for name, fn in ops.items():
    ref = fn().clone(); torch.cuda.synchronize()
    assert torch.equal(ref, fn())
    bad = 0
    for blk in range(reps // 20):
        cur = torch.cuda.current_stream(dev); sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sb):
            for _ in range(2): big(t, xb)
        with torch.cuda.stream(sa):
            outs = [fn() for _ in range(20)]
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    print(f"{name:36s}: {bad} of {reps // 20 * 20} co-scheduled results differ", flush=True)
