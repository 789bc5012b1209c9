"""GPU parity: HIP DiT path (through the C ABI) vs the CPU oracle / reference golden vectors.

Tolerances (SURVEY.md §8c rule 4; fp16 operands, fp32 accumulate, fp32 residual stream):
  per-forward velocity rel-L2 <= 2e-3 against the fp32 oracle.
"""
import functools
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dit_ref  # checker only


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), map_location="cpu", weights_only=False)


# ----------------------------------------------------------------------------- building blocks
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 384, 1024), (200, 132, 64), (1, 1536, 768), (4096, 1024, 4096)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_epilogues(dev, M, N, K, epi):
    from lfm_amd import hip

    g = torch.Generator().manual_seed(M * 7 + N + K + epi)
    A = (torch.randn(M, K, generator=g) * 0.5).half()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    bias = torch.randn(N, generator=g) * 0.1
    ref = A.float() @ W.float().t() + bias
    tokens = 8 if M % 8 == 0 else 1
    if epi == 1:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    out = None
    gate = None
    if epi == 3:
        X = torch.randn(M, N, generator=g)
        gate = torch.randn(M // tokens, N, generator=g)
        ref = X + gate.repeat_interleave(tokens, 0) * ref
        out = X.clone().to(dev)
        gate = gate.to(dev)
    got = hip.gemm_f16(A.to(dev), W.to(dev), bias.to(dev), epilogue=epi, out=out, gate=gate, gate_stride=N, tokens=tokens)
    torch.cuda.synchronize()
    assert rel_l2(got, ref) < (2e-3 if epi in (0, 1) else 2e-4)


@pytest.mark.parametrize("M,N,K", [(256, 3072, 1024), (256, 1024, 4096), (128, 192, 64), (192, 1152, 1152), (64, 64, 128), (256, 4608, 1152), (256, 2304, 768)])
@pytest.mark.parametrize("kernel", [7, 8])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_latency_mode_kernels_alone(dev, kernel, M, N, K, epi):
    """The two batch-1 kernels through lfm_gemm_f16 (lfm_gemm_select 7 = 64x64 tiles with the XCD-grouped tile order where the column tiles are a multiple
    of 8 and the plain order elsewhere, csrc/gemm_sq64_kernel.h; 8 = all rows x 16 columns, csrc/gemm_skinny_kernel.h) against the fp32 product; K-tile counts
    from 1 (every DMA in flight at once) to 64 (the ring wraps eight times); the same summation order per output element, so the two agree bit for bit."""
    from lfm_amd import hip

    g = torch.Generator().manual_seed(M * 5 + N + K + epi)
    A = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
    bias = torch.randn(N, generator=g) * 0.1
    ref = A.float().cpu() @ W.float().cpu().t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    outs = {}
    try:
        for k in (kernel, 15 - kernel):
            hip.gemm_select(k)
            outs[k] = hip.gemm_f16(A, W, bias.to(dev), epilogue=epi).clone()
    finally:
        hip.gemm_select(0)
    torch.cuda.synchronize()
    assert rel_l2(outs[kernel], ref) < (2e-3 if epi in (0, 1) else 2e-4)
    assert torch.equal(outs[7], outs[8])


def test_latency_mode_kernels_refuse_other_shapes(dev):
    from lfm_amd import hip

    A = torch.zeros(200, 128, device=dev).half()
    W = torch.zeros(64, 128, device=dev).half()
    b = torch.zeros(64, device=dev)
    try:
        hip.gemm_select(7)
        with pytest.raises(RuntimeError):
            hip.gemm_f16(A, W, b, epilogue=0)  # 200 rows: not whole 64-row tiles
        hip.gemm_select(8)
        hip.gemm_f16(A, W, b, epilogue=0)  # the all-rows kernel takes ragged row counts
        with pytest.raises(RuntimeError):
            hip.gemm_f16(torch.zeros(320, 128, device=dev).half(), W, b, epilogue=0)  # more than 256 rows
    finally:
        hip.gemm_select(0)


@functools.lru_cache(maxsize=64)
def _gemm_case(M, N, K, epi):
    """Host operands and the fp32 reference of one (shape, epilogue) case: shared by every kernel the case is run with."""
    g = torch.Generator().manual_seed(M + N * 3 + K + epi)
    A = (torch.randn(M, K, generator=g) * 0.5).half()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    bias = torch.randn(N, generator=g) * 0.1
    ref = A.float() @ W.float().t() + bias
    tokens = 4
    if epi == 1:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    X = gate = None
    if epi == 3:
        X = torch.randn(M, N, generator=g)
        gate = torch.randn(M // tokens, N, generator=g)
        ref = X + gate.repeat_interleave(tokens, 0) * ref
    return A, W, bias, X, gate, ref, tokens


@pytest.mark.parametrize("M,N,K", [(512, 512, 128), (1024, 768, 1024), (300, 260, 64), (4096, 1024, 4096), (8192, 3072, 1024),
                                   (512, 256, 192), (256, 512, 320), (768, 512, 576), (512, 128, 96), (384, 132, 160), (256, 128, 32),
                                   (65536, 128, 1152)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
@pytest.mark.parametrize("kernel", [4, 4 | (1024 << 4), 5, 5 | (1024 << 4), 6, 6 | (1024 << 4)])
def test_gemm256_kernels(dev, M, N, K, epi, kernel):
    """Same checks with a 256-row kernel forced (lfm_gemm_select: 4 = the 256x128 two-workgroups-per-CU kernel, 4 | 1024<<4 = with the 8-byte-store
    epilogue, 5 = the quadrant-phased 256x256 kernel on 16x16x32 MFMAs, the default for chip-filling shapes, 6 = the one-wave-per-SIMD 256x256
    kernel with 128x128 wave tiles, gemm256w_kernel.h); K covers 1, 2, 3, 5, odd and even
    numbers of 64-deep (and, for the 256x128 kernel, 32-deep) K-tiles, i.e. every prologue / tail path; N = 128 / 132 the narrow shapes; repeated
    launches screen for races."""
    if (kernel & 15) != 4 and M > 8192 and N == 128:
        pytest.skip("the narrow-N convolution shape is the 256x128 kernel's")
    if (kernel & 15) in (5, 6) and K % 64:
        pytest.skip("64-deep K-tiles: every K on the reference path is a multiple of 64 (32-deep tails are the 256x128 kernel's)")
    from lfm_amd import hip

    A, W, bias, X, gate, ref, tokens = _gemm_case(M, N, K, epi)
    if gate is not None:
        gate = gate.to(dev)
    hip.gemm_select(kernel)
    try:
        Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
        outs = []
        for _ in range(4):
            out = X.clone().to(dev) if epi == 3 else None
            outs.append(hip.gemm_f16(Ad, Wd, bd, epilogue=epi, out=out, gate=gate, gate_stride=N, tokens=tokens))
        torch.cuda.synchronize()
    finally:
        hip.gemm_select(0)
    assert rel_l2(outs[0], ref) < (2e-3 if epi in (0, 1) else 2e-4)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])  # deterministic across launches (no data race on the LDS stages)


@pytest.mark.parametrize("kernel", [4, 5, 6])
def test_gemm256_detects_transpose(dev, kernel):
    from lfm_amd import hip

    A = torch.eye(256, 256).half()
    W = (torch.arange(512 * 256).reshape(512, 256) % 97).half()
    hip.gemm_select(kernel)
    try:
        got = hip.gemm_f16(A.to(dev), W.to(dev), None, epilogue=2)
    finally:
        hip.gemm_select(0)
    assert torch.equal(got.cpu(), W.float().t())


@functools.lru_cache(maxsize=8)
def _qkv_case(batch, tokens, D):
    M = batch * tokens
    g = torch.Generator().manual_seed(batch + tokens + D)
    A = (torch.randn(M, D, generator=g) * 0.5).half()
    W = (torch.randn(3 * D, D, generator=g) / D ** 0.5).half()
    bias = torch.randn(3 * D, generator=g) * 0.1
    return A, W, bias, A.float() @ W.float().t() + bias


@pytest.mark.parametrize("kernel", [1, 4, 5, 5 | (1024 << 4), 6, 6 | (1024 << 4)])
@pytest.mark.parametrize("batch,tokens,D,hd", [(3, 256, 384, 64), (8, 64, 512, 64), (2, 256, 1024, 64), (2, 256, 1152, 72), (3, 64, 576, 72)])
def test_gemm_qkv_split(dev, batch, tokens, D, hd, kernel):
    """Fused QKV projection: Q, K row-major, V transposed per head (timm Attention's qkv + reshape + permute, DiT.py:120), with
    every GEMM kernel (fragment-direct V^T stores in v1/v2 and v3's two-barrier schedule, operand-swapped V tiles in v3)."""
    from lfm_amd import hip

    A, W, bias, ref = _qkv_case(batch, tokens, D)
    hip.gemm_select(kernel)
    try:
        Q, K, Vt = hip.gemm_qkv_f16(A.to(dev), W.to(dev), bias.to(dev), hd, tokens)
        torch.cuda.synchronize()
    finally:
        hip.gemm_select(0)
    assert rel_l2(Q, ref[:, :D]) < 2e-3 and rel_l2(K, ref[:, D:2 * D]) < 2e-3
    v_ref = ref[:, 2 * D:].reshape(batch, tokens, D // hd, hd).permute(0, 2, 3, 1)  # [b, head, d, tok]
    assert rel_l2(Vt[..., hip.vt_token_perm(tokens, dev)], v_ref) < 2e-3  # V^T rows are stored with every 16-token group permuted (hip.vt_token_perm)


def test_gemm_detects_transpose(dev):
    """A = I with an asymmetric W: a swapped C write cannot pass (cdna guide: always A=I-check)."""
    from lfm_amd import hip

    A = torch.eye(128, 128).half()
    W = (torch.arange(256 * 128).reshape(256, 128) % 97).half()
    got = hip.gemm_f16(A.to(dev), W.to(dev), None, epilogue=2)
    assert torch.equal(got.cpu(), W.float().t())


def test_gemm_rejects_bad_shapes(dev):
    from lfm_amd import hip

    A = torch.zeros(64, 72, device=dev, dtype=torch.float16)  # K not a multiple of 64
    W = torch.zeros(64, 72, device=dev, dtype=torch.float16)
    with pytest.raises(hip.LfmHipError):
        hip.gemm_f16(A, W)


@pytest.mark.parametrize("D,tokens,shared,n_img", [(1024, 256, False, 3), (768, 256, True, 3), (384, 64, False, 3), (128, 256, False, 3),
                                                   (1024, 256, True, 20), (384, 100, False, 43), (1152, 64, False, 70)])
def test_ln_modulate(dev, D, tokens, shared, n_img):
    """Two rows per wave; up to 5120 rows and a ragged last block.  (A row-walking variant -- 8 rows per wave with the next row's loads in
    flight -- measured 22.5 us vs 21.4 us at the benchmark shape and was not kept.)"""
    from lfm_amd import hip

    g = torch.Generator().manual_seed(D)
    X = torch.randn(n_img * tokens, D, generator=g) * 2 + 0.3
    rows = 1 if shared else n_img
    shift, scale = torch.randn(rows, D, generator=g) * 0.2, torch.randn(rows, D, generator=g) * 0.2
    ln = torch.nn.functional.layer_norm(X, (D,), eps=1e-6).reshape(n_img, tokens, D)
    ref = (ln * (1 + scale[:, None]) + shift[:, None]).reshape(-1, D)
    mod = torch.stack([shift, scale], 1).contiguous().to(dev)  # [rows, 2, D]
    got = hip.ln_modulate(X.to(dev), mod[:, 0], mod[:, 1], tokens, 0 if shared else 2 * D)
    assert rel_l2(got, ref) < 1e-3


@pytest.mark.parametrize("T,heads,batch,hd", [(256, 16, 3, 64), (256, 2, 1, 64), (64, 6, 2, 64), (128, 4, 2, 64), (256, 16, 17, 64), (256, 16, 64, 64),
                                              (256, 12, 43, 64), (256, 16, 3, 72), (64, 16, 5, 72), (128, 3, 2, 72), (256, 16, 33, 72), (16, 6, 7, 64), (16, 16, 3, 72),
                                              (1024, 6, 2, 64), (1024, 16, 5, 64), (1024, 16, 2, 72)])
def test_attention(dev, T, heads, batch, hd):
    """Up to the benchmark's own shape (64 images x 16 heads x 256 tokens); every (image, head) item is checked on its own.
    head_dim 72 = the DiT-XL family (4.5 MFMA k-slots, 2.25 output row blocks: the padding lanes must contribute exactly nothing)."""
    from lfm_amd import hip

    g = torch.Generator().manual_seed(T + heads)
    D = heads * hd
    q = (torch.randn(batch, heads, T, hd, generator=g) * 1.5).half()
    k = (torch.randn(batch, heads, T, hd, generator=g) * 1.5).half()
    v = torch.randn(batch, heads, T, hd, generator=g).half()
    k[0, 0, 5] *= 6  # a spiky key row exercises the running-max rescale of the online softmax
    if T == 1024:
        k[0, 0, 700] *= 8  # ... and one in a LATER key chunk of the 1024-token kernel (the state carried across chunks must rescale)
    ref = torch.softmax((q.float() @ k.float().transpose(-1, -2)) * hd ** -0.5, -1) @ v.float()
    ref = ref.transpose(1, 2).reshape(batch * T, D)
    Q = q.transpose(1, 2).reshape(batch * T, D).contiguous().to(dev)
    K = k.transpose(1, 2).reshape(batch * T, D).contiguous().to(dev)
    Vt = v.transpose(-1, -2)[..., hip.vt_token_perm(T)].contiguous().to(dev)  # [b, h, hd, T] in the library's token order
    got = hip.dit_attention(Q, K, Vt, batch, heads, T, head_dim=hd)
    got2 = hip.dit_attention(Q, K, Vt, batch, heads, T, head_dim=hd)
    torch.cuda.synchronize()
    assert rel_l2(got, ref) < 2e-3
    assert torch.equal(got, got2)
    per_item = (got.float().cpu().reshape(batch, T, heads, hd) - ref.reshape(batch, T, heads, hd)).pow(2).sum((1, 3)).sqrt() / \
        ref.reshape(batch, T, heads, hd).pow(2).sum((1, 3)).sqrt()
    assert float(per_item.max()) < 4e-3  # no single (image, head) item is off (a stale-buffer bug would hit whole items)


@pytest.mark.parametrize("small", [1, 2, 4])
def test_attention_query_split_matches(dev, small):
    """<= 64 (image, head) items of 256 tokens x hd 64 (latency mode: one to four DiT-L images) run the query-split kernel -- two four-wave workgroups per item,
    each staging every key: the same arithmetic per query in the same key order, so the leading `small` images of a batch of 8 (128 items: the eight-wave
    kernel) come out bit for bit the same when they are evaluated on their own."""
    from lfm_amd import hip

    T, heads, hd, batch = 256, 16, 64, 8
    g = torch.Generator().manual_seed(small)
    D = heads * hd
    Q = (torch.randn(batch * T, D, generator=g) * 1.5).half().to(dev)
    K = (torch.randn(batch * T, D, generator=g) * 1.5).half().to(dev)
    Vt = torch.randn(batch, heads, hd, T, generator=g).half().to(dev)
    K[5, :hd] *= 6  # a spiky key row of the first item
    full = hip.dit_attention(Q, K, Vt, batch, heads, T, head_dim=hd)
    part = hip.dit_attention(Q[: small * T].contiguous(), K[: small * T].contiguous(), Vt[:small].contiguous(), small, heads, T, head_dim=hd)
    torch.cuda.synchronize()
    assert torch.equal(part, full[: small * T])
    assert float(part.float().abs().mean()) > 1e-3


@pytest.mark.parametrize("batch,heads", [(5, 16), (8, 16), (33, 16), (64, 16), (100, 16), (43, 12), (171, 6)])
def test_attention_stream_matches_per_item(dev, batch, heads):
    """256 tokens x hd 64 with more than 64 (image, head) items run on PERSISTENT workgroups that stream K / V^T of consecutive items through a four-slot LDS ring
    (csrc/attention_stream_kernel.h: counted waits across item boundaries, out-of-range stages at the tail).  Same arithmetic per query in the same key order as the
    one-workgroup-per-item kernel: bit-identical for every item count -- one item per workgroup (80, 128 items), a ragged second round (528, 516), two to four items
    per workgroup (1024, 1600, 1026) -- with spiky keys that force the online-softmax rescale in every stage, twice in a row (bit-repeatable), and inside the
    tolerance of the fp32 reference item by item."""
    from lfm_amd import hip

    T, hd = 256, 64
    D = heads * hd
    g = torch.Generator().manual_seed(batch * 100 + heads)
    Q = (torch.randn(batch * T, D, generator=g) * 1.5).half().to(dev)
    K = (torch.randn(batch * T, D, generator=g) * 1.5).half().to(dev)
    Vt = torch.randn(batch, heads, hd, T, generator=g).half().to(dev)
    for img, tok in ((0, 5), (batch // 2, 70), (batch - 1, 130), (batch - 1, 250), (1, 200)):  # one spike per 64-key stage somewhere, first / middle / last items
        K[img * T + tok, : 2 * hd] *= 6
    try:
        hip.set_option(hip.OPT_ATTENTION_STREAM, 0)
        ref = hip.dit_attention(Q, K, Vt, batch, heads, T, head_dim=hd)
        hip.set_option(hip.OPT_ATTENTION_STREAM, 1)
        out = hip.dit_attention(Q, K, Vt, batch, heads, T, head_dim=hd)
        out2 = hip.dit_attention(Q, K, Vt, batch, heads, T, head_dim=hd)
        torch.cuda.synchronize()
    finally:
        hip.set_option(hip.OPT_ATTENTION_STREAM, 1)
    assert torch.equal(out, out2)
    bad = (out != ref).reshape(batch, T, heads, hd).any(dim=3).any(dim=1)  # [batch, heads]: which items differ
    assert not bool(bad.any()), f"{int(bad.sum())} of {batch * heads} items differ, first {bad.nonzero()[:4].tolist()}"
    # and against fp32 (V^T rows are stored in the library's token order: vt_token_perm is an involution)
    q = Q.float().reshape(batch, T, heads, hd).permute(0, 2, 1, 3)
    k = K.float().reshape(batch, T, heads, hd).permute(0, 2, 1, 3)
    v = Vt.float()[..., hip.vt_token_perm(T, device=dev)].transpose(2, 3)  # [batch, heads, T, hd]
    want = torch.softmax(q @ k.transpose(2, 3) * hd ** -0.5, dim=-1) @ v
    got = out.float().reshape(batch, T, heads, hd).permute(0, 2, 1, 3)
    per_item = (got - want).pow(2).sum((2, 3)).sqrt() / want.pow(2).sum((2, 3)).sqrt()
    assert float(per_item.max()) < 4e-3


def test_attention_refuses_unbuilt_head_sizes(dev):
    from lfm_amd import hip

    z = torch.zeros(256, 96, device=dev, dtype=torch.float16)
    with pytest.raises(hip.LfmHipError):
        hip.dit_attention(z, z, z.reshape(1, 1, 96, 256), 1, 1, 256, head_dim=96)


# ----------------------------------------------------------------------------- whole model
def _model_from_state(cfgkw, sd, dev):
    from lfm_amd.models import DiT

    m = DiT(img_resolution=cfgkw["img_resolution"], patch_size=cfgkw["patch"], in_channels=cfgkw["in_channels"],
            hidden_size=cfgkw["hidden"], depth=cfgkw["depth"], num_heads=cfgkw["heads"],
            label_dropout=cfgkw["label_dropout"], num_classes=cfgkw["num_classes"])
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval()


@pytest.mark.parametrize("which", ["cond", "uncond"])
def test_dit_matches_reference_golden(dev, golden_dir, which):
    """Same weights, same latents as the unmodified reference (tests/golden/dit_tiny.pt)."""
    rec = _load(golden_dir, "dit_tiny.pt")[which]
    m = _model_from_state(rec["cfg"], rec["state_dict"], dev)
    x = rec["x"].to(dev)
    y = rec["y"].to(dev) if "y" in rec else None
    assert rel_l2(m(torch.tensor(0.37, device=dev), x, y), rec["v_t0d"]) < 2e-3
    assert rel_l2(m(torch.tensor([0.9, 0.5, 0.02], device=dev), x, y), rec["v_tN"]) < 2e-3
    if which == "cond":
        assert rel_l2(m(torch.tensor(0.37), x), rec["v_ynone"]) < 2e-3
        got = m.forward_with_cfg(torch.tensor(0.37, device=dev), rec["x_cfg"].to(dev), rec["y_cfg"].to(dev), cfg_scale=rec["cfg_scale"])
        assert rel_l2(got, rec["v_cfg"]) < 2e-3


def test_dit_head_dim_72_matches_reference_golden(dev, golden_dir):
    """The DiT-XL family's head size (1152 / 16 = 72, reference models/DiT.py:354-363): golden from the unmodified reference on a 576-wide,
    8-head, depth-2 model (tests/golden/dit_hd72.pt; weights regenerated from the seeded state maker, checksum-checked), then
    DiT-XL/2 and DiT-XL/4 at full size against the oracle (XL/2 with every GEMM kernel forced: test_dit_with_every_gemm_kernel)."""
    rec = _load(golden_dir, "dit_hd72.pt")
    cfg = dit_ref.DiTCfg(**rec["cfg"])
    sd = dit_ref.make_dit_state(cfg, seed=rec["state_seed"])
    assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - rec["state_checksum"]) < 1e-6 * rec["state_checksum"]
    m = _model_from_state(rec["cfg"], sd, dev)
    x, y = rec["x"].to(dev), rec["y"].to(dev)
    assert rel_l2(m(torch.tensor([0.9, 0.5, 0.02], device=dev), x, y), rec["v_tN"]) < 2e-3
    got = m.forward_with_cfg(torch.tensor(0.37, device=dev), rec["x_cfg"].to(dev), rec["y_cfg"].to(dev), cfg_scale=rec["cfg_scale"])
    assert rel_l2(got, rec["v_cfg"]) < 2e-3
    from lfm_amd.models import DiT_models

    for name, batch in (("DiT-XL/2", 2), ("DiT-XL/4", 3)):
        kw = dict(num_classes=10, label_dropout=0.1)
        cfg = dit_ref.DiTCfg.named(name, **kw)
        sd = dit_ref.make_dit_state(cfg, seed=5)
        mm = DiT_models[name](img_resolution=32, in_channels=4, **kw)
        mm.load_state_dict(sd, strict=True)
        mm = mm.to(dev).eval()
        g = torch.Generator().manual_seed(9)
        xx = torch.randn(batch, 4, 32, 32, generator=g)
        yy = torch.randint(0, 10, (batch,), generator=g)
        t = torch.linspace(0.2, 0.9, batch)
        assert rel_l2(mm(t.to(dev), xx.to(dev), yy.to(dev)), dit_ref.dit_forward(sd, cfg, t, xx, yy)) < 2e-3, (name, batch)
        del mm


def test_dit_patch4_matches_reference_golden(dev, golden_dir):
    """DiT-x/4 family: the patch embedding runs on the MFMA GEMM (K = 64), the final layer emits 64 outputs per token in four passes.
    Golden from the unmodified reference (tests/golden/dit_p4.pt); then DiT-S/4 and DiT-B/4 at full size against the oracle."""
    rec = _load(golden_dir, "dit_p4.pt")
    m = _model_from_state(rec["cfg"], rec["state_dict"], dev)
    x, y = rec["x"].to(dev), rec["y"].to(dev)
    assert rel_l2(m(torch.tensor([0.9, 0.5, 0.02], device=dev), x, y), rec["v_tN"]) < 2e-3
    got = m.forward_with_cfg(torch.tensor(0.37, device=dev), rec["x_cfg"].to(dev), rec["y_cfg"].to(dev), cfg_scale=rec["cfg_scale"])
    assert rel_l2(got, rec["v_cfg"]) < 2e-3
    from lfm_amd.models import DiT_models

    for name, batch in (("DiT-S/4", 5), ("DiT-B/4", 2), ("DiT-S/8", 9), ("DiT-B/8", 3)):  # x/8 on 32x32 latents: 16 tokens of 256 inputs
        kw = dict(num_classes=10, label_dropout=0.1)
        cfg = dit_ref.DiTCfg.named(name, **kw)
        sd = dit_ref.make_dit_state(cfg, seed=4)
        mm = DiT_models[name](img_resolution=32, in_channels=4, **kw)
        mm.load_state_dict(sd, strict=True)
        mm = mm.to(dev).eval()
        g = torch.Generator().manual_seed(8)
        xx = torch.randn(batch, 4, 32, 32, generator=g)
        yy = torch.randint(0, 10, (batch,), generator=g)
        t = torch.linspace(0.2, 0.9, batch)
        assert rel_l2(mm(t.to(dev), xx.to(dev), yy.to(dev)), dit_ref.dit_forward(sd, cfg, t, xx, yy)) < 2e-3, name


@pytest.mark.parametrize("name,batch,kw", [
    ("DiT-B/2", 4, dict(num_classes=1, label_dropout=0.0)),      # BASELINE config 1 shape
    ("DiT-B/2", 6, dict(num_classes=1000, label_dropout=0.1)),   # config 4 shape (class-conditional)
    ("DiT-L/2", 3, dict(num_classes=1, label_dropout=0.0)),      # config 2 shape
    ("DiT-S/2", 1, dict(num_classes=1, label_dropout=0.0)),      # batch 1 (--measure_time mode): a single 256-row tile
    ("DiT-L/2", 1, dict(num_classes=1, label_dropout=0.0)),      # the reference's --measure_time configuration: split-K GEMMs whose finish is also the LayerNorm
    ("DiT-XL/2", 2, dict(num_classes=1000, label_dropout=0.1)),  # 1152-wide rows (two column groups per thread in the fused finish), one conditioning row per image
    ("DiT-S/2", 65, dict(num_classes=10, label_dropout=0.1)),    # ragged everywhere: 65 M-tiles, N = 1152/384 not multiples of 256
])
def test_dit_matches_oracle_fullsize(dev, name, batch, kw):
    from lfm_amd.models import DiT_models

    cfg = dit_ref.DiTCfg.named(name, **kw)
    sd = dit_ref.make_dit_state(cfg, seed=1)
    m = DiT_models[name](img_resolution=32, in_channels=4, **kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(42)
    x = torch.randn(batch, 4, 32, 32, generator=g)
    y = torch.randint(0, kw["num_classes"], (batch,), generator=g) if kw["num_classes"] > 1 else None
    for t in (torch.tensor(1.0), torch.tensor(0.5), torch.linspace(0.1, 0.9, batch)):
        ref = dit_ref.dit_forward(sd, cfg, t, x, y)
        got = m(t.to(dev), x.to(dev), y.to(dev) if y is not None else None)
        assert float(ref.abs().mean()) > 1e-3
        assert rel_l2(got, ref) < 2e-3, (name, t)


@pytest.mark.parametrize("name,batch,kw", [
    ("DiT-L/2", 1, dict(num_classes=1, label_dropout=0.0)),      # the reference's --measure_time configuration (test_flow_latent.py:223-246)
    ("DiT-S/2", 1, dict(num_classes=10, label_dropout=0.1)),     # 384 wide: 24 / 72 / 96 column slices, two K-slices for proj / fc2
    ("DiT-XL/2", 1, dict(num_classes=1000, label_dropout=0.1)),  # 1152 wide, head_dim 72, two column groups per thread in the finish
    ("DiT-B/4", 3, dict(num_classes=1, label_dropout=0.0)),      # three images of 64 tokens: 192 rows (ragged against the kernel's 256), patch-4 embedding
])
def test_skinny_latency_kernel_vs_splitk_path_and_oracle(dev, name, batch, kw):
    """Evaluations of <= 256 token rows run their four block linears on the latency-mode kernels (default: 64x64 tiles, csrc/gemm_sq64_kernel.h, where the
    rows are whole 64-token tiles, else all rows x 16 columns, csrc/gemm_skinny_kernel.h; option value 2 forces the latter): the same forward as the
    split-K 128x128 path of rounds 2-4 (lfm_set_option(LFM_OPT_SKINNY_GEMM, 0)) up to the fp32 summation order, bit-repeatable, and inside the
    per-forward budget against the CPU oracle."""
    from lfm_amd import hip
    from lfm_amd.models import DiT_models

    cfg = dit_ref.DiTCfg.named(name, **kw)
    sd = dit_ref.make_dit_state(cfg, seed=3)
    m = DiT_models[name](img_resolution=32, in_channels=4, **kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(batch, 4, 32, 32, generator=g)
    y = torch.randint(0, kw["num_classes"], (batch,), generator=g) if kw["num_classes"] > 1 else None
    t = torch.tensor(0.37)
    xd, td, yd = x.to(dev), t.to(dev), (y.to(dev) if y is not None else None)
    a = m(td, xd, yd).clone()
    b = m(td, xd, yd).clone()
    try:
        hip.set_option(hip.OPT_SKINNY_GEMM, 2)  # always all rows x 16 columns (1 = default: 64x64 tiles where the rows are whole 64-token tiles)
        sk = m(td, xd, yd).clone()
        sk2 = m(td, xd, yd).clone()
        hip.set_option(hip.OPT_SKINNY_GEMM, 0)
        old = m(td, xd, yd).clone()
    finally:
        hip.set_option(hip.OPT_SKINNY_GEMM, 1)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(sk, sk2)
    assert not torch.equal(a, old) and not torch.equal(sk, old), "the latency-mode kernels did not run (preconditions?)"
    assert rel_l2(a, old) < 2e-4 and rel_l2(sk, old) < 2e-4  # same fp16 operands, fp32 accumulation in another order
    assert torch.equal(a, sk)  # the two latency-mode kernels slice K alike and sum in the same order (which one ran: profiles/r05_latency_mode.txt)
    ref = dit_ref.dit_forward(sd, cfg, t, x, y)
    assert rel_l2(a, ref) < 2e-3 and rel_l2(sk, ref) < 2e-3 and rel_l2(old, ref) < 2e-3


@pytest.mark.parametrize("name,batch,kw", [
    ("DiT-S/2", 3, dict(num_classes=10, label_dropout=0.1)),     # 384 wide: the 128x128 GEMM tiles
    ("DiT-L/2", 2, dict(num_classes=1, label_dropout=0.0)),      # 1024 wide: the 256x256 tiles and the folded LayerNorm-modulate over 1024-token images
    ("DiT-XL/2", 1, dict(num_classes=1000, label_dropout=0.1)),  # head_dim 72
])
def test_dit_1024_tokens_matches_oracle(dev, name, batch, kw):
    """64x64 latents (the f8 latent of a 512x512 image, train_flow_latent.py --image_size 512 with models/DiT.py:179-182's patch 2): 1024 tokens per
    image, past what the attention kernel keeps in the LDS at once -- its four key chunks, the token split of the QKV epilogue (shift 10) and the
    positional table at grid 32 against the oracle."""
    from lfm_amd.models import DiT_models

    cfg = dit_ref.DiTCfg.named(name, img_resolution=64, **kw)
    sd = dit_ref.make_dit_state(cfg, seed=5)
    m = DiT_models[name](img_resolution=64, in_channels=4, **kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(64)
    x = torch.randn(batch, 4, 64, 64, generator=g)
    y = torch.randint(0, kw["num_classes"], (batch,), generator=g) if kw["num_classes"] > 1 else None
    t = torch.linspace(0.3, 0.8, batch)
    ref = dit_ref.dit_forward(sd, cfg, t, x, y)
    got = m(t.to(dev), x.to(dev), y.to(dev) if y is not None else None)
    got2 = m(t.to(dev), x.to(dev), y.to(dev) if y is not None else None)
    assert float(ref.abs().mean()) > 1e-3
    assert rel_l2(got, ref) < 2e-3, name
    assert torch.equal(got, got2)


@functools.lru_cache(maxsize=4)
def _every_kernel_case(name, batch):
    """Model on the GPU, inputs and the oracle's answer: built once per (model, batch), run with every kernel."""
    from lfm_amd.models import DiT_models

    kw = dict(num_classes=10, label_dropout=0.1)
    cfg = dit_ref.DiTCfg.named(name, **kw)
    sd = dit_ref.make_dit_state(cfg, seed=3)
    m = DiT_models[name](img_resolution=32, in_channels=4, **kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(torch.device("cuda:0")).eval()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(batch, 4, 32, 32, generator=g)
    y = torch.randint(0, 10, (batch,), generator=g)
    t = torch.linspace(0.2, 0.8, batch)
    return m, x, y, t, dit_ref.dit_forward(sd, cfg, t, x, y)


@pytest.mark.parametrize("kernel", [1, 4, 5, 6])
@pytest.mark.parametrize("name,batch", [("DiT-S/2", 5), ("DiT-B/2", 3), ("DiT-XL/2", 2)])
def test_dit_with_every_gemm_kernel(dev, name, batch, kernel):
    """The same forward with each GEMM kernel forced (auto picks by size, so small test batches would never reach the 256x256
    kernels through the model): covers the fused epilogues in situ, in particular the QKV split with V written transposed
    (fragment-direct in the 128x128 kernel, operand-swapped tiles + transposed epilogue in the 256-row ones)."""
    from lfm_amd import hip
    from lfm_amd.models import DiT_models

    m, x, y, t, ref = _every_kernel_case(name, batch)
    hip.gemm_select(kernel)
    try:
        got = m(t.to(dev), x.to(dev), y.to(dev))
        torch.cuda.synchronize()
    finally:
        hip.gemm_select(0)
    assert rel_l2(got, ref) < 2e-3, (name, kernel)


def test_dit_fused_euler_update(dev):
    """out = base + dt*v fused into the final-layer kernel equals the unfused v."""
    import ctypes as C

    from lfm_amd import hip
    from lfm_amd.models import DiT_models

    cfg = dit_ref.DiTCfg.named("DiT-S/2", num_classes=1, label_dropout=0.0)
    sd = dit_ref.make_dit_state(cfg, seed=5)
    m = DiT_models["DiT-S/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(0)).to(dev)
    t = torch.tensor(0.7, device=dev)
    v = m(t, x)
    dt = torch.tensor([-0.02], device=dev)
    xn = x.clone()
    m._run(t, xn, None, False, 1.0, out=xn, axpy_base=xn, axpy_dt=dt)  # in place
    torch.testing.assert_close(xn, x + dt * v, rtol=1e-6, atol=1e-6)
    _ = (C, hip)


def test_forward_refuses_cpu():
    from lfm_amd import hip
    from lfm_amd.models import DiT_models

    m = DiT_models["DiT-S/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0).eval()
    with pytest.raises(hip.LfmHipError):
        m(torch.tensor(0.5), torch.zeros(1, 4, 32, 32))


@pytest.mark.parametrize("kernel", [0, 6])
@pytest.mark.parametrize("name,batch,labels", [("DiT-L/2", 48, False), ("DiT-B/2", 64, True)])
def test_folded_ln_epilogues_match_separate_launches_and_the_oracle(dev, name, batch, labels, kernel):
    """lfm_set_option(LFM_OPT_FOLD_LN) (default on): LayerNorm-modulate folded into the proj / fc2 / qkv / fc1 epilogues must give the forward of the
    separate ln_modulate launches up to the fp16 rounding of the GEMM operand, be bit-repeatable (no atomics, no inter-workgroup waits), and stay
    inside the per-forward budget against the CPU oracle on its own.  The smallest batches that meet its preconditions: DiT-L/2 (4 column tiles,
    one shared conditioning row) at 48, class-conditional DiT-B/2 (3 column tiles, one conditioning row PER IMAGE) at 64.  Both with the default
    kernels and with the one-wave-per-SIMD kernel forced."""
    from lfm_amd import hip
    from lfm_amd.models import DiT_models

    kw = dict(num_classes=1000, label_dropout=0.1) if labels else dict(num_classes=1, label_dropout=0.0)
    cfg = dit_ref.DiTCfg.named(name, **kw)
    sd = dit_ref.make_dit_state(cfg, seed=6)
    m = DiT_models[name](img_resolution=32, in_channels=4, **kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(batch, 4, 32, 32, generator=g)
    y = torch.randint(0, 1001, (batch,), generator=g) if labels else None
    t = torch.linspace(0.05, 0.95, batch) if labels else torch.tensor(0.6)
    xd, td, yd = x.to(dev), t.to(dev), (y.to(dev) if labels else None)
    hip.gemm_select(kernel)  # 0: the default kernels; 6: producer / consumer epilogues of the one-wave-per-SIMD kernel (gemm256w_kernel.h)
    try:
        hip.set_option(hip.OPT_FOLD_LN, 0)
        try:
            base = m(td, xd, yd).clone()
        finally:
            hip.set_option(hip.OPT_FOLD_LN, 1)
        a = m(td, xd, yd).clone()
        b = m(td, xd, yd).clone()
        torch.cuda.synchronize()
    finally:
        hip.gemm_select(0)
    assert torch.equal(a, b)
    assert not torch.equal(a, base), "the folded path did not run (preconditions?)"
    assert rel_l2(a, base) < 1e-3
    k = 4  # oracle on a few images (each image is independent of the others)
    ref = dit_ref.dit_forward(sd, cfg, t[:k] if labels else t, x[:k], y[:k] if labels else None)
    assert rel_l2(a[:k].cpu(), ref) < 2e-3 and rel_l2(base[:k].cpu(), ref) < 2e-3


@pytest.mark.parametrize("name,batch,labels", [("DiT-L/2", 48, False), ("DiT-L/2", 64, False), ("DiT-B/2", 64, True), ("DiT-B/2", 67, False)])
def test_fused_qkv_attention_matches_two_kernels(dev, name, batch, labels):
    """lfm_set_option(LFM_OPT_FUSED_QKV_ATTENTION) (default on): on the folded path at 256 tokens x head_dim 64 the QKV projection and the attention core of a block
    run as ONE kernel, a workgroup per (image, head) (csrc/qkv_attention_kernel.h: the 256 x 192 slice of the projection on the quadrant-phased K loop, Q / K / V^T
    handed over through the LDS, the key loop of the per-item attention kernel).  Same K-tile order, same row affine, same softmax block: the forward must be
    BIT-IDENTICAL to the two-kernel path and bit-repeatable -- one shared conditioning row (DiT-L/2) and one row per image (class-conditional DiT-B/2: u / v per
    image), D = 1024 (16 K-tiles, 16 heads) and 768 (12 K-tiles, 12 heads: the tile order's group size changes), a batch that is not a multiple of the
    XCD count."""
    from lfm_amd import hip
    from lfm_amd.models import DiT_models

    kw = dict(num_classes=1000, label_dropout=0.1) if labels else dict(num_classes=1, label_dropout=0.0)
    cfg = dit_ref.DiTCfg.named(name, **kw)
    sd = dit_ref.make_dit_state(cfg, seed=9)
    m = DiT_models[name](img_resolution=32, in_channels=4, **kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(batch, 4, 32, 32, generator=g)
    y = torch.randint(0, 1001, (batch,), generator=g) if labels else None
    t = torch.linspace(0.05, 0.95, batch) if labels else torch.tensor(0.4)
    xd, td, yd = x.to(dev), t.to(dev), (y.to(dev) if labels else None)
    plan = hip.dit_plan(m.shape_struct(), batch, t_len=batch if labels else 1, labels=labels)
    assert plan == hip.PLAN_FOLDED_LN | hip.PLAN_FUSED_QKV_ATTENTION, "the fused kernel would not run for this case: the comparison below would be vacuous"
    try:
        hip.set_option(hip.OPT_FUSED_QKV_ATTENTION, 0)
        two = m(td, xd, yd).clone()
        hip.set_option(hip.OPT_FUSED_QKV_ATTENTION, 1)
        a = m(td, xd, yd).clone()
        b = m(td, xd, yd).clone()
        torch.cuda.synchronize()
    finally:
        hip.set_option(hip.OPT_FUSED_QKV_ATTENTION, 1)
    assert bool(torch.isfinite(a).all())
    assert torch.equal(a, b)
    bad = (a != two).flatten(1).any(dim=1)
    assert not bool(bad.any()), f"{int(bad.sum())} of {batch} images differ (first {bad.nonzero()[:4].flatten().tolist()}), rel-L2 {rel_l2(a, two):.3e}"
    k = 2
    ref = dit_ref.dit_forward(sd, cfg, t[:k] if labels else t, x[:k], y[:k] if labels else None)
    assert rel_l2(a[:k].cpu(), ref) < 2e-3


@pytest.mark.parametrize("fold", [1, 0])
def test_trained_like_dynamic_range(dev, fold):
    """Every other parity test runs on xavier / N(0, 0.02) weights.  A trained DiT has a few 'massive' residual channels, large adaLN scales and a wide
    fc1: this case puts four residual channels at +-3000 (through the patch-embedding bias, so they ride the residual stream through every block),
    multiplies the adaLN scale rows by 8 and the fc1 weights by 48 (fp16 fc1 activations beyond 64), and checks the forward against the fp32 oracle inside the usual per-forward budget
    -- with the LayerNorm folded into the GEMM epilogues (centred fp16 operand, one-pass shifted variance: the case this test was written for) and
    with the separate launches.  Finite output implies finite Q / K / V^T / H everywhere upstream (nothing masks an inf or NaN on this path); the last
    block's fp16 fc1 activation, still in the workspace, is checked directly."""
    from lfm_amd import hip
    from lfm_amd.models import DiT_models

    name, batch = "DiT-B/2", 64
    kw = dict(num_classes=1, label_dropout=0.0)
    cfg = dit_ref.DiTCfg.named(name, **kw)
    sd = dit_ref.make_dit_state(cfg, seed=11)
    D = cfg.hidden
    sd["x_embedder.proj.bias"][[5, 100, 333, 700]] += torch.tensor([3000.0, 3000.0, -3000.0, 3000.0])
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        for lo in (D, 4 * D):  # scale_msa, scale_mlp rows of the adaLN table
            sd[b + "adaLN_modulation.1.weight"][lo:lo + D] *= 8.0
            sd[b + "adaLN_modulation.1.bias"][lo:lo + D] *= 8.0
        sd[b + "mlp.fc1.weight"] *= 48.0
    m = DiT_models[name](img_resolution=32, in_channels=4, **kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    x = torch.randn(batch, 4, 32, 32, generator=torch.Generator().manual_seed(2))
    t = torch.tensor(0.35)
    hip.set_option(hip.OPT_FOLD_LN, fold)
    try:
        out = m(t.to(dev), x.to(dev)).clone()
        torch.cuda.synchronize()
    finally:
        hip.set_option(hip.OPT_FOLD_LN, 1)
    assert bool(torch.isfinite(out).all())
    k = 4
    ref = dit_ref.dit_forward(sd, cfg, t, x[:k])
    err = rel_l2(out[:k].cpu(), ref)
    assert err < 2e-3, err
    # the fc1 activation of the last block: fp16 [M, H] at the start of the Q | K | V^T region (csrc/dit.hip::carve: X fp32, A fp16, then QKVH)
    M, H = batch * 256, cfg.mlp_hidden
    ws = m._ws[1]
    off = M * D * 4 + M * D * 2
    hb = ws[off:off + M * H * 2].view(torch.float16)
    print(f"dynamic range (fold={fold}): fc1 activation peak {float(hb.abs().max()):.1f}, residual peak {float(ws[:M * D * 4].view(torch.float32).abs().max()):.0f}, "
          f"rel-L2 vs oracle {err:.2e}")
    assert bool(torch.isfinite(hb).all()) and float(hb.abs().max()) >= 64.0  # round 4: two orders wider than on N(0, 0.02) weights, still far from 65504
    assert float(ws[:M * D * 4].view(torch.float32).abs().max()) >= 2900.0  # the residual stream really carries the massive channels to the last block


def test_gemm_operands_beyond_2gb(dev):
    """The 256x256 kernels address row-major operands through buffer resources with UNSIGNED 32-bit byte offsets: an A operand of 2^30 .. 2^31 elements
    (2.1 .. 4.3 GB; e.g. the VAE's full-resolution 1x1 shortcut at decode_chunk 64: 64 x 256 x 256 pixels x 256 channels = 2^30) must give the result
    of the 128x128 kernel (plain pointer arithmetic) in its LAST rows too, where the byte offsets exceed 2^31."""
    from lfm_amd import hip

    M, N, K = (1 << 22) + 4096, 256, 256  # M * K = 2^30 + 2^20 elements
    g = torch.Generator(device=dev).manual_seed(5)
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).half()
    b = torch.randn(N, device=dev, generator=g) * 0.1
    outs = {}
    for k in (1, 5, 6):
        hip.gemm_select(k)
        try:
            outs[k] = hip.gemm_f16(A, W, b, epilogue=0)
        finally:
            hip.gemm_select(0)
    torch.cuda.synchronize()
    tail = slice(M - 2048, M)
    ref = A[tail].float() @ W.float().t() + b
    for k in (1, 5, 6):
        assert rel_l2(outs[k][tail], ref) < 2e-3, k
    for k in (5, 6):  # whole outputs, compared on the device (1 G elements)
        d = float((outs[k].float() - outs[1].float()).norm() / outs[1].float().norm())
        assert d < 1e-3, (k, d)
