"""CPU tests of the host logic: samplers / RNG against reference-generated golden vectors, the solver
module against the oracle restatement and analytic known answers, and the C-ABI export list."""
import math
import os
import re

import pytest
import torch

from oracle import ode_ref


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), map_location="cpu", weights_only=False)


# ----------------------------------------------------------------------------- RNG (reference sampler/random_util.py)
def test_random_util_matches_reference(golden_dir):
    from lfm_amd.sampler.random_util import get_generator

    gold = _load(golden_dir, "randgen.pt")
    for n, seed, bs in ((64, 42, 8), (10, 7, 4)):
        gen = get_generator("determ", n, seed)
        assert torch.equal(gen.randn(bs, 4, 8, 8), gold[f"determ_n{n}_s{seed}_randn"])
        assert torch.equal(gen.randn(bs, 4, 8, 8), gold[f"determ_n{n}_s{seed}_randn2"])
        assert torch.equal(gen.randint(0, 1000, (bs,)), gold[f"determ_n{n}_s{seed}_randint"])
    gen = get_generator("determ-indiv", 6, 3)
    assert torch.equal(gen.randn(4, 2, 3, 3), gold["indiv_n6_s3_randn"])
    gen = get_generator("determ", 64, 42)
    gen.rank, gen.world_size = 1, 4
    assert torch.equal(gen.randn(8, 4, 8, 8), gold["determ_n64_s42_rank1of4_randn"])


def test_random_util_batch_size_independence():
    from lfm_amd.sampler.random_util import get_generator

    a = get_generator("determ", 32, 1).randn(8, 3)
    b = get_generator("determ", 32, 1).randn(4, 3)
    assert torch.equal(a[:4], b)  # same leading rows whatever the batch size (the point of the class)
    g = get_generator("determ", 8, 1)
    g.rank, g.world_size = 3, 4
    idx = g._indices(4)
    assert idx.tolist() == [3, 7, 7, 7]  # clamped at num_samples-1 (random_util.py:64)


# ----------------------------------------------------------------------------- Karras samplers (reference sampler/karras_sample.py)
class _Field:
    def __init__(self, A):
        self.A = A

    def __call__(self, t, x, **kw):
        flat = x.flatten(1)
        return (torch.tanh(flat @ self.A) * (1.0 + t[:, None]) - 0.5 * flat).reshape(x.shape)


@pytest.mark.parametrize("sampler,steps", [("euler", 11), ("euler", 51), ("heun", 11), ("heun", 50), ("heun", 40)])
def test_karras_sample_matches_reference(golden_dir, sampler, steps):
    from lfm_amd.sampler.karras_sample import karras_sample

    gold = _load(golden_dir, "karras.pt")
    out = karras_sample(_Field(gold["A"]), gold["x"].clone(), steps=steps, model_kwargs={}, device="cpu", clip_denoised=False,
                        sigma_min=1e-5, sigma_max=1.0, s_tmin=0.0, s_tmax=1.0, s_churn=0.0, sampler=sampler, rho=1.0, ts=range(0, steps, 15))
    torch.testing.assert_close(out, gold[f"{sampler}_{steps}"], rtol=0, atol=0)


@pytest.mark.parametrize("kind", ["determ", "determ-indiv"])
def test_heun_keeps_the_reference_rng_stream(golden_dir, kind):
    """Two consecutive Heun batches from ONE stateful generator: batch 2's x_T depends on the (zero-weighted) noise tensors the
    reference draws during batch 1 (karras_sample.py:143-145).  With the default reference_rng (on for determ*) the product
    reproduces both batches bit-for-bit; with the draws skipped only the first one."""
    from lfm_amd.sampler.karras_sample import karras_sample
    from lfm_amd.sampler.random_util import get_generator

    gold = _load(golden_dir, "karras_rng.pt")
    field = _Field(gold["A"])
    kw = dict(steps=11, model_kwargs={}, device="cpu", clip_denoised=False, sigma_min=1e-5, sigma_max=1.0, s_tmin=0.0, s_tmax=1.0,
              s_churn=0.0, sampler="heun")
    gen = get_generator(kind, 12, 5)
    for b in range(2):
        x = gen.randn(4, 1, 4, 4)
        assert torch.equal(x, gold[f"{kind}_x{b}"]), b
        assert torch.equal(karras_sample(field, x.clone(), generator=gen, **kw), gold[f"{kind}_out{b}"])
    gen = get_generator(kind, 12, 5)
    x0 = gen.randn(4, 1, 4, 4)
    assert torch.equal(karras_sample(field, x0.clone(), generator=gen, reference_rng=False, **kw), gold[f"{kind}_out0"])  # same batch result
    assert not torch.equal(gen.randn(4, 1, 4, 4), gold[f"{kind}_x1"])  # ... but the stream has left the reference's


def test_heun_quirk_counts_nfe():
    """steps=40 default is frozen: a 50-point grid does 49 predictor + 39 corrector evaluations = 88 NFE."""
    from lfm_amd.sampler.karras_sample import karras_sample

    calls = []

    def model(t, x, **kw):
        calls.append(float(t[0]))
        return -x

    x = torch.ones(2, 1, 2, 2)
    karras_sample(model, x, steps=50, model_kwargs={}, device="cpu", clip_denoised=False, sigma_min=1e-5, sigma_max=1.0, sampler="heun")
    assert len(calls) == 88
    calls.clear()
    karras_sample(model, x, steps=50, model_kwargs={}, device="cpu", clip_denoised=False, sigma_min=1e-5, sigma_max=1.0, sampler="heun",
                  heun_reference_quirk=False)
    assert len(calls) == 49 + 49  # quirk off: `i < steps - 1` with steps = the real grid length covers every interval
    calls.clear()
    karras_sample(model, x, steps=51, model_kwargs={}, device="cpu", clip_denoised=False, sigma_min=1e-5, sigma_max=1.0, sampler="euler")
    assert len(calls) == 50


# ----------------------------------------------------------------------------- odeint: product vs oracle vs analytic
def _lin(t, y):
    return -y * (1.0 + 0.5 * torch.sin(3 * t))


@pytest.mark.parametrize("method,h", [("euler", 0.02), ("euler", 0.03), ("midpoint", 0.1), ("rk4", 0.1)])
def test_fixed_grid_matches_oracle(method, h):
    from lfm_amd.solvers import odeint

    y0 = torch.randn(3, 4, generator=torch.Generator().manual_seed(0))
    t = torch.tensor([1.0, 0.0])
    a = odeint(_lin, y0, t, method=method, options={"step_size": h})
    b = ode_ref.odeint(_lin, y0, t, method=method, options={"step_size": h})
    assert a.shape == (2, 3, 4)
    torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_euler_grid_and_nfe():
    from lfm_amd.solvers import odeint, torchdiffeq_euler_grid

    for h, n in ((0.02, 50), (0.01, 100), (0.1, 10), (0.03, 34)):
        ts, dts = torchdiffeq_euler_grid(h)
        assert dts.numel() == n and ts.numel() == n + 1
        assert float(ts[0]) == 1.0 and float(ts[-1]) == 0.0
        assert abs(float(dts.sum()) + 1.0) < 1e-6
        seen = []
        odeint(lambda t, y: (seen.append(float(t)), -y)[1], torch.ones(1), torch.tensor([1.0, 0.0]), method="euler", options={"step_size": h})
        assert len(seen) == n
        torch.testing.assert_close(torch.tensor(seen), ts[:-1], rtol=0, atol=0)
    ts, dts = torchdiffeq_euler_grid(0.03)
    assert abs(float(dts[-1]) + 0.01) < 1e-6  # short last step


def test_dopri5_matches_oracle_and_analytic():
    from lfm_amd.solvers import odeint

    y0 = torch.tensor([[1.0, 0.0], [0.5, -0.3]])
    w = 2.0

    def osc(t, y):  # harmonic oscillator, integrated backwards from t=1 to 0
        return torch.stack([y[:, 1] * w, -y[:, 0] * w], 1)

    t = torch.tensor([1.0, 0.0])
    st_a, st_b = {}, {}
    a = odeint(osc, y0, t, method="dopri5", rtol=1e-5, atol=1e-5, stats=st_a)
    b = ode_ref.odeint(osc, y0, t, method="dopri5", rtol=1e-5, atol=1e-5, stats=st_b)
    torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)
    assert st_a["steps"] == st_b["steps"] and st_a["accepted"] == st_b["accepted"]
    c, s = math.cos(-w), math.sin(-w)  # exact rotation by angle w*(0-1)
    exact = torch.stack([y0[:, 0] * c + y0[:, 1] * s, -y0[:, 0] * s + y0[:, 1] * c], 1)
    assert float((a[-1] - exact).abs().max()) < 2e-4


def test_fixed_grid_order_of_convergence():
    from lfm_amd.solvers import odeint

    y0 = torch.ones(1, dtype=torch.float64)
    f = lambda t, y: -y  # noqa
    exact = math.e  # y(0) = y(1) * e
    errs = {}
    for m in ("euler", "midpoint", "rk4"):
        errs[m] = [abs(float(odeint(f, y0, torch.tensor([1.0, 0.0], dtype=torch.float64), method=m, options={"step_size": h})[-1]) - exact)
                   for h in (0.1, 0.05)]
    assert 1.8 < errs["euler"][0] / errs["euler"][1] < 2.2
    assert 3.5 < errs["midpoint"][0] / errs["midpoint"][1] < 4.5
    assert 14 < errs["rk4"][0] / errs["rk4"][1] < 18


# ----------------------------------------------------------------------------- model-constructor boundary
def test_create_network_state_dict_matches_reference_names(golden_dir):
    from argparse import Namespace

    from lfm_amd.models import DiT, create_network

    rec = _load(golden_dir, "dit_tiny.pt")["cond"]
    c = rec["cfg"]
    m = DiT(img_resolution=c["img_resolution"], patch_size=c["patch"], in_channels=c["in_channels"], hidden_size=c["hidden"],
            depth=c["depth"], num_heads=c["heads"], label_dropout=c["label_dropout"], num_classes=c["num_classes"])
    ref_sd = rec["state_dict"]
    assert list(m.state_dict().keys()) == list(ref_sd.keys())
    assert all(m.state_dict()[k].shape == v.shape for k, v in ref_sd.items())
    m.load_state_dict(ref_sd, strict=True)
    net = create_network(Namespace(use_origin_adm=False, model_type="DiT-B/2", image_size=256, f=8, num_in_channels=4,
                                   label_dropout=0.0, num_classes=1))
    assert sum(p.numel() for p in net.parameters()) == 129_732_112  # == the reference DiT-B/2 constructor (checked against /root/reference; incl. the frozen pos_embed)
    # default init is adaLN-Zero (DiT.py:219-228)
    assert not bool(net.final_layer.linear.weight.any()) and not bool(net.blocks[0].adaLN_modulation[1].weight.any())
    torch.testing.assert_close(m.pos_embed, ref_sd["pos_embed"], rtol=0, atol=1e-6)


def test_cfg_without_classes_is_an_error():
    from argparse import Namespace

    from lfm_amd.test_flow_latent import make_model_kwargs

    with pytest.raises(ValueError):
        make_model_kwargs(Namespace(num_classes=1, cfg_scale=1.5, model_type="DiT-B/2"), torch.zeros(2, 4, 8, 8), None, "cpu")


# ----------------------------------------------------------------------------- C ABI
def test_c_abi_exports_every_declared_symbol():
    import ctypes

    from lfm_amd import _build

    lib_path = _build.build()
    lib = ctypes.CDLL(lib_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "lfm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # declarations only (comments name struct fields like lfm_dit_call.fold_ln)
    measure = re.findall(r"#ifdef LFM_MEASURE(.*?)#endif", hdr, flags=re.S)
    measure_names = set(re.findall(r"\b(lfm_[a-z0-9_]+)\s*\(", "".join(measure)))
    names = sorted(set(re.findall(r"\b(lfm_[a-z0-9_]+)\s*\(", hdr)) - measure_names)
    assert len(names) >= 12 and len(measure_names) == 5  # trace readers (3) + the per-kernel checksum pair
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lfm_hip.h but not exported"
    if os.environ.get("LFM_MEASURE") != "1":  # the shipped library carries no measurement-only entry point
        for n in measure_names:
            assert not hasattr(lib, n), f"{n} is measurement-only but exported by the shipped build"
    lib.lfm_strerror.restype = ctypes.c_char_p
    assert lib.lfm_strerror(0) == b"ok" and lib.lfm_abi_version() >= 1


def test_no_packed_fp32_op_sel():
    """Guard of the co-scheduling fix (csrc/common.h: fma_v; profiles/r05_cosched_root_cause.txt): a packed-fp32 instruction whose half takes its source
    from the OTHER register of the pair (op_sel) read that operand as 0.0 in lanes 48-63 when a foreign wave shared the SIMD.  The shipped library must not
    contain the form in ANY kernel: every gfx950 code object of the built .so is disassembled and scanned."""
    from lfm_amd import _build

    if not os.path.exists(_build.OBJDUMP):
        # wherever the library can be BUILT the guard must run: a toolchain without its disassembler is a broken install, not a reason to skip
        assert not os.path.exists(_build.HIPCC), f"hipcc is installed but {_build.OBJDUMP} is not: the op_sel guard cannot run"
        pytest.skip("no ROCm toolchain here (nothing can be built either)")
    found = _build.opsel_scan(_build.build())
    assert not found, "packed-fp32 instructions with op_sel (write the expression with fma_v / scalars instead):\n" + "\n".join(
        f"  {n} x {ins} op_sel:[{sel}] in {k}" for (k, ins, sel), n in sorted(found.items(), key=lambda x: -x[1])[:10])
    assert "-fno-slp-vectorize" in _build.FLAGS


def test_unet_and_edm_state_dict_names_match_reference(golden_dir):
    """Key-for-key (and shape-for-shape) identity with state dicts produced by the unmodified reference classes."""
    from lfm_amd.models.EDM import DhariwalUNet
    from lfm_amd.models.unet import UNetModel

    for which, rec in _load(golden_dir, "unet_tiny.pt").items():
        m = UNetModel(**rec["cfg"])
        assert list(m.state_dict().keys()) == list(rec["state_dict"].keys()), which
        assert all(m.state_dict()[k].shape == v.shape for k, v in rec["state_dict"].items())
    rec = _load(golden_dir, "edm_tiny.pt")
    m = DhariwalUNet(**rec["cfg"])
    assert list(m.state_dict().keys()) == list(rec["state_dict"].keys())
    assert all(m.state_dict()[k].shape == v.shape for k, v in rec["state_dict"].items())
    assert any(k.endswith("resample_filter") for k in rec["state_dict"])  # the up/down convolutions' buffers are mirrored too


def test_dit_workspace_requirement_is_monotone_in_the_batch():
    """lfm_dit_workspace_bytes is host-only arithmetic: small batches add split-K slabs, and the size for max_batch must still cover every
    smaller batch (lfm_amd/models/DiT.py keeps ONE workspace sized for the largest batch it has seen)."""
    import ctypes as C

    from lfm_amd import hip

    L = hip.lib()
    for kw in (dict(depth=24, hidden=1024, heads=16, mlp_hidden=4096, patch=2, in_ch=4, res=32, label_rows=1),     # DiT-L/2, T = 256
               dict(depth=12, hidden=768, heads=12, mlp_hidden=3072, patch=2, in_ch=4, res=32, label_rows=1001),   # DiT-B/2 class-conditional
               dict(depth=12, hidden=384, heads=6, mlp_hidden=1536, patch=2, in_ch=4, res=16, label_rows=11),     # T = 64
               dict(depth=28, hidden=1152, heads=16, mlp_hidden=4608, patch=2, in_ch=4, res=32, label_rows=1)):   # DiT-XL/2: head_dim 72
        s = hip.DitShape(**kw)
        sizes = [L.lfm_dit_workspace_bytes(C.byref(s), b) for b in range(1, 70)]
        assert all(v > 0 for v in sizes)
        assert all(b >= a for a, b in zip(sizes, sizes[1:])), kw
    bad = hip.DitShape(depth=28, hidden=1280, heads=16, mlp_hidden=5120, patch=2, in_ch=4, res=32, label_rows=1)  # head_dim 80: no kernel
    assert L.lfm_dit_workspace_bytes(C.byref(bad), 1) == 0


# ----------------------------------------------------------------------------- label bounds, buffer ownership, invalidation
def test_label_bounds_raise_like_nn_embedding():
    from lfm_amd import hip

    hip.check_labels(torch.tensor([0, 3, 9]), 10, "t")
    with pytest.raises(IndexError):
        hip.check_labels(torch.tensor([0, 10]), 10, "t")   # e.g. CFG's null class on a model built with label_dropout = 0
    with pytest.raises(IndexError):
        hip.check_labels(torch.tensor([-1, 2]), 10, "t")


def test_apply_without_a_move_keeps_the_packed_operands():
    """NFECount(model).to(device) / model.to(same device) must not drop packed weights, workspaces or captured graphs."""
    from lfm_amd.models import DiT_models

    m = DiT_models["DiT-S/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0).eval()
    m._packed, gen = "sentinel", m._gen
    m.to("cpu")
    m.float()
    assert m._packed == "sentinel" and m._gen == gen
    m.double()
    assert m._packed is None and m._gen == gen + 1


# ----------------------------------------------------------------------------- training pair (train_flow_latent.py:143-155)
def test_flow_matching_pair_defines_the_sampled_ode():
    from lfm_amd.train_flow_latent import SIGMA_MIN, flow_matching_loss, flow_matching_pair

    g = torch.Generator().manual_seed(0)
    z0, z1 = torch.randn(3, 4, 8, 8, generator=g), torch.randn(3, 4, 8, 8, generator=g)
    t = torch.tensor([0.0, 0.3, 1.0])
    zt, u, _ = flow_matching_pair(z0, t, z1)
    tt = t.view(-1, 1, 1, 1)
    assert torch.equal(zt, (1 - tt) * z0 + (1e-5 + (1 - 1e-5) * tt) * z1)   # the reference's line, verbatim arithmetic
    assert torch.equal(u, (1 - 1e-5) * z1 - z0)
    torch.testing.assert_close(zt[0], z0[0] + SIGMA_MIN * z1[0])            # t = 0: data (+ sigma_min noise)
    torch.testing.assert_close(zt[2], z1[2])                                # t = 1: noise
    # d z_t / dt = u: the ODE the samplers integrate from t = 1 to 0 carries z_1 back to z_0 + sigma_min z_1 when v = u
    eps = 1e-3
    zt2, _, _ = flow_matching_pair(z0, t + eps, z1)
    torch.testing.assert_close((zt2 - zt) / eps, u, rtol=1e-2, atol=1e-3)
    loss = flow_matching_loss(lambda t, x, y: torch.zeros_like(x), z0, generator=torch.Generator().manual_seed(1))
    assert float(loss) > 0


def test_unbuilt_dit_shapes_are_refused_at_construction():
    from lfm_amd.models import DiT_models

    from lfm_amd.models import DiT

    with pytest.raises(NotImplementedError):
        DiT(img_resolution=32, hidden_size=1280, depth=2, num_heads=16, num_classes=1, label_dropout=0.0)   # head_dim 80: no attention kernel
    DiT_models["DiT-XL/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)       # head_dim 72: built
    DiT_models["DiT-B/8"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)        # 16 tokens: built
    DiT_models["DiT-S/2"](img_resolution=64, in_channels=4, num_classes=1, label_dropout=0.0)        # 1024 tokens: built (four key chunks)
    with pytest.raises(NotImplementedError):
        DiT_models["DiT-S/2"](img_resolution=128, in_channels=4, num_classes=1, label_dropout=0.0)   # 4096 tokens: no attention kernel
    DiT_models["DiT-S/4"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0)        # 64 tokens, K = 64: built
    DiT_models["DiT-S/8"](img_resolution=64, in_channels=4, num_classes=1, label_dropout=0.0)        # 64 tokens, K = 256: built


def test_dit_plan_says_which_block_loop_runs():
    """lfm_dit_plan (no launch): the folded LayerNorm path and the fused QKV + attention kernel are taken exactly where their preconditions hold -- residual width a
    multiple of 256, whole 256-row tiles, a chip-filling batch; 256 tokens per image and head_dim 64 for the fused kernel -- and follow the library options and
    the per-call fold switch."""
    from lfm_amd import hip

    def shape(depth, hidden, heads, patch=2, res=32):
        return hip.DitShape(depth, hidden, heads, patch, 4, res, 4 * hidden, 1)

    L2, B2, XL2, S2, L4 = shape(24, 1024, 16), shape(12, 768, 12), shape(28, 1152, 16), shape(12, 384, 6), shape(24, 1024, 16, patch=4)
    both = hip.PLAN_FOLDED_LN | hip.PLAN_FUSED_QKV_ATTENTION
    assert hip.dit_plan(L2, 64) == both and hip.dit_plan(L2, 48) == both
    assert hip.dit_plan(L2, 47) == 0  # 188 tiles: not chip-filling
    assert hip.dit_plan(B2, 64, t_len=64, labels=True) == both and hip.dit_plan(B2, 512) == both
    assert hip.dit_plan(XL2, 64) == 0 and hip.dit_plan(S2, 256) == 0  # 1152 / 384 are not multiples of 256
    assert hip.dit_plan(L4, 256) == hip.PLAN_FOLDED_LN  # 64 tokens per image, one shared conditioning row: folded, but an image is not one 256-row tile
    assert hip.dit_plan(L4, 256, t_len=256, labels=True) == 0  # per-image rows need whole tiles of ONE image
    assert hip.dit_plan(L2, 64, fold_ln=hip.CALL_OFF) == 0
    hip.set_option(hip.OPT_FUSED_QKV_ATTENTION, 0)
    try:
        assert hip.dit_plan(L2, 64) == hip.PLAN_FOLDED_LN
    finally:
        hip.set_option(hip.OPT_FUSED_QKV_ATTENTION, 1)
    assert hip.dit_plan(L2, 64, gemm_select=hip.call_gemm_select(6)) == hip.PLAN_FOLDED_LN  # the one-wave-per-SIMD kernels keep the two-kernel form
    assert hip.dit_plan(L2, 64, gemm_select=hip.call_gemm_select(1)) == 0  # a forced 128x128 kernel: separate LayerNorm launches


def test_library_options_are_host_state_only():
    """lfm_set_option touches no device: unknown keys are refused, the folded-LayerNorm switch toggles (default on)."""
    from lfm_amd import hip

    L = hip.lib()
    assert L.lfm_set_option(99, 1) < 0
    with pytest.raises(hip.LfmHipError):
        hip.set_option(99, 1)
    hip.set_option(hip.OPT_FOLD_LN, 0)
    hip.set_option(hip.OPT_FOLD_LN, 1)
    hip.set_option(hip.OPT_FUSED_QKV_ATTENTION, 0)  # QKV projection + attention as one kernel (default on)
    hip.set_option(hip.OPT_FUSED_QKV_ATTENTION, 1)


def test_conv_split_k_plan_for_the_small_maps():
    """lfm_conv3x3_workspace_bytes is host-only arithmetic = the split-K plan of the UNets' low-resolution 3x3 convolutions (csrc/ops.hip): up to 256
    tiles of 128x128 split (the 16x16 maps of the celeb512 UNet at batch 32 would otherwise run one four-wave workgroup per CU), to at most 512
    workgroups, slices at least 128 deep and a multiple of 64; chip-filling problems need no workspace."""
    from lfm_amd import hip

    L = hip.lib()
    ws = lambda n, h, cin, cout: L.lfm_conv3x3_workspace_bytes(n, h, h, cin, cout)
    assert ws(32, 64, 256, 256) == 0 and ws(32, 32, 512, 512) == 0          # 2048 / 1024 tiles: no split
    # round 5: deep problems with >= 2048 rows whose 256x256 tiles x slices keep at least half the CUs busy slice on the 256x256 kernel (gemm_kernel.h:
    # splitk256_slices: the largest divisor of the K-tile count with slices >= 24 K-tiles and at most 256 workgroups); the workspace covers the larger plan
    assert ws(32, 16, 512, 512) == 3 * 8192 * 512 * 4                        # 64 tiles of 256x256 x 3 slices of 24 K-tiles (128x128 plan: two slices)
    assert ws(32, 16, 1024, 512) == 4 * 8192 * 512 * 4                       # 64 tiles x 4 slices of 36 K-tiles
    assert ws(64, 8, 768, 768) == 4 * 4096 * 768 * 4                         # EDM ffhq_adm 8x8 maps at batch 64: 48 tiles x 4 slices of 27 K-tiles
    assert ws(32, 8, 512, 512) == 8 * 2048 * 512 * 4                         # 64 tiles -> eight slices of K / 8 = 576
    assert ws(32, 4, 1024, 1024) == 16 * 512 * 1024 * 4                      # 32 tiles -> sixteen slices of 576
    assert ws(2, 8, 64, 128) == 0                                            # K = 576: a half (288) is not a multiple of the 64-deep K-tile
    assert ws(0, 8, 64, 128) == 0


def test_seeded_edm_state_is_order_independent_and_fp16_exact():
    """oracle/edm_state.py: the weights of the full-size EDM fixtures are regenerated from (tensor name, shape, seed) on both sides -- the reference module in
    oracle/make_golden.py, the product module in the GPU tests -- so the draw must not depend on the order parameters are registered in, and every value must
    be fp16-representable (the product packs GEMM operands in fp16: the fixture is about arithmetic, not weight rounding)."""
    from oracle.edm_state import seeded_edm_state

    names = [("enc.32x32_block0.conv0.weight", (8, 4, 3, 3)), ("enc.32x32_block0.conv1.weight", (8, 8, 3, 3)), ("enc.32x32_block0.norm0.weight", (8,)),
             ("enc.32x32_block0.conv0.bias", (8,)), ("map_label.weight", (16, 10)), ("enc.32x32_block0.conv0.resample_filter", (1, 1, 2, 2))]
    a = seeded_edm_state(names, 7)
    b = seeded_edm_state(list(reversed(names)), 7)
    assert set(a) == set(b) and "enc.32x32_block0.conv0.resample_filter" not in a
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], a[k].half().float())
    assert not torch.equal(a["enc.32x32_block0.conv0.weight"], seeded_edm_state(names, 8)["enc.32x32_block0.conv0.weight"])
    assert float(a["enc.32x32_block0.conv1.weight"].std()) < float(a["enc.32x32_block0.conv0.weight"].std())  # the reference's zero-init layers stay small
    assert abs(float(a["enc.32x32_block0.norm0.weight"].mean()) - 1.0) < 0.2



def test_concurrency_twin_shares_weights_not_scratch():
    """solvers.concurrency_twin: a second handle for a second HIP stream -- the same parameter tensors, its own workspace and its own solver cache."""
    from lfm_amd.models import DiT_models
    from lfm_amd.solvers import concurrency_twin

    m = DiT_models["DiT-S/2"](img_resolution=32, in_channels=4, num_classes=1, label_dropout=0.0).eval()
    m.__dict__["_fused_solvers"] = {"k": object()}
    m._ws = (4, torch.zeros(8))
    t = concurrency_twin(m)
    assert t is not m and type(t) is type(m)
    assert all(a is b for a, b in zip(m.parameters(), t.parameters()))  # shared weights
    assert t._ws is None and m._ws is not None  # own scratch
    assert "_fused_solvers" not in t.__dict__ and "_fused_solvers" in m.__dict__  # own captured solvers
    t._gen += 5
    assert m._gen != t._gen  # plain attributes are per handle
