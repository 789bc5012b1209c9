"""GPU parity of the EDM-style ADM UNet (``DhariwalUNet``, model_type "adm" without --use_origin_adm) against golden vectors
produced by the unmodified reference ``models/EDM.py`` (tests/golden/edm_tiny.pt, oracle/make_golden.py).  Tolerance 3e-3."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def test_edm_adm_matches_reference_golden(golden_dir):
    from lfm_amd.models.EDM import DhariwalUNet

    rec = torch.load(os.path.join(golden_dir, "edm_tiny.pt"), map_location="cpu", weights_only=False)
    dev = torch.device("cuda:0")
    m = DhariwalUNet(**rec["cfg"])
    m.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in rec["state_dict"].items()}, strict=True)
    m = m.to(dev).eval()
    x, y = rec["x"].to(dev), rec["y"].to(dev)
    assert float(rec["v_t0d"].abs().mean()) > 1e-2
    assert rel_l2(m(torch.tensor(0.6, device=dev), x, y), rec["v_t0d"]) < 3e-3
    assert rel_l2(m(torch.tensor([0.9, 0.5, 0.3, 0.05], device=dev), x, y), rec["v_tN"]) < 3e-3
    assert rel_l2(m(torch.tensor(0.6, device=dev), x), rec["v_nolabel"]) < 3e-3
    got = m.forward_with_cfg(torch.tensor(0.6, device=dev), x, y, cfg_scale=1.7)
    assert rel_l2(got, rec["v_cfg"]) < 3e-3
    assert torch.equal(got[:2], got[2:])


def test_create_network_dispatches_edm_adm():
    from argparse import Namespace

    from lfm_amd.models import create_network
    from lfm_amd.models.EDM import DhariwalUNet

    a = Namespace(use_origin_adm=False, model_type="adm", image_size=128, f=8, num_in_channels=4, num_out_channels=4, label_dim=0, nf=64,
                  ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), dropout=0.0, label_dropout=0.0)
    m = create_network(a)
    assert isinstance(m, DhariwalUNet)
    m = m.cuda().eval()
    for p in m.parameters():  # de-zero (init_zero convs make the default model output 0)
        if not bool(p.any()):
            torch.nn.init.normal_(p, std=0.02)
    m._packed = None
    v = m(torch.tensor(0.3).cuda(), torch.randn(2, 4, 16, 16).cuda())
    assert v.shape == (2, 4, 16, 16) and torch.isfinite(v).all() and float(v.abs().mean()) > 0


def _full(name, golden_dir, dev):
    from lfm_amd.models.EDM import DhariwalUNet
    from oracle.edm_state import load_seeded  # checker-side weight maker (the fixture holds inputs / reference outputs only)

    rec = torch.load(os.path.join(golden_dir, "edm_full.pt"), map_location="cpu", weights_only=False)[name]
    m = DhariwalUNet(**rec["cfg"])
    checksum = load_seeded(m, rec["state_seed"])
    assert abs(checksum - rec["state_checksum"]) <= 1e-9 * rec["state_checksum"], "the seeded state differs from the one the reference was run with"
    assert sum(p.numel() for p in m.parameters()) == rec["params"]
    return rec, m.to(dev).eval()


def test_edm_ffhq_adm_production_size_vs_reference(golden_dir):
    """test_args/{ffhq,bed}_adm.txt: nf 256, ch_mult 1 2 3 4 (256 / 512 / 768 / 1024 channels -- 768 and the 1792 / 1280-wide concatenations are not
    powers of two), attention at 16x16 (8 heads, 256 tokens), 8x8 (12 heads, 64 tokens) and 4x4 (16 heads, 16 tokens), two blocks per level, 406 M
    parameters; outputs of the unmodified reference models/EDM.py (oracle/make_golden.py::golden_edm_full).  Then the 10-step Euler solve on the
    torchdiffeq grid through the graph-captured fixed-grid solver against the reference's own loop."""
    from argparse import Namespace

    from lfm_amd.test_flow_latent import sample_from_model

    dev = torch.device("cuda:0")
    rec, m = _full("ffhq_adm", golden_dir, dev)
    x = rec["x"].to(dev)
    assert float(rec["v_t0d"].abs().mean()) > 1e-2
    assert rel_l2(m(torch.tensor(0.6, device=dev), x[:1]), rec["v_t0d"]) < 3e-3
    a = m(torch.tensor([0.9, 0.3], device=dev), x).clone()
    assert rel_l2(a, rec["v_tN"]) < 3e-3
    assert torch.equal(a, m(torch.tensor([0.9, 0.3], device=dev), x))  # bit-repeatable (deterministic GroupNorm statistics, no atomics)
    args = Namespace(method="euler", step_size=0.1, perturb=False, compute_nfe=False, cfg_scale=1.0, atol=1e-5, rtol=1e-5)
    fused = sample_from_model(m, x, {}, args)[-1]
    assert rel_l2(fused, rec["x_euler10"]) < 1e-3
    args.fused = False
    assert rel_l2(sample_from_model(m, x, {}, args)[-1], fused) < 1e-5


def test_edm_imnet_adm_production_size_vs_reference(golden_dir):
    """test_args/imnet_adm.txt (bash_scripts/run_test_cls.sh: label_dim 1000, label_dropout 0.1, CFG 1.25): class-conditional evaluation and
    forward_with_cfg (the dropped label of the second half, EDM.py:825-826) at production size."""
    dev = torch.device("cuda:0")
    rec, m = _full("imnet_adm", golden_dir, dev)
    x, y = rec["x"].to(dev), rec["y"].to(dev)
    assert rel_l2(m(torch.tensor(0.6, device=dev), x[:1], y[:1]), rec["v_t0d"]) < 3e-3
    assert rel_l2(m(torch.tensor([0.9, 0.3], device=dev), x, y), rec["v_tN"]) < 3e-3
    got = m.forward_with_cfg(torch.tensor(0.37, device=dev), rec["x_cfg"].to(dev), rec["y_cfg"].to(dev), cfg_scale=rec["cfg_scale"])
    assert rel_l2(got, rec["v_cfg"]) < 3e-3
    assert torch.equal(got[:1], got[1:])
