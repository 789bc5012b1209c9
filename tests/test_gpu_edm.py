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
