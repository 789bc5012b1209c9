"""Pin the oracle restatements against fixtures produced by the unmodified reference
(oracle/make_golden.py).  CPU only."""
import os

import pytest
import torch

from oracle import dit_ref


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), map_location="cpu", weights_only=False)


@pytest.mark.parametrize("which", ["cond", "uncond"])
def test_dit_ref_matches_reference(golden_dir, which):
    rec = _load(golden_dir, "dit_tiny.pt")[which]
    cfg = dit_ref.DiTCfg(**rec["cfg"])
    sd, x = rec["state_dict"], rec["x"]
    assert rec["dezeroed"] >= 6  # adaLN w/b per block + final: the zero-init trap is defused
    y = rec.get("y")
    v = dit_ref.dit_forward(sd, cfg, torch.tensor(0.37), x, y)
    assert float(rec["v_t0d"].abs().mean()) > 1e-3  # not comparing 0 with 0
    torch.testing.assert_close(v, rec["v_t0d"], rtol=1e-5, atol=1e-6)
    v = dit_ref.dit_forward(sd, cfg, torch.tensor([0.9, 0.5, 0.02]), x, y)
    torch.testing.assert_close(v, rec["v_tN"], rtol=1e-5, atol=1e-6)
    if which == "cond":
        v = dit_ref.dit_forward(sd, cfg, torch.tensor(0.37), x, None)
        torch.testing.assert_close(v, rec["v_ynone"], rtol=1e-5, atol=1e-6)
        v = dit_ref.dit_forward_with_cfg(sd, cfg, torch.tensor(0.37), rec["x_cfg"], rec["y_cfg"], rec["cfg_scale"])
        torch.testing.assert_close(v, rec["v_cfg"], rtol=1e-5, atol=1e-6)


def _hd72_state(rec):
    cfg = dit_ref.DiTCfg(**rec["cfg"])
    sd = dit_ref.make_dit_state(cfg, seed=rec["state_seed"])
    # the fixture holds no weights: the seeded state must be the one the reference ran with
    assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - rec["state_checksum"]) < 1e-6 * rec["state_checksum"]
    return cfg, sd


def test_dit_ref_head_dim_72_and_patch_4_match_reference(golden_dir):
    """The DiT-XL family's head size (1152 / 16 = 72, models/DiT.py:354-363) and the DiT-x/4 patch size, both produced by the unmodified
    reference (tests/golden/dit_hd72.pt, dit_p4.pt)."""
    rec = _load(golden_dir, "dit_hd72.pt")
    cfg, sd = _hd72_state(rec)
    assert cfg.hidden // cfg.heads == 72
    assert float(rec["v_tN"].abs().mean()) > 1e-3
    v = dit_ref.dit_forward(sd, cfg, torch.tensor([0.9, 0.5, 0.02]), rec["x"], rec["y"])
    torch.testing.assert_close(v, rec["v_tN"], rtol=1e-5, atol=1e-6)
    v = dit_ref.dit_forward_with_cfg(sd, cfg, torch.tensor(0.37), rec["x_cfg"], rec["y_cfg"], rec["cfg_scale"])
    torch.testing.assert_close(v, rec["v_cfg"], rtol=1e-5, atol=1e-6)
    rec = _load(golden_dir, "dit_p4.pt")
    cfg = dit_ref.DiTCfg(**rec["cfg"])
    v = dit_ref.dit_forward(rec["state_dict"], cfg, torch.tensor([0.9, 0.5, 0.02]), rec["x"], rec["y"])
    torch.testing.assert_close(v, rec["v_tN"], rtol=1e-5, atol=1e-6)


def test_pos_embed_closed_form(golden_dir):
    rec = _load(golden_dir, "dit_tiny.pt")["cond"]
    pe = dit_ref.sincos_pos_embed_2d(128, 16)
    torch.testing.assert_close(pe, rec["state_dict"]["pos_embed"], rtol=0, atol=1e-6)


def test_make_dit_state_names_match_reference(golden_dir):
    rec = _load(golden_dir, "dit_tiny.pt")["cond"]
    cfg = dit_ref.DiTCfg(**rec["cfg"])
    mine = dit_ref.make_dit_state(cfg, seed=3)
    assert set(mine) == set(rec["state_dict"])
    for k, v in rec["state_dict"].items():
        assert tuple(mine[k].shape) == tuple(v.shape), k
    v = dit_ref.dit_forward(mine, cfg, torch.tensor(0.5), rec["x"], rec["y"])
    assert torch.isfinite(v).all() and float(v.abs().mean()) > 1e-3


def test_flops_closed_form():
    assert abs(dit_ref.dit_flops_per_image(dit_ref.DiTCfg.named("DiT-L/2")) / 1e9 - 161.4) < 0.5
    assert abs(dit_ref.dit_flops_per_image(dit_ref.DiTCfg.named("DiT-B/2")) / 1e9 - 46.0) < 0.3


@pytest.mark.parametrize("which", ["ssn", "cls"])
def test_unet_ref_matches_reference(golden_dir, which):
    from oracle import unet_ref

    rec = _load(golden_dir, "unet_tiny.pt")[which]
    v = unet_ref.unet_forward(rec["state_dict"], rec["cfg"], rec["t"], rec["x"], rec.get("y"))
    assert float(rec["v"].abs().mean()) > 1e-2
    torch.testing.assert_close(v, rec["v"], rtol=1e-4, atol=1e-5)
    mine = unet_ref.make_unet_state(rec["cfg"], seed=1)
    assert set(mine) == set(rec["state_dict"])
    assert all(tuple(mine[k].shape) == tuple(rec["state_dict"][k].shape) for k in mine)
