"""Wire / disk formats either side of the sampling path (lfm_amd/io_formats.py).  CPU only.

The Frechet distance is pinned by ``tests/golden/fid.pt``, produced by the reference's own ``calculate_frechet_distance``
(oracle/make_golden.py::golden_fid).  torchvision is not installable here, so ``save_image`` / ``make_grid`` themselves are
parity-unpinned: the tests state the documented conversion rules (rounding vs truncation) instead."""
import argparse
import os

import numpy as np
import pytest
import torch

from lfm_amd import io_formats as io

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_frechet_distance_matches_reference_golden():
    g = torch.load(os.path.join(GOLDEN, "fid.pt"), map_location="cpu", weights_only=False)
    assert len(g["cases"]) == 3
    for c in g["cases"]:
        m1, s1 = io.activation_statistics(c["act1"].numpy())
        m2, s2 = io.activation_statistics(c["act2"].numpy())
        np.testing.assert_allclose(m1, c["mu1"].numpy(), rtol=0, atol=1e-12)
        np.testing.assert_allclose(s1, c["sigma1"].numpy(), rtol=0, atol=1e-12)
        got = io.frechet_distance(m1, s1, m2, s2)
        assert abs(got - c["fid"]) <= 1e-9 * max(1.0, abs(c["fid"]))
    # identical statistics -> 0 (up to sqrtm round-off); shape mismatch raises like the reference's assert
    c = g["cases"][0]
    assert abs(io.frechet_distance(c["mu1"].numpy(), c["sigma1"].numpy(), c["mu1"].numpy(), c["sigma1"].numpy())) < 1e-6
    with pytest.raises(ValueError):
        io.frechet_distance(np.zeros(3), np.eye(3), np.zeros(4), np.eye(4))


def test_fid_stat_reader_accepts_both_layouts(tmp_path):
    mu, sigma = np.arange(5.0), np.eye(5) * 2
    np.savez(tmp_path / "s.npz", mu=mu, sigma=sigma)                         # pytorch_fid --save-stats layout
    np.save(tmp_path / "s.npy", {"mu": mu, "sigma": sigma}, allow_pickle=True)  # the repo's *_stat.npy layout (fid_score.py:258-259)
    for name in ("s.npz", "s.npy"):
        m, s = io.read_fid_stats(str(tmp_path / name))
        np.testing.assert_array_equal(m, mu)
        np.testing.assert_array_equal(s, sigma)


def test_checkpoint_layouts(tmp_path):
    sd = {"pos_embed": torch.randn(1, 4, 8), "blocks.0.attn.qkv.weight": torch.randn(24, 8), "modulex.bias": torch.zeros(2)}
    ddp = {"module." + k: v for k, v in sd.items()}
    content = {"epoch": 3, "global_step": 99, "args": argparse.Namespace(exp="x"), "model_dict": ddp, "optimizer": {"state": {}}, "scheduler": {}}
    torch.save(sd, tmp_path / "flat.pth")
    torch.save(ddp, tmp_path / "model_3.pth")
    torch.save(content, tmp_path / "content.pth")
    for name in ("flat.pth", "model_3.pth", "content.pth"):
        got = io.load_state_dict_file(str(tmp_path / name))
        assert sorted(got) == sorted(sd)          # 'modulex.bias' keeps its name: only a real 'module.' prefix is stripped
        for k in sd:
            assert torch.equal(got[k], sd[k])
    # a real content.pth also carries optimizer state with non-tensor leaves: still the safe unpickler
    content["optimizer"] = {"state": {0: {"step": 5, "exp_avg": torch.zeros(2)}}, "param_groups": [{"lr": 1e-4, "betas": (0.9, 0.999)}]}
    torch.save(content, tmp_path / "content2.pth")
    assert sorted(io.load_state_dict_file(str(tmp_path / "content2.pth"))) == sorted(sd)
    with pytest.raises(ValueError):
        io.extract_state_dict({"epoch": 1, "args": None})
    with pytest.raises(ValueError):
        io.extract_state_dict([1, 2, 3])


class _Evil:
    def __reduce__(self):
        return (os.system, ("echo pwned > /dev/null",))


def test_checkpoint_loader_never_falls_back_to_the_unsafe_unpickler(tmp_path):
    """A pickle the safe loader rejects must stay rejected: no automatic weights_only=False retry."""
    import pickle

    torch.save({"w": torch.zeros(1), "payload": _Evil()}, tmp_path / "evil.pth")
    with pytest.raises(pickle.UnpicklingError):
        io.load_state_dict_file(str(tmp_path / "evil.pth"))


def test_uint8_conversions_and_grid(tmp_path):
    # values chosen on both sides of the .5 boundary: rounding (save_image) and truncation (the DDP script) differ exactly there
    v01 = torch.tensor([0.0, 0.4 / 255, 0.5 / 255, 0.6 / 255, 100.49 / 255, 100.51 / 255, 254.6 / 255, 1.0, 1.2, -0.3])
    img01 = v01.clamp(0, 1).reshape(1, 1, 1, -1).expand(1, 3, 1, -1)
    r = io.to_uint8_rounding(img01)[0, 0, :, 0].tolist()
    assert r == [0, 0, 1, 1, 100, 101, 255, 255, 255, 0]
    t = io.to_uint8_truncating(img01 * 2 - 1)[0, 0, :, 0].tolist()
    assert t[:2] == [0, 0] and t[7:] == [255, 255, 0] and all(abs(a - b) <= 1 for a, b in zip(t, r))
    assert t[4] == 100 and t[3] in (0, 1)
    # sample sheet: 10 images, nrow 8 -> 2 rows x 8 columns, row-major, empty cells black
    imgs = torch.stack([torch.full((4, 6, 3), k, dtype=torch.uint8) for k in range(1, 11)])
    grid = io.make_grid_nhwc(imgs, nrow=8, padding=0)
    assert grid.shape == (8, 48, 3)
    assert int(grid[0, 0, 0]) == 1 and int(grid[0, 47, 0]) == 8 and int(grid[4, 0, 0]) == 9 and int(grid[4, 6, 0]) == 10 and int(grid[4, 12, 0]) == 0
    assert io.make_grid_nhwc(imgs[:3], nrow=8).shape == (4, 18, 3)     # fewer images than nrow: one row of 3
    with pytest.raises(NotImplementedError):
        io.make_grid_nhwc(imgs, padding=2)
    # files: global index naming j * world + rank + total, readable JPEGs of the right size
    from PIL import Image

    io.save_indexed_jpegs(imgs[:2], str(tmp_path / "out"), start_index=16, world_size=8, rank=3)
    assert sorted(os.listdir(tmp_path / "out")) == ["19.jpg", "27.jpg"]
    assert Image.open(tmp_path / "out" / "19.jpg").size == (6, 4)
    io.save_image_grid(torch.rand(10, 3, 4, 6), str(tmp_path / "sheet.jpg"))
    assert Image.open(tmp_path / "sheet.jpg").size == (48, 8)
