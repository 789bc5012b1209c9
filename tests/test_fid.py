"""On-device FID statistics (lfm_amd/fid.py) against the reference's numpy path.  CPU only (gloo for the 2-rank reduction).

Anchors: ``tests/golden/fid.pt`` holds activations, mu / sigma and the Frechet distances produced by the reference's own
``pytorch_fid/fid_score.py`` functions (oracle/make_golden.py::golden_fid); the streaming, rank-sharded accumulation must land on the same
numbers, and the DDP driver must report the same score as a direct computation over every rank's features."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _load(golden_dir):
    return torch.load(os.path.join(golden_dir, "fid.pt"), map_location="cpu", weights_only=False)["cases"]


def test_streaming_statistics_equal_the_reference_numpy_path(golden_dir):
    from lfm_amd.fid import FeatureStatistics
    from lfm_amd.io_formats import frechet_distance

    for case in _load(golden_dir):
        a1, a2 = case["act1"], case["act2"]
        s1, s2 = FeatureStatistics(a1.shape[1]), FeatureStatistics(a2.shape[1])
        for chunk in torch.split(a1.float().double(), 17):  # ragged batches
            s1.update(chunk)
        s2.update(a2.reshape(a2.shape[0], a2.shape[1], 1, 1))  # a pool3 map [n, dims, 1, 1] is squeezed as the reference does
        mu1, sg1 = s1.all_reduce().finalize()
        mu2, sg2 = s2.all_reduce().finalize()
        np.testing.assert_allclose(mu2, case["mu2"].numpy(), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(sg2, case["sigma2"].numpy(), rtol=1e-10, atol=1e-12)
        # act1 went through float32 on purpose (what a GPU extractor hands over): statistics agree to fp32 rounding of the inputs
        np.testing.assert_allclose(mu1, case["mu1"].numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(sg1, case["sigma1"].numpy(), rtol=0, atol=1e-5)
        fid = frechet_distance(case["mu1"].numpy(), case["sigma1"].numpy(), mu2, sg2)
        assert abs(fid - case["fid"]) < 1e-8 * max(1.0, abs(case["fid"]))
    with pytest.raises(ValueError):
        FeatureStatistics(8).update(torch.zeros(3, 9))
    with pytest.raises(ValueError):
        FeatureStatistics(8).update(torch.zeros(1, 8)).finalize()
    done = FeatureStatistics(4).update(torch.randn(5, 4)).all_reduce()
    with pytest.raises(RuntimeError):
        done.update(torch.randn(2, 4))


def test_feature_extractor_plug():
    from lfm_amd.fid import load_feature_extractor

    with pytest.raises(RuntimeError, match="Inception"):
        load_feature_extractor("", "cpu")
    with pytest.raises(ValueError):
        load_feature_extractor("tests.test_fid", "cpu")
    f = load_feature_extractor("tests.test_fid:stub_extractor_factory", "cpu")
    assert f(torch.zeros(2, 8, 8, 3, dtype=torch.uint8)).shape == (2, 6)


def stub_extractor_factory(device):
    """A deterministic stand-in for Inception pool3: per-channel mean and standard deviation of the uint8 image (6 features)."""

    def f(u8):
        x = u8.to(torch.float64) / 255.0
        return torch.cat([x.mean((1, 2)), x.std((1, 2))], 1)

    return f


# ----------------------------------------------------------------------------- 2 ranks: the sums are reduced, not the pixels
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_fid_worker(rank, world, port, q, stats_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from lfm_amd import test_flow_latent_ddp as ddp
    from lfm_amd.io_formats import to_uint8_truncating
    from tests.test_multi_rank import _StubModel, _StubVae

    feats = []
    extract = stub_extractor_factory("cpu")

    def recording(u8):
        f = extract(u8)
        feats.append(f.clone())
        return f

    def never_save(block, start):
        raise AssertionError("the statistics path must not write images")

    hooks = dict(backend="gloo", device="cpu", build_models=lambda a, d: (_StubModel().eval(), _StubVae()), to_uint8=to_uint8_truncating,
                 save=never_save, extractor=recording)
    argv = ["--model_type", "DiT-S/2", "--image_size", "32", "--num_in_channels", "4", "--n_sample", "20", "--batch_size", "5", "--method", "euler",
            "--step_size", "0.25", "--generator", "determ", "--seed", "3", "--compute_fid", "--real_img_dir", stats_path]
    res = ddp.main(argv, hooks=hooks)
    mu, sigma = res["fid_stats"].finalize()
    q.put((rank, res["total"], res["iters"], res.get("fid"), torch.cat(feats).numpy().tolist(), mu.tolist(), sigma.tolist()))


def test_world2_ddp_driver_reduces_feature_statistics(tmp_path):
    """--compute_fid with a feature extractor: each rank accumulates (n, sum x, sum x x^T) over its own batches, one all_reduce merges them,
    rank 0 reports the Frechet distance against the reference statistics file -- and nothing is gathered or written as pixels."""
    from lfm_amd.io_formats import activation_statistics, frechet_distance

    rng = np.random.default_rng(5)
    ref_act = rng.normal(size=(40, 6)) * 0.1 + 0.4
    mu_r, sg_r = activation_statistics(ref_act)
    stats_path = str(tmp_path / "ref_stats.npz")
    np.savez(stats_path, mu=mu_r, sigma=sg_r)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_fid_worker, args=(r, world, port, q, stats_path)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, total, iters, fid0, f0, mu0, sg0), (r1, total1, iters1, fid1, f1, mu1, sg1) = res
    assert (total, iters, total1, iters1) == (20, 2, 20, 2) and fid1 is None
    allf = np.concatenate([np.array(f0), np.array(f1)])
    assert allf.shape == (20, 6) and np.abs(np.array(f0) - np.array(f1)).max() > 0  # rank-local noise: different images per rank
    mu, sg = activation_statistics(allf)
    for got_mu, got_sg in ((mu0, sg0), (mu1, sg1)):  # every rank holds the global statistics after the all_reduce
        np.testing.assert_allclose(np.array(got_mu), mu, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(np.array(got_sg), sg, rtol=1e-9, atol=1e-13)
    assert abs(fid0 - frechet_distance(mu, sg, mu_r, sg_r)) < 1e-9
