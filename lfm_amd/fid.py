"""On-device FID statistics: the half of the reference's FID evaluation that can be built offline.

The reference decodes to JPEGs on a shared filesystem, re-reads them on rank 0, runs Inception-v3 there and reduces the
[n, 2048] pool3 activations with numpy (``pytorch_fid/fid_score.py:114-174`` get_activations, ``:230-251``
calculate_activation_statistics, ``:177-225`` calculate_frechet_distance).  Here the images never leave the GPU that
made them: every rank feeds its own uint8 NHWC batches to a feature extractor, accumulates

    n,  sum_i x_i,  sum_i x_i x_i^T        (fp64, on the device)

and ONE ``all_reduce`` of those sums (2048^2 + 2048 + 1 doubles = 33.6 MB, once per evaluation) replaces the all-gather of
pixels: mu = s / n, sigma = (S - n mu mu^T) / (n - 1) are exactly ``np.mean(act, 0)`` / ``np.cov(act, rowvar=False)`` of the
concatenated activations (tests/test_fid.py pins that against the reference-produced golden ``fid.pt`` and runs the reduction over a
world-2 gloo group).  The Frechet distance itself is ``lfm_amd.io_formats.frechet_distance`` (pinned to the reference's function).

What is NOT here: the Inception-v3 network.  ``pytorch_fid/inception.py:23`` downloads its weights by URL and builds on
torchvision; neither exists offline, and a from-memory restatement could not be checked against anything.  The extractor is
therefore a plug: ``--fid_feature_extractor pkg.module:factory`` names a callable ``factory(device) -> f`` with
``f(uint8 NHWC batch on device) -> [n, dims] float features``; without it ``--compute_fid`` keeps writing the reference's JPEG
tree for an external ``pytorch_fid`` run (lfm_amd/test_flow_latent{,_ddp}.py).
"""
import importlib

import numpy as np
import torch
import torch.distributed as dist


class FeatureStatistics:
    """Streaming, rank-sharded replacement of get_activations + calculate_activation_statistics (fid_score.py:114-174, 230-251)."""

    def __init__(self, dims=2048, device="cpu"):
        self.dims, self.device = int(dims), torch.device(device)
        self.n = torch.zeros((), dtype=torch.float64, device=self.device)
        self.s = torch.zeros(self.dims, dtype=torch.float64, device=self.device)
        self.ss = torch.zeros(self.dims, self.dims, dtype=torch.float64, device=self.device)
        self._reduced = False

    def update(self, feats):
        """feats: [n, dims] (any float dtype) on ``self.device``; pool3 maps [n, dims, 1, 1] are squeezed as the reference does (:163)."""
        if self._reduced:
            raise RuntimeError("statistics were already reduced across ranks; create a new FeatureStatistics")
        if feats.dim() == 4 and feats.shape[2:] == (1, 1):
            feats = feats[:, :, 0, 0]
        if feats.dim() != 2 or feats.shape[1] != self.dims:
            raise ValueError(f"expected [n, {self.dims}] features, got {tuple(feats.shape)}")
        x = feats.to(self.device, torch.float64)
        self.n += x.shape[0]
        self.s += x.sum(0)
        self.ss.addmm_(x.t(), x)
        return self

    def all_reduce(self):
        """Sum the three accumulators over the process group (a no-op outside one).  RCCL on GPUs, gloo in the CPU rehearsal."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not self._reduced:
            flat = torch.cat([self.n.reshape(1), self.s, self.ss.reshape(-1)])
            dist.all_reduce(flat)
            self.n, self.s, self.ss = flat[0], flat[1:1 + self.dims].clone(), flat[1 + self.dims:].reshape(self.dims, self.dims).clone()
        self._reduced = True
        return self

    def finalize(self):
        """(mu, sigma) as float64 numpy arrays: np.mean(act, axis=0), np.cov(act, rowvar=False) of everything that was fed."""
        n = float(self.n)
        if n < 2:
            raise ValueError("need at least two samples for a covariance")
        mu = self.s / n
        sigma = (self.ss - n * torch.outer(mu, mu)) / (n - 1.0)
        sigma = 0.5 * (sigma + sigma.t())  # the accumulated outer products are symmetric up to rounding; np.cov's output is exactly so
        return mu.cpu().numpy(), sigma.cpu().numpy()


def load_feature_extractor(spec, device):
    """``pkg.module:factory`` -> factory(device).  Raises with the reason when no extractor is configured."""
    if not spec:
        raise RuntimeError("no FID feature extractor configured: Inception-v3 (pytorch_fid/inception.py:23) needs torchvision and "
                           "downloaded weights, neither available offline; pass --fid_feature_extractor pkg.module:factory")
    mod, _, attr = spec.partition(":")
    if not attr:
        raise ValueError(f"--fid_feature_extractor expects 'pkg.module:factory', got {spec!r}")
    return getattr(importlib.import_module(mod), attr)(device)


def fid_against_reference_stats(stats, ref_stats_path):
    """fid_score.py:254-283: Frechet distance between the accumulated statistics and a precomputed (mu, sigma) file."""
    from .io_formats import frechet_distance, read_fid_stats

    mu, sigma = stats.finalize()
    mu_r, sigma_r = read_fid_stats(ref_stats_path)
    return float(frechet_distance(mu, sigma, np.asarray(mu_r), np.asarray(sigma_r)))
