"""Latent-flow semantic synthesis (mask-to-image) sampler (drop-in for /root/reference/downstream_tasks/test_flow_latent_semantic_syn.py).

Per batch (reference :130-140): one-hot label map -> ``SpatialRescaler`` (three bilinear x0.5 stages + a 1x1 channel mapper,
models/encoder.py:90-112) -> 4-channel conditioning latent at the latent resolution, concatenated to the state (8 input channels of the
origin-ADM UNet); integrate from noise; decode.  The rescaler is a few hundred kFLOP per image and stays torch ops on the GPU (plumbing);
the UNet velocity field and the VAE decode run on the HIP path."""
from functools import partial

import torch
from torch import nn

from . import ADAPTIVE_SOLVER, FIXER_SOLVER, WrapperCondFlow, sample_from_model  # noqa: F401

to_range_0_1 = lambda x: (x + 1.0) / 2.0  # noqa: E731


class SpatialRescaler(nn.Module):
    """Reference models/encoder.py:90-112, same parameter names (``channel_mapper``) so ``cond_stage_model_*.pth`` loads unchanged."""

    def __init__(self, n_stages=1, method="bilinear", multiplier=0.5, in_channels=3, out_channels=None, bias=False):
        super().__init__()
        assert n_stages >= 0 and method in ["nearest", "linear", "bilinear", "trilinear", "bicubic", "area"]
        self.n_stages, self.multiplier = n_stages, multiplier
        self.interpolator = partial(torch.nn.functional.interpolate, mode=method)
        self.remap_output = out_channels is not None
        if self.remap_output:
            self.channel_mapper = nn.Conv2d(in_channels, out_channels, 1, bias=bias)

    def forward(self, x):
        for _ in range(self.n_stages):
            x = self.interpolator(x, scale_factor=self.multiplier)
        return self.channel_mapper(x) if self.remap_output else x

    def encode(self, x):
        return self(x)


@torch.no_grad()
def synthesize_batch(model_cond, first_stage_model, cond_stage_model, segmentation, num_cls, args, z_0=None, generator=None):
    """segmentation: [N,S,S] int64 class ids.  Returns images in [0,1] (reference :131-140)."""
    seg = torch.nn.functional.one_hot(segmentation, num_cls).permute(0, 3, 1, 2).float()
    model_cond.cond = cond_stage_model(seg)
    if z_0 is None:
        z_0 = torch.randn(seg.size(0), 4, args.image_size // 8, args.image_size // 8, device=seg.device, generator=generator)
    fake_sample = sample_from_model(model_cond, z_0, args)[-1]
    return to_range_0_1(first_stage_model.decode(fake_sample / args.scale_factor).sample), fake_sample
