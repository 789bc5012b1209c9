"""Latent-flow semantic synthesis (mask-to-image) sampler (drop-in for /root/reference/downstream_tasks/test_flow_latent_semantic_syn.py).

Per batch (reference :130-140): one-hot label map -> ``SpatialRescaler`` (three bilinear x0.5 stages + a 1x1 channel mapper,
models/encoder.py:90-112) -> 4-channel conditioning latent at the latent resolution, concatenated to the state (8 input channels of the
origin-ADM UNet); integrate from noise; decode.  The rescaler is a few hundred kFLOP per image: its bilinear resampling stays a torch elementwise op
(data preparation), its 1x1 channel mapper runs on the library's MFMA GEMM (``lfm_gemm_f16``, fp16 hi + lo split operands: the fp32 product to 2^-21) --
no vendor BLAS / MIOpen call on the product path; the UNet velocity field and the VAE decode run on the HIP path."""
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
        if not self.remap_output:
            return x
        return self._map_channels_hip(x) if x.is_cuda else self.channel_mapper(x)  # (CPU tensors: the oracle side of the tests)

    def _map_channels_hip(self, x):
        """The 1x1 ``channel_mapper`` convolution as C[pixels, Cout] = X[pixels, Cin] W^T on lfm_gemm_f16 (fp32 accumulate, fp32 output).  Activation and
        weight are split into fp16 hi + lo parts and three products are summed (hi hi + lo hi + hi lo; the dropped lo lo term is 2^-22 relative): the
        fp32 convolution of the reference to rounding, like the library's own input / output layers."""
        from .. import hip

        N, Cin, h, w = x.shape
        Cout = self.channel_mapper.out_channels
        K, Np = -(-Cin // 64) * 64, -(-Cout // 4) * 4  # the GEMM kernel's K-tile depth / 4-column granularity
        a32 = torch.zeros(N * h * w, K, device=x.device, dtype=torch.float32)
        a32[:, :Cin] = x.permute(0, 2, 3, 1).reshape(-1, Cin)
        w32 = torch.zeros(Np, K, device=x.device, dtype=torch.float32)
        w32[:Cout, :Cin] = self.channel_mapper.weight.detach().reshape(Cout, Cin).to(x.device, torch.float32)
        a_hi, w_hi = a32.half(), w32.half()
        a_lo, w_lo = (a32 - a_hi.float()).half(), (w32 - w_hi.float()).half()
        bias = torch.zeros(Np, device=x.device)
        if self.channel_mapper.bias is not None:
            bias[:Cout] = self.channel_mapper.bias.detach().to(x.device, torch.float32)
        out = hip.gemm_f16(a_hi, w_hi, bias, epilogue=2) + hip.gemm_f16(a_lo, w_hi, None, epilogue=2) + hip.gemm_f16(a_hi, w_lo, None, epilogue=2)
        return out[:, :Cout].reshape(N, h, w, Cout).permute(0, 3, 1, 2).contiguous()

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
