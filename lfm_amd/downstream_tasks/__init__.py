"""Conditional samplers of the reference's downstream tasks: the same ODE loop as the main sampling path, with a
conditioning latent concatenated to the state on every velocity evaluation (9 input channels for inpainting: 4 state + 4 masked-image
latent + 1 mask; 8 for semantic synthesis: 4 state + 4 rescaled label map).

Drop-in for /root/reference/downstream_tasks/test_flow_latent_inpainting.py:58-91 and test_flow_latent_semantic_syn.py:33-67
(``sample_from_model``, ``WrapperCondFlow``); the velocity field is the origin-ADM ``UNetModel`` on the HIP path
(``models.get_flow_model`` with ``num_in_channels`` 9 / 8, ``num_out_channels`` 4).
"""
import torch
from torch import nn

from ..solvers import ADAPTIVE_SOLVER, FIXER_SOLVER, fused_fixed_grid_available, odeint, sample_torchdiffeq_euler_fused

__all__ = ["ADAPTIVE_SOLVER", "FIXER_SOLVER", "WrapperCondFlow", "sample_from_model"]


class WrapperCondFlow(nn.Module):
    """``forward(t, x) = model(t, cat([x, cond], 1))`` (reference test_flow_latent_inpainting.py:80-88).

    The reference assigns ``wrapper.cond = c`` once per batch.  Here the assignment COPIES into a wrapper-owned buffer when the shape is
    unchanged, so a solver step captured in a hipGraph keeps reading the right memory; a new shape allocates and invalidates the graphs."""

    def __init__(self, model, cond=None):
        super().__init__()
        self.model = model
        self._cond = None
        self._own_gen = 0
        self.count_nfe = False
        self.nfe = 0
        if cond is not None:
            self.cond = cond

    @property
    def cond(self):
        return self._cond

    @cond.setter
    def cond(self, c):
        if c is None:
            self._cond = None
        elif self._cond is not None and self._cond.shape == c.shape and self._cond.device == c.device and self._cond.dtype == c.dtype:
            self._cond.copy_(c)
        else:
            self._cond = c.detach().clone()
            self._own_gen += 1

    # what the graph-captured fixed-grid solver asks of a velocity field
    @property
    def in_channels(self):  # channels of the ODE state, not of the wrapped network's input
        return self.model.out_channels

    @property
    def image_size(self):
        return self.model.image_size

    @property
    def _gen(self):
        return getattr(self.model, "_gen", 0) * 1_000_003 + self._own_gen

    def forward(self, t, x, y=None):
        if self._cond is None:
            raise RuntimeError("WrapperCondFlow: set .cond before sampling")
        if self.count_nfe:
            self.nfe += 1
        return self.model(t, torch.cat([x, self._cond], 1))


def sample_from_model(model, x_0, args):
    """Reference downstream_tasks/test_flow_latent_inpainting.py:58-77: integrate from t = 1 (noise) to t = 0; returns the [2, N, 4, h, w]
    end points (callers take [-1]).  Fixed-step Euler on the HIP UNet runs as one captured graph per interval (as the main sampler)."""
    if not getattr(args, "compute_fid", False) and hasattr(model, "count_nfe"):
        model.count_nfe = True  # reference :65-66
    if (args.method == "euler" and getattr(args, "fused", True) and not getattr(args, "perturb", False) and fused_fixed_grid_available(model, x_0)
            and not getattr(model, "count_nfe", False)):
        x1 = sample_torchdiffeq_euler_fused(model, x_0, args.step_size, {})
        return torch.stack([x_0, x1], 0)
    t = torch.tensor([1.0, 0.0], device=x_0.device)
    opts = {} if args.method in ADAPTIVE_SOLVER else {"step_size": args.step_size, "perturb": getattr(args, "perturb", False)}
    return odeint(model, x_0, t, method=args.method, atol=args.atol, rtol=args.rtol, options=opts)
