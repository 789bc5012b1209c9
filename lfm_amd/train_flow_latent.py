"""The flow-matching training pair of the reference's training step (train_flow_latent.py:143-155), host-side.

    z_t = (1 - t) z_0 + (1e-5 + (1 - 1e-5) t) z_1          z_1 ~ N(0, I) is noise (t = 1), z_0 the data latent (t = 0)
    u   = (1 - 1e-5) z_1 - z_0                              regression target of the velocity field
    loss = mse(model(t, z_t, y), u)

``encode_latents`` is the data side of that step (VAE encode on the HIP path, :141-143).  The backward pass / optimiser are outside the
scope of this package (training is out of scope; the HIP models are inference-only and raise in .train() mode): these
helpers exist so that a training loop written against the reference finds the same arithmetic and so the sampler tests can check that
the ODE they integrate is the one this pair defines (dz_t/dt = u)."""
import torch
import torch.nn.functional as F

SIGMA_MIN = 1e-5


def flow_matching_pair(z_0, t, z_1=None, generator=None):
    """(z_t, u, z_1) for data latents z_0 [N,...] and times t [N] (or broadcastable), reference :146-151."""
    t = t.reshape(-1, *([1] * (z_0.dim() - 1))).to(z_0.dtype)
    if z_1 is None:
        z_1 = torch.randn(z_0.shape, generator=generator, device=z_0.device, dtype=z_0.dtype)
    z_t = (1 - t) * z_0 + (SIGMA_MIN + (1 - SIGMA_MIN) * t) * z_1
    u = (1 - SIGMA_MIN) * z_1 - z_0
    return z_t, u, z_1


def flow_matching_loss(model, z_0, y=None, t=None, generator=None):
    """mse(model(t, z_t, y), u) with t ~ U(0, 1) per sample (reference :144-154)."""
    if t is None:
        t = torch.rand((z_0.size(0),), generator=generator, device=z_0.device, dtype=z_0.dtype)
    z_t, u, _ = flow_matching_pair(z_0, t, generator=generator)
    return F.mse_loss(model(t, z_t, y), u)


@torch.no_grad()
def encode_latents(first_stage_model, x_0, scale_factor=0.18215, is_latent_data=False, generator=None):
    """Reference :139-143: pre-encoded latents are only rescaled, images go through the VAE encoder."""
    if is_latent_data:
        return x_0 * scale_factor
    return first_stage_model.encode(x_0).latent_dist.sample(generator=generator).mul_(scale_factor)
