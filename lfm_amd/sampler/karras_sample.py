"""Fixed-grid Euler / Heun integrators on a linear time grid (drop-in for /root/reference/sampler/karras_sample.py).

``karras_sample(model, x_T, steps, ...)`` keeps the reference signature (karras_sample.py:7-26) and additionally
tolerates the two keyword arguments the reference's own caller passes but the reference function rejects
(``rho``, ``ts``; test_flow_latent.py:93-94 -- SURVEY.md fact 4), so the wrapper in lfm_amd.test_flow_latent works.

Two execution paths, same arithmetic:
  * generic: any callable ``model(t, x, **kw)`` -- a Python loop of tensor ops (used for parity tests on CPU and for
    models that are not ours);
  * fused (lfm_amd.solvers.GraphedFixedGrid): when ``model`` is the HIP DiT the whole step -- time-grid advance, velocity
    field, CFG combine and the x += dt*v update -- is one captured hipGraph replayed per step.

Reference quirk kept by default (``heun_reference_quirk=True``): ``sample_heun`` compares the step index with its own
``steps=40`` default, which ``karras_sample`` never forwards (karras_sample.py:37-40,129,155) -- the 2nd-order correction
is applied to intervals i < 39 only, whatever the grid length.
"""
import numpy as np
import torch as th

from .random_util import _Deterministic, get_generator


def karras_sample(model, x_T, steps, clip_denoised=True, progress=False, callback=None, model_kwargs=None, device=None,
                  sigma_min=0.002, sigma_max=80, sampler="heun", s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0,
                  generator=None, classifier=None, cond_func=None, rho=None, ts=None, heun_reference_quirk=True, fused=None,
                  reference_rng=None):
    if generator is None:
        generator = get_generator("dummy")
    model_kwargs = model_kwargs or {}
    sigmas = th.linspace(sigma_max, sigma_min, steps, device=device)
    if sampler not in ("heun", "euler"):
        raise KeyError(sampler)

    # sample_heun draws one noise tensor per interval even when it is multiplied by zero (karras_sample.py:143-145).  The value is
    # irrelevant, the RNG STATE is not: with the stateful determ / determ-indiv generators every later draw (the next batch's x_T and
    # labels) depends on it.  reference_rng (default: on for exactly those generators) performs and discards the draws.
    if reference_rng is None:
        reference_rng = isinstance(generator, _Deterministic)
    use_cfg = model_kwargs.get("cfg_scale", 1.0) > 1.0
    if classifier is None and not clip_denoised and callback is None:
        from ..solvers import fused_fixed_grid_available, sample_fixed_grid_fused

        if fused is not False and fused_fixed_grid_available(model, x_T) and (sampler == "euler" or s_churn == 0.0):
            heun_limit = (40 if heun_reference_quirk else steps) if sampler == "heun" else 0
            if sampler == "heun" and reference_rng:
                for _ in range(steps - 1):  # the draws do not depend on the trajectory: advance the RNG as the reference does
                    generator.randn_like(x_T)
            return sample_fixed_grid_fused(model, x_T, sigmas, model_kwargs, heun_limit=heun_limit)
    if fused is True:
        raise RuntimeError("fused=True but the fused path is not applicable (needs the HIP DiT on a GPU, no classifier/clip/callback)")

    def denoiser(x_t, sigma):
        if use_cfg:
            out = model.forward_with_cfg(sigma, x_t, **model_kwargs)
        else:
            out = model(sigma, x_t, **{k: v for k, v in model_kwargs.items() if k != "cfg_scale"})
        return out.clamp(-1, 1) if clip_denoised else out

    def cls_denoiser(x_t, sigma):
        return model(sigma, x_t) + cond_func(classifier, x_t, 1.0 - sigma, **model_kwargs)

    fn = cls_denoiser if classifier is not None else denoiser
    if sampler == "euler":
        return sample_euler(fn, x_T, sigmas, generator, progress=progress, callback=callback)
    return sample_heun(fn, x_T, sigmas, generator, progress=progress, callback=callback, s_churn=s_churn, s_tmin=s_tmin,
                       s_tmax=s_tmax, s_noise=s_noise, steps=40 if heun_reference_quirk else steps, reference_rng=reference_rng)


@th.no_grad()
def sample_euler(denoiser, x, sigmas, generator, progress=False, callback=None):
    """x <- x + v(x, sigma_i) * (sigma_{i+1} - sigma_i); NFE = len(sigmas) - 1 (karras_sample.py:85-118)."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        sigma = sigmas[i]
        d = denoiser(x, sigma * s_in)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "denoised": d})
        x = x + d * (sigmas[i + 1] - sigma)
    return x


@th.no_grad()
def sample_heun(distiller, x, sigmas, generator, progress=False, callback=None, steps=40, s_churn=0.0, s_tmin=0.0,
                s_tmax=float("inf"), s_noise=1.0, reference_rng=False):
    """Heun (karras_sample.py:121-161).  With s_churn = 0 the noise term is exactly zero; the reference still draws it
    (an 819 MB RNG call per interval at 50k samples).  The result of THIS batch is identical without the draw, so it is skipped
    unless ``reference_rng`` asks for the reference's RNG stream (stateful generators: see karras_sample)."""
    s_in = x.new_ones([x.shape[0]])
    x_next = x
    n = len(sigmas) - 1
    for i in range(n):
        t_cur, t_next = sigmas[i], sigmas[i + 1]
        x_cur = x_next
        gamma = min(s_churn / steps, np.sqrt(2) - 1) if s_tmin <= t_cur <= s_tmax else 0
        t_hat = th.as_tensor(t_cur + gamma * t_cur)
        if gamma == 0:
            x_hat = x_cur
            if reference_rng:
                generator.randn_like(x_cur)  # drawn and discarded: keeps the generator state on the reference's stream
        else:
            x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).sqrt() * s_noise * generator.randn_like(x_cur)
        d_cur = distiller(x_hat, t_hat * s_in)
        x_next = x_hat + (t_next - t_hat) * d_cur
        if i < steps - 1:
            d_prime = distiller(x_next, t_next * s_in)
            x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
    return x_next
