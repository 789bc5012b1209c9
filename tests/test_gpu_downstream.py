"""GPU parity of the downstream-task samplers: the conditional velocity field (9-channel origin-ADM UNet behind
WrapperCondFlow) against a golden produced by the UNMODIFIED reference UNetModel (tests/golden/inpaint_tiny.pt, oracle/make_golden.py::
golden_inpaint), the fused / graph-captured conditional Euler solve, and the whole inpainting batch (VAE encode -> solve -> decode ->
composite) against the oracle pieces."""
import os
from argparse import Namespace

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ode_ref, unet_ref, vae_ref  # checkers only


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def _golden_model(golden_dir, dev):
    from lfm_amd.models.unet import UNetModel

    rec = torch.load(os.path.join(golden_dir, "inpaint_tiny.pt"), map_location="cpu", weights_only=False)
    m = UNetModel(**rec["cfg"])
    m.load_state_dict({k: v.float() for k, v in rec["state_dict"].items()}, strict=True)
    return rec, m.to(dev).eval()


def test_conditional_velocity_and_euler_match_reference_golden(golden_dir):
    from lfm_amd.downstream_tasks import WrapperCondFlow, sample_from_model

    dev = torch.device("cuda:0")
    rec, m = _golden_model(golden_dir, dev)
    w = WrapperCondFlow(m, cond=rec["cond"].to(dev))
    x = rec["x"].to(dev)
    assert rel_l2(w(torch.tensor([1.0, 1.0], device=dev), x), rec["v_t1"]) < 3e-3
    assert rel_l2(w(torch.tensor([0.7, 0.2], device=dev), x), rec["v_tN"]) < 3e-3
    args = Namespace(method="euler", step_size=0.25, perturb=False, compute_fid=True, atol=1e-5, rtol=1e-5)
    fused = sample_from_model(w, x, args)
    assert fused.shape == (2, 2, 4, 16, 16) and torch.equal(fused[0], x)
    assert rel_l2(fused[-1], rec["x_euler4"]) < 2e-3
    args.fused = False
    eager = sample_from_model(w, x, args)[-1]
    assert rel_l2(fused[-1], eager) < 1e-6
    # a new condition of the same shape goes INTO the captured graph's buffer: the replayed solve must follow it
    cond2 = torch.roll(rec["cond"], 1, 0).to(dev)
    w.cond = cond2
    args.fused = True
    f2 = sample_from_model(w, x, args)[-1]
    args.fused = False
    assert rel_l2(f2, sample_from_model(w, x, args)[-1]) < 1e-6
    assert rel_l2(f2, fused[-1]) > 1e-3
    # NFE counting mode of the reference (:65-66): the generic loop, counted on the wrapper
    args.compute_fid = False
    w.nfe = 0
    sample_from_model(w, x, args)
    assert w.nfe == 4


def test_inpainting_batch_vs_oracle():
    """One full iteration of the reference loop (test_flow_latent_inpainting.py:141-160) at 64x64: VAE encode (posterior MODE, to take
    the sampling noise out of the comparison) -> 9-channel UNet flow, 4 Euler steps -> VAE decode -> composite."""
    from lfm_amd.autoencoder import AutoencoderKL
    from lfm_amd.downstream_tasks import WrapperCondFlow
    from lfm_amd.downstream_tasks.test_flow_latent_inpainting import inpaint_batch, synthetic_batch
    from lfm_amd.models import get_flow_model

    dev = torch.device("cuda:0")
    args = Namespace(image_size=64, num_in_channels=9, num_out_channels=4, nf=64, num_res_blocks=1, attn_resolutions=(2,), dropout=0.0,
                     ch_mult=(1, 2), num_classes=None, num_heads=2, num_head_channels=-1, num_head_upsample=-1, layout=False,
                     scale_factor=0.18215, method="euler", step_size=0.25, perturb=False, compute_fid=True, atol=1e-5, rtol=1e-5)
    cfg = dict(image_size=8, in_channels=9, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2),
               num_classes=None, num_heads=2, num_head_channels=-1, num_heads_upsample=-1)
    usd = unet_ref.make_unet_state(cfg, seed=7)
    m = get_flow_model(args)
    m.load_state_dict(usd, strict=True)
    m = m.to(dev).eval()
    vsd = vae_ref.make_vae_state(seed=3, with_encoder=True)
    vae = AutoencoderKL(with_encoder=True)
    vae.load_state_dict(vsd, strict=True)
    vae = vae.to(dev)
    image, mask, masked = synthetic_batch(2, 64, "cpu", seed=1)
    z0 = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(2))

    out, lat = inpaint_batch(WrapperCondFlow(m), vae, image.to(dev), mask.to(dev), masked.to(dev), args, z_0=z0.to(dev), sample_posterior=False)
    # oracle
    c = vae_ref.vae_encode_moments(vsd, masked)[:, :4] * args.scale_factor
    cond = torch.cat([c, F.interpolate(mask, size=c.shape[-2:])], 1)
    ref_lat = ode_ref.odeint(lambda t, x: unet_ref.unet_forward(usd, cfg, t, torch.cat([x, cond], 1)), z0, torch.tensor([1.0, 0.0]), method="euler",
                             options={"step_size": 0.25})[-1]
    ref_img = (vae_ref.vae_decode(vsd, ref_lat / args.scale_factor) + 1) / 2
    m01, i01 = (mask + 1) / 2, (image + 1) / 2
    ref_out = ref_img * m01 + (1 - m01) * i01
    assert rel_l2(lat, ref_lat) < 3e-3
    assert rel_l2(out, ref_out) < 1e-2
    keep = (m01 == 0).expand_as(i01)
    assert torch.equal(out.cpu()[keep], i01[keep])  # pixels outside the hole are the input image, exactly


def test_semantic_synthesis_batch_vs_oracle():
    """One batch of the reference loop (downstream_tasks/test_flow_latent_semantic_syn.py:130-140) at 64x64 against the CPU oracle: one-hot label map ->
    SpatialRescaler (three bilinear x0.5 stages + the 1x1 channel mapper, restated below on the CPU with the same weights) -> 8-channel origin-ADM
    UNet flow (oracle: unet_ref on the concatenated state), 2 Euler steps -> VAE decode."""
    from lfm_amd.autoencoder import AutoencoderKL
    from lfm_amd.downstream_tasks import WrapperCondFlow
    from lfm_amd.downstream_tasks.test_flow_latent_semantic_syn import SpatialRescaler, synthesize_batch
    from lfm_amd.models import get_flow_model

    dev = torch.device("cuda:0")
    args = Namespace(image_size=64, num_in_channels=8, num_out_channels=4, nf=64, num_res_blocks=1, attn_resolutions=(2,), dropout=0.0,
                     ch_mult=(1, 2), num_classes=None, num_heads=2, num_head_channels=-1, num_head_upsample=-1, layout=False,
                     scale_factor=0.18215, method="euler", step_size=0.5, perturb=False, compute_fid=True, atol=1e-5, rtol=1e-5)
    cfg = dict(image_size=8, in_channels=8, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2),
               num_classes=None, num_heads=2, num_head_channels=-1, num_heads_upsample=-1)
    usd = unet_ref.make_unet_state(cfg, seed=9)
    m = get_flow_model(args)
    m.load_state_dict(usd, strict=True)
    m = m.to(dev).eval()
    vsd = vae_ref.make_vae_state(seed=3)
    vae = AutoencoderKL()
    vae.load_state_dict(vsd, strict=True)
    vae = vae.to(dev)
    torch.manual_seed(5)
    cs = SpatialRescaler(n_stages=3, in_channels=19, out_channels=4, multiplier=0.5)
    wmap = cs.channel_mapper.weight.detach().clone()
    seg = torch.randint(0, 19, (2, 64, 64), generator=torch.Generator().manual_seed(3))
    z0 = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(4))
    img, lat = synthesize_batch(WrapperCondFlow(m), vae, cs.to(dev), seg.to(dev), 19, args, z_0=z0.to(dev))
    # oracle
    c = F.one_hot(seg, 19).permute(0, 3, 1, 2).float()
    for _ in range(3):
        c = F.interpolate(c, scale_factor=0.5, mode="bilinear")
    cond = F.conv2d(c, wmap)
    ref_lat = ode_ref.odeint(lambda t, x: unet_ref.unet_forward(usd, cfg, t, torch.cat([x, cond], 1)), z0, torch.tensor([1.0, 0.0]), method="euler",
                             options={"step_size": 0.5})[-1]
    ref_img = (vae_ref.vae_decode(vsd, ref_lat / args.scale_factor) + 1) / 2
    assert img.shape == (2, 3, 64, 64) and lat.shape == (2, 4, 8, 8)
    assert rel_l2(lat, ref_lat) < 3e-3
    assert rel_l2(img, ref_img) < 1e-2
