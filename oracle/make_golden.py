"""Generate ``tests/golden/*.pt`` by running the UNMODIFIED reference (oracle tooling).

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python -m oracle.make_golden            # rewrites tests/golden/

What is imported from the reference, as-is:
  * models/DiT.py           (behind oracle/timm_shim.py -- timm is not installed)
  * models/guided_diffusion/unet.py  (UNetModel)
  * sampler/karras_sample.py, sampler/random_util.py
  * pytorch_fid/fid_score.py::calculate_frechet_distance (loaded from its source file, torchvision / Inception imports stubbed)
Every fixture stores the reference state_dict (tiny configs, so the files stay small), the
seeded inputs and the reference outputs.  torchdiffeq / diffusers have no fixture: they are not
installable here (parity unpinned, see oracle/__init__.py).
"""
import os
import sys

import torch

REF = os.environ.get("LFM_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    from oracle import timm_shim

    timm_shim.install()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import models.DiT as ref_dit  # noqa
    import sampler.karras_sample as ref_karras  # noqa
    import sampler.random_util as ref_rand  # noqa

    return ref_dit, ref_karras, ref_rand


def _dezero_module(m, seed):
    from oracle.dit_ref import dezero_

    sd = m.state_dict()
    n = dezero_(sd, seed=seed)
    m.load_state_dict(sd)
    return n


def golden_dit(ref_dit):
    out = {}
    g = torch.Generator().manual_seed(42)
    for name, kw in {
        "cond": dict(num_classes=10, label_dropout=0.1),
        "uncond": dict(num_classes=1, label_dropout=0.0),
    }.items():
        torch.manual_seed(0)
        m = ref_dit.DiT(img_resolution=32, patch_size=2, in_channels=4, hidden_size=128, depth=2, num_heads=2, **kw).eval()
        nz = _dezero_module(m, 1234)
        # give biases some signal too (reference init zeroes every Linear bias, DiT.py:196-200)
        sd = m.state_dict()
        for k in sd:
            if k.endswith(".bias"):
                sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.02)
        m.load_state_dict(sd)
        x = torch.randn(3, 4, 32, 32, generator=g)
        rec = {"cfg": dict(depth=2, hidden=128, patch=2, heads=2, img_resolution=32, in_channels=4, **kw),
               "state_dict": {k: v.clone() for k, v in m.state_dict().items()}, "x": x, "dezeroed": nz}
        with torch.no_grad():
            t0 = torch.tensor(0.37)
            tN = torch.tensor([0.9, 0.5, 0.02])
            if name == "cond":
                y = torch.tensor([3, 0, 9])
                rec["y"] = y
                rec["v_t0d"] = m(t0, x, y)
                rec["v_tN"] = m(tN, x, y)
                rec["v_ynone"] = m(t0, x)  # y=None => null class row
                x2 = torch.cat([x[:2], x[:2]], 0)
                y2 = torch.tensor([3, 7, 10, 10])
                rec["x_cfg"], rec["y_cfg"], rec["cfg_scale"] = x2, y2, 1.5
                rec["v_cfg"] = m.forward_with_cfg(t0, x2, y2, cfg_scale=1.5)
            else:
                rec["v_t0d"] = m(t0, x)
                rec["v_tN"] = m(tN, x)
        out[name] = rec
    return out


def golden_dit_p4(ref_dit):
    """Patch size 4 (the DiT-x/4 family, models/DiT.py:358-371): 64 tokens of 4x4x4 = 64 inputs each; forward and forward_with_cfg."""
    g = torch.Generator().manual_seed(52)
    kw = dict(num_classes=10, label_dropout=0.1)
    torch.manual_seed(0)
    m = ref_dit.DiT(img_resolution=32, patch_size=4, in_channels=4, hidden_size=128, depth=1, num_heads=2, **kw).eval()
    nz = _dezero_module(m, 4321)
    sd = m.state_dict()
    for k in sd:
        if k.endswith(".bias"):
            sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.02)
    m.load_state_dict(sd)
    x = torch.randn(3, 4, 32, 32, generator=g)
    y = torch.tensor([3, 0, 9])
    rec = {"cfg": dict(depth=1, hidden=128, patch=4, heads=2, img_resolution=32, in_channels=4, **kw),
           "state_dict": {k: v.clone() for k, v in m.state_dict().items()}, "x": x, "y": y, "dezeroed": nz}
    with torch.no_grad():
        rec["v_tN"] = m(torch.tensor([0.9, 0.5, 0.02]), x, y)
        x2 = torch.cat([x[:2], x[:2]], 0)
        y2 = torch.tensor([3, 7, 10, 10])
        rec["x_cfg"], rec["y_cfg"], rec["cfg_scale"] = x2, y2, 1.5
        rec["v_cfg"] = m.forward_with_cfg(torch.tensor(0.37), x2, y2, cfg_scale=1.5)
    return rec


def golden_karras(ref_karras, ref_rand):
    """sample_euler / sample_heun on a cheap nonlinear field; includes the steps=40 quirk."""
    g = torch.Generator().manual_seed(7)
    A = torch.randn(16, 16, generator=g) * 0.3

    class Field:
        def __call__(self, t, x, **kw):
            flat = x.flatten(1)
            return (torch.tanh(flat @ A) * (1.0 + t[:, None]) - 0.5 * flat).reshape(x.shape)

        forward_with_cfg = None

    x = torch.randn(5, 1, 4, 4, generator=g)
    out = {"A": A, "x": x}
    gen = ref_rand.get_generator("dummy")
    for sampler, steps in (("euler", 11), ("euler", 51), ("heun", 11), ("heun", 50), ("heun", 40)):
        r = ref_karras.karras_sample(Field(), x.clone(), steps=steps, model_kwargs={}, device="cpu",
                                     clip_denoised=False, sigma_min=1e-5, sigma_max=1.0, s_tmin=0.0,
                                     s_tmax=1.0, s_churn=0.0, sampler=sampler, generator=gen)
        out[f"{sampler}_{steps}"] = r
    return out


def golden_karras_rng(ref_karras, ref_rand):
    """Two consecutive Heun batches drawn from ONE stateful reference generator: the second batch's x_T depends on the noise
    tensors sample_heun draws (and multiplies by zero) during the first (karras_sample.py:143-145, random_util.py:72-75)."""
    g = torch.Generator().manual_seed(7)
    A = torch.randn(16, 16, generator=g) * 0.3

    def field(t, x, **kw):
        flat = x.flatten(1)
        return (torch.tanh(flat @ A) * (1.0 + t[:, None]) - 0.5 * flat).reshape(x.shape)

    out = {"A": A}
    for kind in ("determ", "determ-indiv"):
        gen = ref_rand.get_generator(kind, 12, 5)
        for b in range(2):
            x = gen.randn(4, 1, 4, 4)
            gen.done_samples = 0  # the reference driver never advances it either (test_flow_latent.py:161-162)
            out[f"{kind}_x{b}"] = x
            out[f"{kind}_out{b}"] = ref_karras.karras_sample(field, x.clone(), steps=11, model_kwargs={}, device="cpu", clip_denoised=False,
                                                             sigma_min=1e-5, sigma_max=1.0, s_tmin=0.0, s_tmax=1.0, s_churn=0.0,
                                                             sampler="heun", generator=gen)
    return out


def golden_randgen(ref_rand):
    out = {}
    for n, seed, bs in ((64, 42, 8), (10, 7, 4)):
        gen = ref_rand.get_generator("determ", n, seed)
        out[f"determ_n{n}_s{seed}_randn"] = gen.randn(bs, 4, 8, 8)
        out[f"determ_n{n}_s{seed}_randn2"] = gen.randn(bs, 4, 8, 8)  # second draw advances the stream
        out[f"determ_n{n}_s{seed}_randint"] = gen.randint(0, 1000, (bs,))
    gen = ref_rand.get_generator("determ-indiv", 6, 3)
    out["indiv_n6_s3_randn"] = gen.randn(4, 2, 3, 3)
    # rank/world slicing (random_util.py:58-67) emulated by poking the attributes dist would set
    gen = ref_rand.get_generator("determ", 64, 42)
    gen.rank, gen.world_size = 1, 4
    out["determ_n64_s42_rank1of4_randn"] = gen.randn(8, 4, 8, 8)
    return out


def golden_unet():
    sys.path.insert(0, REF)
    from models.guided_diffusion.unet import UNetModel

    out = {}
    g = torch.Generator().manual_seed(11)
    base = dict(image_size=16, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, dropout=0.0, conv_resample=True,
                dims=2, use_checkpoint=False, use_fp16=False, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=True,
                resblock_updown=False, use_new_attention_order=False)
    cfgs = {
        # channels are multiples of 64 (the implicit-GEMM conv contract); attention at ds=2 -> T = 8x8 = 64 tokens, like celeb512's 8x8 level
        "ssn": dict(base, attention_resolutions=(2,), channel_mult=(1, 2), num_classes=None, num_heads=2),
        "cls": dict(base, attention_resolutions=(1, 2), channel_mult=(1, 1), num_classes=5, num_heads=4),
    }
    for name, kw in cfgs.items():
        torch.manual_seed(0)
        m = UNetModel(**kw).eval()
        _dezero_module(m, 4321)
        sd = m.state_dict()
        for k in sd:  # weights made fp16-representable so the fixture can be stored in half the bytes without changing the outputs
            sd[k].copy_(sd[k].half().float())
        m.load_state_dict(sd)
        x = torch.randn(2, 4, 16, 16, generator=g)
        t = torch.tensor([0.8, 0.1])
        rec = {"cfg": kw, "state_dict": {k: v.clone().half() for k, v in m.state_dict().items()}, "x": x, "t": t}
        with torch.no_grad():
            if kw["num_classes"]:
                y = torch.tensor([1, 4])
                rec["y"] = y
                rec["v"] = m(t, x, y)
            else:
                rec["v"] = m(t, x)
        out[name] = rec
    return out


def golden_unet_opts():
    """The three UNetModel options no shipped test_args file turns on (unet.py:131-238,341-369): plain emb-add ResBlocks
    (use_scale_shift_norm=False), ResBlock up/down sampling (resblock_updown=True), QKVAttention's channel order
    (use_new_attention_order=True) -- all on at once, and the up/down ResBlocks with the default FiLM conditioning."""
    sys.path.insert(0, REF)
    from models.guided_diffusion.unet import UNetModel

    out = {}
    g = torch.Generator().manual_seed(41)
    base = dict(image_size=16, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, dropout=0.0, conv_resample=True, dims=2,
                use_checkpoint=False, use_fp16=False, num_head_channels=-1, num_heads_upsample=-1, attention_resolutions=(2,), channel_mult=(1, 1),
                num_classes=None, num_heads=2)  # 64 channels throughout: the fixture stays ~1.5 MB per configuration
    for name, opt in {"all": (False, True, True), "updown": (True, True, False)}.items():  # every option on; up/down ResBlocks with FiLM and the legacy order
        kw = dict(base, use_scale_shift_norm=opt[0], resblock_updown=opt[1], use_new_attention_order=opt[2])
        torch.manual_seed(0)
        m = UNetModel(**kw).eval()
        _dezero_module(m, 555)
        sd = m.state_dict()
        for k in sd:
            sd[k].copy_(sd[k].half().float())
        m.load_state_dict(sd)
        x = torch.randn(2, 4, 16, 16, generator=g)
        t = torch.tensor([0.6, 0.3])
        with torch.no_grad():
            v = m(t, x)
        out[name] = {"cfg": kw, "state_dict": {k: v_.clone().half() for k, v_ in m.state_dict().items()}, "x": x, "t": t, "v": v}
    return out


def golden_inpaint():
    """The conditional velocity field of the downstream samplers: the unmodified reference UNetModel with 9 input channels behind the
    reference's WrapperCondFlow (downstream_tasks/test_flow_latent_inpainting.py:80-88 -- restated here in three lines because that file
    imports torchdiffeq / diffusers / torchvision and cannot be imported), two velocity evaluations and a 4-step explicit Euler solve
    from t = 1 to t = 0 (x <- x + dt * v(t, x), dt = -0.25: what torchdiffeq's fixed-grid euler does on this grid)."""
    sys.path.insert(0, REF)
    from models.guided_diffusion.unet import UNetModel

    g = torch.Generator().manual_seed(31)
    kw = dict(image_size=16, in_channels=9, model_channels=64, out_channels=4, num_res_blocks=1, dropout=0.0, conv_resample=True,
              dims=2, use_checkpoint=False, use_fp16=False, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=True,
              resblock_updown=False, use_new_attention_order=False, attention_resolutions=(2,), channel_mult=(1, 2), num_classes=None, num_heads=2)
    torch.manual_seed(0)
    m = UNetModel(**kw).eval()
    _dezero_module(m, 977)
    sd = m.state_dict()
    for k in sd:
        sd[k].copy_(sd[k].half().float())
    m.load_state_dict(sd)
    x = torch.randn(2, 4, 16, 16, generator=g)
    cond = torch.cat([torch.randn(2, 4, 16, 16, generator=g), (torch.rand(2, 1, 16, 16, generator=g) > 0.5).float() * 2 - 1], 1)

    def wrapped(t, xx):  # WrapperCondFlow.forward
        return m(t, torch.cat([xx, cond], 1))

    rec = {"cfg": kw, "state_dict": {k: v.clone().half() for k, v in m.state_dict().items()}, "x": x, "cond": cond}
    with torch.no_grad():
        rec["v_t1"] = wrapped(torch.tensor([1.0, 1.0]), x)
        rec["v_tN"] = wrapped(torch.tensor([0.7, 0.2]), x)
        xx = x.clone()
        for k in range(4):
            t = torch.full((2,), 1.0 - 0.25 * k)
            xx = xx + (-0.25) * wrapped(t, xx)
        rec["x_euler4"] = xx
    return rec


def golden_edm():
    """DhariwalUNet (models/EDM.py:716-861): plain forward with labels, and forward_with_cfg."""
    import models.EDM as ref_edm

    g = torch.Generator().manual_seed(21)
    kw = dict(img_resolution=16, in_channels=4, out_channels=4, label_dim=5, augment_dim=0, model_channels=64, channel_mult=[1, 2],
              channel_mult_emb=4, num_blocks=1, attn_resolutions=[8], dropout=0.0, label_dropout=0.1)
    torch.manual_seed(0)
    m = ref_edm.DhariwalUNet(**kw).eval()
    _dezero_module(m, 999)
    sd = m.state_dict()
    for k in sd:
        if sd[k].is_floating_point():
            sd[k].copy_(sd[k].half().float())
    m.load_state_dict(sd)
    x = torch.randn(4, 4, 16, 16, generator=g)
    y = torch.tensor([1, 4, 0, 2])
    rec = {"cfg": kw, "state_dict": {k: (v.clone().half() if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}, "x": x, "y": y}
    with torch.no_grad():
        rec["v_t0d"] = m(torch.tensor(0.6), x, y)
        rec["v_tN"] = m(torch.tensor([0.9, 0.5, 0.3, 0.05]), x, y)
        rec["v_nolabel"] = m(torch.tensor(0.6), x)
        rec["v_cfg"] = m.forward_with_cfg(torch.tensor(0.6), x, y, cfg_scale=1.7)
    return rec


def golden_fid():
    """pytorch_fid/fid_score.py::calculate_frechet_distance (pure numpy/scipy) on seeded statistics, incl. a rank-deficient pair.
    The module imports torchvision / the Inception wrapper at import time; neither is installed, so the function is loaded from its
    source file with those two imports stubbed (nothing of them is used by the function)."""
    import importlib.util
    import types

    import numpy as np

    for name in ("torchvision", "torchvision.transforms", "inception"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["inception"].InceptionV3 = type("InceptionV3", (), {"BLOCK_INDEX_BY_DIM": {64: 0, 192: 1, 768: 2, 2048: 3}})
    spec = importlib.util.spec_from_file_location("ref_fid_score", os.path.join(REF, "pytorch_fid", "fid_score.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(7)
    out = {"cases": []}
    for dims, n1, n2 in [(16, 60, 50), (64, 150, 130), (48, 30, 40)]:  # the last pair is rank-deficient (n < dims)
        a1 = rng.normal(size=(n1, dims)) * rng.uniform(0.5, 2.0, size=dims) + rng.normal(size=dims)
        a2 = rng.normal(size=(n2, dims)) * rng.uniform(0.5, 2.0, size=dims)
        m1, s1 = np.mean(a1, axis=0), np.cov(a1, rowvar=False)
        m2, s2 = np.mean(a2, axis=0), np.cov(a2, rowvar=False)
        out["cases"].append({"act1": torch.from_numpy(a1), "act2": torch.from_numpy(a2), "mu1": torch.from_numpy(m1), "sigma1": torch.from_numpy(s1),
                             "mu2": torch.from_numpy(m2), "sigma2": torch.from_numpy(s2), "fid": float(mod.calculate_frechet_distance(m1, s1, m2, s2))})
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_dit, ref_karras, ref_rand = _import_reference()
    torch.save(golden_dit(ref_dit), os.path.join(OUT, "dit_tiny.pt"))
    torch.save(golden_dit_p4(ref_dit), os.path.join(OUT, "dit_p4.pt"))
    torch.save(golden_karras(ref_karras, ref_rand), os.path.join(OUT, "karras.pt"))
    torch.save(golden_karras_rng(ref_karras, ref_rand), os.path.join(OUT, "karras_rng.pt"))
    torch.save(golden_randgen(ref_rand), os.path.join(OUT, "randgen.pt"))
    torch.save(golden_unet(), os.path.join(OUT, "unet_tiny.pt"))
    torch.save(golden_edm(), os.path.join(OUT, "edm_tiny.pt"))
    torch.save(golden_inpaint(), os.path.join(OUT, "inpaint_tiny.pt"))
    torch.save(golden_unet_opts(), os.path.join(OUT, "unet_opts.pt"))
    torch.save(golden_fid(), os.path.join(OUT, "fid.pt"))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()

