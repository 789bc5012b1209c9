"""The VAE restatement (oracle/vae_ref.py) against an INDEPENDENT implementation of the same published architecture.  CPU only.

diffusers is not installable, so ``vae_ref`` stays unpinned against diffusers itself.  But ``AutoencoderKL`` of sd-vae-ft-mse is diffusers'
port of the CompVis latent-diffusion ``Encoder`` / ``Decoder`` (kl-f8: ch 128, ch_mult (1, 2, 4, 4), 2 res blocks, no attention outside
the middle block, GroupNorm(32, eps 1e-6), swish, nearest-2x + conv upsampling, asymmetric-pad stride-2 downsampling) -- and the
``transformers`` package installed in this image carries ANOTHER port of that same code: the VQ-VAE of ``models/janus`` (written by other
people, sharing nothing with this repository).  Loading OUR diffusers-named state dict into THEIR modules under a key map and comparing
the outputs pins the block semantics (resnet order, 1-head middle attention with ``C ** -0.5``, where the upsamplers / downsamplers sit,
the (0, 1, 0, 1) padding, norm-out + swish + conv-out) to an independent source.  Janus adds attention blocks at its lowest-resolution
level, which kl-f8 does not have (``attn_resolutions = []``); they are removed -- its forward skips an empty list.  ``post_quant_conv`` /
``quant_conv`` live outside the Decoder / Encoder in both families and are applied here by hand."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import vae_ref

janus = pytest.importorskip("transformers.models.janus.modeling_janus")
janus_cfg = pytest.importorskip("transformers.models.janus.configuration_janus")


def _cfg(double_latent):
    return janus_cfg.JanusVQVAEConfig(base_channels=128, channel_multiplier=[1, 2, 4, 4], num_res_blocks=2, latent_channels=4, in_channels=3,
                                      out_channels=3, double_latent=double_latent, dropout=0.0)


def _put(conv_or_norm, sd, pre):
    w = sd[pre + ".weight"]
    if isinstance(conv_or_norm, nn.Conv2d) and w.dim() == 2:  # diffusers Linear (to_q, ...) == 1x1 convolution
        w = w[:, :, None, None]
    with torch.no_grad():
        assert conv_or_norm.weight.shape == w.shape, (pre, conv_or_norm.weight.shape, w.shape)
        conv_or_norm.weight.copy_(w)
        conv_or_norm.bias.copy_(sd[pre + ".bias"])


def _load_resnet(block, sd, pre):
    _put(block.norm1, sd, pre + ".norm1")
    _put(block.conv1, sd, pre + ".conv1")
    _put(block.norm2, sd, pre + ".norm2")
    _put(block.conv2, sd, pre + ".conv2")
    if pre + ".conv_shortcut.weight" in sd:
        _put(block.nin_shortcut, sd, pre + ".conv_shortcut")  # diffusers names its 1x1 shortcut conv_shortcut, ldm nin_shortcut
    else:
        assert not hasattr(block, "nin_shortcut")


def _load_mid(mid, sd, pre):
    _load_resnet(mid.block_1, sd, pre + ".resnets.0")
    _load_resnet(mid.block_2, sd, pre + ".resnets.1")
    a = pre + ".attentions.0"
    _put(mid.attn_1.norm, sd, a + ".group_norm")
    _put(mid.attn_1.q, sd, a + ".to_q")
    _put(mid.attn_1.k, sd, a + ".to_k")
    _put(mid.attn_1.v, sd, a + ".to_v")
    _put(mid.attn_1.proj_out, sd, a + ".to_out.0")


@pytest.fixture(scope="module")
def sd():
    return vae_ref.make_vae_state(seed=11, with_encoder=True)


def test_decoder_equals_the_independent_ldm_port(sd):
    dec = janus.JanusVQVAEDecoder(_cfg(False)).eval()
    dec.up[0].attn = nn.ModuleList()  # kl-f8 has no attention outside the middle block
    _put(dec.conv_in, sd, "decoder.conv_in")
    _load_mid(dec.mid, sd, "decoder.mid_block")
    for b in range(4):
        for r in range(3):
            _load_resnet(dec.up[b].block[r], sd, f"decoder.up_blocks.{b}.resnets.{r}")
        if b < 3:
            _put(dec.up[b].upsample.conv, sd, f"decoder.up_blocks.{b}.upsamplers.0.conv")
        else:
            assert not hasattr(dec.up[b], "upsample")
    _put(dec.norm_out, sd, "decoder.conv_norm_out")
    _put(dec.conv_out, sd, "decoder.conv_out")
    consumed = sum(p.numel() for p in dec.parameters())
    assert consumed == 49_490_179  # every decoder parameter of the published network found a home (and nothing else exists in theirs)
    z = torch.randn(2, 4, 6, 6, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = dec(F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"]))
        got = vae_ref.vae_decode(sd, z)
    assert got.shape == (2, 3, 48, 48) and float(want.abs().mean()) > 1e-3
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)


def test_encoder_equals_the_independent_ldm_port(sd):
    enc = janus.JanusVQVAEEncoder(_cfg(True)).eval()
    enc.down[-1].attn = nn.ModuleList()
    _put(enc.conv_in, sd, "encoder.conv_in")
    for b in range(4):
        for r in range(2):
            _load_resnet(enc.down[b].block[r], sd, f"encoder.down_blocks.{b}.resnets.{r}")
        if b < 3:
            _put(enc.down[b].downsample.conv, sd, f"encoder.down_blocks.{b}.downsamplers.0.conv")
    _load_mid(enc.mid, sd, "encoder.mid_block")
    _put(enc.norm_out, sd, "encoder.conv_norm_out")
    _put(enc.conv_out, sd, "encoder.conv_out")
    assert sum(p.numel() for p in enc.parameters()) == 34_163_592
    x = torch.randn(2, 3, 40, 40, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        want = F.conv2d(enc(x), sd["quant_conv.weight"], sd["quant_conv.bias"])
        got = vae_ref.vae_encode_moments(sd, x)
    assert got.shape == (2, 8, 5, 5)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)
