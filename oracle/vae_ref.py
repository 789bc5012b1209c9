"""Oracle restatement of the SD f8 KL-VAE *decoder* (TEST INFRASTRUCTURE ONLY).

**parity unpinned**: the reference obtains this network from the un-vendored, un-versioned
``diffusers`` package (requirements.txt:2) --
    /root/reference/test_flow_latent.py:101,131  AutoencoderKL.from_pretrained("stabilityai/sd-vae-ft-mse")
    /root/reference/test_flow_latent.py:193      first_stage_model.decode(fake_sample / scale_factor).sample
-- and neither diffusers nor the checkpoint is available here.  This file restates the published
architecture of ``AutoencoderKL`` (decoder half) for the ``sd-vae-ft-mse`` config:
latent 4, block_out_channels [128,256,512,512], layers_per_block 2 (=> 3 resnets per up block),
norm_num_groups 32, GN eps 1e-6, SiLU, single-head mid attention.
State-dict key names follow diffusers (>=0.2x) so a real checkpoint loads unchanged.
Pinned against an INDEPENDENT port of the same network (``tests/test_vae_ref_ldm.py``): the ``transformers`` package's Janus VQ-VAE is the
CompVis latent-diffusion Encoder / Decoder that ``AutoencoderKL`` ports; loaded with this file's state dict it reproduces ``vae_decode`` and
``vae_encode_moments`` to 1e-4.  Still unpinned against diffusers' own code and the real weights.
Further anchors (``tests/test_vae_ref.py``): the published parameter count (49,490,179 decoder + 20 post_quant_conv), the frozen diffusers key
list ``tests/golden/vae_decoder_keys.json``, mid attention vs ``F.scaled_dot_product_attention``, resnet / GroupNorm / upsample
identities, FLOP count 622.2 G.
"""
import math

import torch
import torch.nn.functional as F

BLOCK_OUT = (128, 256, 512, 512)
GROUPS = 32
EPS = 1e-6
LATENT = 4


def _gn(sd, pre, x):
    return F.group_norm(x, GROUPS, sd[pre + ".weight"], sd[pre + ".bias"], eps=EPS)


def _conv(sd, pre, x, pad):
    return F.conv2d(x, sd[pre + ".weight"], sd[pre + ".bias"], padding=pad)


def resnet(sd, pre, x):
    """diffusers ResnetBlock2D with temb=None, output_scale_factor=1."""
    h = _conv(sd, pre + ".conv1", F.silu(_gn(sd, pre + ".norm1", x)), 1)
    h = _conv(sd, pre + ".conv2", F.silu(_gn(sd, pre + ".norm2", h)), 1)
    if pre + ".conv_shortcut.weight" in sd:
        x = _conv(sd, pre + ".conv_shortcut", x, 0)
    return x + h


def mid_attention(sd, pre, x):
    """diffusers Attention, 1 head of dim C, GroupNorm on [B,C,HW], residual connection."""
    B, C, H, W = x.shape
    h = _gn(sd, pre + ".group_norm", x.reshape(B, C, H * W)).transpose(1, 2)  # [B,T,C]
    q = F.linear(h, sd[pre + ".to_q.weight"], sd[pre + ".to_q.bias"])
    k = F.linear(h, sd[pre + ".to_k.weight"], sd[pre + ".to_k.bias"])
    v = F.linear(h, sd[pre + ".to_v.weight"], sd[pre + ".to_v.bias"])
    a = (q @ k.transpose(1, 2) * C ** -0.5).softmax(dim=-1)
    o = F.linear(a @ v, sd[pre + ".to_out.0.weight"], sd[pre + ".to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


@torch.no_grad()
def vae_decode(sd, z):
    """``AutoencoderKL.decode(z).sample`` : [N,4,R,R] -> [N,3,8R,8R]."""
    h = _conv(sd, "post_quant_conv", z, 0)
    h = _conv(sd, "decoder.conv_in", h, 1)
    h = resnet(sd, "decoder.mid_block.resnets.0", h)
    h = mid_attention(sd, "decoder.mid_block.attentions.0", h)
    h = resnet(sd, "decoder.mid_block.resnets.1", h)
    for i in range(4):
        for j in range(3):
            h = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h)
        if i < 3:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", h, 1)
    h = F.silu(_gn(sd, "decoder.conv_norm_out", h))
    return _conv(sd, "decoder.conv_out", h, 1)


def downsample(sd, pre, x):
    """diffusers Downsample2D(use_conv=True, padding=0): F.pad(x, (0, 1, 0, 1)) then conv3x3 stride 2."""
    return F.conv2d(F.pad(x, (0, 1, 0, 1)), sd[pre + ".weight"], sd[pre + ".bias"], stride=2)


@torch.no_grad()
def vae_encode_moments(sd, x):
    """``AutoencoderKL.encode(x).latent_dist.parameters`` : [N,3,8R,8R] -> [N,8,R,R] (mean | logvar before the clamp).
    Restates diffusers' Encoder for the sd-vae-ft-mse config (block_out_channels [128,256,512,512], layers_per_block 2,
    double_z) followed by quant_conv; call sites train_flow_latent.py:143, downstream_tasks/test_flow_latent_inpainting.py:146."""
    h = _conv(sd, "encoder.conv_in", x, 1)
    for i in range(4):
        for j in range(2):
            h = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h)
        if i < 3:
            h = downsample(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", h)
    h = resnet(sd, "encoder.mid_block.resnets.0", h)
    h = mid_attention(sd, "encoder.mid_block.attentions.0", h)
    h = resnet(sd, "encoder.mid_block.resnets.1", h)
    h = F.silu(_gn(sd, "encoder.conv_norm_out", h))
    h = _conv(sd, "encoder.conv_out", h, 1)
    return _conv(sd, "quant_conv", h, 0)


def vae_encoder_layout():
    """(key prefix, kind, cin, cout) for the encoder half + quant_conv, in execution order."""
    L = [("encoder.conv_in", "conv3", 3, 128)]

    def res(pre, cin, cout):
        L.append((pre + ".norm1", "gn", cin, cin))
        L.append((pre + ".conv1", "conv3", cin, cout))
        L.append((pre + ".norm2", "gn", cout, cout))
        L.append((pre + ".conv2", "conv3", cout, cout))
        if cin != cout:
            L.append((pre + ".conv_shortcut", "conv1", cin, cout))

    cin = 128
    for i, cout in enumerate(BLOCK_OUT):
        for j in range(2):
            res(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i < 3:
            L.append((f"encoder.down_blocks.{i}.downsamplers.0.conv", "conv3", cout, cout))
        cin = cout
    res("encoder.mid_block.resnets.0", 512, 512)
    a = "encoder.mid_block.attentions.0"
    L.append((a + ".group_norm", "gn", 512, 512))
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        L.append((f"{a}.{n}", "linear", 512, 512))
    res("encoder.mid_block.resnets.1", 512, 512)
    L.append(("encoder.conv_norm_out", "gn", 512, 512))
    L.append(("encoder.conv_out", "conv3", 512, 8))
    L.append(("quant_conv", "conv1", 8, 8))
    return L


def vae_decoder_layout():
    """(key prefix, kind, cin, cout) for every parameterised layer, in execution order."""
    L = [("post_quant_conv", "conv1", LATENT, LATENT), ("decoder.conv_in", "conv3", LATENT, 512)]

    def res(pre, cin, cout):
        L.append((pre + ".norm1", "gn", cin, cin))
        L.append((pre + ".conv1", "conv3", cin, cout))
        L.append((pre + ".norm2", "gn", cout, cout))
        L.append((pre + ".conv2", "conv3", cout, cout))
        if cin != cout:
            L.append((pre + ".conv_shortcut", "conv1", cin, cout))

    res("decoder.mid_block.resnets.0", 512, 512)
    a = "decoder.mid_block.attentions.0"
    L.append((a + ".group_norm", "gn", 512, 512))
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        L.append((f"{a}.{n}", "linear", 512, 512))
    res("decoder.mid_block.resnets.1", 512, 512)
    cin = 512
    for i, cout in enumerate(reversed(BLOCK_OUT)):
        for j in range(3):
            res(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i < 3:
            L.append((f"decoder.up_blocks.{i}.upsamplers.0.conv", "conv3", cout, cout))
        cin = cout
    L.append(("decoder.conv_norm_out", "gn", 128, 128))
    L.append(("decoder.conv_out", "conv3", 128, 3))
    return L


def make_vae_state(seed=0, with_encoder=False):
    """Seeded random weights of the sd-vae-ft-mse architecture (no checkpoint on disk); the encoder half is drawn AFTER the decoder
    from the same generator, so the decoder weights of a seed do not depend on ``with_encoder``.

    Conv/linear weights ~ U(+-sqrt(3/fan_in)) (unit gain), biases N(0,0.02), GN affine 1+N(0,0.1)/N(0,0.1):
    keeps activations O(1) through 30 layers so fp16 paths are exercised without overflow."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for pre, kind, cin, cout in vae_decoder_layout() + (vae_encoder_layout() if with_encoder else []):
        if kind == "gn":
            sd[pre + ".weight"] = 1 + 0.1 * torch.randn(cin, generator=g)
            sd[pre + ".bias"] = 0.1 * torch.randn(cin, generator=g)
            continue
        k = 3 if kind == "conv3" else 1
        a = math.sqrt(3.0 / (cin * k * k))
        shape = (cout, cin) if kind == "linear" else (cout, cin, k, k)
        sd[pre + ".weight"] = (torch.rand(shape, generator=g) * 2 - 1) * a
        sd[pre + ".bias"] = 0.02 * torch.randn(cout, generator=g)
    return sd


def vae_decode_flops(R):
    """2*MAC of one decode at latent side R (622.2 GFLOP at R=32)."""
    mac, side = 0, R
    for pre, kind, cin, cout in vae_decoder_layout():
        if kind == "gn":
            continue
        if "upsamplers" in pre:
            side *= 2
        k = {"conv3": 9, "conv1": 1, "linear": 1}[kind]
        mac += side * side * cin * cout * k
    mac += 2 * (R * R) ** 2 * 512  # mid attention QK^T and PV
    return 2 * mac
