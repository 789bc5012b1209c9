"""First-stage f8 KL-VAE, MI355X-native (drop-in for ``diffusers.models.AutoencoderKL`` as the reference uses it:
/root/reference/test_flow_latent.py:101,131,193, test_flow_latent_ddp.py:13,57,110 for decode;
train_flow_latent.py:143 and downstream_tasks/test_flow_latent_inpainting.py:146 for encode).

    vae = AutoencoderKL.from_pretrained(path_or_id).to(device)
    img = vae.decode(z / 0.18215).sample                              # [N,3,8R,8R]
    z   = vae.encode(img).latent_dist.sample().mul_(0.18215)          # [N,4,R,R]

Parameter names follow diffusers (``quant_conv``, ``post_quant_conv``, ``encoder.down_blocks.0.resnets.0.conv1``,
``decoder.mid_block.resnets.0.conv1`` ...), so a ``diffusion_pytorch_model.safetensors`` of ``stabilityai/sd-vae-ft-mse``
loads unchanged.  The arithmetic runs in liblfm_hip.so (lfm_vae_decode / lfm_vae_encode); there is no PyTorch fallback.
"""
import ctypes as C
import json
import math
import os

import torch
import torch.nn as nn

from . import hip

BLOCK_OUT = (128, 256, 512, 512)


class _Res(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("n1_g", "n1_b", "c1_w", "c1_b", "n2_g", "n2_b", "c2_w", "c2_b", "sc_w", "sc_b")] + [
        ("cin", C.c_int), ("cout", C.c_int)]


class _VaeWeights(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("pq_w", "pq_b", "cin_w", "cin_b")] + [("mid", _Res * 2)] +
                [(n, C.c_void_p) for n in ("at_g", "at_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b")] +
                [("up", (_Res * 3) * 4), ("ups_w", C.c_void_p * 3), ("ups_b", C.c_void_p * 3)] +
                [(n, C.c_void_p) for n in ("no_g", "no_b", "cout_w", "cout_b")])


class _VaeEncWeights(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("cin_w", "cin_b")] + [("down", (_Res * 2) * 4), ("ds_w", C.c_void_p * 3), ("ds_b", C.c_void_p * 3)] +
                [("mid", _Res * 2)] + [(n, C.c_void_p) for n in ("at_g", "at_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b")] +
                [(n, C.c_void_p) for n in ("no_g", "no_b", "cout_w", "cout_b")])


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class DiagonalGaussianDistribution:
    """diffusers' posterior object: ``parameters`` = [N, 2C, h, w] moments (mean | logvar), logvar clamped to [-30, 20]."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean

    def kl(self):
        return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


def _resnet(cin, cout):
    m = nn.Module()
    m.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
    m.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
    m.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
    m.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
    if cin != cout:
        m.conv_shortcut = nn.Conv2d(cin, cout, 1)
    return m


class _Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv_in = nn.Conv2d(4, 512, 3, padding=1)
        self.mid_block = nn.Module()
        self.mid_block.resnets = nn.ModuleList([_resnet(512, 512), _resnet(512, 512)])
        att = nn.Module()
        att.group_norm = nn.GroupNorm(32, 512, eps=1e-6)
        att.to_q, att.to_k, att.to_v = nn.Linear(512, 512), nn.Linear(512, 512), nn.Linear(512, 512)
        att.to_out = nn.ModuleList([nn.Linear(512, 512)])
        self.mid_block.attentions = nn.ModuleList([att])
        self.up_blocks = nn.ModuleList()
        cin = 512
        for i, cout in enumerate(reversed(BLOCK_OUT)):
            blk = nn.Module()
            blk.resnets = nn.ModuleList([_resnet(cin if j == 0 else cout, cout) for j in range(3)])
            if i < 3:
                up = nn.Module()
                up.conv = nn.Conv2d(cout, cout, 3, padding=1)
                blk.upsamplers = nn.ModuleList([up])
            self.up_blocks.append(blk)
            cin = cout
        self.conv_norm_out = nn.GroupNorm(32, 128, eps=1e-6)
        self.conv_out = nn.Conv2d(128, 3, 3, padding=1)


def _attention():
    att = nn.Module()
    att.group_norm = nn.GroupNorm(32, 512, eps=1e-6)
    att.to_q, att.to_k, att.to_v = nn.Linear(512, 512), nn.Linear(512, 512), nn.Linear(512, 512)
    att.to_out = nn.ModuleList([nn.Linear(512, 512)])
    return att


class _Encoder(nn.Module):
    """diffusers Encoder for block_out_channels (128, 256, 512, 512), layers_per_block 2, double_z (8 output channels)."""

    def __init__(self):
        super().__init__()
        self.conv_in = nn.Conv2d(3, 128, 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cin = 128
        for i, cout in enumerate(BLOCK_OUT):
            blk = nn.Module()
            blk.resnets = nn.ModuleList([_resnet(cin if j == 0 else cout, cout) for j in range(2)])
            if i < 3:
                ds = nn.Module()
                ds.conv = nn.Conv2d(cout, cout, 3, stride=2, padding=0)  # applied after F.pad(x, (0, 1, 0, 1))
                blk.downsamplers = nn.ModuleList([ds])
            self.down_blocks.append(blk)
            cin = cout
        self.mid_block = nn.Module()
        self.mid_block.resnets = nn.ModuleList([_resnet(512, 512), _resnet(512, 512)])
        self.mid_block.attentions = nn.ModuleList([_attention()])
        self.conv_norm_out = nn.GroupNorm(32, 512, eps=1e-6)
        self.conv_out = nn.Conv2d(512, 8, 3, padding=1)


class AutoencoderKL(nn.Module):
    """AutoencoderKL of the sd-vae-ft-mse architecture (encoder optional: the sampling path only decodes)."""

    @staticmethod
    def _max_chunk(R):
        """Images per launch such that every row-major GEMM operand of the full-resolution level (chunk * (8R)^2 pixels x up to 256 channels) stays below the
        2^31-element addressing limit of the GEMM kernels' 32-bit row offsets (csrc/gemm_kernel.h: ASrcRowMajor::fits): a user-set
        decode_chunk beyond it would otherwise be refused by the library (LFM_ERR_SHAPE) instead of being split."""
        return max(1, ((1 << 31) - 1) // (64 * R * R * 256))

    def __init__(self, decode_chunk=None, with_encoder=False):
        super().__init__()
        if with_encoder:
            self.encoder = _Encoder()
            self.quant_conv = nn.Conv2d(8, 8, 1)
        self.post_quant_conv = nn.Conv2d(4, 4, 1)
        self.decoder = _Decoder()
        self.decode_chunk = decode_chunk
        self.with_encoder = with_encoder
        self._packed = None
        self._packed_enc = None
        self._ws = None
        self.requires_grad_(False)

    # ---- construction
    @classmethod
    def from_pretrained(cls, path, **kw):
        """Load ``<path>/diffusion_pytorch_model.safetensors`` (or ``.bin``).  ``path`` must be a local directory:
        there is no network here, and unlike the reference we refuse to silently continue without weights."""
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"AutoencoderKL.from_pretrained({path!r}): not a local directory (hub ids cannot be fetched offline); "
                "use AutoencoderKL.from_random(seed) for synthetic-weight benchmarking")
        cfgp = os.path.join(path, "config.json")
        if os.path.exists(cfgp):
            cfg = json.load(open(cfgp))
            if tuple(cfg.get("block_out_channels", BLOCK_OUT)) != BLOCK_OUT or cfg.get("latent_channels", 4) != 4:
                raise ValueError(f"unsupported AutoencoderKL config in {cfgp}: only the sd-vae-ft-mse architecture is built")
        st = os.path.join(path, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file

            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "diffusion_pytorch_model.bin"), map_location="cpu", weights_only=True)  # safe unpickler only
        kw.setdefault("with_encoder", any(k.startswith("encoder.") for k in sd))
        m = cls(**kw)
        m.load_state_dict(sd)
        return m

    @classmethod
    def from_random(cls, seed=0, **kw):
        """Synthetic weights of the right architecture (unit-gain fan-in init keeps activations O(1))."""
        m = cls(**kw)
        g = torch.Generator().manual_seed(seed)
        for mod in m.modules():
            if isinstance(mod, (nn.Conv2d, nn.Linear)):
                fan_in = mod.weight[0].numel()
                a = math.sqrt(3.0 / fan_in)
                mod.weight.copy_((torch.rand(mod.weight.shape, generator=g) * 2 - 1) * a)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.02)
            elif isinstance(mod, nn.GroupNorm):
                mod.weight.copy_(1 + 0.1 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
        return m

    def load_state_dict(self, sd, strict=True):
        """Accepts a full AutoencoderKL checkpoint: encoder / quant_conv keys are dropped unless the module was built
        ``with_encoder``; the pre-0.2x attention names (query/key/value/proj_attn) are mapped to to_q/to_k/to_v/to_out.0."""
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        out = {}
        for k, v in sd.items():
            if not self.with_encoder and (k.startswith("encoder.") or k.startswith("quant_conv.")):
                continue
            parts = k.split(".")
            if "attentions" in parts and parts[-2] in ren:
                k = ".".join(parts[:-2] + [ren[parts[-2]], parts[-1]])
                if v.dim() == 4:
                    v = v.reshape(v.shape[0], v.shape[1])
            out[k] = v
        self._packed = None
        self._packed_enc = None
        return super().load_state_dict(out, strict=strict)

    def _apply(self, fn, *a, **k):
        self._packed = None
        self._packed_enc = None
        self._ws = None
        return super()._apply(fn, *a, **k)

    # ---- packing
    @torch.no_grad()
    def _pack(self):
        dev = self.post_quant_conv.weight.device
        hip.require_gpu(self.post_quant_conv.weight, "AutoencoderKL")
        keep = []

        def f32(t):
            keep.append(t.detach().to(dev, torch.float32).contiguous())
            return keep[-1].data_ptr()

        def f16(t):
            keep.append(t.detach().to(dev, torch.float16).contiguous())
            return keep[-1].data_ptr()

        def conv3(w):  # [Cout,Cin,3,3] -> [Cout, tap, Cin]
            return f16(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1))

        def res(m):
            r = _Res()
            r.n1_g, r.n1_b, r.c1_w, r.c1_b = f32(m.norm1.weight), f32(m.norm1.bias), conv3(m.conv1.weight), f32(m.conv1.bias)
            r.n2_g, r.n2_b, r.c2_w, r.c2_b = f32(m.norm2.weight), f32(m.norm2.bias), conv3(m.conv2.weight), f32(m.conv2.bias)
            if hasattr(m, "conv_shortcut"):
                r.sc_w, r.sc_b = f16(m.conv_shortcut.weight.reshape(m.conv_shortcut.weight.shape[0], -1)), f32(m.conv_shortcut.bias)
            r.cin, r.cout = m.conv1.in_channels, m.conv1.out_channels
            return r

        d = self.decoder
        w = _VaeWeights()
        w.pq_w, w.pq_b = f32(self.post_quant_conv.weight.reshape(4, 4)), f32(self.post_quant_conv.bias)
        w.cin_w, w.cin_b = f32(d.conv_in.weight), f32(d.conv_in.bias)
        w.mid[0], w.mid[1] = res(d.mid_block.resnets[0]), res(d.mid_block.resnets[1])
        a = d.mid_block.attentions[0]
        w.at_g, w.at_b = f32(a.group_norm.weight), f32(a.group_norm.bias)
        w.q_w, w.q_b, w.k_w, w.k_b = f16(a.to_q.weight), f32(a.to_q.bias), f16(a.to_k.weight), f32(a.to_k.bias)
        w.v_w, w.v_b, w.o_w, w.o_b = f16(a.to_v.weight), f32(a.to_v.bias), f16(a.to_out[0].weight), f32(a.to_out[0].bias)
        for i, blk in enumerate(d.up_blocks):
            for j in range(3):
                w.up[i][j] = res(blk.resnets[j])
            if i < 3:
                w.ups_w[i], w.ups_b[i] = conv3(blk.upsamplers[0].conv.weight), f32(blk.upsamplers[0].conv.bias)
        w.no_g, w.no_b = f32(d.conv_norm_out.weight), f32(d.conv_norm_out.bias)
        cw = torch.zeros(4, 128, 3, 3, device=dev)
        cw[:3] = d.conv_out.weight
        cb = torch.zeros(4, device=dev)
        cb[:3] = d.conv_out.bias
        w.cout_w, w.cout_b = conv3(cw), f32(cb)
        self._packed = (w, keep)
        return self._packed

    @torch.no_grad()
    def _pack_enc(self):
        if not self.with_encoder:
            raise hip.LfmHipError("this AutoencoderKL was built without the encoder half: AutoencoderKL(with_encoder=True) / from_pretrained")
        dev = self.post_quant_conv.weight.device
        hip.require_gpu(self.post_quant_conv.weight, "AutoencoderKL")
        keep = []

        def f32(t):
            keep.append(t.detach().to(dev, torch.float32).contiguous())
            return keep[-1].data_ptr()

        def f16(t):
            keep.append(t.detach().to(dev, torch.float16).contiguous())
            return keep[-1].data_ptr()

        def conv3(w):
            return f16(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1))

        def res(m):
            r = _Res()
            r.n1_g, r.n1_b, r.c1_w, r.c1_b = f32(m.norm1.weight), f32(m.norm1.bias), conv3(m.conv1.weight), f32(m.conv1.bias)
            r.n2_g, r.n2_b, r.c2_w, r.c2_b = f32(m.norm2.weight), f32(m.norm2.bias), conv3(m.conv2.weight), f32(m.conv2.bias)
            if hasattr(m, "conv_shortcut"):
                r.sc_w, r.sc_b = f16(m.conv_shortcut.weight.reshape(m.conv_shortcut.weight.shape[0], -1)), f32(m.conv_shortcut.bias)
            r.cin, r.cout = m.conv1.in_channels, m.conv1.out_channels
            return r

        e = self.encoder
        w = _VaeEncWeights()
        w.cin_w, w.cin_b = f32(e.conv_in.weight), f32(e.conv_in.bias)
        for i, blk in enumerate(e.down_blocks):
            for j in range(2):
                w.down[i][j] = res(blk.resnets[j])
            if i < 3:
                w.ds_w[i], w.ds_b[i] = conv3(blk.downsamplers[0].conv.weight), f32(blk.downsamplers[0].conv.bias)
        w.mid[0], w.mid[1] = res(e.mid_block.resnets[0]), res(e.mid_block.resnets[1])
        a = e.mid_block.attentions[0]
        w.at_g, w.at_b = f32(a.group_norm.weight), f32(a.group_norm.bias)
        w.q_w, w.q_b, w.k_w, w.k_b = f16(a.to_q.weight), f32(a.to_q.bias), f16(a.to_k.weight), f32(a.to_k.bias)
        w.v_w, w.v_b, w.o_w, w.o_b = f16(a.to_v.weight), f32(a.to_v.bias), f16(a.to_out[0].weight), f32(a.to_out[0].bias)
        w.no_g, w.no_b = f32(e.conv_norm_out.weight), f32(e.conv_norm_out.bias)
        # quant_conv (1x1, 8 -> 8) folded into conv_out (exact algebra, done in fp64 before the fp16 rounding of the GEMM operand)
        q = self.quant_conv.weight.detach().double().reshape(8, 8).to(dev)
        cw = torch.einsum("oi,ichw->ochw", q, e.conv_out.weight.detach().double().to(dev))
        cb = q @ e.conv_out.bias.detach().double().to(dev) + self.quant_conv.bias.detach().double().to(dev)
        w.cout_w, w.cout_b = conv3(cw.float()), f32(cb.float())
        self._packed_enc = (w, keep)
        return self._packed_enc

    # ---- encode
    @torch.no_grad()
    def encode(self, x, return_dict=True):
        """``AutoencoderKL.encode(x).latent_dist``: x [N,3,8R,8R] in [-1, 1] -> DiagonalGaussianDistribution over [N,4,R,R]."""
        hip.require_gpu(x, "AutoencoderKL.encode")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3] or x.shape[2] % 64:
            raise ValueError(f"x must be [N,3,S,S] with S % 64 == 0, got {tuple(x.shape)}")
        w, _ = self._packed_enc or self._pack_enc()
        x = x.contiguous().float()
        N, R = x.shape[0], x.shape[2] // 8
        chunk = min(self.decode_chunk or max(1, (64 * 32 * 32) // (R * R)), N, self._max_chunk(R))
        L = hip.lib()
        need = L.lfm_vae_workspace_bytes(R, chunk)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        moments = torch.empty(N, 8, R, R, device=x.device, dtype=torch.float32)
        rc = L.lfm_vae_encode(C.byref(w), hip.ptr(self._ws), self._ws.numel(), hip.ptr(x), hip.ptr(moments), N, R, chunk, hip.stream_ptr(x.device))
        hip.check(rc, "lfm_vae_encode")
        dist = DiagonalGaussianDistribution(moments)
        return AutoencoderKLOutput(dist) if return_dict else (dist,)

    # ---- decode
    @torch.no_grad()
    def decode(self, z, return_dict=True):
        hip.require_gpu(z, "AutoencoderKL.decode")
        if z.dim() != 4 or z.shape[1] != 4 or z.shape[2] != z.shape[3] or z.shape[2] % 8:
            raise ValueError(f"z must be [N,4,R,R] with R % 8 == 0, got {tuple(z.shape)}")
        w, _ = self._packed or self._pack()
        z = z.contiguous().float()
        N, R = z.shape[0], z.shape[2]
        # images per pass: bigger is faster (57.9 ms vs 65.9 ms per 64 images for 64 vs 16 at 256x256) and bounded by the
        # 4 ping-pong activation buffers: auto = 64 images at 256x256 (8.6 GB), scaled by resolution (16 at 512x512)
        chunk = self.decode_chunk or max(1, (64 * 32 * 32) // (R * R))
        chunk = min(chunk, N, self._max_chunk(R))
        L = hip.lib()
        need = L.lfm_vae_workspace_bytes(R, chunk)
        if self._ws is None or self._ws.numel() < need or self._ws.device != z.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=z.device)
        out = torch.empty(N, 3, 8 * R, 8 * R, device=z.device, dtype=torch.float32)
        rc = L.lfm_vae_decode(C.byref(w), hip.ptr(self._ws), self._ws.numel(), hip.ptr(z), hip.ptr(out), N, R, chunk, hip.stream_ptr(z.device))
        hip.check(rc, "lfm_vae_decode")
        return DecoderOutput(out) if return_dict else (out,)

    def forward(self, z):
        return self.decode(z).sample


def images_to_uint8(x, rounding=False):
    """u8 NHWC = trunc(clamp((x+1)/2,0,1)*255) on the GPU (reference test_flow_latent_ddp.py:131-135); ``rounding=True`` adds the 0.5 of
    torchvision.utils.save_image, the single-process script's writer (test_flow_latent.py:264-269,297)."""
    hip.require_gpu(x, "images_to_uint8")
    x = x.contiguous().float()
    N, _, H, W = x.shape
    out = torch.empty(N, H, W, 3, dtype=torch.uint8, device=x.device)
    hip.check(hip.lib().lfm_images_to_uint8_mode(hip.ptr(x), hip.ptr(out), N, H, W, int(bool(rounding)), hip.stream_ptr(x.device)),
              "lfm_images_to_uint8_mode")
    return out
