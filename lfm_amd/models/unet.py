"""Origin-ADM UNet velocity field, MI355X-native (drop-in for ``UNetModel`` of
/root/reference/models/guided_diffusion/unet.py:376-655, the backbone behind ``--use_origin_adm``).

Same constructor arguments, same parameter tree (``time_embed.{0,2}``, ``label_emb``, ``input_blocks.i.j.{in_layers.{0,2},
emb_layers.1,out_layers.{0,3},skip_connection,norm,qkv,proj_out,op}``, ``middle_block.{0,1,2}``, ``output_blocks.i.j...{conv}``,
``out.{0,2}``) so reference checkpoints load with ``strict=True``; same call contract ``model(t, x, y=None) -> v``.

The layer list is data-dependent, so the forward is sequenced here, on the host, over the NHWC-fp16 building blocks of
liblfm_hip.so (implicit-GEMM 3x3 convolutions incl. stride-2 / fused nearest-2x upsample, GroupNorm32 with the FiLM
scale-shift, small-T legacy attention, 1x1 convolutions with fused residual).  Graph-capturable; no PyTorch compute ops.

Built: ``use_scale_shift_norm=True`` (the sampler default, test_flow_latent.py:356), ``resblock_updown=False``,
``use_new_attention_order=False``, ``dims=2``, conv resampling -- i.e. every ``test_args/*_adm.txt`` with
``USE_ORIGIN_ADM=true``.  Other combinations raise ``NotImplementedError``.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import hip


class GroupNorm32(nn.GroupNorm):  # nn.py:17-19 (fp32 statistics; ours accumulates in fp32 too)
    pass


def normalization(channels):
    return GroupNorm32(32, channels)


class Upsample(nn.Module):  # unet.py:73-100
    def __init__(self, channels, use_conv, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if not use_conv:
            raise NotImplementedError("Upsample without conv (conv_resample=False) is not built")
        self.conv = nn.Conv2d(self.channels, self.out_channels, 3, padding=1)


class Downsample(nn.Module):  # unet.py:103-128
    def __init__(self, channels, use_conv, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if not use_conv:
            raise NotImplementedError("Downsample by average pooling (conv_resample=False) is not built")
        self.op = nn.Conv2d(self.channels, self.out_channels, 3, stride=2, padding=1)


class _Resample(nn.Module):
    """Parameter-free Upsample / Downsample (use_conv=False) inside a ResBlock(up= / down=): nearest 2x / 2x2 average pool (unet.py:73-128)."""

    def __init__(self, channels, up):
        super().__init__()
        self.channels, self.up = channels, up


class ResBlock(nn.Module):  # unet.py:131-238
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_scale_shift_norm=False, up=False, down=False):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_scale_shift_norm = use_scale_shift_norm
        self.updown = up or down
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), nn.Conv2d(channels, self.out_channels, 3, padding=1))
        if self.updown:  # same attribute names as the reference (they hold no parameters)
            self.h_upd, self.x_upd = _Resample(channels, up), _Resample(channels, up)
        else:
            self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1))
        for p in self.out_layers[-1].parameters():  # zero_module (unet.py:198)
            p.detach().zero_()
        self.skip_connection = nn.Identity() if self.out_channels == channels else nn.Conv2d(channels, self.out_channels, 1)


class AttentionBlock(nn.Module):  # unet.py:241-287
    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        self.use_new_attention_order = use_new_attention_order
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        assert channels % self.num_heads == 0
        self.norm = normalization(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = nn.Conv1d(channels, channels, 1)
        for p in self.proj_out.parameters():
            p.detach().zero_()


class TimestepEmbedSequential(nn.Sequential):
    pass


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, use_fp16=False,
                 num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False):
        super().__init__()
        if dims != 2 or use_fp16:
            raise NotImplementedError("only dims=2, use_fp16=False are built (the HIP path computes in fp16 operands / fp32 accumulate by itself)")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_classes = num_classes
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, ted)
        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, ch, 3, padding=1))])
        chans, ds = [ch], 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, dropout, out_channels=int(mult * model_channels), use_scale_shift_norm=use_scale_shift_norm)]
                ch = int(mult * model_channels)
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, num_heads=num_heads, num_head_channels=num_head_channels, use_new_attention_order=use_new_attention_order))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(
                    ResBlock(ch, ted, dropout, out_channels=ch, use_scale_shift_norm=use_scale_shift_norm, down=True) if resblock_updown
                    else Downsample(ch, conv_resample, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, ted, dropout, use_scale_shift_norm=use_scale_shift_norm),
            AttentionBlock(ch, num_heads=num_heads, num_head_channels=num_head_channels, use_new_attention_order=use_new_attention_order),
            ResBlock(ch, ted, dropout, use_scale_shift_norm=use_scale_shift_norm))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, out_channels=int(model_channels * mult), use_scale_shift_norm=use_scale_shift_norm)]
                ch = int(model_channels * mult)
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, num_heads=num_heads_upsample, num_head_channels=num_head_channels,
                                                 use_new_attention_order=use_new_attention_order))
                if level and i == num_res_blocks:
                    layers.append(ResBlock(ch, ted, dropout, out_channels=ch, use_scale_shift_norm=use_scale_shift_norm, up=True) if resblock_updown
                                  else Upsample(ch, conv_resample, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), nn.Conv2d(input_ch, out_channels, 3, padding=1))
        for p in self.out[-1].parameters():
            p.detach().zero_()
        self._packed = None
        self._scratch = None
        self._conv_ws = None
        self._film_all = None
        self._gen = 0  # bumped whenever device buffers a captured graph may point to are replaced

    # ---- packing ------------------------------------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        before = [(p.data_ptr(), p.dtype, p.device) for p in self.parameters()]
        out = super()._apply(fn, *a, **k)
        if before != [(p.data_ptr(), p.dtype, p.device) for p in self.parameters()]:  # only a real move / cast invalidates
            self._packed = None
            self._scratch = None
            self._conv_ws = None
            self._gen = getattr(self, "_gen", 0) + 1
        return out

    def load_state_dict(self, *a, **k):
        self._packed = None
        self._gen = getattr(self, "_gen", 0) + 1
        return super().load_state_dict(*a, **k)

    @torch.no_grad()
    def _pack(self):
        dev = self.out[2].weight.device
        hip.require_gpu(self.out[2].weight, "UNetModel")
        P = {}

        def f32(t):
            return t.detach().to(dev, torch.float32).contiguous()

        def f16(t):
            return t.detach().to(dev, torch.float16).contiguous()

        def conv3(m):
            w = m.weight
            if w.shape[1] % 64:
                raise hip.LfmHipError(f"3x3 conv with Cin={w.shape[1]}: the implicit-GEMM path needs Cin % 64 == 0")
            return f16(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)), f32(m.bias)

        for name, m in self.named_modules():
            if isinstance(m, ResBlock):
                P[name] = dict(gn1=(f32(m.in_layers[0].weight), f32(m.in_layers[0].bias)), c1=conv3(m.in_layers[2]),
                               emb=(f16(m.emb_layers[1].weight), f32(m.emb_layers[1].bias)),
                               gn2=(f32(m.out_layers[0].weight), f32(m.out_layers[0].bias)), c2=conv3(m.out_layers[3]),
                               skip=None if isinstance(m.skip_connection, nn.Identity) else
                               (f16(m.skip_connection.weight.reshape(m.out_channels, -1)), f32(m.skip_connection.bias)))
            elif isinstance(m, AttentionBlock):
                qw, qb = m.qkv.weight.reshape(3 * m.channels, -1), m.qkv.bias
                if m.use_new_attention_order:
                    # QKVAttention (unet.py:341-369) chunks [q | k | v] first and heads second; the kernel reads QKVAttentionLegacy's
                    # [head][q | k | v][ch] rows: a one-time row permutation of qkv.weight / bias makes the two the same computation
                    H, Cc = m.num_heads, m.channels // m.num_heads
                    perm = torch.arange(3 * m.channels, device=qw.device).reshape(3, H, Cc).permute(1, 0, 2).reshape(-1)
                    qw, qb = qw[perm], qb[perm]
                P[name] = dict(gn=(f32(m.norm.weight), f32(m.norm.bias)), qkv=(f16(qw), f32(qb)),
                               proj=(f16(m.proj_out.weight.reshape(m.channels, -1)), f32(m.proj_out.bias)))
            elif isinstance(m, Downsample):
                P[name] = conv3(m.op)
            elif isinstance(m, Upsample):
                P[name] = conv3(m.conv)
        # every ResBlock projects the SAME silu(emb) row through its own emb_layers Linear (unet.py:205-207): one GEMM per evaluation over
        # the concatenated weights (M = batch rows only -- 27 tiny launches otherwise), each block then reads its column slice
        names = [n for n, m in self.named_modules() if isinstance(m, ResBlock)]
        P["emb_all"] = (torch.cat([P[n]["emb"][0] for n in names], 0).contiguous(), torch.cat([P[n]["emb"][1] for n in names], 0).contiguous())
        off = 0
        for n in names:
            P[n]["emb_slice"] = (off, P[n]["emb"][0].shape[0])
            off += P[n]["emb"][0].shape[0]
            P[n]["emb"] = None  # the per-block copies are not needed on the device
        c0 = self.input_blocks[0][0]
        P["conv_in"] = (f32(c0.weight), f32(c0.bias))
        P["time"] = (f32(self.time_embed[0].weight), f32(self.time_embed[0].bias), f32(self.time_embed[2].weight), f32(self.time_embed[2].bias))
        P["label"] = f32(self.label_emb.weight) if self.num_classes is not None else None
        P["gn_out"] = (f32(self.out[0].weight), f32(self.out[0].bias))
        wo = self.out[2].weight
        w4 = torch.zeros(4, wo.shape[1], 3, 3, device=dev)
        w4[: wo.shape[0]] = wo
        b4 = torch.zeros(4, device=dev)
        b4[: wo.shape[0]] = self.out[2].bias
        if wo.shape[0] > 4:
            raise hip.LfmHipError("output conv with more than 4 channels is not built")
        P["conv_out"] = (f16(w4.permute(0, 2, 3, 1).reshape(4, -1)), f32(b4))
        self._packed = P
        self._gen += 1
        return P

    # ---- op helpers (all enqueue on torch's current stream) -------------------------------------------------------------
    def _gn(self, x, N, HW, Cch, gb, film, silu):
        y = torch.empty_like(x)
        need = hip.lib().lfm_groupnorm_scratch_bytes(N, Cch)
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != x.device:
            self._scratch = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=x.device)
            self._gen += 1
        hip.check(hip.lib().lfm_groupnorm_f16(hip.ptr(x), hip.ptr(y), hip.ptr(gb[0]), hip.ptr(gb[1]), hip.ptr(film),
                                              film.stride(0) if film is not None else 0, hip.ptr(self._scratch), N, HW, Cch, 32, 1e-5,
                                              1 if silu else 0, hip.stream_ptr(x.device)), "lfm_groupnorm_f16")
        return y

    def _gn2(self, xa, xb, N, HW, gb, film, silu):
        """GroupNorm of the channel concat [xa | xb] read in place (``th.cat([h, hs.pop()], dim=1)`` is never materialised, unet.py:649)."""
        Ca, Cb = xa.shape[1], xb.shape[1]
        y = torch.empty(xa.shape[0], Ca + Cb, dtype=torch.float16, device=xa.device)
        need = hip.lib().lfm_groupnorm_scratch_bytes(N, Ca + Cb)
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != xa.device:
            self._scratch = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=xa.device)
            self._gen += 1
        hip.check(hip.lib().lfm_groupnorm2_f16(hip.ptr(xa), Ca, hip.ptr(xb), Cb, hip.ptr(y), hip.ptr(gb[0]), hip.ptr(gb[1]), hip.ptr(film),
                                               film.stride(0) if film is not None else 0, hip.ptr(self._scratch), N, HW, 32, 1e-5,
                                               1 if silu else 0, hip.stream_ptr(xa.device)), "lfm_groupnorm2_f16")
        return y

    def _linear2(self, xa, xb, wb):
        M, Nout = xa.shape[0], wb[0].shape[0]
        out = torch.empty(M, Nout, dtype=torch.float16, device=xa.device)
        hip.check(hip.lib().lfm_linear2_f16(hip.ptr(xa), xa.shape[1], hip.ptr(xb), xb.shape[1], hip.ptr(wb[0]), wb[0].stride(0), hip.ptr(out), Nout, M, Nout,
                                            hip.ptr(wb[1]), None, hip.stream_ptr(xa.device)), "lfm_linear2_f16")
        return out

    def _cat(self, pair):
        h, skip = pair
        cat = torch.empty(h.shape[0], h.shape[1] + skip.shape[1], dtype=torch.float16, device=h.device)
        hip.check(hip.lib().lfm_concat_channels_f16(hip.ptr(h), hip.ptr(skip), hip.ptr(cat), h.shape[0], h.shape[1], skip.shape[1],
                                                    hip.stream_ptr(h.device)), "lfm_concat_channels_f16")
        return cat

    def _conv(self, x, wb, N, H, W, Cin, Cout, mode=0, resid=None):
        out = torch.empty(N * H * W, Cout, dtype=torch.float16, device=x.device)
        L = hip.lib()
        need = L.lfm_conv3x3_workspace_bytes(N, H, W, Cin, Cout)  # > 0 for the small-M / huge-K low-resolution levels: split-K slabs
        if need and (self._conv_ws is None or self._conv_ws.numel() < need or self._conv_ws.device != x.device):
            self._conv_ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            self._gen += 1
        ws = self._conv_ws if need else None
        hip.check(L.lfm_conv3x3_f16_ws(hip.ptr(x), hip.ptr(wb[0]), hip.ptr(wb[1]), hip.ptr(resid), hip.ptr(out), N, H, W, Cin, Cout, mode,
                                       hip.ptr(ws), ws.numel() if ws is not None else 0, hip.stream_ptr(x.device)), "lfm_conv3x3_f16_ws")
        return out

    def _linear(self, x, wb, resid=None):
        M, K = x.shape
        Nout = wb[0].shape[0]
        out = torch.empty(M, Nout, dtype=torch.float16, device=x.device)
        hip.check(hip.lib().lfm_linear_f16(hip.ptr(x), x.stride(0), hip.ptr(wb[0]), wb[0].stride(0), hip.ptr(out), Nout, M, Nout, K,
                                           hip.ptr(wb[1]), hip.ptr(resid), hip.stream_ptr(x.device)), "lfm_linear_f16")
        return out

    def _resblock(self, name, m, h, N, H, W, emb_silu):
        p = self._packed[name]
        Cin, Cout = m.channels, m.out_channels
        pair = None
        if isinstance(h, tuple):  # (h, skip) of an output block: read in place by the first GroupNorm and the 1x1 skip convolution
            if m.updown or p["skip"] is None or h[0].shape[1] % 64 or h[1].shape[1] % 8:
                h = self._cat(h)  # resampled / identity-skip inputs need the tensor itself
            else:
                pair = h
        if pair is not None:
            t1 = self._gn2(pair[0], pair[1], N, H * W, p["gn1"], None, True)
        else:
            t1 = self._gn(h, N, H * W, Cin, p["gn1"], None, True)
        if m.updown:  # in_rest -> h_upd / x_upd -> in_conv (unet.py:219-224): nearest 2x or 2x2 mean on both branches
            H2, W2 = (H * 2, W * 2) if m.h_upd.up else (H // 2, W // 2)
            t1, h = self._resample(t1, N, H2, W2, Cin, m.h_upd.up), self._resample(h, N, H2, W2, Cin, m.h_upd.up)
            H, W = H2, W2
        a = self._conv(t1, p["c1"], N, H, W, Cin, Cout)
        o, wdt = p["emb_slice"]
        emb_out = self._film_all[:, o:o + wdt]  # a column slice of the one emb GEMM of this evaluation
        if m.use_scale_shift_norm:  # fp32 [N, 2*Cout] = [scale | shift] folded into the GroupNorm affine (unet.py:228-232)
            t2 = self._gn(a, N, H * W, Cout, p["gn2"], emb_out, True)
        else:                       # h = h + emb_out[..., None, None]; out_layers(h)  (unet.py:233-235)
            a2 = torch.empty_like(a)
            hip.check(hip.lib().lfm_add_image_vec_f16(hip.ptr(a), hip.ptr(emb_out), emb_out.stride(0), hip.ptr(a2), N, H * W, Cout,
                                                      hip.stream_ptr(a.device)), "lfm_add_image_vec_f16")
            t2 = self._gn(a2, N, H * W, Cout, p["gn2"], None, True)
        if pair is not None:
            skip = self._linear2(pair[0], pair[1], p["skip"])
        else:
            skip = h if p["skip"] is None else self._linear(h, p["skip"])
        return self._conv(t2, p["c2"], N, H, W, Cout, Cout, resid=skip)

    def _resample(self, x, N, Ho, Wo, C, up):
        y = torch.empty(N * Ho * Wo, C, dtype=torch.float16, device=x.device)
        fn = hip.lib().lfm_upsample2_f16 if up else hip.lib().lfm_avgpool2_f16
        hip.check(fn(hip.ptr(x), hip.ptr(y), N, Ho, Wo, C, hip.stream_ptr(x.device)), "resample")
        return y

    def _attention(self, name, m, h, N, H, W):
        p = self._packed[name]
        Cch, T = m.channels, H * W
        t = self._gn(h, N, T, Cch, p["gn"], None, False)
        qkv = self._linear(t, p["qkv"])
        a = torch.empty(N * T, Cch, dtype=torch.float16, device=h.device)
        hip.check(hip.lib().lfm_attention_small_f16(hip.ptr(qkv), hip.ptr(a), N, T, m.num_heads, Cch // m.num_heads,
                                                    hip.stream_ptr(h.device)), "lfm_attention_small_f16")
        return self._linear(a, p["proj"], resid=h)

    def _run_block(self, prefix, block, h, N, H, W, emb_silu):
        for j, layer in enumerate(block):
            name = f"{prefix}.{j}"
            if isinstance(layer, ResBlock):
                h = self._resblock(name, layer, h, N, H, W, emb_silu)
                if layer.updown:
                    H, W = (H * 2, W * 2) if layer.h_upd.up else (H // 2, W // 2)
            elif isinstance(layer, AttentionBlock):
                h = self._attention(name, layer, h, N, H, W)
            elif isinstance(layer, Downsample):
                H, W = H // 2, W // 2
                h = self._conv(h, self._packed[name], N, H, W, layer.channels, layer.out_channels, mode=2)
            elif isinstance(layer, Upsample):
                H, W = H * 2, W * 2
                h = self._conv(h, self._packed[name], N, H, W, layer.channels, layer.out_channels, mode=1)
            else:
                raise TypeError(type(layer))
        return h, H, W

    # ---- forward ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, timesteps, x, y=None, **kwargs):
        """v = model(t, x, y) (unet.py:613-655).  t: 0-d / [1] / [N] (a scalar is broadcast; the reference's hard-coded
        ``device="cuda"`` at :630 is the tensor's device here)."""
        assert (y is not None) == (self.num_classes is not None), "must specify y if and only if the model is class-conditional"
        hip.require_gpu(x, "UNetModel.forward")
        if self.training:
            raise hip.LfmHipError("the HIP UNet is inference-only: call .eval()")
        if self._packed is None:
            self._pack()
        L, dev = hip.lib(), x.device
        x = x.contiguous().float()
        N, Cin, H, W = x.shape
        assert Cin == self.in_channels
        t = torch.as_tensor(timesteps, device=dev).float().reshape(-1).contiguous()
        if t.numel() not in (1, N):
            raise ValueError(f"timesteps must have 1 or {N} elements")
        E, F = self.model_channels * 4, self.model_channels
        if y is not None:
            y = y.to(dev, torch.long).contiguous()
            assert y.shape == (N,)
        n_labels = 0 if self._packed["label"] is None else int(self._packed["label"].shape[0])
        if y is not None and n_labels:
            hip.check_labels(y, n_labels, "UNetModel")
        emb = torch.empty(N, E, device=dev)
        emb_silu = torch.empty(N, E, device=dev, dtype=torch.float16)
        h1 = torch.empty(N, E, device=dev)
        tw = self._packed["time"]
        hip.check(L.lfm_time_embed(hip.ptr(t), t.numel(), hip.ptr(tw[0]), hip.ptr(tw[1]), hip.ptr(tw[2]), hip.ptr(tw[3]),
                                   hip.ptr(self._packed["label"]), hip.ptr(y), n_labels, hip.ptr(h1), hip.ptr(emb), hip.ptr(emb_silu), N, F, E,
                                   hip.stream_ptr(dev)), "lfm_time_embed")
        ea = self._packed["emb_all"]
        self._film_all = hip.gemm_f16(emb_silu, ea[0], ea[1], epilogue=2)  # every ResBlock's emb_layers in one launch
        ci = self._packed["conv_in"]
        ch0 = ci[0].shape[0]
        h = torch.empty(N * H * W, ch0, dtype=torch.float16, device=dev)
        hip.check(L.lfm_conv3x3_in_f32(hip.ptr(x), hip.ptr(ci[0]), hip.ptr(ci[1]), hip.ptr(h), N, H, W, Cin, ch0, hip.stream_ptr(dev)),
                  "lfm_conv3x3_in_f32")
        hs = [(h, ch0)]
        ch = ch0
        for i, block in enumerate(self.input_blocks):
            if i == 0:
                continue
            h, H, W = self._run_block(f"input_blocks.{i}", block, h, N, H, W, emb_silu)
            ch = h.shape[1]
            hs.append((h, ch))
        h, H, W = self._run_block("middle_block", self.middle_block, h, N, H, W, emb_silu)
        for i, block in enumerate(self.output_blocks):
            skip, cs = hs.pop()
            first = block[0]
            if isinstance(first, ResBlock):  # th.cat([h, hs.pop()], dim=1) (unet.py:649) is consumed in place by the ResBlock's GroupNorm and skip conv
                h, H, W = self._run_block(f"output_blocks.{i}", block, (h, skip), N, H, W, emb_silu)
            else:
                h, H, W = self._run_block(f"output_blocks.{i}", block, self._cat((h, skip)), N, H, W, emb_silu)
        t1 = self._gn(h, N, H * W, h.shape[1], self._packed["gn_out"], None, True)
        out = torch.empty(N, self.out_channels, H, W, device=dev)
        co = self._packed["conv_out"]
        hip.check(L.lfm_conv3x3_out_f32(hip.ptr(t1), hip.ptr(co[0]), hip.ptr(co[1]), hip.ptr(out), N, H, W, h.shape[1], self.out_channels,
                                        hip.stream_ptr(dev)), "lfm_conv3x3_out_f32")
        return out

    def forward_with_cfg(self, *a, **k):
        raise NotImplementedError("UNetModel has no forward_with_cfg in the reference either (unet.py:376-655): CFG needs DiT or the EDM adm")


_ = C
