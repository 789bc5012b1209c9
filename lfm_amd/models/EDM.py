"""EDM-style ADM UNet (``DhariwalUNet``), MI355X-native -- drop-in for the ``model_type == "adm"`` branch of
/root/reference/models/EDM.py (``DhariwalUNet`` :716-861, ``UNetBlock`` :188-292, ``get_edm_network`` :864-939): the
backbone of ``test_args/{bed,ffhq,imnet}_adm.txt`` (USE_ORIGIN_ADM=false), including its own ``forward_with_cfg``.

Same constructor arguments and parameter/buffer names (``map_layer0/1``, ``map_label``, ``enc.<res>x<res>_{conv,down,block<i>}``,
``dec.<res>x<res>_{in0,in1,up,block<i>}``, ``out_norm``, ``out_conv``, the ``resample_filter`` buffers of up/down convolutions), so
reference checkpoints load with ``strict=True``.  The forward is sequenced on the host over the same NHWC-fp16 building blocks of
liblfm_hip.so as the origin-ADM UNet (``lfm_amd/models/unet.py``):

* ``Conv2d(up=True)`` with the [1,1] filter is a nearest-2x upsample followed by the 3x3 convolution -> one implicit-GEMM launch
  (mode 1); ``Conv2d(down=True)`` is a 2x2 mean (``lfm_avgpool2_f16``) followed by the 3x3 convolution (EDM.py:96-125);
* ``silu(addcmul(shift, norm1(x), scale + 1))`` (EDM.py:266-269) is the FiLM GroupNorm of ``lfm_groupnorm_f16``;
* attention: the reference's qkv channel order is [head][ch][q,k,v] (``reshape(N*heads, ch, 3, -1).unbind(2)``, EDM.py:277-281);
  the rows of ``qkv.weight`` are permuted ONCE at pack time into [head][q|k|v][ch], which is what ``lfm_attention_small_f16`` reads;
* the class embedding ``map_label(one_hot(y))`` (no bias) is a column lookup of ``map_label.weight``; the label dropped for the
  unconditional half under CFG (``drop_half_label``, EDM.py:825-826) is an extra all-zero row.

Built: ``use_context=False``, ``augment_dim=0``, channels that are multiples of 64 (the implicit-GEMM contract).  SongUNet
(``ncsn++`` / ``ddpm++``) stays out of scope.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import hip


def _weight_init(shape, mode, fan_in, fan_out):  # EDM.py:27-36
    if mode == "xavier_uniform":
        return np.sqrt(6 / (fan_in + fan_out)) * (torch.rand(*shape) * 2 - 1)
    if mode == "xavier_normal":
        return np.sqrt(2 / (fan_in + fan_out)) * torch.randn(*shape)
    if mode == "kaiming_uniform":
        return np.sqrt(3 / fan_in) * (torch.rand(*shape) * 2 - 1)
    if mode == "kaiming_normal":
        return np.sqrt(1 / fan_in) * torch.randn(*shape)
    raise ValueError(f'Invalid init mode "{mode}"')


class Linear(nn.Module):  # EDM.py:43-56 (parameter container)
    def __init__(self, in_features, out_features, bias=True, init_mode="kaiming_normal", init_weight=1, init_bias=0):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        kw = dict(mode=init_mode, fan_in=in_features, fan_out=out_features)
        self.weight = nn.Parameter(_weight_init([out_features, in_features], **kw) * init_weight)
        self.bias = nn.Parameter(_weight_init([out_features], **kw) * init_bias) if bias else None


class Conv2d(nn.Module):  # EDM.py:63-98 (parameter container)
    def __init__(self, in_channels, out_channels, kernel, bias=True, up=False, down=False, resample_filter=(1, 1),
                 init_mode="kaiming_normal", init_weight=1, init_bias=0):
        assert not (up and down)
        super().__init__()
        self.in_channels, self.out_channels, self.up, self.down, self.kernel = in_channels, out_channels, up, down, kernel
        kw = dict(mode=init_mode, fan_in=in_channels * kernel * kernel, fan_out=out_channels * kernel * kernel)
        self.weight = nn.Parameter(_weight_init([out_channels, in_channels, kernel, kernel], **kw) * init_weight) if kernel else None
        self.bias = nn.Parameter(_weight_init([out_channels], **kw) * init_bias) if kernel and bias else None
        if tuple(resample_filter) != (1, 1):
            raise NotImplementedError("only the [1,1] resample filter of DhariwalUNet is built")
        f = torch.as_tensor(resample_filter, dtype=torch.float32)
        f = f.ger(f).unsqueeze(0).unsqueeze(1) / f.sum().square()
        self.register_buffer("resample_filter", f if up or down else None)


class GroupNorm(nn.Module):  # EDM.py:139-151
    def __init__(self, num_channels, num_groups=32, min_channels_per_group=4, eps=1e-5):
        super().__init__()
        self.num_groups = min(num_groups, num_channels // min_channels_per_group)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))


class UNetBlock(nn.Module):  # EDM.py:188-292 (parameter container)
    def __init__(self, in_channels, out_channels, emb_channels, up=False, down=False, attention=False, num_heads=None,
                 channels_per_head=64, dropout=0, skip_scale=1, eps=1e-5, init=None, init_zero=None):
        super().__init__()
        init, init_zero = init or {}, init_zero or dict(init_weight=0)
        self.in_channels, self.out_channels, self.up, self.down = in_channels, out_channels, up, down
        self.num_heads = 0 if not attention else (num_heads if num_heads is not None else out_channels // channels_per_head)
        self.skip_scale = skip_scale
        if skip_scale != 1:
            raise NotImplementedError("skip_scale != 1 is not built (DhariwalUNet uses 1)")
        self.norm0 = GroupNorm(in_channels, eps=eps)
        self.conv0 = Conv2d(in_channels, out_channels, 3, up=up, down=down, **init)
        self.affine = Linear(emb_channels, out_channels * 2, **init)
        self.norm1 = GroupNorm(out_channels, eps=eps)
        self.conv1 = Conv2d(out_channels, out_channels, 3, **init_zero)
        self.skip = None
        if out_channels != in_channels or up or down:
            self.skip = Conv2d(in_channels, out_channels, 1 if out_channels != in_channels else 0, up=up, down=down, **init)
        if self.num_heads:
            self.norm2 = GroupNorm(out_channels, eps=eps)
            self.qkv = Conv2d(out_channels, out_channels * 3, 1, **init)
            self.proj = Conv2d(out_channels, out_channels, 1, **init_zero)


class DhariwalUNet(nn.Module):
    def __init__(self, img_resolution, in_channels, out_channels, label_dim=0, augment_dim=0, model_channels=192, channel_mult=(1, 2, 3, 4),
                 channel_mult_emb=4, num_blocks=3, attn_resolutions=(32, 16, 8), dropout=0.10, label_dropout=0, use_context=False):
        super().__init__()
        if use_context or augment_dim:
            raise NotImplementedError("use_context / augment_dim are not used by the sampling path and are not built")
        self.label_dim, self.label_dropout = label_dim, label_dropout
        self.img_resolution, self.in_channels, self.out_channels, self.model_channels = img_resolution, in_channels, out_channels, model_channels
        emb_channels = model_channels * channel_mult_emb
        self.emb_channels = emb_channels
        init = dict(init_mode="kaiming_uniform", init_weight=np.sqrt(1 / 3), init_bias=np.sqrt(1 / 3))
        init_zero = dict(init_mode="kaiming_uniform", init_weight=0, init_bias=0)
        bk = dict(emb_channels=emb_channels, channels_per_head=64, dropout=dropout, init=init, init_zero=init_zero)
        self.map_layer0 = Linear(model_channels, emb_channels, **init)
        self.map_layer1 = Linear(emb_channels, emb_channels, **init)
        self.map_label = Linear(label_dim, emb_channels, bias=False, init_mode="kaiming_normal", init_weight=np.sqrt(label_dim)) if label_dim else None
        self.enc = nn.ModuleDict()
        cout = in_channels
        for level, mult in enumerate(channel_mult):
            res = img_resolution >> level
            if level == 0:
                cin, cout = cout, model_channels * mult
                self.enc[f"{res}x{res}_conv"] = Conv2d(cin, cout, 3, **init)
            else:
                self.enc[f"{res}x{res}_down"] = UNetBlock(cout, cout, down=True, **bk)
            for idx in range(num_blocks):
                cin, cout = cout, model_channels * mult
                self.enc[f"{res}x{res}_block{idx}"] = UNetBlock(cin, cout, attention=(res in attn_resolutions), **bk)
        skips = [b.out_channels for b in self.enc.values()]
        self.dec = nn.ModuleDict()
        for level, mult in reversed(list(enumerate(channel_mult))):
            res = img_resolution >> level
            if level == len(channel_mult) - 1:
                self.dec[f"{res}x{res}_in0"] = UNetBlock(cout, cout, attention=True, **bk)
                self.dec[f"{res}x{res}_in1"] = UNetBlock(cout, cout, **bk)
            else:
                self.dec[f"{res}x{res}_up"] = UNetBlock(cout, cout, up=True, **bk)
            for idx in range(num_blocks + 1):
                cin = cout + skips.pop()
                cout = model_channels * mult
                self.dec[f"{res}x{res}_block{idx}"] = UNetBlock(cin, cout, attention=(res in attn_resolutions), **bk)
        self.out_norm = GroupNorm(cout)
        self.out_conv = Conv2d(cout, out_channels, 3, **init_zero)
        self._packed = None
        self._scratch = None
        self._gen = 0

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
        dev = self.out_conv.weight.device
        hip.require_gpu(self.out_conv.weight, "DhariwalUNet")
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()  # noqa: E731
        f16 = lambda t: t.detach().to(dev, torch.float16).contiguous()  # noqa: E731

        def conv3(m):
            w = m.weight
            if w.shape[1] % 64:
                raise hip.LfmHipError(f"3x3 conv with Cin={w.shape[1]}: the implicit-GEMM path needs Cin % 64 == 0")
            return f16(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)), f32(m.bias)

        def conv1(m):
            return f16(m.weight.reshape(m.weight.shape[0], -1)), f32(m.bias)

        P = {}
        for group in (self.enc, self.dec):
            pre = "enc." if group is self.enc else "dec."
            for name, b in group.items():
                if isinstance(b, Conv2d):
                    P[pre + name] = (f32(b.weight), f32(b.bias))
                    continue
                d = dict(gn0=(f32(b.norm0.weight), f32(b.norm0.bias)), c0=conv3(b.conv0), aff=(f16(b.affine.weight), f32(b.affine.bias)),
                         gn1=(f32(b.norm1.weight), f32(b.norm1.bias)), c1=conv3(b.conv1),
                         skip=conv1(b.skip) if (b.skip is not None and b.skip.weight is not None) else None)
                if b.num_heads:
                    C, ch = b.out_channels, b.out_channels // b.num_heads
                    # reference row (head, c, which) -> our row (head, which, c)
                    perm = torch.arange(3 * C).reshape(b.num_heads, ch, 3).permute(0, 2, 1).reshape(-1)
                    d.update(gn2=(f32(b.norm2.weight), f32(b.norm2.bias)),
                             qkv=(f16(b.qkv.weight.reshape(3 * C, C)[perm]), f32(b.qkv.bias[perm])), proj=conv1(b.proj))
                P[pre + name] = d
        # every block's `affine` projection of the embedding (EDM.py:263-265) in ONE GEMM per evaluation: weights stacked row-wise, a block reads its
        # [scale | shift] columns of the result (28 launches of ~16 us each at the ffhq_adm size otherwise: profiles/r04_config6_kernel_stats.csv)
        offs, ws_, bs_, off = {}, [], [], 0
        for key, d in P.items():
            if isinstance(d, dict):
                offs[key] = (off, d["aff"][0].shape[0])
                ws_.append(d["aff"][0])
                bs_.append(d["aff"][1])
                off += d["aff"][0].shape[0]
        P["aff_all"] = (torch.cat(ws_, 0).contiguous(), torch.cat(bs_, 0).contiguous(), offs)
        P["time"] = (f32(self.map_layer0.weight), f32(self.map_layer0.bias), f32(self.map_layer1.weight), f32(self.map_layer1.bias))
        if self.map_label is not None:  # [label_dim + 1, E]: column lookup + one all-zero row for the dropped label
            P["label"] = torch.cat([f32(self.map_label.weight).t(), torch.zeros(1, self.emb_channels, device=dev)], 0).contiguous()
        else:
            P["label"] = None
        P["gn_out"] = (f32(self.out_norm.weight), f32(self.out_norm.bias))
        wo = self.out_conv.weight
        if wo.shape[0] > 4:
            raise hip.LfmHipError("output conv with more than 4 channels is not built")
        w4 = torch.zeros(4, wo.shape[1], 3, 3, device=dev)
        w4[: wo.shape[0]] = wo
        b4 = torch.zeros(4, device=dev)
        b4[: wo.shape[0]] = self.out_conv.bias
        P["conv_out"] = (f16(w4.permute(0, 2, 3, 1).reshape(4, -1)), f32(b4))
        self._packed = P
        self._gen += 1
        return P

    # ---- ops --------------------------------------------------------------------------------------------------------------
    def _gn(self, x, N, HW, C, gb, film, silu, eps=1e-5):
        groups = min(32, C // 4)  # EDM GroupNorm (EDM.py:139-143)
        y = torch.empty_like(x)
        need = hip.lib().lfm_groupnorm_scratch_bytes(N, C)
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != x.device:
            self._scratch = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=x.device)
            self._gen += 1
        hip.check(hip.lib().lfm_groupnorm_f16(hip.ptr(x), hip.ptr(y), hip.ptr(gb[0]), hip.ptr(gb[1]), hip.ptr(film),
                                              film.stride(0) if film is not None else 0, hip.ptr(self._scratch), N, HW, C, groups, eps,
                                              1 if silu else 0, hip.stream_ptr(x.device)), "lfm_groupnorm_f16")
        return y

    def _conv(self, x, wb, N, H, W, Cin, Cout, mode=0, resid=None):
        out = torch.empty(N * H * W, Cout, dtype=torch.float16, device=x.device)
        L = hip.lib()
        need = L.lfm_conv3x3_workspace_bytes(N, H, W, Cin, Cout)  # split-K slabs of the small-M / huge-K low-resolution levels
        if need and (getattr(self, "_conv_ws", None) is None or self._conv_ws.numel() < need or self._conv_ws.device != x.device):
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

    def _pool(self, x, N, Ho, Wo, C):
        y = torch.empty(N * Ho * Wo, C, dtype=torch.float16, device=x.device)
        hip.check(hip.lib().lfm_avgpool2_f16(hip.ptr(x), hip.ptr(y), N, Ho, Wo, C, hip.stream_ptr(x.device)), "lfm_avgpool2_f16")
        return y

    def _gn2(self, xa, xb, N, HW, gb, silu, eps=1e-5):
        """GroupNorm of the channel concat [xa | xb] read in place (``torch.cat([x, skips.pop()], dim=1)``, EDM.py:840-842, is never materialised)."""
        Ca, Cb = xa.shape[1], xb.shape[1]
        y = torch.empty(xa.shape[0], Ca + Cb, dtype=torch.float16, device=xa.device)
        need = hip.lib().lfm_groupnorm_scratch_bytes(N, Ca + Cb)
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != xa.device:
            self._scratch = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=xa.device)
            self._gen += 1
        hip.check(hip.lib().lfm_groupnorm2_f16(hip.ptr(xa), Ca, hip.ptr(xb), Cb, hip.ptr(y), hip.ptr(gb[0]), hip.ptr(gb[1]), None, 0, hip.ptr(self._scratch),
                                               N, HW, min(32, (Ca + Cb) // 4), eps, 1 if silu else 0, hip.stream_ptr(xa.device)), "lfm_groupnorm2_f16")
        return y

    def _linear2(self, xa, xb, wb):
        M, Nout = xa.shape[0], wb[0].shape[0]
        out = torch.empty(M, Nout, dtype=torch.float16, device=xa.device)
        hip.check(hip.lib().lfm_linear2_f16(hip.ptr(xa), xa.shape[1], hip.ptr(xb), xb.shape[1], hip.ptr(wb[0]), wb[0].stride(0), hip.ptr(out), Nout, M, Nout,
                                            hip.ptr(wb[1]), None, hip.stream_ptr(xa.device)), "lfm_linear2_f16")
        return out

    def _cat(self, pair):
        h, s = pair
        cat = torch.empty(h.shape[0], h.shape[1] + s.shape[1], dtype=torch.float16, device=h.device)
        hip.check(hip.lib().lfm_concat_channels_f16(hip.ptr(h), hip.ptr(s), hip.ptr(cat), h.shape[0], h.shape[1], s.shape[1], hip.stream_ptr(h.device)),
                  "lfm_concat_channels_f16")
        return cat

    def _block(self, name, b, x, N, H, W, film_all):
        """UNetBlock.forward (EDM.py:258-292).  Returns (out, H, W).  `x` may be the pair (h, skip) of a decoder block: its channel concat is then read in place
        by the block's two consumers -- the first GroupNorm and the 1x1 skip convolution -- as in the origin-ADM UNet (round 6; the concat kernel was 2 % of an
        evaluation at the ffhq_adm size)."""
        p = self._packed[name]
        Cin, Cout = b.in_channels, b.out_channels
        pair = None
        if isinstance(x, tuple):
            if b.down or b.up or p["skip"] is None or x[0].shape[1] % 64 or x[1].shape[1] % 8:
                x = self._cat(x)  # resampled / identity-skip inputs need the tensor itself
            else:
                pair = x
        orig = x
        t = self._gn2(pair[0], pair[1], N, H * W, p["gn0"], True) if pair is not None else self._gn(x, N, H * W, Cin, p["gn0"], None, True)
        if b.down:
            H, W = H // 2, W // 2
            t = self._pool(t, N, H, W, Cin)
            orig = self._pool(orig, N, H, W, Cin)
            h = self._conv(t, p["c0"], N, H, W, Cin, Cout)
        elif b.up:
            H, W = H * 2, W * 2
            h = self._conv(t, p["c0"], N, H, W, Cin, Cout, mode=1)
        else:
            h = self._conv(t, p["c0"], N, H, W, Cin, Cout)
        fo, fw = self._packed["aff_all"][2][name]
        film = film_all[:, fo:fo + fw]  # fp32 [N, 2*Cout] = [scale | shift]: this block's columns of the one affine GEMM (row stride = all blocks' columns)
        t = self._gn(h, N, H * W, Cout, p["gn1"], film, True)
        if b.up:  # skip(orig) with kernel 0 = nearest 2x upsample of the input (conv_transpose with the all-ones 2x2 filter)
            up = torch.empty(N * H * W, Cin, dtype=torch.float16, device=x.device)
            hip.check(hip.lib().lfm_upsample2_f16(hip.ptr(orig), hip.ptr(up), N, H, W, Cin, hip.stream_ptr(x.device)), "lfm_upsample2_f16")
            orig = up
        if pair is not None:
            skip = self._linear2(pair[0], pair[1], p["skip"])
        else:
            skip = orig if p["skip"] is None else self._linear(orig, p["skip"])
        x = self._conv(t, p["c1"], N, H, W, Cout, Cout, resid=skip)
        if b.num_heads:
            T, ch = H * W, Cout // b.num_heads
            t = self._gn(x, N, T, Cout, p["gn2"], None, False)
            qkv = self._linear(t, p["qkv"])
            a = torch.empty(N * T, Cout, dtype=torch.float16, device=x.device)
            hip.check(hip.lib().lfm_attention_small_f16(hip.ptr(qkv), hip.ptr(a), N, T, b.num_heads, ch, hip.stream_ptr(x.device)),
                      "lfm_attention_small_f16")
            x = self._linear(a, p["proj"], resid=x)
        return x, H, W

    # ---- forward ----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _forward(self, noise_labels, x, y, drop_half_label):
        hip.require_gpu(x, "DhariwalUNet.forward")
        if self.training:
            raise hip.LfmHipError("the HIP DhariwalUNet is inference-only: call .eval()")
        if self._packed is None:
            self._pack()
        L, dev, P = hip.lib(), x.device, self._packed
        x = x.contiguous().float()
        N, Cin, H, W = x.shape
        t = torch.as_tensor(noise_labels, device=dev).float().reshape(-1).contiguous()
        if t.numel() not in (1, N):
            raise ValueError(f"noise_labels must have 1 or {N} elements")
        E, F = self.emb_channels, self.model_channels
        yy = None
        if P["label"] is not None and y is not None:
            yy = y.to(dev, torch.long).clone()
            if drop_half_label:
                yy[N // 2:] = self.label_dim  # the all-zero row (EDM.py:825-826)
        n_labels = 0 if P["label"] is None else int(P["label"].shape[0])
        if yy is not None:
            hip.check_labels(yy, n_labels, "DhariwalUNet")
        emb = torch.empty(N, E, device=dev)
        emb_silu = torch.empty(N, E, device=dev, dtype=torch.float16)
        h1 = torch.empty(N, E, device=dev)
        tw = P["time"]
        hip.check(L.lfm_time_embed(hip.ptr(t), t.numel(), hip.ptr(tw[0]), hip.ptr(tw[1]), hip.ptr(tw[2]), hip.ptr(tw[3]),
                                   hip.ptr(P["label"] if yy is not None else None), hip.ptr(yy), n_labels, hip.ptr(h1), hip.ptr(emb),
                                   hip.ptr(emb_silu), N, F, E, hip.stream_ptr(dev)), "lfm_time_embed")
        film_all = hip.gemm_f16(emb_silu, P["aff_all"][0], P["aff_all"][1], epilogue=2)  # fp32 [N, sum of 2*Cout over the blocks]
        skips, h = [], None
        for name, b in self.enc.items():
            if isinstance(b, Conv2d):
                wb = P["enc." + name]
                h = torch.empty(N * H * W, b.out_channels, dtype=torch.float16, device=dev)
                hip.check(L.lfm_conv3x3_in_f32(hip.ptr(x), hip.ptr(wb[0]), hip.ptr(wb[1]), hip.ptr(h), N, H, W, Cin, b.out_channels,
                                               hip.stream_ptr(dev)), "lfm_conv3x3_in_f32")
            else:
                h, H, W = self._block("enc." + name, b, h, N, H, W, film_all)
            skips.append((h, h.shape[1]))
        for name, b in self.dec.items():
            if h.shape[1] != b.in_channels:
                s, _ = skips.pop()
                h = (h, s)  # torch.cat([x, skips.pop()], dim=1) (EDM.py:840-842): consumed in place by the block where its shape allows it
            h, H, W = self._block("dec." + name, b, h, N, H, W, film_all)
        t1 = self._gn(h, N, H * W, h.shape[1], P["gn_out"], None, True)
        out = torch.empty(N, self.out_channels, H, W, device=dev)
        co = P["conv_out"]
        hip.check(L.lfm_conv3x3_out_f32(hip.ptr(t1), hip.ptr(co[0]), hip.ptr(co[1]), hip.ptr(out), N, H, W, h.shape[1], self.out_channels,
                                        hip.stream_ptr(dev)), "lfm_conv3x3_out_f32")
        return out

    def forward(self, noise_labels, x, y=None, augment_labels=None, drop_half_label=False, **kwargs):
        """v = model(t, x, y) (EDM.py:808-845)."""
        return self._forward(noise_labels, x, y, drop_half_label)

    def forward_with_cfg(self, noise_labels, x, y=None, augment_labels=None, cfg_scale=1.0, **kwargs):
        """EDM.py:847-861: x[:N/2] evaluated with labels y (first half) and with the label dropped (second half)."""
        n2 = len(x) // 2
        xin = x.contiguous().float().clone()
        xin[n2:].copy_(xin[:n2])  # combined = cat([half, half])
        out = self._forward(noise_labels, xin, y, True)
        cond, uncond = out[:n2], out[n2:]
        coef = torch.tensor([cfg_scale, 1.0 - cfg_scale], device=x.device)  # uncond + s*(cond - uncond)
        res = torch.empty_like(out)
        hip.lincomb(res[:n2], None, [cond, uncond], coef)
        hip.lincomb(res[n2:], None, [cond, uncond], coef)
        return res


def get_edm_network(config):
    """Reference models/EDM.py:864-939: only the ``adm`` branch is on the sampling path of the reference's test_args."""
    if config.model_type != "adm":
        raise NotImplementedError(f"model_type {config.model_type!r}: SongUNet (ncsn++/ddpm++) and adm_context are out of scope (SURVEY.md §2)")
    return DhariwalUNet(img_resolution=config.image_size // config.f, in_channels=config.num_in_channels, out_channels=config.num_out_channels,
                        label_dim=config.label_dim, augment_dim=0, model_channels=config.nf, channel_mult=config.ch_mult, channel_mult_emb=4,
                        num_blocks=config.num_res_blocks, attn_resolutions=config.attn_resolutions, dropout=config.dropout,
                        label_dropout=config.label_dropout)
