"""Oracle restatement of the origin-ADM ``UNetModel`` forward (TEST INFRASTRUCTURE ONLY).

Plain fp32 PyTorch, functional, driven by a reference-format ``state_dict`` + the constructor arguments.  Follows
/root/reference/models/guided_diffusion/unet.py:376-655 (layer plan :474-596, forward :613-655), ResBlock :218-238,
AttentionBlock/QKVAttentionLegacy :281-334, nn.py:17-19,103-121.  PINNED: checked against ``tests/golden/unet_tiny.pt``
(produced by the unmodified reference) in ``tests/test_oracle_golden.py``.
Scope: use_scale_shift_norm=True, resblock_updown=False, use_new_attention_order=False, dims=2, conv resampling.
"""
import math

import torch
import torch.nn.functional as F


def layer_plan(cfg):
    """[(block_name, [(kind, name, info)])] in execution order, mirroring the constructor loops (unet.py:474-590)."""
    mc, mult, nrb, attn = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"], set(cfg["attention_resolutions"])
    ch = int(mult[0] * mc)
    inp, chans, ds = [("input_blocks.0", [("conv_in", "input_blocks.0.0", None)])], [ch], 1
    idx = 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            layers = [("res", f"input_blocks.{idx}.0", (ch, int(m * mc)))]
            ch = int(m * mc)
            if ds in attn:
                layers.append(("attn", f"input_blocks.{idx}.1", ch))
            inp.append((f"input_blocks.{idx}", layers))
            chans.append(ch)
            idx += 1
        if level != len(mult) - 1:
            inp.append((f"input_blocks.{idx}", [("down", f"input_blocks.{idx}.0", ch)]))
            chans.append(ch)
            ds *= 2
            idx += 1
    mid = [("res", "middle_block.0", (ch, ch)), ("attn", "middle_block.1", ch), ("res", "middle_block.2", (ch, ch))]
    out, idx = [], 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [("res", f"output_blocks.{idx}.0", (ch + ich, int(mc * m)))]
            ch = int(mc * m)
            j = 1
            if ds in attn:
                layers.append(("attn", f"output_blocks.{idx}.{j}", ch))
                j += 1
            if level and i == nrb:
                layers.append(("up", f"output_blocks.{idx}.{j}", ch))
                ds //= 2
            out.append((f"output_blocks.{idx}", layers))
            idx += 1
    return inp, mid, out


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(sd, p, x):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def _res(sd, p, x, emb):
    h = F.conv2d(F.silu(_gn(sd, p + ".in_layers.0", x)), sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[..., None, None]
    scale, shift = torch.chunk(e, 2, dim=1)
    h = _gn(sd, p + ".out_layers.0", h) * (1 + scale) + shift
    h = F.conv2d(F.silu(h), sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def _attn(sd, p, x, heads):
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(sd, p + ".norm", xf), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    ch = c // heads
    q, k, v = qkv.reshape(b * heads, ch * 3, -1).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s).float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, xf.shape[-1])
    h = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + h).reshape(b, c, hh, ww)


def _heads(cfg, ch, upsample=False):
    if cfg.get("num_head_channels", -1) != -1:
        return ch // cfg["num_head_channels"]
    nh = cfg["num_heads"]
    if upsample and cfg.get("num_heads_upsample", -1) != -1:
        nh = cfg["num_heads_upsample"]
    return nh


def _run(sd, cfg, layers, h, emb, upsample=False):
    for kind, name, info in layers:
        if kind == "conv_in":
            h = F.conv2d(h, sd[name + ".weight"], sd[name + ".bias"], padding=1)
        elif kind == "res":
            h = _res(sd, name, h, emb)
        elif kind == "attn":
            h = _attn(sd, name, h, _heads(cfg, info, upsample))
        elif kind == "down":
            h = F.conv2d(h, sd[name + ".op.weight"], sd[name + ".op.bias"], stride=2, padding=1)
        elif kind == "up":
            h = F.conv2d(F.interpolate(h, scale_factor=2, mode="nearest"), sd[name + ".conv.weight"], sd[name + ".conv.bias"], padding=1)
    return h


@torch.no_grad()
def unet_forward(sd, cfg, t, x, y=None):
    """unet.py:613-655.  t: [N] (a scalar / [1] is broadcast, as :629-630 intends)."""
    sd = {k: v.float() for k, v in sd.items()}
    t = t.reshape(-1).float()
    if t.numel() != x.shape[0]:
        t = t * torch.ones(x.shape[0])
    emb = F.linear(timestep_embedding(t, cfg["model_channels"]), sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if cfg.get("num_classes") is not None:
        emb = emb + sd["label_emb.weight"][y]
    inp, mid, out = layer_plan(cfg)
    h, hs = x.float(), []
    for _, layers in inp:
        h = _run(sd, cfg, layers, h, emb)
        hs.append(h)
    h = _run(sd, cfg, mid, h, emb)
    for _, layers in out:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run(sd, cfg, layers, h, emb, upsample=True)
    return F.conv2d(F.silu(_gn(sd, "out.0", h)), sd["out.2.weight"], sd["out.2.bias"], padding=1)


def make_unet_state(cfg, seed=0):
    """Seeded random weights with the reference's names/shapes (zero-init tensors de-zeroed: SURVEY.md fact 3)."""
    g = torch.Generator().manual_seed(seed)
    mc, ted = cfg["model_channels"], cfg["model_channels"] * 4
    sd = {}

    def lin(name, o, i, std=None):
        a = std if std is not None else math.sqrt(1.0 / i)
        sd[name + ".weight"] = torch.randn(o, i, generator=g) * a
        sd[name + ".bias"] = torch.randn(o, generator=g) * 0.02

    def conv(name, o, i, k, gain=1.0):
        shape = (o, i, k, k) if k else (o, i, 1)
        kk = k * k if k else 1
        sd[name + ".weight"] = torch.randn(shape, generator=g) * (gain * math.sqrt(1.0 / (i * kk)))
        sd[name + ".bias"] = torch.randn(o, generator=g) * 0.02

    def gn(name, c):
        sd[name + ".weight"] = 1 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)

    lin("time_embed.0", ted, mc)
    lin("time_embed.2", ted, ted)
    if cfg.get("num_classes") is not None:
        sd["label_emb.weight"] = torch.randn(cfg["num_classes"], ted, generator=g)
    inp, mid, out = layer_plan(cfg)
    for _, layers in inp + [("mid", mid)] + out:
        for kind, name, info in layers:
            if kind == "conv_in":
                conv(name, int(cfg["channel_mult"][0] * mc), cfg["in_channels"], 3)
            elif kind == "res":
                cin, cout = info
                gn(name + ".in_layers.0", cin)
                conv(name + ".in_layers.2", cout, cin, 3)
                lin(name + ".emb_layers.1", 2 * cout, ted, std=0.02)
                gn(name + ".out_layers.0", cout)
                conv(name + ".out_layers.3", cout, cout, 3, gain=0.5)
                if cin != cout:
                    conv(name + ".skip_connection", cout, cin, 1)
            elif kind == "attn":
                gn(name + ".norm", info)
                conv(name + ".qkv", 3 * info, info, 0)
                conv(name + ".proj_out", info, info, 0, gain=0.5)
            elif kind == "down":
                conv(name + ".op", info, info, 3)
            elif kind == "up":
                conv(name + ".conv", info, info, 3)
    ch0 = int(cfg["channel_mult"][0] * mc)
    gn("out.0", ch0)
    conv("out.2", cfg["out_channels"], ch0, 3)
    return sd
