"""Oracle restatement of the DiT velocity field (TEST INFRASTRUCTURE ONLY).

Plain fp32 PyTorch, functional, driven by a reference-format ``state_dict``.
Every function cites the reference lines it follows (paths relative to
/root/reference).  PINNED: ``tests/test_oracle_golden.py`` checks this file
against ``tests/golden/dit_tiny.pt``, which ``oracle/make_golden.py`` produced
from the unmodified reference ``models/DiT.py``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# models/DiT.py:354-415 -- name -> (depth, hidden, patch, heads)
DIT_CONFIGS = {
    "DiT-XL/2": (28, 1152, 2, 16), "DiT-XL/4": (28, 1152, 4, 16), "DiT-XL/8": (28, 1152, 8, 16),
    "DiT-L/2": (24, 1024, 2, 16), "DiT-L/4": (24, 1024, 4, 16), "DiT-L/8": (24, 1024, 8, 16),
    "DiT-B/2": (12, 768, 2, 12), "DiT-B/4": (12, 768, 4, 12), "DiT-B/8": (12, 768, 8, 12),
    "DiT-S/2": (12, 384, 2, 6), "DiT-S/4": (12, 384, 4, 6), "DiT-S/8": (12, 384, 8, 6),
}


class DiTCfg:
    """Shape description of one DiT (models/DiT.py:157-184)."""

    def __init__(self, depth, hidden, patch, heads, img_resolution=32, in_channels=4,
                 num_classes=1000, label_dropout=0.1, mlp_ratio=4.0):
        self.depth, self.hidden, self.patch, self.heads = depth, hidden, patch, heads
        self.res, self.in_ch = img_resolution, in_channels
        self.num_classes, self.label_dropout = num_classes, label_dropout
        self.mlp_hidden = int(hidden * mlp_ratio)
        self.grid = img_resolution // patch
        self.tokens = self.grid * self.grid
        # LabelEmbedder rows: num_classes + (dropout > 0)  (models/DiT.py:79-81)
        self.label_rows = num_classes + (1 if label_dropout > 0 else 0)

    @staticmethod
    def named(name, **kw):
        d, h, p, nh = DIT_CONFIGS[name]
        return DiTCfg(d, h, p, nh, **kw)


# ----------------------------------------------------------------------------- pos-embed
def sincos_pos_embed_2d(embed_dim, grid_size):
    """models/DiT.py:299-346.  [emb_h | emb_w], each [sin | cos]; meshgrid with w first,
    so ``grid[0]`` (fed to the first half) is the COLUMN index of the token."""
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, -1)

    def one(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float64) / (dim / 2.0)
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.astype(np.float64), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one(embed_dim // 2, grid[0]), one(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).float().unsqueeze(0)


# ----------------------------------------------------------------------------- pieces
def timestep_embedding(t, dim=256, max_period=10000):
    """models/DiT.py:43-62 -- cos first, then sin; t is NOT scaled by 1000."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def t_embedder(sd, t):
    """models/DiT.py:64-69 (0-d t is promoted to [1])."""
    if t.dim() == 0:
        t = t[None]
    h = F.linear(timestep_embedding(t), sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])
    return F.linear(F.silu(h), sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])


def modulate(x, shift, scale):
    """models/DiT.py:20-21."""
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def attention(sd, pre, x, heads):
    """timm Attention as called at models/DiT.py:120 (restated; see oracle/timm_shim.py)."""
    B, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def mlp(sd, pre, x):
    """timm Mlp with GELU(tanh) (models/DiT.py:123-124)."""
    h = F.gelu(F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"]), approximate="tanh")
    return F.linear(h, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def dit_block(sd, i, x, c, heads):
    """models/DiT.py:127-131."""
    p = f"blocks.{i}."
    mod = F.linear(F.silu(c), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"])
    s_msa, sc_msa, g_msa, s_mlp, sc_mlp, g_mlp = mod.chunk(6, dim=1)
    D = x.shape[-1]
    x = x + g_msa.unsqueeze(1) * attention(sd, p + "attn.", modulate(F.layer_norm(x, (D,), eps=1e-6), s_msa, sc_msa), heads)
    x = x + g_mlp.unsqueeze(1) * mlp(sd, p + "mlp.", modulate(F.layer_norm(x, (D,), eps=1e-6), s_mlp, sc_mlp))
    return x


def unpatchify(x, cfg):
    """models/DiT.py:230-243."""
    c, p, h = cfg.in_ch, cfg.patch, cfg.grid
    x = x.reshape(x.shape[0], h, h, p, p, c)
    x = torch.einsum("nhwpqc->nchpwq", x)
    return x.reshape(x.shape[0], c, h * p, h * p)


# ----------------------------------------------------------------------------- forward
@torch.no_grad()
def dit_forward(sd, cfg, t, x, y=None):
    """models/DiT.py:252-272.  t: 0-d / [1] / [N]; x: [N,C,H,W] fp32; y: [N] int64 or None."""
    N = x.shape[0]
    if y is None:  # :259-260 -- "in_channels" here is the label-table row count
        y = torch.full((N,), cfg.label_rows - 1, dtype=torch.long)
    tok = F.conv2d(x, sd["x_embedder.proj.weight"], sd["x_embedder.proj.bias"], stride=cfg.patch)
    tok = tok.flatten(2).transpose(1, 2) + sd["pos_embed"]
    c = t_embedder(sd, t) + sd["y_embedder.embedding_table.weight"][y]
    for i in range(cfg.depth):
        tok = dit_block(sd, i, tok, c, cfg.heads)
    D = cfg.hidden
    mod = F.linear(F.silu(c), sd["final_layer.adaLN_modulation.1.weight"], sd["final_layer.adaLN_modulation.1.bias"])
    shift, scale = mod.chunk(2, dim=1)
    tok = modulate(F.layer_norm(tok, (D,), eps=1e-6), shift, scale)
    tok = F.linear(tok, sd["final_layer.linear.weight"], sd["final_layer.linear.bias"])
    return unpatchify(tok, cfg)


@torch.no_grad()
def dit_forward_with_cfg(sd, cfg, t, x, y, cfg_scale):
    """models/DiT.py:274-290 -- both output halves carry the guided velocity."""
    half = x[: len(x) // 2]
    out = dit_forward(sd, cfg, t, torch.cat([half, half], 0), y)
    cond, uncond = torch.split(out, len(out) // 2, dim=0)
    g = uncond + cfg_scale * (cond - uncond)
    return torch.cat([g, g], 0)


# ----------------------------------------------------------------------------- weights
ZERO_INIT_KEYS = ("adaLN_modulation.1.", "final_layer.linear.")


def make_dit_state(cfg, seed=0, dezero_std=0.02):
    """Seeded random weights with the reference's tensor names/shapes.

    Statistics follow models/DiT.py:193-228 (xavier-uniform Linear weights, zero biases,
    N(0,0.02) embedders) EXCEPT that the tensors the reference zero-initialises (adaLN and the
    final linear, :219-228) and all biases are drawn N(0, dezero_std): with the default init the
    model outputs exactly 0 and a parity test would compare 0 with 0.
    """
    g = torch.Generator().manual_seed(seed)
    D, H, p, C = cfg.hidden, cfg.mlp_hidden, cfg.patch, cfg.in_ch

    def xavier(o, i, *rest):
        fan_in, fan_out = i * int(np.prod(rest or (1,))), o
        a = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand((o, i) + tuple(rest), generator=g) * 2 - 1) * a

    def normal(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    sd = {"pos_embed": sincos_pos_embed_2d(D, cfg.grid)}
    sd["x_embedder.proj.weight"] = xavier(D, C, p, p)
    sd["x_embedder.proj.bias"] = normal(D, std=dezero_std)
    sd["t_embedder.mlp.0.weight"] = normal(D, 256)
    sd["t_embedder.mlp.0.bias"] = normal(D, std=dezero_std)
    sd["t_embedder.mlp.2.weight"] = normal(D, D)
    sd["t_embedder.mlp.2.bias"] = normal(D, std=dezero_std)
    sd["y_embedder.embedding_table.weight"] = normal(cfg.label_rows, D)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        sd[b + "attn.qkv.weight"] = xavier(3 * D, D)
        sd[b + "attn.qkv.bias"] = normal(3 * D, std=dezero_std)
        sd[b + "attn.proj.weight"] = xavier(D, D)
        sd[b + "attn.proj.bias"] = normal(D, std=dezero_std)
        sd[b + "mlp.fc1.weight"] = xavier(H, D)
        sd[b + "mlp.fc1.bias"] = normal(H, std=dezero_std)
        sd[b + "mlp.fc2.weight"] = xavier(D, H)
        sd[b + "mlp.fc2.bias"] = normal(D, std=dezero_std)
        sd[b + "adaLN_modulation.1.weight"] = normal(6 * D, D, std=dezero_std)
        sd[b + "adaLN_modulation.1.bias"] = normal(6 * D, std=dezero_std)
    sd["final_layer.linear.weight"] = normal(p * p * C, D, std=dezero_std)
    sd["final_layer.linear.bias"] = normal(p * p * C, std=dezero_std)
    sd["final_layer.adaLN_modulation.1.weight"] = normal(2 * D, D, std=dezero_std)
    sd["final_layer.adaLN_modulation.1.bias"] = normal(2 * D, std=dezero_std)
    return sd


def dezero_(sd, seed=1234, std=0.02):
    """Re-draw every all-zero floating tensor of a state dict from N(0,std) (SURVEY.md §8c rule 1)."""
    g = torch.Generator().manual_seed(seed)
    n = 0
    for k in sorted(sd):
        v = sd[k]
        if v.is_floating_point() and v.numel() > 0 and not bool(v.any()):
            v.copy_(torch.randn(v.shape, generator=g) * std)
            n += 1
    return n


def dit_flops_per_image(cfg):
    """F = 2*[depth*(12*T*D^2 + 2*T^2*D + 6*D^2) + 2*T*16*D... ]"""
    T, D, L = cfg.tokens, cfg.hidden, cfg.depth
    pp = cfg.patch * cfg.patch * cfg.in_ch
    mac = L * (12 * T * D * D + 2 * T * T * D + 6 * D * D) + T * pp * D * 2 + 2 * D * D + 256 * D + D * D
    return 2 * mac
