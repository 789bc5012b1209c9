"""Three-class stand-in for ``timm.models.vision_transformer`` (oracle only).

The reference imports ``Attention, Mlp, PatchEmbed`` from timm
(/root/reference/models/DiT.py:17; un-pinned in requirements.txt:6) and timm is
not installed in this image.  This module restates the published behaviour of
those three classes so that the UNMODIFIED reference ``models/DiT.py`` can be
imported by ``oracle/make_golden.py`` to generate golden vectors.

Call sites that constrain the behaviour:
  * models/DiT.py:120  Attention(hidden, num_heads=, qkv_bias=True)
  * models/DiT.py:124  Mlp(in_features=, hidden_features=, act_layer=, drop=0)
  * models/DiT.py:179  PatchEmbed(img, patch, in_chans, embed, bias=True)
  * models/DiT.py:182  .num_patches   :208-210 .proj   :236 .patch_size[0]

parity unpinned w.r.t. timm itself (no version, not on disk).
"""
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, **_):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        attn = attn.softmax(dim=-1)
        x = attn @ v
        x = x.transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, **_):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True, bias=True):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
        self.norm = nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


def install():
    """Register the shim as ``timm.models.vision_transformer`` in sys.modules."""
    if "timm" in sys.modules and not getattr(sys.modules["timm"], "_lfm_oracle_shim", False):
        return  # a real timm is importable; leave it alone
    timm = types.ModuleType("timm")
    timm._lfm_oracle_shim = True
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.Attention, vt.Mlp, vt.PatchEmbed = Attention, Mlp, PatchEmbed
    timm.models = models
    models.vision_transformer = vt
    sys.modules["timm"] = timm
    sys.modules["timm.models"] = models
    sys.modules["timm.models.vision_transformer"] = vt


_ = (torch, F)
