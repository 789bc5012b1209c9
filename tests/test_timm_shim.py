"""Independent anchors for the three-class timm stand-in (oracle/timm_shim.py).  CPU only.

timm is not installable here, so the shim stays **parity unpinned** against timm itself.  What CAN be checked is that its three classes
compute what timm's published ``vision_transformer.Attention / Mlp / PatchEmbed`` are documented to compute, against implementations that share
no code with this repository:

* ``Attention`` == ``torch.nn.MultiheadAttention`` (PyTorch's own multi-head attention) loaded with the same parameters: the fused
  ``qkv`` weight is MHA's ``in_proj_weight`` ([q; k; v] blocks, heads as contiguous slices of the embedding), ``proj`` its ``out_proj``,
  scaling ``head_dim ** -0.5`` on the scores -- and == ``F.scaled_dot_product_attention`` on the split heads;
* ``PatchEmbed`` == ``F.unfold`` (non-overlapping p x p patches, channel-major inside a patch, row-major over the grid) times the
  flattened convolution weight;
* ``Mlp`` == fc2(act(fc1(x))) with the activation the reference passes (GELU tanh, models/DiT.py:123-124), dropouts inert in eval;
* the attributes the reference reads off them (models/DiT.py:182, 208-210, 236) exist with the documented types.
The product's head split (csrc EpiQKV + attention kernels) is tested against the same layout on the GPU (tests/test_gpu_dit.py)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import timm_shim


@pytest.mark.parametrize("dim,heads,tokens", [(64, 4, 16), (144, 2, 9), (96, 6, 33)])  # head sizes 16, 72 (the DiT-XL size), 16
def test_attention_equals_torch_multihead_attention(dim, heads, tokens):
    g = torch.Generator().manual_seed(dim + heads)
    att = timm_shim.Attention(dim, num_heads=heads, qkv_bias=True).eval()
    with torch.no_grad():
        for p in att.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    mha = nn.MultiheadAttention(dim, heads, bias=True, batch_first=True).eval()
    with torch.no_grad():
        mha.in_proj_weight.copy_(att.qkv.weight)
        mha.in_proj_bias.copy_(att.qkv.bias)
        mha.out_proj.weight.copy_(att.proj.weight)
        mha.out_proj.bias.copy_(att.proj.bias)
    x = torch.randn(3, tokens, dim, generator=g)
    with torch.no_grad():
        want, _ = mha(x, x, x, need_weights=False)
        got = att(x)
        q, k, v = att.qkv(x).reshape(3, tokens, 3, heads, dim // heads).permute(2, 0, 3, 1, 4).unbind(0)
        sdpa = att.proj(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(3, tokens, dim))
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)  # fp32 summation orders differ (values are O(1..10) at 0.2-scale weights)
    torch.testing.assert_close(got, sdpa, rtol=1e-4, atol=1e-4)
    assert att.scale == (dim // heads) ** -0.5


@pytest.mark.parametrize("p,c,res,dim", [(2, 4, 32, 48), (4, 4, 32, 40), (8, 3, 16, 24)])
def test_patch_embed_equals_unfold(p, c, res, dim):
    g = torch.Generator().manual_seed(p * res)
    pe = timm_shim.PatchEmbed(res, p, c, dim, bias=True).eval()
    with torch.no_grad():
        pe.proj.weight.copy_(torch.randn(pe.proj.weight.shape, generator=g))
        pe.proj.bias.copy_(torch.randn(pe.proj.bias.shape, generator=g))
    x = torch.randn(2, c, res, res, generator=g)
    with torch.no_grad():
        got = pe(x)
        patches = F.unfold(x, kernel_size=p, stride=p).transpose(1, 2)  # [N, grid*grid, c*p*p]: token order row-major, (c, py, px) inside
        want = patches @ pe.proj.weight.reshape(dim, -1).t() + pe.proj.bias
    assert got.shape == (2, (res // p) ** 2, dim)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    assert pe.num_patches == (res // p) ** 2 and pe.patch_size == (p, p) and isinstance(pe.proj, nn.Conv2d)  # models/DiT.py:182,208-210,236


def test_mlp_is_fc2_act_fc1_with_the_reference_activation():
    g = torch.Generator().manual_seed(1)
    mlp = timm_shim.Mlp(in_features=32, hidden_features=128, act_layer=lambda: nn.GELU(approximate="tanh"), drop=0).eval()  # models/DiT.py:123-124
    x = torch.randn(5, 7, 32, generator=g)
    with torch.no_grad():
        want = F.linear(F.gelu(F.linear(x, mlp.fc1.weight, mlp.fc1.bias), approximate="tanh"), mlp.fc2.weight, mlp.fc2.bias)
        torch.testing.assert_close(mlp(x), want, rtol=1e-6, atol=1e-6)
    assert [n for n, _ in mlp.named_children()] == ["fc1", "act", "drop1", "norm", "fc2", "drop2"]  # attribute names of the published class
    assert sorted(k for k, _ in mlp.named_parameters()) == ["fc1.bias", "fc1.weight", "fc2.bias", "fc2.weight"]


def test_attention_and_patch_embed_equal_the_transformers_vit_port():
    """A third implementation: the ViT of the ``transformers`` package (its own port of the timm / google-research ViT): separate q / k / v
    projections = the three row blocks of timm's fused ``qkv``, heads as contiguous slices, ``head_dim ** -0.5``; patch embedding = strided
    convolution, ``flatten(2).transpose(1, 2)``."""
    vit = pytest.importorskip("transformers.models.vit.modeling_vit")
    cfgmod = pytest.importorskip("transformers.models.vit.configuration_vit")
    dim, heads, tokens = 144, 2, 16  # head size 72
    cfg = cfgmod.ViTConfig(hidden_size=dim, num_attention_heads=heads, qkv_bias=True, attention_probs_dropout_prob=0.0, image_size=8, patch_size=2,
                           num_channels=4)
    cfg._attn_implementation = "eager"
    g = torch.Generator().manual_seed(9)
    att = timm_shim.Attention(dim, num_heads=heads, qkv_bias=True).eval()
    with torch.no_grad():
        for p in att.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    theirs = vit.ViTAttention(cfg).eval()
    with torch.no_grad():
        for i, proj in enumerate((theirs.q_proj, theirs.k_proj, theirs.v_proj)):
            proj.weight.copy_(att.qkv.weight[i * dim:(i + 1) * dim])
            proj.bias.copy_(att.qkv.bias[i * dim:(i + 1) * dim])
        theirs.o_proj.weight.copy_(att.proj.weight)
        theirs.o_proj.bias.copy_(att.proj.bias)
    x = torch.randn(2, tokens, dim, generator=g)
    with torch.no_grad():
        torch.testing.assert_close(att(x), theirs(x)[0], rtol=1e-4, atol=1e-4)
    pe = timm_shim.PatchEmbed(8, 2, 4, dim, bias=True).eval()
    tpe = vit.ViTPatchEmbeddings(cfg).eval()
    with torch.no_grad():
        tpe.projection.weight.copy_(pe.proj.weight)
        tpe.projection.bias.copy_(pe.proj.bias)
        img = torch.randn(3, 4, 8, 8, generator=g)
        torch.testing.assert_close(pe(img), tpe(img), rtol=1e-6, atol=1e-6)
