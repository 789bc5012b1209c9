"""DiT velocity field, MI355X-native (drop-in for /root/reference/models/DiT.py).

Same constructor, same parameter names/shapes (reference checkpoints load with strict=True), same call
contract ``model(t, x, y=None) -> v`` and ``model.forward_with_cfg(t, x, y, cfg_scale)``; the arithmetic
runs in liblfm_hip.so (lfm_dit_forward): fp16 MFMA GEMMs with fp32 accumulate, fp32 residual stream,
fp32 timestep/label path.  There is no PyTorch fallback: a forward on a non-GPU tensor raises.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from .. import hip


# --------------------------------------------------------------------------- parameter containers
class _Attention(nn.Module):  # names of timm Attention's parameters (reference DiT.py:120)
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):  # names of timm Mlp's parameters (reference DiT.py:124)
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _PatchEmbed(nn.Module):  # timm PatchEmbed (reference DiT.py:179)
    def __init__(self, img, patch, in_chans, dim):
        super().__init__()
        self.patch_size = (patch, patch)
        self.num_patches = (img // patch) ** 2
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch, bias=True)


class TimestepEmbedder(nn.Module):  # reference DiT.py:29-69
    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size), nn.SiLU(), nn.Linear(hidden_size, hidden_size))
        self.frequency_embedding_size = frequency_embedding_size


class LabelEmbedder(nn.Module):  # reference DiT.py:72-104
    def __init__(self, num_classes, hidden_size, dropout_prob):
        super().__init__()
        self.in_channels = num_classes + (dropout_prob > 0)
        self.embedding_table = nn.Embedding(self.in_channels, hidden_size)
        self.num_classes = num_classes
        self.dropout_prob = dropout_prob

    def get_in_channels(self):
        return self.in_channels


class DiTBlock(nn.Module):  # reference DiT.py:112-131
    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.attn = _Attention(hidden_size, num_heads)
        self.mlp = _Mlp(hidden_size, int(hidden_size * mlp_ratio))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))


class FinalLayer(nn.Module):  # reference DiT.py:134-149
    def __init__(self, hidden_size, patch_size, out_channels):
        super().__init__()
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))


def get_2d_sincos_pos_embed(embed_dim, grid_size):
    """Fixed 2-D sin-cos table (reference DiT.py:299-346): [emb(col) | emb(row)], each [sin | cos]."""
    pos = np.arange(grid_size, dtype=np.float64)
    col, row = np.meshgrid(pos, pos)  # col varies fastest over tokens

    def one(dim, p):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = p.reshape(-1)[:, None] * omega[None]
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    return np.concatenate([one(embed_dim // 2, col), one(embed_dim // 2, row)], axis=1)


# --------------------------------------------------------------------------- the model
class DiT(nn.Module):
    def __init__(self, img_resolution=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4.0, label_dropout=0.1, num_classes=1000, learn_sigma=False):
        super().__init__()
        if learn_sigma:
            raise NotImplementedError("learn_sigma=True is never used by the LFM sampling path")
        # what the HIP kernels are built for (csrc/dit.hip::check_shape) -- refuse at construction, not at the first forward
        tokens = (img_resolution // patch_size) ** 2 if patch_size and img_resolution % patch_size == 0 else 0
        kk = patch_size * patch_size * in_channels
        if hidden_size % num_heads or hidden_size // num_heads not in (64, 72):
            raise NotImplementedError(f"DiT with head_dim {hidden_size / num_heads:g}: the LDS-resident attention kernel is built for head_dim 64 "
                                      "(DiT-S / B / L) and 72 (DiT-XL: 1152 / 16)")
        if tokens not in (16, 64, 128, 256, 1024) or (kk > 16 and kk % 64) or kk > 256 or hidden_size % 64 or hidden_size > 1280:
            raise NotImplementedError(f"DiT shape not built: {tokens} tokens (need 16 / 64 / 128 / 256 / 1024), patch inputs {kk} (need <= 16 or a multiple of 64 up "
                                      f"to 256), hidden {hidden_size} (multiple of 64, <= 1280)")
        self.learn_sigma = learn_sigma
        self.in_channels = in_channels
        self.out_channels = in_channels
        self.patch_size = patch_size
        self.num_heads = num_heads
        self.num_classes = num_classes
        self.img_resolution = img_resolution
        self.hidden_size = hidden_size
        self.depth = depth
        self.mlp_hidden = int(hidden_size * mlp_ratio)

        self.x_embedder = _PatchEmbed(img_resolution, patch_size, in_channels, hidden_size)
        self.t_embedder = TimestepEmbedder(hidden_size)
        self.y_embedder = LabelEmbedder(num_classes, hidden_size, label_dropout)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.x_embedder.num_patches, hidden_size), requires_grad=False)
        self.blocks = nn.ModuleList([DiTBlock(hidden_size, num_heads, mlp_ratio=mlp_ratio) for _ in range(depth)])
        self.final_layer = FinalLayer(hidden_size, patch_size, self.out_channels)
        self.initialize_weights()
        self._packed = None
        self._ws = None
        self._gen = 0  # bumped whenever device buffers a captured graph may point to are replaced

    # ---- init: same distributions as reference DiT.py:193-228 (incl. the adaLN-Zero / zero output layer)
    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)
        grid = int(self.x_embedder.num_patches ** 0.5)
        self.pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(self.hidden_size, grid)).float().unsqueeze(0))
        w = self.x_embedder.proj.weight.data
        nn.init.xavier_uniform_(w.view(w.shape[0], -1))
        nn.init.zeros_(self.x_embedder.proj.bias)
        nn.init.normal_(self.y_embedder.embedding_table.weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        for blk in self.blocks:
            nn.init.zeros_(blk.adaLN_modulation[-1].weight)
            nn.init.zeros_(blk.adaLN_modulation[-1].bias)
        nn.init.zeros_(self.final_layer.adaLN_modulation[-1].weight)
        nn.init.zeros_(self.final_layer.adaLN_modulation[-1].bias)
        nn.init.zeros_(self.final_layer.linear.weight)
        nn.init.zeros_(self.final_layer.linear.bias)

    # ---- any parameter movement / reload invalidates the packed fp16 operands
    def _apply(self, fn, *a, **k):
        # invalidate the packed operands / workspace / captured graphs only if a parameter really moved or changed dtype
        # (NFECount(model).to(device) on an already-placed model must not re-pack 0.9 GB of weights per call)
        before = [(p.data_ptr(), p.dtype, p.device) for p in self.parameters()]
        out = super()._apply(fn, *a, **k)
        if before != [(p.data_ptr(), p.dtype, p.device) for p in self.parameters()]:
            self._packed = None
            self._gen = getattr(self, "_gen", 0) + 1
            self._ws = None
        return out

    def load_state_dict(self, *a, **k):
        self._packed = None
        self._gen = getattr(self, "_gen", 0) + 1
        return super().load_state_dict(*a, **k)

    def shape_struct(self):
        return hip.DitShape(self.depth, self.hidden_size, self.num_heads, self.patch_size, self.in_channels, self.img_resolution,
                            self.mlp_hidden, self.y_embedder.get_in_channels())

    @torch.no_grad()
    def _pack(self):
        """Build the lfm_dit_weights operand set on the parameters' device (fp16 GEMM operands, fp32 rest)."""
        dev = self.pos_embed.device
        hip.require_gpu(self.pos_embed, "DiT")
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        f16 = lambda t: t.detach().to(dev, torch.float16).contiguous()
        B = self.blocks
        keep = {
            "pos_embed": f32(self.pos_embed[0]),
            "patch_w": f32(self.x_embedder.proj.weight.reshape(self.hidden_size, -1)),
            "patch_b": f32(self.x_embedder.proj.bias),
            "t_w0": f32(self.t_embedder.mlp[0].weight), "t_b0": f32(self.t_embedder.mlp[0].bias),
            "t_w2": f32(self.t_embedder.mlp[2].weight), "t_b2": f32(self.t_embedder.mlp[2].bias),
            "y_table": f32(self.y_embedder.embedding_table.weight),
            "ada_w": f16(torch.cat([b.adaLN_modulation[1].weight for b in B] + [self.final_layer.adaLN_modulation[1].weight], 0)),
            "ada_b": f32(torch.cat([b.adaLN_modulation[1].bias for b in B] + [self.final_layer.adaLN_modulation[1].bias], 0)),
            "qkv_w": f16(torch.stack([b.attn.qkv.weight for b in B])), "qkv_b": f32(torch.stack([b.attn.qkv.bias for b in B])),
            "proj_w": f16(torch.stack([b.attn.proj.weight for b in B])), "proj_b": f32(torch.stack([b.attn.proj.bias for b in B])),
            "fc1_w": f16(torch.stack([b.mlp.fc1.weight for b in B])), "fc1_b": f32(torch.stack([b.mlp.fc1.bias for b in B])),
            "fc2_w": f16(torch.stack([b.mlp.fc2.weight for b in B])), "fc2_b": f32(torch.stack([b.mlp.fc2.bias for b in B])),
            "final_w": f32(self.final_layer.linear.weight), "final_b": f32(self.final_layer.linear.bias),
            "patch_w16": f16(self.x_embedder.proj.weight.reshape(self.hidden_size, -1)),  # GEMM operand of the patch embedding (patch 4 / 8)
        }
        w = hip.DitWeights(**{k: v.data_ptr() for k, v in keep.items()})
        self._packed = (w, keep, self.shape_struct())
        self._gen += 1
        return self._packed

    def _workspace(self, batch, device):
        if self._ws is None or self._ws[0] < batch or self._ws[1].device != device:
            shape = self.shape_struct()
            nbytes = hip.lib().lfm_dit_workspace_bytes(C.byref(shape), batch)
            if nbytes == 0:
                raise hip.LfmHipError(f"DiT shape not supported by the HIP path: {self.extra_repr()}")
            self._ws = (batch, torch.empty(nbytes, dtype=torch.uint8, device=device))
            self._gen += 1
        return self._ws[1]

    def extra_repr(self):
        return (f"depth={self.depth}, hidden={self.hidden_size}, heads={self.num_heads}, patch={self.patch_size}, "
                f"res={self.img_resolution}, in_ch={self.in_channels}")

    # ---- per-grid conditioning tables (unconditional models under a fixed-grid solver)
    @torch.no_grad()
    def cond_table(self, ts, batch):
        """Per-grid conditioning table (include/lfm_hip.h: lfm_dit_cond_table_build) for the grid times `ts` (device fp32 [n]): what every evaluation of
        an UNCONDITIONAL model at scalar time derives from t alone, one row per time.  Returns a device byte tensor to pass to _run(cond=...)."""
        hip.require_gpu(ts, "DiT.cond_table")
        w, _, shape = self._packed or self._pack()
        ts = ts.detach().to(torch.float32).contiguous()
        n = ts.numel()
        nbytes = hip.lib().lfm_dit_cond_table_bytes(C.byref(shape), n)
        if nbytes == 0:
            raise hip.LfmHipError("lfm_dit_cond_table_bytes refused the shape")
        table = torch.empty(nbytes, dtype=torch.uint8, device=ts.device)
        ws = self._workspace(batch, ts.device)
        rc = hip.lib().lfm_dit_cond_table_build(C.byref(shape), C.byref(w), hip.ptr(ws), ws.numel(), batch, hip.ptr(ts), n, hip.ptr(table), nbytes,
                                                hip.stream_ptr(ts.device))
        hip.check(rc, "lfm_dit_cond_table_build")
        return table

    # ---- the single entry to the device code
    @torch.no_grad()
    def _run(self, t, x, y, cfg, cfg_scale, out=None, axpy_base=None, axpy_dt=None, cond=None, fold_ln=0, gemm_select=0):
        hip.require_gpu(x, "DiT.forward")
        if self.training:
            raise hip.LfmHipError("the HIP DiT is inference-only: call .eval() (label dropout / autograd are training features)")
        if x.dim() != 4 or x.shape[1] != self.in_channels or x.shape[2] != self.img_resolution or x.shape[3] != self.img_resolution:
            raise ValueError(f"x must be [N,{self.in_channels},{self.img_resolution},{self.img_resolution}], got {tuple(x.shape)}")
        packed = self._packed or self._pack()
        w, _, shape = packed
        N = x.shape[0]
        x = x.contiguous().float()
        t = torch.as_tensor(t, device=x.device).float().reshape(-1).contiguous()
        if t.numel() not in (1, N):
            raise ValueError(f"t must have 1 or {N} elements, got {t.numel()}")
        if cond is not None and (y is not None or t.numel() != 1):
            raise ValueError("a per-grid conditioning table serves evaluations with ONE shared conditioning row (scalar t, no labels)")
        if y is not None:
            y = y.to(device=x.device, dtype=torch.long).contiguous()
            if y.numel() != N:
                raise ValueError(f"y must have {N} elements")
            hip.check_labels(y, self.y_embedder.get_in_channels(), "DiT")  # nn.Embedding's IndexError (reference DiT.py:99-103)
        if out is None:
            out = torch.empty_like(x)
        ws = self._workspace(N, x.device)
        call = hip.DitCall(N, x.data_ptr(), t.data_ptr(), t.numel(), y.data_ptr() if y is not None else None, 1 if cfg else 0,
                           float(cfg_scale), out.data_ptr(), axpy_base.data_ptr() if axpy_base is not None else None,
                           axpy_dt.data_ptr() if axpy_dt is not None else None,
                           cond[0].data_ptr() if cond is not None else None, cond[1].data_ptr() if cond is not None else None,
                           int(cond[2]) if cond is not None else 0,  # cond = (table, device step counter, row offset, rows): see cond_table()
                           int(cond[3]) if cond is not None and len(cond) > 3 else 0,
                           int(fold_ln), int(gemm_select))  # per-call settings (ABI 4): 0 = the library defaults
        rc = hip.lib().lfm_dit_forward(C.byref(shape), C.byref(w), hip.ptr(ws), ws.numel(), C.byref(call), hip.stream_ptr(x.device))
        hip.check(rc, "lfm_dit_forward")
        return out

    def forward(self, t, x, y=None, **kwargs):
        """v = model(t, x, y)  (reference DiT.py:252-272).  t: 0-d / [1] / [N];  x: [N,C,R,R];  y: [N] int64 or None."""
        return self._run(t, x, y, False, 1.0)

    def forward_with_cfg(self, t, x, y=None, cfg_scale=1.0, **kwargs):
        """Reference DiT.py:274-290: x[:N/2] is evaluated with y[:N/2] (cond) and y[N/2:] (uncond); both output halves
        carry uncond + cfg_scale*(cond-uncond)."""
        if x.shape[0] % 2:
            raise ValueError("forward_with_cfg needs an even batch (cond | uncond)")
        return self._run(t, x, y, True, cfg_scale)


# --------------------------------------------------------------------------- configs (reference DiT.py:354-415)
def _cfg(depth, hidden, patch, heads):
    return lambda **kw: DiT(depth=depth, hidden_size=hidden, patch_size=patch, num_heads=heads, **kw)


DiT_models = {
    "DiT-XL/2": _cfg(28, 1152, 2, 16), "DiT-XL/4": _cfg(28, 1152, 4, 16), "DiT-XL/8": _cfg(28, 1152, 8, 16),
    "DiT-L/2": _cfg(24, 1024, 2, 16), "DiT-L/4": _cfg(24, 1024, 4, 16), "DiT-L/8": _cfg(24, 1024, 8, 16),
    "DiT-B/2": _cfg(12, 768, 2, 12), "DiT-B/4": _cfg(12, 768, 4, 12), "DiT-B/8": _cfg(12, 768, 8, 12),
    "DiT-S/2": _cfg(12, 384, 2, 6), "DiT-S/4": _cfg(12, 384, 4, 6), "DiT-S/8": _cfg(12, 384, 8, 6),
}
_ = math
