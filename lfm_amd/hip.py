"""ctypes binding of liblfm_hip.so (C ABI: include/lfm_hip.h).  Fails loudly -- no fallback."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "liblfm_hip.so")
_lib = None


class LfmHipError(RuntimeError):
    pass


class DitShape(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("depth", "hidden", "heads", "patch", "in_ch", "res", "mlp_hidden", "label_rows")]


_DIT_WEIGHT_FIELDS = ("pos_embed", "patch_w", "patch_b", "t_w0", "t_b0", "t_w2", "t_b2", "y_table", "ada_w", "ada_b",
                      "qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "final_w", "final_b", "patch_w16")


class DitWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _DIT_WEIGHT_FIELDS]


class DitCall(C.Structure):
    _fields_ = [("batch", C.c_int), ("x", C.c_void_p), ("t", C.c_void_p), ("t_len", C.c_int), ("y", C.c_void_p),
                ("cfg", C.c_int), ("cfg_scale", C.c_float), ("out", C.c_void_p), ("axpy_base", C.c_void_p), ("axpy_dt", C.c_void_p),
                ("cond_table", C.c_void_p), ("cond_step", C.c_void_p), ("cond_offset", C.c_int),
                ("cond_rows", C.c_int), ("fold_ln", C.c_int), ("gemm_select", C.c_int)]  # ABI 4; zero = "as before" (ctypes zero-fills omitted fields)


ABI_VERSION = 4  # include/lfm_hip.h: lfm_abi_version()
CALL_OFF, CALL_ON = 1, 2  # lfm_dit_call.fold_ln


def call_gemm_select(which):
    """lfm_dit_call.gemm_select value for the kernel selection `which` (what lfm_gemm_select would take): per call, not library-wide."""
    return int(which) + 1


def lib():
    """Load the shared library.  With hipcc present the stamp-checked in-tree build runs first (a no-op when the sources are
    unchanged, a rebuild when they are stale); without hipcc the prebuilt library is used as shipped."""
    global _lib
    if _lib is not None:
        return _lib
    from . import _build

    override = os.environ.get("LFM_HIP_LIBRARY")  # measurement A/B only (tools/): a prebuilt experiment build of the SAME ABI instead of the in-tree one
    if override:
        if not os.path.exists(override):
            raise LfmHipError(f"LFM_HIP_LIBRARY={override} does not exist")
        path = override
    else:
        path = LIB_PATH
    if not override and (os.path.exists(_build.HIPCC) or not os.path.exists(LIB_PATH)):
        try:
            _build.build()
        except Exception as e:  # noqa
            if not os.path.exists(LIB_PATH):
                raise LfmHipError(f"liblfm_hip.so is missing and could not be built: {e}") from e
            raise LfmHipError(f"liblfm_hip.so is stale and could not be rebuilt: {e}") from e
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise LfmHipError(f"cannot load {path}: {e}") from e
    L.lfm_strerror.restype = C.c_char_p
    L.lfm_strerror.argtypes = [C.c_int]
    L.lfm_abi_version.restype = C.c_int
    if L.lfm_abi_version() != ABI_VERSION:  # a stale prebuilt library would read the call structs of another layout
        raise LfmHipError(f"{path} has ABI {L.lfm_abi_version()}, this package binds ABI {ABI_VERSION}: rebuild it (python -m lfm_amd._build --force)")
    L.lfm_dit_workspace_bytes.restype = C.c_size_t
    L.lfm_dit_workspace_bytes.argtypes = [C.POINTER(DitShape), C.c_int]
    L.lfm_dit_forward.restype = C.c_int
    L.lfm_dit_forward.argtypes = [C.POINTER(DitShape), C.POINTER(DitWeights), C.c_void_p, C.c_size_t, C.POINTER(DitCall), C.c_void_p]
    L.lfm_gemm_f16.restype = C.c_int
    L.lfm_gemm_f16.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    L.lfm_dit_cond_table_bytes.restype = C.c_size_t
    L.lfm_dit_cond_table_bytes.argtypes = [C.c_void_p, C.c_int]
    L.lfm_dit_cond_table_build.restype = C.c_int
    L.lfm_dit_cond_table_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.lfm_clock_probe.restype = C.c_int
    L.lfm_clock_probe.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.lfm_gemm_select.restype = C.c_int
    L.lfm_gemm_select.argtypes = [C.c_int]
    L.lfm_gemm_qkv_f16.restype = C.c_int
    L.lfm_gemm_qkv_f16.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    if hasattr(L, "lfm_gemm_trace_read"):  # LFM_MEASURE builds only
        L.lfm_gemm_trace_read.restype = C.c_int
        L.lfm_gemm_trace_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        L.lfm_attention_trace_read.restype = C.c_int
        L.lfm_attention_trace_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    L.lfm_profile_fc1.restype = C.c_int
    L.lfm_profile_fc1.argtypes = [C.c_int]
    L.lfm_profile_fc1_read.restype = C.c_int
    L.lfm_profile_fc1_read.argtypes = [C.POINTER(C.c_float), C.c_int]
    L.lfm_profile_blocks_read.restype = C.c_int
    L.lfm_profile_blocks_read.argtypes = [C.POINTER(C.c_float), C.c_int]
    L.lfm_ln_modulate.restype = C.c_int
    L.lfm_ln_modulate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    L.lfm_dit_attention.restype = C.c_int
    L.lfm_dit_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.lfm_set_option.restype = C.c_int
    L.lfm_set_option.argtypes = [C.c_int, C.c_int]
    L.lfm_dit_call_settings.restype = C.c_int
    L.lfm_dit_call_settings.argtypes = [C.POINTER(DitCall), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lfm_dit_plan.restype = C.c_int
    L.lfm_dit_plan.argtypes = [C.POINTER(DitShape), C.POINTER(DitCall), C.POINTER(C.c_int)]
    L.lfm_dit_attention_hd.restype = C.c_int
    L.lfm_dit_attention_hd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.lfm_grid_advance.restype = C.c_int
    L.lfm_grid_advance.argtypes = [C.c_void_p] * 7
    L.lfm_lincomb.restype = C.c_int
    L.lfm_lincomb.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_void_p]
    L.lfm_rk_error_norm.restype = C.c_int
    L.lfm_rk_error_norm.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
    L.lfm_vae_workspace_bytes.restype = C.c_size_t
    L.lfm_vae_workspace_bytes.argtypes = [C.c_int, C.c_int]
    L.lfm_vae_decode.restype = C.c_int
    L.lfm_vae_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.lfm_vae_encode.restype = C.c_int
    L.lfm_vae_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.lfm_images_to_uint8.restype = C.c_int
    L.lfm_images_to_uint8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.lfm_images_to_uint8_mode.restype = C.c_int
    L.lfm_images_to_uint8_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    V, I, LG, F = C.c_void_p, C.c_int, C.c_long, C.c_float
    L.lfm_conv3x3_f16.restype = I
    L.lfm_conv3x3_f16.argtypes = [V, V, V, V, V, I, I, I, I, I, I, V]
    L.lfm_conv3x3_workspace_bytes.restype = C.c_size_t
    L.lfm_conv3x3_workspace_bytes.argtypes = [I, I, I, I, I]
    L.lfm_conv3x3_f16_ws.restype = I
    L.lfm_conv3x3_f16_ws.argtypes = [V, V, V, V, V, I, I, I, I, I, I, V, C.c_size_t, V]
    L.lfm_conv3x3_in_f32.restype = I
    L.lfm_conv3x3_in_f32.argtypes = [V, V, V, V, I, I, I, I, I, V]
    L.lfm_conv3x3_out_f32.restype = I
    L.lfm_conv3x3_out_f32.argtypes = [V, V, V, V, I, I, I, I, I, V]
    L.lfm_linear_f16.restype = I
    L.lfm_linear_f16.argtypes = [V, LG, V, LG, V, LG, I, I, I, V, V, V]
    L.lfm_groupnorm_scratch_bytes.restype = C.c_size_t
    L.lfm_groupnorm_scratch_bytes.argtypes = [I, I]
    L.lfm_groupnorm_f16.restype = I
    L.lfm_groupnorm_f16.argtypes = [V, V, V, V, V, LG, V, I, I, I, I, F, I, V]
    L.lfm_groupnorm2_f16.restype = I
    L.lfm_groupnorm2_f16.argtypes = [V, I, V, I, V, V, V, V, LG, V, I, I, I, F, I, V]
    L.lfm_linear2_f16.restype = I
    L.lfm_linear2_f16.argtypes = [V, I, V, I, V, LG, V, LG, I, I, V, V, V]
    L.lfm_avgpool2_f16.restype = I
    L.lfm_avgpool2_f16.argtypes = [V, V, I, I, I, I, V]
    L.lfm_upsample2_f16.restype = I
    L.lfm_upsample2_f16.argtypes = [V, V, I, I, I, I, V]
    L.lfm_concat_channels_f16.restype = I
    L.lfm_concat_channels_f16.argtypes = [V, V, V, LG, I, I, V]
    L.lfm_add_image_vec_f16.restype = I
    L.lfm_add_image_vec_f16.argtypes = [V, V, LG, V, I, I, I, V]
    L.lfm_attention_small_f16.restype = I
    L.lfm_attention_small_f16.argtypes = [V, V, I, I, I, I, V]
    L.lfm_time_embed.restype = I
    L.lfm_time_embed.argtypes = [V, I, V, V, V, V, V, V, I, V, V, V, I, I, I, V]
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise LfmHipError(f"{what} failed: {lib().lfm_strerror(rc).decode()} (code {rc})")


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream (so launches are ordered with torch ops and graph-capturable)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def check_labels(y, rows, what):
    """Labels must index the embedding table (the reference's nn.Embedding raises IndexError otherwise; a kernel cannot).  The check
    reads two scalars back from the device, so the verdict is remembered ON THE TENSOR OBJECT (attribute, keyed by its version counter and
    the table size) -- a solver loop that passes the same label tensor to every evaluation pays once, while a fresh tensor that merely
    reuses a freed tensor's address is checked again.  Skipped during graph capture (no readback possible there)."""
    if y is None or y.numel() == 0:
        return
    if y.is_cuda and torch.cuda.is_current_stream_capturing():
        return
    key = (y._version, y.numel(), rows)
    if getattr(y, "_lfm_labels_ok", None) == key:
        return
    lo, hi = (int(v) for v in torch.stack([y.min(), y.max()]).tolist())  # one readback
    if lo < 0 or hi >= rows:
        raise IndexError(f"{what}: label {hi if hi >= rows else lo} is outside the embedding table [0, {rows}) "
                         "(classifier-free guidance needs a model built with label_dropout > 0: its null class is row num_classes)")
    try:
        y._lfm_labels_ok = key
    except Exception:  # noqa
        pass


def require_gpu(t, what):
    if not t.is_cuda:
        raise LfmHipError(f"{what}: tensor is on {t.device}; the LFM hot path only runs on an MI355X (no CPU fallback)")


# ----------------------------------------------------------------------------- thin op wrappers (used by tests)
def gemm_select(which):
    """Force a GEMM kernel (0 auto, 1 128x128, 4 256x128, 5 256x256 eight waves, 6 256x256 four waves; 7 / 8: gemm_f16 on the batch-1 kernels, 64x64 tiles /
    all rows x 16 columns) | ablation flags << 4; A/B measurements
    and parity tests only.  Raises on a value the library rejects (a silently ignored selection invalidates an A/B)."""
    check(lib().lfm_gemm_select(int(which)), "lfm_gemm_select")


def gemm_f16(A, W, bias=None, epilogue=0, out=None, gate=None, gate_stride=0, tokens=1):
    require_gpu(A, "gemm_f16")
    M, K = A.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32 if epilogue in (2, 3) else torch.float16)
    check(lib().lfm_gemm_f16(ptr(A), A.stride(0), ptr(W), W.stride(0), ptr(out), out.stride(0), M, N, K, ptr(bias), epilogue,
                             ptr(gate), gate_stride, tokens, stream_ptr()), "lfm_gemm_f16")
    return out


def gemm_qkv_f16(A, W, bias, head_dim, tokens):
    """Q, K [M, D] and V^T [M/tokens, D/head_dim, head_dim, tokens] of the fused attention input projection."""
    require_gpu(A, "gemm_qkv_f16")
    M, K = A.shape
    D = W.shape[0] // 3
    Q = torch.empty(M, D, device=A.device, dtype=torch.float16)
    Kt = torch.empty_like(Q)
    Vt = torch.empty(M // tokens, D // head_dim, head_dim, tokens, device=A.device, dtype=torch.float16)
    check(lib().lfm_gemm_qkv_f16(ptr(A), A.stride(0), ptr(W), W.stride(0), ptr(Q), ptr(Kt), ptr(Vt), M, D, K, ptr(bias), head_dim, tokens,
                                 stream_ptr()), "lfm_gemm_qkv_f16")
    return Q, Kt, Vt


def ln_modulate(X, shift, scale, tokens, mod_stride):
    require_gpu(X, "ln_modulate")
    M, D = X.shape
    A = torch.empty(M, D, device=X.device, dtype=torch.float16)
    check(lib().lfm_ln_modulate(ptr(X), ptr(A), M, D, tokens, ptr(shift), ptr(scale), mod_stride, stream_ptr()), "lfm_ln_modulate")
    return A


OPT_FOLD_LN = 1  # adaLN LayerNorm-modulate folded into the GEMM epilogues, default on (include/lfm_hip.h)
OPT_SKINNY_GEMM = 4  # batch-1 DiT linears on the latency-mode kernels (1, default: csrc/gemm_sq64_kernel.h where rows % 64 == 0, else csrc/gemm_skinny_kernel.h; 2: always the latter; 0: split-K path)
OPT_ATTENTION_STREAM = 5  # 256-token hd-64 attention on persistent workgroups with an LDS ring of K / V^T stages (csrc/attention_stream_kernel.h), default on; 0: one workgroup per item
OPT_FUSED_QKV_ATTENTION = 6  # folded path, 256 tokens x hd 64: QKV projection + attention in one kernel (csrc/qkv_attention_kernel.h), default on; 0: two kernels
OPT_GEMM_V6 = 2  # chip-filling row-major GEMMs on the one-wave-per-SIMD 256x256 kernel (csrc/gemm256w_kernel.h) instead of the 8-wave one


def effective_clock_mhz(device=None, iters=20000):
    """The clock this GPU sustains under matrix load (lfm_clock_probe: one workgroup per CU streaming MFMAs on random operands for ~50 ms): s_memtime
    ticks of the slowest workgroup over the launch's wall time.  Boxes differ by several percent under the board power cap."""
    dev = torch.device(device or "cuda:0")
    n = torch.cuda.get_device_properties(dev).multi_processor_count
    ticks = torch.zeros(n, dtype=torch.int64, device=dev)
    for it in (200, iters):  # a short warm-up launch, then the measured one
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
        check(lib().lfm_clock_probe(n, it, ptr(ticks), stream_ptr(dev)), "lfm_clock_probe")
        e1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1)
    # Calibration (tools/clock_probe.py on an MI355X, round 4): the tick count is proportional to the MFMA count, not to wall time (20 802 844 ticks for
    # 2 560 000 MFMAs per SIMD in a 21.6 ms and in a 23.7 ms run) -- 8.13 ticks per 16-pass MFMA, i.e. one s_memtime tick = TWO shader cycles here; the
    # issue-bound estimate 16 cycles x MFMAs / wall time agrees within 2 %.
    return 2.0 * float(ticks.max()) / (ms * 1e3)


PLAN_FOLDED_LN, PLAN_FUSED_QKV_ATTENTION = 1, 2  # lfm_dit_plan bits


def dit_plan(shape, batch, t_len=1, labels=False, fold_ln=0, gemm_select=0):
    """LFM_PLAN_* bits of the block loop lfm_dit_forward would run for this shape / batch under the current library options (no launch, no GPU needed)."""
    call = DitCall()
    call.batch, call.t_len, call.y = int(batch), int(t_len), (1 if labels else None)  # y: only NULL-ness is read
    call.fold_ln, call.gemm_select = fold_ln, gemm_select
    plan = C.c_int(-1)
    check(lib().lfm_dit_plan(C.byref(shape), C.byref(call), C.byref(plan)), "lfm_dit_plan")
    return plan.value


def set_option(key, value):
    check(lib().lfm_set_option(int(key), int(value)), "lfm_set_option")


def vt_token_perm(T, device=None):
    """Index tensor p with Vt_library[..., i] = Vt_plain[..., p[i]]: the V^T token order of the library (tokens of every 16-group stored as 0-3, 8-11, 4-7,
    12-15; include/lfm_hip.h: lfm_gemm_qkv_f16).  An involution: the same index restores the plain order."""
    i = torch.arange(T, device=device)
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)


def dit_attention(Q, K, Vt, batch, heads, T, head_dim=64):
    """Q, K, O: fp16 [batch*T, heads*head_dim]; Vt: fp16 [batch, heads, head_dim, T] in the library's token order (vt_token_perm).  head_dim 64 or 72."""
    require_gpu(Q, "dit_attention")
    O = torch.empty_like(Q)
    check(lib().lfm_dit_attention_hd(ptr(Q), ptr(K), ptr(Vt), ptr(O), batch, heads, head_dim, T, stream_ptr()), "lfm_dit_attention_hd")
    return O


def rk_error_norm(y0, y1, ks, e_coef, dt, rtol, atol, scratch, out):
    """out[0] = sqrt(mean(((dt * sum e_j k_j) / (atol + rtol max(|y0|, |y1|)))^2)) -- device fp32 tensors; e_coef, dt, out on the device."""
    require_gpu(y0, "rk_error_norm")
    arr = (C.c_void_p * len(ks))(*[k.data_ptr() for k in ks])
    check(lib().lfm_rk_error_norm(ptr(y0), ptr(y1), arr, ptr(e_coef), ptr(dt), len(ks), y0.numel(), float(rtol), float(atol), ptr(scratch), ptr(out),
                                  stream_ptr(y0.device)), "lfm_rk_error_norm")
    return out


def lincomb(out, base, ks, coef, scale=None):
    """out = base + (*scale) * sum coef[i]*ks[i]  (device fp32 tensors; coef/scale are device tensors)."""
    require_gpu(out, "lincomb")
    arr = (C.c_void_p * len(ks))(*[k.data_ptr() for k in ks])
    check(lib().lfm_lincomb(ptr(out), ptr(base), arr, ptr(coef), ptr(scale), len(ks), out.numel(), stream_ptr(out.device)), "lfm_lincomb")
    return out
