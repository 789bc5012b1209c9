/* liblfm_hip.so -- C ABI of the MI355X (gfx950) sampling hot path of LFM.
 *
 * The reference (VinAIResearch/LFM) is pure Python and has no FFI; the drop-in boundary is the Python
 * surface (create_network / model(t, x, y) / sample_from_model / AutoencoderKL.decode).  This header is
 * the C boundary UNDER that surface: each entry point names the reference function it replaces
 * (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes; every pointer is a DEVICE pointer unless named host_*;
 *   - the caller owns every buffer, including the workspace (no allocation, no sync, no throw);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*), so calls are graph-capturable;
 *   - return 0 on success, a negative LFM_ERR_* otherwise (see lfm_strerror).
 */
#ifndef LFM_HIP_H
#define LFM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lfm_stream_t; /* hipStream_t */

#define LFM_OK 0
#define LFM_ERR_SHAPE (-1)
#define LFM_ERR_ALIGN (-2)
#define LFM_ERR_WORKSPACE (-3)
#define LFM_ERR_LAUNCH (-4)
#define LFM_ERR_ARG (-5)

const char* lfm_strerror(int code);
int lfm_abi_version(void);

/* ------------------------------------------------------------------ DiT velocity field
 * Shape of one DiT (models/DiT.py:157-184, configs :354-415). */
typedef struct lfm_dit_shape {
  int depth;      /* number of DiTBlocks */
  int hidden;     /* D */
  int heads;      /* D / heads must be 64 */
  int patch;      /* p */
  int in_ch;      /* C (4 for f8 latents) */
  int res;        /* latent side R = image_size / f */
  int mlp_hidden; /* int(D * mlp_ratio) */
  int label_rows; /* rows of y_embedder.embedding_table = num_classes + (label_dropout > 0) */
} lfm_dit_shape;

/* Packed weights.  fp16 tensors are the GEMM operands, row-major [out, in] exactly like nn.Linear;
 * everything else stays fp32.  J = depth*6*D + 2*D: the adaLN_modulation.1 rows of blocks 0..depth-1
 * (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp; models/DiT.py:128) followed by
 * final_layer.adaLN_modulation.1 (shift, scale; :146). */
typedef struct lfm_dit_weights {
  const float* pos_embed; /* [T, D]            pos_embed (models/DiT.py:184,203-205)                  */
  const float* patch_w;   /* [D, C*p*p]        x_embedder.proj.weight flattened (c, p, q)             */
  const float* patch_b;   /* [D]                                                                      */
  const float* t_w0;      /* [D, 256]          t_embedder.mlp.0                                       */
  const float* t_b0;      /* [D]                                                                      */
  const float* t_w2;      /* [D, D]            t_embedder.mlp.2                                       */
  const float* t_b2;      /* [D]                                                                      */
  const float* y_table;   /* [label_rows, D]   y_embedder.embedding_table.weight                      */
  const void* ada_w;      /* fp16 [J, D]                                                              */
  const float* ada_b;     /* [J]                                                                      */
  const void* qkv_w;      /* fp16 [depth, 3D, D]   blocks.i.attn.qkv.weight                           */
  const float* qkv_b;     /* [depth, 3D]                                                              */
  const void* proj_w;     /* fp16 [depth, D, D]    blocks.i.attn.proj.weight                          */
  const float* proj_b;    /* [depth, D]                                                               */
  const void* fc1_w;      /* fp16 [depth, H, D]    blocks.i.mlp.fc1.weight                            */
  const float* fc1_b;     /* [depth, H]                                                               */
  const void* fc2_w;      /* fp16 [depth, D, H]    blocks.i.mlp.fc2.weight                            */
  const float* fc2_b;     /* [depth, D]                                                               */
  const float* final_w;   /* [p*p*C, D]        final_layer.linear.weight                              */
  const float* final_b;   /* [p*p*C]                                                                  */
} lfm_dit_weights;

/* One evaluation of the velocity field, with the solver update optionally fused into the last kernel. */
typedef struct lfm_dit_call {
  int batch;              /* rows the network evaluates (already doubled under CFG)                   */
  const float* x;         /* [batch, C, R, R] fp32 NCHW                                               */
  const float* t;         /* [t_len] fp32, t_len = 1 (0-d / [1] time) or batch                        */
  int t_len;
  const int64_t* y;       /* [batch] class labels, or NULL => row label_rows-1 (models/DiT.py:259-260) */
  int cfg;                /* 0: DiT.forward.  1: DiT.forward_with_cfg (models/DiT.py:274-290): rows
                             [0,batch/2) of x are used for BOTH halves, and both output halves carry
                             uncond + cfg_scale*(cond - uncond)                                        */
  float cfg_scale;
  float* out;             /* [batch, C, R, R]                                                         */
  const float* axpy_base; /* NULL: out = v.   else out = axpy_base + (*axpy_dt) * v  (may alias out)  */
  const float* axpy_dt;   /* device scalar (read at kernel run time => one captured graph per solver) */
} lfm_dit_call;

size_t lfm_dit_workspace_bytes(const lfm_dit_shape* shape, int max_batch);

/* Replaces DiT.forward / DiT.forward_with_cfg (models/DiT.py:252-290) as called by the solver closure
 * `denoiser` (test_flow_latent.py:55-59) and sampler/karras_sample.py:42-46. */
int lfm_dit_forward(const lfm_dit_shape* shape, const lfm_dit_weights* w, void* workspace, size_t workspace_bytes,
                    const lfm_dit_call* call, lfm_stream_t stream);

/* ------------------------------------------------------------------ building blocks (exported for parity tests)
 * C[M,N] (+)= A[M,K] * W[N,K]^T on MFMA, fp16 operands, fp32 accumulate.  epilogue:
 *   0: C fp16 = acc + bias          1: C fp16 = gelu_tanh(acc + bias)       2: C fp32 = acc + bias
 *   3: C fp32 += gate[m / tokens][n] * (acc + bias)   (gate row stride gate_stride, 0 = shared row)
 * Replaces the nn.Linear calls inside timm Attention/Mlp (models/DiT.py:120,124) and adaLN (:128). */
int lfm_gemm_f16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias,
                 int epilogue, const float* gate, long gate_stride, int tokens, lfm_stream_t stream);

/* A fp16 [M,D] = LayerNorm(X fp32 [M,D], eps 1e-6, no affine) * (1 + scale[img]) + shift[img]
 * (models/DiT.py:20-21,119,121,129-130).  mod_stride = floats between images' rows (0 = shared). */
int lfm_ln_modulate(const float* X, void* A, int M, int D, int tokens, const float* shift, const float* scale, long mod_stride,
                    lfm_stream_t stream);

/* softmax(q k^T / sqrt(hd)) v for hd = 64, T in {64,128,256} (timm Attention as called at models/DiT.py:120).
 * Q,K: fp16 [batch*T, D] token-major; Vt: fp16 [batch, heads, hd, T]; O: fp16 [batch*T, D]. */
int lfm_dit_attention(const void* Q, const void* K, const void* Vt, void* O, int batch, int heads, int T, lfm_stream_t stream);

/* ------------------------------------------------------------------ solver helpers (device-resident time grid)
 * Advance the captured step: t_cur[0] = ts[*step]; dt_cur[0] = dts[*step]; ++*step.  Lets ONE captured
 * graph be replayed for every step of the fixed grids of test_flow_latent.py:42-76 (torchdiffeq euler)
 * and sampler/karras_sample.py:85-161. */
int lfm_grid_advance(const float* ts, const float* dts, int* step, float* t_cur, float* dt_cur, lfm_stream_t stream);

/* out = base + sum_i coef[i] * k[i]  (i < nk <= 8), n elements; coef is a DEVICE array (RK stage combos). */
int lfm_lincomb(float* out, const float* base, const float* const* k_host_ptrs, const float* coef, int nk, long n,
                lfm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LFM_HIP_H */
