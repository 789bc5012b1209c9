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
/* The ABI this header describes.  The call structs below GROW between ABI versions (lfm_dit_call gained fields in 3 and 4) and the library reads every
 * field of the struct it is handed, so a caller compiled against an older header would pass a struct that is too short: a C / cgo / JNI caller checks
 * `lfm_abi_version() == LFM_ABI_VERSION` ONCE, after loading the library and before the first call (the Python binding refuses a mismatching library at
 * load time: lfm_amd/hip.py; tests/test_c_abi.py does it from C). */
#define LFM_ABI_VERSION 4
int lfm_abi_version(void);

/* ------------------------------------------------------------------ DiT velocity field
 * Shape of one DiT (models/DiT.py:157-184, configs :354-415). */
typedef struct lfm_dit_shape {
  int depth;      /* number of DiTBlocks */
  int hidden;     /* D */
  int heads;      /* head_dim = D / heads must be 64 or 72 (DiT-S / B / L: 64; DiT-XL: 1152 / 16 = 72) */
  int patch;      /* p: 2 (register patch-embed kernel), 4 or 8 (p*p*C % 64 == 0: patch embedding on the MFMA GEMM) */
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
  const void* patch_w16;  /* fp16 [D, C*p*p]: x_embedder.proj.weight as a GEMM operand; needed when p*p*C > 16 (DiT-x/4, x/8), else may be NULL */
} lfm_dit_weights;

/* One evaluation of the velocity field, with the solver update optionally fused into the last kernel. */
typedef struct lfm_dit_call {
  int batch;              /* rows the network evaluates (already doubled under CFG)                   */
  const float* x;         /* [batch, C, R, R] fp32 NCHW                                               */
  const float* t;         /* [t_len] fp32, t_len = 1 (0-d / [1] time) or batch                        */
  int t_len;
  const int64_t* y;       /* [batch] class labels in [0, label_rows) (else the row is NaN), or NULL => row label_rows-1 (models/DiT.py:259-260) */
  int cfg;                /* 0: DiT.forward.  1: DiT.forward_with_cfg (models/DiT.py:274-290): rows
                             [0,batch/2) of x are used for BOTH halves, and both output halves carry
                             uncond + cfg_scale*(cond - uncond)                                        */
  float cfg_scale;
  float* out;             /* [batch, C, R, R]                                                         */
  const float* axpy_base; /* NULL: out = v.   else out = axpy_base + (*axpy_dt) * v  (may alias out)  */
  const float* axpy_dt;   /* device scalar (read at kernel run time => one captured graph per solver) */
  /* ABI 3: per-grid conditioning table (lfm_dit_cond_table_build), or NULL.  Only for evaluations with ONE shared conditioning row (t_len == 1, y == NULL):
   * everything the forward derives from t alone is then copied from table row *cond_step + cond_offset instead of being recomputed; t is not read. */
  const void* cond_table;
  const int* cond_step;   /* device int (a fixed-grid solver's interval counter) */
  int cond_offset;
  /* ABI 4.  Callers ZERO-INITIALISE the struct (memset / `= {0}`): every field below reads 0 as "as before". */
  int cond_rows;          /* rows of cond_table (the n_times it was built with).  > 0: a row index *cond_step + cond_offset outside [0, cond_rows) is not
                             read -- the evaluation's conditioning, and with it its output, becomes NaN (a kernel cannot return an error); 0: unchecked */
  int fold_ln;            /* per-call LFM_OPT_FOLD_LN: 0 = the library default (lfm_set_option), LFM_CALL_OFF = separate LayerNorm launches, LFM_CALL_ON = folded */
  int gemm_select;        /* per-call kernel selection: 0 = the library default (lfm_gemm_select), else LFM_CALL_GEMM_SELECT(value lfm_gemm_select would take) */
} lfm_dit_call;
#define LFM_CALL_OFF 1
#define LFM_CALL_ON 2
#define LFM_CALL_GEMM_SELECT(which) ((which) + 1)

/* Bytes of caller-owned scratch for batches up to max_batch: residual stream, LN / attention buffers, Q|K|Vt (reused for the fc1
 * activation), conditioning vectors, the adaLN table and -- for small batches (<= 1024 tokens, latency mode) -- the fp32 slabs of the
 * split-K GEMMs (reserved for every max_batch, so the requirement is monotone: a workspace sized for max_batch serves every smaller batch). */
size_t lfm_dit_workspace_bytes(const lfm_dit_shape* shape, int max_batch);

/* Replaces DiT.forward / DiT.forward_with_cfg (models/DiT.py:252-290) as called by the solver closure
 * `denoiser` (test_flow_latent.py:55-59) and sampler/karras_sample.py:42-46. */
int lfm_dit_forward(const lfm_dit_shape* shape, const lfm_dit_weights* w, void* workspace, size_t workspace_bytes,
                    const lfm_dit_call* call, lfm_stream_t stream);

/* Which form of the block loop lfm_dit_forward would take for `call` if it were enqueued by the calling thread now (library defaults, per-call fields, shape and
 * batch): *plan_out = LFM_PLAN_FOLDED_LN (adaLN LayerNorm-modulate folded into the GEMM epilogues: LFM_OPT_FOLD_LN) | LFM_PLAN_FUSED_QKV_ATTENTION (QKV projection
 * + attention core as one kernel: LFM_OPT_FUSED_QKV_ATTENTION).  Reads call->batch, t_len, y (NULL or not), fold_ln, gemm_select; no launch; usable without a GPU. */
#define LFM_PLAN_FOLDED_LN 1
#define LFM_PLAN_FUSED_QKV_ATTENTION 2
int lfm_dit_plan(const lfm_dit_shape* shape, const lfm_dit_call* call, int* plan_out);

/* Per-grid conditioning tables for the unconditional models (test_args/{celeb256,ffhq,bed,church}_dit.txt: num_classes 1, no labels, scalar t -- the
 * closure `denoiser` of test_flow_latent.py:55-59 calls model(t, x) with the solver's grid time): c = t_embedder(t), the adaLN modulation of every block and
 * of the final layer (models/DiT.py:128, 170, 259-262) and the u / v rows of the folded LayerNorm path are pure functions of t, so a fixed-grid solver
 * computes them ONCE per grid time.  Rows are written by the same launches an evaluation would make: results are bit-identical.
 * t_values: device [n_times]; table: device, lfm_dit_cond_table_bytes(shape, n_times) bytes, 16-byte aligned; workspace / batch: as for lfm_dit_forward
 * (scratch).  The table depends on the weights and on the grid, not on x. */
size_t lfm_dit_cond_table_bytes(const lfm_dit_shape* shape, int n_times);
int lfm_dit_cond_table_build(const lfm_dit_shape* shape, const lfm_dit_weights* w, void* workspace, size_t workspace_bytes, int batch, const float* t_values,
                             int n_times, void* table, size_t table_bytes, lfm_stream_t stream);

/* ------------------------------------------------------------------ building blocks (exported for parity tests)
 * C[M,N] (+)= A[M,K] * W[N,K]^T on MFMA, fp16 operands, fp32 accumulate.  epilogue:
 *   0: C fp16 = acc + bias          1: C fp16 = gelu_tanh(acc + bias)       2: C fp32 = acc + bias
 *   3: C fp32 += gate[m / tokens][n] * (acc + bias)   (gate row stride gate_stride, 0 = shared row)
 * Replaces the nn.Linear calls inside timm Attention/Mlp (models/DiT.py:120,124) and adaLN (:128). */
int lfm_gemm_f16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias,
                 int epilogue, const float* gate, long gate_stride, int tokens, lfm_stream_t stream);

/* The attention input projection with its fused split (timm Attention.qkv + reshape/permute, models/DiT.py:120):
 * [Q | K | V] = A[M,K] * W[3D,K]^T + bias;  Q, K fp16 [M, D] row-major;  V is written TRANSPOSED per head as
 * Vt[M / tokens][D / head_dim][head_dim][tokens] (what lfm_dit_attention reads), with the tokens of every group of 16 stored in the order
 * 0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15 (position p of a row holds token p with bits 2 and 3 exchanged: the operand order of the attention kernel's
 * P V MFMA, csrc/gemm_kernel.h: vt_pos; ABI 3).  head_dim % 8 == 0, tokens % 16 == 0. */
int lfm_gemm_qkv_f16(const void* A, long lda, const void* W, long ldw, void* Q, void* K_out, void* Vt, int M, int D, int K,
                     const float* bias, int head_dim, int tokens, lfm_stream_t stream);

/* Scope of the three library-wide switches below (lfm_gemm_select, lfm_set_option, lfm_profile_fc1): they set process-wide DEFAULTS and are meant for
 * measurement and tests; they are read when a call is ENQUEUED, on the calling thread.  A caller that needs a setting for ONE evaluation -- two
 * lanes in flight, several host threads -- passes it in lfm_dit_call (fold_ln, gemm_select): the per-call values live in thread-local state for the
 * duration of lfm_dit_forward and never touch the defaults, so concurrent callers do not see each other's choices (tests/test_c_abi.py).
 * Kernel selection for the GEMMs: 0 = automatic (the 256x256 quadrant-phased kernel on 16x16x32 MFMAs for chip-filling shapes with K % 64 == 0,
 * the 256x128 two-workgroups-per-CU kernel for chip-filling shapes that are only 128 columns wide, the 128x128 kernel otherwise), 1 = force
 * 128x128, 4 = force 256x128, 5 = force 256x256 with eight waves, 6 = force 256x256 with one wave per SIMD (128x128 wave tiles; row-major A
 * operands, K % 64 == 0 -- other cases take kernel 5); 7 / 8 = lfm_gemm_f16 (epilogues 0-2, <= 256 rows) on the latency-mode kernels -- 64x64 tiles /
 * all rows x 16 columns -- which refuse other shapes, every other entry point treats them as 1 (other values are refused); | flags << 4 =
 * ablation / A-B switches.  For measurement and parity tests. */
int lfm_gemm_select(int which);

/* Measurement only: when enabled, every eager lfm_dit_forward records a HIP event pair around each block's fc1 GEMM (the
 * dominant kernel); lfm_profile_fc1_read synchronises and returns the per-launch durations in ms (bench.py roofline row).  One stream at a time:
 * the first stream that launches an evaluation while the probe is on owns it; an evaluation enqueued on ANOTHER stream meanwhile is not recorded
 * and makes lfm_profile_fc1_read return LFM_ERR_ARG (event pairs of interleaved streams would time the other stream's kernels too). */
int lfm_profile_fc1(int enable);
/* With the same probe on, every recorded evaluation also brackets its WHOLE block loop (all DiTBlocks: qkv .. fc2, models/DiT.py:112-131) with one
 * event pair: host_ms_out receives one duration per evaluation (at most 16); block time = duration / depth (bench.py `roofline_block`).
 * lfm_profile_fc1(2) records ONLY these (no events between the block's kernels). */
int lfm_profile_blocks_read(float* host_ms_out, int max_n);
/* Library-wide options.  key 1 (LFM_OPT_FOLD_LN), value 0 / 1, default 1: modulate(LayerNorm(x), shift, scale) (models/DiT.py:20-21, 129-130) FOLDED
 * into the GEMM epilogues around it -- the gated-residual GEMMs (proj, fc2) emit the centred, (1 + scale)-weighted fp16 operand and per-row partial
 * sums, the consuming GEMMs (qkv, fc1) apply rstd, the mean correction and the shift term in their epilogues -- wherever the shape allows it
 * (residual width a multiple of 256, whole 256-row tiles of one image or one shared conditioning row, chip-filling batch); 0 = always the separate
 * lfm_ln_modulate launches (the path every other shape takes).  Same result up to fp16 rounding of the operand (tests/test_gpu_dit.py). */
#define LFM_OPT_FOLD_LN 1
/* key 2 (LFM_OPT_GEMM_V6), value 0 / 1: the chip-filling row-major GEMMs (the four linears of a DiT block) on the one-wave-per-SIMD 256x256 kernel
 * (csrc/gemm256w_kernel.h) instead of the eight-wave one.  Same accumulation order per output element: bit-identical results. */
#define LFM_OPT_GEMM_V6 2
/* key 4 (LFM_OPT_SKINNY_GEMM), value 0 / 1 / 2, default 1: evaluations of <= 256 token rows (ONE image of 256 tokens: --measure_time, test_flow_latent.py:223-246)
 * run the four linears of a DiTBlock on the latency-mode kernels -- 64x64 tiles (csrc/gemm_sq64_kernel.h) where the rows are whole 64-row tiles, else all rows x
 * 16 columns (csrc/gemm_skinny_kernel.h); 2 = always the latter; 0 = the split-K 128x128 path (A/B and parity tests). */
#define LFM_OPT_SKINNY_GEMM 4
/* key 5 (LFM_OPT_ATTENTION_STREAM), value 0 / 1, default 1: lfm_dit_attention at 256 tokens x head_dim 64 with more than 64 (image, head) items runs on persistent
 * workgroups that stream K / V^T of consecutive items through an LDS ring (csrc/attention_stream_kernel.h); 0 = one workgroup per item (csrc/attention_kernel.h).
 * Same arithmetic in the same order per query: bit-identical results (tests/test_gpu_dit.py). */
#define LFM_OPT_ATTENTION_STREAM 5
/* key 6 (LFM_OPT_FUSED_QKV_ATTENTION), value 0 / 1, default 1: on the folded path, at 256 tokens per image and head_dim 64, the QKV projection and the attention
 * core of a DiTBlock (models/DiT.py:120) run as ONE kernel -- a workgroup per (image, head) computes the head's 256 x 192 slice of the projection and attends on
 * it out of the LDS; Q, K, V never reach HBM (csrc/qkv_attention_kernel.h); 0 = two kernels.  Same arithmetic in the same order: bit-identical results. */
#define LFM_OPT_FUSED_QKV_ATTENTION 6
int lfm_set_option(int key, int value);
/* The settings lfm_dit_forward would run `call` with if it were enqueued by the calling thread now (per-call fields over the library defaults):
 * *gemm_select_out = kernel | flags << 4, *fold_ln_out = 0 / 1.  No launch; usable without a GPU. */
int lfm_dit_call_settings(const lfm_dit_call* call, int* gemm_select_out, int* fold_ln_out);

/* Measurement aid (bench.py): the clock the chip sustains under matrix load.  `blocks` workgroups of 512 threads each stream iters x 64 MFMAs per wave on
 * pseudo-random fp16 operands between two s_memtime reads; ticks_out[blocks] (device) receives the tick count of every workgroup (one tick = one shader
 * cycle).  Sustained clock = ticks / the launch's wall time (HIP events around the call). */
int lfm_clock_probe(int blocks, int iters, unsigned long long* ticks_out, lfm_stream_t stream);

/* ---- measurement-only entry points: exported by LFM_MEASURE builds of the library only (LFM_MEASURE=1 python -m lfm_amd._build; tools/README.md) */
#ifdef LFM_MEASURE
/* Measurement only: s_memtime stamps written by the quadrant-phased GEMM (select flag 2) after every barrier of block 0,
 * wave groups 0 and 1; host_out receives 2 x n_per_group values. */
int lfm_gemm_trace_read(unsigned long long* host_out, int n_per_group);
/* Measurement only: s_memtime stamps of the attention kernel's trace build (select flags 33554432 | 67108864; slot map in csrc/attention_kernel.h). */
int lfm_attention_trace_read(unsigned long long* host_out, int n);
/* Measurement only (LFM_MEASURE builds, same trace build): per attention workgroup {HW_ID | XCC_ID << 32, start, loads landed, end} -- which CU it ran on
 * and when; host_out receives 4 x n_wg values (n_wg <= 2048). */
int lfm_attention_wg_trace_read(unsigned long long* host_out, int n_wg);
/* Measurement only: per-kernel checksums of the evaluations that run on `workspace` (csrc/dit.hip: slot [block][8]); tools/concurrency_ws_diff.py. */
int lfm_dit_chk_arm(const void* workspace);
int lfm_dit_chk_read(unsigned long long* host_out, int n);
#endif /* LFM_MEASURE */
int lfm_profile_fc1_read(float* host_ms_out, int max_n);

/* A fp16 [M,D] = LayerNorm(X fp32 [M,D], eps 1e-6, no affine) * (1 + scale[img]) + shift[img]
 * (models/DiT.py:20-21,119,121,129-130).  mod_stride = floats between images' rows (0 = shared). */
int lfm_ln_modulate(const float* X, void* A, int M, int D, int tokens, const float* shift, const float* scale, long mod_stride,
                    lfm_stream_t stream);

/* softmax(q k^T / sqrt(hd)) v for hd = 64, T in {16,64,128,256,1024} (timm Attention as called at models/DiT.py:120; 1024 = four key chunks of 256).
 * Q,K: fp16 [batch*T, D] token-major; Vt: fp16 [batch, heads, hd, T] in the token order lfm_gemm_qkv_f16 writes (16-groups permuted); O: fp16 [batch*T, D]. */
int lfm_dit_attention(const void* Q, const void* K, const void* Vt, void* O, int batch, int heads, int T, lfm_stream_t stream);
/* The same with the head size as an argument: head_dim 64 (DiT-S / B / L) or 72 (DiT-XL/{2,4,8}: 1152 / 16, models/DiT.py:354-363);
 * D = heads * head_dim. */
int lfm_dit_attention_hd(const void* Q, const void* K, const void* Vt, void* O, int batch, int heads, int head_dim, int T,
                         lfm_stream_t stream);

/* ------------------------------------------------------------------ first-stage VAE decoder
 * Decode half of diffusers AutoencoderKL, config stabilityai/sd-vae-ft-mse (latent 4, block_out_channels
 * [128,256,512,512], 2 layers/block, 32 GN groups, eps 1e-6), as the reference calls it:
 *   test_flow_latent.py:131,193   first_stage_model.decode(fake_sample / args.scale_factor).sample
 * conv3x3 weights are fp16 [Cout][tap = ky*3+kx][Cin]; 1x1 / linear weights fp16 [Cout][Cin]; GN affine and
 * biases fp32.  NULL sc_w = identity shortcut. */
typedef struct lfm_vae_resnet {
  const float* n1_g; const float* n1_b; const void* c1_w; const float* c1_b;
  const float* n2_g; const float* n2_b; const void* c2_w; const float* c2_b;
  const void* sc_w; const float* sc_b;
  int cin, cout;
} lfm_vae_resnet;

typedef struct lfm_vae_weights {
  const float* pq_w; const float* pq_b;   /* post_quant_conv [4,4], [4]           fp32 */
  const float* cin_w; const float* cin_b; /* decoder.conv_in [512,4,3,3], [512]   fp32 */
  lfm_vae_resnet mid[2];                  /* decoder.mid_block.resnets.{0,1}           */
  const float* at_g; const float* at_b;   /* decoder.mid_block.attentions.0.group_norm */
  const void* q_w; const float* q_b; const void* k_w; const float* k_b;
  const void* v_w; const float* v_b; const void* o_w; const float* o_b;   /* to_q/to_k/to_v/to_out.0 */
  lfm_vae_resnet up[4][3];                /* decoder.up_blocks.i.resnets.j             */
  const void* ups_w[3]; const float* ups_b[3]; /* decoder.up_blocks.i.upsamplers.0.conv */
  const float* no_g; const float* no_b;   /* decoder.conv_norm_out                     */
  const void* cout_w; const float* cout_b; /* decoder.conv_out padded to 4 outputs: fp16 [4][9][128], fp32 [4] */
} lfm_vae_weights;

size_t lfm_vae_workspace_bytes(int R, int chunk);

/* out[N,3,8R,8R] fp32 NCHW = decode(z[N,4,R,R] fp32 NCHW); images are processed `chunk` at a time so the
 * activation working set (4 x chunk x (8R)^2 x 256 fp16) stays bounded. */
int lfm_vae_decode(const lfm_vae_weights* w, void* workspace, size_t workspace_bytes, const float* z, float* out, int N, int R,
                   int chunk, lfm_stream_t stream);

/* Encoder half (diffusers AutoencoderKL.encode as called at train_flow_latent.py:143 and
 * downstream_tasks/test_flow_latent_inpainting.py:146).  fp16 tensors are GEMM operands [Cout][tap][Cin]. */
typedef struct lfm_vae_enc_weights {
  const float* cin_w; const float* cin_b;   /* encoder.conv_in [128,3,3,3], [128]         fp32 */
  lfm_vae_resnet down[4][2];                /* encoder.down_blocks.i.resnets.j                 */
  const void* ds_w[3]; const float* ds_b[3]; /* encoder.down_blocks.i.downsamplers.0.conv (stride 2, pad (0,1,0,1)) */
  lfm_vae_resnet mid[2];                    /* encoder.mid_block.resnets.{0,1}                 */
  const float* at_g; const float* at_b;     /* encoder.mid_block.attentions.0.group_norm       */
  const void* q_w; const float* q_b; const void* k_w; const float* k_b;
  const void* v_w; const float* v_b; const void* o_w; const float* o_b;
  const float* no_g; const float* no_b;     /* encoder.conv_norm_out                           */
  const void* cout_w; const float* cout_b;  /* quant_conv o encoder.conv_out folded: fp16 [8][9][512], fp32 [8] */
} lfm_vae_enc_weights;

/* moments[N,8,R,R] fp32 NCHW (mean | log-variance of the latent distribution) = quant_conv(Encoder(x)), x[N,3,8R,8R] fp32 NCHW.
 * Workspace: lfm_vae_workspace_bytes(R, chunk), as for the decoder. */
int lfm_vae_encode(const lfm_vae_enc_weights* w, void* workspace, size_t workspace_bytes, const float* x, float* moments, int N, int R,
                   int chunk, lfm_stream_t stream);

/* u8 NHWC = trunc(clamp((x+1)/2, 0, 1) * 255) of fp32 NCHW images (test_flow_latent_ddp.py:131-135). */
int lfm_images_to_uint8(const float* x, uint8_t* out, int N, int H, int W, lfm_stream_t stream);
/* Same with rounding != 0: trunc(clamp((x+1)/2, 0, 1) * 255 + 0.5), the conversion of torchvision.utils.save_image that the
 * single-process script applies (test_flow_latent.py:264-269,297). */
int lfm_images_to_uint8_mode(const float* x, uint8_t* out, int N, int H, int W, int rounding, lfm_stream_t stream);

/* ------------------------------------------------------------------ NHWC-fp16 building blocks (origin-ADM UNet)
 * The guided-diffusion UNet (models/guided_diffusion/unet.py:376-655) has a data-dependent layer list, so its forward is
 * sequenced by the host (lfm_amd/models/unet.py) over these ops.  Activations: fp16 NHWC [N*H*W, C].
 *
 * lfm_conv3x3_f16: out[N,H,W,Cout] = conv3x3(in, pad 1) + bias (+ resid); w fp16 [Cout][ky*3+kx][Cin], Cin % 64 == 0.
 *   mode 0: in is [N,H,W,Cin] (ResBlock convs, unet.py:171-175,193-198);  mode 1: in is [N,H/2,W/2,Cin], nearest-2x upsampled on
 *   the fly (Upsample, :73-100);  mode 2: in is [N,2H,2W,Cin], stride 2 (Downsample, :103-128).
 *   Kernel choice (same result up to the fp32 summation order): modes 0 / 1 with H, W multiples of 16, Cout % 128 == 0 and at least 256
 *   (16x16-pixel tile, 128-channel block) pairs run on the halo-tiled direct kernel (csrc/conv_halo_kernel.h); everything else is an
 *   implicit GEMM (split-K with a workspace for the small maps). */
int lfm_conv3x3_f16(const void* in, const void* w, const float* bias, const void* resid, void* out, int N, int H, int W, int Cin, int Cout,
                    int mode, lfm_stream_t stream);
/* The same with a caller-owned workspace of lfm_conv3x3_workspace_bytes(...) bytes (0 = none needed): small-M / huge-K convolutions (the
 * low-resolution levels of the UNets) then run split-K with a deterministic slab reduction.  workspace may be NULL. */
size_t lfm_conv3x3_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int lfm_conv3x3_f16_ws(const void* in, const void* w, const float* bias, const void* resid, void* out, int N, int H, int W, int Cin, int Cout,
                       int mode, void* workspace, size_t workspace_bytes, lfm_stream_t stream);
/* first conv (unet.py:475): fp32 NCHW [N,Cin<=16,H,W] -> fp16 NHWC [N,H,W,Cout]; w fp32 [Cout,Cin,3,3] */
int lfm_conv3x3_in_f32(const float* x_nchw, const float* w, const float* bias, void* out_nhwc, int N, int H, int W, int Cin, int Cout,
                       lfm_stream_t stream);
/* last conv (unet.py:594): fp16 NHWC -> fp32 NCHW [N,nch<=4,H,W]; w4 fp16 [4][9][Cin] (rows >= nch zero), bias4 fp32 [4] */
int lfm_conv3x3_out_f32(const void* in, const void* w4, const float* bias4, float* out_nchw, int N, int H, int W, int Cin, int nch,
                        lfm_stream_t stream);
/* C fp16 [M,N] = A[M,K] W[N,K]^T + bias (+ resid fp16 [M,N]): 1x1 convs / Conv1d(k=1) / skip connections (unet.py:204,266,276) */
int lfm_linear_f16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias,
                   const void* resid, lfm_stream_t stream);
/* The same with A = the channel concat [A1 (K1 columns) | A2 (K2 columns)] of two dense fp16 tensors read in place: the 1x1 skip convolution of an
 * output-block ResBlock, whose input is th.cat([h, hs.pop()], dim=1) (unet.py:649 + :236).  K1 % 64 == 0, K2 % 8 == 0. */
int lfm_linear2_f16(const void* A1, int K1, const void* A2, int K2, const void* W, long ldw, void* C, long ldc, int M, int N, const float* bias,
                    const void* resid, lfm_stream_t stream);
/* y = silu?( GroupNorm(groups <= 32)(x; gamma, beta, eps) * (1 + scale[n]) + shift[n] );  film = fp32 [N][scale(C) | shift(C)] rows film_stride apart,
 * or NULL (nn.py:17-19,93-100; scale-shift-norm unet.py:228-233).  scratch: lfm_groupnorm_scratch_bytes(N, C) bytes. */
size_t lfm_groupnorm_scratch_bytes(int N, int C);
int lfm_groupnorm_f16(const void* x, void* y, const float* gamma, const float* beta, const float* film, long film_stride, void* scratch, int N,
                      int HW, int C, int groups, float eps, int silu, lfm_stream_t stream);
/* The same on the channel concat [xa (Ca channels) | xb (Cb channels)] read in place (groups may straddle the seam); y is dense [N*HW, Ca + Cb];
 * scratch: lfm_groupnorm_scratch_bytes(N, Ca + Cb).  Ca % 8 == Cb % 8 == 0.  Bit-identical to lfm_concat_channels_f16 + lfm_groupnorm_f16. */
int lfm_groupnorm2_f16(const void* xa, int Ca, const void* xb, int Cb, void* y, const float* gamma, const float* beta, const float* film,
                       long film_stride, void* scratch, int N, int HW, int groups, float eps, int silu, lfm_stream_t stream);
/* y[N,Ho,Wo,C] = 2x2 mean of x[N,2Ho,2Wo,C]: EDM Conv2d(down=True) with the [1,1] resample filter (models/EDM.py:96-98,122-125) */
int lfm_avgpool2_f16(const void* x, void* y, int N, int Ho, int Wo, int C, lfm_stream_t stream);
/* y[N,Ho,Wo,C] = nearest-2x upsample of x[N,Ho/2,Wo/2,C]: EDM Conv2d(up=True, kernel=0) skip path (models/EDM.py:117-121) */
int lfm_upsample2_f16(const void* x, void* y, int N, int Ho, int Wo, int C, lfm_stream_t stream);
/* out[p][0:Ca | Ca:Ca+Cb] = a[p], b[p]   (th.cat([h, hs.pop()], dim=1), unet.py:649) */
int lfm_concat_channels_f16(const void* a, const void* b, void* out, long pixels, int Ca, int Cb, lfm_stream_t stream);
/* y = x + e[n][c] broadcast over the pixels of image n (ResBlock without scale-shift norm: h + emb_out[..., None, None], unet.py:233-235);
 * x, y fp16 NHWC [N*HW, C], e fp32 rows e_stride apart */
int lfm_add_image_vec_f16(const void* x, const float* e, long e_stride, void* y, int N, int HW, int C, lfm_stream_t stream);
/* QKVAttentionLegacy (unet.py:310-334): qkv fp16 [N*T, 3C], columns [head][q|k|v][ch]; out fp16 [N*T, C] columns [head][ch].
 * T = 64 / 256 with ch = 64 / 128 (16-byte aligned pointers): MFMA kernel; any other shape: a VALU kernel that keeps K, V and a 64-query score
 * block in the LDS (LFM_ERR_SHAPE when that exceeds 160 KiB). */
int lfm_attention_small_f16(const void* qkv, void* out, int N, int T, int heads, int ch, lfm_stream_t stream);
/* emb = time_embed(timestep_embedding(t, F)) (+ label_emb[y]) (nn.py:103-121, unet.py:633-641): fp32 [N,E] and fp16 silu(emb).
 * label_table has label_rows rows; a label outside [0, label_rows) (an IndexError in the reference) poisons its row with NaN. */
int lfm_time_embed(const float* t, int t_len, const float* w0, const float* b0, const float* w2, const float* b2, const float* label_table,
                   const int64_t* y, int label_rows, float* scratch_h1, float* emb, void* emb_silu_f16, int N, int F, int E,
                   lfm_stream_t stream);

/* ------------------------------------------------------------------ solver helpers (device-resident time grid)
 * Advance the captured step:  s = *step;  t_cur[0] = ts[s];  t_next[0] = ts[s+1];  dt_cur[0] = dts[s];  *step = s+1.
 * ts has n+1 entries, dts n.  Lets ONE captured graph be replayed for every interval of the fixed grids of
 * test_flow_latent.py:42-76 (torchdiffeq euler) and sampler/karras_sample.py:85-161 (Euler / Heun). */
int lfm_grid_advance(const float* ts, const float* dts, int* step, float* t_cur, float* t_next, float* dt_cur, lfm_stream_t stream);

/* out = base + (scale ? *scale : 1) * sum_i coef[i] * k[i]   (i < nk <= 8), n fp32 elements (n % 4 == 0).
 * coef and scale are DEVICE memory (read at run time); k_host_ptrs is a HOST array of nk device pointers.
 * Covers the Heun update x + dt*(0.5 d + 0.5 d') (karras_sample.py:157-159) and the Runge-Kutta stage sums of
 * torchdiffeq's rk4 / dopri5. */
int lfm_lincomb(float* out, const float* base, const float* const* k_host_ptrs, const float* coef, const float* scale, int nk, long n,
                lfm_stream_t stream);

/* Error ratio of one adaptive Runge-Kutta step, as torchdiffeq's controller takes it (RMS over the WHOLE state tensor; reference call site
 * test_flow_latent.py:61-73, dopri5 at rtol = atol = 1e-5):  out[0] = sqrt(mean_i ((*dt) * sum_j e_coef[j] k_j[i] / (atol + rtol max(|y0[i]|, |y1[i]|)))^2).
 * e_coef, dt, out: DEVICE memory; k_host_ptrs: HOST array of nk <= 8 device pointers; scratch: >= 1024 floats of device memory; n % 4 == 0.
 * Deterministic (two fixed-order stages); the only value of the step the host has to read. */
int lfm_rk_error_norm(const float* y0, const float* y1, const float* const* k_host_ptrs, const float* e_coef, const float* dt, int nk, long n,
                      float rtol, float atol, float* scratch, float* out, lfm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LFM_HIP_H */
