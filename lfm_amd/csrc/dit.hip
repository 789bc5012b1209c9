// DiT velocity field on gfx950: the kernels around the MFMA GEMMs and the per-call driver.
// Reference behaviour: /root/reference/models/DiT.py (cited per kernel).
#include <mutex>

#include "../../include/lfm_hip.h"
#include "gemm_dispatch.h"
#include "gemm_skinny_kernel.h"
#include "gemm_sq64_kernel.h"
#include "qkv_attention_kernel.h"

// Library-wide switches = process-wide DEFAULTS (lfm_gemm_select, lfm_set_option); a call that carries its own values (lfm_dit_call.fold_ln /
// .gemm_select, ABI 4) overrides them in THREAD-LOCAL state for the duration of lfm_dit_forward: every launcher reads the effective value on the
// calling thread while it enqueues, so two host threads (or two lanes in flight with different settings) never see each other's choice.
static int g_gemm_sel_default = 0, g_gemm_dbg_default = 0;
static int g_opt_fold_ln_default = 1;
static thread_local int tl_sel_set = 0, tl_gemm_sel = 0, tl_gemm_dbg = 0;  // per-call kernel selection active on this thread
static thread_local int tl_fold = 0;                                       // 0: default, LFM_CALL_OFF, LFM_CALL_ON
#define g_gemm_sel (tl_sel_set ? tl_gemm_sel : g_gemm_sel_default)
#define g_gemm_dbg (tl_sel_set ? tl_gemm_dbg : g_gemm_dbg_default)
#define g_opt_fold_ln (tl_fold ? (tl_fold == LFM_CALL_ON ? 1 : 0) : g_opt_fold_ln_default)
int lfm_gemm_selected() { return g_gemm_sel; }
int lfm_gemm_debug_flags() { return g_gemm_dbg; }
int lfm_gemm_selected_v1_ok() { return g_gemm_sel < 2 && !(g_gemm_dbg & 512); }  // flag 512: split-K off (A/B)
// flag 4096: never pick v4 for 256-wide-capable shapes; flag 8192: always (A/B of the automatic choice)
int lfm_gemm_prefers_v4(int M, int N, int K) {
  (void)M;
  (void)N;
  (void)K;
  if (g_gemm_dbg & 4096) return 0;
  if (g_gemm_dbg & 8192) return 1;
  return 0;
}
static int g_opt_v6 = 0;
int lfm_gemm_v6_default() { return g_opt_v6; }
static int g_opt_skinny = 1;  // LFM_OPT_SKINNY_GEMM: the batch-1 DiT linears on their own kernels (1: 64x64 tiles where the image has whole 64-token tiles,
                               // gemm_sq64_kernel.h, else all rows x 16 columns, gemm_skinny_kernel.h; 2: always the latter; 0: the rounds 2-4 split-K path)
static int g_stagger = 0;
int lfm_stagger_ticks() { return g_stagger; }
static int g_opt_att_stream = 1;  // LFM_OPT_ATTENTION_STREAM: 256 tokens x head_dim 64 with more than 64 (image, head) items on the persistent streamed kernel
int lfm_attention_stream_enabled() { return g_opt_att_stream; }
static int g_opt_fused_qkv = 1;  // LFM_OPT_FUSED_QKV_ATTENTION: folded path at 256 tokens x head_dim 64: QKV projection + attention in one kernel (qkv_attention_kernel.h)
static inline bool gemm_select_valid(int which) {
  const int k = which & 15;
  return which >= 0 && (k == 0 || k == 1 || k == 4 || k == 5 || k == 6 || k == 7 || k == 8);  // 7, 8: the latency-mode kernels through lfm_gemm_f16 (tests)
}
struct CallScope {  // per-call settings of one lfm_dit_forward on this thread (restored on every return path)
  int sel_set, sel, dbg, fold;
  CallScope() : sel_set(tl_sel_set), sel(tl_gemm_sel), dbg(tl_gemm_dbg), fold(tl_fold) {}
  ~CallScope() {
    tl_sel_set = sel_set;
    tl_gemm_sel = sel;
    tl_gemm_dbg = dbg;
    tl_fold = fold;
  }
};

static int call_scope_enter(const lfm_dit_call* c) {  // after a CallScope was opened on this thread
  if (c->fold_ln < 0 || c->fold_ln > LFM_CALL_ON || c->gemm_select < 0 || (c->gemm_select && !gemm_select_valid(c->gemm_select - 1)) || c->cond_rows < 0)
    return LFM_ERR_ARG;
  if (c->fold_ln) tl_fold = c->fold_ln;
  if (c->gemm_select) {
    tl_sel_set = 1;
    tl_gemm_sel = (c->gemm_select - 1) & 15;
    tl_gemm_dbg = (c->gemm_select - 1) >> 4;
  }
  return LFM_OK;
}
// What lfm_dit_forward would run `call` with on the calling thread, right now: gemm_select_out = kernel | flags << 4, fold_ln_out = 0 / 1.  Goes through
// the same scope code as the forward (no launch, no GPU needed): the CPU test of the per-call / per-thread scoping (tests/test_c_abi.py).
extern "C" int lfm_dit_call_settings(const lfm_dit_call* call, int* gemm_select_out, int* fold_ln_out) {
  if (!call || !gemm_select_out || !fold_ln_out) return LFM_ERR_ARG;
  CallScope scope;
  const int rc = call_scope_enter(call);
  if (rc) return rc;
  *gemm_select_out = g_gemm_sel | (g_gemm_dbg << 4);
  *fold_ln_out = g_opt_fold_ln;
  return LFM_OK;
}

extern "C" int lfm_set_option(int key, int value) {  // key 1 (LFM_OPT_FOLD_LN): adaLN LayerNorm-modulate folded into the GEMM epilogues (default 1)
  if (key == 1) {
    g_opt_fold_ln_default = value != 0;
    return LFM_OK;
  }
  if (key == 2) {  // LFM_OPT_GEMM_V6: the one-wave-per-SIMD 256x256 kernel for the chip-filling row-major GEMMs
    g_opt_v6 = value != 0;
    return LFM_OK;
  }
  if (key == 4) {  // LFM_OPT_SKINNY_GEMM: 0 = the split-K 128x128 path of rounds 2-4 for M <= 256 (A/B, parity)
    g_opt_skinny = value < 0 || value > 2 ? 1 : value;
    return LFM_OK;
  }
  if (key == 5) {  // LFM_OPT_ATTENTION_STREAM: 0 = one workgroup per (image, head) item (the rounds 1-5 kernel; A/B and the bit-equality test)
    g_opt_att_stream = value != 0;
    return LFM_OK;
  }
  if (key == 6) {  // LFM_OPT_FUSED_QKV_ATTENTION: 0 = the QKV GEMM and the attention kernel as two launches (A/B and the bit-equality test)
    g_opt_fused_qkv = value != 0;
    return LFM_OK;
  }
#ifdef LFM_MEASURE
  if (key == 3) {  // measurement: start offset (s_memtime ticks) of the second resident workgroups of the two-per-CU kernels (gemm256_common.h)
    g_stagger = value > 0 ? value : 0;
    return LFM_OK;
  }
#endif
  return LFM_ERR_ARG;
}
extern "C" int lfm_gemm_select(int which) {  // low 4 bits: kernel choice (0 auto, 1, 4, 5, 6; 7, 8 for lfm_gemm_f16 only); bits 4+: ablation flags (measurement only)
  if (!gemm_select_valid(which)) return LFM_ERR_ARG;
  g_gemm_sel_default = which & 15;
  g_gemm_dbg_default = which >> 4;
  return LFM_OK;
}

// ------------------------------------------------------------------ timestep embedder (DiT.py:29-69)
// temb[r] = W2 * silu(W0 * [cos(t f) | sin(t f)] + b0) + b2, fp32 throughout; one wave per output element.
__global__ __launch_bounds__(256) void temb1_kernel(const float* __restrict__ t, const float* __restrict__ w0, const float* __restrict__ b0,
                                                    float* __restrict__ h1, int D) {
  const int r = blockIdx.y, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= D) return;
  const float tv = t[r];
  const float* w = w0 + (long)j * 256;
  float s = 0.f;
#pragma unroll
  for (int k = lane; k < 256; k += 64) {
    const int i = k & 127;
    const float a = tv * expf(-9.210340371976184f * (float)i / 128.0f);  // t * exp(-ln(1e4) i/half)
    s += w[k] * (k < 128 ? cosf(a) : sinf(a));
  }
  s = wave_sum(s);
  if (lane == 0) h1[(long)r * D + j] = silu_f(s + b0[j]);
}
__global__ __launch_bounds__(256) void temb2_kernel(const float* __restrict__ h1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                    float* __restrict__ temb, int D) {
  const int r = blockIdx.y, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= D) return;
  const float* w = w2 + (long)j * D;
  const float* h = h1 + (long)r * D;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) s += w[k] * h[k];
  s = wave_sum(s);
  if (lane == 0) temb[(long)r * D + j] = s + b2[j];
}

// c_half[r] = fp16(silu(temb[t_len==1 ? 0 : r] + y_table[y ? y[r] : null_row]))   (DiT.py:259-264 + the SiLU of :125)
// A label outside [0, label_rows) is an IndexError in the reference (nn.Embedding); a kernel inside a captured graph cannot raise,
// so the row is POISONED with NaN instead of reading out of bounds (the host wrapper validates labels before they get here).
__global__ void cond_kernel(const float* __restrict__ temb, int t_len, const float* __restrict__ y_table, const int64_t* __restrict__ y,
                            int label_rows, half_t* __restrict__ c_half, int D, int rows) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * D) return;
  const int r = (int)(i / D), j = (int)(i - (long)r * D);
  const long yr = y ? (long)y[r] : (long)(label_rows - 1);
  if (yr < 0 || yr >= label_rows) {
    c_half[i] = (half_t)__builtin_nanf("");
    return;
  }
  const float v = temb[(t_len == 1 ? 0 : (long)r * D) + j] + y_table[yr * D + j];
  c_half[i] = (half_t)silu_f(v);
}

// ------------------------------------------------------------------ patch embed (timm PatchEmbed + pos_embed, DiT.py:179,261)
// X[n*T + tok][j] = b[j] + pos[tok][j] + sum_{c,p,q} W[j][c][p][q] * x[n % xmod][c][hp+p][wp+q]
// blockDim = D/4 threads, thread = 4 consecutive output channels whose weight rows stay in registers (KK <= 16 here);
// a block walks PE_TOK tokens, whose KK input values are wave-uniform loads.
#define PE_TOK 16
#define PE_MAXK 16
__global__ __launch_bounds__(320) void patch_embed_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                          const float* __restrict__ pos, float* __restrict__ X, int M, int xmod, int C, int R,
                                                          int p, int D) {
  __shared__ float xs[PE_TOK][PE_MAXK];
  const int grid = R / p, T = grid * grid, KK = C * p * p;
  const long m_begin = (long)blockIdx.x * PE_TOK;
  for (int e = threadIdx.x; e < PE_TOK * PE_MAXK; e += blockDim.x) {  // stage the inputs: the token loop has no dependent global loads
    const int tt = e / PE_MAXK, k = e % PE_MAXK;
    const long m = m_begin + tt;
    float v = 0.f;
    if (m < M && k < KK) {
      const int tok = (int)(m % T), n = (int)(m / T) % xmod;
      const int c = k / (p * p), pp = (k / p) % p, q = k % p;
      v = x[(((long)n * C + c) * R + (tok / grid) * p + pp) * R + (tok % grid) * p + q];
    }
    xs[tt][k] = v;
  }
  const int j = threadIdx.x * 4;
  float wr[4][PE_MAXK];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < PE_MAXK; ++k) wr[i][k] = (k < KK) ? w[(long)(j + i) * KK + k] : 0.f;
  const f32x4 bias = *(const f32x4*)(b + j);
  // every position-embedding row of the block's tokens is fetched BEFORE the first store: vmcnt counts stores too and returns in order, so a
  // load issued behind a store waits for that store's round trip (58 us per launch with the load inside the token loop)
  f32x4 pv[PE_TOK];
#pragma unroll
  for (int tt = 0; tt < PE_TOK; ++tt) {
    const long m = (m_begin + tt < M) ? m_begin + tt : M - 1;
    pv[tt] = *(const f32x4*)(pos + (long)(m % T) * D + j);
  }
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < PE_TOK; ++tt) {
    const long m = m_begin + tt;
    if (m >= M) break;
    f32x4 acc = bias + pv[tt];
#pragma unroll
    for (int k = 0; k < PE_MAXK; ++k) {
      const float xv = xs[tt][k];
      acc.x += wr[0][k] * xv;
      acc.y += wr[1][k] * xv;
      acc.z += wr[2][k] * xv;
      acc.w += wr[3][k] * xv;
    }
    *(f32x4*)(X + m * D + j) = acc;
  }
}

// Round 3: the patch embedding of the */2 models (K = p*p*C = 16) on v_mfma_f32_16x16x16_f16, fused with the FIRST LayerNorm of the forward.
// patch_embed_kernel above spends its time on 256 LDS broadcast reads and 1024 scalar FMAs per thread (55 us for 67 MB); the LayerNorm after it
// re-read the 67 MB it had just written (ln_center_mod_kernel / ln_modulate, 17-20 us).  Here a block of D / 256 waves walks 16-token tiles; wave w
// owns channels 256 w .. + 255 with its weight fragments (fp16 hi + lo, three MFMAs per tile pair = the fp32 dot product to 2^-22) and bias rows
// resident in registers.  The W rows of a 32-channel pair are fed through the permutation n = 8 (a >> 2) + 4 e + (a & 3) (a = fragment row, e = which
// MFMA of the pair), so a lane ends up with EIGHT CONSECUTIVE channels of one token: X leaves as 2 x 16-byte stores (four lanes = one 128-byte
// line), the fp16 operand of the first qkv GEMM as one.  The row statistics are in-lane sums + two lane exchanges + one LDS hand-over between the
// waves; the variance is the exact two-pass one (the values stay in registers).  With A == nullptr only X is written.
__global__ __launch_bounds__(320) void patch_embed_ln_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                             const float* __restrict__ pos, float* __restrict__ X, int M, int xmod, int R, int D,
                                                             int tiles_per_block, half_t* __restrict__ A, const float* __restrict__ scale,
                                                             long mod_stride, float* __restrict__ part, int tiles_p, float* __restrict__ cen) {
  typedef half_t half4v __attribute__((ext_vector_type(4)));
  __shared__ float red[2][5][16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, a = lane & 15, q = lane >> 4;
  const int grid = R >> 1, T = grid * grid;
  // resident weight fragments and bias rows of this wave's 8 channel pairs
  half4v wh[8][2], wl[8][2];
  f32x4 bias[8][2];
#pragma unroll
  for (int pr = 0; pr < 8; ++pr)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int n = 256 * wv + 32 * pr + 8 * (a >> 2) + 4 * e + (a & 3);  // the W row this lane feeds as fragment row a
      const f32x4 wf = *(const f32x4*)(w + (long)n * 16 + 4 * q);
      const float wa[4] = {wf.x, wf.y, wf.z, wf.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wh[pr][e][i] = (half_t)wa[i];
        wl[pr][e][i] = (half_t)(wa[i] - (float)wh[pr][e][i]);
      }
      bias[pr][e] = *(const f32x4*)(b + 256 * wv + 32 * pr + 8 * q + 4 * e);  // the channels this lane OWNS in the result
    }
  for (int it = 0; it < tiles_per_block; ++it) {
    const long tile = (long)blockIdx.x * tiles_per_block + it;
    if (tile * 16 >= M) break;  // (whole block)
    const long m = tile * 16 + a < M ? tile * 16 + a : M - 1;
    const int tok = (int)(m % T), n_img = (int)(m / T) % xmod;
    // the token's patch values k = 4 q .. 4 q + 3 = channel q, 2 x 2 pixels
    const float* xp = x + (((long)n_img * 4 + q) * R + (tok / grid) * 2) * R + (tok % grid) * 2;
    const f32x2 r0 = *(const f32x2*)xp, r1 = *(const f32x2*)(xp + R);
    const float xa[4] = {r0.x, r0.y, r1.x, r1.y};
    half4v xh, xl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xh[i] = (half_t)xa[i];
      xl[i] = (half_t)(xa[i] - (float)xh[i]);
    }
    f32x4 val[8][2];
    float sx = 0.f;
    const float* prow = pos + (long)tok * D + 256 * wv + 8 * q;
    float* xrow = X + m * D + 256 * wv + 8 * q;
    const bool live = tile * 16 + a < M;
#pragma unroll
    for (int pr = 0; pr < 8; ++pr)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wh[pr][e], xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wl[pr][e], xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wh[pr][e], xl, acc, 0, 0, 0);
        const f32x4 v = (f32x4){acc[0], acc[1], acc[2], acc[3]} + bias[pr][e] + *(const f32x4*)(prow + 32 * pr + 4 * e);
        val[pr][e] = v;
        if (live) *(f32x4*)(xrow + 32 * pr + 4 * e) = v;
        sx += (v.x + v.y) + (v.z + v.w);
      }
    if (!A) continue;
    sx += __shfl_xor(sx, 16, 64);
    sx += __shfl_xor(sx, 32, 64);
    if (q == 0) red[0][wv][a] = sx;
    __syncthreads();
    float sum = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) sum += red[0][i][a];
    const float mean = sum / (float)D;
    const float* srow = scale + (m / T) * mod_stride + 256 * wv + 8 * q;
    half_t* arow = A + m * D + 256 * wv + 8 * q;
    float sq = 0.f;
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) {
      const f32x4 d0 = val[pr][0] - mean, d1 = val[pr][1] - mean;
      sq += (d0.x * d0.x + d0.y * d0.y) + (d0.z * d0.z + d0.w * d0.w) + (d1.x * d1.x + d1.y * d1.y) + (d1.z * d1.z + d1.w * d1.w);
      const f32x4 o0 = d0 * (1.0f + *(const f32x4*)(srow + 32 * pr)), o1 = d1 * (1.0f + *(const f32x4*)(srow + 32 * pr + 4));
      const half8_t h = {(half_t)o0.x, (half_t)o0.y, (half_t)o0.z, (half_t)o0.w, (half_t)o1.x, (half_t)o1.y, (half_t)o1.z, (half_t)o1.w};
      if (live) *(half8_t*)(arow + 32 * pr) = h;
    }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (q == 0) red[1][wv][a] = sq;
    __syncthreads();
    if (wv == 0 && q == 0 && live) {
      float qs = 0.f;
      for (int i = 0; i < (int)(blockDim.x >> 6); ++i) qs += red[1][i][a];
      float* pp = part + m * tiles_p * 2;
      pp[0] = sum;
      pp[1] = qs;
      for (int t2 = 1; t2 < tiles_p; ++t2) {
        pp[2 * t2] = 0.f;
        pp[2 * t2 + 1] = 0.f;
      }
      cen[m] = mean;
    }
  }
}

// Patch sizes with p*p*C > 16 (DiT-*/4, */8): the patch embedding is a real GEMM (K = p*p*C = 64 / 256).  This kernel gathers the patches
// into the fp16 A operand Ap[m][k], k = (c, pp, q) as x_embedder.proj.weight flattens, and pre-fills the residual stream with the
// position embedding; the GEMM then adds  1 * (patches W^T + bias)  through the gated-residual epilogue (gate = a row of ones).
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, const float* __restrict__ pos, half_t* __restrict__ Ap,
                                                       float* __restrict__ X, float* __restrict__ ones, int M, int xmod, int C, int R, int p, int D) {
  const int grid = R / p, T = grid * grid, KK = C * p * p;
  const long m = blockIdx.x;
  const int tok = (int)(m % T), n = (int)(m / T) % xmod;
  for (int k = threadIdx.x; k < KK; k += 256) {
    const int c = k / (p * p), pp = (k / p) % p, q = k % p;
    Ap[m * KK + k] = (half_t)x[(((long)n * C + c) * R + (tok / grid) * p + pp) * R + (tok % grid) * p + q];
  }
  for (int j = threadIdx.x; j < D; j += 256) X[m * D + j] = pos[(long)tok * D + j];
  if (m == 0)
    for (int j = threadIdx.x; j < D; j += 256) ones[j] = 1.0f;
}

// ------------------------------------------------------------------ LayerNorm + modulate -> fp16 (DiT.py:20-21,119,129-130)
// one wave per token row; the row stays in registers (<= 5 float4 per lane => D <= 1280).
#define LN_MAXV 5
#define LN_ROWS 2  // rows per wave: both rows' loads are issued before either reduction, doubling the bytes in flight per wave
__global__ __launch_bounds__(256) void ln_modulate_kernel(const float* __restrict__ X, half_t* __restrict__ A, int M, int D, int tokens,
                                                          const float* __restrict__ shift, const float* __restrict__ scale, long mod_stride) {
  const int lane = threadIdx.x & 63;
  const long m0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_ROWS;
  if (m0 >= M) return;
  const int nv = D >> 2;
  f32x4 v[LN_ROWS][LN_MAXV];
  float s[LN_ROWS];
#pragma unroll
  for (int r = 0; r < LN_ROWS; ++r) {
    const long m = (m0 + r < M) ? m0 + r : M - 1;
    const f32x4* xr = (const f32x4*)(X + m * D);
    s[r] = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv) {
        v[r][i] = xr[c];
        s[r] += v[r][i].x + v[r][i].y + v[r][i].z + v[r][i].w;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < LN_ROWS; ++r) {
    const long m = m0 + r;
    if (m >= M) break;
    const float mean = wave_sum(s[r]) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv) {
        v[r][i] -= mean;
        q += v[r][i].x * v[r][i].x + v[r][i].y * v[r][i].y + v[r][i].z * v[r][i].z + v[r][i].w * v[r][i].w;
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-6f);
    const long mo = (m / tokens) * mod_stride;
    const f32x4* sh = (const f32x4*)(shift + mo);
    const f32x4* sc = (const f32x4*)(scale + mo);
    half4_t* ar = (half4_t*)(A + m * D);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv) {
        const f32x4 o = v[r][i] * rstd * (1.0f + sc[c]) + sh[c];
        half4_t h = {(half_t)o.x, (half_t)o.y, (half_t)o.z, (half_t)o.w};
        ar[c] = h;
      }
    }
  }
}

// Same computation with 16-byte stores: lane l owns EIGHT consecutive columns 8 (l + 64 i) .. + 7 (two adjacent float4 loads), so a row
// of the fp16 output goes out as 16 B per lane instead of 8 (the GEMM epilogues gained 3-4 % from the same change).  D % 8 == 0.
#define LN_MAXP 3  // column octets per lane: D <= 1536
// NR = rows per wave, DPP = row sums on the DPP cross-lane network instead of ds_bpermute (<1, true> ships; the others are A/B variants)
template <int NR, bool DPP = false>
__global__ __launch_bounds__(256) void ln_modulate8_kernel(const float* __restrict__ X, half_t* __restrict__ A, int M, int D, int tokens,
                                                           const float* __restrict__ shift, const float* __restrict__ scale, long mod_stride) {
  const int lane = threadIdx.x & 63;
  const long m0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * NR;
  if (m0 >= M) return;
  const int np = D >> 3;
  f32x4 v[NR][LN_MAXP][2];
  float s[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const long m = (m0 + r < M) ? m0 + r : M - 1;
    const f32x4* xr = (const f32x4*)(X + m * D);
    s[r] = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXP; ++i) {
      const int c = lane + 64 * i;
      if (c < np) {
        v[r][i][0] = xr[2 * c];
        v[r][i][1] = xr[2 * c + 1];
        const f32x4 t = v[r][i][0] + v[r][i][1];
        s[r] += (t.x + t.y) + (t.z + t.w);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const long m = m0 + r;
    if (m >= M) break;
    const float mean = (DPP ? wave_sum_dpp(s[r]) : wave_sum(s[r])) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXP; ++i) {
      if (lane + 64 * i < np) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          v[r][i][h] -= mean;
          const f32x4 t = v[r][i][h] * v[r][i][h];
          q += (t.x + t.y) + (t.z + t.w);
        }
      }
    }
    const float rstd = rsqrtf((DPP ? wave_sum_dpp(q) : wave_sum(q)) / (float)D + 1e-6f);
    const long mo = (m / tokens) * mod_stride;
    const f32x4* sh = (const f32x4*)(shift + mo);
    const f32x4* sc = (const f32x4*)(scale + mo);
    half8_t* ar = (half8_t*)(A + m * D);
#pragma unroll
    for (int i = 0; i < LN_MAXP; ++i) {
      const int c = lane + 64 * i;
      if (c < np) {
        const f32x4 lo = v[r][i][0] * rstd * (1.0f + sc[2 * c]) + sh[2 * c];
        const f32x4 hi = v[r][i][1] * rstd * (1.0f + sc[2 * c + 1]) + sh[2 * c + 1];
        half8_t h = {(half_t)lo.x, (half_t)lo.y, (half_t)lo.z, (half_t)lo.w, (half_t)hi.x, (half_t)hi.y, (half_t)hi.z, (half_t)hi.w};
        ar[c] = h;
      }
    }
  }
}

// Folded LayerNorm-modulate (gemm_kernel.h, "adaLN LayerNorm-modulate FOLDED into the GEMM epilogues"): the FIRST LayerNorm of a forward has no
// producer GEMM in front of it (x comes from the patch embedding), so this kernel plays the producer: A' = fp16((x - mu)(1 + scale)) with the exact
// row mean as the centring constant, partial slot 0 = (sum x, sum (x - mu)^2), the other slots 0, cen[m] = mu.  One row per wave, DPP sums.
__global__ __launch_bounds__(256) void ln_center_mod_kernel(const float* __restrict__ X, half_t* __restrict__ A, int M, int D, int tokens,
                                                            const float* __restrict__ scale, long mod_stride, float* __restrict__ part, int tiles_p,
                                                            float* __restrict__ cen) {
  const int lane = threadIdx.x & 63;
  const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int np = D >> 3;
  f32x4 v[LN_MAXP][2];
  const f32x4* xr = (const f32x4*)(X + m * D);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXP; ++i) {
    const int c = lane + 64 * i;
    if (c < np) {
      v[i][0] = xr[2 * c];
      v[i][1] = xr[2 * c + 1];
      const f32x4 t = v[i][0] + v[i][1];
      s += (t.x + t.y) + (t.z + t.w);
    }
  }
  const float sum = wave_sum_dpp(s), mean = sum / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXP; ++i) {
    if (lane + 64 * i < np) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v[i][h] -= mean;
        const f32x4 t = v[i][h] * v[i][h];
        q += (t.x + t.y) + (t.z + t.w);
      }
    }
  }
  const float qs = wave_sum_dpp(q);
  const f32x4* sc = (const f32x4*)(scale + (m / tokens) * mod_stride);
  half8_t* ar = (half8_t*)(A + m * D);
#pragma unroll
  for (int i = 0; i < LN_MAXP; ++i) {
    const int c = lane + 64 * i;
    if (c < np) {
      const f32x4 lo = v[i][0] * (1.0f + sc[2 * c]), hi = v[i][1] * (1.0f + sc[2 * c + 1]);
      half8_t h = {(half_t)lo.x, (half_t)lo.y, (half_t)lo.z, (half_t)lo.w, (half_t)hi.x, (half_t)hi.y, (half_t)hi.z, (half_t)hi.w};
      ar[c] = h;
    }
  }
  if (lane < tiles_p) *(f32x2*)(part + (m * tiles_p + lane) * 2) = lane == 0 ? (f32x2){sum, qs} : (f32x2){0.f, 0.f};
  if (lane == 0) cen[m] = mean;
}

// A operand of the u / v GEMMs of the folded path: for block i and branch b (0 = msa, 1 = mlp) rows [0, R) = fp16(1 + scale), rows [R, 2R) = fp16(shift);
// Amod[((i * 2 + b) * 2 + h) * R + r][k].  (u only ever multiplies rstd (mu - c), a small correction, and v takes the place of a term that used
// to be rounded to fp16 inside the LN output anyway: fp16 operands cost nothing here.)
__global__ __launch_bounds__(256) void mod_rows_f16_kernel(const float* __restrict__ mod, long mod_stride, int depth, int R, int D,
                                                           half_t* __restrict__ Amod) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // over depth * 2 * 2 * R * D / 4
  const int d4 = D >> 2;
  const long total = (long)depth * 4 * R * d4;
  if (idx >= total) return;
  const int k = (int)(idx % d4) * 4;
  long t = idx / d4;
  const int r = (int)(t % R);
  t /= R;
  const int h = (int)(t & 1), b = (int)((t >> 1) & 1), i = (int)(t >> 2);
  const float* src = mod + (long)r * mod_stride + (long)i * 6 * D + (b ? 3 * D : 0) + (h ? 0 : D) + k;  // h = 0: scale (+ 1), h = 1: shift
  f32x4 v = *(const f32x4*)src;
  if (!h) v += 1.0f;
  half4_t o = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
  *(half4_t*)(Amod + idx * 4) = o;
}

// u / v rows of the folded path when ONE conditioning row serves the whole batch (scalar time, no labels): a GEMV pair per weight row,
//   u[n] = sum_k (1 + scale[k]) W[n][k],   v[n] = sum_k shift[k] W[n][k] + bias[n],
// streamed straight from the fp16 weights with the fp32 modulation vectors in registers (no fp16 rounding of them at all).  One wave = eight
// weight rows (sixteen 16-byte loads in flight per lane); grid.y = block index.  This is pure weight streaming (352 MB per DiT-L/2 forward):
// the batched 128x128 MFMA GEMM it replaces for this case moved the same bytes at 4 TB/s with 126 of its 128 tile rows padding.
#define UV_ROWS 8
__global__ __launch_bounds__(256) void uv_gemv_kernel(const half_t* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ mod,
                                                      int N, int D, int scale_off, int shift_off, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, i = blockIdx.y;
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * UV_ROWS;
  if (n0 >= N) return;
  const int nch = D >> 3;  // 16-byte chunks per row
  const float* sc = mod + (long)i * 6 * D + scale_off;
  const float* sh = mod + (long)i * 6 * D + shift_off;
  f32x4 au[LN_MAXP][2], av[LN_MAXP][2];
#pragma unroll
  for (int j = 0; j < LN_MAXP; ++j) {
    const int c = lane + 64 * j;
    if (c < nch) {
      au[j][0] = *(const f32x4*)(sc + 8 * c) + 1.0f;
      au[j][1] = *(const f32x4*)(sc + 8 * c + 4) + 1.0f;
      av[j][0] = *(const f32x4*)(sh + 8 * c);
      av[j][1] = *(const f32x4*)(sh + 8 * c + 4);
    }
  }
  const half_t* wb = W + ((long)i * N + n0) * D;
  half8_t wv[UV_ROWS][LN_MAXP];
#pragma unroll
  for (int r = 0; r < UV_ROWS; ++r)
#pragma unroll
    for (int j = 0; j < LN_MAXP; ++j) {
      const int c = lane + 64 * j;
      if (c < nch && n0 + r < N) wv[r][j] = *(const half8_t*)(wb + (long)r * D + 8 * c);
    }
  float* ob = out + (long)i * 2 * N;
#pragma unroll
  for (int r = 0; r < UV_ROWS; ++r) {
    float u = 0.f, v = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXP; ++j) {
      if (lane + 64 * j < nch && n0 + r < N) {
        const half8_t h = wv[r][j];
        const f32x4 w0 = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]}, w1 = {(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
        const f32x4 pu = au[j][0] * w0 + au[j][1] * w1, pv = av[j][0] * w0 + av[j][1] * w1;
        u += (pu.x + pu.y) + (pu.z + pu.w);
        v += (pv.x + pv.y) + (pv.z + pv.w);
      }
    }
    u = wave_sum_dpp(u);
    v = wave_sum_dpp(v);
    if (lane == 0 && n0 + r < N) {
      ob[n0 + r] = u;
      ob[N + n0 + r] = v + bias[(long)i * N + n0 + r];
    }
  }
}

#include "attention_kernel.h"  // dit_attention_kernel<T, JQ, HD> + attention_launch

// ------------------------------------------------------------------ final layer + unpatchify + solver update
// (DiT.py:134-149,230-243,270-271; CFG combine :285-287; Euler update test_flow_latent.py:61-73 via torchdiffeq)
// out[n][c][hp+p][wp+q] = base + dt * v,  v = linear(modulate(LN(x)))[(p*P+q)*C + c].
// One wave owns FOUR token rows (under CFG: two conditional tokens and their two unconditional twins), so every row of the
// output matrix Wf is fetched once per four tokens; the 4 x 16 per-lane partial dot products are then reduced with a 6-step
// butterfly reduce-scatter (63 exchanges) that leaves lane l with the finished value of (row l>>4, output l&15).
#define FIN_MAXO 256  // outputs per token p*p*C, processed 16 per pass
template <bool CFG>
__global__ __launch_bounds__(256) void final_layer_kernel(const float* __restrict__ X, int M, int D, int tokens, const float* __restrict__ shift,
                                                          const float* __restrict__ scale, long mod_stride, const float* __restrict__ Wf,
                                                          const float* __restrict__ bf, int C, int R, int p, float cfg_scale,
                                                          float* out, const float* base, const float* __restrict__ dt_ptr) {
  const int lane = threadIdx.x & 63;
  const long wg = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int Mh = CFG ? M / 2 : M;
  const long mfirst = CFG ? wg * 2 : wg * 4;
  if (mfirst >= Mh) return;
  const int nv = D >> 2, NO = p * p * C;
  long mrow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    long m = CFG ? mfirst + (r & 1) + (long)(r >> 1) * Mh : mfirst + r;
    mrow[r] = m < M ? m : M - 1;
  }
  f32x4 v[4][LN_MAXV];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f32x4* xr = (const f32x4*)(X + mrow[r] * D);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv) v[r][i] = xr[c];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (lane + 64 * i < nv) s += v[r][i].x + v[r][i].y + v[r][i].z + v[r][i].w;
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (lane + 64 * i < nv) {
        v[r][i] -= mean;
        q += v[r][i].x * v[r][i].x + v[r][i].y * v[r][i].y + v[r][i].z * v[r][i].z + v[r][i].w * v[r][i].w;
      }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-6f);
    const long mo = (mrow[r] / tokens) * mod_stride;
    const f32x4* sh = (const f32x4*)(shift + mo);
    const f32x4* sc = (const f32x4*)(scale + mo);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv) v[r][i] = v[r][i] * rstd * (1.0f + sc[c]) + sh[c];
    }
  }
  const int r = lane >> 4, ol = lane & 15;
  const long m = CFG ? mfirst + (r & 1) + (long)(r >> 1) * Mh : mfirst + r;
  for (int o0 = 0; o0 < NO; o0 += 16) {  // 16 outputs per pass: p*p*C = 16 (patch 2) is one pass, 64 / 256 (patch 4 / 8) four / sixteen
    float part[64];  // [r][o]
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      f32x4 w4[LN_MAXV];
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        w4[i] = (o0 + o < NO && c < nv) ? ((const f32x4*)(Wf + (long)(o0 + o) * D))[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i)
          if (lane + 64 * i < nv) a += v[rr][i].x * w4[i].x + v[rr][i].y * w4[i].y + v[rr][i].z * w4[i].z + v[rr][i].w * w4[i].w;
        part[rr * 16 + o] = a;
      }
    }
    // reduce-scatter: at step s the lane bit (32 >> s) picks the upper/lower half of the remaining index range
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int half_w = 32 >> s, mask = 32 >> s;
      const bool upper = (lane & mask) != 0;
#pragma unroll
      for (int k = 0; k < half_w; ++k) {
        const float keep = upper ? part[k + half_w] : part[k];
        const float send = upper ? part[k] : part[k + half_w];
        part[k] = keep + __shfl_xor(send, mask, 64);
      }
    }
    const int o = o0 + ol;
    float val = part[0] + (o < NO ? bf[o] : 0.f);
    if (CFG) {  // rows 0,1 conditional, rows 2,3 their unconditional twins: lane ^ 32 holds the twin's value
      const float other = xhalf(val);
      const float cond = r < 2 ? val : other, uncond = r < 2 ? other : val;
      val = uncond + cfg_scale * (cond - uncond);
    }
    if (o < NO && m < M && (CFG || m < Mh)) {
      const int grid = R / p;
      const int n = (int)(m / tokens), tok = (int)(m % tokens);
      const int pp = o / (p * C), qq = (o / C) % p, c = o % C;
      const long off = (((long)n * C + c) * R + (tok / grid) * p + pp) * R + (tok % grid) * p + qq;
      if (base) out[off] = base[off] + (*dt_ptr) * val;
      else out[off] = val;
    }
  }
}

// Round 3: the same layer as a skinny MFMA GEMM.  final_layer_kernel above keeps four whole rows per wave in registers (434+ VGPRs: one wave per SIMD)
// and reduces 64 partial dot products through 63 ds_bpermute exchanges: 64 us for 67 MB = 1.0 TB/s.  Here one wave owns SIXTEEN token rows as ONE
// v_mfma_f32_16x16x32_f16 row tile and N = p*p*C / 16 column tiles; K = D is walked in 32-deep steps with the operands built in registers in the
// MFMA fragment layout (lane l: row l & 15, eight consecutive k at 8 (l >> 4)), so
//   * the LayerNorm statistics are in-lane sums plus two lane exchanges (the four lanes l, l^16, l^32, l^48 share a row);
//   * X is streamed twice (statistics, then operands): the second pass hits the L2 (64 KiB per wave), HBM sees the 67 MB once;
//   * fp32 fidelity on an fp16 matrix core: activation and weight are each split into fp16 hi + lo and three MFMAs (hi*hi + lo*hi + hi*lo)
//     accumulate in fp32 -- the dropped lo*lo term is 2^-22 relative, i.e. the result is the fp32 dot product to rounding, as before.
// Under CFG a tile holds eight conditional rows and their eight unconditional twins, which land in lanes l and l ^ 32 of the result.
// (D % 128 == 0: the four waves of a block take the 32-deep k-steps round-robin.)
template <bool CFG, int NT>
__global__ __launch_bounds__(256) void final_layer_mfma_kernel(const float* __restrict__ X, int M, int D, int tokens, const float* __restrict__ shift,
                                                               const float* __restrict__ scale, long mod_stride, const float* __restrict__ Wf,
                                                               const float* __restrict__ bf, int C, int R, int p, float cfg_scale, float* out,
                                                               const float* base, const float* __restrict__ dt_ptr) {
  // one BLOCK per 16-row tile; its four waves split K (k-steps wv, wv + 4, ...) so that sixteen waves per CU cover the memory latency, and
  // combine their partial statistics / partial accumulators through the LDS in a fixed order
  __shared__ float st_s[4][16][2];
  __shared__ float acc_s[4][NT][64][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, a = lane & 15, q = lane >> 4;
  const long tile = blockIdx.x;
  const int Mh = CFG ? M / 2 : M;
  const long m = CFG ? tile * 8 + (a & 7) + (a >> 3) * (long)Mh : tile * 16 + a;  // this lane's operand row
  const float* xr = X + m * D + 8 * q;
  const int nks = D >> 5;
  // ---- pass 1: shifted one-pass statistics (shift = the row's first element)
  const float c0 = X[m * D];
  float sx = 0.f, sq = 0.f;
#pragma unroll 4
  for (int ks = wv; ks < nks; ks += 4) {
    const f32x4 x0 = *(const f32x4*)(xr + 32 * ks) - c0, x1 = *(const f32x4*)(xr + 32 * ks + 4) - c0;
    sx += (x0.x + x0.y) + (x0.z + x0.w) + (x1.x + x1.y) + (x1.z + x1.w);
    sq += (x0.x * x0.x + x0.y * x0.y) + (x0.z * x0.z + x0.w * x0.w) + (x1.x * x1.x + x1.y * x1.y) + (x1.z * x1.z + x1.w * x1.w);
  }
  sx += __shfl_xor(sx, 16, 64);
  sq += __shfl_xor(sq, 16, 64);
  sx += __shfl_xor(sx, 32, 64);
  sq += __shfl_xor(sq, 32, 64);
  if (q == 0) {
    st_s[wv][a][0] = sx;
    st_s[wv][a][1] = sq;
  }
  __syncthreads();
  sx = (st_s[0][a][0] + st_s[1][a][0]) + (st_s[2][a][0] + st_s[3][a][0]);
  sq = (st_s[0][a][1] + st_s[1][a][1]) + (st_s[2][a][1] + st_s[3][a][1]);
  const float dl = sx / (float)D, mean = c0 + dl;
  const float rstd = rsqrtf(fmaxf(sq / (float)D - dl * dl, 0.f) + 1e-6f);
  const long mo = (m / tokens) * mod_stride + 8 * q;
  // ---- pass 2: operands + MFMAs (X again, now from the L2)
  f32x4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto split = [](const f32x4& lo4, const f32x4& hi4, half8_t& h, half8_t& l) {
    const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h[e] = (half_t)v[e];
      l[e] = (half_t)(v[e] - (float)h[e]);
    }
  };
#pragma unroll 2
  for (int ks = wv; ks < nks; ks += 4) {
    const f32x4 x0 = *(const f32x4*)(xr + 32 * ks), x1 = *(const f32x4*)(xr + 32 * ks + 4);
    const f32x4 s0 = *(const f32x4*)(scale + mo + 32 * ks), s1 = *(const f32x4*)(scale + mo + 32 * ks + 4);
    const f32x4 h0 = *(const f32x4*)(shift + mo + 32 * ks), h1 = *(const f32x4*)(shift + mo + 32 * ks + 4);
    const f32x4 a0 = (x0 - mean) * rstd * (1.0f + s0) + h0, a1 = (x1 - mean) * rstd * (1.0f + s1) + h1;
    half8_t ah, al;
    split(a0, a1, ah, al);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float* wr = Wf + (long)(t * 16 + a) * D + 32 * ks + 8 * q;  // output column t * 16 + (lane & 15), same k slice
      half8_t wh, wl;
      split(*(const f32x4*)wr, *(const f32x4*)(wr + 4), wh, wl);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) *(f32x4_t*)acc_s[wv][t][lane] = acc[t];
  __syncthreads();
  if (wv != 0) return;
  // ---- lane holds out[row 4 q + r][column o = t * 16 + (lane & 15)], r = 0..3
  const float dt = base ? *dt_ptr : 0.f;
  const int grid = R / p;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4_t tot = (*(const f32x4_t*)acc_s[0][t][lane] + *(const f32x4_t*)acc_s[1][t][lane]) +
                        (*(const f32x4_t*)acc_s[2][t][lane] + *(const f32x4_t*)acc_s[3][t][lane]);
    const int o = t * 16 + a;
    const float bo = bf[o];
    const int pp = o / (p * C), qq = (o / C) % p, c = o % C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * q + r;
      float val = tot[r] + bo;
      if (CFG) {  // rows 0..7 conditional, 8..15 their unconditional twins: lane ^ 32 holds the twin's value
        const float other = xhalf(val);
        const float cond = row < 8 ? val : other, uncond = row < 8 ? other : val;
        val = uncond + cfg_scale * (cond - uncond);
      }
      const long mr = CFG ? tile * 8 + (row & 7) + (row >> 3) * (long)Mh : tile * 16 + row;
      const int n = (int)(mr / tokens), tok = (int)(mr % tokens);
      const long off = (((long)n * C + c) * R + (tok / grid) * p + pp) * R + (tok % grid) * p + qq;
      out[off] = base ? base[off] + dt * val : val;
    }
  }
}

// ------------------------------------------------------------------ solver helpers
__global__ void grid_advance_kernel(const float* ts, const float* dts, int* step, float* t_cur, float* t_next, float* dt_cur) {
  const int s = *step;
  *t_cur = ts[s];
  *t_next = ts[s + 1];
  *dt_cur = dts[s];
  *step = s + 1;
}

struct LinPtrs {
  const float* k[8];
};
__global__ void lincomb_kernel(float* out, const float* base, LinPtrs ks, const float* __restrict__ coef, const float* __restrict__ scale,
                               int nk, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < nk; ++j) {
    const float c = coef[j];
    if (c != 0.f) acc += c * ((const f32x4*)ks.k[j])[i];
  }
  if (scale) acc *= *scale;
  if (base) acc += ((const f32x4*)base)[i];
  ((f32x4*)out)[i] = acc;
}

// Error ratio of an adaptive Runge-Kutta step as torchdiffeq takes it (whole-tensor RMS): sqrt(mean(((dt * sum_j e_j k_j) / (atol + rtol max(|y0|, |y1|)))^2)).
// Two fixed-order stages (block partials, then one block over them): deterministic, one float for the host to read.
#define RK_BLOCKS 1024
__global__ __launch_bounds__(256) void rk_err_partial_kernel(const float* __restrict__ y0, const float* __restrict__ y1, LinPtrs ks,
                                                             const float* __restrict__ coef, const float* __restrict__ dt, int nk, long n4, float rtol,
                                                             float atol, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  const float h = *dt;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < nk; ++j) {
      const float c = coef[j];
      if (c != 0.f) e += c * ((const f32x4*)ks.k[j])[i];
    }
    const f32x4 a = ((const f32x4*)y0)[i], b = ((const f32x4*)y1)[i];
    const f32x4 r = {h * e.x / (atol + rtol * fmaxf(fabsf(a.x), fabsf(b.x))), h * e.y / (atol + rtol * fmaxf(fabsf(a.y), fabsf(b.y))),
                     h * e.z / (atol + rtol * fmaxf(fabsf(a.z), fabsf(b.z))), h * e.w / (atol + rtol * fmaxf(fabsf(a.w), fabsf(b.w)))};
    acc += (r.x * r.x + r.y * r.y) + (r.z * r.z + r.w * r.w);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void rk_err_finish_kernel(const float* __restrict__ part, int nb, float inv_n, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += part[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sqrtf(((red[0] + red[1]) + (red[2] + red[3])) * inv_n);
}

// ------------------------------------------------------------------ host side
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct DitWs {
  float* X;       // [M, D] fp32 residual stream
  half_t* A;      // [M, D] LN output / attention output
  half_t* A2;     // [M, D] folded path: the proj GEMM's A' output (fc1's operand).  NOT ws.A: proj's operand IS ws.A (the attention output), and a tile of
                  // a row panel may still be streaming columns that a sibling tile's epilogue would overwrite (round 4: found by the full-size parity test)
  half_t* QKVH;   // max(3*M*D, M*H): Q | K | Vt, later the fc1 activation
  float* temb;    // [B, D]
  float* temb_h;  // [B, D] hidden layer of the t-MLP
  half_t* c_half; // [B, D]
  float* mod;     // [B, J]
  float* ones;    // [D] of 1.0f: gate row of the patch-embedding GEMM (large patches)
  // folded LayerNorm-modulate (gemm_kernel.h): row partials, two centring-constant arrays (ping-pong), the u / v GEMM's operand and results
  float* ln_part;  // [M][ceil(D / 256)][2]
  float* cen[2];   // [M] each
  half_t* amod;    // [depth][2 branches][2][rows][D] fp16: (1 + scale | shift) rows
  float* uvq;      // [depth][2 rows][3D]: u, v of the qkv projections
  float* uvf;      // [depth][2 rows][H]:  u, v of fc1
  float* slab;    // split-K partial tiles (small M only, else null)
  size_t slab_bytes;
  size_t total;
};

static DitWs carve(const lfm_dit_shape* s, int B, void* ws, bool sizing = false) {
  const size_t T = (size_t)(s->res / s->patch) * (s->res / s->patch), M = (size_t)B * T, D = s->hidden, H = s->mlp_hidden;
  const size_t J = (size_t)s->depth * 6 * D + 2 * D;
  size_t off = 0;
  char* base = (char*)ws;
  DitWs w;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  w.X = (float*)take(M * D * 4);
  w.A = (half_t*)take(M * D * 2);
  const size_t qkvh = (3 * M * D > M * H ? 3 * M * D : M * H) * 2;
  w.QKVH = (half_t*)take(qkvh);
  w.temb = (float*)take((size_t)B * D * 4);
  w.temb_h = (float*)take((size_t)B * D * 4);
  w.c_half = (half_t*)take((size_t)B * D * 2);
  w.mod = (float*)take((size_t)B * J * 4);
  w.ones = (float*)take(D * 4);
  w.ln_part = (float*)take(M * ((D + 255) / 256) * 8);
  w.cen[0] = (float*)take(M * 4);
  w.cen[1] = (float*)take(M * 4);
  // only shapes that can ever take the folded LayerNorm path (residual width a multiple of 256, H % 64 == 0: lfm_dit_forward's `fold`) pay for its operands
  const bool uv = (D % 256 == 0) && (H % 64 == 0);
  w.amod = (half_t*)take(uv ? (size_t)s->depth * 4 * B * D * 2 : 0);
  w.uvq = (float*)take(uv ? (size_t)s->depth * 2 * B * 3 * D * 4 : 0);
  w.uvf = (float*)take(uv ? (size_t)s->depth * 2 * B * H * 4 : 0);
  w.A2 = (half_t*)take(uv ? M * D * 2 : 0);
  // latency mode: room for up to 4 K slices of the widest GEMM output (fc1), when the token count is small
  // (when SIZING for a maximum batch, reserve the slabs of the largest small batch too, so that the requirement is monotone in the
  // batch and a workspace sized for max_batch serves every smaller batch)
  const size_t Ms = sizing ? (M < 1024 ? M : (size_t)1024 / T * T) : M;
  w.slab_bytes = (sizing || M <= 1024) && Ms > 0 ? 4 * Ms * (H > 3 * D ? H : 3 * D) * 4 : 0;
  w.slab = w.slab_bytes ? (float*)take(w.slab_bytes) : nullptr;
  if (!base) w.slab = nullptr;
  w.total = off;
  return w;
}

static int check_shape(const lfm_dit_shape* s) {
  if (!s) return LFM_ERR_ARG;
  if (s->depth <= 0 || s->hidden <= 0 || s->heads <= 0 || s->patch <= 0 || s->in_ch <= 0 || s->res <= 0) return LFM_ERR_SHAPE;
  if (s->hidden % s->heads) return LFM_ERR_SHAPE;
  const int hd = s->hidden / s->heads;
  if (hd != 64 && hd != 72) return LFM_ERR_SHAPE;  // S / B / L: 64; XL: 1152 / 16 = 72
  if (s->res % s->patch) return LFM_ERR_SHAPE;
  const int T = (s->res / s->patch) * (s->res / s->patch);
  if (T != 16 && T != 64 && T != 128 && T != 256 && T != 1024) return LFM_ERR_SHAPE;  // attention kernels: LDS-resident K / V^T up to 256 tokens; 1024 = four key chunks
  if (s->hidden % 64 || s->hidden > 256 * LN_MAXV || s->mlp_hidden % 64) return LFM_ERR_SHAPE;
  const int kk = s->patch * s->patch * s->in_ch;
  if (kk > FIN_MAXO || (kk > PE_MAXK && (kk % 64))) return LFM_ERR_SHAPE;  // small patches: register kernel; large: GEMM (K % 64 == 0)
  if (s->label_rows <= 0) return LFM_ERR_SHAPE;
  return LFM_OK;
}

extern "C" const char* lfm_strerror(int code) {
  switch (code) {
    case LFM_OK: return "ok";
    case LFM_ERR_SHAPE: return "unsupported or inconsistent shape";
    case LFM_ERR_ALIGN: return "pointer / leading dimension alignment";
    case LFM_ERR_WORKSPACE: return "workspace too small";
    case LFM_ERR_LAUNCH: return "kernel launch failed";
    case LFM_ERR_ARG: return "bad argument";
  }
  return "unknown";
}
extern "C" int lfm_abi_version(void) { return LFM_ABI_VERSION; }  // 2: lfm_time_embed takes label_rows; 3: lfm_dit_call carries the per-grid conditioning table; 4: + cond_rows, per-call fold_ln / gemm_select

extern "C" size_t lfm_dit_workspace_bytes(const lfm_dit_shape* shape, int max_batch) {
  if (check_shape(shape) != LFM_OK || max_batch <= 0) return 0;
  return carve(shape, max_batch, nullptr, true).total;
}

extern "C" int lfm_dit_attention_hd(const void* Q, const void* K, const void* Vt, void* O, int batch, int heads, int head_dim, int T,
                                    lfm_stream_t stream) {
  if (!Q || !K || !Vt || !O) return LFM_ERR_ARG;
  if (batch <= 0 || heads <= 0) return LFM_ERR_SHAPE;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)O) & 15) return LFM_ERR_ALIGN;
  return attention_launch((const half_t*)Q, (const half_t*)K, (const half_t*)Vt, (half_t*)O, batch, heads, head_dim, T, (hipStream_t)stream);
}
extern "C" int lfm_dit_attention(const void* Q, const void* K, const void* Vt, void* O, int batch, int heads, int T, lfm_stream_t stream) {
  return lfm_dit_attention_hd(Q, K, Vt, O, batch, heads, 64, T, stream);
}

static int ln_modulate_launch(const float* X, half_t* A, int M, int D, int tokens, const float* shift, const float* scale, long stride,
                              hipStream_t st) {
  if (D % 4 || D > 256 * LN_MAXV) return LFM_ERR_SHAPE;
  if (D % 8 == 0 && !(((uintptr_t)A | (uintptr_t)X) & 15) && !(lfm_gemm_debug_flags() & 32768)) {  // flag 32768: the 8-byte-store kernel (A/B)
    // ONE row per wave and the two row sums on the DPP network (r02_probe3: 17.0-17.2 us = 5.9 TB/s at M 16384 x D 1024; two rows per wave with
    // ds_bpermute sums -- the round-1 choice -- 20.4-20.9 us, one row with ds_bpermute 17.5-17.8 us, four rows 23.3-24.0 us).  A/B flags: 65536 =
    // one row + ds_bpermute sums, 524288 = two rows, 262144 = four rows.
    const int f = lfm_gemm_debug_flags();
    if (f & 65536) hipLaunchKernelGGL(ln_modulate8_kernel<1>, dim3(cdiv(M, 4)), dim3(256), 0, st, X, A, M, D, tokens, shift, scale, stride);
    else if (f & 524288) hipLaunchKernelGGL(ln_modulate8_kernel<2>, dim3(cdiv(M, 8)), dim3(256), 0, st, X, A, M, D, tokens, shift, scale, stride);
    else if (f & 262144) hipLaunchKernelGGL(ln_modulate8_kernel<4>, dim3(cdiv(M, 16)), dim3(256), 0, st, X, A, M, D, tokens, shift, scale, stride);
    else hipLaunchKernelGGL((ln_modulate8_kernel<1, true>), dim3(cdiv(M, 4)), dim3(256), 0, st, X, A, M, D, tokens, shift, scale, stride);
    LFM_CHECK_LAUNCH();
    return LFM_OK;
  }
  hipLaunchKernelGGL(ln_modulate_kernel, dim3(cdiv(M, 4 * LN_ROWS)), dim3(256), 0, st, X, A, M, D, tokens, shift, scale, stride);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

extern "C" int lfm_ln_modulate(const float* X, void* A, int M, int D, int tokens, const float* shift, const float* scale, long mod_stride,
                               lfm_stream_t stream) {
  if (!X || !A || !shift || !scale) return LFM_ERR_ARG;
  if (M <= 0 || tokens <= 0) return LFM_ERR_SHAPE;
  return ln_modulate_launch(X, (half_t*)A, M, D, tokens, shift, scale, mod_stride, (hipStream_t)stream);
}

extern "C" int lfm_gemm_f16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias,
                            int epilogue, const float* gate, long gate_stride, int tokens, lfm_stream_t stream) {
  if (!A || !W || !C) return LFM_ERR_ARG;
  if ((lda % 8) || ((uintptr_t)A & 15)) return LFM_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  ASrcRowMajor a{(const half_t*)A, lda, M, 0};
  if (g_gemm_sel == 7 || g_gemm_sel == 8) {  // the latency-mode kernels on their own (parity tests): 7 = 64x64 tiles, 8 = all rows x 16 columns; <= 256 rows
    const int which = g_gemm_sel;
    auto lat = [&](const auto& e) {
      return which == 7 ? launch_gemm_sq64((const half_t*)A, lda, (const half_t*)W, ldw, M, N, K, e, 1, st)
                        : launch_gemm_skinny((const half_t*)A, lda, (const half_t*)W, ldw, M, N, K, e, 1, st);
    };
    switch (epilogue) {
      case 0: return lat(EpiBiasF16{(half_t*)C, ldc, bias});
      case 1: return bias ? lat(EpiBiasGeluF16{(half_t*)C, ldc, bias}) : LFM_ERR_ARG;
      case 2: return lat(EpiBiasF32{(float*)C, ldc, bias});
      default: return LFM_ERR_ARG;
    }
  }
  switch (epilogue) {
    case 0:
#ifdef LFM_MEASURE
      if (g_gemm_sel == 5 && (g_gemm_dbg & 2) && K % G256Q_BK == 0)  // measurement: the epilogue-stamped build of the 16x16x32 kernel
        return launch_gemm256h_tn<ASrcRowMajor, EpiBiasF16, true>(a, (const half_t*)W, ldw, M, N, K, EpiBiasF16{(half_t*)C, ldc, bias}, st);
#endif
      return launch_gemm_auto(a, (const half_t*)W, ldw, M, N, K, EpiBiasF16{(half_t*)C, ldc, bias}, st);
    case 1:
      if (!bias) return LFM_ERR_ARG;
#ifdef LFM_MEASURE
      if (g_gemm_sel == 5 && (g_gemm_dbg & 2) && K % G256Q_BK == 0)  // measurement: the epilogue-stamped build of the 16x16x32 kernel
        return launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, true>(a, (const half_t*)W, ldw, M, N, K, EpiBiasGeluF16{(half_t*)C, ldc, bias}, st);
      if (g_gemm_sel == 5 && ((g_gemm_dbg >> 21) & 7) && K % G256Q_BK == 0) {  // measurement: main-loop ablations (flags 1..7 << 21)
        const EpiBiasGeluF16 e{(half_t*)C, ldc, bias};
        switch ((g_gemm_dbg >> 21) & 7) {
          case 1: return launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, false, 1>(a, (const half_t*)W, ldw, M, N, K, e, st);
          case 2: return launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, false, 2>(a, (const half_t*)W, ldw, M, N, K, e, st);
          case 3: return launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, false, 3>(a, (const half_t*)W, ldw, M, N, K, e, st);
          case 4: return launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, false, 4>(a, (const half_t*)W, ldw, M, N, K, e, st);
          case 5: return launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, false, 5>(a, (const half_t*)W, ldw, M, N, K, e, st);
          case 6: return launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, false, 6>(a, (const half_t*)W, ldw, M, N, K, e, st);
          default: return (g_gemm_dbg & (1 << 24)) ? launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, false, 8>(a, (const half_t*)W, ldw, M, N, K, e, st)
                                                   : launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, false, 7>(a, (const half_t*)W, ldw, M, N, K, e, st);
        }
      }
      if ((g_gemm_sel == 5 || g_gemm_sel == 6) && ((g_gemm_dbg >> 25) & 3) && K % G256Q_BK == 0) {  // measurement: OPT variants (1..3 << 25)
        const EpiBiasGeluF16 e{(half_t*)C, ldc, bias};
        const int o = (g_gemm_dbg >> 25) & 3;
        if (g_gemm_sel == 6) return launch_gemm256w_tn<ASrcRowMajor, EpiBiasGeluF16, 0, 0, 1>(a, (const half_t*)W, ldw, M, N, K, e, st);
        if (o == 1) return launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, false, 0, 1>(a, (const half_t*)W, ldw, M, N, K, e, st);
        if (o == 2) return launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, false, 0, 2>(a, (const half_t*)W, ldw, M, N, K, e, st);
        return launch_gemm256h_tn<ASrcRowMajor, EpiBiasGeluF16, false, 0, 3>(a, (const half_t*)W, ldw, M, N, K, e, st);
      }
      if (g_gemm_sel == 6 && ((g_gemm_dbg >> 21) & 15) && K % G256Q_BK == 0) {  // measurement: v6 main-loop ablations (1..3 << 21) and DMA placement (8 << 21)
        const EpiBiasGeluF16 e{(half_t*)C, ldc, bias};
        switch ((g_gemm_dbg >> 21) & 15) {
          case 1: return launch_gemm256w_tn<ASrcRowMajor, EpiBiasGeluF16, 1>(a, (const half_t*)W, ldw, M, N, K, e, st);
          case 2: return launch_gemm256w_tn<ASrcRowMajor, EpiBiasGeluF16, 2>(a, (const half_t*)W, ldw, M, N, K, e, st);
          case 3: return launch_gemm256w_tn<ASrcRowMajor, EpiBiasGeluF16, 3>(a, (const half_t*)W, ldw, M, N, K, e, st);
          case 4: return launch_gemm256w_tn<ASrcRowMajor, EpiBiasGeluF16, 4>(a, (const half_t*)W, ldw, M, N, K, e, st);
          case 8: return launch_gemm256w_tn<ASrcRowMajor, EpiBiasGeluF16, 0, 1>(a, (const half_t*)W, ldw, M, N, K, e, st);
          default: return LFM_ERR_ARG;
        }
      }
#endif
      return launch_gemm_auto(a, (const half_t*)W, ldw, M, N, K, EpiBiasGeluF16{(half_t*)C, ldc, bias}, st);
    case 2: return launch_gemm_auto(a, (const half_t*)W, ldw, M, N, K, EpiBiasF32{(float*)C, ldc, bias}, st);
    case 3:
      if (!bias || !gate || tokens <= 0) return LFM_ERR_ARG;
#ifdef LFM_MEASURE
      if (g_gemm_sel == 5 && (g_gemm_dbg & 2) && K % G256Q_BK == 0)
        return launch_gemm256h_tn<ASrcRowMajor, EpiGateResidF32, true>(a, (const half_t*)W, ldw, M, N, K,
                                                                       EpiGateResidF32{(float*)C, ldc, bias, gate, gate_stride, tokens}, st);
#endif
      return launch_gemm_auto(a, (const half_t*)W, ldw, M, N, K, EpiGateResidF32{(float*)C, ldc, bias, gate, gate_stride, tokens}, st);
  }
  return LFM_ERR_ARG;
}

// ------------------------------------------------------------------ in-situ timing of the dominant kernel (measurement only)
// bench.py's roofline row needs the fc1 GEMM's duration INSIDE a real forward (real activations, real cache state); the
// captured graph cannot be bracketed from outside, so an eager forward can record one HIP event pair per block here.
#define LFM_PROF_MAX 64
#define LFM_PROF_BLK_MAX 16
static hipEvent_t g_prof_ev[2 * LFM_PROF_MAX];
static bool g_prof_init = false, g_prof_on = false;
static int g_prof_count = 0;
static hipEvent_t g_prof_blk_ev[2 * LFM_PROF_BLK_MAX];  // around the whole block loop of an evaluation (all blocks' qkv .. fc2)
static int g_prof_blk_count = 0;
static int g_prof_mode = 0;  // 1: an event pair around every fc1 launch (+ the block loop); 2: around the block loop only (no events between the kernels)
static hipStream_t g_prof_stream = nullptr;  // the stream that owns the probe (the first one that launches while it is on)
static bool g_prof_owned = false, g_prof_conflict = false;
// The probe is a measurement device for ONE stream driven by ONE host thread (bench.py).  Switching it, claiming it and reading it are serialised by a mutex, so
// that a second host thread enqueueing evaluations at the same time gets a clean refusal (g_prof_conflict) instead of a data race on the flags; the sample
// counters are touched only by the thread whose stream owns the probe (prof_claim returned true for it).
static std::mutex g_prof_mu;
extern "C" int lfm_profile_fc1(int enable) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (enable && !g_prof_init) {
    for (int i = 0; i < 2 * LFM_PROF_MAX; ++i)
      if (hipEventCreate(&g_prof_ev[i]) != hipSuccess) return LFM_ERR_LAUNCH;
    for (int i = 0; i < 2 * LFM_PROF_BLK_MAX; ++i)
      if (hipEventCreate(&g_prof_blk_ev[i]) != hipSuccess) return LFM_ERR_LAUNCH;
    g_prof_init = true;
  }
  g_prof_on = enable != 0;
  g_prof_mode = enable;
  if (enable) {
    g_prof_count = 0;
    g_prof_blk_count = 0;
    g_prof_owned = false;
    g_prof_conflict = false;
  }
  return LFM_OK;
}
static bool prof_claim(hipStream_t st) {  // may THIS evaluation record its fc1 launches?
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_on) return false;
  if (!g_prof_owned) {
    g_prof_owned = true;
    g_prof_stream = st;
  }
  if (st != g_prof_stream) {
    g_prof_conflict = true;
    return false;
  }
  return true;
}
extern "C" int lfm_profile_fc1_read(float* ms_out, int max_n) {  // synchronises; returns the number of samples written
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!ms_out || !g_prof_init) return LFM_ERR_ARG;
  if (g_prof_conflict) return LFM_ERR_ARG;  // a second stream launched evaluations while the probe was on: the samples would time its kernels too
  const int n = g_prof_count < max_n ? g_prof_count : max_n;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof_ev[2 * i + 1]) != hipSuccess) return LFM_ERR_LAUNCH;
    if (hipEventElapsedTime(&ms_out[i], g_prof_ev[2 * i], g_prof_ev[2 * i + 1]) != hipSuccess) return LFM_ERR_LAUNCH;
  }
  return n;
}

extern "C" int lfm_profile_blocks_read(float* ms_out, int max_n) {  // one sample per recorded evaluation: its whole block loop; synchronises
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!ms_out || !g_prof_init) return LFM_ERR_ARG;
  if (g_prof_conflict) return LFM_ERR_ARG;
  const int n = g_prof_blk_count < max_n ? g_prof_blk_count : max_n;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof_blk_ev[2 * i + 1]) != hipSuccess) return LFM_ERR_LAUNCH;
    if (hipEventElapsedTime(&ms_out[i], g_prof_blk_ev[2 * i], g_prof_blk_ev[2 * i + 1]) != hipSuccess) return LFM_ERR_LAUNCH;
  }
  return n;
}

extern "C" int lfm_gemm_qkv_f16(const void* A, long lda, const void* W, long ldw, void* Q, void* Kout, void* Vt, int M, int D, int K,
                                const float* bias, int head_dim, int tokens, lfm_stream_t stream) {
  if (!A || !W || !Q || !Kout || !Vt || !bias) return LFM_ERR_ARG;
  if ((lda % 8) || ((uintptr_t)A & 15)) return LFM_ERR_ALIGN;
  if (head_dim <= 0 || tokens <= 0 || (D % head_dim) || (head_dim % 8) || (tokens % 16) || (M % tokens)) return LFM_ERR_SHAPE;  // 16: the V^T token groups (vt_pos)
#ifdef LFM_MEASURE
  if (g_gemm_sel == 5 && (g_gemm_dbg & 2) && K % G256Q_BK == 0)
    return launch_gemm256h_tn<ASrcRowMajor, EpiQKV, true>(ASrcRowMajor{(const half_t*)A, lda, M, 0}, (const half_t*)W, ldw, M, 3 * D, K,
                                                          EpiQKV::make((half_t*)Q, (half_t*)Kout, (half_t*)Vt, bias, D, head_dim, tokens),
                                                          (hipStream_t)stream);
#endif
  return launch_gemm_auto(ASrcRowMajor{(const half_t*)A, lda, M, 0}, (const half_t*)W, ldw, M, 3 * D, K,
                          EpiQKV::make((half_t*)Q, (half_t*)Kout, (half_t*)Vt, bias, D, head_dim, tokens), (hipStream_t)stream);
}

#ifdef LFM_MEASURE  // s_memtime trace readers: measurement builds only (include/lfm_hip.h)
extern "C" int lfm_gemm_trace_read(unsigned long long* host_out, int n_per_group) {  // 2 x n stamps (group 0, group 1)
  if (!host_out || n_per_group <= 0 || n_per_group > G256Q_TRACE_MAX) return LFM_ERR_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return LFM_ERR_LAUNCH;
  for (int g = 0; g < 2; ++g)
    if (hipMemcpyFromSymbol(host_out + (size_t)g * n_per_group, HIP_SYMBOL(g256q_trace), sizeof(unsigned long long) * n_per_group,
                            sizeof(unsigned long long) * G256Q_TRACE_MAX * g, hipMemcpyDeviceToHost) != hipSuccess)
      return LFM_ERR_LAUNCH;
  return LFM_OK;
}

extern "C" int lfm_attention_trace_read(unsigned long long* host_out, int n) {  // the s_memtime stamps of the MODE 3 attention build
  if (!host_out || n <= 0 || n > ATT_TRACE_SLOTS) return LFM_ERR_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return LFM_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(att_trace), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost) != hipSuccess) return LFM_ERR_LAUNCH;
  return LFM_OK;
}

extern "C" int lfm_attention_wg_trace_read(unsigned long long* host_out, int n_wg) {  // MODE 3: {hw id, start, landed, end} per workgroup
  if (!host_out || n_wg <= 0 || n_wg > ATT_WG_TRACE) return LFM_ERR_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return LFM_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(att_wg_trace), sizeof(unsigned long long) * 4 * n_wg, 0, hipMemcpyDeviceToHost) != hipSuccess) return LFM_ERR_LAUNCH;
  return LFM_OK;
}
// Per-kernel checksums of one armed evaluation (tools/concurrency_ws_diff.py): after every kernel of the folded block loop the 64-bit wrap-around sum of
// its output buffer's 32-bit words (integer adds: order-independent, so equal data <=> equal sum whatever the reduction order) goes into a slot
// [block][8]: 0 Q|K|V^T after qkv, 1 O after attention, 2 X / 3 A' (A2) / 4 row partials after proj, 5 H after fc1, 6 X / 7 A' (A) after fc2.
#define DIT_CHK_SLOTS (64 * 8)
static unsigned long long* g_chk = nullptr;
static const void* g_chk_ws = nullptr;
__global__ __launch_bounds__(256) void chk_kernel(const unsigned* __restrict__ p, long nwords, unsigned long long* __restrict__ out) {
  unsigned long long s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (long)gridDim.x * 256) s += p[i];
  atomicAdd(out, s);
}
static void dit_chk(const void* buf, size_t bytes, int block, int slot, hipStream_t st) {
  if (!g_chk || block >= 64) return;
  hipLaunchKernelGGL(chk_kernel, dim3(1024), dim3(256), 0, st, (const unsigned*)buf, (long)(bytes / 4), g_chk + block * 8 + slot);
}
extern "C" int lfm_dit_chk_arm(const void* workspace) {  // the evaluations that run on THIS workspace record their checksums (nullptr: off)
  if (!g_chk && hipMalloc((void**)&g_chk, DIT_CHK_SLOTS * 8) != hipSuccess) return LFM_ERR_LAUNCH;
  g_chk_ws = workspace;
  return LFM_OK;
}
extern "C" int lfm_dit_chk_read(unsigned long long* host_out, int n) {
  if (!host_out || n <= 0 || n > DIT_CHK_SLOTS || !g_chk) return LFM_ERR_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return LFM_ERR_LAUNCH;
  if (hipMemcpy(host_out, g_chk, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return LFM_ERR_LAUNCH;
  return LFM_OK;
}
#ifdef LFM_EXP_DUMP
// (experiment build, tools/cosched_dump.py) the folded fc1 epilogue of every block of the evaluations on THIS workspace dumps the operands of its affine
static float* g_dbg = nullptr;
static long g_dbg_stride = 0;
static const void* g_dbg_ws = nullptr;
extern "C" int lfm_dit_dbg_arm(const void* workspace, float* dump, long stride_floats) {
  g_dbg = dump;
  g_dbg_stride = stride_floats;
  g_dbg_ws = workspace;
  return LFM_OK;
}
#endif
#endif  // LFM_MEASURE

// ------------------------------------------------------------------ conditioning (everything the forward derives from t and y alone)
// c = t_emb(t) (+ y_emb), the adaLN modulation rows of every block and of the final layer (DiT.py:252-262, 128, 170) and -- for the folded
// LayerNorm path -- the u / v rows of the qkv and fc1 projections (gemm_kernel.h).  One function, so that the per-grid tables below are written by
// exactly the launches a forward would make.
static int dit_conditioning(const lfm_dit_shape* s, const lfm_dit_weights* w, const DitWs& ws, const float* t, int t_len, const int64_t* y, int rows,
                            bool want_uv, hipStream_t st) {
  const int D = s->hidden, H = s->mlp_hidden;
  const long J = (long)s->depth * 6 * D + 2 * D;
  const long mstride = rows == 1 ? 0 : J;
  hipLaunchKernelGGL(temb1_kernel, dim3(cdiv(D, 4), t_len), dim3(256), 0, st, t, w->t_w0, w->t_b0, ws.temb_h, D);
  LFM_CHECK_LAUNCH();
  hipLaunchKernelGGL(temb2_kernel, dim3(cdiv(D, 4), t_len), dim3(256), 0, st, ws.temb_h, w->t_w2, w->t_b2, ws.temb, D);
  LFM_CHECK_LAUNCH();
  hipLaunchKernelGGL(cond_kernel, dim3(cdiv((long)rows * D, 256)), dim3(256), 0, st, ws.temb, t_len, w->y_table, y, s->label_rows, ws.c_half, D, rows);
  LFM_CHECK_LAUNCH();
  int rc = launch_gemm_tn(ASrcRowMajor{ws.c_half, D, rows, 0}, (const half_t*)w->ada_w, D, rows, (int)J, D, EpiBiasF32{ws.mod, J, w->ada_b}, st);
  if (rc || !want_uv) return rc;
  if (rows == 1 && D <= 8 * 64 * LN_MAXP) {  // one shared conditioning row: weight-streaming GEMVs (u, v of every block)
    hipLaunchKernelGGL(uv_gemv_kernel, dim3(cdiv(3 * D, 4 * UV_ROWS), s->depth), dim3(256), 0, st, (const half_t*)w->qkv_w, w->qkv_b, ws.mod, 3 * D, D, D, 0,
                       ws.uvq);
    LFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(uv_gemv_kernel, dim3(cdiv(H, 4 * UV_ROWS), s->depth), dim3(256), 0, st, (const half_t*)w->fc1_w, w->fc1_b, ws.mod, H, D, 4 * D, 3 * D,
                       ws.uvf);
    LFM_CHECK_LAUNCH();
    return LFM_OK;
  }
  const long nmod = (long)s->depth * 4 * rows * (D / 4);
  hipLaunchKernelGGL(mod_rows_f16_kernel, dim3(cdiv(nmod, 256)), dim3(256), 0, st, ws.mod, mstride, s->depth, rows, D, ws.amod);
  LFM_CHECK_LAUNCH();
  // u, v of every block in two batched GEMMs (batch = depth): [2 rows x D] x [D x 3D] and [2 rows x D] x [D x H]
  rc = launch_gemm_auto(ASrcRowMajor{ws.amod, D, 2 * rows, 0}, (const half_t*)w->qkv_w, D, 2 * rows, 3 * D, D, EpiUV{ws.uvq, 3L * D, w->qkv_b, rows, 3L * D}, st,
                        s->depth, 4L * rows * D, 3L * D * D, 2L * rows * 3 * D);
  if (rc) return rc;
  return launch_gemm_auto(ASrcRowMajor{ws.amod + 2L * rows * D, D, 2 * rows, 0}, (const half_t*)w->fc1_w, D, 2 * rows, H, D,
                          EpiUV{ws.uvf, (long)H, w->fc1_b, rows, (long)H}, st, s->depth, 4L * rows * D, (long)H * D, 2L * rows * H);
}

// Per-grid conditioning tables (round 4).  For the unconditional models (one shared conditioning row: scalar t, no labels -- celeb256 / ffhq / bed /
// church_dit.txt) the conditioning is a pure function of the grid time, yet every evaluation re-streamed the adaLN table (304 MB of weights for
// DiT-L/2) and both u / v GEMVs: ~168 us = 1.6 % of an evaluation (profiles/r03_final_bench_kernel_stats.csv).  A table row holds, for one time,
// [mod: J floats | uvq: depth * 2 * 3D | uvf: depth * 2 * H]; rows are written by dit_conditioning itself (bit-identical to the per-evaluation path)
// and an evaluation copies its row into the workspace (one ~2 MB copy launch) -- the row index is read on the device, so one captured graph serves
// every interval of the grid.
static inline bool cond_uv_shape(const lfm_dit_shape* s) { return (s->hidden % 256) == 0 && (s->mlp_hidden % 64) == 0; }
static inline long cond_row_floats(const lfm_dit_shape* s) {
  const long D = s->hidden, H = s->mlp_hidden, J = (long)s->depth * 6 * D + 2 * D;
  return J + (cond_uv_shape(s) ? (long)s->depth * 2 * (3 * D + H) : 0);
}
__global__ __launch_bounds__(256) void cond_row_copy_kernel(const float* __restrict__ table, long row_floats, const int* __restrict__ step, int offset,
                                                            int fixed_row, float* __restrict__ mod, long nmod, float* __restrict__ uvq, long nq,
                                                            float* __restrict__ uvf, long nf, int to_table, int rows) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= nmod + nq + nf) return;
  const long row = step ? (long)(*step + offset) : (long)fixed_row;
  float* ws = i < nmod ? mod + i : (i < nmod + nq ? uvq + (i - nmod) : uvf + (i - nmod - nq));  // nmod, nq, nf are multiples of 4
  if (rows > 0 && (row < 0 || row >= rows)) {  // a row the table does not have (wrong offset / counter): poison instead of reading out of bounds
    if (!to_table) *(f32x4*)ws = (f32x4){__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
    return;
  }
  float* tb = (float*)table + row * row_floats + i;
  if (to_table) *(f32x4*)tb = *(const f32x4*)ws;
  else *(f32x4*)ws = *(const f32x4*)tb;
}
static int dit_cond_copy(const lfm_dit_shape* s, const DitWs& ws, const float* table, const int* step, int offset, int fixed_row, int to_table,
                         hipStream_t st, int rows = 0) {
  const long D = s->hidden, H = s->mlp_hidden, J = (long)s->depth * 6 * D + 2 * D;
  const long nq = cond_uv_shape(s) ? (long)s->depth * 2 * 3 * D : 0, nf = cond_uv_shape(s) ? (long)s->depth * 2 * H : 0;
  hipLaunchKernelGGL(cond_row_copy_kernel, dim3(cdiv((J + nq + nf) / 4, 256)), dim3(256), 0, st, table, cond_row_floats(s), step, offset, fixed_row, ws.mod, J,
                     ws.uvq, nq, ws.uvf, nf, to_table, rows);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
static int dit_cond_select(const lfm_dit_shape* s, const DitWs& ws, const float* table, const int* step, int offset, int rows, hipStream_t st) {
  return dit_cond_copy(s, ws, table, step, offset, 0, 0, st, rows);
}

extern "C" size_t lfm_dit_cond_table_bytes(const lfm_dit_shape* shape, int n_times) {
  if (check_shape(shape) != LFM_OK || n_times <= 0) return 0;
  return (size_t)cond_row_floats(shape) * 4 * (size_t)n_times;
}

extern "C" int lfm_dit_cond_table_build(const lfm_dit_shape* s, const lfm_dit_weights* w, void* workspace, size_t workspace_bytes, int batch,
                                        const float* t_values, int n_times, void* table, size_t table_bytes, lfm_stream_t stream) {
  int rc = check_shape(s);
  if (rc) return rc;
  if (!w || !workspace || !t_values || !table || n_times <= 0 || batch <= 0) return LFM_ERR_ARG;
  if (table_bytes < lfm_dit_cond_table_bytes(s, n_times)) return LFM_ERR_WORKSPACE;
  const DitWs ws = carve(s, batch, workspace);  // the buffers an evaluation at this batch would use (the conditioning part does not depend on it)
  if (ws.total > workspace_bytes) return LFM_ERR_WORKSPACE;
  if (((uintptr_t)workspace & 255) || ((uintptr_t)table & 15)) return LFM_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n_times; ++i) {
    rc = dit_conditioning(s, w, ws, t_values + i, 1, nullptr, 1, cond_uv_shape(s), st);
    if (rc) return rc;
    rc = dit_cond_copy(s, ws, (const float*)table, nullptr, 0, i, 1, st, n_times);
    if (rc) return rc;
  }
  return LFM_OK;
}

// Which form of the block loop an evaluation takes (read on the calling thread, inside the call's scope): the folded LayerNorm-modulate path, its GEMMs on the
// one-wave-per-SIMD kernel, the QKV projection + attention core as one kernel.  Shared by lfm_dit_forward and lfm_dit_plan.
struct DitPlan {
  bool fold, w6, fused;
};
static DitPlan dit_plan(const lfm_dit_shape* s, int B, int rows) {
  const int D = s->hidden, H = s->mlp_hidden, grid = s->res / s->patch, T = grid * grid, M = B * T;
  const int tiles_p = D / 256;
  DitPlan p;
  // folded LayerNorm-modulate: only where its preconditions hold -- whole 256-row tiles of ONE image each (or one shared modulation row), row partials in D / 256
  // slots, all four GEMMs chip-filling on the 16x16x32 kernel
  p.fold = g_opt_fold_ln && (D % 256 == 0) && (M % 256 == 0) && (rows == 1 || T % 256 == 0) && (long)(M / 256) * tiles_p >= 192 &&
           (g_gemm_sel == 0 || g_gemm_sel == 6) && (H % 64 == 0) && s->depth >= 1;
  p.w6 = g_gemm_sel == 6 || (g_gemm_sel == 0 && g_opt_v6);  // the block GEMMs of the folded path on the one-wave-per-SIMD kernel
  // QKV projection + attention in one kernel (qkv_attention_kernel.h): one (image, head) per work item -- images of exactly one 256-token tile, head_dim 64,
  // operands inside the unsigned 32-bit byte offsets of its LDS-DMAs
  p.fused = p.fold && g_opt_fused_qkv && !p.w6 && T == 256 && D == s->heads * 64 && (long)M * D < (1L << 31) && (long)3 * D * D < (1L << 31);
  return p;
}
// The plan of an evaluation without launching it (tests; a caller that sizes its expectations): *plan_out = LFM_PLAN_* bits.  Same scope code as the forward.
extern "C" int lfm_dit_plan(const lfm_dit_shape* s, const lfm_dit_call* c, int* plan_out) {
  int rc = check_shape(s);
  if (rc) return rc;
  if (!c || !plan_out) return LFM_ERR_ARG;
  const int B = c->batch;
  if (B <= 0 || (c->t_len != 1 && c->t_len != B)) return LFM_ERR_SHAPE;
  CallScope scope;
  if ((rc = call_scope_enter(c)) != LFM_OK) return rc;
  const DitPlan p = dit_plan(s, B, (c->t_len == 1 && !c->y) ? 1 : B);
  *plan_out = (p.fold ? LFM_PLAN_FOLDED_LN : 0) | (p.fused ? LFM_PLAN_FUSED_QKV_ATTENTION : 0);
  return LFM_OK;
}

extern "C" int lfm_dit_forward(const lfm_dit_shape* s, const lfm_dit_weights* w, void* workspace, size_t workspace_bytes,
                               const lfm_dit_call* c, lfm_stream_t stream) {
  int rc = check_shape(s);
  if (rc) return rc;
  if (!w || !workspace || !c || !c->x || !c->t || !c->out) return LFM_ERR_ARG;
  const int B = c->batch;
  if (B <= 0 || (c->t_len != 1 && c->t_len != B)) return LFM_ERR_SHAPE;
  const bool cfg = c->cfg != 0;
  if (cfg && (B & 1)) return LFM_ERR_SHAPE;
  if (c->axpy_base && !c->axpy_dt) return LFM_ERR_ARG;
  const DitWs ws = carve(s, B, workspace);
  if (ws.total > workspace_bytes) return LFM_ERR_WORKSPACE;
  if ((uintptr_t)workspace & 255) return LFM_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int D = s->hidden, H = s->mlp_hidden, grid = s->res / s->patch, T = grid * grid, M = B * T;
  const long J = (long)s->depth * 6 * D + 2 * D;
  // per-call settings (ABI 4): thread-local for the duration of this call, the library defaults are not touched
  CallScope scope;
  if ((rc = call_scope_enter(c)) != LFM_OK) return rc;
  const bool prof_ok = prof_claim(st);

  // conditioning rows: one shared row when time is scalar and there are no labels
  const int rows = (c->t_len == 1 && !c->y) ? 1 : B;
  const long mstride = rows == 1 ? 0 : J;
  const bool tab = c->cond_table != nullptr;  // everything derived from t alone comes from a per-grid table (lfm_dit_cond_table_build)
  if (tab && (rows != 1 || !c->cond_step)) return LFM_ERR_ARG;

  const int KK = s->in_ch * s->patch * s->patch;
  // folded LayerNorm-modulate: decided here because the */2 patch embedding can already play the first producer (see below)
  const int tiles_p = D / 256;
  const DitPlan plan = dit_plan(s, B, rows);
  const bool fold = plan.fold, w6 = plan.w6;
  const bool pe_mfma = s->patch == 2 && s->in_ch == 4 && (D % 256 == 0) && D <= 1280 && (s->res % 2 == 0) && !(g_gemm_dbg & 2097152);  // flag: round-1 kernel
  if (tab) {
    rc = dit_cond_select(s, ws, (const float*)c->cond_table, c->cond_step, c->cond_offset, c->cond_rows, st);
  } else {
    rc = dit_conditioning(s, w, ws, c->t, c->t_len, c->y, rows, fold, st);
  }
  if (rc) return rc;
  if (pe_mfma) {
    const int tpb = M <= 1024 ? 1 : 2;  // latency mode: twice the blocks (16 for one image), one 16-token tile each behind the weight-fragment prologue
    hipLaunchKernelGGL(patch_embed_ln_kernel, dim3(cdiv(M, 16 * tpb)), dim3(64 * (D / 256)), 0, st, c->x, w->patch_w, w->patch_b, w->pos_embed, ws.X,
                       M, cfg ? B / 2 : B, s->res, D, tpb, fold ? ws.A : (half_t*)nullptr, ws.mod + D, mstride, ws.ln_part, tiles_p, ws.cen[0]);
    LFM_CHECK_LAUNCH();
  } else if (KK <= PE_MAXK) {
    hipLaunchKernelGGL(patch_embed_kernel, dim3(cdiv(M, PE_TOK)), dim3(D / 4), 0, st, c->x, w->patch_w, w->patch_b, w->pos_embed, ws.X, M,
                       cfg ? B / 2 : B, s->in_ch, s->res, s->patch, D);
    LFM_CHECK_LAUNCH();
  } else {  // DiT-*/4, */8: patches -> fp16 A operand (in the LN buffer, free at this point), X = pos_embed, then X += patches W^T + bias on MFMA
    if (!w->patch_w16) return LFM_ERR_ARG;
    hipLaunchKernelGGL(patchify_kernel, dim3(M), dim3(256), 0, st, c->x, w->pos_embed, ws.A, ws.X, ws.ones, M, cfg ? B / 2 : B, s->in_ch, s->res,
                       s->patch, D);
    LFM_CHECK_LAUNCH();
    rc = launch_gemm_auto(ASrcRowMajor{ws.A, KK, M, 0}, (const half_t*)w->patch_w16, KK, M, D, KK,
                          EpiGateResidF32{ws.X, D, w->patch_b, ws.ones, 0, T}, st);
    if (rc) return rc;
  }

  half_t* Qb = ws.QKVH;
  half_t* Kb = Qb + (size_t)M * D;
  half_t* Vb = Kb + (size_t)M * D;
  // FOLDED LayerNorm-modulate (default; gemm_kernel.h): only where its preconditions hold -- whole 256-row tiles of ONE image each (or one shared
  // modulation row), row partials in D / 256 slots, all four GEMMs chip-filling on the 16x16x32 kernel.  Everything else (small batches, DiT-S / XL
  // widths, patch 4 / 8 token counts with per-image conditioning, forced kernels) takes the separate ln_modulate launches below.
  const long uvs_q = rows == 1 ? 0 : 3 * D, uvs_f = rows == 1 ? 0 : H;
  int cen_cur = 0;
  if (fold) {
    if (!pe_mfma) {  // (the MFMA patch embedding has already written A', the partials and the row means)
      hipLaunchKernelGGL(ln_center_mod_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, ws.X, ws.A, M, D, T, ws.mod + D, mstride, ws.ln_part, tiles_p,
                         ws.cen[0]);
      LFM_CHECK_LAUNCH();
    }
  }
  auto rowstat_src = [&]() {  // consumer: reads cen[cen_cur], publishes the new row means into the other array, which becomes current
    RowStatSrc r{ws.ln_part, ws.cen[cen_cur], ws.cen[cen_cur ^ 1], tiles_p, 1.0f / (float)D, 1e-6f};
    cen_cur ^= 1;
    return r;
  };
  auto launch_fold = [&](const ASrcRowMajor& a, const half_t* Wp, long ldw_, int M_, int N_, int K_, const auto& e) {
    return w6 ? launch_gemm256w_tn(a, Wp, ldw_, M_, N_, K_, e, st) : launch_gemm256h_tn(a, Wp, ldw_, M_, N_, K_, e, st);
  };
#ifdef LFM_MEASURE
  const bool chk = g_chk && workspace == g_chk_ws;
  if (chk) (void)hipMemsetAsync(g_chk, 0, DIT_CHK_SLOTS * 8, st);
#define DIT_CHK(buf, bytes, slot) \
  if (chk) dit_chk(buf, bytes, i, slot, st)
#else
#define DIT_CHK(buf, bytes, slot)
#endif
  const bool prof_blk = prof_ok && g_prof_blk_count < LFM_PROF_BLK_MAX;
  if (prof_blk) (void)hipEventRecord(g_prof_blk_ev[2 * g_prof_blk_count], st);
  const bool fused = plan.fused;  // QKV projection + attention in one kernel (qkv_attention_kernel.h)
  if (fold) {
    for (int i = 0; i < s->depth; ++i) {
      const float* mod = ws.mod + (long)i * 6 * D;  // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
      const float* uq = ws.uvq + (long)i * 2 * rows * 3 * D;
      const float* uf = ws.uvf + (long)i * 2 * rows * H;
      // attention output / proj operand `Ob` and the fc1 operand `A1b`: the two-kernel path reuses ws.A for O (the QKV GEMM is over); the fused kernel writes
      // an image's O while other heads of the image still read its A' rows, so O goes to ws.A2 and proj hands A' back through ws.A
      half_t* Ob = fused ? ws.A2 : ws.A;
      half_t* A1b = fused ? ws.A : ws.A2;
      if (fused) {
        const QkvAttnArgs e_qa{uq, uq + (long)rows * 3 * D, uvs_q, rowstat_src(), Ob, D, s->heads, 0.125f * 1.4426950408889634f, nullptr, 0};
        rc = launch_qkv_attention(ws.A, D, (const half_t*)w->qkv_w + (size_t)i * 3 * D * D, D, M, D, s->heads, D, e_qa, st);
        if (rc) return rc;
      } else {
        const EpiQKVMod e_qkv{Qb, Kb, Vb, uq, uq + (long)rows * 3 * D, uvs_q, D, D / s->heads, T, EpiQKV::log2_or_neg(T), rowstat_src(), nullptr, 0};
        rc = launch_fold(ASrcRowMajor{ws.A, D, M, 0}, (const half_t*)w->qkv_w + (size_t)i * 3 * D * D, D, M, 3 * D, D, e_qkv);
        if (rc) return rc;
        DIT_CHK(ws.QKVH, (size_t)3 * M * D * 2, 0);
        rc = attention_launch(Qb, Kb, Vb, Ob, B, s->heads, D / s->heads, T, st);
        if (rc) return rc;
      }
      DIT_CHK(Ob, (size_t)M * D * 2, 1);
      // proj: X += gate_msa * (.), A' for fc1 with scale_mlp, partials; c = the row means the qkv GEMM just published
      const EpiGateResidMod e_proj{ws.X, D, w->proj_b + (size_t)i * D, mod + 2 * D, mstride, T, A1b, mod + 4 * D, mstride, ws.cen[cen_cur], ws.ln_part, tiles_p};
      rc = launch_fold(ASrcRowMajor{Ob, D, M, 0}, (const half_t*)w->proj_w + (size_t)i * D * D, D, M, D, D, e_proj);
      if (rc) return rc;
      DIT_CHK(ws.X, (size_t)M * D * 4, 2);
      DIT_CHK(A1b, (size_t)M * D * 2, 3);
      DIT_CHK(ws.ln_part, (size_t)M * tiles_p * 8, 4);
      const bool prof = prof_ok && g_prof_mode == 1 && g_prof_count < LFM_PROF_MAX;
      if (prof) (void)hipEventRecord(g_prof_ev[2 * g_prof_count], st);
#if defined(LFM_MEASURE) && defined(LFM_EXP_DUMP)
      const EpiModGeluF16 e_fc1{ws.QKVH, H, uf, uf + (long)rows * H, uvs_f, T, rowstat_src(), nullptr, 0,
                                (g_dbg && workspace == g_dbg_ws) ? g_dbg + (long)i * g_dbg_stride : (float*)nullptr};
#else
      const EpiModGeluF16 e_fc1{ws.QKVH, H, uf, uf + (long)rows * H, uvs_f, T, rowstat_src(), nullptr, 0};
#endif
      rc = launch_fold(ASrcRowMajor{A1b, D, M, 0}, (const half_t*)w->fc1_w + (size_t)i * H * D, D, M, H, D, e_fc1);
      if (rc) return rc;
      if (prof) (void)hipEventRecord(g_prof_ev[2 * g_prof_count++ + 1], st);
      DIT_CHK(ws.QKVH, (size_t)M * H * 2, 5);
      if (i + 1 < s->depth) {  // fc2 writes the NEXT block's A' (its scale_msa)
        const EpiGateResidMod e_fc2{ws.X, D, w->fc2_b + (size_t)i * D, mod + 5 * D, mstride, T, ws.A, mod + 7 * D, mstride, ws.cen[cen_cur], ws.ln_part, tiles_p};
        rc = launch_fold(ASrcRowMajor{ws.QKVH, H, M, 0}, (const half_t*)w->fc2_w + (size_t)i * D * H, H, M, D, H, e_fc2);
      } else {
        const EpiGateResidF32 e_fc2{ws.X, D, w->fc2_b + (size_t)i * D, mod + 5 * D, mstride, T};
        rc = launch_fold(ASrcRowMajor{ws.QKVH, H, M, 0}, (const half_t*)w->fc2_w + (size_t)i * D * H, H, M, D, H, e_fc2);
      }
      if (rc) return rc;
      DIT_CHK(ws.X, (size_t)M * D * 4, 6);
      DIT_CHK(ws.A, (size_t)M * D * 2, 7);
    }
  }
#undef DIT_CHK
  // Latency mode (one image of <= 256 tokens): the four linears of a block on the skinny kernel (gemm_skinny_kernel.h): qkv / fc1 with their epilogues in the
  // kernel, proj / fc2 as K-slices (256 workgroups) into the slabs of the row-owning finish + LayerNorm kernel
  auto sk_slices = [&](int N_, int K_) {  // K-slices so that column slices x slices ~ 256 workgroups; slabs must fit
    int sl = 1;
    while ((N_ / SK_BN) * (sl * 2) <= 256 && (K_ % (sl * 2 * SK_BK)) == 0 && K_ / (sl * 2) >= 2 * SK_BK && (size_t)(sl * 2) * M * N_ * 4 <= ws.slab_bytes) sl *= 2;
    return sl;
  };
  // ... or, for images of whole 64-token tiles, on the 64x64 kernel (gemm_sq64_kernel.h: half the bytes per workgroup)
  auto sq_slices = [&](int N_, int K_) {
    int sl = 1;
    while ((N_ / SQ_T) * (M / SQ_T) * (sl * 2) <= 256 && (K_ % (sl * 2 * SQ_BK)) == 0 && K_ / (sl * 2) >= 2 * SQ_BK && (size_t)(sl * 2) * M * N_ * 4 <= ws.slab_bytes) sl *= 2;
    return sl;
  };
  const bool sq64 = g_opt_skinny == 1 && M >= SQ_T && (M % SQ_T) == 0 && gemm_sq64_ok(ws.A, D, w->qkv_w, D, M, 3 * D, D, 1) && gemm_sq64_ok(ws.A, D, w->fc1_w, D, M, H, D, 1) &&
                    gemm_sq64_ok(ws.A, D, w->proj_w, D, M, D, D, sq_slices(D, D)) && gemm_sq64_ok(ws.QKVH, H, w->fc2_w, H, M, D, H, sq_slices(D, H));
  const int sk_s_proj = sq64 ? sq_slices(D, D) : sk_slices(D, D), sk_s_fc2 = sq64 ? sq_slices(D, H) : sk_slices(D, H);
  const bool skinny = !fold && g_opt_skinny && g_gemm_sel == 0 && M <= SK_ROWS && ws.slab && D <= 1024 * SPLITK_LN_MAXJ && (D % 4) == 0 &&
                      gemm_skinny_ok(ws.A, D, w->qkv_w, D, M, 3 * D, D, 1) && gemm_skinny_ok(ws.A, D, w->fc1_w, D, M, H, D, 1) &&
                      gemm_skinny_ok(ws.A, D, w->proj_w, D, M, D, D, sk_s_proj) && gemm_skinny_ok(ws.QKVH, H, w->fc2_w, H, M, D, H, sk_s_fc2);
  auto lat_gemm = [&](const half_t* A_, long lda_, const half_t* W_, int N_, int K_, const auto& epi_, int S_) {  // the latency-mode linear: lda == ldw == K
    return sq64 ? launch_gemm_sq64(A_, lda_, W_, (long)K_, M, N_, K_, epi_, S_, st) : launch_gemm_skinny(A_, lda_, W_, (long)K_, M, N_, K_, epi_, S_, st);
  };
  bool a_ready = false;  // latency mode: the previous split-K finish already wrote this LayerNorm's output
  for (int i = 0; i < (fold ? 0 : s->depth); ++i) {
    const float* mod = ws.mod + (long)i * 6 * D;  // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
    if (!a_ready) {
      rc = ln_modulate_launch(ws.X, ws.A, M, D, T, mod, mod + D, mstride, st);
      if (rc) return rc;
    }
    a_ready = false;
    const EpiQKV e_qkv = EpiQKV::make(Qb, Kb, Vb, w->qkv_b + (size_t)i * 3 * D, D, D / s->heads, T);
    if (skinny) rc = lat_gemm(ws.A, D, (const half_t*)w->qkv_w + (size_t)i * 3 * D * D, 3 * D, D, e_qkv, 1);
    else {
      rc = launch_gemm_splitk(ws.A, D, (const half_t*)w->qkv_w + (size_t)i * 3 * D * D, D, M, 3 * D, D, e_qkv, ws.slab, ws.slab_bytes, st);
      if (rc == 1) rc = launch_gemm_auto(ASrcRowMajor{ws.A, D, M, 0}, (const half_t*)w->qkv_w + (size_t)i * 3 * D * D, D, M, 3 * D, D, e_qkv, st);
    }
    if (rc) return rc;
    rc = attention_launch(Qb, Kb, Vb, ws.A, B, s->heads, D / s->heads, T, st);
    if (rc) return rc;
    const EpiGateResidF32 e_proj{ws.X, D, w->proj_b + (size_t)i * D, mod + 2 * D, mstride, T};
    // small M: split-K whose finish kernel is also the LayerNorm-modulate in front of fc1 (its input ws.A is consumed by the slab GEMM before the finish writes it)
    if (skinny) {  // K-slices of the skinny kernel into slabs, then the row-owning finish that is also the LayerNorm-modulate in front of fc1
      rc = lat_gemm(ws.A, D, (const half_t*)w->proj_w + (size_t)i * D * D, D, D, EpiSlabF32{ws.slab, (long)D, (long)M * D}, sk_s_proj);
      if (rc) return rc;
      hipLaunchKernelGGL(splitk_finish_resid_ln_kernel, dim3(M), dim3(256), 0, st, ws.slab, sk_s_proj, (long)M * D, D, e_proj.X, e_proj.ldx, e_proj.bias, e_proj.gate,
                         e_proj.gate_stride, e_proj.tokens, ws.A, mod + 3 * D, mod + 4 * D, mstride);
      LFM_CHECK_LAUNCH();
    } else
      rc = launch_gemm_splitk_resid_ln(ws.A, D, (const half_t*)w->proj_w + (size_t)i * D * D, D, M, D, D, e_proj, ws.A, mod + 3 * D, mod + 4 * D, mstride, ws.slab,
                                       ws.slab_bytes, st);
    if (rc == 1) {
      rc = launch_gemm_auto(ASrcRowMajor{ws.A, D, M, 0}, (const half_t*)w->proj_w + (size_t)i * D * D, D, M, D, D, e_proj, st);
      if (rc) return rc;
      rc = ln_modulate_launch(ws.X, ws.A, M, D, T, mod + 3 * D, mod + 4 * D, mstride, st);
    }
    if (rc) return rc;
    const bool prof = prof_ok && g_prof_mode == 1 && g_prof_count < LFM_PROF_MAX;
    if (prof) (void)hipEventRecord(g_prof_ev[2 * g_prof_count], st);
    const EpiBiasGeluF16 e_fc1{ws.QKVH, H, w->fc1_b + (size_t)i * H};
    if (skinny) rc = lat_gemm(ws.A, D, (const half_t*)w->fc1_w + (size_t)i * H * D, H, D, e_fc1, 1);
    else {
      rc = launch_gemm_splitk(ws.A, D, (const half_t*)w->fc1_w + (size_t)i * H * D, D, M, H, D, e_fc1, ws.slab, ws.slab_bytes, st);
      if (rc == 1) rc = launch_gemm_auto(ASrcRowMajor{ws.A, D, M, 0}, (const half_t*)w->fc1_w + (size_t)i * H * D, D, M, H, D, e_fc1, st);
    }
    if (rc) return rc;
    if (prof) (void)hipEventRecord(g_prof_ev[2 * g_prof_count++ + 1], st);
    const EpiGateResidF32 e_fc2{ws.X, D, w->fc2_b + (size_t)i * D, mod + 5 * D, mstride, T};
    const bool more = i + 1 < s->depth;  // ... and the one in front of the NEXT block's qkv (its shift_msa / scale_msa)
    if (skinny) {
      rc = lat_gemm(ws.QKVH, H, (const half_t*)w->fc2_w + (size_t)i * D * H, D, H, EpiSlabF32{ws.slab, (long)D, (long)M * D}, sk_s_fc2);
      if (rc) return rc;
      hipLaunchKernelGGL(splitk_finish_resid_ln_kernel, dim3(M), dim3(256), 0, st, ws.slab, sk_s_fc2, (long)M * D, D, e_fc2.X, e_fc2.ldx, e_fc2.bias, e_fc2.gate,
                         e_fc2.gate_stride, e_fc2.tokens, more ? ws.A : (half_t*)nullptr, mod + 6 * D, mod + 7 * D, mstride);
      LFM_CHECK_LAUNCH();
    } else
      rc = launch_gemm_splitk_resid_ln(ws.QKVH, H, (const half_t*)w->fc2_w + (size_t)i * D * H, H, M, D, H, e_fc2, more ? ws.A : (half_t*)nullptr, mod + 6 * D, mod + 7 * D,
                                       mstride, ws.slab, ws.slab_bytes, st);
    if (rc == LFM_OK) a_ready = more;
    if (rc == 1) rc = launch_gemm_auto(ASrcRowMajor{ws.QKVH, H, M, 0}, (const half_t*)w->fc2_w + (size_t)i * D * H, H, M, D, H, e_fc2, st);
    if (rc) return rc;
  }
  if (prof_blk) (void)hipEventRecord(g_prof_blk_ev[2 * g_prof_blk_count++ + 1], st);
  const float* fmod = ws.mod + (long)s->depth * 6 * D;
  const int Mh = cfg ? M / 2 : M;
  // skinny MFMA GEMM (16 rows per wave) when the shape allows it: whole 16-row tiles inside one image half, 16-column output tiles, D % 32 == 0
  const int NO = s->in_ch * s->patch * s->patch;
  const bool fin_mfma = (NO == 16 || NO == 64) && (D % 32 == 0) && (T % 16 == 0) && (M % 16 == 0) && !(g_gemm_dbg & 1048576);  // flag 1048576: the round-1 kernel (A/B)
  if (fin_mfma) {
    const long ntiles = cfg ? Mh / 8 : M / 16;
#define FIN_LAUNCH(CF, NTT)                                                                                                                 \
  hipLaunchKernelGGL((final_layer_mfma_kernel<CF, NTT>), dim3((unsigned)ntiles), dim3(256), 0, st, ws.X, M, D, T, fmod, fmod + D, mstride,   \
                     w->final_w, w->final_b, s->in_ch, s->res, s->patch, cfg ? c->cfg_scale : 1.0f, c->out, c->axpy_base, c->axpy_dt)
    if (cfg) {
      if (NO == 16) FIN_LAUNCH(true, 1);
      else FIN_LAUNCH(true, 4);
    } else {
      if (NO == 16) FIN_LAUNCH(false, 1);
      else FIN_LAUNCH(false, 4);
    }
#undef FIN_LAUNCH
  } else if (cfg)
    hipLaunchKernelGGL(final_layer_kernel<true>, dim3(cdiv(Mh, 8)), dim3(256), 0, st, ws.X, M, D, T, fmod, fmod + D, mstride, w->final_w,
                       w->final_b, s->in_ch, s->res, s->patch, c->cfg_scale, c->out, c->axpy_base, c->axpy_dt);
  else
    hipLaunchKernelGGL(final_layer_kernel<false>, dim3(cdiv(Mh, 16)), dim3(256), 0, st, ws.X, M, D, T, fmod, fmod + D, mstride, w->final_w,
                       w->final_b, s->in_ch, s->res, s->patch, 1.0f, c->out, c->axpy_base, c->axpy_dt);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

// ------------------------------------------------------------------ effective clock under matrix load (measurement aid for bench.py)
// The GEMMs of this path run under the board power cap: boxes of the pool differ by 6-7 % in the clock they sustain, and a
// roofline fraction means little without it.  One workgroup per CU streams v_mfma_f32_16x16x32_f16 on pseudo-random fp16 operands (the load the GEMMs
// put on the chip) for `iters` x 64 instructions per wave and brackets the stream with s_memtime (one tick = one shader cycle, MI355X_MICROARCH.md):
// ticks / wall time = the sustained clock.  out[0 .. blocks) = ticks per workgroup.
__global__ __launch_bounds__(512) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters, unsigned seed) {
  half8_t a[4], b[4];
  unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u;
      a[i][e] = (half_t)(((int)(h >> 16) & 1023) * (1.0f / 512.0f) - 1.0f);
      h = h * 1664525u + 1013904223u;
      b[i][e] = (half_t)(((int)(h >> 16) & 1023) * (1.0f / 512.0f) - 1.0f);
    }
  f32x4 c[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) c[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 16; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j & 3], b[(j >> 2) & 3], c[j], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += c[j][0] + c[j][3];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (s == 123.456f) out[blockIdx.x] = 0;  // keeps the accumulators live
}

extern "C" int lfm_clock_probe(int blocks, int iters, unsigned long long* ticks_out, lfm_stream_t stream) {
  if (!ticks_out || blocks <= 0 || iters <= 0) return LFM_ERR_ARG;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, ticks_out, iters, 12345u);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

extern "C" int lfm_grid_advance(const float* ts, const float* dts, int* step, float* t_cur, float* t_next, float* dt_cur, lfm_stream_t stream) {
  if (!ts || !dts || !step || !t_cur || !t_next || !dt_cur) return LFM_ERR_ARG;
  hipLaunchKernelGGL(grid_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ts, dts, step, t_cur, t_next, dt_cur);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

extern "C" int lfm_rk_error_norm(const float* y0, const float* y1, const float* const* k_host_ptrs, const float* e_coef, const float* dt, int nk, long n,
                                 float rtol, float atol, float* scratch, float* out, lfm_stream_t stream) {
  if (!y0 || !y1 || !k_host_ptrs || !e_coef || !dt || !scratch || !out || nk <= 0 || nk > 8) return LFM_ERR_ARG;
  if (n <= 0 || n % 4 || (((uintptr_t)y0 | (uintptr_t)y1) & 15)) return LFM_ERR_ALIGN;
  LinPtrs p;
  for (int i = 0; i < 8; ++i) p.k[i] = i < nk ? k_host_ptrs[i] : nullptr;
  const int nb = (int)(cdiv(n / 4, 256) < RK_BLOCKS ? cdiv(n / 4, 256) : RK_BLOCKS);
  hipLaunchKernelGGL(rk_err_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, y0, y1, p, e_coef, dt, nk, n / 4, rtol, atol, scratch);
  LFM_CHECK_LAUNCH();
  hipLaunchKernelGGL(rk_err_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, nb, 1.0f / (float)n, out);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

extern "C" int lfm_lincomb(float* out, const float* base, const float* const* k_host_ptrs, const float* coef, const float* scale, int nk,
                           long n, lfm_stream_t stream) {
  if (!out || !coef || nk < 0 || nk > 8 || (nk && !k_host_ptrs)) return LFM_ERR_ARG;
  if (n % 4 || ((uintptr_t)out & 15)) return LFM_ERR_ALIGN;
  LinPtrs p;
  for (int i = 0; i < 8; ++i) p.k[i] = i < nk ? k_host_ptrs[i] : nullptr;
  hipLaunchKernelGGL(lincomb_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, base, p, coef, scale, nk, n / 4);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
