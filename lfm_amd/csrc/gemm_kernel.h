// fp16 x fp16 -> fp32 MFMA GEMM for gfx950:  C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue).
//
// Both operands are K-contiguous ("TN"): activations A[M,K] row-major and weights W[N,K] exactly as
// torch.nn.Linear stores them, so an MFMA fragment is one 16-byte load per lane.
//
// Structure (v1, "2-barrier"): 128x128x64 tile, 256 threads = 4 waves in 2(M) x 2(N), each wave a
// 64x64 sub-tile = 2x2 v_mfma_f32_32x32x16_f16 fragments.  Operands are staged by 16-byte LDS-DMA
// (global_load_lds_dwordx4) into a double-buffered, XOR-swizzled LDS image:
//   LDS row = 64 halves = 128 B = 8 chunks of 16 B;  chunk c of row r lives at chunk c ^ ((r>>1)&7).
// ds_read_b128 is serviced in 16-lane groups over a 256-B bank row (= two tile rows), so the key is
// (r>>1)&7: the 16 rows a group touches land on 16 distinct 16-B slots (conflict-free).  The DMA
// destination is lane-linear, so the swizzle is applied to the per-lane SOURCE address and to the
// read address (both-sides rule).  The next tile's DMA stays in flight across the barrier
// (counted s_waitcnt vmcnt + raw s_barrier).
//
// The MFMA is issued with W as the A-operand and the activations as the B-operand, i.e. it computes
// the C^T fragment: lane l then owns column m = l&31 and FOUR CONSECUTIVE n per register group, so
// the epilogue sees (m, n..n+3, float4) and can do 8/16-byte stores and per-n vector loads.
#pragma once
#include "common.h"

int lfm_gemm_selected_v1_ok();  // 1 unless a 256x256 kernel is being forced (lfm_gemm_select 2 / 3): split-K runs on the 128x128 kernel

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_STAGE_BYTES (2 * GEMM_BM * GEMM_BK * 2)  // A + W tile, 32 KiB
#define GEMM_LDS_BYTES (2 * GEMM_STAGE_BYTES)         // double buffered, 64 KiB

// ------------------------------------------------------------------ A-operand sources
// Plain row-major activations; rows >= M are clamped (their results are never stored).
struct ASrcRowMajor {
  const half_t* A;
  long lda;
  int M;
  __device__ __forceinline__ void init(int bz, long bs) { A += (long)bz * bs; }
  // a row is a 32-BIT element offset from the (uniform) base A + k0: an LDS-DMA address is then SGPR base + VGPR offset -- one register per row
  // instead of a 64-bit pointer and no 64-bit VALU add per issue (the 256x256 kernels are register-bound).  M * lda < 2^31 (checked at launch).
  struct Row {
    unsigned off;
  };
  __device__ __forceinline__ Row row(int m) const {
    Row r;
    r.off = (unsigned)((m < M ? m : M - 1) * (int)lda);
    return r;
  }
  int k0;
  __device__ __forceinline__ void begin_tile(int kt, int bk) { k0 = kt * bk; }
  __device__ __forceinline__ const half_t* ptr(const Row& r, int koff) const { return (A + k0) + (r.off + (unsigned)koff); }
  bool fits() const { return (long)M * lda < (1L << 31); }
  // buffer-addressed LDS-DMA (round 4; tools/ubench/ldsdma_rate.hip: buffer_load_dwordx4 .. offen lds with an SGPR resource, a 32-bit VGPR byte
  // offset and the K offset in an SGPR sustains 47.9 B/clk/CU from eight waves where global_load_lds with 64-bit VGPR addresses sustains 42.2, and
  // needs no 64-bit VALU add per issue): resource over A (num_records = 2^32 - 1: the range check never fires), per-row byte offset, per-K-tile
  // scalar offset.  Byte offsets are UNSIGNED 32-bit: operands up to 2^31 elements (tests/test_gpu_dit.py::test_gemm_operands_beyond_2gb).
  static constexpr bool buffer_form = true;
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc() const { return __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, -1, 0x00020000); }
  __device__ __forceinline__ unsigned voff(const Row& r, int koff) const { return (r.off + (unsigned)koff) * 2u; }
  __device__ __forceinline__ unsigned soff() const { return (unsigned)k0 * 2u; }
  bool fits_buffer() const { return (long)M * lda < (1L << 31); }
};
template <class ASrc, class = void>
struct asrc_has_buffer {
  static constexpr bool value = false;
};
template <class ASrc>
struct asrc_has_buffer<ASrc, decltype((void)ASrc::buffer_form)> {
  static constexpr bool value = ASrc::buffer_form;
};
template <class ASrc>
static inline auto asrc_fits_buffer(const ASrc& a, int) -> decltype(a.fits_buffer()) {
  return a.fits_buffer();
}
template <class ASrc>
static inline bool asrc_fits_buffer(const ASrc&, long) {
  return false;
}
// 16-byte LDS-DMA through a buffer resource: address = resource base + voff (VGPR, bytes) + soff (SGPR, bytes)
__device__ __forceinline__ void glds16_buf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds_wave_base), 16, (int)voff, (int)soff, 0, 0);
}
// A = [A1 | A2]: the K range [0, Ca) from A1 [M, Ca], [Ca, Ca + Cb) from A2 [M, Cb] (dense rows); Ca is a multiple of every kernel's K-tile depth,
// so a K-tile lies in one of the two.  Not batched.
struct ASrcRowMajor2 {
  const half_t* A;
  const half_t* B;
  int Ca, Cb, M;
  int k0;
  __device__ __forceinline__ void init(int, long) {}
  struct Row {
    unsigned ra, rb;
  };
  __device__ __forceinline__ Row row(int m) const {
    const int mm = m < M ? m : M - 1;
    Row r;
    r.ra = (unsigned)(mm * Ca);
    r.rb = (unsigned)(mm * Cb);
    return r;
  }
  __device__ __forceinline__ void begin_tile(int kt, int bk) { k0 = kt * bk; }
  __device__ __forceinline__ const half_t* ptr(const Row& r, int koff) const {
    return k0 < Ca ? (A + k0) + (r.ra + (unsigned)koff) : (B + (k0 - Ca)) + (r.rb + (unsigned)koff);
  }
  bool fits() const { return (long)M * (Ca > Cb ? Ca : Cb) < (1L << 31); }
};
// A sources with addressing limits say so through fits(); every launcher asks
template <class ASrc>
static inline auto asrc_fits(const ASrc& a, int) -> decltype(a.fits()) {
  return a.fits();
}
template <class ASrc>
static inline bool asrc_fits(const ASrc&, long) {
  return true;
}

// ------------------------------------------------------------------ epilogues
// Two-phase so the interior-tile path can issue ALL its loads before the first store (the compiler
// will not move a load above a possibly-aliasing store):  aux = epi.load(m, n);  epi.store(m, n, v, aux)
// with v = C[m][n..n+3] (fp32 accumulators).  m < M and n+3 < N are guaranteed by the caller.
struct EpiBiasF16 {  // C = acc + bias  -> fp16
  half_t* C;
  long ldc;
  const float* bias;  // may be null
  typedef f32x4 Aux;
  static constexpr bool column_aux = true;
  __device__ __forceinline__ Aux load(int, int n) const { return bias ? *(const f32x4*)(bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& b) const {
    v += b;
    half4_t h = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
    *(half4_t*)(C + (long)m * ldc + n) = h;
  }
  __device__ __forceinline__ bool wide_ok() const { return (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0; }  // 16-byte stores are aligned
  __device__ __forceinline__ void store8(int m, int n, f32x4 lo, f32x4 hi, const Aux& bl, const Aux& bh) const {
    lo += bl;
    hi += bh;
    half8_t h = {(half_t)lo.x, (half_t)lo.y, (half_t)lo.z, (half_t)lo.w, (half_t)hi.x, (half_t)hi.y, (half_t)hi.z, (half_t)hi.w};
    *(half8_t*)(C + (long)m * ldc + n) = h;
  }
};

struct EpiBiasGeluF16 {  // C = gelu_tanh(acc + bias) -> fp16   (timm Mlp fc1, DiT.py:123-124)
  half_t* C;
  long ldc;
  const float* bias;
  typedef f32x4 Aux;
  static constexpr bool column_aux = true;
  __device__ __forceinline__ Aux load(int, int n) const { return *(const f32x4*)(bias + n); }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& b) const {
    v += b;
    const f32x2_t a = gelu_tanh_pk((f32x2_t){v.x, v.y}), c = gelu_tanh_pk((f32x2_t){v.z, v.w});
    half4_t h = {(half_t)a.x, (half_t)a.y, (half_t)c.x, (half_t)c.y};
    *(half4_t*)(C + (long)m * ldc + n) = h;
  }
  __device__ __forceinline__ bool wide_ok() const { return (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0; }  // 16-byte stores are aligned
  __device__ __forceinline__ void store8(int m, int n, f32x4 lo, f32x4 hi, const Aux& bl, const Aux& bh) const {
    lo += bl;
    hi += bh;
    const f32x2_t a = gelu_tanh_pk((f32x2_t){lo.x, lo.y}), c = gelu_tanh_pk((f32x2_t){lo.z, lo.w});
    const f32x2_t e = gelu_tanh_pk((f32x2_t){hi.x, hi.y}), g = gelu_tanh_pk((f32x2_t){hi.z, hi.w});
    half8_t h = {(half_t)a.x, (half_t)a.y, (half_t)c.x, (half_t)c.y, (half_t)e.x, (half_t)e.y, (half_t)g.x, (half_t)g.y};
    *(half8_t*)(C + (long)m * ldc + n) = h;
  }
};

struct EpiBiasF32 {  // C = acc + bias -> fp32   (adaLN modulation table)
  float* C;
  long ldc;
  const float* bias;
  typedef f32x4 Aux;
  static constexpr bool column_aux = true;
  __device__ __forceinline__ Aux load(int, int n) const { return bias ? *(const f32x4*)(bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& b) const { *(f32x4*)(C + (long)m * ldc + n) = v + b; }
};

// X[m][n] += gate[img(m)][n] * (acc + bias[n]);  fp32 residual stream (DiT.py:129-130)
struct EpiGateResidF32 {
  float* X;
  long ldx;
  const float* bias;
  const float* gate;  // gate + img*gate_stride + n
  long gate_stride;   // floats between images' modulation rows (0 => one shared row)
  int tokens;         // rows per image
  struct Aux {
    f32x4 b, g, x;
  };
  __device__ __forceinline__ Aux load(int m, int n) const {
    Aux a;
    a.b = *(const f32x4*)(bias + n);
    a.g = *(const f32x4*)(gate + (long)(m / tokens) * gate_stride + n);
    a.x = *(const f32x4*)(X + (long)m * ldx + n);
    return a;
  }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& a) const {
    *(f32x4*)(X + (long)m * ldx + n) = a.x + a.g * (v + a.b);
  }
};

// ------------------------------------------------------------------ adaLN LayerNorm-modulate FOLDED into the GEMM epilogues (round 3)
// modulate(LayerNorm(x), shift, scale) (DiT.py:20-21, 129-130) feeds a Linear, so with mu, rstd the row statistics of x and c any per-row constant
//   (LN(x) (1 + s) + sh) W^T  =  rstd * [ ((x - c)(1 + s)) W^T  -  (mu - c) * u ]  +  v,     u[n] = sum_k (1 + s[k]) W[n][k],  v[n] = sum_k sh[k] W[n][k] + bias[n].
// PRODUCER = the gated-residual GEMM that updates x (proj, fc2; EpiGateResidMod): its epilogue already holds the new x in registers, so it also
//   writes A' = fp16((x - c)(1 + s)) -- the consumer's A operand -- and per-row partials (sum x, sum (x - c)^2) of its 256 columns into a fixed slot
//   part[m][tile_n] (plain stores, deterministic).  c = cen[m] = the row mean BEFORE this update (written by the previous consumer), so x - c is
//   centred up to the mean shift of one residual update: no cancellation in the fp16 rounding of A' or in the one-pass variance.
// CONSUMER = the GEMM that reads A' (qkv, fc1; rowstat epilogues below): before its K loop every tile reduces its 256 rows' partials to
//   a = rstd, b = -rstd (mu - c) in LDS (and tile column 0 publishes mu as the next producer's c); its epilogue evaluates a * acc + (b * u + v).
// No inter-workgroup synchronisation, no second pass over x: both LN-modulate launches of a block (2 x 100 MB of HBM traffic) disappear for
// 33.5 MB of extra stores in each producer epilogue.  u, v: one small batched GEMM per forward over all blocks (lfm_dit_forward).
// (Rejected first, measured in round 3: normalising inside the producer epilogue behind an inter-workgroup panel counter -- correct, bit-stable,
// but 12.03 vs 11.92 ms per DiT-L/2 evaluation: the wait for the three sibling tiles cost more than the two launches it saved.)
struct RowStatSrc {
  const float* part;    // [M][tiles_p][2]
  const float* cen_in;  // [M] the c the producer used
  float* cen_out;       // [M] <- mu (written by tile column 0 only)
  int tiles_p;          // partial slots per row (the producer's column tiles)
  float inv_n, eps;     // 1 / row length, LayerNorm eps
};

// X[m][n] += gate * (acc + bias)  AND  A'[m][n] = fp16((X[m][n] - c[m]) (1 + scale[n])),  part[m][tile_n] = (sum X, sum (X - c)^2)
// (Round 4, measured and removed: a variant whose epilogue requested the X rows of the next 32-row pass before the stores of the current one --
// bit-identical, 10.97 vs 10.95 ms per DiT-L/2 forward.  The four-wave kernel's own producer epilogue keeps that order: it costs nothing there.)
struct EpiGateResidMod {
  float* X;
  long ldx;
  const float* bias;
  const float* gate;
  long gate_stride;
  int tokens;
  half_t* A;           // [M][N], leading dimension N
  const float* scale;  // the consumer LayerNorm's scale row (+ img * mod_stride)
  long mod_stride;
  const float* cen;    // [M]
  float* part;         // [M][tiles_n][2]
  int tiles_n;
  static constexpr bool producer_mod = true;
  struct Aux {
    f32x4 b, g, x;
  };
  __device__ __forceinline__ Aux load(int m, int n) const {
    Aux a;
    a.b = *(const f32x4*)(bias + n);
    a.g = *(const f32x4*)(gate + (long)(m / tokens) * gate_stride + n);
    a.x = *(const f32x4*)(X + (long)m * ldx + n);
    return a;
  }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& a) const {
    *(f32x4*)(X + (long)m * ldx + n) = a.x + a.g * (v + a.b);
  }
};
template <class Epi, class = void>
struct epi_is_producer_mod {
  static constexpr bool value = false;
};
template <class Epi>
struct epi_is_producer_mod<Epi, decltype((void)Epi::producer_mod)> {
  static constexpr bool value = Epi::producer_mod;
};

// C = a[m] * acc + (b[m] * u[n] + v[n]) -> fp16;  (a, b) of the tile's rows sit in LDS at rs[2 (m - m0)] (filled by the kernel's prologue)
struct EpiModF16 {
  half_t* C;
  long ldc;
  const float* u;  // + img * uv_stride + n
  const float* v;
  long uv_stride;
  int tokens;
  const float* rs;
  int m0;
  struct Aux {
    f32x4 u, v;
  };
  static constexpr bool column_aux = true;
  __device__ __forceinline__ Aux load(int m, int n) const {
    const long o = (long)(m / tokens) * uv_stride + n;
    Aux a;
    a.u = *(const f32x4*)(u + o);
    a.v = *(const f32x4*)(v + o);
    return a;
  }
  __device__ __forceinline__ f32x4 affine(int m, f32x4 acc, const Aux& c) const {
    const f32x2 ab = *(const f32x2*)(rs + 2 * (m - m0));
    return row_affine4(ab.x, ab.y, acc, c.u, c.v);  // plain FMAs, not the packed form: common.h fma_v
  }
  __device__ __forceinline__ void store(int m, int n, f32x4 acc, const Aux& c) const {
    const f32x4 o = affine(m, acc, c);
    half4_t h = {(half_t)o.x, (half_t)o.y, (half_t)o.z, (half_t)o.w};
    *(half4_t*)(C + (long)m * ldc + n) = h;
  }
  __device__ __forceinline__ bool wide_ok() const { return (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0; }
  __device__ __forceinline__ f32x2 row_aux(int m) const { return *(const f32x2*)(rs + 2 * (m - m0)); }
  __device__ __forceinline__ void store8r(int m, int n, f32x4 lo, f32x4 hi, const Aux& cl, const Aux& ch, f32x2 ab) const {
    lo = row_affine4(ab.x, ab.y, lo, cl.u, cl.v);
    hi = row_affine4(ab.x, ab.y, hi, ch.u, ch.v);
    half8_t h = {(half_t)lo.x, (half_t)lo.y, (half_t)lo.z, (half_t)lo.w, (half_t)hi.x, (half_t)hi.y, (half_t)hi.z, (half_t)hi.w};
    *(half8_t*)(C + (long)m * ldc + n) = h;
  }
  __device__ __forceinline__ void store8(int m, int n, f32x4 lo, f32x4 hi, const Aux& cl, const Aux& ch) const {
    store8r(m, n, lo, hi, cl, ch, row_aux(m));
  }
};

// fc1 of the folded path: C = gelu_tanh(a[m] * acc + (b[m] * u[n] + v[n])) -> fp16   (v carries the fc1 bias)
struct EpiModGeluF16 {
  half_t* C;
  long ldc;
  const float* u;
  const float* v;
  long uv_stride;
  int tokens;
  RowStatSrc st;
  const float* rs;
  int m0;
  static constexpr bool rowstat = true;
  struct Aux {
    f32x4 u, v;
  };
  static constexpr bool column_aux = true;
  __device__ __forceinline__ Aux load(int m, int n) const {
    const long o = (long)(m / tokens) * uv_stride + n;
    Aux a;
    a.u = *(const f32x4*)(u + o);
    a.v = *(const f32x4*)(v + o);
    return a;
  }
  __device__ __forceinline__ void store(int m, int n, f32x4 acc, const Aux& c) const {
    const f32x2 ab = *(const f32x2*)(rs + 2 * (m - m0));
    const f32x4 x = row_affine4(ab.x, ab.y, acc, c.u, c.v);
    const f32x2_t p = gelu_tanh_pk((f32x2_t){x.x, x.y}), q = gelu_tanh_pk((f32x2_t){x.z, x.w});
    half4_t h = {(half_t)p.x, (half_t)p.y, (half_t)q.x, (half_t)q.y};
    *(half4_t*)(C + (long)m * ldc + n) = h;
  }
  __device__ __forceinline__ bool wide_ok() const { return (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0; }
  __device__ __forceinline__ f32x2 row_aux(int m) const { return *(const f32x2*)(rs + 2 * (m - m0)); }
  __device__ __forceinline__ void store8(int m, int n, f32x4 lo, f32x4 hi, const Aux& cl, const Aux& ch) const {
    store8r(m, n, lo, hi, cl, ch, row_aux(m));
  }
  __device__ __forceinline__ void store8r(int m, int n, f32x4 lo, f32x4 hi, const Aux& cl, const Aux& ch, f32x2 ab) const {
#if defined(LFM_MEASURE) && defined(LFM_EXP_DUMP)
    // (experiment build, tools/cosched_dump.py) every operand of the affine as THIS lane saw it, 16 floats per (row, 8-column group):
    // acc lo | acc hi | t = b u + v for columns 0, 2, 4, 6 | (a, b) | x for columns 0, 2
    const f32x4 acc_lo = lo, acc_hi = hi;
    const f32x4 tl = ab.y * cl.u + cl.v, th = ab.y * ch.u + ch.v;
    lo = ab.x * lo + tl;
    hi = ab.x * hi + th;
    if (dbg && (m & 7) >= 6) {  // lanes 48-63 of the hand-over
      f32x4* d = (f32x4*)(dbg + ((long)m * (ldc / 8) + n / 8) * 16);
      d[0] = acc_lo;
      d[1] = acc_hi;
      d[2] = (f32x4){tl.x, tl.z, th.x, th.z};
      d[3] = (f32x4){ab.x, ab.y, lo.x, lo.z};
    }
#elif defined(LFM_EXP_AFFINE_PACKED)
    // (experiment build) the vector expression: v_pk_fma_f32 with op_sel -- the form that leaves the solo result under co-scheduling
    lo = ab.x * lo + (ab.y * cl.u + cl.v);
    hi = ab.x * hi + (ab.y * ch.u + ch.v);
#else
    lo = row_affine4(ab.x, ab.y, lo, cl.u, cl.v);
    hi = row_affine4(ab.x, ab.y, hi, ch.u, ch.v);
#endif
    const f32x2_t a = gelu_tanh_pk((f32x2_t){lo.x, lo.y}), c = gelu_tanh_pk((f32x2_t){lo.z, lo.w});
    const f32x2_t e = gelu_tanh_pk((f32x2_t){hi.x, hi.y}), g = gelu_tanh_pk((f32x2_t){hi.z, hi.w});
    half8_t h = {(half_t)a.x, (half_t)a.y, (half_t)c.x, (half_t)c.y, (half_t)e.x, (half_t)e.y, (half_t)g.x, (half_t)g.y};
    *(half8_t*)(C + (long)m * ldc + n) = h;
  }
#if defined(LFM_MEASURE) && defined(LFM_EXP_DUMP)
  float* dbg;  // [M][N / 8][16] or null
#endif
};
template <class Epi, class = void>
struct epi_has_rowstat {
  static constexpr bool value = false;
};
template <class Epi>
struct epi_has_rowstat<Epi, decltype((void)Epi::rowstat)> {
  static constexpr bool value = Epi::rowstat;
};

// u / v rows of the folded path (lfm_dit_forward): rows [0, R) = (1 + scale) W^T, rows [R, 2R) = shift W^T + bias; batched over the blocks
struct EpiUV {
  float* C;
  long ldc;
  const float* bias;
  int R;
  long bs_bias;
  typedef f32x4 Aux;
  __device__ __forceinline__ void batch(int bz, long bs) {
    C += (long)bz * bs;
    bias += (long)bz * bs_bias;
  }
  __device__ __forceinline__ Aux load(int m, int n) const { return m >= R ? *(const f32x4*)(bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& b) const { *(f32x4*)(C + (long)m * ldc + n) = v + b; }
};

// Split-K partial tile: slice bz of the K range writes its fp32 partial product to slab[bz][M][N] (see launch_gemm_splitk).
struct EpiSlabF32 {
  float* slab;
  long ldn, slice_stride;
  typedef int Aux;
  __device__ __forceinline__ void batch(int bz, long) { slab += (long)bz * slice_stride; }
  __device__ __forceinline__ Aux load(int, int) const { return 0; }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux&) const { *(f32x4*)(slab + (long)m * ldn + n) = v; }
};

// V^T token order (round 4).  Inside every group of 16 tokens a V^T row stores the tokens in the order 0 1 2 3 8 9 10 11 | 4 5 6 7 12 13 14 15 (bits 2 and
// 3 of the token index exchanged: an involution).  Why: the attention kernel's P V MFMA takes, per k-slot of 16 keys, keys {4 h + r} and {8 + 4 h + r}
// (h = lane >> 5) from one lane -- the order its S^T = K Q^T accumulators already hold P in -- so with plain token order a lane needed TWO 8-byte LDS reads
// per fragment, and the 32 lanes of a half-wave could only reach 16 of the 32 eight-byte slots of a bank row (2-way conflicts by construction: 40 % of the
// kernel's LDS cycles, profiles/r03_final_pmc_in_situ.txt).  In this order the lane's eight keys are ONE 16-byte chunk: one conflict-free ds_read_b128.
// V^T is private to the QKV projection (writer) and the attention kernels (readers).
__host__ __device__ __forceinline__ int vt_pos(int tok) { return (tok & ~12) | ((tok & 4) << 1) | ((tok & 8) >> 1); }

// QKV projection of timm Attention (DiT.py:120): columns [q | k | v], each [head][hd].
// Q,K are stored token-major [M, D]; V is stored TRANSPOSED per (image, head): Vt[img][head][d][token],
// which is the key-contiguous layout the attention kernel's P*V MFMA operand wants.
struct EpiQKV {
  half_t* Q;
  half_t* K;
  half_t* Vt;
  const float* bias;
  int D, hd, tokens;
  int hd_sh, tok_sh;  // log2 when hd / tokens are powers of two (every DiT configuration), else -1: no integer division per store
  static __host__ __device__ int log2_or_neg(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
  }
  static EpiQKV make(half_t* Q, half_t* K, half_t* Vt, const float* bias, int D, int hd, int tokens) {
    return EpiQKV{Q, K, Vt, bias, D, hd, tokens, log2_or_neg(hd), log2_or_neg(tokens)};
  }
  typedef f32x4 Aux;
  __device__ __forceinline__ half_t* vt_ptr(int n, int m) const {  // &Vt[img][head][d][tok] for column n (>= 2D) and row m
    // ((img * heads + head) * hd + d) * tokens + tok  with  head * hd + d = c  and  heads * hd = D:  no head / d split is needed
    const int c = n - 2 * D;
    const int img = tok_sh >= 0 ? (m >> tok_sh) : m / tokens, tok = m - img * tokens;
    return Vt + ((long)img * D + c) * tokens + vt_pos(tok);
  }
  __device__ __forceinline__ bool direct(int n0) const { return n0 >= 2 * D; }  // V tiles: 32 consecutive tokens per lane group
  __device__ __forceinline__ Aux load(int, int n) const { return *(const f32x4*)(bias + n); }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& b) const {
    v += b;
    if (n < 2 * D) {
      half_t* dst = (n < D) ? (Q + (long)m * D + n) : (K + (long)m * D + (n - D));
      half4_t h = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
      *(half4_t*)dst = h;
    } else {
      half_t* dst = vt_ptr(n, m);  // n..n+3 stay inside one head (hd % 4 == 0)
      dst[0] = (half_t)v.x;
      dst[tokens] = (half_t)v.y;
      dst[2 * tokens] = (half_t)v.z;
      dst[3 * tokens] = (half_t)v.w;
    }
  }
  // A 256-column tile that lies inside Q or inside K is a plain "acc + bias -> fp16" tile of a [M, D] matrix: kernels that
  // ask get that epilogue (no per-store Q/K/V case distinction).  C is shifted so that C[m*D + n] is the right element.
  __device__ __forceinline__ bool plain_tile(int n0, int bn) const { return n0 + bn <= D || (n0 >= D && n0 + bn <= 2 * D); }
  __device__ __forceinline__ EpiBiasF16 plain(int n0) const { return EpiBiasF16{n0 < D ? Q : K - D, (long)D, bias}; }
  // Kernels that can compute a tile TRANSPOSED (operands swapped in the MFMA) hand V tiles over as rows of V^T:
  // v = C[m..m+3][n], four consecutive tokens of one (head, d) column -> one 8-byte store.  tokens % 4 == 0.
  __device__ __forceinline__ bool transposed(int n0) const { return n0 >= 2 * D; }
  __device__ __forceinline__ float load_t(int n) const { return bias[n]; }
  __device__ __forceinline__ void store_t(int n, int m, f32x4 v, float b) const {
    half4_t h = {(half_t)(v.x + b), (half_t)(v.y + b), (half_t)(v.z + b), (half_t)(v.w + b)};
    *(half4_t*)vt_ptr(n, m) = h;
  }
  __device__ __forceinline__ bool wide_t_ok() const { return (tokens & 15) == 0 && ((uintptr_t)Vt & 15) == 0; }
  // one 16-byte chunk of a V^T row: lo = tokens m .. m + 3, hi = tokens m + 8 .. m + 11 (m % 16 == 0 or 4: see vt_pos)
  __device__ __forceinline__ void store_t8(int n, int m, f32x4 lo, f32x4 hi, float b) const {
    half8_t h = {(half_t)(lo.x + b), (half_t)(lo.y + b), (half_t)(lo.z + b), (half_t)(lo.w + b),
                 (half_t)(hi.x + b), (half_t)(hi.y + b), (half_t)(hi.z + b), (half_t)(hi.w + b)};
    *(half8_t*)vt_ptr(n, m) = h;
  }
};

// QKV projection of the folded path: EpiQKV's layout with the row-affine correction instead of the bias (v carries the qkv bias)
struct EpiQKVMod {
  half_t* Q;
  half_t* K;
  half_t* Vt;
  const float* u;  // [rows][3D] (+ img * uv_stride)
  const float* v;
  long uv_stride;
  int D, hd, tokens;
  int tok_sh;
  RowStatSrc st;
  const float* rs;
  int m0;
  static constexpr bool rowstat = true;
  typedef EpiModF16::Aux Aux;
  __device__ __forceinline__ half_t* vt_ptr(int n, int m) const {
    const int c = n - 2 * D;
    const int img = tok_sh >= 0 ? (m >> tok_sh) : m / tokens, tok = m - img * tokens;
    return Vt + ((long)img * D + c) * tokens + vt_pos(tok);
  }
  __device__ __forceinline__ Aux load(int m, int n) const {
    const long o = (long)(m / tokens) * uv_stride + n;
    Aux a;
    a.u = *(const f32x4*)(u + o);
    a.v = *(const f32x4*)(v + o);
    return a;
  }
  __device__ __forceinline__ void store(int m, int n, f32x4 acc, const Aux& c) const {  // generic path (edge tiles of odd shapes only)
    const f32x2 ab = *(const f32x2*)(rs + 2 * (m - m0));
    const f32x4 o = row_affine4(ab.x, ab.y, acc, c.u, c.v);
    if (n < 2 * D) {
      half_t* dst = (n < D) ? (Q + (long)m * D + n) : (K + (long)m * D + (n - D));
      half4_t h = {(half_t)o.x, (half_t)o.y, (half_t)o.z, (half_t)o.w};
      *(half4_t*)dst = h;
    } else {
      half_t* dst = vt_ptr(n, m);
      dst[0] = (half_t)o.x;
      dst[tokens] = (half_t)o.y;
      dst[2 * tokens] = (half_t)o.z;
      dst[3 * tokens] = (half_t)o.w;
    }
  }
  __device__ __forceinline__ bool plain_tile(int n0, int bn) const { return n0 + bn <= D || (n0 >= D && n0 + bn <= 2 * D); }
  __device__ __forceinline__ EpiModF16 plain(int n0) const { return EpiModF16{n0 < D ? Q : K - D, (long)D, u, v, uv_stride, tokens, rs, m0}; }
  __device__ __forceinline__ bool transposed(int n0) const { return n0 >= 2 * D; }
  // column constants of a V^T row: (u[n], v[n]) of the tile's image
  __device__ __forceinline__ f32x2 load_t(int n) const {
    const long o = (long)(m0 / tokens) * uv_stride + n;
    return (f32x2){u[o], v[o]};
  }
  __device__ __forceinline__ void store_t(int n, int m, f32x4 acc, f32x2 c) const {  // four consecutive tokens m .. m + 3 of column n
    const f32x4 r0 = *(const f32x4*)(rs + 2 * (m - m0)), r1 = *(const f32x4*)(rs + 2 * (m - m0) + 4);  // (a, b) x 4 rows
    half4_t h = {(half_t)fma_v(r0.x, acc.x, fma_v(r0.y, c.x, c.y)), (half_t)fma_v(r0.z, acc.y, fma_v(r0.w, c.x, c.y)),
                 (half_t)fma_v(r1.x, acc.z, fma_v(r1.y, c.x, c.y)), (half_t)fma_v(r1.z, acc.w, fma_v(r1.w, c.x, c.y))};
    *(half4_t*)vt_ptr(n, m) = h;
  }
  __device__ __forceinline__ bool wide_t_ok() const { return (tokens & 15) == 0 && ((uintptr_t)Vt & 15) == 0; }
  __device__ __forceinline__ void store_t8(int n, int m, f32x4 lo, f32x4 hi, f32x2 c) const {  // lo = tokens m .. m + 3, hi = tokens m + 8 .. m + 11
    const float* r = rs + 2 * (m - m0);
    const f32x4 r0 = *(const f32x4*)r, r1 = *(const f32x4*)(r + 4), r2 = *(const f32x4*)(r + 16), r3 = *(const f32x4*)(r + 20);
    half8_t h = {(half_t)fma_v(r0.x, lo.x, fma_v(r0.y, c.x, c.y)), (half_t)fma_v(r0.z, lo.y, fma_v(r0.w, c.x, c.y)),
                 (half_t)fma_v(r1.x, lo.z, fma_v(r1.y, c.x, c.y)), (half_t)fma_v(r1.z, lo.w, fma_v(r1.w, c.x, c.y)),
                 (half_t)fma_v(r2.x, hi.x, fma_v(r2.y, c.x, c.y)), (half_t)fma_v(r2.z, hi.y, fma_v(r2.w, c.x, c.y)),
                 (half_t)fma_v(r3.x, hi.z, fma_v(r3.y, c.x, c.y)), (half_t)fma_v(r3.z, hi.w, fma_v(r3.w, c.x, c.y))};
    *(half8_t*)vt_ptr(n, m) = h;
  }
};

// ------------------------------------------------------------------ kernel
// default: epilogues ignore the batch index; batched epilogues provide a member batch(bz, stride).
template <class Epi>
__device__ __forceinline__ auto epi_batch(Epi& e, int bz, long bs, int) -> decltype(e.batch(bz, bs), void()) {
  e.batch(bz, bs);
}
template <class Epi>
__device__ __forceinline__ void epi_batch(Epi&, int, long, long) {}

// epilogues that need the raw MFMA fragment layout (lane = 32 consecutive m) for some column range provide direct(n0).
template <class Epi>
__device__ __forceinline__ auto epi_direct(const Epi& e, int n0, int) -> decltype(e.direct(n0)) {
  return e.direct(n0);
}
template <class Epi>
__device__ __forceinline__ bool epi_direct(const Epi&, int, long) {
  return false;
}

// epilogues with transposed(n0) / store_t(n, m, v): see EpiQKV
template <class Epi, class = void>
struct epi_has_transposed {
  static constexpr bool value = false;
};
template <class Epi>
struct epi_has_transposed<Epi, decltype((void)((const Epi*)nullptr)->transposed(0))> {
  static constexpr bool value = true;
};

template <class Epi, class = void>
struct epi_has_plain {
  static constexpr bool value = false;
};
template <class Epi>
struct epi_has_plain<Epi, decltype((void)((const Epi*)nullptr)->plain(0))> {
  static constexpr bool value = true;
};

template <class ASrc, class Epi>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(ASrc asrc, const half_t* __restrict__ W, long ldw, int M, int N, int K,
                                                          int tiles_n, Epi epi, long bsA, long bsW, long bsC) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // XCD-aware tile order: block b runs on XCD b%8; give every XCD a contiguous range of M-panels so an
  // activation panel is fetched into ONE L2 and reused by all N-tiles there.
  int bid = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * GEMM_BM, n0 = tile_n * GEMM_BN;
  const int wm = wave >> 1, wn = wave & 1;

  // batched GEMM (blockIdx.y): operand/result bases advance by the batch strides
  const int bz = blockIdx.y;
  asrc.init(bz, bsA);
  W += (long)bz * bsW;

  // ---- per-thread DMA source rows: 4 passes over the 128-row tile, 32 rows per pass
  typename ASrc::Row arow[4];
  const half_t* wrow[4];
  int cswz[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = p * 32 + (tid >> 3);
    arow[p] = asrc.row(m0 + r);
    const int n = n0 + r;
    wrow[p] = W + (long)(n < N ? n : N - 1) * ldw;
    cswz[p] = ((tid & 7) ^ ((r >> 1) & 7)) * 8;  // source k-offset (halves) of the chunk this lane fetches
  }
  const int nk = K / GEMM_BK;

  auto issue = [&](int kt, int stage) {
    char* sA = smem + stage * GEMM_STAGE_BYTES;
    char* sW = sA + GEMM_BM * GEMM_BK * 2;
    const int k0 = kt * GEMM_BK;
    asrc.begin_tile(kt, GEMM_BK);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      glds16(asrc.ptr(arow[p], cswz[p]), sA + (p * 256 + wave * 64) * 16);
      glds16(wrow[p] + k0 + cswz[p], sW + (p * 256 + wave * 64) * 16);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read offsets (bytes within a tile image), constant over K
  int a_off[2], w_off[2], a_key[2], w_key[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ra = wm * 64 + j * 32 + (lane & 31);
    a_off[j] = ra * 128;
    a_key[j] = (ra >> 1) & 7;
    const int rw = wn * 64 + j * 32 + (lane & 31);
    w_off[j] = rw * 128;
    w_key[j] = (rw >> 1) & 7;
  }
  const int chalf = lane >> 5;

  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nk) {
      issue(kt + 1, stage ^ 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this wave's 8 DMAs of tile kt have landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // ... and everyone else's
    asm volatile("" ::: "memory");
    const char* sA = smem + stage * GEMM_STAGE_BYTES;
    const char* sW = sA + GEMM_BM * GEMM_BK * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 2 + chalf;
      half8_t af[2], wf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        af[j] = *(const half8_t*)(sA + a_off[j] + ((c ^ a_key[j]) << 4));
        wf[j] = *(const half8_t*)(sW + w_off[j] + ((c ^ w_key[j]) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // all reads of this stage done before it is refilled
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: acc[i][j][4g+r] = C[m][n+r], m = ..+(lane&31), n = ..+8g+4*(lane>>5)
  epi_batch(epi, bz, bsC, 0);
  if (m0 + GEMM_BM <= M && n0 + GEMM_BN <= N) {  // interior tile: all loads first, then all stores
    typename Epi::Aux aux[2][2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          aux[j][i][g] = epi.load(m0 + wm * 64 + j * 32 + (lane & 31), n0 + wn * 64 + i * 32 + 8 * g + 4 * chalf);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + wm * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * chalf;
          f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          epi.store(m, n, v, aux[j][i][g]);
        }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + wm * 64 + j * 32 + (lane & 31);
      if (m >= M) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * chalf;
          if (n + 3 < N) {
            f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            epi.store(m, n, v, epi.load(m, n));
          }
        }
    }
  }
}

template <class ASrc, class Epi>
static inline int launch_gemm_tn(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int K, const Epi& epi,
                                 hipStream_t stream, int batch = 1, long bsA = 0, long bsW = 0, long bsC = 0) {
  if (!asrc_fits(asrc, 0)) return LFM_ERR_SHAPE;
  if (M <= 0 || N <= 0 || K <= 0 || (K % GEMM_BK) != 0 || (N % 4) != 0) return LFM_ERR_SHAPE;
  if ((ldw % 8) != 0 || ((uintptr_t)W & 15)) return LFM_ERR_ALIGN;
  const int tm = cdiv(M, GEMM_BM), tn = cdiv(N, GEMM_BN);
  static lfm_device_mask attr_set{0};  // one attribute call per instantiation and device
  const unsigned long long dbit = lfm_device_bit();
  if (lfm_device_todo(attr_set, dbit)) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<ASrc, Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    lfm_device_done(attr_set, dbit);
  }
  hipLaunchKernelGGL((gemm_tn_kernel<ASrc, Epi>), dim3(tm * tn, batch), dim3(256), GEMM_LDS_BYTES, stream, asrc, W, ldw, M, N, K, tn,
                     epi, bsA, bsW, bsC);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

// ------------------------------------------------------------------ split-K for small M (latency mode)
// --measure_time runs ONE latent: M = 256 rows, so a DiT GEMM has 16-64 tiles of 128x128 on 256 CUs and up to 64 K-tiles in
// sequence per tile (fc2) -- 3.07 ms per DiT-L/2 evaluation, 0.3 TB/s of weight streaming.  Split the K range over blockIdx.y:
// every slice writes an fp32 partial tile to a slab in the caller's workspace, and a second small kernel sums the slabs in a
// FIXED order and applies the real epilogue (deterministic, unlike atomics -- which were also measured 1.7x slower: device-scope
// atomics from 8 XCDs serialise at the memory side).
template <class Epi>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slabs, int S, long slice_stride, int M, int N, Epi epi) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N >> 2;
  if (idx >= (long)M * n4) return;
  const int m = (int)(idx / n4), n = (int)(idx - (long)m * n4) * 4;
  const float* p = slabs + (long)m * N + n;
  f32x4 v = *(const f32x4*)p;
  for (int s = 1; s < S; ++s) v += *(const f32x4*)(p + (long)s * slice_stride);
  epi.store(m, n, v, epi.load(m, n));
}

// Latency mode (round 3): the finish of a split-K gated-residual GEMM (proj, fc2) owns WHOLE rows when one block takes one row, so it can also be
// the LayerNorm-modulate that follows (DiT.py:129-130): X' = X + gate (sum of slabs + bias), written back, then A = fp16(LN(X') (1 + scale) + shift)
// for the next GEMM -- exact two-pass statistics, the row stays in registers.  Removes both ln_modulate launches of a block at batch 1 (a launch +
// one dependent HBM round trip each at this size).  Block = 256 threads, thread t owns columns 4 t + 1024 j.
#define SPLITK_LN_MAXJ 2  // N <= 2048
static __global__ __launch_bounds__(256) void splitk_finish_resid_ln_kernel(const float* __restrict__ slabs, int S, long slice_stride, int N, float* __restrict__ X,
                                                                     long ldx, const float* __restrict__ bias, const float* __restrict__ gate,
                                                                     long gate_stride, int tokens, half_t* __restrict__ A, const float* __restrict__ shift,
                                                                     const float* __restrict__ scale, long mod_stride) {
  __shared__ float red[4];
  const int m = blockIdx.x, img = m / tokens, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4 x[SPLITK_LN_MAXJ], sc1[SPLITK_LN_MAXJ], sh[SPLITK_LN_MAXJ];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < SPLITK_LN_MAXJ; ++j) {
    const int n = threadIdx.x * 4 + 1024 * j;
    x[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (n < N) {
      if (A) {  // the modulation rows are requested WITH the slabs (round 5): behind the reductions' barriers they were one more dependent round trip
        sc1[j] = *(const f32x4*)(scale + (long)img * mod_stride + n);
        sh[j] = *(const f32x4*)(shift + (long)img * mod_stride + n);
      }
      const float* p = slabs + (long)m * N + n;
      f32x4 v = *(const f32x4*)p;
      for (int q = 1; q < S; ++q) v += *(const f32x4*)(p + (long)q * slice_stride);
      x[j] = *(const f32x4*)(X + (long)m * ldx + n) + *(const f32x4*)(gate + (long)img * gate_stride + n) * (v + *(const f32x4*)(bias + n));
      *(f32x4*)(X + (long)m * ldx + n) = x[j];
      s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
    }
  }
  if (!A) return;
  s = wave_sum_dpp(s);  // DPP cross-lane network: six VALU adds instead of six ds_bpermute round trips
  if (lane == 0) red[wv] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)N;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < SPLITK_LN_MAXJ; ++j)
    if (threadIdx.x * 4 + 1024 * j < N) {
      x[j] -= mean;
      q += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
    }
  q = wave_sum_dpp(q);
  if (lane == 0) red[wv] = q;
  __syncthreads();
  const float rstd = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)N + 1e-6f);
#pragma unroll
  for (int j = 0; j < SPLITK_LN_MAXJ; ++j) {
    const int n = threadIdx.x * 4 + 1024 * j;
    if (n < N) {
      const f32x4 o = x[j] * rstd * (1.0f + sc1[j]) + sh[j];
      const half4_t h = {(half_t)o.x, (half_t)o.y, (half_t)o.z, (half_t)o.w};
      *(half4_t*)(A + (long)m * N + n) = h;
    }
  }
}

// slices for a GEMM of this size: only when the 128x128 tiling leaves most of the chip idle; slices >= 128 deep; slabs fit
// max_tiles / max_wg: the 3x3 convolutions of the UNets' 16x16 maps (celeb512 at batch 32: 8192 rows x 512 columns = 256 tiles = ONE four-wave
// workgroup per CU, K = 4608 .. 9216) also split, into two slices = two workgroups per CU (90 us each unsplit); the DiT linears keep 64 / 256
// (their K is 1024 .. 4096: the finish kernel would eat the gain).
static inline int splitk_slices(int M, int N, int K, size_t slab_bytes, int max_tiles = 64, int max_wg = 256) {
  const long tiles = (long)cdiv(M, 128) * cdiv(N, 128);
  if (tiles > max_tiles || (K % 128) != 0 || (N % 4) != 0) return 1;
  int s = 1;
  while (tiles * (s * 2) <= max_wg && K / (s * 2) >= 128 && (K / (s * 2)) % 64 == 0 && (size_t)(s * 2) * M * N * 4 <= slab_bytes) s *= 2;
  return s;
}

// returns LFM_OK after launching both kernels, or 1 if split-K does not apply (the caller then takes the ordinary path).
// Slices are addressed through the batch index: asrc.init(bz, ks) makes slice bz start at k = bz * ks (row-major A advances its
// pointer; an implicit-GEMM convolution source starts its tap / channel walk there), W advances by ks columns.
// Round 5: slices of DEEP small-map problems (the UNets' 8x8 / 16x16-map 3x3 convolutions: M = 2048 .. 8192 pixel rows, K = 9 Cin = 4608 .. 18432) on the
// 256x256 eight-wave kernel instead of the 128x128 one: its main loop runs at ~1.3 PFLOP/s where the 128x128 loop reaches 0.45-0.75
// (profiles/r04_conv_small_maps_probe.txt), and a slice that is >= 24 K-tiles deep amortises its longer prologue / epilogue.  S = the largest divisor
// of the K-tile count that keeps tiles x S <= 256 workgroups (one per CU) with slices >= `min_tiles_k` K-tiles.  0 = does not apply.
int lfm_gemm_debug_flags();
static inline int splitk256_slices(int M, int N, int K, size_t slab_bytes, int min_tiles_k = 24) {
  if (M < 2048 || N < 256 || (N % 256) != 0 || (K % 64) != 0 || (lfm_gemm_debug_flags() & 65536)) return 0;  // flag 65536: the 128x128 slices (A/B)
  const long tiles = (long)cdiv(M, 256) * (N / 256);
  if (tiles > 128) return 0;
  const int kt = K / 64;
  int best = 0;
  for (int sl = 2; sl <= 16; ++sl)
    if (kt % sl == 0 && tiles * sl <= 256 && kt / sl >= min_tiles_k && (size_t)sl * M * N * 4 <= slab_bytes) best = sl;
  // at least half the CUs busy, or the many small 128x128 slices win (tools/splitk256_probe.py: 512 -> 512 channels at 8x8 x 32 images = 16 tiles x 3
  // slices = 48 workgroups: 47.8 us against 27.0 us; 768 -> 768 at 8x8 x 64 images = 144 workgroups: 66.6 against 79.1 us)
  return tiles * best >= 128 ? best : 0;
}
template <class ASrc, class Epi>
static inline int launch_gemm_splitk256(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int ks, const EpiSlabF32& e, hipStream_t stream, int S);

template <class ASrc, class Epi>
static inline int launch_gemm_splitk_src(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, float* slab,
                                         size_t slab_bytes, hipStream_t stream, int max_tiles = 64, int max_wg = 256) {
  if (!slab || lfm_gemm_selected_v1_ok() == 0) return 1;
  if (const int S2 = splitk256_slices(M, N, K, slab_bytes)) {
    const int ks = K / S2;
    const long stride = (long)M * N;
    const int rc = launch_gemm_splitk256<ASrc, Epi>(asrc, W, ldw, M, N, ks, EpiSlabF32{slab, (long)N, stride}, stream, S2);
    if (rc != 1) {
      if (rc) return rc;
      const long work = (long)M * (N >> 2);
      hipLaunchKernelGGL((splitk_finish_kernel<Epi>), dim3((unsigned)cdiv(work, 256)), dim3(256), 0, stream, slab, S2, stride, M, N, epi);
      LFM_CHECK_LAUNCH();
      return LFM_OK;
    }
  }
  const int S = splitk_slices(M, N, K, slab_bytes, max_tiles, max_wg);
  if (S < 2) return 1;
  const int ks = K / S;
  const long stride = (long)M * N;
  int rc = launch_gemm_tn(asrc, W, ldw, M, N, ks, EpiSlabF32{slab, (long)N, stride}, stream, S, ks, ks, 0);
  if (rc) return rc;
  const long work = (long)M * (N >> 2);
  hipLaunchKernelGGL((splitk_finish_kernel<Epi>), dim3((unsigned)cdiv(work, 256)), dim3(256), 0, stream, slab, S, stride, M, N, epi);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
template <class Epi>
static inline int launch_gemm_splitk(const half_t* A, long lda, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, float* slab,
                                     size_t slab_bytes, hipStream_t stream) {
  return launch_gemm_splitk_src(ASrcRowMajor{A, lda, M, 0}, W, ldw, M, N, K, epi, slab, slab_bytes, stream);
}

// split-K gated-residual GEMM whose finish is also the following LayerNorm-modulate (A_out may be null: plain finish).  Returns 1 if split-K does
// not apply to the shape (the caller takes the ordinary GEMM + ln_modulate path).
static inline int launch_gemm_splitk_resid_ln(const half_t* A, long lda, const half_t* W, long ldw, int M, int N, int K, const EpiGateResidF32& e,
                                              half_t* A_out, const float* shift, const float* scale, long mod_stride, float* slab, size_t slab_bytes,
                                              hipStream_t stream) {
  if (!slab || lfm_gemm_selected_v1_ok() == 0 || N > 1024 * SPLITK_LN_MAXJ || (N % 4) != 0) return 1;
  const int S = splitk_slices(M, N, K, slab_bytes);
  if (S < 2) return 1;
  const int ks = K / S;
  const long stride = (long)M * N;
  int rc = launch_gemm_tn(ASrcRowMajor{A, lda, M, 0}, W, ldw, M, N, ks, EpiSlabF32{slab, (long)N, stride}, stream, S, ks, ks, 0);
  if (rc) return rc;
  hipLaunchKernelGGL(splitk_finish_resid_ln_kernel, dim3(M), dim3(256), 0, stream, slab, S, stride, N, e.X, e.ldx, e.bias, e.gate, e.gate_stride, e.tokens, A_out,
                     shift, scale, mod_stride);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
