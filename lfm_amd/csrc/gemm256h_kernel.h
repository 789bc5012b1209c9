// 256x256 quadrant-phased MFMA GEMM on v_mfma_f32_16x16x32_f16 (v5):  C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue).
//
// Why (tools/ubench/mfma_shape.hip, profiles/r02_mfma_shape.txt): the GEMMs of this path run under the board POWER cap, not under an
// issue or bandwidth limit -- and on random fp16 operands a pure stream of 16x16x32 MFMAs sustains 1930 TFLOP/s where the 32x32x16
// stream that v1-v4 use sustains 1660 (both 2450 on zeros): the small shape moves 20 % fewer register-file bytes per flop (4
// accumulator registers per instruction instead of 16).  Energy per flop is the lever under a power cap, so this generation keeps
// v3's structure unchanged -- LDS image, piece-granular LDS-DMA ring, quadrant phases, one barrier per phase, counted waits,
// ping-pong wave groups, tile order (gemm256q_kernel.h) -- and changes only the instruction and what follows from its fragment maps:
//
//   operand fragments  lane l holds 8 consecutive k of row (l & 15) at k-group (l >> 4): one ds_read_b128 of logical chunk
//                      4*ks + (l >> 4) of a 128-byte LDS row (ks = 0, 1 per 64-deep K-tile); the same 12 / 4 / 8 / 0 reads per phase as v3;
//   accumulators       a wave's 128 x 64 block = 8 (m) x 4 (n) tiles of f32x4; issued with W as the A operand, so lane l owns
//                      C[m = 16 i + (l & 15)][n = 16 j + 4 (l >> 4) + r], r = 0..3: four consecutive n per lane as before;
//   epilogue           the same LDS-transposed row-major hand-over (shared Epi interface), with the scratch filled from the new map.
#pragma once
#include "gemm256n_kernel.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---- epilogue for the 16x16 accumulator map.  acc[i][j]: tile i (16 rows) x j (16 columns) of the wave's 128 x 64 block.
// fill32(I, scr): rows 32 I .. 32 I + 31 of the block -> scratch [32][64 + pad] fp32, row stride 272 B (as g256_epilogue_rows)
// Measurement only (lfm_gemm_select flag 2 with kernel 5): waves 0 and 4 of the tile at row 0, column (flags >> 21) & 15 stamp s_memtime at the end of the K loop and after the
// scratch fill / read-back / store issue of each of the four 32-row blocks of the row-major epilogue (g256q_trace, lfm_gemm_trace_read).
template <bool TRACE>
__device__ __forceinline__ void g256h_stamp(bool tr, int g, int wn, int lane, int slot) {
  if constexpr (TRACE) {
    if (tr && wn == 0 && lane == 0) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      g256q_trace[g][slot] = t;
    }
  }
}

// epilogues with per-row operands in LDS provide row_aux(m) -> f32x2 and store8r(m, n, lo, hi, aux_lo, aux_hi, row_aux)
template <class Epi, class = void>
struct epi_has_row_aux {
  static constexpr bool value = false;
};
template <class Epi>
struct epi_has_row_aux<Epi, decltype((void)&Epi::row_aux)> {
  static constexpr bool value = true;
};

template <int BN, bool TRACE = false, class Epi>
__device__ __forceinline__ void g256h_epilogue_rows(f32x4_t (&acc)[8][4], char* smem, const Epi& epi, int m0, int n0, int M, int N, int g, int wn,
                                                    int lane, int wave, bool narrow, bool tr = false) {
  char* scr = smem + wave * (32 * 272);
  const bool interior = (m0 + G256_BM <= M) && (n0 + BN <= N);
  constexpr bool COL = epi_column_aux<Epi>::value;  // bias-only auxiliary operand: loaded once per tile, ahead of the first store
  const int l15 = lane & 15, l4 = lane >> 4;
  auto fill32 = [&](int I) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4_t*)(scr + (h2 * 16 + l15) * 272 + (j * 16 + l4 * 4) * 4) = acc[2 * I + h2][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  if constexpr (epi_has_store8<Epi>::value) {
    if (!narrow && epi.wide_ok()) {
      const int rrow = lane >> 3, rcol = lane & 7;
      typename Epi::Aux cl, ch;
      if constexpr (COL) {
        if (interior) {
          cl = epi.load(m0, n0 + wn * 64 + rcol * 8);
          ch = epi.load(m0, n0 + wn * 64 + rcol * 8 + 4);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fill32(i);
        g256h_stamp<TRACE>(tr, g, wn, lane, 1 + 4 * i);
        f32x4 lo[4], hi[4];
        const int mb = m0 + g * 128 + i * 32 + rrow, n = n0 + wn * 64 + rcol * 8;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          lo[ps] = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + rcol * 32);
          hi[ps] = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + rcol * 32 + 16);
        }
        if constexpr (epi_has_row_aux<Epi>::value) {  // per-row operands of the epilogue (LDS), fetched with the read-back: one wait covers both
          if (COL && interior) {
            f32x2 ra[4];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) ra[ps] = epi.row_aux(mb + ps * 8);
#ifdef LFM_EXP_WAIT_ALL  // (experiment build) every LDS read of the pass has landed, plus 16 idle cycles, before the first VALU instruction that consumes one
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) epi.store8r(mb + ps * 8, n, lo[ps], hi[ps], cl, ch, ra[ps]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
          }
        }
        if constexpr (TRACE) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          g256h_stamp<TRACE>(tr, g, wn, lane, 2 + 4 * i);
        }
        if (COL && interior) {
          if constexpr (COL) {
            g256h_stamp<TRACE>(tr, g, wn, lane, 3 + 4 * i);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) epi.store8(mb + ps * 8, n, lo[ps], hi[ps], cl, ch);
            g256h_stamp<TRACE>(tr, g, wn, lane, 4 + 4 * i);
          }
        } else if (interior) {
          typename Epi::Aux al[4], ah[4];
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            al[ps] = epi.load(mb + ps * 8, n);
            ah[ps] = epi.load(mb + ps * 8, n + 4);
          }
          if constexpr (TRACE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (trace build only) the auxiliary loads have returned
            g256h_stamp<TRACE>(tr, g, wn, lane, 3 + 4 * i);
          }
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) epi.store8(mb + ps * 8, n, lo[ps], hi[ps], al[ps], ah[ps]);
          g256h_stamp<TRACE>(tr, g, wn, lane, 4 + 4 * i);
        } else {
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int m = mb + ps * 8;
            if (m >= M) continue;
            if (n + 7 < N) epi.store8(m, n, lo[ps], hi[ps], epi.load(m, n), epi.load(m, n + 4));
            else if (n + 3 < N) epi.store(m, n, lo[ps], epi.load(m, n));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      return;
    }
  }
  const int rrow = lane >> 4, rcol = lane & 15;
  typename Epi::Aux cx;
  if constexpr (COL) {
    if (interior) cx = epi.load(m0, n0 + wn * 64 + rcol * 4);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    fill32(i);
    g256h_stamp<TRACE>(tr, g, wn, lane, 1 + 4 * i);
    f32x4 v[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) v[ps] = *(const f32x4*)(scr + (ps * 4 + rrow) * 272 + rcol * 16);
    if constexpr (TRACE) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      g256h_stamp<TRACE>(tr, g, wn, lane, 2 + 4 * i);
    }
    const int mb = m0 + g * 128 + i * 32 + rrow, n = n0 + wn * 64 + rcol * 4;
    if (COL && interior) {
      if constexpr (COL) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) epi.store(mb + ps * 4, n, v[ps], cx);
        g256h_stamp<TRACE>(tr, g, wn, lane, 4 + 4 * i);
      }
    } else if (interior) {
      typename Epi::Aux aux[8];
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) aux[ps] = epi.load(mb + ps * 4, n);
      if constexpr (TRACE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g256h_stamp<TRACE>(tr, g, wn, lane, 3 + 4 * i);
      }
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) epi.store(mb + ps * 4, n, v[ps], aux[ps]);
      g256h_stamp<TRACE>(tr, g, wn, lane, 4 + 4 * i);
    } else if (n + 3 < N) {
#pragma unroll
      for (int ps = 0; ps < 8; ++ps)
        if (mb + ps * 4 < M) epi.store(mb + ps * 4, n, v[ps], epi.load(mb + ps * 4, n));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// sum over the 16 lanes of a DPP row (every lane of the row receives it): quad swaps, then two row rotations
__device__ __forceinline__ float g256h_row16_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x124>(v);
  v = dpp_add<0x128>(v);
  return v;
}

// Producer epilogue of the folded LayerNorm-modulate (EpiGateResidMod, gemm_kernel.h): the gated-residual read-modify-write of X, plus the consumer
// GEMM's A operand A' = fp16((X' - c)(1 + scale)) and this tile's per-row partials (sum X', sum (X' - c)^2).  Interior tiles of ONE image only (the
// host guarantees it).  Row-major hand-over as g256h_epilogue_rows (4-column form): per pass a lane owns rows mb + 4 ps (ps = 0..7), columns n .. n + 3;
// the 16 lanes that share a row are one DPP row, so a row's sums over the wave's 64 columns cost four DPP adds each.
template <class Epi>
__device__ __forceinline__ void g256h_epilogue_mod(f32x4_t (&acc)[8][4], char* smem, const Epi& epi, int m0, int n0, int tile_n, int N, int g, int wn,
                                                   int lane, int wave) {
  char* scr = smem + wave * (32 * 272);
  float* red = (float*)(smem + 8 * 32 * 272);           // [g][wn][128 rows][2]: 8 KiB behind the eight scratch areas
  float* cen_s = (float*)(smem + 8 * 32 * 272 + 8192);  // [256 rows] centring constants of the tile's rows
  const int l15 = lane & 15, l4 = lane >> 4;
  const int rrow = lane >> 4, rcol = lane & 15;
  const int n = n0 + wn * 64 + rcol * 4;
  const int img = m0 / epi.tokens;
  if (threadIdx.x < 256) cen_s[threadIdx.x] = epi.cen[m0 + threadIdx.x];
  const f32x4 bias = *(const f32x4*)(epi.bias + n);
  const f32x4 gate = *(const f32x4*)(epi.gate + (long)img * epi.gate_stride + n);
  const f32x4 sc1 = *(const f32x4*)(epi.scale + (long)img * epi.mod_stride + n) + 1.0f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4_t*)(scr + (h2 * 16 + l15) * 272 + (j * 16 + l4 * 4) * 4) = acc[2 * i + h2][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int rl = g * 128 + i * 32 + rrow;  // + 4 ps: row inside the tile
    f32x4 xo[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) xo[ps] = *(const f32x4*)(epi.X + (long)(m0 + rl + ps * 4) * epi.ldx + n);
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const f32x4 v = *(const f32x4*)(scr + (ps * 4 + rrow) * 272 + rcol * 16);
      const float c = cen_s[rl + ps * 4];
      const f32x4 xn = xo[ps] + gate * (v + bias);
      *(f32x4*)(epi.X + (long)(m0 + rl + ps * 4) * epi.ldx + n) = xn;
      const f32x4 d = xn - c;
      const f32x4 ap = d * sc1;
      const half4_t h = {(half_t)ap.x, (half_t)ap.y, (half_t)ap.z, (half_t)ap.w};
      *(half4_t*)(epi.A + (long)(m0 + rl + ps * 4) * N + n) = h;
      const float sx = g256h_row16_sum((xn.x + xn.y) + (xn.z + xn.w));
      const float sq = g256h_row16_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w));
      if (rcol == 0) {
        float* dst = red + (((g * 4 + wn) * 128) + i * 32 + ps * 4 + rrow) * 2;
        dst[0] = sx;
        dst[1] = sq;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // tile-level row sums (fixed order over the four waves) -> this tile's slot of the row's partials
  if (wn == 0) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = lane + 64 * rr;
      float sx = 0.f, sq = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        sx += red[((g * 4 + w4) * 128 + r) * 2];
        sq += red[((g * 4 + w4) * 128 + r) * 2 + 1];
      }
      *(f32x2*)(epi.part + ((long)(m0 + g * 128 + r) * epi.tiles_n + tile_n) * 2) = (f32x2){sx, sq};
    }
  }
}

// Consumer prologue of the folded LayerNorm-modulate: (a, b) = (rstd, -rstd (mu - c)) of the tile's 256 rows -> LDS rs[256][2] (above the operand
// ring); tile column 0 publishes mu[m] as the next producer's centring constant.  Two halves around the K loop's first DMAs: the loads are issued
// BEFORE them (vmcnt returns in order: a load issued behind the twelve prologue DMAs would make its consumer wait for all of them -- measured
// +1.3 us per tile), the arithmetic runs while they fly.
#define G256H_MAX_PARTS 5  // residual width <= 1280
struct G256hRowStatRegs {
  f32x2 p[G256H_MAX_PARTS];
  float c;
};
template <class Epi>
__device__ __forceinline__ G256hRowStatRegs g256h_rowstat_load(const Epi& epi, int m0, int M) {
  G256hRowStatRegs r;
  const int m = m0 + (int)threadIdx.x < M ? m0 + (int)threadIdx.x : M - 1;
  if (threadIdx.x < 256) {
    const f32x2* pp = (const f32x2*)(epi.st.part + (long)m * epi.st.tiles_p * 2);
#pragma unroll
    for (int t = 0; t < G256H_MAX_PARTS; ++t) r.p[t] = t < epi.st.tiles_p ? pp[t] : (f32x2){0.f, 0.f};
    r.c = epi.st.cen_in[m];
  }
  return r;
}
template <class Epi>
__device__ __forceinline__ void g256h_rowstat_finish(Epi& epi, const G256hRowStatRegs& r, char* smem, int m0, int M, int tile_n) {
  float* rs = (float*)(smem + G256Q_LDS_BYTES);
  if (threadIdx.x < 256) {
    float sx = 0.f, sq = 0.f;
#pragma unroll
    for (int t = 0; t < G256H_MAX_PARTS; ++t) {  // fixed order
      sx += r.p[t].x;
      sq += r.p[t].y;
    }
    const float mu = sx * epi.st.inv_n, dl = mu - r.c;
    const float var = fmaxf(sq * epi.st.inv_n - dl * dl, 0.f);
    const float rstd = rsqrtf(var + epi.st.eps);
    *(f32x2*)(rs + 2 * threadIdx.x) = (f32x2){rstd, -rstd * dl};
    if (tile_n == 0 && m0 + (int)threadIdx.x < M) epi.st.cen_out[m0 + threadIdx.x] = mu;
  }
  epi.rs = rs;
  epi.m0 = m0;
}

// One 128 x 64 accumulator block (rows g * 128 .., columns wn * 64 .. of the tile at m0, n0) through the epilogue `epi`; the wave's private scratch is
// slot `wave`.  Shared by the 8-wave kernel below (one block per wave) and the 4-wave kernel of gemm256w_kernel.h (two blocks per wave).
template <int BN, bool TRACE = false, class Epi>
__device__ __forceinline__ void g256h_epilogue_body(f32x4_t (&acc)[8][4], char* smem, Epi& epi, int m0, int n0, int M, int N, int g, int wn, int lane,
                                                    int wave, int dbg, bool swapped, bool tr) {
  const int l15 = lane & 15, l4 = lane >> 4;
  if constexpr (epi_has_transposed<Epi>::value) {
    // The K loop ran this tile with the MFMA operands swapped: acc[i][j][r] = C[m = 16 i + 4 l4 + r][n = 16 j + l15], FOUR CONSECUTIVE m per
    // lane.  Scratch rows = 32 columns n (tiles 2 J, 2 J + 1), scratch columns = 64 rows m (tiles 4 ih .. 4 ih + 3); read back row-major:
    // a lane gets 8 (or 4) consecutive m of one n -> epi.store_t8 / store_t.
    if (swapped) {
      char* scr = smem + wave * (32 * 272);
      const bool wide = !(dbg & 1024) && epi.wide_t_ok();
      // the per-column bias of every pass, loaded ahead of the first store (a load issued after stores waits for them: vmcnt is in order)
      typedef decltype(epi.load_t(0)) AuxT;  // float (a bias) or (u, v) of the folded path
      AuxT bt[2][4];
      if (wide) {
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int n = n0 + wn * 64 + J * 32 + (lane >> 3) + ps * 8;
            bt[J][ps] = n < N ? epi.load_t(n) : AuxT{};
          }
      }
#pragma unroll
      for (int J = 0; J < 2; ++J) {
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
#pragma unroll
          for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
              *(f32x4_t*)(scr + (j2 * 16 + l15) * 272 + (i4 * 16 + l4 * 4) * 4) = acc[4 * ih + i4][2 * J + j2];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          g256h_stamp<TRACE>(tr, g, wn, lane, 1 + 4 * (2 * J + ih));
          // wide stores: a lane owns tokens ml .. ml + 3 and ml + 8 .. ml + 11 of a 16-token group (one 16-byte chunk of the permuted V^T row: vt_pos)
          const int ml = 16 * ((lane & 7) >> 1) + 4 * (lane & 1);
          if (wide && m0 + G256_BM <= M && n0 + BN <= N) {  // interior tile: no per-store bounds checks
            const int rrow = lane >> 3;
            const int nb = n0 + wn * 64 + J * 32 + rrow, m = m0 + g * 128 + ih * 64 + ml;
            f32x4 lo[4], hi[4];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
              lo[ps] = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + ml * 4);
              hi[ps] = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + ml * 4 + 32);
            }
            if constexpr (TRACE) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              g256h_stamp<TRACE>(tr, g, wn, lane, 2 + 4 * (2 * J + ih));
            }
            if (!(TRACE && (dbg & 131072))) {  // (trace build, flag 131072: the pass without its stores)
#pragma unroll
              for (int ps = 0; ps < 4; ++ps) epi.store_t8(nb + ps * 8, m, lo[ps], hi[ps], bt[J][ps]);
            }
          } else if (wide) {
            const int rrow = lane >> 3;
            const int nb = n0 + wn * 64 + J * 32 + rrow, m = m0 + g * 128 + ih * 64 + ml;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
              const f32x4 lo = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + ml * 4);
              const f32x4 hi = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + ml * 4 + 32);
              const int n = nb + ps * 8;
              if (n >= N) continue;
              const AuxT b = bt[J][ps];
              if (m + 11 < M) epi.store_t8(n, m, lo, hi, b);
              else {
                if (m + 3 < M) epi.store_t(n, m, lo, b);
                // (the hi half -- tokens m + 8 .. m + 11 -- lies beyond M here; wide_t_ok() implies M % 16 == 0, so this branch only trims whole tails)
              }
            }
          } else {
            const int rrow = lane >> 4, rcol = lane & 15;
            const int nb = n0 + wn * 64 + J * 32 + rrow, m = m0 + g * 128 + ih * 64 + rcol * 4;
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
              const f32x4 v = *(const f32x4*)(scr + (ps * 4 + rrow) * 272 + rcol * 16);
              const int n = nb + ps * 4;
              if (n < N && m + 3 < M) epi.store_t(n, m, v, epi.load_t(n));
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          g256h_stamp<TRACE>(tr, g, wn, lane, 4 + 4 * (2 * J + ih));
        }
      }
      g256h_stamp<TRACE>(tr, g, wn, lane, 17);
      return;
    }
  }
  if (epi_direct(epi, n0, 0)) {  // epilogues that want the fragment layout: (m, n..n+3) per lane straight from the accumulators
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + g * 128 + i * 16 + l15;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + l4 * 4;
        if (n + 3 < N) {
          f32x4 v = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          epi.store(m, n, v, epi.load(m, n));
        }
      }
    }
    return;
  }
  if constexpr (epi_has_plain<Epi>::value) {
    if (epi.plain_tile(n0, BN)) {
      auto pe = epi.plain(n0);
      g256h_epilogue_rows<BN, TRACE>(acc, smem, pe, m0, n0, M, N, g, wn, lane, wave, (dbg & 1024) != 0, tr);
      return;
    }
  }
  g256h_epilogue_rows<BN, TRACE>(acc, smem, epi, m0, n0, M, N, g, wn, lane, wave, (dbg & 1024) != 0, tr);
  if constexpr (epi_has_finish_tile<Epi>::value) epi.finish_tile(m0, n0, g, wn, lane);
  g256h_stamp<TRACE>(tr, g, wn, lane, 17);
}

template <int BN, bool TRACE = false, class Epi>
__device__ __forceinline__ void g256h_epilogue(f32x4_t (&acc)[8][4], char* smem, Epi& epi, int m0, int n0, int M, int N, int g, int wn, int lane,
                                               int wave, int bz, long bsC, int dbg, bool swapped) {
  const bool tr = TRACE && m0 == 0 && n0 == ((dbg >> 21) & 15) * BN && bz == 0;  // the stamped tile: row 0, column (flags >> 21) & 15
  epi_batch(epi, bz, bsC, 0);
  g256h_stamp<TRACE>(tr, g, wn, lane, 0);
  if (dbg & 4) return;  // ablation: no epilogue
  if constexpr (epi_is_producer_mod<Epi>::value) {
    g256h_epilogue_mod(acc, smem, epi, m0, n0, n0 / BN, N, g, wn, lane, wave);
    return;
  }
  g256h_epilogue_body<BN, TRACE>(acc, smem, epi, m0, n0, M, N, g, wn, lane, wave, dbg, swapped, tr);
}

// Round 3, where the main loop goes (tools/mainloop_ablation.py on an LFM_MEASURE build, profiles/r03_mainloop_ablation.txt; epilogue off, fc2 shape
// M 16384 x N 1024 x K 4096): full loop 97.5 us (1410 TFLOP/s); without the LDS-DMA issues 82.4; without the fragment reads 69.9; with neither 60.6
// (2267 TFLOP/s, constant operands); barriers: free (60.0 without them, 60.1 with one per K-tile); static priority for the second wave group: WORSE
// (110 us); accumulators pinned to D = C registers through inline-asm MFMAs: nothing (the compiler rotates them through D != C register tuples, which
// costs nothing).  So the LOAD part of a phase (~450-600 cycles: six fragment reads on average + two LDS-DMA issues of 60-185 cycles each) is what
// keeps the matrix pipe at 62 % -- a phase is 256 + L, not 512.  Tried on that evidence and NOT kept: a software-pipelined loop in which every wave
// issues the next step's six reads and its two LDS-DMAs between its own MFMAs (in-place refill of the A fragments, one counted lgkmcnt(5) per group
// of four MFMAs; bit-identical results).  With both wave groups on the same instruction stream all eight waves hit the LDS-DMA issue together and
// the pipe starved (fc2 122 vs 116 us); staggering the groups with two copies of the loop did not fit the register allocator's 256 VGPRs (136-168 B
// of scratch, 180+ us); staggering only the LDS-DMA issue behind a uniform branch (one loop copy, 44 B of scratch) was as slow as un-staggered.  The
// same ablation on the pipelined loop (profiles/r03_mainloop_ablation_incl_pipelined.txt) settles it: its bare MFMA + waits + barriers skeleton runs
// in 69.6 us, the loop in 116.3, without the LDS-DMAs in 92.4 -- reads and DMAs cost as much BETWEEN the MFMAs as they do in a LOAD part of their
// own (~30 cycles of the issuing wave per ds_read_b128, 60-185 per LDS-DMA, and the partner wave does not win that time back).  With a 128x64 wave
// tile the loop needs 0.375 fragment reads and 0.125 LDS-DMAs per MFMA; only a larger wave tile (fewer operand bytes per MFMA) changes that.  Kept from it: ASrcRowMajor rows as 32-bit
// offsets from a uniform base (SGPR base + VGPR offset LDS-DMAs), which took this kernel from 256 VGPRs + 12 B of scratch to 252 VGPRs and none.
// ABL (LFM_MEASURE builds only; results are garbage, timings are the point): 1 = no LDS-DMA after the prologue, 2 = no fragment reads,
// 3 = neither (the bare MFMA stream + barriers), 4 = static priority (s_setprio 1 for the second wave group, no per-phase flips),
// 5 = 3 without the per-phase barriers (the bare MFMA stream), 6 = 3 with one barrier per K-tile instead of per phase,
// 7 = 5 with the accumulators pinned (inline-asm MFMA, D = C), 8 = the full kernel with pinned accumulators
// OPT (round 4): bit 0 = LDS-DMAs through buffer resources (ASrc::buffer_form sources; byte offsets < 2^31, checked at launch), bit 1 = a LOAD part
// issues its first LDS-DMA BEFORE its fragment reads and the second one after them (a wave's VMEM instructions serialise at ~110 cycles each --
// tools/ubench/ldsdma_rate.hip, one wave: 105-116 cycles per 1-KiB instruction in every addressing form -- so the second of two back-to-back issues
// waits for the first; the reads in between cover that wait).
// Measured (tools/dma_opt_probe.py, profiles/r04_dma_opt_probe.txt; bit-identical results): bit 0 -0.6 .. -2.1 us on the main loops of both 256x256 kernels
// (252 -> 218 VGPRs in this one) = the default; bit 1 +1.5 .. +3 us = rejected.
// OPT bit 2 (round 6): the pieces' LDS-DMAs placed against the LOAD parts' fragment reads (none | B1 | A1 | A0 B0 instead of B1 | A1 | A0 | B0; counted vmcnt 4 4 4 6):
// -0.2 % per DiT-L/2 forward, bit-identical (profiles/r06_dma_phase_balance.txt) -- not the default.
#ifndef G256H_DEFAULT_OPT
#define G256H_DEFAULT_OPT 1
#endif
template <class ASrc, class Epi, bool TRACE = false, int ABL = 0, int OPT = G256H_DEFAULT_OPT>
__global__ __launch_bounds__(512) void gemm256h_tn_kernel(ASrc asrc, const half_t* __restrict__ W, long ldw, int M, int N, int K, int tiles_n,
                                                           Epi epi, long bsA, long bsW, long bsC, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wn = wave & 3;

  int tile_m, tile_n;
  g256_tile_order(blockIdx.x, gridDim.x, tiles_n, dbg, tile_m, tile_n);
  const int m0 = tile_m * G256_BM, n0 = tile_n * G256_BN;
  bool swapped = false;
  if constexpr (epi_has_transposed<Epi>::value) swapped = epi.transposed(n0);
  const int bz = blockIdx.y;
  asrc.init(bz, bsA);
  W += (long)bz * bsW;

  // ---- DMA sources: identical to v3 (piece = 128 local rows x 128 B; thread tid stages chunks tid and 512 + tid)
  typename ASrc::Row arow[2][2];  // [sub][pass]
  const half_t* wrow[2][2];
  const int cswz = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      arow[s][p] = asrc.row(m0 + p * 128 + s * 64 + (tid >> 3));
      const int n = n0 + (p * 2 + (tid >> 8)) * 64 + s * 32 + ((tid >> 3) & 31);
      wrow[s][p] = W + (long)(n < N ? n : N - 1) * ldw + cswz;
    }
  const int nk = K / G256Q_BK;
  const int dma_off = wave * 1024;
  bool dma_on = true;  // (ABL 1 / 3 turn it off after the prologue)
  constexpr bool BUFDMA = (OPT & 1) != 0 && asrc_has_buffer<ASrc>::value;
  __amdgpu_buffer_rsrc_t rsa, rsw;
  unsigned avoff[2][2], wvoff[2][2];
  if constexpr (BUFDMA) {
    rsa = asrc.rsrc();
    rsw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, -1, 0x00020000);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        avoff[s][p] = asrc.voff(arow[s][p], cswz);
        const int n = n0 + (p * 2 + (tid >> 8)) * 64 + s * 32 + ((tid >> 3) & 31);
        wvoff[s][p] = ((unsigned)(n < N ? n : N - 1) * (unsigned)ldw + (unsigned)cswz) * 2u;
      }
  }
  // the two LDS-DMAs of a piece: P = 0 / 1 / 2 = both / the first / the second
  auto issue_a = [&](int s, char* slot, int P = 0) {
    if ((ABL == 1 || ABL == 3 || (ABL >= 5 && ABL <= 7)) && !dma_on) return;
    if constexpr (BUFDMA) {
      if (P != 2) glds16_buf(rsa, avoff[s][0], asrc.soff(), slot + dma_off);
      if (P != 1) glds16_buf(rsa, avoff[s][1], asrc.soff(), slot + 8192 + dma_off);
    } else {
      if (P != 2) glds16(asrc.ptr(arow[s][0], cswz), slot + dma_off);
      if (P != 1) glds16(asrc.ptr(arow[s][1], cswz), slot + 8192 + dma_off);
    }
  };
  auto issue_b = [&](int s, int kt, char* slot, int P = 0) {
    if ((ABL == 1 || ABL == 3 || (ABL >= 5 && ABL <= 7)) && !dma_on) return;
    if constexpr (BUFDMA) {
      if (P != 2) glds16_buf(rsw, wvoff[s][0], (unsigned)kt * (G256Q_BK * 2), slot + dma_off);
      if (P != 1) glds16_buf(rsw, wvoff[s][1], (unsigned)kt * (G256Q_BK * 2), slot + 8192 + dma_off);
    } else {
      if (P != 2) glds16(wrow[s][0] + kt * G256Q_BK, slot + dma_off);
      if (P != 1) glds16(wrow[s][1] + kt * G256Q_BK, slot + 8192 + dma_off);
    }
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // fragment reads: lane (r = lane&15, q = lane>>4) reads logical chunk 4*ks + q of local row base + r; key(row) = (row>>1)&7 and the
  // 16-row tile bases are multiples of 16, so the key depends on the lane only
  const int rkey = ((lane & 15) >> 1) & 7, q4 = lane >> 4;
  int a_addr[2], w_addr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_addr[ks] = (g * 64 + (lane & 15)) * 128 + (((ks * 4 + q4) ^ rkey) << 4);   // + i4 * 2048 (16 rows)
    w_addr[ks] = (wn * 32 + (lane & 15)) * 128 + (((ks * 4 + q4) ^ rkey) << 4);  // + j2 * 2048
  }
  half8_t af[4][2], wf[4][2];  // A: [16-row tile of the current sub][k32 step];  W: [16-column tile 0..3 (sub0: 0,1; sub1: 2,3)][k32 step]
  auto lds_read = [&](half8_t& dst, int addr, auto OFFC) {
    constexpr int OFF = decltype(OFFC)::value;
    if constexpr (ABL == 2 || ABL == 3 || (ABL >= 5 && ABL <= 7)) asm volatile("" : "+v"(dst));  // (ablation: the fragment keeps whatever it held)
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
  };
#define G256H_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define G256H_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

  // LOAD part of phase PH of K-tile t: fragment reads in consumption order (k32-step major), one piece staged, counted wait (v3, SCHED 1)
  auto load_part = [&](auto PHC, auto BUFC, int t, bool s1, bool s2) {
    constexpr int PH = decltype(PHC)::value, BUF = decltype(BUFC)::value, HI = BUF * G256Q_BUF_BYTES;
    char* cur = smem + BUF * G256Q_BUF_BYTES;
    char* oth = smem + (BUF ^ 1) * G256Q_BUF_BYTES;
    auto stage = [&](int P) {  // this phase's piece (P: which of its two LDS-DMAs)
      if constexpr ((OPT & 4) != 0) {  // (round 6 experiment) pieces placed against the parts' fragment reads (12 | 4 | 8 | 0): none | B1(t+1) | A1(t+1) | A0(t+2) B0(t+2)
        if constexpr (PH == 1) {
          if (s1) issue_b(1, t + 1, oth + G256Q_SLOT_B1);
        } else if constexpr (PH == 2) {
          if (s1) issue_a(1, oth + G256Q_SLOT_A1);
        } else if constexpr (PH == 3) {
          if (s2) {
            asrc.begin_tile(t + 2, G256Q_BK);
            issue_a(0, cur + G256Q_SLOT_A0);
            issue_b(0, t + 2, cur + G256Q_SLOT_B0);
          }
        }
        return;
      }
      if constexpr (PH == 0) {
        if (s1) issue_b(1, t + 1, oth + G256Q_SLOT_B1, P);
      } else if constexpr (PH == 1) {
        if (s1) issue_a(1, oth + G256Q_SLOT_A1, P);
      } else if constexpr (PH == 2) {
        if (s2) {
          if (P != 2) asrc.begin_tile(t + 2, G256Q_BK);
          issue_a(0, cur + G256Q_SLOT_A0, P);
        }
      } else {
        if (s2) issue_b(0, t + 2, cur + G256Q_SLOT_B0, P);
      }
    };
    if constexpr ((OPT & 2) != 0) {
      stage(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PH == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        lds_read(wf[0][ks], w_addr[ks] + HI, g256q_ic<G256Q_SLOT_B0>{});
        lds_read(wf[1][ks], w_addr[ks] + HI, g256q_ic<G256Q_SLOT_B0 + 2048>{});
        lds_read(af[0][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A0>{});
        lds_read(af[1][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A0 + 2048>{});
        lds_read(af[2][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A0 + 4096>{});
        lds_read(af[3][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A0 + 6144>{});
      }
    } else if constexpr (PH == 1) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        lds_read(wf[2][ks], w_addr[ks] + HI, g256q_ic<G256Q_SLOT_B1>{});
        lds_read(wf[3][ks], w_addr[ks] + HI, g256q_ic<G256Q_SLOT_B1 + 2048>{});
      }
    } else if constexpr (PH == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        lds_read(af[0][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A1>{});
        lds_read(af[1][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A1 + 2048>{});
        lds_read(af[2][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A1 + 4096>{});
        lds_read(af[3][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A1 + 6144>{});
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    stage((OPT & 2) != 0 ? 2 : 0);
    if constexpr (ABL == 1 || ABL == 3 || (ABL >= 5 && ABL <= 7)) {
      G256H_VMCNT(0);
    } else if constexpr ((OPT & 4) != 0) {  // issue order per K-tile: B1 B1 | A1 A1 | A0 A0 B0 B0; the LOAD part after next reads what has landed here
      if (s1) {
        if constexpr (PH < 3) G256H_VMCNT(4);
        else if (s2) G256H_VMCNT(6);
        else G256H_VMCNT(2);
      } else G256H_VMCNT(0);
    } else if (s2) G256H_VMCNT(6);  // pieces allowed in flight: 3 3 3 3 | 3 3 2 1 | 0 0 0 0
    else if (s1) {
      if constexpr (PH < 2) G256H_VMCNT(6);
      else if constexpr (PH == 2) G256H_VMCNT(4);
      else G256H_VMCNT(2);
    } else G256H_VMCNT(0);
  };
  // MFMA part of phase PH: one 64 x 32 quadrant x K = 64 = 16 MFMAs of 16x16x32, every group of 2 behind a counted lgkmcnt
  auto mfma_part = [&](auto PHC, auto SWC) {
    constexpr int PH = decltype(PHC)::value;
    constexpr bool SW = decltype(SWC)::value != 0;
    constexpr int I0 = (PH >= 2) ? 4 : 0, J0 = (PH == 1 || PH == 2) ? 2 : 0;
    if constexpr (ABL != 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        if constexpr (PH == 0) {  // 12 reads, per k32 step: W0 W1 A0 A1 A2 A3
          if (ks == 0) {
            if (i4 == 0) G256H_LGKM(9);
            else if (i4 == 1) G256H_LGKM(8);
            else if (i4 == 2) G256H_LGKM(7);
            else G256H_LGKM(6);
          } else {
            if (i4 == 0) G256H_LGKM(3);
            else if (i4 == 1) G256H_LGKM(2);
            else if (i4 == 2) G256H_LGKM(1);
            else G256H_LGKM(0);
          }
        } else if constexpr (PH == 1) {  // 4 reads: W2 W3 (ks 0), W2 W3 (ks 1)
          if (i4 == 0) {
            if (ks == 0) G256H_LGKM(2);
            else G256H_LGKM(0);
          }
        } else if constexpr (PH == 2) {  // 8 reads: A0..A3 (ks 0), A0..A3 (ks 1)
          if (ks == 0) {
            if (i4 == 0) G256H_LGKM(7);
            else if (i4 == 1) G256H_LGKM(6);
            else if (i4 == 2) G256H_LGKM(5);
            else G256H_LGKM(4);
          } else {
            if (i4 == 0) G256H_LGKM(3);
            else if (i4 == 1) G256H_LGKM(2);
            else if (i4 == 2) G256H_LGKM(1);
            else G256H_LGKM(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
          if constexpr (ABL == 7 || ABL == 8) {  // (experiment) the accumulator pinned: D and C the SAME registers
            if constexpr (SW) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[I0 + i4][J0 + j2]) : "v"(af[i4][ks]), "v"(wf[J0 + j2][ks]));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[I0 + i4][J0 + j2]) : "v"(wf[J0 + j2][ks]), "v"(af[i4][ks]));
          } else if constexpr (SW) acc[I0 + i4][J0 + j2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i4][ks], wf[J0 + j2][ks], acc[I0 + i4][J0 + j2], 0, 0, 0);
          else acc[I0 + i4][J0 + j2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[J0 + j2][ks], af[i4][ks], acc[I0 + i4][J0 + j2], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    if constexpr (ABL != 4) __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: pieces A0(0) B0(0) B1(0) A1(0) [A0(1) B0(1)]
  G256hRowStatRegs rsr;
  if constexpr (epi_has_rowstat<Epi>::value) {
    rsr = g256h_rowstat_load(epi, m0, M);
    __builtin_amdgcn_sched_barrier(0);
  }
  asrc.begin_tile(0, G256Q_BK);
  issue_a(0, smem + G256Q_SLOT_A0);
  issue_b(0, 0, smem + G256Q_SLOT_B0);
  issue_b(1, 0, smem + G256Q_SLOT_B1);
  issue_a(1, smem + G256Q_SLOT_A1);
  if (nk > 1) {
    asrc.begin_tile(1, G256Q_BK);
    issue_a(0, smem + G256Q_BUF_BYTES + G256Q_SLOT_A0);
    issue_b(0, 1, smem + G256Q_BUF_BYTES + G256Q_SLOT_B0);
  }
  if constexpr (epi_has_rowstat<Epi>::value) g256h_rowstat_finish(epi, rsr, smem, m0, M, tile_n);
  if (nk > 1) G256H_VMCNT(6);
  else G256H_VMCNT(2);
  if constexpr (ABL == 1 || ABL == 3 || (ABL >= 5 && ABL <= 7)) {
    G256H_VMCNT(0);
    dma_on = false;
  }
  if constexpr (ABL == 2 || ABL == 3 || (ABL >= 5 && ABL <= 7)) {  // defined fragment contents for the ablated reads
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        af[i][ks] = (half8_t){1, 2, 3, 4, 5, 6, 7, 8};
        wf[i][ks] = (half8_t){1, -1, 1, -1, 1, -1, 1, -1};
      }
  }
  if constexpr (ABL == 4) {
    if (g == 1) __builtin_amdgcn_s_setprio(1);
  }
  G256_BARRIER();
  // ONE barrier per phase: group 0 runs MFMA(p), LOAD(p+1); group 1 runs LOAD(p), MFMA(p) (hazard analysis: gemm256q_kernel.h)
  auto run = [&](auto GC, auto SWC) {
    constexpr int G = decltype(GC)::value;
    if constexpr (G == 0) load_part(g256q_ic<0>{}, g256q_ic<0>{}, 0, 1 < nk, 2 < nk);
    G256_BARRIER();
    auto tile = [&](auto BUFC, int t) {
      constexpr int BUF = decltype(BUFC)::value;
      const bool s1 = t + 1 < nk, s2 = t + 2 < nk, s3 = t + 3 < nk;
      auto phase = [&](auto PHC) {
        constexpr int PH = decltype(PHC)::value;
        if constexpr (G == 0) {
          mfma_part(PHC, SWC);
          if constexpr (PH < 3) load_part(g256q_ic<PH + 1>{}, BUFC, t, s1, s2);
          else if (s1) load_part(g256q_ic<0>{}, g256q_ic<(BUF ^ 1)>{}, t + 1, s2, s3);
        } else {
          load_part(PHC, BUFC, t, s1, s2);
          mfma_part(PHC, SWC);
        }
        if constexpr (ABL == 5 || ABL == 7) {
        } else if constexpr (ABL == 6) {
          if constexpr (PH == 3) G256_BARRIER();
        } else {
          G256_BARRIER();
        }
      };
      phase(g256q_ic<0>{});
      phase(g256q_ic<1>{});
      phase(g256q_ic<2>{});
      phase(g256q_ic<3>{});
    };
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      tile(g256q_ic<0>{}, t);
      tile(g256q_ic<1>{}, t + 1);
    }
    if (t < nk) tile(g256q_ic<0>{}, t);
  };
  if constexpr (epi_has_transposed<Epi>::value) {
    if (swapped) {
      if (g == 0) run(g256q_ic<0>{}, g256q_ic<1>{});
      else run(g256q_ic<1>{}, g256q_ic<1>{});
    } else {
      if (g == 0) run(g256q_ic<0>{}, g256q_ic<0>{});
      else run(g256q_ic<1>{}, g256q_ic<0>{});
    }
  } else {
    if (g == 0) run(g256q_ic<0>{}, g256q_ic<0>{});
    else run(g256q_ic<1>{}, g256q_ic<0>{});
  }
#undef G256H_VMCNT
#undef G256H_LGKM
  if constexpr (ABL == 7 || ABL == 8) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // inline-asm MFMAs: their results must have landed
  g256h_epilogue<G256_BN, TRACE>(acc, smem, epi, m0, n0, M, N, g, wn, lane, wave, bz, bsC, dbg, swapped);
}

template <class ASrc, class Epi, bool TRACE = false, int ABL = 0, int OPT = G256H_DEFAULT_OPT>
static inline int launch_gemm256h_tn(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t stream,
                                     int batch = 1, long bsA = 0, long bsW = 0, long bsC = 0) {
  if (!asrc_fits(asrc, 0)) return LFM_ERR_SHAPE;
  if constexpr ((OPT & 1) != 0 && asrc_has_buffer<ASrc>::value) {  // byte offsets of the buffer-addressed LDS-DMAs (batched operands: per batch element)
    if (!asrc_fits_buffer(asrc, 0) || (long)N * ldw >= (1L << 31)) return LFM_ERR_SHAPE;
  }
  if (M <= 0 || N <= 0 || K <= 0 || (K % G256Q_BK) != 0 || (N % 4) != 0) return LFM_ERR_SHAPE;
  if ((ldw % 8) != 0 || ((uintptr_t)W & 15)) return LFM_ERR_ALIGN;
  const int tm = cdiv(M, G256_BM), tn = cdiv(N, G256_BN);
  constexpr int LDS = G256Q_LDS_BYTES + (epi_has_rowstat<Epi>::value ? 2048 : 0);  // + rs[256][2] of the folded LayerNorm consumers
  static lfm_device_mask attr_set{0};  // one bit per device: the attribute is per (function, device)
  const unsigned long long dbit = lfm_device_bit();
  if (lfm_device_todo(attr_set, dbit)) {
    if (hipFuncSetAttribute((const void*)gemm256h_tn_kernel<ASrc, Epi, TRACE, ABL, OPT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LFM_ERR_LAUNCH;
    lfm_device_done(attr_set, dbit);
  }
  hipLaunchKernelGGL((gemm256h_tn_kernel<ASrc, Epi, TRACE, ABL, OPT>), dim3(tm * tn, batch), dim3(512), LDS, stream, asrc, W, ldw, M, N, K, tn, epi, bsA,
                     bsW, bsC, lfm_gemm_debug_flags());
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
