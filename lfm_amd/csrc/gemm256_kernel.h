// 256x256 "ping-pong" MFMA GEMM for gfx950 (v2):  C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue).
// Same operand / epilogue / A-source interfaces as gemm_kernel.h (v1); used for the large GEMMs of the hot path.
//
// 512 threads = 8 waves = two wave-GROUPS of four (one wave per SIMD each).  Group g owns rows g*128..+128 of the
// tile, wave (g, wn) the 128x64 block at columns wn*64: 4(m) x 2(n) fragments of v_mfma_f32_32x32x16_f16 = 128
// accumulator registers.  K is consumed in tiles of 32: per K-tile a wave runs a LOAD segment (12 ds_read_b128
// fragments into registers + its share of a future tile's LDS-DMA) and a COMPUTE segment (16 MFMAs, nothing else).
// The two groups run the same program shifted by ONE segment, separated by workgroup barriers:
//
//   slot      2t       2t+1     2t+2     2t+3
//   group 0   L(t)     C(t)     L(t+1)   C(t+1)
//   group 1   C(t-1)   L(t)     C(t)     L(t+1)
//
// so on every SIMD one wave is always in a COMPUTE segment while its partner loads: the matrix pipe sees a
// back-to-back MFMA stream.
//
// LDS = a RING of four stages of (A 256x32 + W 256x32) fp16 = 4 x 32 KiB.  Measurement drove this shape: with two
// 64-deep stages only ONE K-tile (64 KiB per CU) could be in flight and the loop ran at the DMA round-trip latency
// (~1.9 us per 64-deep K-tile, equal with and without the MFMAs).  With the ring, K-tile t+3 is issued in L(t) and is
// not needed before slot 2t+6: three tiles (96 KiB per CU) are always in flight and a load has ~6 segments to land.
// Waits are COUNTED (s_waitcnt vmcnt(8): "everything except my last two tiles has landed"), never 0 in steady state.
// Rows are 64 B, so the XOR swizzle key is (row>>2)&3 on 16-B chunks (a 256-B bank row = four tile rows); as in v1 it
// is applied to the DMA source address and to the fragment read.  Every LOAD segment drains its own ds_reads
// (lgkmcnt(0)) before its barrier, so a stage is never refilled while a read of it is in flight.
#pragma once
#include "gemm_kernel.h"

int lfm_gemm_selected();     // 0 auto, 1 force v1, 2 force v2 (set by lfm_gemm_select)
int lfm_gemm_debug_flags();  // ablation switches, measurement only
int lfm_gemm_prefers_v4(int M, int N, int K);  // shapes where the 256x128 two-workgroups-per-CU kernel measured faster than the 256x256 one

#define G256_BM 256
#define G256_BN 256
#define G256_BK 32
#define G256_NSTAGE 4
#define G256_TILE_BYTES (256 * G256_BK * 2)     // one operand tile, 16 KiB
#define G256_STAGE_BYTES (2 * G256_TILE_BYTES)  // A + W, 32 KiB
#define G256_LDS_BYTES (G256_NSTAGE * G256_STAGE_BYTES)  // 128 KiB (the epilogue scratch reuses it)

// Tile order.  Block b runs on XCD b%8: give each XCD a contiguous range of tile ids, and inside a range walk groups of
// GM = 4 M-panels column-major, so the ~32 tiles an XCD runs concurrently form a 4 x 8 patch (12 operand panels in its
// L2) instead of a 1 x 32 / 2 x 16 strip (33 / 18 panels).
__device__ __forceinline__ void g256_tile_order(int bid, int nb, int tiles_n, int dbg, int& tile_m, int& tile_n) {
  if ((nb & 7) == 0 && !(dbg & 128)) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  // GM x tiles_n should be a multiple of the ~32 tiles an XCD runs at once: 8 for 12 tile columns (QKV, measured -3 %), else 4
  const int tiles_m = nb / tiles_n, GM = (dbg & 32) ? 8 : ((dbg & 64) ? 2 : ((dbg & 256) ? 4 : ((tiles_n & 7) && tiles_n > 8 ? 8 : 4)));
  const int grp = bid / (GM * tiles_n), within = bid - grp * (GM * tiles_n);
  const int gm = (tiles_m - grp * GM) < GM ? (tiles_m - grp * GM) : GM;  // last group may be short
  tile_m = grp * GM + within % gm;
  tile_n = within / gm;
}

// epilogues with fp16 outputs provide store8(m, n, lo, hi, aux_lo, aux_hi): EIGHT consecutive columns = one 16-byte store
template <class Epi, class = void>
struct epi_has_store8 {
  static constexpr bool value = false;
};
template <class Epi>
struct epi_has_store8<Epi, decltype((void)&Epi::store8)> {
  static constexpr bool value = true;
};

// epilogues whose auxiliary operand depends on the column only (a bias row) say so with  static constexpr bool column_aux = true :
// kernels may then load it once per tile, AHEAD of the first store (vmcnt counts stores too and returns in order, so a bias load issued
// after a block's stores waits for those stores to drain -- profiles/r02_epilogue_trace.txt, the "aux" column of the bias epilogues)
template <class Epi, class = void>
struct epi_column_aux {
  static constexpr bool value = false;
};
template <class Epi>
struct epi_column_aux<Epi, decltype((void)Epi::column_aux)> {
  static constexpr bool value = Epi::column_aux;
};

// The row-major path of the epilogue: each wave transposes its accumulators through a private LDS scratch, see g256_epilogue.
// fp16 outputs (store8): a lane re-reads EIGHT consecutive columns of a row (two ds_read_b128) and issues ONE 16-byte store, so a
// store instruction covers 8 rows x one full 128-B line -- half the store instructions of the 4-column form.  The fp16 epilogues
// were store-ISSUE bound (3.8 TB/s ~ 7 B/cycle/CU with 8-byte stores, the guide's T21 case), not bandwidth bound.
template <int BN = G256_BN, class Epi>
__device__ __forceinline__ void g256_epilogue_rows(f32x16 (&acc)[4][2], char* smem, const Epi& epi, int m0, int n0, int M, int N, int g, int wn,
                                                   int lane, int wave, bool narrow = false) {
  const int chalf = lane >> 5;
  char* scr = smem + wave * (32 * 272);
  const bool interior = (m0 + G256_BM <= M) && (n0 + BN <= N);
  if constexpr (epi_has_store8<Epi>::value) {
    if (!narrow && epi.wide_ok()) {
      const int rrow = lane >> 3, rcol = lane & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *(f32x4*)(scr + (lane & 31) * 272 + (j * 32 + 8 * q + 4 * chalf) * 4) = v;
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 lo[4], hi[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          lo[ps] = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + rcol * 32);
          hi[ps] = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + rcol * 32 + 16);
        }
        const int mb = m0 + g * 128 + i * 32 + rrow, n = n0 + wn * 64 + rcol * 8;
        if (interior) {
          typename Epi::Aux al[4], ah[4];
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            al[ps] = epi.load(mb + ps * 8, n);
            ah[ps] = epi.load(mb + ps * 8, n + 4);
          }
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) epi.store8(mb + ps * 8, n, lo[ps], hi[ps], al[ps], ah[ps]);
        } else {
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int m = mb + ps * 8;
            if (m >= M) continue;
            if (n + 7 < N) epi.store8(m, n, lo[ps], hi[ps], epi.load(m, n), epi.load(m, n + 4));
            else if (n + 3 < N) epi.store(m, n, lo[ps], epi.load(m, n));  // N % 4 == 0: a ragged edge ends on a 4-column boundary
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      return;
    }
  }
  const int rrow = lane >> 4, rcol = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        *(f32x4*)(scr + (lane & 31) * 272 + (j * 32 + 8 * q + 4 * chalf) * 4) = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 v[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) v[ps] = *(const f32x4*)(scr + (ps * 4 + rrow) * 272 + rcol * 16);
    const int mb = m0 + g * 128 + i * 32 + rrow, n = n0 + wn * 64 + rcol * 4;
    if (interior) {
      typename Epi::Aux aux[8];
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) aux[ps] = epi.load(mb + ps * 4, n);
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) epi.store(mb + ps * 4, n, v[ps], aux[ps]);
    } else if (n + 3 < N) {
#pragma unroll
      for (int ps = 0; ps < 8; ++ps)
        if (mb + ps * 4 < M) epi.store(mb + ps * 4, n, v[ps], epi.load(mb + ps * 4, n));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// Shared epilogue of the 256x256 kernels (wave (g, wn) owns rows g*128.., columns wn*64.., acc[i][j] = 32x32 block i, j).
template <int BN = G256_BN, class Epi>
__device__ __forceinline__ void g256_epilogue(f32x16 (&acc)[4][2], char* smem, Epi& epi, int m0, int n0, int M, int N, int g, int wn, int lane,
                                              int wave, int bz, long bsC, int dbg, bool swapped = false) {
  const int chalf = lane >> 5;
  // ---- epilogue.  The MFMA leaves lane (m = lane&31, h = lane>>5) with 4 consecutive n per register group: storing that
  // directly makes every store instruction touch 32 different 128-B lines with 16-32 B each (measured: ~12 us per tile,
  // L2-request-bound).  Instead each wave transposes its block through a PRIVATE 32 x 64 fp32 LDS scratch (row stride
  // 272 B: conflict-free ds_write_b128) and re-reads it row-major: 16 lanes cover one 256-B row, so a global access
  // instruction touches 4 rows x full lines.  Epilogues that want the fragment layout (V^T scatter) opt out.
  epi_batch(epi, bz, bsC, 0);
  if (dbg & 4) return;  // ablation: no epilogue
  if constexpr (epi_has_transposed<Epi>::value) {
    // The K loop ran this tile with the MFMA operands swapped: lane (n = lane&31, h) holds FOUR CONSECUTIVE m per register
    // group.  Same scratch, roles exchanged: rows = 32 columns n of block j, columns = 64 rows m of blocks 2*ih, 2*ih+1; read
    // back row-major, 16 lanes cover 64 consecutive m of one n -> epi.store_t(n, m, C[m..m+3][n]).
    if (swapped && !(dbg & 1024) && epi.wide_t_ok()) {  // 16-byte stores: lane = (column n = lane>>3 of 8 per pass, 8 consecutive rows m)
      char* scr = smem + wave * (32 * 272);
      const int rrow = lane >> 3, rcol = lane & 7;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nb = n0 + wn * 64 + j * 32 + rrow;
        float bt[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) bt[ps] = nb + ps * 8 < N ? epi.load_t(nb + ps * 8) : 0.f;
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 v = {acc[2 * ih + ii][j][4 * q], acc[2 * ih + ii][j][4 * q + 1], acc[2 * ih + ii][j][4 * q + 2], acc[2 * ih + ii][j][4 * q + 3]};
              *(f32x4*)(scr + (lane & 31) * 272 + (ii * 32 + 8 * q + 4 * chalf) * 4) = v;
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          f32x4 lo[4], hi[4];
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            lo[ps] = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + rcol * 32);
            hi[ps] = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + rcol * 32 + 16);
          }
          const int m = m0 + g * 128 + ih * 64 + rcol * 8;
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            if (nb + ps * 8 >= N) continue;
            if (m + 7 < M) epi.store_t8(nb + ps * 8, m, lo[ps], hi[ps], bt[ps]);
            else if (m + 3 < M) epi.store_t(nb + ps * 8, m, lo[ps], bt[ps]);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      return;
    }
    if (swapped) {
      char* scr = smem + wave * (32 * 272);
      const int rrow = lane >> 4, rcol = lane & 15;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nb = n0 + wn * 64 + j * 32 + rrow;
        float bt[8];  // all loads of a block before its first store (the compiler cannot move a load above a possibly-aliasing store)
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) bt[ps] = nb + ps * 4 < N ? epi.load_t(nb + ps * 4) : 0.f;
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 v = {acc[2 * ih + ii][j][4 * q], acc[2 * ih + ii][j][4 * q + 1], acc[2 * ih + ii][j][4 * q + 2], acc[2 * ih + ii][j][4 * q + 3]};
              *(f32x4*)(scr + (lane & 31) * 272 + (ii * 32 + 8 * q + 4 * chalf) * 4) = v;
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          f32x4 v[8];
#pragma unroll
          for (int ps = 0; ps < 8; ++ps) v[ps] = *(const f32x4*)(scr + (ps * 4 + rrow) * 272 + rcol * 16);
          const int m = m0 + g * 128 + ih * 64 + rcol * 4;
#pragma unroll
          for (int ps = 0; ps < 8; ++ps)
            if (nb + ps * 4 < N && m + 3 < M) epi.store_t(nb + ps * 4, m, v[ps], bt[ps]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      return;
    }
  }
  if (epi_direct(epi, n0, 0)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + g * 128 + i * 32 + (lane & 31);
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * chalf;
          if (n + 3 < N) {
            f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            epi.store(m, n, v, epi.load(m, n));
          }
        }
    }
    return;
  }
  if constexpr (epi_has_plain<Epi>::value) {
    if (epi.plain_tile(n0, BN)) {
      auto pe = epi.plain(n0);
      g256_epilogue_rows<BN>(acc, smem, pe, m0, n0, M, N, g, wn, lane, wave, (dbg & 1024) != 0);
      return;
    }
  }
  g256_epilogue_rows<BN>(acc, smem, epi, m0, n0, M, N, g, wn, lane, wave, (dbg & 1024) != 0);  // flag 1024: the 8-byte-store epilogue (A/B)
}

template <class ASrc, class Epi>
__global__ __launch_bounds__(512) void gemm256_tn_kernel(ASrc asrc, const half_t* __restrict__ W, long ldw, int M, int N, int K,
                                                          int tiles_n, Epi epi, long bsA, long bsW, long bsC, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = wave >> 2, wn = wave & 3;
  const bool dbg_noload = dbg & 1, dbg_nomfma = dbg & 2;  // ablation switches (measurement only)

  int tile_m, tile_n;
  g256_tile_order(blockIdx.x, gridDim.x, tiles_n, dbg, tile_m, tile_n);
  const int m0 = tile_m * G256_BM, n0 = tile_n * G256_BN;
  const int bz = blockIdx.y;
  asrc.init(bz, bsA);
  W += (long)bz * bsW;

  // ---- DMA sources: 2 passes of 128 rows per operand, 4 lanes per 64-B row
  typename ASrc::Row arow[2];
  const half_t* wrow[2];
  int cswz[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = p * 128 + (tid >> 2);
    arow[p] = asrc.row(m0 + r);
    const int n = n0 + r;
    wrow[p] = W + (long)(n < N ? n : N - 1) * ldw;
    cswz[p] = ((tid & 3) ^ ((r >> 2) & 3)) * 8;
  }
  const int nk = K / G256_BK;

  // 4 DMAs per thread per K-tile (2 A + 2 W)
  auto issue_tile = [&](int kt) {
    char* sA = smem + (kt & (G256_NSTAGE - 1)) * G256_STAGE_BYTES;
    char* sW = sA + G256_TILE_BYTES;
    const int k0 = kt * G256_BK;
    asrc.begin_tile(kt, G256_BK);
    if (dbg_noload && kt > 0) return;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      glds16(asrc.ptr(arow[p], cswz[p]), sA + (p * 512 + wave * 64) * 16);
      glds16(wrow[p] + k0 + cswz[p], sW + (p * 512 + wave * 64) * 16);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read offsets (bytes inside an operand tile); chunk c of row r sits at c ^ ((r>>2)&3)
  int a_off[4], a_key[4], w_off[2], w_key[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = g * 128 + i * 32 + (lane & 31);
    a_off[i] = r * 64;
    a_key[i] = (r >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wn * 64 + j * 32 + (lane & 31);
    w_off[j] = r * 64;
    w_key[j] = (r >> 2) & 3;
  }
  const int chalf = lane >> 5;

  half8_t af[4][2], wf[2][2];  // [frag][k16 step]

  auto load_frags = [&](int kt) {
    const char* sA = smem + (kt & (G256_NSTAGE - 1)) * G256_STAGE_BYTES;
    const char* sW = sA + G256_TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf[j][ks] = *(const half8_t*)(sW + w_off[j] + (((ks * 2 + chalf) ^ w_key[j]) << 4));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = *(const half8_t*)(sA + a_off[i] + (((ks * 2 + chalf) ^ a_key[i]) << 4));
  };
  auto compute = [&]() {
    if (dbg_nomfma) return;
    if (!(dbg & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][ks], af[i][ks], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
#define G256_BARRIER()                  \
  do {                                  \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_s_barrier();       \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);  \
  } while (0)
#define G256_LGKM0()                                    \
  do {                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    __builtin_amdgcn_sched_barrier(0);                  \
  } while (0)
  // "K-tile kt has landed" for this wave's share: every tile up to min(kt+2, nk-1) has been issued, 4 DMAs each, in order
  auto wait_tile = [&](int kt) {
    const int newer = (nk - 1 - kt) < 2 ? (nk - 1 - kt) : 2;
    if (newer >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---- prologue: K-tiles 0..2 in flight, tile 0 landed
  issue_tile(0);
  if (nk > 1) issue_tile(1);
  if (nk > 2) issue_tile(2);
  wait_tile(0);
  G256_BARRIER();

  if (g == 0) {
    for (int t = 0; t < nk; ++t) {
      if ((dbg & 16) && t + 3 < nk) issue_tile(t + 3);
      load_frags(t);  // slot 2t: L(t)
      if (!(dbg & 16) && t + 3 < nk) issue_tile(t + 3);
      G256_LGKM0();
      G256_BARRIER();
      compute();  // slot 2t+1: C(t)
      if (t + 1 < nk) wait_tile(t + 1);
      G256_BARRIER();  // (after the last tile: every stage read is finished -> the epilogue may reuse the LDS)
    }
  } else {
    G256_BARRIER();  // slot 0: group 1 idles, then stays one segment behind
    for (int t = 0; t < nk; ++t) {
      if ((dbg & 16) && t + 3 < nk) issue_tile(t + 3);
      load_frags(t);  // slot 2t+1: L(t)
      if (!(dbg & 16) && t + 3 < nk) issue_tile(t + 3);
      G256_LGKM0();
      if (t + 1 < nk) wait_tile(t + 1);
      G256_BARRIER();
      compute();  // slot 2t+2: C(t)
      if (t + 1 < nk) G256_BARRIER();
    }
  }

  g256_epilogue(acc, smem, epi, m0, n0, M, N, g, wn, lane, wave, bz, bsC, dbg);
}

template <class ASrc, class Epi>
static inline int launch_gemm256_tn(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t stream,
                                    int batch = 1, long bsA = 0, long bsW = 0, long bsC = 0) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % G256_BK) != 0 || (N % 4) != 0) return LFM_ERR_SHAPE;
  if ((ldw % 8) != 0 || ((uintptr_t)W & 15)) return LFM_ERR_ALIGN;
  const int tm = cdiv(M, G256_BM), tn = cdiv(N, G256_BN);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm256_tn_kernel<ASrc, Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES) != hipSuccess)
      return LFM_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm256_tn_kernel<ASrc, Epi>), dim3(tm * tn, batch), dim3(512), G256_LDS_BYTES, stream, asrc, W, ldw, M, N, K, tn, epi,
                     bsA, bsW, bsC, lfm_gemm_selected() == 2 ? lfm_gemm_debug_flags() : (lfm_gemm_debug_flags() & 4));  // other flags belong to v3
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
