// Shared pieces of the 256-row MFMA GEMM kernels (256x128 "two workgroups per CU", gemm256n_kernel.h; 256x256 quadrant-phased on
// v_mfma_f32_16x16x32_f16, gemm256h_kernel.h): tile order, epilogue traits, the LDS-transposed row-major epilogue for the 32x32x16 accumulator
// map, the barrier macro, the LDS image constants of the quadrant-phased ring.
// History (DESIGN.md section 3 keeps the measurements): the first two 256x256 generations -- a four-stage ping-pong ring of 32-deep K-tiles
// (round 1) and its quadrant-phased successor on 32x32x16 MFMAs with 64-deep K-tiles (round 1 / 2) -- were superseded by the 16x16x32 kernel
// and removed in round 3; no reference shape dispatched to them (every K on the path is a multiple of 64).
#pragma once
#include "gemm_kernel.h"

int lfm_gemm_selected();     // 0 auto, 1 force the 128x128 kernel, 4 the 256x128 one, 5 the 256x256 one (set by lfm_gemm_select)
int lfm_gemm_debug_flags();  // ablation switches, measurement only
int lfm_stagger_ticks();  // measurement builds (lfm_set_option key 3): s_memtime ticks by which workgroups 256..511 of a co-resident-pair kernel start late; else 0
#ifdef LFM_MEASURE
// Two workgroups share a CU in the 256x128 GEMM and the halo convolution; dispatched together they run in lockstep (both in their main loops, then both in
// their epilogues).  The experiment: delay the second resident of every CU in the FIRST wave of workgroups (ids 256..511: consecutive ids go round the
// XCDs and then the CUs, so id + 256 is the partner of id); later workgroups inherit the offset because a slot frees when its predecessor ends.
__device__ __forceinline__ void lfm_stagger_start(int ticks) {
  if (ticks > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)ticks) __builtin_amdgcn_s_sleep(16);
  }
}
#endif
int lfm_gemm_prefers_v4(int M, int N, int K);  // shapes where the 256x128 two-workgroups-per-CU kernel measured faster than the 256x256 one

#define G256_BM 256
#define G256_BN 256

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

// epilogues that accumulate something per lane across their store8 calls (GroupNorm partial sums of the convolution outputs, vae.hip) provide
// finish_tile(m0, n0, g, wn, lane): called once per wave after the row-major hand-over, in which a lane always owns the SAME eight columns
// n0 + 64 wn + 8 (lane & 7) and rows of the 128-row half g
template <class Epi, class = void>
struct epi_has_finish_tile {
  static constexpr bool value = false;
};
template <class Epi>
struct epi_has_finish_tile<Epi, decltype((void)&Epi::finish_tile)> {
  static constexpr bool value = true;
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
      const int rrow = lane >> 3;
      const int ml = 16 * ((lane & 7) >> 1) + 4 * (lane & 1);  // tokens ml .. ml + 3 and ml + 8 .. ml + 11: one 16-byte chunk of the permuted V^T row (vt_pos)
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
            lo[ps] = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + ml * 4);
            hi[ps] = *(const f32x4*)(scr + (ps * 8 + rrow) * 272 + ml * 4 + 32);
          }
          const int m = m0 + g * 128 + ih * 64 + ml;
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            if (nb + ps * 8 >= N) continue;
            if (m + 11 < M) epi.store_t8(nb + ps * 8, m, lo[ps], hi[ps], bt[ps]);
            else {
              if (m + 3 < M) epi.store_t(nb + ps * 8, m, lo[ps], bt[ps]);
              // (the hi half -- tokens m + 8 .. m + 11 -- lies beyond M here; wide_t_ok() implies M % 16 == 0)
            }
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
  if constexpr (epi_has_finish_tile<Epi>::value) epi.finish_tile(m0, n0, g, wn, lane);
}

#define G256_BARRIER()                  \
  do {                                  \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_s_barrier();       \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);  \
  } while (0)

// ---- LDS image of the quadrant-phased 256x256 kernel: 64-deep K-tiles = 128-byte rows, staged in four 16-KiB pieces (A.sub0 / A.sub1 = rows
// {0..63} / {64..127} of both groups' 128-row halves, B.sub0 / B.sub1 = columns {0..31} / {32..63} of every wave's 64-column block); two K-tiles
// = 128 KiB.  Chunk c of row r sits at c ^ ((r >> 1) & 7), applied to the DMA source address and to the fragment read.
#define G256Q_BK 64
#define G256Q_PIECE 16384
#define G256Q_SLOT_A0 0
#define G256Q_SLOT_B0 (1 * G256Q_PIECE)
#define G256Q_SLOT_B1 (2 * G256Q_PIECE)
#define G256Q_SLOT_A1 (3 * G256Q_PIECE)
#define G256Q_BUF_BYTES (4 * G256Q_PIECE)
#define G256Q_LDS_BYTES (2 * G256Q_BUF_BYTES)

// Measurement only (lfm_gemm_select flag 2 with kernel 5): s_memtime stamps of the epilogue, parked here and read back with lfm_gemm_trace_read().
// One copy per translation unit; dit.hip's is read back.
#define G256Q_TRACE_MAX 2048
static __device__ unsigned long long g256q_trace[2][G256Q_TRACE_MAX];

template <int V>
struct g256q_ic {
  static constexpr int value = V;
};
