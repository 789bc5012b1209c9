// 256x256 "quadrant-phased" MFMA GEMM for gfx950 (v3):  C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue).
// Same operand / epilogue / A-source interfaces and the same wave -> output mapping as gemm256_kernel.h (v2), so the
// epilogue code is shared.  What changed, and why (measurements in DESIGN.md section 3):
//
//  * v2 stages 32-deep K-tiles, i.e. 64-B rows: one LDS-DMA instruction then touches 16 rows x HALF a cache line.
//    v3 stages 64-deep K-tiles = full 128-B lines (8 rows per DMA instruction), half the L2 requests per byte: -7 % cycles,
//    and -- the GEMMs run under the board power cap on real data -- the same in time.
//  * A K-tile is consumed in FOUR phases, one 64x32 output quadrant each (8 MFMAs = 256 matrix-pipe cycles), and is
//    staged in four 16-KiB PIECES cut along the same lines:  A.sub0 / A.sub1 = rows {0..63} / {64..127} of both
//    groups' 128-row halves,  B.sub0 / B.sub1 = columns {0..31} / {32..63} of every wave's 64-column block.  A piece is
//    last read in a known phase, so it can be re-staged two phases later: every phase issues exactly one piece
//    (2 DMAs per thread) and ends its LOAD part with the same counted vmcnt wait.
//
//      phase       reads (ds_read_b128)          MFMAs (acc rows x cols)    stages piece        of K-tile
//      P1(t)       B.sub0 (4) + A.sub0 (8)       sub0 x sub0                B.sub1              t+1
//      P2(t)       B.sub1 (4)                    sub0 x sub1                A.sub1              t+1
//      P3(t)       A.sub1 (8)                    sub1 x sub1                A.sub0              t+2
//      P4(t)       -                             sub1 x sub0                B.sub0              t+2
//
//    The two wave groups (rows 0..127 / 128..255 of the tile, one wave per SIMD each) alternate "LOAD part" and "MFMA part"
//    so that a SIMD's matrix pipe always has a wave feeding it.  Two schedules (template SCHED):
//      SCHED 1 (default): ONE barrier per phase.  Between two barriers group 0 runs MFMA(p), LOAD(p+1) and group 1 runs
//              LOAD(p), MFMA(p); the hand-over inside the interval is the matrix pipe's own queueing.  The wait in LOAD(p)
//              covers the pieces of phase p+2 (three pieces = 48 KiB per CU stay in flight).
//      SCHED 0 (A/B switch, flag 1): phase = [LOAD] barrier [MFMA] barrier, group 1 one barrier behind; the wait covers the
//              pieces of phase p+1 (four pieces in flight).
//    Hazards, both schedules: a piece is read only after every wave's wait for it AND a barrier; a piece read in phase r is
//    retired by the counted lgkmcnt waits of MFMA(r) and re-staged in LOAD(r+2) or later, behind a barrier (see the comments
//    at the two schedules below).
//  * Fragment reads are inline asm in the order the MFMAs consume them, every MFMA pair behind its own counted lgkmcnt.
//  * Tiles an epilogue wants transposed (EpiQKV: the V third) run with the MFMA operands swapped (template copy of the loop).
//    LDS = 2 K-tiles x 4 pieces x 16 KiB = 128 KiB; rows are 128 B, chunk c of row r sits at c ^ ((r>>1)&7) (v1's key).
#pragma once
#include "gemm256_kernel.h"
#include "gemm256n_kernel.h"

#define G256Q_BK 64
#define G256Q_PIECE 16384
#define G256Q_SLOT_A0 0
#define G256Q_SLOT_B0 (1 * G256Q_PIECE)
#define G256Q_SLOT_B1 (2 * G256Q_PIECE)
#define G256Q_SLOT_A1 (3 * G256Q_PIECE)
#define G256Q_BUF_BYTES (4 * G256Q_PIECE)
#define G256Q_LDS_BYTES (2 * G256Q_BUF_BYTES)

// Measurement only (lfm_gemm_select flag 2): waves 0 and 4 of block 0 take SIX s_memtime stamps per phase (phase start, reads
// issued, DMAs issued, vmcnt wait done, first barrier passed, MFMAs issued) into SGPRs, wait for them once at the end of the
// phase (when the LDS queue is empty anyway) and park them in the 32 KiB of LDS above the operand ring; they are copied out
// after the K loop.  lfm_gemm_trace_read() / tools/phase_trace.py.  One copy per translation unit; dit.hip's is read back.
#define G256Q_TRACE_MAX 2048
static __device__ unsigned long long g256q_trace[2][G256Q_TRACE_MAX];

template <int V>
struct g256q_ic {
  static constexpr int value = V;
};

template <class ASrc, class Epi, bool TRACE, int SCHED>
__global__ __launch_bounds__(512) void gemm256q_tn_kernel(ASrc asrc, const half_t* __restrict__ W, long ldw, int M, int N, int K,
                                                           int tiles_n, Epi epi, long bsA, long bsW, long bsC, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> SGPR
  const int g = wave >> 2, wn = wave & 3;

  int tile_m, tile_n;
  g256_tile_order(blockIdx.x, gridDim.x, tiles_n, dbg, tile_m, tile_n);
  const int m0 = tile_m * G256_BM, n0 = tile_n * G256_BN;
  // tiles the epilogue wants transposed (V^T of the QKV projection) run with the MFMA operands swapped: C^T blocks for free
  bool swapped = false;
  if constexpr (epi_has_transposed<Epi>::value && SCHED == 1) swapped = epi.transposed(n0);  // (the A/B schedule keeps the direct V^T path)
  const int bz = blockIdx.y;
  asrc.init(bz, bsA);
  W += (long)bz * bsW;

  // ---- DMA sources.  A piece is 128 local rows x 128 B = 1024 chunks of 16 B: thread tid stages chunks tid and 512 + tid,
  // i.e. local rows (tid>>3) and 64 + (tid>>3), physical chunk tid&7, which holds logical chunk (tid&7) ^ key(row).
  //   A.sub_s local row lr -> tile row (lr>>6)*128 + s*64 + (lr&63)         (group = lr>>6)
  //   B.sub_s local row lr -> tile col (lr>>5)*64  + s*32 + (lr&31)         (wave column = lr>>5)
  typename ASrc::Row arow[2][2];  // [sub][pass]
  const half_t* wrow[2][2];
  const int cswz = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;  // key(lr) = (lr>>1)&7 = (tid>>4)&7 for both passes
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      arow[s][p] = asrc.row(m0 + p * 128 + s * 64 + (tid >> 3));
      const int n = n0 + (p * 2 + (tid >> 8)) * 64 + s * 32 + ((tid >> 3) & 31);
      wrow[s][p] = W + (long)(n < N ? n : N - 1) * ldw + cswz;
    }
  const int nk = K / G256Q_BK;
  const int dma_off = wave * 1024;  // wave-uniform destination; the hardware adds lane*16

  auto issue_a = [&](int s, char* slot) {  // the A-source's current K-tile (begin_tile) applies
    glds16(asrc.ptr(arow[s][0], cswz), slot + dma_off);
    glds16(asrc.ptr(arow[s][1], cswz), slot + 8192 + dma_off);
  };
  auto issue_b = [&](int s, int kt, char* slot) {
    glds16(wrow[s][0] + kt * G256Q_BK, slot + dma_off);
    glds16(wrow[s][1] + kt * G256Q_BK, slot + 8192 + dma_off);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment reads: lane (r = lane&31, h = lane>>5) reads logical chunk 2*ks + h of local row base + r
  const int rkey = ((lane & 31) >> 1) & 7, chalf = lane >> 5;
  const int a_rd = (g * 64 + (lane & 31)) * 128;   // + i2 * 4096
  const int w_rd = (wn * 32 + (lane & 31)) * 128;
  half8_t af[2][4], wf[2][4];  // A: [32-row block][k16 step] of the current sub;  W: [sub][k16 step]

  // The reads are inline asm (invisible to the compiler's wait insertion, which otherwise puts a full lgkmcnt(0) in front of
  // the first MFMA of a phase), issued in the order the MFMAs consume them (k16-step major), and every pair of MFMAs is preceded
  // by a COUNTED lgkmcnt: the first MFMAs start as soon as their own fragments are there while the later reads are in flight.
  // per-k16-step read addresses (the swizzle term depends on ks and on the lane, the rest is a compile-time offset)
  int a_addr[4], w_addr[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    a_addr[ks] = a_rd + (((ks * 2 + chalf) ^ rkey) << 4);
    w_addr[ks] = w_rd + (((ks * 2 + chalf) ^ rkey) << 4);
  }
  auto lds_read = [&](half8_t& dst, int addr, auto OFFC) {  // OFF < 65536 (ds offset field); the caller folds the rest into addr
    constexpr int OFF = decltype(OFFC)::value;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
  };
  auto read_a = [&](auto BUFC, auto SLOTC) {
    constexpr int HI = decltype(BUFC)::value * G256Q_BUF_BYTES, SLOT = decltype(SLOTC)::value;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      lds_read(af[0][ks], a_addr[ks] + HI, g256q_ic<SLOT>{});
      lds_read(af[1][ks], a_addr[ks] + HI, g256q_ic<SLOT + 4096>{});
    }
  };
  auto read_w = [&](int s, auto BUFC, auto SLOTC) {
    constexpr int HI = decltype(BUFC)::value * G256Q_BUF_BYTES, SLOT = decltype(SLOTC)::value;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) lds_read(wf[s][ks], w_addr[ks] + HI, g256q_ic<SLOT>{});
  };

#define G256Q_VMCNT(n)                                       \
  do {                                                       \
    stamp(2);                                                \
    asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");    \
  } while (0)
  int trace_n = 0;
  unsigned long long tstamp[6];
  auto stamp = [&](int i) {
    if constexpr (TRACE) asm volatile("s_memtime %0" : "=s"(tstamp[i])::"memory");
  };
  auto stamp_flush = [&]() {
    if constexpr (TRACE) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (wn == 0 && lane == 0 && trace_n + 6 <= G256Q_TRACE_MAX) {
        unsigned long long* dst = (unsigned long long*)(smem + G256Q_LDS_BYTES + g * 16384) + trace_n;
#pragma unroll
        for (int i = 0; i < 6; ++i) dst[i] = tstamp[i];
      }
      trace_n += 6;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };

  // LOAD part of phase PH of K-tile t (BUF = K-tile parity; s1 / s2: K-tiles t+1 / t+2 exist, wave-uniform): the phase's fragment
  // reads, one piece staged, and the counted wait.  SCHED 0 waits for what the NEXT phase reads (four pieces stay in flight),
  // SCHED 1 for what the phase after the next reads (three pieces).
  auto load_part = [&](auto PHC, auto BUFC, int t, bool s1, bool s2) {
    constexpr int PH = decltype(PHC)::value, BUF = decltype(BUFC)::value;
    char* cur = smem + BUF * G256Q_BUF_BYTES;
    char* oth = smem + (BUF ^ 1) * G256Q_BUF_BYTES;
    stamp(0);
    if constexpr (PH == 0) {
      read_w(0, BUFC, g256q_ic<G256Q_SLOT_B0>{});
      read_a(BUFC, g256q_ic<G256Q_SLOT_A0>{});
    } else if constexpr (PH == 1) {
      read_w(1, BUFC, g256q_ic<G256Q_SLOT_B1>{});
    } else if constexpr (PH == 2) {
      read_a(BUFC, g256q_ic<G256Q_SLOT_A1>{});
    }
    stamp(1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PH == 0) {
      if (s1) issue_b(1, t + 1, oth + G256Q_SLOT_B1);
    } else if constexpr (PH == 1) {
      if (s1) issue_a(1, oth + G256Q_SLOT_A1);  // A-source state = K-tile t+1 (set in P3(t-1) / the prologue)
    } else if constexpr (PH == 2) {
      if (s2) {
        asrc.begin_tile(t + 2, G256Q_BK);
        issue_a(0, cur + G256Q_SLOT_A0);
      }
    } else {
      if (s2) issue_b(0, t + 2, cur + G256Q_SLOT_B0);
    }
    if constexpr (SCHED == 0) {  // pieces allowed in flight: 4 4 4 4 | second-to-last tile 4 4 3 2 | last tile 1 0 0 -
      if (s2) G256Q_VMCNT(8);
      else if (s1) {
        if constexpr (PH < 2) G256Q_VMCNT(8);
        else if constexpr (PH == 2) G256Q_VMCNT(6);
        else G256Q_VMCNT(4);
      } else {
        if constexpr (PH == 0) G256Q_VMCNT(2);
        else if constexpr (PH < 3) G256Q_VMCNT(0);
      }
    } else {  // 3 3 3 3 | 3 3 2 1 | 0 0 0 0
      if (s2) G256Q_VMCNT(6);
      else if (s1) {
        if constexpr (PH < 2) G256Q_VMCNT(6);
        else if constexpr (PH == 2) G256Q_VMCNT(4);
        else G256Q_VMCNT(2);
      } else G256Q_VMCNT(0);
    }
    stamp(3);
  };
  // MFMA part of phase PH: one 64x32 quadrant x K = 64, every pair of MFMAs behind a counted lgkmcnt
  auto mfma_part = [&](auto PHC, auto SWC) {
    constexpr int PH = decltype(PHC)::value;
    constexpr bool SW = decltype(SWC)::value != 0;  // operands swapped: the tile is computed transposed
    constexpr int I0 = (PH >= 2) ? 2 : 0, J = (PH == 1 || PH == 2) ? 1 : 0;
    stamp(4);
    if (!(dbg & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (PH < 3) {
        // reads still allowed in flight before k16 step ks:  P1 (4 W + 8 A, ks-major): 6,4,2,0;  P2 (4 W): 3,2,1,0;  P3 (8 A): 6,4,2,0
        if constexpr (PH == 1) {
          if (ks == 0) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
          else if (ks == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
          else if (ks == 2) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
          if (ks == 0) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
          else if (ks == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
          else if (ks == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (SW) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          acc[I0 + i2][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i2][ks], wf[J][ks], acc[I0 + i2][J], 0, 0, 0);
      } else {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          acc[I0 + i2][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[J][ks], af[i2][ks], acc[I0 + i2][J], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    stamp(5);
  };

  // ---- prologue: pieces A0(0) B0(0) B1(0) A1(0) [A0(1) B0(1)]
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
  if constexpr (SCHED == 0) {
    // phase = [LOAD part] barrier [MFMA part] barrier, group 1 one barrier behind: see the header.  P1(0) reads pieces 0, 1.
    if (nk > 1) G256Q_VMCNT(8);
    else G256Q_VMCNT(4);
    G256_BARRIER();
    if (g == 1) G256_BARRIER();
    auto tile = [&](auto BUFC, int t) {
      const bool s1 = t + 1 < nk, s2 = t + 2 < nk;
      auto phase = [&](auto PHC) {
        load_part(PHC, BUFC, t, s1, s2);
        G256_BARRIER();
        mfma_part(PHC, g256q_ic<0>{});
        stamp_flush();
        if (decltype(PHC)::value < 3 || s1 || g == 0) G256_BARRIER();  // group 1 skips the very last barrier
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
  } else {
    // SCHED 1: ONE barrier per phase.  Between two barriers group 0 runs  MFMA(p), LOAD(p+1)  and group 1 runs  LOAD(p), MFMA(p):
    // still "one wave of a SIMD computes while the other loads", but the hand-over inside the interval is not a barrier (the
    // second wave's MFMAs simply queue behind the first wave's), so a K-tile costs 4 barrier round trips instead of 8.
    //   RAW  the wait in LOAD(p) covers the pieces read in phase p+2: for either group a barrier lies between every wave's
    //        wait and the first read (group 0 reads them in LOAD(p+2) one interval later, group 1 two intervals later);
    //   WAR  a piece read in phase r is retired by the counted waits of MFMA(r), before the barrier that ends the interval of
    //        group 1's MFMA(r); it is re-staged in LOAD(r+2) or later, which for either group starts after that barrier.
    if (nk > 1) G256Q_VMCNT(6);  // phases 0 and 1 read pieces 0, 1, 2
    else G256Q_VMCNT(2);
    G256_BARRIER();
    auto run = [&](auto GC, auto SWC) {  // one copy of the loop per group / operand order (per-phase branches made the allocator spill)
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
          stamp_flush();
          G256_BARRIER();
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
  }
#undef G256Q_VMCNT

  // group 0's last barrier is group 1's first barrier of its last phase (P4: no reads), every earlier read has been
  // retired and every piece has landed: the LDS is free for the epilogue scratch
  if constexpr (TRACE) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && wn == 0) {
      const unsigned long long* src = (const unsigned long long*)(smem + G256Q_LDS_BYTES + g * 16384);
      const int n = trace_n < G256Q_TRACE_MAX ? trace_n : G256Q_TRACE_MAX;
      for (int i = lane; i < n; i += 64) g256q_trace[g][i] = src[i];
    }
    __syncthreads();
  }
  g256_epilogue(acc, smem, epi, m0, n0, M, N, g, wn, lane, wave, bz, bsC, dbg, swapped);
}

template <class ASrc, class Epi, bool TRACE = false, int SCHED = 1>
static inline int launch_gemm256q_tn(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t stream,
                                     int batch = 1, long bsA = 0, long bsW = 0, long bsC = 0) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % G256Q_BK) != 0 || (N % 4) != 0) return LFM_ERR_SHAPE;
  if ((ldw % 8) != 0 || ((uintptr_t)W & 15)) return LFM_ERR_ALIGN;
  const int tm = cdiv(M, G256_BM), tn = cdiv(N, G256_BN);
  constexpr int lds = G256Q_LDS_BYTES + (TRACE ? 32768 : 0);  // the trace parks its stamps above the operand ring
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm256q_tn_kernel<ASrc, Epi, TRACE, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return LFM_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm256q_tn_kernel<ASrc, Epi, TRACE, SCHED>), dim3(tm * tn, batch), dim3(512), lds, stream, asrc, W, ldw, M, N, K, tn, epi, bsA,
                     bsW, bsC, lfm_gemm_debug_flags());
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
