// 256x128 "two-workgroups-per-CU" MFMA GEMM for gfx950 (v4):  C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue).
// Same operand / A-source / epilogue interfaces and the same wave -> accumulator mapping as the 256x256 kernels (a wave owns a
// 128x64 block = 4 x 2 fragments of v_mfma_f32_32x32x16_f16), so the epilogue code is shared.
//
// Why another generation: the 256x256 kernels run ONE 8-wave workgroup per CU with all 512 registers of every
// SIMD.  Their epilogues (write 128 KiB fp16, or read-modify-write 256 KiB fp32 per tile) are bound by the memory path and no MFMA
// runs on that CU meanwhile: 18-40 % of a DiT GEMM.  Nothing can overlap them, because nothing else fits on the CU.  Here a
// workgroup is FOUR waves (one per SIMD, 2 x 2 over a 256 x 128 tile) with 72 KiB of LDS, so TWO workgroups are resident per CU,
// each SIMD hosting one wave of either.  They share nothing and run at their own pace:
//   * main loops of both: the SIMD's matrix pipe alternates between the two waves -- the ping-pong of v2/v3 without a barrier
//     between the partners (the arbitration is the pipe's own, oldest wave first);
//   * one workgroup in its epilogue: the other one has the whole matrix pipe.  To use more than half of it alone, a wave overlaps
//     its own LDS reads with its own MFMAs (fragments of k16-step s+1 are in flight while step s multiplies; counted lgkmcnt);
//   * age-ordered arbitration lets the older workgroup of a CU run ahead, so the two drift out of phase by themselves and their
//     epilogues do not coincide (flag 2048: static s_setprio by block parity instead, A/B).
// It is also the kernel for N = 128 problems (the 128-channel 256^2 convolutions of the VAE), which the 256-wide tiles waste.
//
// K is consumed in 32-deep tiles through a three-stage LDS ring (3 x 24 KiB: A 256 x 32 + W 128 x 32 fp16): tile t+2 is issued
// right after the barrier of iteration t, so two tiles (48 KiB per workgroup) are in flight while one is multiplied; waits are
// counted (vmcnt(6): my six DMAs of tile t+1 may still fly).  Rows are 64 B; chunk c of row r sits at c ^ ((r>>2)&3) (v2's key,
// applied to the DMA source and to the fragment read).
//   RAW  a stage is read after every wave's vmcnt for it and the barrier of the iteration;
//   WAR  stage (t+2)%3 held tile t-1, whose last reads were retired (lgkmcnt(0)) before its MFMAs, i.e. before the barrier of
//        iteration t that precedes the DMA issue.
#pragma once
#include "gemm256_common.h"

#define G256N_BN 128
#define G256N_BK 32
#define G256N_A_BYTES (256 * G256N_BK * 2)                    // 16 KiB
#define G256N_STAGE_BYTES (G256N_A_BYTES + 128 * G256N_BK * 2)  // + W 8 KiB = 24 KiB
#define G256N_LDS_BYTES (3 * G256N_STAGE_BYTES)                // 72 KiB

template <int V>
struct g256n_ic {
  static constexpr int value = V;
};

template <class ASrc, class Epi>
__global__ __launch_bounds__(256, 2) void gemm256n_tn_kernel(ASrc asrc, const half_t* __restrict__ W, long ldw, int M, int N, int K, int tiles_n,
                                                              Epi epi, long bsA, long bsW, long bsC, int dbg, int stag) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
#ifdef LFM_MEASURE
  lfm_stagger_start(stag);
#endif

  int tile_m, tile_n;
  g256_tile_order(blockIdx.x, gridDim.x, tiles_n, dbg | 32, tile_m, tile_n);  // GM = 8: an XCD runs ~64 of these tiles at a time
  const int m0 = tile_m * G256_BM, n0 = tile_n * G256N_BN;
  bool swapped = false;
  if constexpr (epi_has_transposed<Epi>::value) swapped = epi.transposed(n0);
  const int bz = blockIdx.y;
  asrc.init(bz, bsA);
  W += (long)bz * bsW;
  if (dbg & 2048) {  // A/B: static priority by block parity instead of the pipe's age order
    if (__builtin_amdgcn_readfirstlane(blockIdx.x) & 1) __builtin_amdgcn_s_setprio(1);
  }

  // ---- DMA sources: A 256 rows x 4 chunks = 4 passes of 256 threads (row = p*64 + tid>>2), W 128 rows = 2 passes
  typename ASrc::Row arow[4];
  const half_t* wrow[2];
  const int cswz = ((tid & 3) ^ ((tid >> 4) & 3)) * 8;  // key(row) = (row>>2)&3 = (tid>>4)&3 for every pass (64 rows per pass)
#pragma unroll
  for (int p = 0; p < 4; ++p) arow[p] = asrc.row(m0 + p * 64 + (tid >> 2));
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int n = n0 + p * 64 + (tid >> 2);
    wrow[p] = W + (long)(n < N ? n : N - 1) * ldw + cswz;
  }
  const int nk = K / G256N_BK;
  const int dma_off = wave * 1024;

  auto issue_tile = [&](int kt, int stage) {
    char* sA = smem + stage * G256N_STAGE_BYTES;
    char* sW = sA + G256N_A_BYTES;
    asrc.begin_tile(kt, G256N_BK);
#pragma unroll
    for (int p = 0; p < 4; ++p) glds16(asrc.ptr(arow[p], cswz), sA + p * 4096 + dma_off);
#pragma unroll
    for (int p = 0; p < 2; ++p) glds16(wrow[p] + kt * G256N_BK, sW + p * 4096 + dma_off);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment reads: lane (r = lane&31, h = lane>>5) reads logical chunk 2*ks + h of its row; rows of a 32-row block share
  // (row>>2)&3 = (r>>2)&3 (block bases are multiples of 32)
  const int rkey = ((lane & 31) >> 2) & 3, chalf = lane >> 5;
  int a_addr[2], w_addr[2];  // per k16 step, byte offsets inside a stage
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_addr[ks] = (wm * 128 + (lane & 31)) * 64 + (((ks * 2 + chalf) ^ rkey) << 4);
    w_addr[ks] = G256N_A_BYTES + (wn * 64 + (lane & 31)) * 64 + (((ks * 2 + chalf) ^ rkey) << 4);
  }
  half8_t af[2][4], wf[2][2];  // [k16 step][block]
  auto lds_read = [&](half8_t& dst, int addr, auto OFFC) {
    constexpr int OFF = decltype(OFFC)::value;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
  };
  // 12 reads of one K-tile in consumption order: per k16 step W0 W1 A0 A1 A2 A3
  auto read_tile = [&](auto STC) {
    constexpr int BASE = decltype(STC)::value * G256N_STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      lds_read(wf[ks][0], w_addr[ks], g256n_ic<BASE>{});
      lds_read(wf[ks][1], w_addr[ks], g256n_ic<BASE + 32 * 64>{});
      lds_read(af[ks][0], a_addr[ks], g256n_ic<BASE>{});
      lds_read(af[ks][1], a_addr[ks], g256n_ic<BASE + 32 * 64>{});
      lds_read(af[ks][2], a_addr[ks], g256n_ic<BASE + 64 * 64>{});
      lds_read(af[ks][3], a_addr[ks], g256n_ic<BASE + 96 * 64>{});
    }
  };
#define G256N_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
  auto mfma_tile = [&](auto SWC) {
    constexpr bool SW = decltype(SWC)::value != 0;
    if (dbg & 8) __builtin_amdgcn_s_setprio(1);  // A/B: raise the priority around the MFMA cluster (T5)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // reads still allowed in flight before block i of step ks (issue order per step: W0 W1 A0 A1 A2 A3)
        if (ks == 0) {
          if (i == 0) G256N_LGKM(9);
          else if (i == 1) G256N_LGKM(8);
          else if (i == 2) G256N_LGKM(7);
          else G256N_LGKM(6);
        } else {
          if (i == 0) G256N_LGKM(3);
          else if (i == 1) G256N_LGKM(2);
          else if (i == 2) G256N_LGKM(1);
          else G256N_LGKM(0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (SW) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][i], wf[ks][j], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    if (dbg & 8) __builtin_amdgcn_s_setprio(0);
  };
#undef G256N_LGKM

  // ---- prologue: tiles 0 and 1 in flight
  issue_tile(0, 0);
  if (nk > 1) issue_tile(1, 1);

  auto run = [&](auto SWC) {
    auto iter = [&](auto STC, int t) {
      constexpr int ST = decltype(STC)::value;
      if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // my share of tile t has landed; tile t+1 may still fly
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      G256_BARRIER();
      if (t + 2 < nk) issue_tile(t + 2, (ST + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);
      read_tile(STC);
      __builtin_amdgcn_sched_barrier(0);
      mfma_tile(SWC);
    };
    int t = 0;
    for (; t + 2 < nk; t += 3) {
      iter(g256n_ic<0>{}, t);
      iter(g256n_ic<1>{}, t + 1);
      iter(g256n_ic<2>{}, t + 2);
    }
    if (t < nk) iter(g256n_ic<0>{}, t);
    if (t + 1 < nk) iter(g256n_ic<1>{}, t + 1);
  };
  if constexpr (epi_has_transposed<Epi>::value) {
    if (swapped) run(g256n_ic<1>{});
    else run(g256n_ic<0>{});
  } else {
    run(g256n_ic<0>{});
  }
  G256_BARRIER();  // every wave's last fragment reads are retired: the ring is free for the epilogue scratch (4 x 8.5 KiB)
  g256_epilogue<G256N_BN>(acc, smem, epi, m0, n0, M, N, wm, wn, lane, wave, bz, bsC, dbg, swapped);
}

template <class ASrc, class Epi>
static inline int launch_gemm256n_tn(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t stream,
                                     int batch = 1, long bsA = 0, long bsW = 0, long bsC = 0) {
  if (!asrc_fits(asrc, 0)) return LFM_ERR_SHAPE;
  if (M <= 0 || N <= 0 || K <= 0 || (K % G256N_BK) != 0 || (N % 4) != 0) return LFM_ERR_SHAPE;
  if ((ldw % 8) != 0 || ((uintptr_t)W & 15)) return LFM_ERR_ALIGN;
  const int tm = cdiv(M, G256_BM), tn = cdiv(N, G256N_BN);
  static lfm_device_mask attr_set{0};
  const unsigned long long dbit = lfm_device_bit();
  if (lfm_device_todo(attr_set, dbit)) {
    if (hipFuncSetAttribute((const void*)gemm256n_tn_kernel<ASrc, Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, G256N_LDS_BYTES) != hipSuccess)
      return LFM_ERR_LAUNCH;
    lfm_device_done(attr_set, dbit);
  }
  hipLaunchKernelGGL((gemm256n_tn_kernel<ASrc, Epi>), dim3(tm * tn, batch), dim3(256), G256N_LDS_BYTES, stream, asrc, W, ldw, M, N, K, tn, epi, bsA,
                     bsW, bsC, lfm_gemm_debug_flags(), lfm_stagger_ticks());
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
