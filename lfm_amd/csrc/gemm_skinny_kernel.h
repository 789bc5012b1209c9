// Skinny-M MFMA GEMM for the batch-1 ("latency mode", --measure_time, /root/reference/test_flow_latent.py:223-246) DiT linears:
//   C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue),  M <= 256 rows (one image of 256 tokens), fp16 operands, fp32 accumulate.
//
// Why a kernel of its own (profiles/r04_latency_mode.txt): at M = 256 the 128x128 tiling gives 16-64 tiles for 256 CUs, so rounds 2-4 sliced K over
// blockIdx.y (fp32 slabs + a finish kernel per GEMM): 11 us per slab GEMM + 5-7 us per finish, both latency chains of a few K-tiles.  At this size the
// bound is WEIGHT streaming from HBM (0.9 GB per DiT-L/2 evaluation = 0.11 ms at 8 TB/s) and, per CU, streaming the activation rows through the L2
// path (every column slice needs all 256 rows).  This kernel inverts the tiling:
//   * a workgroup owns ALL 256 rows x 16 output columns (x one K-slice): its W slice (16 x K) is streamed from HBM exactly once by exactly one
//     workgroup; A (256 x K fp16, 0.5 MB at K = 1024) comes out of the L2 -- 256 workgroups x 0.5 MB = 128 MB of L2 traffic per GEMM, ~4.5 us at the
//     48 B/clk/CU that eight waves of buffer-addressed LDS-DMA sustain (tools/ubench/ldsdma_rate.hip);
//   * eight waves: all of them stage (thread tid moves 16-byte chunk tid & 7 of rows (tid >> 3) + 64 j, j = 0..3, of a 64-deep K-tile: 32 KiB of A
//     per stage, four stages, three K-tiles in flight behind counted vmcnt waits, ONE barrier per K-tile), wave w multiplies M-tiles 2 w, 2 w + 1
//     (v_mfma_f32_16x16x32_f16, W as the A operand: a lane owns four consecutive n of one row);
//   * WPRE (K-slice <= 1024): the workgroup's WHOLE W slice (16 x K-slice, <= 32 KiB) is requested up front, before the first A tile.  W comes from
//     HBM (first touch), A from the L2: with W staged tile by tile behind a three-tile lookahead the kernel was an HBM-LATENCY chain (first build:
//     12.0 us per K = 1024 GEMM = 20 B/clk/CU of A streaming); requested at once the slice costs one HBM round trip.  4 x 32 KiB + 32 KiB = the CU's
//     160 KiB of LDS.  Without WPRE (DiT-XL: K = 1152) W travels with its A tile: 2 KiB per stage;
//   * LDS image as every 128-byte-row kernel here: chunk c of row r at c ^ ((r >> 1) & 7), swizzle on the DMA source and on the fragment read;
//   * epilogue straight from the accumulators through the shared Epi interface (EpiQKV / EpiBiasGeluF16 in-kernel: no slab, no finish launch);
//   * grid.y > 1 slices K (proj / fc2: 64 column slices x 4 K-slices = 256 workgroups) into fp32 slabs (EpiSlabF32) for the row-owning finish kernel
//     that is also the next LayerNorm-modulate (splitk_finish_resid_ln_kernel) -- deterministic, fixed summation order.
// Requirements: N % 16 == 0, K-slice % 64 == 0, lda / ldw % 8 == 0, 16-byte aligned operands below 2^31 bytes (buffer-addressed DMA).
#pragma once
#include "gemm_kernel.h"

#ifndef SK_ROTATE
#define SK_ROTATE 0
#endif
#define SK_BK 64
#define SK_ROWS 256
#define SK_BN 16
#define SK_STAGES 4
#define SK_A_BYTES (SK_ROWS * SK_BK * 2)
#define SK_W_BYTES (SK_BN * SK_BK * 2)
#define SK_STAGE_BYTES (SK_A_BYTES + SK_W_BYTES)
#define SK_LDS_BYTES (SK_STAGES * SK_STAGE_BYTES)

typedef float sk_f32x4 __attribute__((ext_vector_type(4)));

#define SK_WPRE_MAX_KS 1024
#define SK_LDS_BYTES_WPRE (SK_STAGES * SK_A_BYTES + SK_BN * SK_WPRE_MAX_KS * 2)

template <class Epi, bool WPRE>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const half_t* __restrict__ A, long lda, const half_t* __restrict__ W, long ldw, int M, int N, int Ks,
                                                          Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = WPRE ? SK_A_BYTES : SK_STAGE_BYTES;  // bytes per ring stage
  constexpr int WBASE = WPRE ? SK_STAGES * SK_A_BYTES : SK_A_BYTES;  // WPRE: one region behind the ring, tile kt at + 2048 kt; else inside the stage
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * SK_BN, bz = blockIdx.y;
  epi_batch(epi, bz, 0, 0);
  const unsigned kbase = (unsigned)bz * (unsigned)Ks;
  // ---- DMA sources (buffer resources over the operands: unsigned 32-bit byte offsets)
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, -1, 0x00020000);
  const unsigned cswz = (unsigned)(((tid & 7) ^ ((tid >> 4) & 7)) * 8);  // source chunk of the physical slot this thread fills (key (row >> 1) & 7)
  unsigned avoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (tid >> 3) + 64 * j;
    avoff[j] = ((unsigned)(r < M ? r : M - 1) * (unsigned)lda + cswz) * 2u;
  }
  const int nk = Ks / SK_BK;
  if constexpr (WPRE) {
    // the whole W slice first: 2 nk DMAs of 1 KiB (8 rows x 128 B) dealt round-robin to the eight waves -- DMA d = 2 kt + h moves rows 8 h .. 8 h + 7 of
    // K-tile kt; d = wave (mod 8), so h = wave & 1 is fixed per wave.  Issued before every A tile: loads return in order, so the first counted wait
    // below already implies them (the per-wave counts need not be equal).
    const int h = wave & 1, wr = n0 + 8 * h + (lane >> 3);
    const unsigned wv = ((unsigned)(wr < N ? wr : N - 1) * (unsigned)ldw + (unsigned)(((lane & 7) ^ ((4 * h + (lane >> 4)) & 7)) * 8)) * 2u;
    for (int d = wave; d < 2 * nk; d += 8) glds16_buf(rsw, wv, (kbase + (unsigned)(d >> 1) * SK_BK) * 2u, smem + WBASE + d * 1024);
  }
  const int wr_t = n0 + (tid >> 3);  // (!WPRE) waves 0, 1 stage the 16 W rows of a tile (tid < 128)
  const unsigned wvoff = ((unsigned)(wr_t < N ? wr_t : N - 1) * (unsigned)ldw + cswz) * 2u;
  // (experiment, SK_ROTATE = 1; measured, off) K-tile ORDER rotated per workgroup: step t handles K-tile (t + rot) mod nk.  Two rotations were tried against the idea
  // that 256 workgroups streaming the same A rows in lockstep camp on L2 channels / all miss the same lines together: rot = workgroup mod nk and rot =
  // (workgroup / 8) mod nk (de-phasing inside an XCD).  Neither moved the K = 1024 launches (11.1 / 11.4 us vs 10.7): profiles/r05_latency_mode.txt.
  const int rot = SK_ROTATE ? (int)((blockIdx.x >> 3) % (unsigned)nk) : 0;
  auto tile_of = [&](int t) { return t + rot < nk ? t + rot : t + rot - nk; };
  auto issue = [&](int t) {
    const int kt = tile_of(t);
    char* st = smem + (t % SK_STAGES) * STAGE;
    const unsigned soff = (kbase + (unsigned)kt * SK_BK) * 2u;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16_buf(rsa, avoff[j], soff, st + j * 8192 + wave * 1024);
    if constexpr (!WPRE) {
      if (wave < 2) glds16_buf(rsw, wvoff, soff, st + SK_A_BYTES + wave * 1024);
    }
  };
  // ---- fragment read addresses: lane (r = lane & 15, q = lane >> 4) reads logical chunk 4 ks + q of row base + r
  const int rkey = ((lane & 15) >> 1) & 7, q4 = lane >> 4;
  int fa[2], fw[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((ks * 4 + q4) ^ rkey) << 4;
    fa[ks] = (32 * wave + (lane & 15)) * 128 + ch;  // + 2048 for the wave's second M-tile
    fw[ks] = WBASE + (lane & 15) * 128 + ch;
  }
  sk_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#define SK_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
  for (int t = 0; t < 3 && t < nk; ++t) issue(t);
  for (int t = 0; t < nk; ++t) {
    // K-tile t has landed: this thread's DMAs of the (at most two) later tiles may still be in flight -- 4 per tile (5 for the W-staging waves of !WPRE)
    const int later = nk - 1 - t < 2 ? nk - 1 - t : 2;
    if (!WPRE && wave < 2) {
      if (later == 2) SK_VMCNT(10);
      else if (later == 1) SK_VMCNT(5);
      else SK_VMCNT(0);
    } else {
      if (later == 2) SK_VMCNT(8);
      else if (later == 1) SK_VMCNT(4);
      else SK_VMCNT(0);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everyone's share of tile t is in the LDS, and everyone has finished reading tile t - 1 (its stage is free)
    asm volatile("" ::: "memory");
    if (t + 3 < nk) issue(t + 3);
    const int sb = (t % SK_STAGES) * STAGE, wb = WPRE ? tile_of(t) * 2048 : sb;
    half8_t af[2][2], wf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(wf[ks]) : "v"(fw[ks] + wb) : "memory");
      asm volatile("ds_read_b128 %0, %1" : "=v"(af[0][ks]) : "v"(fa[ks] + sb) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[1][ks]) : "v"(fa[ks] + sb) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks], af[0][ks], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks], af[1][ks], acc[1], 0, 0, 0);
    }
  }
#undef SK_VMCNT
  // ---- epilogue: lane l owns C[m = 16 (2 wave + i) + (l & 15)][n0 + 4 (l >> 4) .. + 3]
  const int n = n0 + 4 * q4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = 32 * wave + 16 * i + (lane & 15);
    if (m < M && n + 3 < N) {
      const f32x4 v = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
      epi.store(m, n, v, epi.load(m, n));
    }
  }
}

static inline bool gemm_skinny_ok(const void* A, long lda, const void* W, long ldw, int M, int N, int K, int S) {
  return M > 0 && M <= SK_ROWS && N > 0 && (N % SK_BN) == 0 && S >= 1 && (K % (S * SK_BK)) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 &&
         !(((uintptr_t)A | (uintptr_t)W) & 15) && (long)M * lda < (1L << 30) && (long)N * ldw < (1L << 30);
}
// S K-slices: slice bz covers k in [bz K / S, (bz + 1) K / S); with S > 1 the epilogue must be slab-addressed by the slice (EpiSlabF32)
template <class Epi>
static inline int launch_gemm_skinny(const half_t* A, long lda, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, int S, hipStream_t stream) {
  if (!gemm_skinny_ok(A, lda, W, ldw, M, N, K, S)) return LFM_ERR_SHAPE;
  static lfm_device_mask attr_set{0}, wpre_bad{0};  // one bit per device: the attribute is per (function, device)
  const unsigned long long bit = lfm_device_bit();
  if (lfm_device_todo(attr_set, bit)) {
    if (hipFuncSetAttribute((const void*)gemm_skinny_kernel<Epi, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES) != hipSuccess) return LFM_ERR_LAUNCH;
    // the W-prefetch variant takes the CU's whole 160 KiB: where the runtime refuses that much for one workgroup, the tile-by-tile variant serves every shape
    if (hipFuncSetAttribute((const void*)gemm_skinny_kernel<Epi, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES_WPRE) != hipSuccess) {
      (void)hipGetLastError();
      wpre_bad.fetch_or(bit, std::memory_order_release);
    }
    lfm_device_done(attr_set, bit);
  }
  const int Ks = K / S;
  if (Ks <= SK_WPRE_MAX_KS && !(wpre_bad.load(std::memory_order_acquire) & bit))
    hipLaunchKernelGGL((gemm_skinny_kernel<Epi, true>), dim3(N / SK_BN, S), dim3(512), SK_LDS_BYTES_WPRE, stream, A, lda, W, ldw, M, N, Ks, epi);
  else
    hipLaunchKernelGGL((gemm_skinny_kernel<Epi, false>), dim3(N / SK_BN, S), dim3(512), SK_LDS_BYTES, stream, A, lda, W, ldw, M, N, Ks, epi);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
