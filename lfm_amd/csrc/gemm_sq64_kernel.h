// 64x64-tile MFMA GEMM for the batch-1 ("latency mode", --measure_time, /root/reference/test_flow_latent.py:223-246) DiT linears when the image has 128 / 192 /
// 256 tokens:   C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue),  M % 64 == 0, M <= 256, fp16 operands, fp32 accumulate.
//
// Why next to gemm_skinny_kernel.h (all rows x 16 columns): that kernel streams 0.5 MB of A + 32 KiB of W per workgroup at K = 1024 and was measured to be bound
// PER CU -- ~96 KiB in flight against ~2 us per round trip = 10.7 us, whatever the other workgroups do (profiles/r05_latency_mode.txt).  The bytes a workgroup
// must pull for a fixed number of outputs, (rows + columns) x K x 2, are smallest for a SQUARE tile: 64 x 64 needs 256 KiB at K = 1024 (2.1x fewer), and its
// 16-KiB K-tiles let a ten-stage ring keep 128 KiB in flight.  The price: a W slice (64 x K) is now wanted by M / 64 workgroups instead of one -- they are
// placed on the SAME XCD (workgroup id mod 8 is the XCD under round-robin dispatch), so the slice leaves HBM once and the others hit that XCD's L2.
//   * eight waves; thread tid moves 16-byte chunk tid & 7 of row tid >> 3 of the A tile and of the W tile: one DMA each per K-tile (64 deep, 128-byte rows,
//     chunk c of row r at c ^ ((r >> 1) & 7) as everywhere), a ten-stage ring (160 KiB), K-tiles consumed in pairs behind ONE barrier with eight more in flight;
//   * wave w multiplies n-tile w & 3 with m-tiles 2 (w >> 2), 2 (w >> 2) + 1 (v_mfma_f32_16x16x32_f16, W as the A operand: a lane owns four consecutive n of a row);
//   * epilogues through the shared Epi interface (EpiQKV / EpiBiasGeluF16 in the kernel); grid.y > 1 slices K into fp32 slabs (EpiSlabF32) for
//     splitk_finish_resid_ln_kernel exactly as the skinny kernel does -- fixed summation order, deterministic.
// Requirements: M % 64 == 0, N % 64 == 0, K-slice % 64 == 0, lda / ldw % 8 == 0, 16-byte aligned operands below 2^31 bytes (buffer-addressed DMA).
#pragma once
#include "gemm_kernel.h"

#define SQ_BK 64
#define SQ_T 64
#ifndef SQ_STAGES
#define SQ_STAGES 8
#endif
#define SQ_HALF_BYTES (SQ_T * SQ_BK * 2)
#define SQ_STAGE_BYTES (2 * SQ_HALF_BYTES)
#define SQ_LDS_BYTES (SQ_STAGES * SQ_STAGE_BYTES)

template <class Epi>
__global__ __launch_bounds__(512) void gemm_sq64_kernel(const half_t* __restrict__ A, long lda, const half_t* __restrict__ W, long ldw, int M, int N, int Ks, int mt,
                                                        int xcd_map, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> tile.  xcd_map (column tiles % 8 == 0): ids x, x + 8, .., x + 8 (mt - 1) -- one XCD -- are the mt row tiles of ONE column tile
  const unsigned x = blockIdx.x;
  int m_tile, n_tile;
  if (xcd_map) {
    const unsigned j = x >> 3;
    m_tile = (int)(j % (unsigned)mt);
    n_tile = (int)((j / (unsigned)mt) * 8u + (x & 7u));
  } else {
    m_tile = (int)(x % (unsigned)mt);
    n_tile = (int)(x / (unsigned)mt);
  }
  const int m0 = m_tile * SQ_T, n0 = n_tile * SQ_T, bz = blockIdx.y;
  epi_batch(epi, bz, 0, 0);
  const unsigned kbase = (unsigned)bz * (unsigned)Ks;
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, -1, 0x00020000);
  const unsigned cswz = (unsigned)(((tid & 7) ^ ((tid >> 4) & 7)) * 8);  // source chunk of the physical slot this thread fills (key (row >> 1) & 7)
  const unsigned avoff = ((unsigned)(m0 + (tid >> 3)) * (unsigned)lda + cswz) * 2u;
  const unsigned wvoff = ((unsigned)(n0 + (tid >> 3)) * (unsigned)ldw + cswz) * 2u;
  const int nk = Ks / SQ_BK;
  auto issue = [&](int t) {
    char* st = smem + (t % SQ_STAGES) * SQ_STAGE_BYTES + wave * 1024;
    const unsigned soff = (kbase + (unsigned)t * SQ_BK) * 2u;
    glds16_buf(rsw, wvoff, soff, st + SQ_HALF_BYTES);  // W first: it is the operand that may come from HBM
    glds16_buf(rsa, avoff, soff, st);
  };
  // fragment reads: lane (r = lane & 15, q = lane >> 4) reads logical chunk 4 ks + q of row base + r
  const int rkey = ((lane & 15) >> 1) & 7, q4 = lane >> 4, wn = wave & 3, wm = wave >> 2;
  int fa[2], fw[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((ks * 4 + q4) ^ rkey) << 4;
    fa[ks] = (32 * wm + (lane & 15)) * 128 + ch;  // + 2048 for the wave's second m-tile
    fw[ks] = SQ_HALF_BYTES + (16 * wn + (lane & 15)) * 128 + ch;
  }
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#define SQ_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
  // K-tiles are consumed in PAIRS behind one barrier: with one workgroup per CU all eight waves are in the same phase, so nothing hides a fragment read or an
  // MFMA chain -- a K-tile per barrier cost ~580 clk (tools/latency_gemm_probe.py: 0.275 us per tile with every operand in the L2); per pair the second tile's
  // reads fly under the first tile's MFMAs and there is half the barriers.  Tiles t + SQ_STAGES - 2, t + SQ_STAGES - 1 are requested at the top of pair t into
  // the stages pair t - 2 was read from: SQ_STAGES - 2 tiles (128 KiB) in flight under the pair being multiplied.
  for (int t = 0; t < SQ_STAGES - 2 && t < nk; ++t) issue(t);
  for (int t = 0; t < nk; t += 2) {
    const bool two = t + 1 < nk;
    // tiles t, t + 1 have landed: this thread's DMAs of the (at most SQ_STAGES - 4) later tiles may still be in flight, two per tile
    const int later = nk - 2 - t < SQ_STAGES - 4 ? nk - 2 - t : SQ_STAGES - 4;
    static_assert(SQ_STAGES >= 4 && SQ_STAGES <= 10, "the counted waits below");
    if (later >= 6) SQ_VMCNT(12);
    else if (later == 5) SQ_VMCNT(10);
    else if (later == 4) SQ_VMCNT(8);
    else if (later == 3) SQ_VMCNT(6);
    else if (later == 2) SQ_VMCNT(4);
    else if (later == 1) SQ_VMCNT(2);
    else SQ_VMCNT(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everyone's share of the pair is in the LDS, and everyone has finished reading the previous pair (its stages are free)
    asm volatile("" ::: "memory");
    if (t + SQ_STAGES - 2 < nk) issue(t + SQ_STAGES - 2);
    if (t + SQ_STAGES - 1 < nk) issue(t + SQ_STAGES - 1);
    half8_t af[2][2][2], wf[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      const int sb = ((t + u) % SQ_STAGES) * SQ_STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(wf[u][ks]) : "v"(fw[ks] + sb) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(af[u][0][ks]) : "v"(fa[ks] + sb) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[u][1][ks]) : "v"(fa[ks] + sb) : "memory");
      }
    }
    if (two) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][ks], af[0][0][ks], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][ks], af[0][1][ks], acc[1], 0, 0, 0);
    }
    if (two) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][ks], af[1][0][ks], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][ks], af[1][1][ks], acc[1], 0, 0, 0);
      }
    }
  }
#undef SQ_VMCNT
  // ---- epilogue: lane l owns C[m0 + 32 wm + 16 i + (l & 15)][n0 + 16 wn + 4 (l >> 4) .. + 3]
  const int n = n0 + 16 * wn + 4 * q4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + 32 * wm + 16 * i + (lane & 15);
    if (m < M && n + 3 < N) epi.store(m, n, acc[i], epi.load(m, n));
  }
}

static inline bool gemm_sq64_ok(const void* A, long lda, const void* W, long ldw, int M, int N, int K, int S) {
  return M > 0 && M <= 256 && (M % SQ_T) == 0 && N > 0 && (N % SQ_T) == 0 && S >= 1 && (K % (S * SQ_BK)) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 &&
         !(((uintptr_t)A | (uintptr_t)W) & 15) && (long)M * lda < (1L << 30) && (long)N * ldw < (1L << 30);
}
// S K-slices: slice bz covers k in [bz K / S, (bz + 1) K / S); with S > 1 the epilogue must be slab-addressed by the slice (EpiSlabF32)
template <class Epi>
static inline int launch_gemm_sq64(const half_t* A, long lda, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, int S, hipStream_t stream) {
  if (!gemm_sq64_ok(A, lda, W, ldw, M, N, K, S)) return LFM_ERR_SHAPE;
  static lfm_device_mask attr_set{0};  // one bit per device: the attribute is per (function, device)
  const unsigned long long bit = lfm_device_bit();
  if (lfm_device_todo(attr_set, bit)) {
    if (hipFuncSetAttribute((const void*)gemm_sq64_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, SQ_LDS_BYTES) != hipSuccess) return LFM_ERR_LAUNCH;
    lfm_device_done(attr_set, bit);
  }
  const int mt = M / SQ_T, nt = N / SQ_T;
  hipLaunchKernelGGL((gemm_sq64_kernel<Epi>), dim3(nt * mt, S), dim3(512), SQ_LDS_BYTES, stream, A, lda, W, ldw, M, N, K / S, mt, (nt % 8) == 0 ? 1 : 0, epi);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
