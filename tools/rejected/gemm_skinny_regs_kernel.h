// REJECTED (round 5, measured): the skinny batch-1 GEMM with A streamed straight into REGISTERS in the MFMA fragment layout (8 K-tiles in flight per wave, no
// barrier in the loop) instead of through the LDS ring: 18.9 us per K = 1024 GEMM against 10.7 us -- a fragment-layout load touches 16 rows x 64 bytes per
// instruction (half cache lines, 16 tag lookups), the LDS-DMA form 8 rows x full 128-byte lines.  profiles/r05_latency_mode.txt.  Kept for the record; not built.
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
//   * eight waves; wave w multiplies M-tiles 2 w, 2 w + 1 (v_mfma_f32_16x16x32_f16, W as the A operand: a lane owns four consecutive n of one row) and
//     streams ITS OWN 32 rows of A straight into registers in the fragment layout -- no wave shares A rows with another, so the LDS is not needed for A,
//     and the register file holds far more in flight: DEPTH = 8 K-tiles = 256 KiB per CU.  (Builds 1 and 2 of round 5 staged A through a 4 x 32 KiB LDS
//     ring: 96 KiB in flight per CU against ~2 us of L2 latency = 21 B/clk/CU, 12.0 / 10.7 us per K = 1024 GEMM; profiles/r05_latency_mode.txt.)  No
//     barrier in the main loop: counted vmcnt waits per K-tile, slots refilled right after the MFMAs that read them;
//   * the workgroup's WHOLE W slice (16 x K-slice, 32 KiB at K = 1024) is requested up front by LDS-DMA, before the first A tile: W is the operand that
//     comes from HBM (first touch); requested at once it costs one HBM round trip instead of one per lookahead window.  One barrier after it landed;
//     W fragments are read from the LDS one K-tile ahead (chunk c of row r at c ^ ((r >> 1) & 7), swizzle on the DMA source and on the read);
//   * epilogue straight from the accumulators through the shared Epi interface (EpiQKV / EpiBiasGeluF16 in-kernel: no slab, no finish launch);
//   * grid.y > 1 slices K (proj / fc2: 64 column slices x 4 K-slices = 256 workgroups) into fp32 slabs (EpiSlabF32) for the row-owning finish kernel
//     that is also the next LayerNorm-modulate (splitk_finish_resid_ln_kernel) -- deterministic, fixed summation order.
// Requirements: N % 16 == 0, K-slice % 64 == 0, lda / ldw % 8 == 0, 16-byte aligned operands below 2^31 bytes (buffer-addressed DMA).
#pragma once
#include "gemm_kernel.h"

#define SK_BK 64
#define SK_ROWS 256
#define SK_BN 16
#define SK_MAX_KS 4096  // W slice of a workgroup: 16 x Ks fp16 <= 128 KiB of LDS

typedef float sk_f32x4 __attribute__((ext_vector_type(4)));
template <int V>
struct sk_ic {
  static constexpr int value = V;
};
template <class F, int... I>
__device__ __forceinline__ void sk_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(sk_ic<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sk_for(F&& f) {
  sk_for_impl(f, std::make_integer_sequence<int, N>{});
}
// s_waitcnt vmcnt(4 n): at most n later K-tiles (four loads each) of this lane still in flight
__device__ __forceinline__ void sk_wait_tiles(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
  }
}

template <class Epi, int DEPTH>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const half_t* __restrict__ A, long lda, const half_t* __restrict__ W, long ldw, int M, int N, int Ks,
                                                          Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // the W slice: K-tile kt at 2048 kt, 16 rows x 128 B, chunk c of row r at c ^ ((r >> 1) & 7)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * SK_BN, bz = blockIdx.y;
  epi_batch(epi, bz, 0, 0);
  const unsigned kbase = (unsigned)bz * (unsigned)Ks;
  const int nk = Ks / SK_BK;
  // ---- the whole W slice first: 2 nk LDS-DMAs of 1 KiB (8 rows x 128 B) dealt round-robin to the eight waves -- DMA d = 2 kt + h moves rows 8 h .. 8 h + 7
  // of K-tile kt; d = wave (mod 8), so h = wave & 1 is fixed per wave.  W is the operand that comes from HBM: requested at once it costs one round trip.
  {
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, -1, 0x00020000);
    const int h = wave & 1, wr = n0 + 8 * h + (lane >> 3);
    const unsigned wv = ((unsigned)(wr < N ? wr : N - 1) * (unsigned)ldw + (unsigned)(((lane & 7) ^ ((4 * h + (lane >> 4)) & 7)) * 8)) * 2u;
    for (int d = wave; d < 2 * nk; d += 8) glds16_buf(rsw, wv, (kbase + (unsigned)(d >> 1) * SK_BK) * 2u, smem + d * 1024);
  }
  // ---- A: every wave streams ITS OWN 32 rows straight into registers in the MFMA fragment layout (no wave shares A rows, so the LDS would only be a
  // detour -- and a capacity limit: with a 4 x 32 KiB LDS ring the kernel sat at 96 KiB in flight per CU = 21 B/clk of L2 latency x bandwidth).  Lane
  // (r = lane & 15, q = lane >> 4) of M-tile i loads the 16 bytes A[32 wave + 16 i + r][k0 + 32 ks + 8 q ..]; DEPTH K-tiles (DEPTH x 32 KiB per CU) in flight.
  const int q4 = lane >> 4;
  unsigned aoff[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 32 * wave + 16 * i + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) aoff[i][ks] = ((unsigned)(r < M ? r : M - 1) * (unsigned)lda + (unsigned)(ks * 32 + q4 * 8)) * 2u;
  }
  half8_t af[DEPTH][2][2];
  // (inline asm operands must be lambda PARAMETERS, not captures)
  auto gload = [](half8_t& dst, unsigned off, const char* base) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory"); };
  auto lread = [](half8_t& dst, int addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); };
  auto load_tile = [&](auto SC, int kt) {
    constexpr int S = decltype(SC)::value;
    const char* base = (const char*)A + ((size_t)kbase + (size_t)kt * SK_BK) * 2;  // wave-uniform: an SGPR pair
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) gload(af[S][i][ks], aoff[i][ks], base);
  };
  sk_for<DEPTH>([&](auto SC) {
    if (decltype(SC)::value < nk) load_tile(SC, decltype(SC)::value);
  });
  // my share of W has landed once at most the A loads issued after it are outstanding; behind the barrier so has everyone's
  sk_wait_tiles(nk < DEPTH ? nk : DEPTH);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // W fragment reads: lane (r, q) reads logical chunk 4 ks + q of row r of K-tile t; double-buffered one tile ahead
  const int rkey = ((lane & 15) >> 1) & 7;
  int fw[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) fw[ks] = (lane & 15) * 128 + (((ks * 4 + q4) ^ rkey) << 4);
  half8_t wf[2][2];
  auto read_w = [&](auto BC, int t) {
    constexpr int B = decltype(BC)::value;
    lread(wf[B][0], fw[0] + t * 2048);
    lread(wf[B][1], fw[1] + t * 2048);
  };
  sk_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  read_w(sk_ic<0>{}, 0);
  for (int t0 = 0; t0 < nk; t0 += DEPTH) {
    sk_for<DEPTH>([&](auto SC) {
      constexpr int S = decltype(SC)::value, B = S & 1;
      const int t = t0 + S;
      if (t < nk) {
        const int later = nk - 1 - t < DEPTH - 1 ? nk - 1 - t : DEPTH - 1;
        sk_wait_tiles(later);                               // A fragments of tile t
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // W fragments of tile t (requested a tile ago)
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nk) read_w(sk_ic<(B ^ 1)>{}, t + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[B][ks], af[S][0][ks], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[B][ks], af[S][1][ks], acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + DEPTH < nk) load_tile(SC, t + DEPTH);  // refill the slot the MFMAs above have read
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  }
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
  return M > 0 && M <= SK_ROWS && N > 0 && (N % SK_BN) == 0 && S >= 1 && (K % (S * SK_BK)) == 0 && K / S <= SK_MAX_KS && (lda % 8) == 0 && (ldw % 8) == 0 &&
         !(((uintptr_t)A | (uintptr_t)W) & 15) && (long)M * lda < (1L << 30) && (long)N * ldw < (1L << 30);
}
// S K-slices: slice bz covers k in [bz K / S, (bz + 1) K / S); with S > 1 the epilogue must be slab-addressed by the slice (EpiSlabF32)
template <class Epi>
static inline int launch_gemm_skinny(const half_t* A, long lda, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, int S, hipStream_t stream) {
  if (!gemm_skinny_ok(A, lda, W, ldw, M, N, K, S)) return LFM_ERR_SHAPE;
  static unsigned long long attr_set = 0;  // one bit per device: the attribute is per (function, device)
  int devid = 0;
  (void)hipGetDevice(&devid);
  const unsigned long long bit = 1ull << (devid & 63);
  if (!(attr_set & bit)) {
    if (hipFuncSetAttribute((const void*)gemm_skinny_kernel<Epi, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_BN * SK_MAX_KS * 2) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_skinny_kernel<Epi, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_BN * SK_MAX_KS * 2) != hipSuccess)
      return LFM_ERR_LAUNCH;
    attr_set |= bit;
  }
  const int Ks = K / S, lds = SK_BN * Ks * 2;
  if (Ks / SK_BK >= 8) hipLaunchKernelGGL((gemm_skinny_kernel<Epi, 8>), dim3(N / SK_BN, S), dim3(512), lds, stream, A, lda, W, ldw, M, N, Ks, epi);
  else hipLaunchKernelGGL((gemm_skinny_kernel<Epi, 4>), dim3(N / SK_BN, S), dim3(512), lds, stream, A, lda, W, ldw, M, N, Ks, epi);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
