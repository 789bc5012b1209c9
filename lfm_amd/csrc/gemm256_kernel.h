// 256x256x64 "ping-pong" MFMA GEMM for gfx950 (v2):  C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue).
// Same operand / epilogue / A-source interfaces as gemm_kernel.h (v1); used for the large GEMMs of the hot path.
//
// 512 threads = 8 waves = two wave-GROUPS of four (one wave per SIMD each).  Group g owns rows g*128..+128 of the
// tile, wave (g, wn) the 128x64 block at columns wn*64: 4(m) x 2(n) fragments of v_mfma_f32_32x32x16_f16 = 128
// accumulator registers.  A K-tile is consumed in two PHASES of 16 MFMAs (m-fragments {0,1} then {2,3}); every phase
// has a LOAD segment (ds_read_b128 fragments into registers, plus a share of the next tile's LDS-DMA) and a COMPUTE
// segment (pure MFMA).  The two groups run the same program shifted by ONE segment, separated by workgroup barriers:
//
//   slot      4t        4t+1      4t+2      4t+3      4t+4 ...
//   group 0   L0(t)     C0(t)     L1(t)     C1(t)     L0(t+1)
//   group 1   C1(t-1)   L0(t)     C0(t)     L1(t)     C1(t)
//
// so on every SIMD one wave is always in a COMPUTE segment while its partner loads: the matrix pipe sees a
// back-to-back MFMA stream.  LDS holds two stages of (A 256x64 + W 256x64) fp16 = 2 x 64 KiB, XOR-swizzled exactly
// like v1 (source-side swizzle for the LDS-DMA, same key on the reads).  Tile t+1 is DMA'd into the other stage
// during slots 4t..4t+2 and must have landed (vmcnt(0) on every wave + the barrier ending slot 4t+3) before slot 4t+4;
// every LOAD segment drains its own ds_reads (lgkmcnt(0)) before its barrier, so a stage is never refilled while a
// read of it is in flight.
#pragma once
#include "gemm_kernel.h"

int lfm_gemm_selected();     // 0 auto, 1 force v1, 2 force v2 (set by lfm_gemm_select)
int lfm_gemm_debug_flags();  // ablation switches, measurement only

#define G256_BM 256
#define G256_BN 256
#define G256_BK 64
#define G256_TILE_BYTES (256 * 64 * 2)          // one operand tile, 32 KiB
#define G256_STAGE_BYTES (2 * G256_TILE_BYTES)  // A + W
#define G256_LDS_BYTES (2 * G256_STAGE_BYTES)   // 128 KiB

template <class ASrc, class Epi>
__global__ __launch_bounds__(512) void gemm256_tn_kernel(ASrc asrc, const half_t* __restrict__ W, long ldw, int M, int N, int K,
                                                          int tiles_n, Epi epi, long bsA, long bsW, long bsC, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = wave >> 2, wn = wave & 3;
  const bool dbg_noload = dbg & 1, dbg_nomfma = dbg & 2;  // ablation switches (measurement only)

  // Tile order.  Block b runs on XCD b%8: give each XCD a contiguous range of tile ids, and inside a range walk groups of
  // GM = 4 M-panels column-major, so the ~32 tiles an XCD runs concurrently form a 4 x 8 patch (12 operand panels in its
  // L2) instead of a 1 x 32 / 2 x 16 strip (33 / 18 panels).
  int bid = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  int tile_m, tile_n;
  {
    const int tiles_m = nb / tiles_n, GM = 4;
    const int grp = bid / (GM * tiles_n), within = bid - grp * (GM * tiles_n);
    const int gm = (tiles_m - grp * GM) < GM ? (tiles_m - grp * GM) : GM;  // last group may be short
    tile_m = grp * GM + within % gm;
    tile_n = within / gm;
  }
  const int m0 = tile_m * G256_BM, n0 = tile_n * G256_BN;
  const int bz = blockIdx.y;
  asrc.init(bz, bsA);
  W += (long)bz * bsW;

  // ---- DMA sources: 4 passes of 64 rows per operand, 8 lanes per 128-B row
  typename ASrc::Row arow[4];
  const half_t* wrow[4];
  int cswz[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = p * 64 + (tid >> 3);
    arow[p] = asrc.row(m0 + r);
    const int n = n0 + r;
    wrow[p] = W + (long)(n < N ? n : N - 1) * ldw;
    cswz[p] = ((tid & 7) ^ ((r >> 1) & 7)) * 8;
  }
  const int nk = K / G256_BK;

  // 8 DMAs per thread per K-tile (4 A + 4 W)
  auto issue_tile = [&](int kt, int stage) {
    char* sA = smem + stage * G256_STAGE_BYTES;
    char* sW = sA + G256_TILE_BYTES;
    const int k0 = kt * G256_BK;
    asrc.begin_tile(kt);
    if (dbg_noload && kt > 0) return;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
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

  // fragment read offsets
  int a_off[4], a_key[4], w_off[2], w_key[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = g * 128 + i * 32 + (lane & 31);
    a_off[i] = r * 128;
    a_key[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wn * 64 + j * 32 + (lane & 31);
    w_off[j] = r * 128;
    w_key[j] = (r >> 1) & 7;
  }
  const int chalf = lane >> 5;

  half8_t af[2][4], wf[2][4];  // [frag][k16 step]

  auto load_w = [&](const char* sW) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wf[j][ks] = *(const half8_t*)(sW + w_off[j] + (((ks * 2 + chalf) ^ w_key[j]) << 4));
  };
  auto load_a = [&](const char* sA, int pair) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        af[i][ks] = *(const half8_t*)(sA + a_off[pair * 2 + i] + (((ks * 2 + chalf) ^ a_key[pair * 2 + i]) << 4));
  };
  auto compute = [&](int pair) {
    if (dbg_nomfma) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[pair * 2 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][ks], af[i][ks], acc[pair * 2 + i][j], 0, 0, 0);
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
#define G256_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

  // ---- prologue: tile 0 -> stage 0
  issue_tile(0, 0);
  G256_VM0();
  G256_BARRIER();

  if (g == 0) {
    for (int t = 0; t < nk; ++t) {
      const char* sA = smem + (t & 1) * G256_STAGE_BYTES;
      const char* sW = sA + G256_TILE_BYTES;
      const bool more = t + 1 < nk;
      // slot 4t: L0
      load_w(sW);
      load_a(sA, 0);
      if (more) issue_tile(t + 1, (t + 1) & 1);
      G256_LGKM0();
      G256_BARRIER();
      // slot 4t+1: C0
      compute(0);
      G256_BARRIER();
      // slot 4t+2: L1
      load_a(sA, 1);
      G256_LGKM0();
      G256_BARRIER();
      // slot 4t+3: C1
      compute(1);
      if (more) G256_VM0();
      G256_BARRIER();  // (last tile: all stage reads are finished -> the epilogue may reuse the LDS)
    }
  } else {
    G256_BARRIER();  // slot 0: group 1 idles
    for (int t = 0; t < nk; ++t) {
      const char* sA = smem + (t & 1) * G256_STAGE_BYTES;
      const char* sW = sA + G256_TILE_BYTES;
      const bool more = t + 1 < nk;
      // slot 4t+1: L0
      load_w(sW);
      load_a(sA, 0);
      if (more) issue_tile(t + 1, (t + 1) & 1);
      G256_LGKM0();
      G256_BARRIER();
      // slot 4t+2: C0
      compute(0);
      G256_BARRIER();
      // slot 4t+3: L1
      load_a(sA, 1);
      G256_LGKM0();
      if (more) G256_VM0();
      G256_BARRIER();
      // slot 4t+4: C1
      compute(1);
      if (more) G256_BARRIER();
    }
  }

  // ---- epilogue.  The MFMA leaves lane (m = lane&31, h = lane>>5) with 4 consecutive n per register group: storing that
  // directly makes every store instruction touch 32 different 128-B lines with 16-32 B each (measured: ~12 us per tile,
  // L2-request-bound).  Instead each wave transposes its block through a PRIVATE 32 x 64 fp32 LDS scratch (row stride
  // 272 B: conflict-free ds_write_b128) and re-reads it row-major: 16 lanes cover one 256-B row, so a global access
  // instruction touches 4 rows x full lines.  Epilogues that want the fragment layout (V^T scatter) opt out.
  epi_batch(epi, bz, bsC, 0);
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
  char* scr = smem + wave * (32 * 272);
  const bool interior = (m0 + G256_BM <= M) && (n0 + G256_BN <= N);
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
                     bsA, bsW, bsC, lfm_gemm_debug_flags());
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

// Dispatcher: the ping-pong kernel when the problem fills the chip with 256x256 tiles, v1 otherwise.
// lfm_gemm_select() (0 auto, 1 force v1, 2 force v2) exists for A/B measurements and parity tests of both kernels.
template <class ASrc, class Epi>
static inline int launch_gemm_auto(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t stream,
                                   int batch = 1, long bsA = 0, long bsW = 0, long bsC = 0) {
  const long tiles256 = (long)cdiv(M, 256) * cdiv(N, 256) * batch;
  const int sel = lfm_gemm_selected();
  if (sel == 2 || (sel == 0 && tiles256 >= 192 && N >= 256 && M >= 256)) return launch_gemm256_tn(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
  return launch_gemm_tn(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
}
