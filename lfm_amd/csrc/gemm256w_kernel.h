// 256x256 MFMA GEMM with ONE wave per SIMD (v6):  C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue).
//
// Why (profiles/r03_mainloop_ablation.txt): the 8-wave kernel (gemm256h_kernel.h, v5) keeps the matrix pipe at 62 % because a wave's LOAD part
// (0.375 ds_read_b128 + 0.125 LDS-DMA per MFMA, 450-600 cycles per phase) outlasts its partner's 256 cycles of MFMAs, and neither software
// pipelining nor priorities changed that: only fewer operand bytes per MFMA do.  Here a workgroup is FOUR waves in 2 x 2, each owning a 128 x 128
// block of the 256 x 256 tile:
//   accumulators   8 x 8 tiles of v_mfma_f32_16x16x32_f16 = 256 registers per lane: the wave has its SIMD's whole 512-entry file (launch bound 256
//                  threads, one workgroup per CU by LDS), so the compiler keeps them in the accumulation half and the fragments in the other;
//   fragment reads 8 A + 8 W ds_read_b128 per 64 MFMAs = 0.25 per MFMA (v5: 0.375); LDS-DMA issues per MFMA as v5 (0.125), tile bytes per flop as v5;
//   schedule       no partner wave: a wave overlaps its OWN loads with its own MFMAs.  Fragments are double-buffered in registers (set ks of the
//                  k32 step): while step (t, ks) multiplies, the reads of the next step and -- in the second step of a K-tile -- the sixteen
//                  LDS-DMAs of K-tile t + 2 are issued one at a time between MFMAs (a filler issues in the shadow of a 16-cycle MFMA:
//                  MI355X_MICROARCH "one wave per SIMD": up to 5 fillers per 32-cycle gap hide).  No counted waits inside a step: everything a
//                  step consumes was requested a full step (> 1000 cycles) earlier;
//   LDS            two 64-KiB K-tiles (64 deep, 128-byte rows, chunk c of row r at c ^ ((r >> 1) & 7) as every 128-byte-row kernel here);
//   barriers       ONE per K-tile, between its two steps:  buffer b (tile t) is read during step 1 of tile t - 1 (ks 0) and step 0 of tile t (ks 1);
//                  after the barrier in the middle of tile t every wave has retired those reads (lgkmcnt(0) precedes it), so tile t + 2 may be
//                  staged into b during step 1 of tile t; it is waited for (vmcnt(0), a full K-tile later) before the barrier in the middle of
//                  tile t + 1, behind which its first reads are issued.
// Epilogue: the wave's block is two 128 x 64 halves with v5's accumulator map, handed one after the other to the shared epilogue code.
#pragma once
#include <utility>
#include "gemm256h_kernel.h"

#define G256W_BUF_BYTES 65536
#define G256W_W_OFF 32768
#define G256W_LDS_BYTES (2 * G256W_BUF_BYTES)

template <class F, int... I>
__device__ __forceinline__ void g256w_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(g256q_ic<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void g256w_for(F&& f) {
  g256w_for_impl(f, std::make_integer_sequence<int, N>{});
}

// One MFMA as inline asm with the accumulator TIED and in the accumulation half of the register file: with the builtin the register allocator spread
// the 64 accumulators over both halves and shuffled them on every loop edge (868 v_accvgpr moves in the first build).  W is the A operand: lane l
// owns C[16 i + (l & 15)][16 j + 4 (l >> 4) + r] of tile (i, j), as in v5.
__device__ __forceinline__ void g256w_mfma(f32x4_t& c, const half8_t& a, const half8_t& w) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
}

// Producer epilogue of the folded LayerNorm-modulate for the 4-wave kernel (see g256h_epilogue_mod): wave (wm, wn) owns rows wm * 128 .., column
// quarters 2 wn and 2 wn + 1 of the tile.  The X rows of the NEXT 32-row pass are requested before the stores of the current one are issued (vmcnt
// returns in order: a load behind a pass's 16 stores waits for their round trip -- profiles/r02_epilogue_trace.txt, the "aux" column).
template <class Epi>
__device__ __forceinline__ void g256w_epilogue_mod(f32x4_t (&acc)[2][8][4], char* smem, const Epi& epi, int m0, int n0, int tile_n, int N, int wm, int wn,
                                                   int lane, int wave) {
  char* scr = smem + wave * (32 * 272);
  float* red = (float*)(smem + 8 * 32 * 272);           // [g][quarter][128 rows][2]
  float* cen_s = (float*)(smem + 8 * 32 * 272 + 8192);  // [256 rows]
  const int l15 = lane & 15, l4 = lane >> 4;
  const int rrow = lane >> 4, rcol = lane & 15;
  const int img = m0 / epi.tokens;
  cen_s[threadIdx.x] = epi.cen[m0 + threadIdx.x];  // 256 threads, 256 rows
  f32x4 xo[2][8];
  auto load_x = [&](int buf, int h, int i) {
    const int n = n0 + (2 * wn + h) * 64 + rcol * 4;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) xo[buf][ps] = *(const f32x4*)(epi.X + (long)(m0 + wm * 128 + i * 32 + rrow + ps * 4) * epi.ldx + n);
  };
  load_x(0, 0, 0);
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = n0 + (2 * wn + h) * 64 + rcol * 4;
    const f32x4 bias = *(const f32x4*)(epi.bias + n);
    const f32x4 gate = *(const f32x4*)(epi.gate + (long)img * epi.gate_stride + n);
    const f32x4 sc1 = *(const f32x4*)(epi.scale + (long)img * epi.mod_stride + n) + 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cur = (h * 4 + i) & 1;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4_t*)(scr + (h2 * 16 + l15) * 272 + (j * 16 + l4 * 4) * 4) = acc[h][2 * i + h2][j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (h * 4 + i + 1 < 8) load_x(cur ^ 1, (h * 4 + i + 1) >> 2, (h * 4 + i + 1) & 3);
      const int rl = wm * 128 + i * 32 + rrow;
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const f32x4 v = *(const f32x4*)(scr + (ps * 4 + rrow) * 272 + rcol * 16);
        const float c = cen_s[rl + ps * 4];
        const f32x4 xn = xo[cur][ps] + gate * (v + bias);
        *(f32x4*)(epi.X + (long)(m0 + rl + ps * 4) * epi.ldx + n) = xn;
        const f32x4 d = xn - c;
        const f32x4 ap = d * sc1;
        const half4_t hh = {(half_t)ap.x, (half_t)ap.y, (half_t)ap.z, (half_t)ap.w};
        *(half4_t*)(epi.A + (long)(m0 + rl + ps * 4) * N + n) = hh;
        const float sx = g256h_row16_sum((xn.x + xn.y) + (xn.z + xn.w));
        const float sq = g256h_row16_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w));
        if (rcol == 0) {
          float* dst = red + (((wm * 4 + 2 * wn + h) * 128) + i * 32 + ps * 4 + rrow) * 2;
          dst[0] = sx;
          dst[1] = sq;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (wn == 0) {  // tile-level row sums (fixed order over the four column quarters) -> this tile's slot of the row's partials
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = lane + 64 * rr;
      float sx = 0.f, sq = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        sx += red[((wm * 4 + w4) * 128 + r) * 2];
        sq += red[((wm * 4 + w4) * 128 + r) * 2 + 1];
      }
      *(f32x2*)(epi.part + ((long)(m0 + wm * 128 + r) * epi.tiles_n + tile_n) * 2) = (f32x2){sx, sq};
    }
  }
}

// ABL (LFM_MEASURE builds; results are garbage, timings are the point): 1 = no LDS-DMA after the prologue, 2 = no fragment reads, 3 = neither,
// 4 = 3 without the barrier and the waits in the middle of a K-tile (the bare MFMA stream).
// VAR: where the sixteen LDS-DMAs of a K-tile go in its second step: 0 = one per group of four MFMAs (shipped), 1 = two per group in the first eight groups
//      (measurement builds: 100.8 vs 97.5 us on the fc1 loop)
// OPT bit 0: LDS-DMAs through buffer resources (see gemm256h_kernel.h)
template <class ASrc, class Epi, int ABL = 0, int VAR = 0, int OPT = G256H_DEFAULT_OPT>
__global__ __launch_bounds__(256) void gemm256w_tn_kernel(ASrc asrc, const half_t* __restrict__ W, long ldw, int M, int N, int K, int tiles_n,
                                                           Epi epi, long bsA, long bsW, long bsC, int dbg) {
  static_assert(!epi_has_finish_tile<Epi>::value, "per-lane tile accumulators (GroupNorm statistics) assume one 128 x 64 block per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int tile_m, tile_n;
  g256_tile_order(blockIdx.x, gridDim.x, tiles_n, dbg, tile_m, tile_n);
  const int m0 = tile_m * G256_BM, n0 = tile_n * G256_BN;
  bool swapped = false;
  if constexpr (epi_has_transposed<Epi>::value) swapped = epi.transposed(n0);
  const int bz = blockIdx.y;
  asrc.init(bz, bsA);
  W += (long)bz * bsW;

  // ---- DMA sources: a wave-instruction stages 8 rows x 128 B; pass p of 8 covers rows p * 32 + tid / 8 of the A tile and of the W tile.
  // key(row) = (row >> 1) & 7 = (tid >> 4) & 7 in every pass (32 rows per pass)
  typename ASrc::Row arow[8];
  unsigned woff[8];  // 32-bit element offsets from W (N * ldw < 2^31, checked at launch)
  const int cswz = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    arow[p] = asrc.row(m0 + p * 32 + (tid >> 3));
    const int n = n0 + p * 32 + (tid >> 3);
    woff[p] = (unsigned)((n < N ? n : N - 1) * (int)ldw + cswz);
  }
  const int nk = K / G256Q_BK;
  const int dma_off = wave * 1024;
  constexpr bool BUFDMA = (OPT & 1) != 0 && asrc_has_buffer<ASrc>::value;
  __amdgpu_buffer_rsrc_t rsa, rsw;
  unsigned avoff[8];
  if constexpr (BUFDMA) {
    rsa = asrc.rsrc();
    rsw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, -1, 0x00020000);
#pragma unroll
    for (int p = 0; p < 8; ++p) avoff[p] = asrc.voff(arow[p], cswz);
  }
  auto issue_dma = [&](auto DC, auto BUFC, int kt) {  // DMA d of 16 of K-tile kt into buffer BUF (begin_tile(kt) has been called)
    constexpr int D = decltype(DC)::value, BUF = decltype(BUFC)::value;
    if constexpr (BUFDMA) {
      if constexpr (D < 8) glds16_buf(rsa, avoff[D], asrc.soff(), smem + BUF * G256W_BUF_BYTES + D * 4096 + dma_off);
      else glds16_buf(rsw, woff[D - 8] * 2u, (unsigned)kt * (G256Q_BK * 2), smem + BUF * G256W_BUF_BYTES + G256W_W_OFF + (D - 8) * 4096 + dma_off);
    } else {
      if constexpr (D < 8) glds16(asrc.ptr(arow[D], cswz), smem + BUF * G256W_BUF_BYTES + D * 4096 + dma_off);
      else glds16((W + kt * G256Q_BK) + woff[D - 8], smem + BUF * G256W_BUF_BYTES + G256W_W_OFF + (D - 8) * 4096 + dma_off);
    }
  };

  f32x4_t acc[2][8][4];  // [column half][16-row tile][16-column tile of the half]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // fragment reads: lane (r = lane & 15, q = lane >> 4) reads logical chunk 4 ks + q of row base + r; tile bases are multiples of 16 rows, so
  // the key depends on the lane only.  One address register per (buffer, ks, operand); the 16-row tile index is an immediate offset.
  const int rkey = ((lane & 15) >> 1) & 7, q4 = lane >> 4;
  int a_addr[2][2], w_addr[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ra = b * G256W_BUF_BYTES + (wm * 128 + (lane & 15)) * 128 + (((ks * 4 + q4) ^ rkey) << 4);
      const int rw = b * G256W_BUF_BYTES + G256W_W_OFF + (wn * 128 + (lane & 15)) * 128 + (((ks * 4 + q4) ^ rkey) << 4);
      // a TRANSPOSED tile (EpiQKV::transposed: V^T rows) is the same instruction stream with the two operands' LDS sources exchanged: the "A"
      // fragments are W rows, so tile (i, j) holds C[16 j + 4 (l >> 4) + r][16 i + (l & 15)] -- four consecutive m per lane
      a_addr[b][ks] = swapped ? rw : ra;
      w_addr[b][ks] = swapped ? ra : rw;
    }
  half8_t fa[2][8], fw[2][8];  // [set = ks][16-row tile]
  auto lds_read = [&](half8_t& dst, int addr, auto OFFC) {
    constexpr int OFF = decltype(OFFC)::value;
    if constexpr (ABL >= 2) asm volatile("" : "+v"(dst));
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
  };
  // read r of 16 of the fragments of (buffer BUF, step KS): A tiles 0..7, then W tiles 0..7
  auto read_frag = [&](auto RC, auto BUFC, auto KSC) {
    constexpr int R = decltype(RC)::value, BUF = decltype(BUFC)::value, KS = decltype(KSC)::value;
    if constexpr (R < 8) lds_read(fa[KS][R], a_addr[BUF][KS], g256q_ic<R * 2048>{});
    else lds_read(fw[KS][R - 8], w_addr[BUF][KS], g256q_ic<(R - 8) * 2048>{});
  };
#define G256W_SB() __builtin_amdgcn_sched_barrier(0)

  // One k32 step of K-tile t (buffer BUF): 64 MFMAs on fragment set KS in 16 groups of four, between them the 16 reads of the next step's
  // fragments and, in step 1, the 16 LDS-DMAs of K-tile t + 2 (DMA: 1 always, 0 never, 2 if `dma`).
  auto step = [&](auto KSC, auto BUFC, auto DMAC, int t, bool dma) {
    constexpr int KS = decltype(KSC)::value, BUF = decltype(BUFC)::value, DMA = decltype(DMAC)::value;
    if constexpr (KS == 1 && DMA != 0) asrc.begin_tile(t + 2, G256Q_BK);
    auto mfma = [&](auto IC, auto JC) {
      constexpr int I = decltype(IC)::value, J = decltype(JC)::value;
      g256w_mfma(acc[J >> 2][I][J & 3], fa[KS][I], fw[KS][J]);
    };
    auto dma_slot = [&](auto DC) {
      if constexpr (KS == 1 && DMA != 0) {
        if constexpr (ABL == 1 || ABL >= 3) return;  // (the prologue's two K-tiles are the only DMAs of the ablated builds)
        if constexpr (DMA == 2) {
          if (!dma) return;
        }
        issue_dma(DC, BUFC, t + 2);
        G256W_SB();
      }
    };
    g256w_for<16>([&](auto GC) {
      constexpr int G = decltype(GC)::value, I = G >> 1, J0 = (G & 1) * 4;
      mfma(g256q_ic<I>{}, g256q_ic<J0>{});
      G256W_SB();
      // the next step's fragments: step 0 reads (t, ks 1) from this buffer, step 1 reads (t + 1, ks 0) from the other one
      if constexpr (KS == 0) read_frag(GC, BUFC, g256q_ic<1>{});
      else read_frag(GC, g256q_ic<(BUF ^ 1)>{}, g256q_ic<0>{});
      G256W_SB();
      mfma(g256q_ic<I>{}, g256q_ic<J0 + 1>{});
      G256W_SB();
      if constexpr (VAR == 0) dma_slot(GC);
      else if constexpr (VAR == 1) {
        if constexpr (G < 8) {
          dma_slot(g256q_ic<2 * G>{});
          dma_slot(g256q_ic<2 * G + 1>{});
        }
      }
      mfma(g256q_ic<I>{}, g256q_ic<J0 + 2>{});
      G256W_SB();
      mfma(g256q_ic<I>{}, g256q_ic<J0 + 3>{});
      G256W_SB();
    });
  };
  auto tile = [&](auto BUFC, auto DMAC, int t, bool dma) {
    step(g256q_ic<0>{}, BUFC, DMAC, t, dma);
    // the middle of tile t: my reads of this buffer are retired, my DMAs of tile t + 1 have landed; behind the barrier so have everyone's
    if constexpr (ABL != 4) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      G256_BARRIER();
    }
    step(g256q_ic<1>{}, BUFC, DMAC, t, dma);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next step's fragments (requested a step ago)
    G256W_SB();
  };

  // ---- prologue: K-tiles 0 and 1 in flight, the first fragments
  G256hRowStatRegs rsr;
  if constexpr (epi_has_rowstat<Epi>::value) {
    rsr = g256h_rowstat_load(epi, m0, M);
    G256W_SB();
  }
  asrc.begin_tile(0, G256Q_BK);
  g256w_for<16>([&](auto DC) { issue_dma(DC, g256q_ic<0>{}, 0); });
  if (nk > 1) {
    asrc.begin_tile(1, G256Q_BK);
    g256w_for<16>([&](auto DC) { issue_dma(DC, g256q_ic<1>{}, 1); });
  }
  if constexpr (epi_has_rowstat<Epi>::value) g256h_rowstat_finish(epi, rsr, smem, m0, M, tile_n);
  if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (ABL >= 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        fa[s][i] = (half8_t){1, 2, 3, 4, 5, 6, 7, 8};
        fw[s][i] = (half8_t){1, -1, 1, -1, 1, -1, 1, -1};
      }
  }
  G256_BARRIER();
  g256w_for<16>([&](auto RC) { read_frag(RC, g256q_ic<0>{}, g256q_ic<0>{}); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  G256W_SB();

  {
    int t = 0;
    for (; t + 3 < nk; t += 2) {
      tile(g256q_ic<0>{}, g256q_ic<1>{}, t, true);
      tile(g256q_ic<1>{}, g256q_ic<1>{}, t + 1, true);
    }
    for (; t < nk; t += 2) {  // the last two or three K-tiles: the DMA slots behind a uniform branch
      tile(g256q_ic<0>{}, g256q_ic<2>{}, t, t + 2 < nk);
      if (t + 1 < nk) tile(g256q_ic<1>{}, g256q_ic<2>{}, t + 1, t + 3 < nk);
    }
  }
#undef G256W_SB
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // inline-asm MFMAs: the compiler does not see the write -> read hazard of their results
  G256_BARRIER();  // every wave's last fragment reads are retired: the K-tile buffers are free for the epilogue scratch

  epi_batch(epi, bz, bsC, 0);
  if (dbg & 4) return;  // ablation: no epilogue
  if constexpr (epi_is_producer_mod<Epi>::value) {
    g256w_epilogue_mod(acc, smem, epi, m0, n0, n0 / G256_BN, N, wm, wn, lane, wave);
  } else {
    if constexpr (epi_has_transposed<Epi>::value) {
      if (swapped) {  // acc[h][I][jj] = tile (n-tile I, m-tile 4 h + jj): regroup into v5's [m-tile 0..7][n-tile 0..3] per 64-column half (register renaming)
#pragma unroll
        for (int hn = 0; hn < 2; ++hn) {
          f32x4_t view[8][4];
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) view[i][j] = acc[i >> 2][4 * hn + j][i & 3];
          g256h_epilogue_body<G256_BN>(view, smem, epi, m0, n0, M, N, wm, 2 * wn + hn, lane, wave, dbg, true, false);
        }
        return;
      }
    }
    g256h_epilogue_body<G256_BN>(acc[0], smem, epi, m0, n0, M, N, wm, 2 * wn, lane, wave, dbg, false, false);
    g256h_epilogue_body<G256_BN>(acc[1], smem, epi, m0, n0, M, N, wm, 2 * wn + 1, lane, wave, dbg, false, false);
  }
}

template <class ASrc, class Epi, int ABL = 0, int VAR = 0, int OPT = G256H_DEFAULT_OPT>
static inline int launch_gemm256w_tn(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t stream,
                                     int batch = 1, long bsA = 0, long bsW = 0, long bsC = 0) {
  if (!asrc_fits(asrc, 0)) return LFM_ERR_SHAPE;
  if (M <= 0 || N <= 0 || K <= 0 || (K % G256Q_BK) != 0 || (N % 4) != 0) return LFM_ERR_SHAPE;
  if ((long)N * ldw >= (1L << 31)) return LFM_ERR_SHAPE;  // W rows are 32-bit element offsets (byte offsets with OPT bit 0)
  if constexpr ((OPT & 1) != 0 && asrc_has_buffer<ASrc>::value) {
    if (!asrc_fits_buffer(asrc, 0)) return LFM_ERR_SHAPE;
  }
  if ((ldw % 8) != 0 || ((uintptr_t)W & 15)) return LFM_ERR_ALIGN;
  const int tm = cdiv(M, G256_BM), tn = cdiv(N, G256_BN);
  constexpr int LDS = G256W_LDS_BYTES + (epi_has_rowstat<Epi>::value ? 2048 : 0);  // + rs[256][2] of the folded LayerNorm consumers
  static lfm_device_mask attr_set{0};  // one bit per device: the attribute is per (function, device)
  const unsigned long long dbit = lfm_device_bit();
  if (lfm_device_todo(attr_set, dbit)) {
    if (hipFuncSetAttribute((const void*)gemm256w_tn_kernel<ASrc, Epi, ABL, VAR, OPT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LFM_ERR_LAUNCH;
    lfm_device_done(attr_set, dbit);
  }
  hipLaunchKernelGGL((gemm256w_tn_kernel<ASrc, Epi, ABL, VAR, OPT>), dim3(tm * tn, batch), dim3(256), LDS, stream, asrc, W, ldw, M, N, K, tn, epi, bsA,
                     bsW, bsC, lfm_gemm_debug_flags());
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
