// NOT part of the library (round 4: measured and rejected -- profiles/r04_w7_quadrant_phased_rejected.txt).  Bit-identical to the eight-wave kernel on every
// parity case, but 5-7 % slower than the simpler four-wave kernel (csrc/gemm256w_kernel.h): spreading the LDS-DMAs to one per eight MFMAs did not
// remove their cost (no-DMA ablation: still -15 us of 106), i.e. that cost is ENERGY under the board power cap, not issue stalls.  Kept as the record of
// the experiment; it compiled against csrc/ at commit time (include path: lfm_amd/csrc).
// 256x256 MFMA GEMM, one wave per SIMD, QUADRANT-PHASED (v7):  C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue).
//
// Why (round 4 measurements): the four-wave kernel of gemm256w_kernel.h (v6) has the better fragment-read ratio (0.25 ds_read_b128 per MFMA) but
// stages a whole K-tile in the second half of the previous one: 16 LDS-DMAs per wave 4 MFMAs apart.  A wave's VMEM instructions serialise at ~110
// cycles each (tools/ubench/ldsdma_rate.hip: one wave sustains one 1-KiB LDS-DMA per 105-116 cycles in every addressing form), so those issues
// stall the wave -- 14.6 us of its 93.4 us main loop on the fc2 shape (profiles/r04_w6_ablation.txt), as much as in the eight-wave kernel.  Here the
// K-tile is staged the way the eight-wave kernel does it -- four 16-KiB pieces, one per phase, each re-staged as soon as its last reader is done --
// so that a wave issues ONE LDS-DMA per eight MFMAs (134 cycles of matrix work) and eight fragment reads per phase:
//
//   pieces   A0 / A1 = rows {0..63} / {64..127} of both 128-row halves of the A tile, W0 / W1 likewise for the W tile; LDS = 2 K-tiles x 4 pieces
//            = 128 KiB (the image, swizzle and slot order of gemm256_common.h);
//   phases   P0 (A0, W0), P1 (A0, W1), P2 (A1, W1), P3 (A1, W0): one 64 x 64 quadrant of the wave's 128 x 128 block x K = 64 = 32 MFMAs each;
//   reads    every phase reads the 8 fragments the NEXT phase needs and does not hold yet (software pipelining inside the one wave):
//            P0(t): W1(t), P1(t): A1(t), P2(t): A0(t+1), P3(t): W0(t+1) (W0 is double-buffered in registers: P3 still multiplies by W0(t));
//   staging  the piece a phase's reads retire is re-staged in the next phase, for the K-tile two (or three) ahead:
//            P0(t): W0(t+2), P1(t): W1(t+2), P2(t): A1(t+2), P3(t): A0(t+3) -- six pieces (24 LDS-DMAs per wave) always in flight, each with six
//            phases to land;
//   barriers one per phase, behind lgkmcnt(0) (my reads of the piece that is re-staged next are retired) and a counted vmcnt (my share of the piece
//            the next phase reads has landed): RAW and WAR as the guide prescribes (read a staged buffer one phase after the wait that retires it).
// A transposed tile (EpiQKV::transposed) exchanges the two OPERANDS at the staging level (the A pieces then hold W rows, the W pieces activation
// rows), so there is one instruction stream.  Row-major A operands only (buffer-addressed LDS-DMA: SGPR resource + VGPR byte offset + SGPR K offset).
#pragma once
#include "gemm256w_kernel.h"

// ABL (LFM_MEASURE builds; results are garbage): 1 = no LDS-DMA after the prologue, 2 = no fragment reads, 3 = neither
template <class ASrc, class Epi, int ABL = 0>
__global__ __launch_bounds__(256) void gemm256x_tn_kernel(ASrc asrc, const half_t* __restrict__ W, long ldw, int M, int N, int K, int tiles_n,
                                                           Epi epi, long bsA, long bsW, long bsC, int dbg) {
  static_assert(asrc_has_buffer<ASrc>::value, "row-major (buffer-addressed) A operands only");
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
  const int pa = swapped ? wn : wm, pw = swapped ? wm : wn;  // this wave's half of the A pieces / of the W pieces

  // ---- staging.  Piece-local row r' = p * 32 + w * 8 + (lane >> 3) (p = 0..3: the wave's four LDS-DMAs of a piece) <-> tile row
  // (r' >> 6) * 128 + (r' & 63) (+ 64 for the second piece of an operand).  key(r') = (r' >> 1) & 7 = (tid >> 4) & 7.  X = the operand whose rows
  // fill the A pieces (activations, or W for a transposed tile), Y = the other one.
  const int cswz = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
  __amdgpu_buffer_rsrc_t rs_act = asrc.rsrc(), rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, -1, 0x00020000);
  unsigned xv[2][4], yv[2][4];  // [piece 0 / 1][p]: byte offsets of this lane's 16 bytes at k = 0
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int rl = p * 32 + (tid >> 3), row = (rl >> 6) * 128 + (rl & 63) + s * 64;
      const int m = m0 + row, n = n0 + row;
      const unsigned act = asrc.voff(asrc.row(m), cswz);
      const unsigned wgt = ((unsigned)(n < N ? n : N - 1) * (unsigned)ldw + (unsigned)cswz) * 2u;
      xv[s][p] = swapped ? wgt : act;
      yv[s][p] = swapped ? act : wgt;
    }
  const int nk = K / G256Q_BK;
  const int dma_off = wave * 1024;
  // one piece (four LDS-DMAs of this wave) of K-tile kt into its slot of buffer kt & 1; PIECE: 0 A0, 1 W0, 2 W1, 3 A1 (the slot order of the image)
  auto issue_piece = [&](auto PC, int kt) {
    constexpr int PIECE = decltype(PC)::value;
    constexpr int SLOT = PIECE == 0 ? G256Q_SLOT_A0 : (PIECE == 1 ? G256Q_SLOT_B0 : (PIECE == 2 ? G256Q_SLOT_B1 : G256Q_SLOT_A1));
    constexpr bool ISX = PIECE == 0 || PIECE == 3;
    constexpr int S = (PIECE == 0 || PIECE == 1) ? 0 : 1;
    char* slot = smem + (kt & 1) * G256Q_BUF_BYTES + SLOT + dma_off;
    const unsigned soff = (unsigned)kt * (G256Q_BK * 2);
    const bool from_act = ISX != swapped;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const unsigned v = ISX ? xv[S][p] : yv[S][p];
      if (from_act) glds16_buf(rs_act, v, soff, slot + p * 4096);
      else glds16_buf(rs_w, v, soff, slot + p * 4096);
    }
  };
  // one LDS-DMA (index P of the wave's four) of a piece: the form the phases interleave with their MFMAs
  auto issue_one = [&](auto PC, auto PIDX, int kt) {
    constexpr int PIECE = decltype(PC)::value, P = decltype(PIDX)::value;
    constexpr int SLOT = PIECE == 0 ? G256Q_SLOT_A0 : (PIECE == 1 ? G256Q_SLOT_B0 : (PIECE == 2 ? G256Q_SLOT_B1 : G256Q_SLOT_A1));
    constexpr bool ISX = PIECE == 0 || PIECE == 3;
    constexpr int S = (PIECE == 0 || PIECE == 1) ? 0 : 1;
    char* slot = smem + (kt & 1) * G256Q_BUF_BYTES + SLOT + dma_off + P * 4096;
    const unsigned soff = (unsigned)kt * (G256Q_BK * 2);
    const unsigned v = ISX ? xv[S][P] : yv[S][P];
    if (ISX != swapped) glds16_buf(rs_act, v, soff, slot);
    else glds16_buf(rs_w, v, soff, slot);
  };

  f32x4_t acc[2][8][4];  // [W-piece half jw][A-piece tile 0..7][W-piece tile of the half]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- fragment reads: lane (r = lane & 15, q = lane >> 4) reads logical chunk 4 ks + q of piece-local row half * 64 + tile * 16 + r
  const int rkey = ((lane & 15) >> 1) & 7, q4 = lane >> 4;
  int a_addr[2][2], w_addr[2][2];  // [buffer][ks]; slot and 16-row tile index go into the immediate offset
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_addr[b][ks] = b * G256Q_BUF_BYTES + (pa * 64 + (lane & 15)) * 128 + (((ks * 4 + q4) ^ rkey) << 4);
      w_addr[b][ks] = b * G256Q_BUF_BYTES + (pw * 64 + (lane & 15)) * 128 + (((ks * 4 + q4) ^ rkey) << 4);
    }
  half8_t fa0[4][2], fa1[4][2], fw1[4][2], fw0[2][4][2];  // [16-row tile][ks]; W0 double-buffered by K-tile parity
  auto lds_read = [&](half8_t& dst, int addr, auto OFFC) {
    constexpr int OFF = decltype(OFFC)::value;
    if constexpr (ABL >= 2) asm volatile("" : "+v"(dst));
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
  };
  // read R of 8 (tile R >> 1, ks R & 1) of piece PIECE of the K-tile in buffer BUF into its fragment registers (W0: the copy WB)
  auto read_frag = [&](auto PC, auto RC, auto BUFC, auto WBC) {
    constexpr int PIECE = decltype(PC)::value, R = decltype(RC)::value, BUF = decltype(BUFC)::value, WB = decltype(WBC)::value;
    constexpr int TI = R >> 1, KS = R & 1;
    if constexpr (PIECE == 0) lds_read(fa0[TI][KS], a_addr[BUF][KS], g256q_ic<G256Q_SLOT_A0 + TI * 2048>{});
    else if constexpr (PIECE == 3) lds_read(fa1[TI][KS], a_addr[BUF][KS], g256q_ic<G256Q_SLOT_A1 + TI * 2048>{});
    else if constexpr (PIECE == 1) lds_read(fw0[WB][TI][KS], w_addr[BUF][KS], g256q_ic<G256Q_SLOT_B0 + TI * 2048>{});
    else lds_read(fw1[TI][KS], w_addr[BUF][KS], g256q_ic<G256Q_SLOT_B1 + TI * 2048>{});
  };
#define G256X_SB() __builtin_amdgcn_sched_barrier(0)
  auto wait_vm = [&](int groups) {  // at most `groups` pieces (4 LDS-DMAs each) of mine may still be in flight
    switch (groups) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    }
  };

  // One phase of K-tile t (parity PAR = t & 1): 32 MFMAs in 8 groups of four; a fragment read after the first MFMA of every group, one LDS-DMA after
  // the third MFMA of every second group.  RD / ST: this phase's read / staging exists (uniform run-time conditions near the ends of the K range).
  auto phase = [&](auto PHC, auto PARC, auto FULLC, int t, bool rd_rt, bool st_rt) {
    constexpr int PH = decltype(PHC)::value, PAR = decltype(PARC)::value;
    constexpr bool FULL = decltype(FULLC)::value != 0;  // steady state (t + 3 < nk): every read and every staging exists -- no branches in the stream
    const bool rd = FULL || rd_rt, st = FULL || st_rt;
    constexpr int IA = PH >= 2 ? 1 : 0, JW = (PH == 1 || PH == 2) ? 1 : 0;
    // what this phase reads (for a later phase) and stages
    constexpr int RPIECE = PH == 0 ? 2 : (PH == 1 ? 3 : (PH == 2 ? 0 : 1));      // W1(t), A1(t), A0(t+1), W0(t+1)
    constexpr int RBUF = (PH >= 2) ? (PAR ^ 1) : PAR;
    constexpr int SPIECE = PH == 0 ? 1 : (PH == 1 ? 2 : (PH == 2 ? 3 : 0));      // W0(t+2), W1(t+2), A1(t+2), A0(t+3)
    const int skt = PH == 3 ? t + 3 : t + 2;
    g256w_for<8>([&](auto GC) {
      constexpr int G = decltype(GC)::value, TI = G >> 1, KS = G & 1;  // group G: A-piece tile TI at k-step KS against the four W-piece tiles
      auto mf = [&](auto JC) {
        constexpr int J = decltype(JC)::value;
        const half8_t& af = IA ? fa1[TI][KS] : fa0[TI][KS];
        const half8_t& wf = JW ? fw1[J][KS] : fw0[PAR][J][KS];
        g256w_mfma(acc[JW][IA * 4 + TI][J], af, wf);
      };
      mf(g256q_ic<0>{});
      G256X_SB();
      if (rd) read_frag(g256q_ic<RPIECE>{}, GC, g256q_ic<RBUF>{}, g256q_ic<(PAR ^ 1)>{});
      G256X_SB();
      mf(g256q_ic<1>{});
      G256X_SB();
      mf(g256q_ic<2>{});
      G256X_SB();
      if constexpr ((G & 1) == 1 && ABL != 1 && ABL != 3) {
        if (st) issue_one(g256q_ic<SPIECE>{}, g256q_ic<(G >> 1)>{}, skt);
        G256X_SB();
      }
      mf(g256q_ic<3>{});
      G256X_SB();
    });
  };
  // end of a phase: my reads are retired, my share of the piece the NEXT phase reads has landed (at most `groups` younger pieces in flight); barrier
  auto phase_end = [&](int groups) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (ABL == 1 || ABL == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else wait_vm(groups);
    G256_BARRIER();
  };
  auto ktile = [&](auto PARC, auto FULLC, int t) {
    constexpr bool FULL = decltype(FULLC)::value != 0;
    const int a = FULL || t + 1 < nk, b = FULL || t + 2 < nk, c = FULL || t + 3 < nk;
    phase(g256q_ic<0>{}, PARC, FULLC, t, true, b);
    phase_end(FULL ? 6 : 4 * a + 2 * b);  // next phase reads A1(t)
    phase(g256q_ic<1>{}, PARC, FULLC, t, true, b);
    phase_end(FULL ? 6 : 3 * a + 3 * b);  // A0(t+1)
    phase(g256q_ic<2>{}, PARC, FULLC, t, a, b);
    phase_end(FULL ? 6 : 2 * a + 4 * b);  // W0(t+1)
    phase(g256q_ic<3>{}, PARC, FULLC, t, a, c);
    phase_end(FULL ? 6 : a + 4 * b + c);  // W1(t+1)
  };

  // ---- prologue: K-tiles 0 and 1 staged, A0(0) / W0(0) read, A0(2) staged; the loop starts at P0(0) with W1(0) landed
  G256hRowStatRegs rsr;
  if constexpr (epi_has_rowstat<Epi>::value) {
    rsr = g256h_rowstat_load(epi, m0, M);
    G256X_SB();
  }
  issue_piece(g256q_ic<0>{}, 0);
  issue_piece(g256q_ic<1>{}, 0);
  issue_piece(g256q_ic<2>{}, 0);
  issue_piece(g256q_ic<3>{}, 0);
  if (nk > 1) {
    issue_piece(g256q_ic<0>{}, 1);
    issue_piece(g256q_ic<1>{}, 1);
    issue_piece(g256q_ic<2>{}, 1);
    issue_piece(g256q_ic<3>{}, 1);
  }
  if constexpr (epi_has_rowstat<Epi>::value) g256h_rowstat_finish(epi, rsr, smem, m0, M, tile_n);
  if (nk > 1) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // A0(0), W0(0) have landed (six younger pieces may fly)
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if constexpr (ABL >= 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        fa0[i][ks] = fa1[i][ks] = (half8_t){1, 2, 3, 4, 5, 6, 7, 8};
        fw0[0][i][ks] = fw0[1][i][ks] = fw1[i][ks] = (half8_t){1, -1, 1, -1, 1, -1, 1, -1};
      }
  }
  G256_BARRIER();
  g256w_for<8>([&](auto RC) { read_frag(g256q_ic<0>{}, RC, g256q_ic<0>{}, g256q_ic<0>{}); });
  g256w_for<8>([&](auto RC) { read_frag(g256q_ic<1>{}, RC, g256q_ic<0>{}, g256q_ic<0>{}); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  G256_BARRIER();  // every wave has read A0(0): its slot may take A0(2)
  if constexpr (ABL != 1 && ABL != 3) {
    if (nk > 2) issue_piece(g256q_ic<0>{}, 2);
  }
  // the wait of "P3(-1)": W1(0) has landed; younger pieces: A1(0), the four of tile 1, A0(2)
  if constexpr (ABL == 1 || ABL == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else wait_vm(1 + 4 * (nk > 1) + (nk > 2));
  G256_BARRIER();

  {
    int t = 0;
    for (; t + 4 < nk; t += 2) {  // both tiles of the pair in the steady state
      ktile(g256q_ic<0>{}, g256q_ic<1>{}, t);
      ktile(g256q_ic<1>{}, g256q_ic<1>{}, t + 1);
    }
    for (; t < nk; t += 2) {  // the last three or four K-tiles: reads / stagings behind uniform conditions, counted waits from the formulas
      ktile(g256q_ic<0>{}, g256q_ic<0>{}, t);
      if (t + 1 < nk) ktile(g256q_ic<1>{}, g256q_ic<0>{}, t + 1);
    }
  }
#undef G256X_SB
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // inline-asm MFMAs: the compiler does not see the write -> read hazard of their results
  G256_BARRIER();

  epi_batch(epi, bz, bsC, 0);
  if (dbg & 4) return;  // ablation: no epilogue
  if constexpr (epi_is_producer_mod<Epi>::value) {
    g256w_epilogue_mod(acc, smem, epi, m0, n0, n0 / G256_BN, N, wm, wn, lane, wave);
  } else {
    if constexpr (epi_has_transposed<Epi>::value) {
      if (swapped) {  // acc[h][I][jj] = tile (n-tile I, m-tile 4 h + jj): regroup into v5's [m-tile 0..7][n-tile 0..3] per 64-column half
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

template <class ASrc, class Epi, int ABL = 0>
static inline int launch_gemm256x_tn(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t stream,
                                     int batch = 1, long bsA = 0, long bsW = 0, long bsC = 0) {
  if (!asrc_fits(asrc, 0) || !asrc_fits_buffer(asrc, 0)) return LFM_ERR_SHAPE;
  if (M <= 0 || N <= 0 || K <= 0 || (K % G256Q_BK) != 0 || (N % 4) != 0) return LFM_ERR_SHAPE;
  if ((long)N * ldw >= (1L << 31)) return LFM_ERR_SHAPE;
  if ((ldw % 8) != 0 || ((uintptr_t)W & 15)) return LFM_ERR_ALIGN;
  const int tm = cdiv(M, G256_BM), tn = cdiv(N, G256_BN);
  constexpr int LDS = G256Q_LDS_BYTES + (epi_has_rowstat<Epi>::value ? 2048 : 0);
  static unsigned long long attr_set = 0;
  int devid = 0;
  (void)hipGetDevice(&devid);
  if (!((attr_set >> (devid & 63)) & 1)) {
    if (hipFuncSetAttribute((const void*)gemm256x_tn_kernel<ASrc, Epi, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LFM_ERR_LAUNCH;
    attr_set |= 1ull << (devid & 63);
  }
  hipLaunchKernelGGL((gemm256x_tn_kernel<ASrc, Epi, ABL>), dim3(tm * tn, batch), dim3(256), LDS, stream, asrc, W, ldw, M, N, K, tn, epi, bsA, bsW, bsC,
                     lfm_gemm_debug_flags());
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
