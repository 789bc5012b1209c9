// QKV projection + attention core of a DiT block in ONE kernel (timm Attention as called at /root/reference/models/DiT.py:120, folded LayerNorm-modulate
// path; 256 tokens per image, head_dim 64):  O[img, :, head] = softmax(Q K^T hd^-0.5) V  with  [Q | K | V] = rowaffine(A' W_head^T)  never leaving the CU.
//
// Why (round 6): as two kernels the block writes Q | K | V^T to HBM (100.7 MB at 64 images of DiT-L/2) in an epilogue that no MFMA overlaps and reads them
// back in an attention kernel whose memory and compute phases overlap by two thirds only (profiles/r06_final_bench_kernel_stats.csv: 101.6 + 29.5 us).  One
// 256-token image x one head is exactly one 256-row tile x 192 columns of the QKV GEMM, and its Q, K, V^T (96 KiB of fp16) fit the LDS the operand ring
// leaves behind: a workgroup = (image, head) runs the K loop of gemm256h_kernel.h on a 256 x 192 tile, applies the folded-LayerNorm row affine
// a[m] acc + (b[m] u[n] + v[n]) straight from the accumulators into the LDS images the attention kernel would have staged (K [256][64], V^T [64][256] in the
// vt_pos token order, Q beside them) and then runs the key loop of dit_attention_kernel<256, 1, 64> on them.  HBM sees A' (through the L2: 16 heads share an
// image's panel), W and O: 266 MB -> ~75 MB per block.
//
// Same arithmetic, same order: the K loop accumulates the K-tiles in the same sequence on the same MFMA (v_mfma_f32_16x16x32_f16), the affine is row_affine4 /
// fma_v as in EpiQKVMod, the softmax block is att_softmax_block, S^T / O^T MFMAs and the normalisation as in the per-item kernel => O is BIT-IDENTICAL to
// launch_gemm256h_tn<EpiQKVMod> + attention_launch (tests/test_gpu_dit.py::test_fused_qkv_attention_matches_two_kernels).
//
// K loop = gemm256h_kernel.h's (quadrant phases, piece-granular LDS-DMA ring, ping-pong wave groups, one barrier per phase, counted waits) with a 192-column
// tile: wave (g, wn) owns rows g*128.. x 48 columns of the head, gathered by the DMA source addresses of the W rows -- accumulator tiles j = 0, 1: 32 dims of
// Q (waves wn = 0, 1) or of K (wn = 2, 3), interleaved in fours so that a lane's two tiles are EIGHT CONSECUTIVE dims of a row (one 16-byte write in the
// hand-over); tile j = 2: V dims 16 wn .. + 15.  Piece B0 = the Q and K rows (128 rows, 16 KiB, as before), piece B1 = the V rows (64 rows, 8 KiB: ONE LDS-DMA
// per thread), so the phases are A0xB0 (16 MFMAs) | A0xB1 (8) | A1xB1 (8) | A1xB0 (16); the seven LDS-DMAs of a K-tile are spread over the LOAD parts as 0 | 3 | 1 | 3
// (against the parts' 12 | 2 | 8 | 0 fragment reads), counted vmcnt waits 4 5 4 6.  The V tile is
// issued with the operands swapped (a lane then owns FOUR CONSECUTIVE TOKENS of one head dim: half a 16-byte chunk of a V^T row, the other half in lane +- 32).
//
// PERSISTENT workgroups (one per CU, items strided over the grid).  Measured with one workgroup per item (profiles/r06_fused_qkv_attention.txt): of 122 us the
// K loop beyond its first tile takes 74, the key loop 32 and everything else 28 = 7 us per item -- workgroup dispatch, the row statistics, the first LDS-DMA round
// trip, one K-tile, the hand-over, the stores.  Here the NEXT item's first four operand pieces, its row statistics and column constants are requested inside
// the key loop (the hand-over puts K / V^T into the ring's second half and Q above the ring, so the first half is free from the K loop's last barrier on), and
// the two pieces that land in the second half right behind the key loop: an item starts its K loop on operands that are already there.
// VMEM bookkeeping (vmcnt counts loads, LDS-DMAs and stores alike and retires in issue order, as in attention_stream_kernel.h): per wave and item, in issue
// order, inside the key loop {column-constant and (waves 0-3) row-statistic loads interleaved with A0(0) A0(0) B0(0) B0(0) B1(0)} A1(0) A1(0), then 4 output stores,
// A0(1) A0(1) B0(1) B0(1): the K loop may start once B1(0) has landed = all but the newest 10 (the first item of a workgroup: 6, there are no stores in between).
#pragma once
#include "attention_kernel.h"
#include "gemm256h_kernel.h"

struct QkvAttnArgs {
  const float* u;  // [rows][3 D] (+ img * uv_stride): column constants of the folded LayerNorm-modulate (u = sum_k (1 + s_k) W_nk)
  const float* v;  //                                 (v = sum_k sh_k W_nk + bias)
  long uv_stride;
  RowStatSrc st;
  half_t* O;  // [M, D] token-major, head-major columns
  int D, heads;
  float scale_log2e;
  const float* rs;  // (kernel) (a, b) of the tile's 256 rows in LDS
  int m0;
};

#define QKVA_LDS_BYTES (G256Q_LDS_BYTES + 8 * 4096)  // operand ring | Q rows = output staging (4 KiB per wave)

__global__ __launch_bounds__(512) void qkv_attention_kernel(const half_t* __restrict__ A, long lda, const half_t* __restrict__ W, long ldw, int M, int K,
                                                            QkvAttnArgs ep, int items, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wn = wave & 3;
  const int D = ep.D;

  // item = (image, head) in the tile order of the 256-row GEMMs (an XCD walks 4- or 8-image x all-heads patches: few A' panels and W slices per L2); a workgroup's
  // items are gridDim.x apart -- a multiple of 8 when it matters, so they stay on its XCD's range
  int item = blockIdx.x;
  int tile_m, head;
  g256_tile_order(item, items, ep.heads, dbg, tile_m, head);
  int m0 = tile_m * G256_BM;

  // ---- DMA sources (piece = rows x 128 B; thread tid stages 16-byte chunk tid of an 8-KiB issue; chunk c of LDS row r holds logical chunk c ^ ((r >> 1) & 7))
  const int cswz = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, -1, 0x00020000);
  unsigned avoff[2][2], wvoff0[2], wvoff1;
  auto set_item = [&](int m0_, int head_) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int p = 0; p < 2; ++p) avoff[s][p] = ((unsigned)(m0_ + p * 128 + s * 64 + (tid >> 3)) * (unsigned)lda + (unsigned)cswz) * 2u;
#pragma unroll
    for (int p = 0; p < 2; ++p) {  // B0: LDS row wn' * 32 + c = column cc = c & 15 of wave wn''s tile c >> 4: dim 32 (wn' & 1) + 8 (cc >> 2) + 4 (c >> 4) + (cc & 3) of Q (wn' < 2) / K
      const int wnp = p * 2 + (tid >> 8), c = (tid >> 3) & 31, cc = c & 15;
      const int n = (wnp >> 1) * D + head_ * 64 + (wnp & 1) * 32 + (cc >> 2) * 8 + (c >> 4) * 4 + (cc & 3);
      wvoff0[p] = ((unsigned)n * (unsigned)ldw + (unsigned)cswz) * 2u;
    }
    wvoff1 = ((unsigned)(2 * D + head_ * 64 + (tid >> 3)) * (unsigned)ldw + (unsigned)cswz) * 2u;  // B1: LDS row = V dim
  };
  set_item(m0, head);
#ifdef LFM_MEASURE  // phase split (tools/fused_qkv_phases.py; results are garbage): flag 67108864 = two K-tiles only, 33554432 = no key loop
  const int nk = (dbg & 67108864) ? 2 : K / G256Q_BK;
#else
  const int nk = K / G256Q_BK;
#endif
  const int dma_off = wave * 1024;
  auto issue_a = [&](int s, int kt, char* slot) {
    glds16_buf(rsa, avoff[s][0], (unsigned)kt * (G256Q_BK * 2), slot + dma_off);
    glds16_buf(rsa, avoff[s][1], (unsigned)kt * (G256Q_BK * 2), slot + 8192 + dma_off);
  };
  auto issue_b0 = [&](int kt, char* slot) {
    glds16_buf(rsw, wvoff0[0], (unsigned)kt * (G256Q_BK * 2), slot + dma_off);
    glds16_buf(rsw, wvoff0[1], (unsigned)kt * (G256Q_BK * 2), slot + 8192 + dma_off);
  };
  auto issue_b1 = [&](int kt, char* slot) { glds16_buf(rsw, wvoff1, (unsigned)kt * (G256Q_BK * 2), slot + dma_off); };

  f32x4_t acc[8][3];

  const int l15 = lane & 15, l4 = lane >> 4;
  const int rkey = (l15 >> 1) & 7;
  int a_addr[2], w_addr[2], w1_addr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_addr[ks] = (g * 64 + l15) * 128 + (((ks * 4 + l4) ^ rkey) << 4);   // + i4 * 2048
    w_addr[ks] = (wn * 32 + l15) * 128 + (((ks * 4 + l4) ^ rkey) << 4);  // + j * 2048 (tiles j = 0, 1)
    w1_addr[ks] = (wn * 16 + l15) * 128 + (((ks * 4 + l4) ^ rkey) << 4);
  }
  half8_t af[4][2], wf[3][2];
  auto lds_read = [&](half8_t& dst, int addr, auto OFFC) {
    constexpr int OFF = decltype(OFFC)::value;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
  };
#define QKVA_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define QKVA_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

  // LOAD part of phase PH of K-tile t: fragment reads in consumption order, one piece staged, counted wait (three pieces stay in flight)
  auto load_part = [&](auto PHC, auto BUFC, int t, bool s1, bool s2) {
    constexpr int PH = decltype(PHC)::value, BUF = decltype(BUFC)::value, HI = BUF * G256Q_BUF_BYTES;
    char* cur = smem + BUF * G256Q_BUF_BYTES;
    char* oth = smem + (BUF ^ 1) * G256Q_BUF_BYTES;
    if constexpr (PH == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        lds_read(wf[0][ks], w_addr[ks] + HI, g256q_ic<G256Q_SLOT_B0>{});
        lds_read(wf[1][ks], w_addr[ks] + HI, g256q_ic<G256Q_SLOT_B0 + 2048>{});
        lds_read(af[0][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A0>{});
        lds_read(af[1][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A0 + 2048>{});
        lds_read(af[2][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A0 + 4096>{});
        lds_read(af[3][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A0 + 6144>{});
      }
    } else if constexpr (PH == 1) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_read(wf[2][ks], w1_addr[ks] + HI, g256q_ic<G256Q_SLOT_B1>{});
    } else if constexpr (PH == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        lds_read(af[0][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A1>{});
        lds_read(af[1][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A1 + 2048>{});
        lds_read(af[2][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A1 + 4096>{});
        lds_read(af[3][ks], a_addr[ks] + HI, g256q_ic<G256Q_SLOT_A1 + 6144>{});
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // Which LDS-DMAs go out in which LOAD part is balanced against the part's fragment reads (12 | 2 | 8 | 0): none | B1(t+1) A1(t+1) A1(t+1) | A0(t+2) | A0(t+2)
    // B0(t+2) B0(t+2).  A barrier interval lasts as long as the slower wave group's (LOAD + MFMA) pair -- with a DMA at ~110 cycles of the issuing wave and a read
    // at ~30, the gemm256h placement (1 | 2 | 2 | 2 here) makes the 12-read part the longest of all; see profiles/r06_dma_phase_balance.txt (-0.25 % per forward:
    // far less than that model says; splitting the twelve READS of phase 0 over two parts as well -- B0 fragments of the next K-tile prefetched in part 3 -- was
    // slower by 0.8 %).  Legal: a slot is
    // refilled after both groups' reads of it (A0 / B0 of this tile: read in LOAD 0, one interval apart), and every piece still has >= 3 intervals to land.
    if constexpr (PH == 1) {
      if (s1) {
        issue_b1(t + 1, oth + G256Q_SLOT_B1);
        issue_a(1, t + 1, oth + G256Q_SLOT_A1);
      }
    } else if constexpr (PH == 2) {
      if (s2) glds16_buf(rsa, avoff[0][0], (unsigned)(t + 2) * (G256Q_BK * 2), cur + G256Q_SLOT_A0 + dma_off);
    } else if constexpr (PH == 3) {
      if (s2) {
        glds16_buf(rsa, avoff[0][1], (unsigned)(t + 2) * (G256Q_BK * 2), cur + G256Q_SLOT_A0 + 8192 + dma_off);
        issue_b0(t + 2, cur + G256Q_SLOT_B0);
      }
    }
    // counted waits: what the LOAD part after next reads has landed (a barrier lies in between); newer DMAs may still fly.  Issue order per K-tile:
    // B1 A1 A1 | A0 | A0 B0 B0
    if (s2) {
      if constexpr (PH == 0) QKVA_VMCNT(4);       // A1(t) landed; A0(t+1) x2, B0(t+1) x2 may fly
      else if constexpr (PH == 1) QKVA_VMCNT(5);  // A0(t+1) landed; B0(t+1) x2, B1(t+1), A1(t+1) x2
      else if constexpr (PH == 2) QKVA_VMCNT(4);  // B0(t+1) landed; B1(t+1), A1(t+1) x2, A0(t+2)
      else QKVA_VMCNT(6);                         // B1(t+1) landed; A1(t+1) x2, A0(t+2) x2, B0(t+2) x2
    } else if (s1) {
      if constexpr (PH == 0) QKVA_VMCNT(4);
      else if constexpr (PH == 1) QKVA_VMCNT(5);
      else if constexpr (PH == 2) QKVA_VMCNT(3);  // B0(t+1) landed; B1(t+1), A1(t+1) x2
      else QKVA_VMCNT(2);                         // B1(t+1) landed; A1(t+1) x2
    } else QKVA_VMCNT(0);
  };
  // MFMA part of phase PH: A sub (64 rows) x {Q, K tiles | V tile} x K = 64
  auto mfma_part = [&](auto PHC) {
    constexpr int PH = decltype(PHC)::value;
    constexpr int I0 = (PH >= 2) ? 4 : 0;
    constexpr bool VT = (PH == 1 || PH == 2);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        if constexpr (PH == 0) {  // 12 reads, per k32 step: W0 W1 A0 A1 A2 A3
          if (ks == 0) {
            if (i4 == 0) QKVA_LGKM(9);
            else if (i4 == 1) QKVA_LGKM(8);
            else if (i4 == 2) QKVA_LGKM(7);
            else QKVA_LGKM(6);
          } else {
            if (i4 == 0) QKVA_LGKM(3);
            else if (i4 == 1) QKVA_LGKM(2);
            else if (i4 == 2) QKVA_LGKM(1);
            else QKVA_LGKM(0);
          }
        } else if constexpr (PH == 1) {  // 2 reads: W2 (ks 0), W2 (ks 1)
          if (i4 == 0) {
            if (ks == 0) QKVA_LGKM(1);
            else QKVA_LGKM(0);
          }
        } else if constexpr (PH == 2) {  // 8 reads: A0..A3 (ks 0), A0..A3 (ks 1)
          if (ks == 0) {
            if (i4 == 0) QKVA_LGKM(7);
            else if (i4 == 1) QKVA_LGKM(6);
            else if (i4 == 2) QKVA_LGKM(5);
            else QKVA_LGKM(4);
          } else {
            if (i4 == 0) QKVA_LGKM(3);
            else if (i4 == 1) QKVA_LGKM(2);
            else if (i4 == 2) QKVA_LGKM(1);
            else QKVA_LGKM(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (VT) {  // operands swapped: acc[i][2][r] = C[m = 16 i + 4 l4 + r][V dim 16 wn + l15]
          acc[I0 + i4][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i4][ks], wf[2][ks], acc[I0 + i4][2], 0, 0, 0);
        } else {  // acc[i][j][r] = C[m = 16 i + l15][dim 32 (wn & 1) + 8 l4 + 4 j + r of Q / K]
#pragma unroll
          for (int j2 = 0; j2 < 2; ++j2) acc[I0 + i4][j2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j2][ks], af[i4][ks], acc[I0 + i4][j2], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue of the workgroup's first item: row statistics and column constants requested BEFORE the first DMAs, pieces A0(0) B0(0) B1(0) A1(0) A0(1) B0(1)
  G256hRowStatRegs rsr = g256h_rowstat_load(ep, m0, M);
  f32x4 uq, vq, uk, vk;
  float uv_, vv_;
  auto load_consts = [&](int tile_m_, int head_) {
    const long uvh = (long)tile_m_ * ep.uv_stride + head_ * 64;
    const long uvo = uvh + (wn >> 1) * D + (wn & 1) * 32 + 8 * l4;  // the lane's eight consecutive dims of Q (waves 0, 1) / K (waves 2, 3): tile 0 = the first four, tile 1 = the others
    uq = *(const f32x4*)(ep.u + uvo), vq = *(const f32x4*)(ep.v + uvo);
    uk = *(const f32x4*)(ep.u + uvo + 4), vk = *(const f32x4*)(ep.v + uvo + 4);
    uv_ = ep.u[uvh + 2 * D + wn * 16 + l15], vv_ = ep.v[uvh + 2 * D + wn * 16 + l15];
  };
  load_consts(tile_m, head);
  __builtin_amdgcn_sched_barrier(0);
  issue_a(0, 0, smem + G256Q_SLOT_A0);
  issue_b0(0, smem + G256Q_SLOT_B0);
  issue_b1(0, smem + G256Q_SLOT_B1);
  issue_a(1, 0, smem + G256Q_SLOT_A1);
  issue_a(0, 1, smem + G256Q_BUF_BYTES + G256Q_SLOT_A0);
  issue_b0(1, smem + G256Q_BUF_BYTES + G256Q_SLOT_B0);
  // (a, b) = (rstd, -rstd (mu - c)) of an item's 256 rows (g256h_rowstat_finish with its own destination): the upper 8 KiB of the first ring half's B1 slot, which
  // no LDS-DMA ever touches (piece B1 is 8 KiB of a 16-KiB slot) -- written for item k + 1 at the end of item k, read at item k + 1's hand-over
  float* const rs = (float*)(smem + G256Q_SLOT_B1 + 8192);
  auto rowstat_finish = [&](const G256hRowStatRegs& r, int m0_, int head_) {
    if (threadIdx.x < 256) {
      float sx = 0.f, sq = 0.f;
#pragma unroll
      for (int t = 0; t < G256H_MAX_PARTS; ++t) {  // fixed order
        sx += r.p[t].x;
        sq += r.p[t].y;
      }
      const float mu = sx * ep.st.inv_n, dl = mu - r.c;
      const float var = fmaxf(sq * ep.st.inv_n - dl * dl, 0.f);
      const float rstd = rsqrtf(var + ep.st.eps);
      *(f32x2*)(rs + 2 * threadIdx.x) = (f32x2){rstd, -rstd * dl};
      if (head_ == 0 && m0_ + (int)threadIdx.x < M) ep.st.cen_out[m0_ + threadIdx.x] = mu;  // head 0 publishes the row means: the next producer's centring constants
    }
  };
  rowstat_finish(rsr, m0, head);
  auto run = [&](auto GC) {
    constexpr int G = decltype(GC)::value;
    if constexpr (G == 0) load_part(g256q_ic<0>{}, g256q_ic<0>{}, 0, 1 < nk, 2 < nk);
    G256_BARRIER();
    auto tile = [&](auto BUFC, int t) {
      constexpr int BUF = decltype(BUFC)::value;
      const bool s1 = t + 1 < nk, s2 = t + 2 < nk, s3 = t + 3 < nk;
      auto phase = [&](auto PHC) {
        constexpr int PH = decltype(PHC)::value;
        if constexpr (G == 0) {
          mfma_part(PHC);
          if constexpr (PH < 3) load_part(g256q_ic<PH + 1>{}, BUFC, t, s1, s2);
          else if (s1) load_part(g256q_ic<0>{}, g256q_ic<(BUF ^ 1)>{}, t + 1, s2, s3);
        } else {
          load_part(PHC, BUFC, t, s1, s2);
          mfma_part(PHC);
        }
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
  // LDS in the attention phase: K [256][64] and V^T [64][256] in the SECOND ring half, Q [256][64] above the ring -- the first ring half stays free, so the next
  // item's first four operand pieces can be requested from the moment the K loop ends.  A wave's own 32 Q rows (4 KiB) are its output staging once it holds its Q
  // fragments.  Q, K: 128-byte rows, chunk c of row r at c ^ ((r >> 1) & 7); V^T: 512-byte rows, tokens of a 16-group in the vt_pos order, chunk c of row d at
  // c ^ (d & 15) -- the images dit_attention_kernel<256, 1, 64> stages from HBM.
  char* const Qs = smem + G256Q_LDS_BYTES;
  char* const Ks = smem + G256Q_BUF_BYTES;
  char* const Vs = smem + G256Q_BUF_BYTES + 32768;
  char* const ob = Qs + wave * 4096;

  const float scale_log2e = ep.scale_log2e;
  bool first = true;
  // (measurement builds, flag 2: s_memtime stamps of waves 0 and 4 of workgroup 0 -> att_trace[0..31] / [32..63], read with lfm_attention_trace_read; per item k, slot
  // 7 k + {0 loop top, 1 operands landed (barrier), 2 K loop done, 3 hand-over done (barrier), 4 next item requested, 5 key loop done, 6 stores issued})
  [[maybe_unused]] int tr_k = 0;
  auto stamp = [&](int slot) {
#ifdef LFM_MEASURE
    if ((dbg & 2) && blockIdx.x == 0 && (wave & 3) == 0 && tr_k < 4) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      if (lane == 0) att_trace[(wave >> 2) * 32 + 7 * tr_k + slot] = t;
    }
#else
    (void)slot;
#endif
  };
#pragma unroll 1
  for (;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    stamp(0);
    if (first) QKVA_VMCNT(6);
    else QKVA_VMCNT(10);
    G256_BARRIER();
    stamp(1);
    if (g == 0) run(g256q_ic<0>{});
    else run(g256q_ic<1>{});
    stamp(2);

    // ---- hand-over: the ring is dead (every wave is past the last barrier, its fragment reads retired before its MFMAs).  (The lane id is laundered per phase:
    // the phases' per-lane LDS addresses are loop invariants the compiler would otherwise keep in registers across the K loop, which has none to spare.)
    {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int l15 = ln & 15, l4 = ln >> 4;
      const int d = wn * 16 + l15;  // V dim of this lane
      // the (a, b) rows first, four 16-row tiles at a time: reads behind the Q / K / V^T writes could not be moved up by the compiler (same address space), and a
      // read - compute - write chain per tile costs an LDS round trip each (measured: 3600 ticks of an item's 52600, tools/fused_qkv_trace.py)
#pragma unroll
      for (int ih = 0; ih < 2; ++ih) {
        f32x2 ab[4];
        f32x4 r0[4], r1[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = ih * 4 + ii;
          ab[ii] = *(const f32x2*)(rs + 2 * (g * 128 + i * 16 + l15));
          const int t0 = g * 128 + i * 16 + 4 * l4;  // tokens t0 .. t0 + 3 of V dim d
          r0[ii] = *(const f32x4*)(rs + 2 * t0);
          r1[ii] = *(const f32x4*)(rs + 2 * t0 + 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        char* const QKs = wn < 2 ? Qs : Ks;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = ih * 4 + ii;
          const int m = g * 128 + i * 16 + l15;
          // tiles 0 and 1 of the wave are the two halves of EIGHT consecutive dims of one matrix (the W rows were gathered that way): one 16-byte write per row
          const f32x4 lo = row_affine4(ab[ii].x, ab[ii].y, acc[i][0], uq, vq);
          const f32x4 hi = row_affine4(ab[ii].x, ab[ii].y, acc[i][1], uk, vk);
#ifdef LFM_MEASURE  // (hand-over ablation: flag 524288 = no Q / K writes, 262144 = no V^T writes)
          if (!(dbg & 524288))
#endif
            *(half8_t*)(QKs + m * 128 + (((4 * (wn & 1) + l4) ^ ((m >> 1) & 7)) << 4)) =
                (half8_t){(half_t)lo.x, (half_t)lo.y, (half_t)lo.z, (half_t)lo.w, (half_t)hi.x, (half_t)hi.y, (half_t)hi.z, (half_t)hi.w};
        }
        // V^T: a lane holds tokens 4 l4 .. + 3 of 16-token group 8 g + i for V dim d, i.e. positions {0, 8, 4, 12}[l4] .. + 3 of the group (vt_pos): lanes l and l + 32
        // (l4 and l4 + 2) hold the two halves of one 16-byte chunk.  One v_permlane32_swap per register over a PAIR of groups gives the lower half-wave the whole
        // chunk of the even group and the upper half-wave that of the odd one: a 16-byte write per lane and pair.
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
          half4_t hv[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int ii = ip * 2 + e, i = ih * 4 + ii;
            hv[e] = (half4_t){(half_t)fma_v(r0[ii].x, acc[i][2][0], fma_v(r0[ii].y, uv_, vv_)), (half_t)fma_v(r0[ii].z, acc[i][2][1], fma_v(r0[ii].w, uv_, vv_)),
                              (half_t)fma_v(r1[ii].x, acc[i][2][2], fma_v(r1[ii].y, uv_, vv_)), (half_t)fma_v(r1[ii].z, acc[i][2][3], fma_v(r1[ii].w, uv_, vv_))};
          }
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 ea = __builtin_bit_cast(u32x2, hv[0]), eb = __builtin_bit_cast(u32x2, hv[1]);
          const auto s0 = __builtin_amdgcn_permlane32_swap(ea.x, eb.x, false, false);  // [0]: (even group: own | odd group: lane - 32's), [1]: (even: lane + 32's | odd: own)
          const auto s1 = __builtin_amdgcn_permlane32_swap(ea.y, eb.y, false, false);
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 chunk = {s0[0], s1[0], s0[1], s1[1]};  // positions 0-3 (tokens 0-3 / 4-7 of the group), then 4-7 (tokens 8-11 / 12-15)
          const int G2 = 8 * g + ih * 4 + ip * 2 + (l4 >> 1);  // the group this lane writes
#ifdef LFM_MEASURE
          if (!(dbg & 262144))
#endif
            *(u32x4*)(Vs + d * 512 + (((2 * G2 + (l4 & 1)) ^ (d & 15)) << 4)) = chunk;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G256_BARRIER();
    stamp(3);

    // ---- attention: wave w owns queries 32 w .. 32 w + 31 (dit_attention_kernel<256, 1, 64>: S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x16_f16).
    // (Measured and not kept: the SIMD's two waves taking turns at s_setprio 1 per pair of key blocks -- the second-dispatched wave group runs its key loop
    // ~25 % longer than the first, 9400 vs 7400 ticks -- 10448 vs 10463 us per DiT-L/2 forward.)
    int lna = lane;
    asm volatile("" : "+v"(lna));
    const int hsel = lna >> 5, l31 = lna & 31;
    const int akey = (l31 >> 1) & 7;
    half8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const half8_t*)(Qs + (wave * 32 + l31) * 128 + (((ks * 2 + hsel) ^ akey) << 4));
    // the workgroup's next item: the first ring half has been free since the K loop's last barrier -- its first four operand pieces (seven LDS-DMAs), its row
    // statistics and column constants are requested under the key loop, one group per half key block: a VMEM issue stalls the wave ~110 cycles
    // (tools/ubench/ldsdma_rate.hip), which the SIMD's other wave can use inside the key loop and nobody can in front of it.  Order (the counted wait at the loop
    // top relies on it): every plain load goes out BEFORE the fifth DMA (piece B1(0)).
    const int nitem = item + (int)gridDim.x;
    const bool has_next = nitem < items;
    int ntile_m = 0, nhead = 0;
    long nuvh = 0;
    const float* nrow_part = nullptr;
    int nrow_m = 0;
    if (has_next) {
      g256_tile_order(nitem, items, ep.heads, dbg, ntile_m, nhead);
      set_item(ntile_m * G256_BM, nhead);
      nuvh = (long)ntile_m * ep.uv_stride + nhead * 64;
      nrow_m = ntile_m * G256_BM + (int)threadIdx.x;
      nrow_part = ep.st.part + (long)nrow_m * ep.st.tiles_p * 2;
    }
    auto next_req = [&](int idx) {
      if (!has_next) return;
      __builtin_amdgcn_sched_barrier(0);
      const long uvo = nuvh + (wn >> 1) * D + (wn & 1) * 32 + 8 * l4;
      const bool rows = threadIdx.x < 256;  // waves 0-3: a row's statistics each (g256h_rowstat_load, spread)
      auto part = [&](int t) { return rows && t < ep.st.tiles_p ? ((const f32x2*)nrow_part)[t] : (f32x2){0.f, 0.f}; };
      if (idx == 0) {
        uq = *(const f32x4*)(ep.u + uvo);
        rsr.p[0] = part(0);
        __builtin_amdgcn_sched_barrier(0);
        glds16_buf(rsa, avoff[0][0], 0u, smem + G256Q_SLOT_A0 + dma_off);
      } else if (idx == 1) {
        vq = *(const f32x4*)(ep.v + uvo);
        rsr.p[1] = part(1);
        __builtin_amdgcn_sched_barrier(0);
        glds16_buf(rsa, avoff[0][1], 0u, smem + G256Q_SLOT_A0 + 8192 + dma_off);
      } else if (idx == 2) {
        uk = *(const f32x4*)(ep.u + uvo + 4);
        rsr.p[2] = part(2);
        __builtin_amdgcn_sched_barrier(0);
        glds16_buf(rsw, wvoff0[0], 0u, smem + G256Q_SLOT_B0 + dma_off);
      } else if (idx == 3) {
        vk = *(const f32x4*)(ep.v + uvo + 4);
        uv_ = ep.u[nuvh + 2 * D + wn * 16 + l15];
        rsr.p[3] = part(3);
        rsr.p[4] = part(4);
        __builtin_amdgcn_sched_barrier(0);
        glds16_buf(rsw, wvoff0[1], 0u, smem + G256Q_SLOT_B0 + 8192 + dma_off);
      } else if (idx == 4) {
        vv_ = ep.v[nuvh + 2 * D + wn * 16 + l15];
        if (rows) rsr.c = ep.st.cen_in[nrow_m];
        __builtin_amdgcn_sched_barrier(0);
        glds16_buf(rsw, wvoff1, 0u, smem + G256Q_SLOT_B1 + dma_off);
      } else if (idx == 5) glds16_buf(rsa, avoff[1][0], 0u, smem + G256Q_SLOT_A1 + dma_off);
      else if (idx == 6) glds16_buf(rsa, avoff[1][1], 0u, smem + G256Q_SLOT_A1 + 8192 + dma_off);
      __builtin_amdgcn_sched_barrier(0);
    };
    stamp(4);
    f32x16 Oa[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) Oa[0][e] = 0.f, Oa[1][e] = 0.f;
    float mrun = -3.0e38f, lrun = 0.f;
    f32x16 zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
    auto qk = [&](f32x16& S, int kb) {
      const char* kp = Ks + (kb * 32 + l31) * 128;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8_t kf = *(const half8_t*)(kp + (((ks * 2 + hsel) ^ akey) << 4));
        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? zero16 : S, 0, 0, 0);
      }
    };
    auto softmax_pv = [&](f32x16& S, int kb) {
      half8_t P[2];
      att_softmax_block<2>(S, kb == 0, mrun, lrun, Oa, scale_log2e, P);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int dd = db * 32 + l31;
          const int c0 = kb * 4 + 2 * s + hsel;
          const half8_t vf = *(const half8_t*)(Vs + dd * 512 + ((c0 ^ (dd & 15)) << 4));
          Oa[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, P[s], Oa[db], 0, 0, 0);
        }
    };
    f32x16 Sa, Sb;
#ifdef LFM_MEASURE
    if (!(dbg & 33554432))
#endif
    {
      qk(Sa, 0);
#pragma unroll
      for (int kb = 0; kb < 8; kb += 2) {
        qk(Sb, kb + 1);
        next_req(kb);
        softmax_pv(Sa, kb);
        if (kb + 2 < 8) qk(Sa, kb + 2);
        next_req(kb + 1);
        softmax_pv(Sb, kb + 1);
      }
    }
#ifdef LFM_MEASURE
    else {
#pragma unroll
      for (int i = 0; i < 7; ++i) next_req(i);
    }
#endif
    stamp(5);
    // ---- normalise and store: lane owns query l31, d = db * 32 + 8 g + 4 hsel + r.  One pass through the wave's 4 KiB of staging (32 rows x 128 B; 8-byte position
    // p of row r at p ^ ((r & 7) << 1)), read back as 16-byte chunks: a store instruction covers eight whole 128-byte output rows.  The staging lies above the
    // ring, over the (a, b) rows of THIS item, dead since the hand-over.
    {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float inv = 1.0f / (lrun + xhalf(lrun));
      const unsigned ow = (unsigned)(l31 * 128), okey2 = (unsigned)((l31 & 7) << 1);
      const int orow = lna >> 3, och = lna & 7;
      const unsigned ord0 = (unsigned)(orow * 128 + ((och ^ orow) << 4));
      half_t* const obase = ep.O + ((long)m0 + wave * 32 + orow) * D + head * 64 + och * 8;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          half4_t hv = {(half_t)(Oa[db][4 * gg] * inv), (half_t)(Oa[db][4 * gg + 1] * inv), (half_t)(Oa[db][4 * gg + 2] * inv), (half_t)(Oa[db][4 * gg + 3] * inv)};
          *(half4_t*)(ob + ow + ((((unsigned)(db * 8 + 2 * gg + hsel)) ^ okey2) << 3)) = hv;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private rows: no barrier (also the compiler barrier between the half4 writes and the f32x4 reads)
      f32x4 vrow[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) vrow[h] = *(const f32x4*)(ob + h * 1024 + ord0);
#pragma unroll
      for (int h = 0; h < 4; ++h) *(f32x4*)(obase + (long)(h * 8) * D) = vrow[h];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    stamp(6);
    ++tr_k;
    if (!has_next) break;
    // every wave is done with K / V^T and with its output staging: the next item's (a, b) rows (its statistics have long landed), and the second ring half
    // takes its A0(1) B0(1)
    G256_BARRIER();
    rowstat_finish(rsr, ntile_m * G256_BM, nhead);
    issue_a(0, 1, smem + G256Q_BUF_BYTES + G256Q_SLOT_A0);
    issue_b0(1, smem + G256Q_BUF_BYTES + G256Q_SLOT_B0);
    item = nitem;
    tile_m = ntile_m;
    head = nhead;
    m0 = ntile_m * G256_BM;
    first = false;
  }
#undef QKVA_VMCNT
#undef QKVA_LGKM
}

// A: [M, lda] fp16 (the centred, (1 + scale)-weighted rows A' of the folded path), W: [3 D, ldw] fp16 (rows q | k | v, head-major), O: [M, D].
// M = images x 256 tokens, D = heads x 64, K % 64 == 0.
static inline int launch_qkv_attention(const half_t* A, long lda, const half_t* W, long ldw, int M, int D, int heads, int K, const QkvAttnArgs& ep,
                                       hipStream_t stream) {
  if (M <= 0 || (M % G256_BM) != 0 || heads <= 0 || D != heads * 64 || K < 2 * G256Q_BK || (K % G256Q_BK) != 0) return LFM_ERR_SHAPE;
  if ((long)M * lda >= (1L << 31) || (long)3 * D * ldw >= (1L << 31)) return LFM_ERR_SHAPE;  // unsigned 32-bit byte offsets of the buffer-addressed LDS-DMAs
  if ((lda % 8) != 0 || (ldw % 8) != 0 || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)ep.O & 15) || ((uintptr_t)ep.u & 15) || ((uintptr_t)ep.v & 15) ||
      (ep.uv_stride % 4) != 0)
    return LFM_ERR_ALIGN;
  if (ep.st.tiles_p > G256H_MAX_PARTS) return LFM_ERR_SHAPE;
  int devid = 0;
  (void)hipGetDevice(&devid);
  static lfm_device_mask attr_set{0};
  static std::atomic<int> cus[64];
  const unsigned long long dbit = 1ull << (devid & 63);
  if (lfm_device_todo(attr_set, dbit)) {
    if (hipFuncSetAttribute((const void*)qkv_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QKVA_LDS_BYTES) != hipSuccess) return LFM_ERR_LAUNCH;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess || n <= 0) n = 256;
    cus[devid & 63].store(n, std::memory_order_relaxed);
    lfm_device_done(attr_set, dbit);
  }
  const int items = (M / G256_BM) * heads, ncu = cus[devid & 63].load(std::memory_order_relaxed);
  const int dbg = lfm_gemm_debug_flags();
  const int grid = (items <= ncu || (dbg & 4194304)) ? items : ncu;  // one persistent workgroup per CU (flag 4194304: one workgroup per item, A/B)
  hipLaunchKernelGGL(qkv_attention_kernel, dim3(grid), dim3(512), QKVA_LDS_BYTES, stream, A, lda, W, ldw, M, K, ep, items, dbg);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
