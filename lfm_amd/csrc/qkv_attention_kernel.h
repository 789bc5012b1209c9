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
// tile: wave (g, wn) owns rows g*128.. x the 48 columns {Q dims 16 wn.., K dims 16 wn.., V dims 16 wn..} of the head (the W rows are gathered by the DMA source
// addresses), i.e. accumulator tiles j = 0 (Q), 1 (K), 2 (V).  Piece B0 = the Q and K rows (128 rows, 16 KiB, as before), piece B1 = the V rows (64 rows,
// 8 KiB: ONE LDS-DMA per thread), so the phases are A0xB0 (16 MFMAs) | A0xB1 (8) | A1xB1 (8) | A1xB0 (16) and the counted vmcnt waits 5 5 5 6 instead of
// 6 6 6 6.  The V tile is issued with the operands swapped (a lane then owns FOUR CONSECUTIVE TOKENS of one head dim = one 8-byte write into a V^T row).
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

#define QKVA_LDS_BYTES (G256Q_LDS_BYTES + 2048)

__global__ __launch_bounds__(512) void qkv_attention_kernel(const half_t* __restrict__ A, long lda, const half_t* __restrict__ W, long ldw, int M, int K,
                                                            QkvAttnArgs ep, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wn = wave & 3;
  const int D = ep.D;

  int tile_m, head;
  g256_tile_order(blockIdx.x, gridDim.x, ep.heads, dbg, tile_m, head);
  const int m0 = tile_m * G256_BM;

  // ---- DMA sources (piece = rows x 128 B; thread tid stages 16-byte chunk tid of an 8-KiB issue; chunk c of LDS row r holds logical chunk c ^ ((r >> 1) & 7))
  const int cswz = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, -1, 0x00020000);
  unsigned avoff[2][2], wvoff0[2], wvoff1;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < 2; ++p) avoff[s][p] = ((unsigned)(m0 + p * 128 + s * 64 + (tid >> 3)) * (unsigned)lda + (unsigned)cswz) * 2u;
#pragma unroll
  for (int p = 0; p < 2; ++p) {  // B0: LDS row wn' * 32 + c, c < 16: Q dim 16 wn' + c, else K dim 16 wn' + c - 16
    const int wnp = p * 2 + (tid >> 8), c = (tid >> 3) & 31;
    const int n = (c >> 4) * D + head * 64 + wnp * 16 + (c & 15);
    wvoff0[p] = ((unsigned)n * (unsigned)ldw + (unsigned)cswz) * 2u;
  }
  wvoff1 = ((unsigned)(2 * D + head * 64 + (tid >> 3)) * (unsigned)ldw + (unsigned)cswz) * 2u;  // B1: LDS row = V dim
  const int nk = K / G256Q_BK;
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
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, l4 = lane >> 4;
  const int rkey = (l15 >> 1) & 7;
  int a_addr[2], w_addr[2], w1_addr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_addr[ks] = (g * 64 + l15) * 128 + (((ks * 4 + l4) ^ rkey) << 4);   // + i4 * 2048
    w_addr[ks] = (wn * 32 + l15) * 128 + (((ks * 4 + l4) ^ rkey) << 4);  // + j * 2048 (j = 0: Q, 1: K)
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
    if constexpr (PH == 0) {
      if (s1) issue_b1(t + 1, oth + G256Q_SLOT_B1);
    } else if constexpr (PH == 1) {
      if (s1) issue_a(1, t + 1, oth + G256Q_SLOT_A1);
    } else if constexpr (PH == 2) {
      if (s2) issue_a(0, t + 2, cur + G256Q_SLOT_A0);
    } else {
      if (s2) issue_b0(t + 2, cur + G256Q_SLOT_B0);
    }
    // DMAs of the three newest pieces may still fly (A0 / B0 / A1: two per wave, B1: one)
    if (s2) {
      if constexpr (PH < 3) QKVA_VMCNT(5);
      else QKVA_VMCNT(6);
    } else if (s1) {
      if constexpr (PH < 2) QKVA_VMCNT(5);
      else if constexpr (PH == 2) QKVA_VMCNT(3);
      else QKVA_VMCNT(2);
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
        if constexpr (VT) {  // operands swapped: acc[i][2][r] = C[m = 16 i + 4 l4 + r][V dim l15]
          acc[I0 + i4][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i4][ks], wf[2][ks], acc[I0 + i4][2], 0, 0, 0);
        } else {  // acc[i][j][r] = C[m = 16 i + l15][dim 4 l4 + r]
#pragma unroll
          for (int j2 = 0; j2 < 2; ++j2) acc[I0 + i4][j2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j2][ks], af[i4][ks], acc[I0 + i4][j2], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: row statistics and column constants requested BEFORE the first DMAs (vmcnt retires in order), pieces A0(0) B0(0) B1(0) A1(0) [A0(1) B0(1)]
  G256hRowStatRegs rsr = g256h_rowstat_load(ep, m0, M);
  const long uvo = (long)tile_m * ep.uv_stride + head * 64 + wn * 16;
  const f32x4 uq = *(const f32x4*)(ep.u + uvo + 4 * l4), vq = *(const f32x4*)(ep.v + uvo + 4 * l4);
  const f32x4 uk = *(const f32x4*)(ep.u + uvo + D + 4 * l4), vk = *(const f32x4*)(ep.v + uvo + D + 4 * l4);
  const float uv_ = ep.u[uvo + 2 * D + l15], vv_ = ep.v[uvo + 2 * D + l15];
  __builtin_amdgcn_sched_barrier(0);
  issue_a(0, 0, smem + G256Q_SLOT_A0);
  issue_b0(0, smem + G256Q_SLOT_B0);
  issue_b1(0, smem + G256Q_SLOT_B1);
  issue_a(1, 0, smem + G256Q_SLOT_A1);
  if (nk > 1) {
    issue_a(0, 1, smem + G256Q_BUF_BYTES + G256Q_SLOT_A0);
    issue_b0(1, smem + G256Q_BUF_BYTES + G256Q_SLOT_B0);
  }
  g256h_rowstat_finish(ep, rsr, smem, m0, M, head);
  if (nk > 1) QKVA_VMCNT(6);
  else QKVA_VMCNT(2);
  G256_BARRIER();
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
  if (g == 0) run(g256q_ic<0>{});
  else run(g256q_ic<1>{});
#undef QKVA_VMCNT
#undef QKVA_LGKM

  // ---- hand-over: the ring is dead (every wave is past the last barrier, its fragment reads retired before its MFMAs).  Q [256][64], K [256][64]: 128-byte rows,
  // chunk c of row r at c ^ ((r >> 1) & 7); V^T [64][256]: 512-byte rows, tokens of a 16-group in the vt_pos order, chunk c of row d at c ^ (d & 15) -- the images
  // dit_attention_kernel<256, 1, 64> stages from HBM.
  char* const Qs = smem;
  char* const Ks = smem + 32768;
  char* const Vs = smem + 65536;
  {
    const float* rs = ep.rs;
    const int d = wn * 16 + l15;  // V dim of this lane
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = g * 128 + i * 16 + l15;
      const f32x2 ab = *(const f32x2*)(rs + 2 * m);
      const f32x4 q = row_affine4(ab.x, ab.y, acc[i][0], uq, vq);
      const f32x4 k = row_affine4(ab.x, ab.y, acc[i][1], uk, vk);
      const int off = m * 128 + (((2 * wn + (l4 >> 1)) ^ ((m >> 1) & 7)) << 4) + (l4 & 1) * 8;
      *(half4_t*)(Qs + off) = (half4_t){(half_t)q.x, (half_t)q.y, (half_t)q.z, (half_t)q.w};
      *(half4_t*)(Ks + off) = (half4_t){(half_t)k.x, (half_t)k.y, (half_t)k.z, (half_t)k.w};
      const int t0 = g * 128 + i * 16 + 4 * l4;  // tokens t0 .. t0 + 3 of V dim d
      const f32x4 r0 = *(const f32x4*)(rs + 2 * t0), r1 = *(const f32x4*)(rs + 2 * t0 + 4);
      const half4_t h = {(half_t)fma_v(r0.x, acc[i][2][0], fma_v(r0.y, uv_, vv_)), (half_t)fma_v(r0.z, acc[i][2][1], fma_v(r0.w, uv_, vv_)),
                         (half_t)fma_v(r1.x, acc[i][2][2], fma_v(r1.y, uv_, vv_)), (half_t)fma_v(r1.z, acc[i][2][3], fma_v(r1.w, uv_, vv_))};
      // tokens 4 l4 .. + 3 of the 16-group sit at positions {0, 8, 4, 12}[l4] .. + 3: 16-byte chunk 2 group + (l4 & 1), upper half for l4 >= 2
      *(half4_t*)(Vs + d * 512 + (((2 * (8 * g + i) + (l4 & 1)) ^ (d & 15)) << 4) + (l4 >> 1) * 8) = h;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  G256_BARRIER();

  // ---- attention: wave w owns queries 32 w .. 32 w + 31 (dit_attention_kernel<256, 1, 64>: S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x16_f16)
  const int hsel = lane >> 5, l31 = lane & 31;
  const int akey = (l31 >> 1) & 7;
  half8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const half8_t*)(Qs + (wave * 32 + l31) * 128 + (((ks * 2 + hsel) ^ akey) << 4));
  f32x16 zero16;
#pragma unroll
  for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
  f32x16 Oa[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) Oa[0][e] = 0.f, Oa[1][e] = 0.f;
  float mrun = -3.0e38f, lrun = 0.f;
  const float scale_log2e = ep.scale_log2e;
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
  qk(Sa, 0);
#pragma unroll
  for (int kb = 0; kb < 8; kb += 2) {
    qk(Sb, kb + 1);
    softmax_pv(Sa, kb);
    if (kb + 2 < 8) qk(Sa, kb + 2);
    softmax_pv(Sb, kb + 1);
  }
  // ---- normalise and store: lane owns query l31, d = db * 32 + 8 g + 4 hsel + r.  Four passes of 8 rows through 1 KiB of the wave's own (dead) Q rows:
  // 8-byte position p of staging row r at p ^ (r << 1), read back as 16-byte chunks -- a store instruction covers eight whole 128-byte output rows.
  {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float inv = 1.0f / (lrun + xhalf(lrun));
    char* const ob = Qs + wave * 4096;
    const unsigned ow = (unsigned)((l31 & 7) * 128), okey2 = (unsigned)((l31 & 7) << 1);
    const int orow = lane >> 3, och = lane & 7;
    const unsigned ord0 = (unsigned)(orow * 128 + ((och ^ orow) << 4));
    half_t* const obase = ep.O + ((long)m0 + wave * 32 + orow) * D + head * 64 + och * 8;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if ((l31 >> 3) == h) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) {
            half4_t hv = {(half_t)(Oa[db][4 * gg] * inv), (half_t)(Oa[db][4 * gg + 1] * inv), (half_t)(Oa[db][4 * gg + 2] * inv), (half_t)(Oa[db][4 * gg + 3] * inv)};
            *(half4_t*)(ob + ow + ((((unsigned)(db * 8 + 2 * gg + hsel)) ^ okey2) << 3)) = hv;
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private rows: no barrier (also the compiler barrier between the half4 writes and the f32x4 read)
      const f32x4 vrow = *(const f32x4*)(ob + ord0);
      *(f32x4*)(obase + (long)(h * 8) * D) = vrow;
      if (h < 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows of pass h are in registers before pass h + 1 overwrites them
    }
  }
}

// A: [M, lda] fp16 (the centred, (1 + scale)-weighted rows A' of the folded path), W: [3 D, ldw] fp16 (rows q | k | v, head-major), O: [M, D].
// M = images x 256 tokens, D = heads x 64, K % 64 == 0.
static inline int launch_qkv_attention(const half_t* A, long lda, const half_t* W, long ldw, int M, int D, int heads, int K, const QkvAttnArgs& ep,
                                       hipStream_t stream) {
  if (M <= 0 || (M % G256_BM) != 0 || heads <= 0 || D != heads * 64 || K <= 0 || (K % G256Q_BK) != 0) return LFM_ERR_SHAPE;
  if ((long)M * lda >= (1L << 30) || (long)3 * D * ldw >= (1L << 30)) return LFM_ERR_SHAPE;  // 32-bit byte offsets of the buffer-addressed LDS-DMAs
  if ((lda % 8) != 0 || (ldw % 8) != 0 || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)ep.O & 15) || ((uintptr_t)ep.u & 15) || ((uintptr_t)ep.v & 15) ||
      (ep.uv_stride % 4) != 0)
    return LFM_ERR_ALIGN;
  if (ep.st.tiles_p > G256H_MAX_PARTS) return LFM_ERR_SHAPE;
  static lfm_device_mask attr_set{0};
  const unsigned long long dbit = lfm_device_bit();
  if (lfm_device_todo(attr_set, dbit)) {
    if (hipFuncSetAttribute((const void*)qkv_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QKVA_LDS_BYTES) != hipSuccess) return LFM_ERR_LAUNCH;
    lfm_device_done(attr_set, dbit);
  }
  hipLaunchKernelGGL(qkv_attention_kernel, dim3((M / G256_BM) * heads), dim3(512), QKVA_LDS_BYTES, stream, A, lda, W, ldw, M, K, ep, lfm_gemm_debug_flags());
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
