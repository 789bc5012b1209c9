// Streamed, persistent attention core for the benchmarked DiT shape: 256 tokens x head_dim 64 (timm Attention as called at
// /root/reference/models/DiT.py:120): O = softmax(Q K^T * hd^-0.5) V per (image, head) item.  Round 6.
//
// Why a second kernel (the per-item one stays for every other shape, attention_kernel.h): with one workgroup per item, 1024 items on 256 CUs x 2 resident
// workgroups are exactly TWO rounds, and a round is load-everything (96 KiB per workgroup, ~11k cycles with all 512 workgroups bursting at once), then
// compute (~15k), then drain the stores (~5k) -- the memory system idles while the matrix pipe works and the other way round (profiles/r03_attention_wg_timeline.txt,
// r04_attention_narrow_timeline.txt: memory-only 22.6 us, compute-only 28.5 us, whole kernel 33.4 us).  Here a workgroup is PERSISTENT (grid = two per CU, items
// strided over the grid) and K / V^T never stop streaming: the LDS holds a four-slot ring of 64-key STAGES (K rows 64 x 128 B + V^T rows 64 dims x 128 B = 16 KiB per
// slot, whole 128-byte lines on both operands), slot s always carries stage s of whatever item is current, and the LDS-DMA of a slot's next tenant is issued at
// the first barrier after its last reader: K runs three stages ahead of the MFMAs that consume it, V^T two -- across item boundaries, so only the first stage of a
// workgroup's first item is ever waited for with nothing else to do.
//
// Same arithmetic as dit_attention_kernel<256, 1, 64>: eight waves x 32 queries, S^T = K Q^T on v_mfma_f32_32x32x16_f16 (a lane owns one query: row max / row sum
// are in-lane plus one lane^32 exchange), online softmax over 32-key blocks (att_softmax_block below, shared by both kernels), P rounded to fp16 straight into the
// B operand of O^T = V^T P^T, V^T rows in the vt_pos token order -- the results are BIT-IDENTICAL to that kernel's (tests/test_gpu_dit.py::
// test_attention_stream_matches_per_item).
//   * every LDS address of the key loop is one of four per-lane registers + an immediate (slot, key block and k-slot are compile-time: the item body is fully unrolled);
//     K and V^T stages share the swizzle (16-byte chunk c of row r stored at c ^ ((r >> 1) & 7): conflict-free for the 16-lane groups of ds_read_b128), so the SAME
//     four registers address both;
//   * Q fragments and O rows go through buffer instructions: per item only two SGPR offsets change.
// VMEM bookkeeping (vmcnt counts loads, LDS-DMAs and stores alike and retires in issue order).  Per wave and item: 8 LDS-DMAs (one per stage and operand), 4 Q loads,
// 4 O stores.  Issue order, g = 4 item + stage: prologue K0 Q Q Q Q V0 K1 V1 K2 (four operations behind the
// Q loads, as in the steady state: the compiler's own wait for the fragments is vmcnt(4) on both paths); after barrier b: K(b+3), V(b+2); behind barrier 4i+3 also Q(i+1) x 4 (once the last
// S MFMAs of item i have consumed the old fragments) and O(i) x 4.  Barrier b needs K(b) and V(b) (and Q for b = 4i): the counted waits are 4 / 10 / 2 / 2 for stage
// 0 / 1 / 2 / 3 (first item: 3 / 3 / 2 / 2).  Stages past the workgroup's last item are issued with an out-of-range buffer offset (no memory traffic, zeros, but they
// COUNT), so the waits are the same to the end.
// LDS: 4 x 16 KiB ring + 8 x 1 KiB output staging (a wave's 8 rows x 128 B per pass, four passes: a store instruction covers eight whole 128-byte rows) = 72 KiB: two
// workgroups per CU, four waves per SIMD, <= 128 VGPRs.  (80 KiB -- 2 KiB of staging per wave, two passes -- measured the same time; the occupancy query answers two workgroups per CU for both sizes and the
// per-workgroup trace shows both resident from the start: profiles/r06_attention_stream.txt.)
#pragma once
#include <stdlib.h>
#include "gemm_kernel.h"

#define ATS_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define ATS_BARRIER()                              \
  do {                                             \
    __builtin_amdgcn_s_barrier();                  \
    asm volatile("" ::: "memory");                 \
  } while (0)

// One 32-key block of the online softmax for the query a lane owns (S: its 16 scores of the block, the other 16 live in lane ^ 32), shared by the streamed and the
// per-item kernels (same arithmetic in the same order => bit-identical results whichever kernel evaluates an item).  Round 6: the key loop is bound by instruction
// ISSUE, not by a pipe -- per SIMD, whether it holds two or four waves, one 32-key block of one wave goes through in ~850 cycles while its 8 MFMAs occupy the matrix
// pipe for 256 and ~5 single-issue instructions hide under each of them (profiles/r06_attention_stream.txt) -- so the block is written for instruction count:
//   * OPTIMISTIC exponentials: p = 2^((s - mrun) scale) is taken against the running reference mrun WITHOUT first looking for the block's maximum; the lane's own sum of
//     its 16 p (needed anyway) tells whether that was safe -- every p <= sum <= 2^14 stays far inside fp16 (P is the fp16 operand of the P V MFMA; row sums and O are
//     fp32).  Only when some lane's sum exceeds 2^14 (or is not a number), and for the first block of an item, the block takes the FULL path: row maximum (3-input
//     maxima + one lane ^ 32 exchange), mrun <- max, O and l rescaled by 2^((old - new) scale), exponentials again.  The reference follows the maximum lazily, as
//     before (rounds 3-5 moved it when the maximum had grown by more than 2^8); fp16 rounds P relative to its size, so the result does not depend on where in
//     [2^-14 .. 2^14] the block's largest p lands.
//   * row sum as a TREE of packed adds (8 issue slots; the serial chain of rounds 1-5 drew a wait state per link: 18).
// `first` is wave-uniform.  Returns the packed P of the block (k-slots 0 and 1) in P.
template <int NDB>
__device__ __forceinline__ void att_softmax_block(const f32x16& S, bool first, float& mrun, float& lrun, f32x16 (&Oa)[NDB], float scale_log2e, half8_t (&P)[2]) {
  f32x2 p[8];
  const f32x2 sc2 = {scale_log2e, scale_log2e};
  auto expo = [&]() {
    const float mbs = mrun * scale_log2e;
    const f32x2 nmb2 = {-mbs, -mbs};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const f32x2 s2 = {S[2 * e], S[2 * e + 1]};
      const f32x2 a2 = __builtin_elementwise_fma(s2, sc2, nmb2);  // ONE fused multiply-add on every path and in every kernel that inlines this (v_pk_fma_f32 / v_fma_f32)
      p[e] = (f32x2){__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
    }
  };
  auto lane_sum = [&]() {
    const f32x2 t = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));  // v_pk_add_f32 x 7
    return t.x + t.y;
  };
  float ls = 0.f;
  bool full = first;
  if (!first) {
    expo();
    ls = lane_sum();
    full = !__all(ls <= 16384.0f);
  }
  if (full) {  // wave-uniform
    float mx = fmaxf(fmaxf(S[0], S[1]), S[2]);
#pragma unroll
    for (int e = 3; e < 15; e += 2) mx = fmaxf(fmaxf(mx, S[e]), S[e + 1]);
    mx = fmaxf(mx, S[15]);
    mx = fmaxf(mx, xhalf(mx));
    const float mnew = fmaxf(mrun, mx);
    const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * scale_log2e);
    mrun = mnew;
    lrun *= alpha;
#pragma unroll
    for (int db = 0; db < NDB; ++db) Oa[db] *= alpha;
    expo();
    ls = lane_sum();
  }
  lrun += ls;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    P[e >> 2][(e & 3) * 2] = (half_t)p[e].x;
    P[e >> 2][(e & 3) * 2 + 1] = (half_t)p[e].y;
  }
}

#ifndef ATT_TRACE_SLOTS
#define ATT_TRACE_SLOTS 64
#define ATT_WG_TRACE 2048
static __device__ unsigned long long att_trace[ATT_TRACE_SLOTS];
static __device__ unsigned long long att_wg_trace[ATT_WG_TRACE][4];
#endif
// MODE (measurement builds only, flags 33554432 / 67108864 like the per-item kernel): 1 = memory only (LDS-DMA ring, waits, barriers, Q loads, O stores; no MFMA / softmax),
// 2 = compute only (every LDS-DMA and Q load out of range: no operand traffic), 3 = s_memtime trace (attention_kernel.h: att_trace = wave 0 of the first / last
// workgroup: slot 0 start, 1 prologue issued, 2 + 14 i + 3 j + {0 before the wait, 1 after the wait, 2 after the barrier} for stage j of the workgroup's item i < 2,
// 14 + 14 i key loop done, 15 + 14 i stores issued; att_wg_trace = {HW_ID | XCC_ID << 32, start, first barrier passed, end} per workgroup)
template <int MODE = 0>
__global__ __launch_bounds__(512, 4) void dit_attention_stream_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ Vt,
                                                                      half_t* __restrict__ O, int D, int heads, int items, int gq, int gr, unsigned kbytes,
                                                                      float scale_log2e) {
  constexpr int T = 256, HD = 64;
  constexpr int SLOT = 16384, VOFF = 8192, OSTG = 4 * SLOT;  // ring slot (K stage | V^T stage), V^T inside a slot, output staging
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hsel = lane >> 5, l31 = lane & 31;
  const int G = gridDim.x;
  [[maybe_unused]] const bool tr_first = blockIdx.x == 0, tr_last = blockIdx.x == gridDim.x - 1;
  [[maybe_unused]] int tr_item = 0;
  auto stamp = [&](int slot) {
    if constexpr (MODE == 3) {
      if (wave == 0 && (tr_first || tr_last) && slot < 32) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (lane == 0) att_trace[(tr_first ? 0 : 32) + slot] = t;
      }
    }
  };
  auto wg_stamp = [&](int slot) {
    if constexpr (MODE == 3) {
      if (wave == 0 && blockIdx.x < ATT_WG_TRACE) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (slot == 0) {
          const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
          if (lane == 0) att_wg_trace[blockIdx.x][0] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        }
        if (lane == 0) att_wg_trace[blockIdx.x][slot + 1] = t;
      }
    }
  };
  stamp(0);
  wg_stamp(0);
  int item = blockIdx.x;
  int img = item / heads, head = item - img * heads;

  // resources over the whole tensors (num_records = their size: an offset >= 2^31 is out of range -> the tail stages fetch nothing)
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)Vt, 0, (int)kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)Q, 0, (int)kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)O, 0, (int)kbytes, 0x00020000);
  // LDS-DMA: lane i of the workgroup fills 16-byte position (i & 7) of stage row (i >> 3) with the row's chunk (i & 7) ^ key
  const int srow = tid >> 3, sch = (tid & 7) ^ ((srow >> 1) & 7);
  const unsigned koff = (unsigned)(srow * D + sch * 8) * 2u;  // K rows: tokens, D halves apart
  const unsigned voff = (unsigned)(srow * T + sch * 8) * 2u;  // V^T rows: head dims, T halves apart
  constexpr unsigned POISON = 0x80000000u;
  char* const dma_dst = smem + wave * 1024;  // + slot * SLOT (+ VOFF): lane-linear 1 KiB per wave
  // fragment reads: row (lane & 31) of a 32-row block, chunk ((2 j + hsel) ^ key) = one of four per-lane offsets
  unsigned fa[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) fa[j] = (unsigned)(l31 * 128 + ((((j * 2 + hsel) ^ ((l31 >> 1) & 7))) << 4));
  const unsigned qoff = (unsigned)((wave * 32 + l31) * D + hsel * 8) * 2u;
  // output staging, per wave: pass h holds rows 8 h .. 8 h + 7 of the wave's 32 queries as 128-byte rows, 8-byte position p of row r at p ^ (r << 1)
  char* const ob = smem + OSTG + wave * 1024;
  const unsigned ow = (unsigned)((l31 & 7) * 128);  // + ((c ^ key2) << 3)
  const unsigned okey2 = (unsigned)((l31 & 7) << 1);
  const int orow = lane >> 3, och = lane & 7;
  const unsigned ord0 = (unsigned)(orow * 128 + ((och ^ orow) << 4));
  const unsigned ooff = (unsigned)((wave * 32 + orow) * D + och * 8) * 2u;

  auto item_k = [&](int im, int hd) { return (unsigned)((im * T) * D + hd * HD) * 2u; };   // byte offset of the item's K / Q / O block (row 0, head column 0)
  auto item_v = [&](int im, int hd) { return (unsigned)((im * heads + hd) * HD * T) * 2u; };
  auto dma_k = [&](unsigned base, bool live, int stage) {
    glds16_buf(rs_k, live && MODE < 2 ? koff : POISON, base + (unsigned)(stage * 64 * D) * 2u, dma_dst + stage * SLOT);
  };
  auto dma_v = [&](unsigned base, bool live, int stage) {
    glds16_buf(rs_v, live && MODE < 2 ? voff : POISON, base + (unsigned)(stage * 64) * 2u, dma_dst + stage * SLOT + VOFF);
  };
  half8_t qf[4];
  auto load_q = [&](unsigned base) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_q, (int)((MODE < 2 ? qoff : POISON) + ks * 32), (int)base, 0));
      qf[ks] = __builtin_bit_cast(half8_t, r);
    }
  };

  unsigned kb_cur = item_k(img, head), vb_cur = item_v(img, head);
  // ---- prologue: the first three K stages and two V^T stages of the first item, its Q fragments
  dma_k(kb_cur, true, 0);
  load_q(kb_cur);
  dma_v(vb_cur, true, 0);
  dma_k(kb_cur, true, 1);
  dma_v(vb_cur, true, 1);
  dma_k(kb_cur, true, 2);
  stamp(1);

  f32x16 zero16;
#pragma unroll
  for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
  f32x16 Oa[2];
  float mrun, lrun;

  // S^T block: 32 keys (block kbl of the stage in `slot`) x the wave's 32 queries
  auto qk = [&](f32x16& S, int slot, int kbl) {
    if constexpr (MODE == 1) return;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const half8_t kf = *(const half8_t*)(smem + slot * SLOT + kbl * 4096 + fa[ks]);
      S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? zero16 : S, 0, 0, 0);
    }
  };
  // online softmax of one 32-key block for the query this lane owns, then O^T += V^T P^T
  auto softmax_pv = [&](f32x16& S, int slot, int kbl, bool first_block) {
    if constexpr (MODE == 1) return;
    half8_t P[2];
    att_softmax_block<2>(S, first_block, mrun, lrun, Oa, scale_log2e, P);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        // keys {4 h + r} and {8 + 4 h + r} of k-slot s are ONE 16-byte chunk of the permuted V^T row (gemm_kernel.h: vt_pos): chunk 4 kbl + 2 s + h of the stage
        const half8_t vf = *(const half8_t*)(smem + slot * SLOT + VOFF + db * 4096 + fa[kbl * 2 + s]);
        Oa[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, P[s], Oa[db], 0, 0, 0);
      }
  };

  bool first = true;
#pragma unroll 1
  for (;;) {
    // the next item of this workgroup (strided by the grid); past the end its stages are issued out of range
    int nimg = img + gq, nhead = head + gr;
    if (nhead >= heads) {
      nhead -= heads;
      ++nimg;
    }
    const bool has_next = item + G < items;
    const unsigned kb_next = has_next ? item_k(nimg, nhead) : 0u, vb_next = has_next ? item_v(nimg, nhead) : 0u;

#pragma unroll
    for (int e = 0; e < 16; ++e) Oa[0][e] = 0.f, Oa[1][e] = 0.f;
    mrun = -3.0e38f;
    lrun = 0.f;

    // ---- stage 0 landed (and this item's Q): everything older than the previous item's four output stores
    if (tr_item < 2) stamp(2 + 14 * tr_item);
    if (first) ATS_VMCNT(3);
    else ATS_VMCNT(4);
    if (tr_item < 2) stamp(3 + 14 * tr_item);
    ATS_BARRIER();
    if (tr_item < 2) stamp(4 + 14 * tr_item);
    if (first) wg_stamp(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks]));  // the compiler's own wait for the Q loads goes HERE, before the next LDS-DMAs are issued
    dma_k(kb_cur, true, 3);
    dma_v(vb_cur, true, 2);
    // ---- stage j + 1 landed; every wave is past its reads of K stage j and V^T stage j - 1: counted wait, barrier, the next two LDS-DMAs
    auto next_stage = [&](int j) {
      if (tr_item < 2) stamp(2 + 14 * tr_item + 3 * (j + 1));
      if (j == 0) {
        if (first) ATS_VMCNT(3);
        else ATS_VMCNT(10);
      } else {
        ATS_VMCNT(2);
      }
      if (tr_item < 2) stamp(3 + 14 * tr_item + 3 * (j + 1));
      ATS_BARRIER();
      if (tr_item < 2) stamp(4 + 14 * tr_item + 3 * (j + 1));
      if (j == 0) {
        dma_k(kb_next, has_next, 0);
        dma_v(vb_cur, true, 3);
      } else if (j == 1) {
        dma_k(kb_next, has_next, 1);
        dma_v(vb_next, has_next, 0);
      } else {
        dma_k(kb_next, has_next, 2);
        dma_v(vb_next, has_next, 1);
      }
    };
    // The eight key blocks of the item, MFMA-first: the S MFMAs of block b + 1 are issued before the softmax arithmetic of block b, so the matrix pipe works under the
    // VALU-heavy part.  (Measured and not kept, round 6: waves 4-7 in VALU-first order so that a SIMD's two waves of this workgroup want the pipe and the issue port
    // at different times -- 33.4 vs 31.8 us, 6144 items 180.7 vs 170.0: the VALU-first half loses its software pipeline, and the second code path costs 8 spilled
    // dwords under the 128-register budget.)
    f32x16 Sa, Sb;
    qk(Sa, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qk(Sb, j, 1);
      if (j == 3 && has_next) load_q(kb_next);  // the old fragments are dead: the next item's arrive under the last two softmax blocks and the output pass
      softmax_pv(Sa, j, 0, j == 0);
      if (j < 3) {
        next_stage(j);
        qk(Sa, j + 1, 0);
      }
      softmax_pv(Sb, j, 1, false);
    }
    // ---- normalise and store: lane owns query (lane & 31), d = db * 32 + 8 g + 4 hsel + r.  Four passes of 8 rows through the wave's staging rows.
    if (tr_item < 2) stamp(14 + 14 * tr_item);
    {
      const float inv = 1.0f / (lrun + xhalf(lrun));
      const unsigned obase = kb_cur;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        if ((l31 >> 3) == h) {
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              half4_t hv = {(half_t)(Oa[db][4 * g] * inv), (half_t)(Oa[db][4 * g + 1] * inv), (half_t)(Oa[db][4 * g + 2] * inv), (half_t)(Oa[db][4 * g + 3] * inv)};
              *(half4_t*)(ob + ow + ((((unsigned)(db * 8 + 2 * g + hsel)) ^ okey2) << 3)) = hv;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private rows: no barrier (the asm is also the COMPILER barrier between the half4 writes and the
                                                            // f32x4 read of the same bytes: without it type-based alias analysis lets the read move up -- measured: wrong rows)
        const f32x4 v = *(const f32x4*)(ob + ord0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rs_o, (int)ooff,
                                               (int)(obase + (unsigned)(h * 8 * D) * 2u), 0);
        if (h < 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows of pass h are in registers before pass h + 1 overwrites them
      }
    }
    if (tr_item < 2) stamp(15 + 14 * tr_item);
    ++tr_item;
    if (!has_next) break;
    first = false;
    item += G;
    img = nimg;
    head = nhead;
    kb_cur = kb_next;
    vb_cur = vb_next;
  }
  if constexpr (MODE == 3) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_stamp(2);
  }
}

// grid = min(items, 2 x CUs): two resident workgroups per CU (72 KiB of LDS, <= 128 VGPRs)
static int attention_stream_launch(const half_t* Q, const half_t* K, const half_t* Vt, half_t* O, int batch, int heads, hipStream_t st, int mode = 0) {
  constexpr int T = 256, HD = 64, LDS = 4 * 16384 + 8 * 1024;
  const long items = (long)batch * heads;
  const long bytes = items * T * HD * 2;  // = batch * T * D * 2: each of Q, K, V^T, O
  if (items <= 0 || bytes >= (1L << 31)) return 1;  // not for this kernel (32-bit buffer offsets, bit 31 = the out-of-range mark)
  int devid = 0;
  (void)hipGetDevice(&devid);
  static lfm_device_mask set{0};
  static std::atomic<int> cus[64];
  const unsigned long long dbit = 1ull << (devid & 63);
  if (lfm_device_todo(set, dbit)) {
    (void)hipFuncSetAttribute((const void*)dit_attention_stream_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
#ifdef LFM_MEASURE
    (void)hipFuncSetAttribute((const void*)dit_attention_stream_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)dit_attention_stream_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)dit_attention_stream_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
#endif
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess || n <= 0) n = 256;
    cus[devid & 63].store(n, std::memory_order_relaxed);
    lfm_device_done(set, dbit);
  }
  long per_cu = 2;
#ifdef LFM_MEASURE
  if (const char* e = getenv("LFM_ATS_WG_PER_CU")) per_cu = atoi(e) == 1 ? 1 : 2;  // measurement: one persistent workgroup per CU (twice the items each)
#endif
  const int G = (int)(items < per_cu * cus[devid & 63].load(std::memory_order_relaxed) ? items : per_cu * cus[devid & 63].load(std::memory_order_relaxed));
  const float sl2 = 0.125f * 1.4426950408889634f;  // hd^-0.5 * log2(e)
#define ATS_LAUNCH(M) hipLaunchKernelGGL(dit_attention_stream_kernel<M>, dim3(G), dim3(512), LDS, st, Q, K, Vt, O, heads * HD, heads, (int)items, G / heads, G % heads, (unsigned)bytes, sl2)
#ifdef LFM_MEASURE
  if (mode == 1) ATS_LAUNCH(1);
  else if (mode == 2) ATS_LAUNCH(2);
  else if (mode == 3) ATS_LAUNCH(3);
  else
#endif
    ATS_LAUNCH(0);
#undef ATS_LAUNCH
  (void)mode;
  return hipGetLastError() == hipSuccess ? 0 : LFM_ERR_LAUNCH;
}
