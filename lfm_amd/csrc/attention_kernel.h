// Attention core of a DiT block (timm Attention as called at /root/reference/models/DiT.py:120): O = softmax(Q K^T * hd^-0.5) V
// per (image, head), head_dim HD = 64 (DiT-S / B / L) or 72 (DiT-XL: 1152 / 16).
//
// One workgroup per (head, image); T / (32 JQ) waves, each owning JQ blocks of 32 queries (processed together so every K / V^T
// fragment read from LDS feeds JQ MFMAs).
// K [T][HD] and V^T [HD][T] of the head are staged once into LDS by LDS-DMA.  K rows are HD * 2 bytes: 128-B rows (hd 64) are
// XOR-swizzled at the DMA source (chunk' = chunk ^ ((row>>1)&7)); 144-B rows (hd 72) need no swizzle -- 36 dwords per row put 16
// consecutive rows on 16 disjoint groups of four banks.  V^T rows (2T bytes) are swizzled by row (chunk' = chunk ^ (row & VKEY)).
// S^T = K Q^T on v_mfma_f32_32x32x16_f16: a lane then holds, for ONE query (lane&31), the scores of
// keys kb*32 + 8g + 4*(lane>>5) + r -- row max / row sum are in-lane plus one lane^32 exchange, and the
// fp32->fp16 packed P registers are directly the B-operand of O^T = V^T P^T (the key order inside an
// MFMA k-slot is the same permutation on both operands, so no shuffle is needed).
// hd 72 = 4.5 k-slots of 16: the fifth slot's upper half (dims 72..79) is fed zeros on BOTH operands (LDS past a row end is
// another row or stale bytes, possibly NaN patterns); O^T has 2.25 blocks of 32 rows: the third block computes 8 live rows.
// V^T rows hold the tokens of every 16-group in the order 0-3, 8-11, 4-7, 12-15 (gemm_kernel.h: vt_pos), which makes a lane's P V operand one 16-byte read.
// Keys are consumed in 32-key blocks with an online softmax (running max m, running sum l), which keeps
// the live state at S 32 + P 16 + O 64 + Q 32 registers for hd 64, JQ 2 (2 waves / SIMD).
#pragma once
#include "gemm_kernel.h"
#include "attention_stream_kernel.h"

int lfm_gemm_debug_flags();
int lfm_attention_stream_enabled();  // LFM_OPT_ATTENTION_STREAM (dit.hip)

// MODE 3 (measurement only): s_memtime stamps of wave 0 of the first workgroup (slots 0..31) and of the last one (32..63), read back with
// lfm_attention_trace_read.  Slots: 0 start, 1 all DMAs / Q loads issued, 2 K and Q landed (first barrier), 3 + 4 k + {0: S(next) issued,
// 1: softmax + PV of the even block done, 2: S(next even) issued, 3: softmax + PV of the odd block done} for the k-th loop iteration,
// 19 stores issued, 20 stores acknowledged; inside the first softmax_pv: 21 softmax VALU done, 22 V^T landed (barrier), 23 PV MFMAs issued.
// (the arrays are declared by attention_stream_kernel.h, included above: both kernels stamp into them)
// MODE 3 also records, per workgroup (linear id < 2048): {HW_ID | XCC_ID << 32, start, loads landed, end} -- which CU it ran on and when

// MODE (measurement only, tools/r2_probe3.py): 0 = the kernel; 1 = memory phases only (stage K / V^T, fetch Q, store a row per query, no
// S / softmax / PV); 2 = compute only (K / V^T are never fetched: the loop runs on whatever the LDS holds).  Round 2, 64 images x 16 heads x 256
// tokens (profiles/r02_probe3_attention_phases_ln_rows.txt): whole kernel 40-41 us, memory phases 22-23 us, compute (with its Q / O traffic) 33-36 us
// -- the kernel is bound by the instruction stream of its two waves per SIMD, and the memory phases already hide behind the co-resident workgroup.
// Round 3 (profiles/r03_attention_*.txt): the per-workgroup timeline (MODE 3: HW_ID + s_memtime per workgroup) shows every CU holding two workgroups
// that de-phase by themselves (the second of a pair lags by ~14k cycles), a workgroup living ~33k cycles of which the loads take ~11k, the key-block
// loop ~15.5k and the store drain ~5k: the kernel is bound by memory LATENCY at an occupancy of two workgroups per CU (64 KiB of LDS each), not by
// its instruction streams.  Measured and not kept: sleeping one workgroup of every first-round pair (37.7-43 us vs 38.5), and a one-basic-block
// step with sched_group_barrier interleaving every MFMA with the VALU that covers its 32 pipe cycles (ISA as prescribed, 40.4 vs 39.5 us) -- the
// partner wave already fills those gaps.  Kept: the lazy rescale (below).
// Measured and NOT kept on that evidence (round 2): a streamed schedule (DMAs issued key block by key block, V^T slices as 64-byte pieces, the online-softmax
// loop started after block 0 behind one counted vmcnt + barrier per block): correct, 40.8-41.7 us vs 40.1-42.8 us.
// (Round 4, tried and dropped: the row maxima as inline-asm v_max3_f32 to skip the few canonicalising v_max_f32 x, x that IEEE fmaxf puts in front of MFMA
// results -- the compiler's hazard recogniser does not see an asm reading MFMA accumulators, and the hd-72 runs stopped being bit-repeatable.)
// NCH > 1 (round 4: DiT at 1024 tokens, models/DiT.py:179-182 with --image_size 512): the sequence has NCH * T tokens.  A workgroup owns T QUERIES
// (blockIdx.z = which block of T) and walks the keys in NCH chunks of T through the same LDS image -- the online softmax state carries over; a chunk
// is re-staged behind a barrier (no cross-chunk prefetch: this shape is off the benchmarked path).
// QS > 1 (round 5, latency mode: one image = `heads` workgroups for 256 CUs): the queries of an (image, head) item are split over QS workgroups (blockIdx.z) that
// each stage ALL keys; T / (32 JQ QS) waves, one per SIMD instead of two for QS = 2 -- the softmax VALU stream of a wave no longer shares its SIMD.  Same
// arithmetic per query, same key order: bit-identical to QS = 1 (tests/test_gpu_dit.py::test_attention_query_split_matches).
template <int T, int JQ, int HD, int MODE = 0, int NCH = 1, int QS = 1>
__global__ __launch_bounds__((T / (32 * JQ * QS)) * 64, HD == 64 && QS == 1 ? (T / (32 * JQ)) / 2 : 2) void dit_attention_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ Vt, half_t* __restrict__ O, int D, int heads,
    float scale_log2e, int stag) {
  constexpr int TT = T * NCH;  // tokens of the sequence
#ifdef LFM_MEASURE
  if (stag > 0) {  // measurement (tools/stagger_probe.py): the second residents of the first wave of workgroups start late
    const int lin = blockIdx.x + blockIdx.y * gridDim.x;
    if (lin >= 256 && lin < 512) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)stag) __builtin_amdgcn_s_sleep(16);
    }
  }
#endif
  static_assert(NCH == 1 || MODE == 0, "measurement variants are single-chunk");
  static_assert(QS == 1 || (NCH == 1 && MODE == 0 && T % (32 * JQ * QS) == 0), "the query split is for the single-chunk product kernel");
  static_assert(HD % 8 == 0 && HD >= 32 && HD <= 128, "head_dim: whole 16-byte chunks");
  constexpr int NKB = T / 32;         // 32-key blocks
  constexpr int NW = T / (32 * JQ * QS);  // waves: each owns JQ blocks of 32 queries
  constexpr int NTHR = NW * 64;
  constexpr int KS = (HD + 15) / 16;  // k-slots of the S MFMAs (the last one half empty when HD % 16 == 8)
  constexpr int NDB = (HD + 31) / 32; // 32-row blocks of O^T
  constexpr int KCH = HD / 8;         // 16-B chunks per K row
  constexpr int KROW = KCH * 16;      // bytes per K row in LDS
  constexpr bool KSWZ = HD == 64;     // 128-B rows: XOR swizzle; rows of an odd number of chunks are conflict-free as they are
  constexpr int VKEY = (T / 8 - 1) < 15 ? (T / 8 - 1) : 15;  // V^T swizzle key mask (stays inside the row)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;             // [T][HD] halves
  char* Vs = smem + T * KROW;  // [HD][T] halves, 2T-B rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.x, img = blockIdx.y;
  const bool tr_first = blockIdx.x == 0 && blockIdx.y == 0, tr_last = blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1;
  auto stamp = [&](int slot) {
    if constexpr (MODE == 3) {
      if (wave == 0 && (tr_first || tr_last)) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (lane == 0) att_trace[(tr_first ? 0 : 32) + slot] = t;
      }
    }
  };
  stamp(0);
  auto wg_stamp = [&](int slot) {
    if constexpr (MODE == 3) {
      const int lin = blockIdx.y * gridDim.x + blockIdx.x;
      if (wave == 0 && lin < ATT_WG_TRACE) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (slot == 0) {
          const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
          if (lane == 0) att_wg_trace[lin][0] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        }
        if (lane == 0) att_wg_trace[lin][slot + 1] = t;
      }
    }
  };
  wg_stamp(0);
  const int qblk = NCH > 1 ? (int)blockIdx.z : 0;
  const half_t* Kg = K + (long)img * TT * D + head * HD;
  const half_t* Vg = Vt + ((long)img * heads + head) * HD * TT;
  // buffer-addressed LDS-DMA (SGPR resource + 32-bit VGPR byte offset: +13 % LDS-DMA throughput per CU over 64-bit VGPR addresses,
  // tools/ubench/ldsdma_rate.hip -- the load phase is a third of a workgroup's life)
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, -1, 0x00020000);

  // ---- stage K, fetch this wave's Q fragments, then stage V^T.  Issue order = completion order for loads, so
  // vmcnt(VMIN) below means "K and Q are here, V^T may still be flying"; V^T is awaited before the first PV.
  // T * HD / 8 slots of 16 B over NTHR lanes: 8 or 4 full passes for hd 64, 4.5 for hd 72 (the last pass is issued by the lower
  // half of the waves only: whole waves, so the DMA's implicit lane*16 destination stays dense).
  constexpr int SLOTS = T * KCH;
  constexpr int NPASS = (SLOTS + NTHR - 1) / NTHR;
  constexpr int VMIN = SLOTS / NTHR;  // V^T DMAs every wave issues
  static_assert(VMIN == 8 || VMIN == 4, "the counted wait below is vmcnt(VMIN)");
  static_assert(SLOTS % NTHR == 0 || (SLOTS % NTHR) % 64 == 0, "a partial pass is made of whole waves");
  // (the chunk loop of the hd-64 kernel passes a laundered copy of tid: the 2 * NPASS per-lane offsets and LDS destinations are recomputed at every re-stage
  // instead of living -- spilled, 19 dwords under the 128-VGPR budget of two workgroups per CU -- across the key-block loop: 8 x 16 x 1024 x 64 59-62 -> 53-55 us,
  // tools/attn_1024_time.py; the NCH = 1 kernels issue both stages once, before anything else is live)
  auto stage_k = [&](int chunk, unsigned tid_) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const unsigned s = p * NTHR + tid_, wave_ = tid_ >> 6;
      if (SLOTS % NTHR == 0 || p + 1 < NPASS || s < SLOTS) {
        const int row = s / KCH, ch = s - row * KCH;
        const int c = KSWZ ? (ch ^ ((row >> 1) & 7)) : ch;
        if (MODE != 2) glds16_buf(rs_k, (unsigned)((chunk * T + row) * D + c * 8) * 2u, 0u, Ks + (p * NTHR + wave_ * 64) * 16);
      }
    }
  };
  auto stage_v = [&](int chunk, unsigned tid_) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const unsigned s = p * NTHR + tid_, wave_ = tid_ >> 6;
      if (SLOTS % NTHR == 0 || p + 1 < NPASS || s < SLOTS) {
        constexpr int CPR = T / 8;  // 16-B chunks per V^T row of the staged key chunk
        const int row = s / CPR, c = (s % CPR) ^ (row & VKEY);
        if (MODE != 2) glds16_buf(rs_v, (unsigned)(row * TT + chunk * T + c * 8) * 2u, 0u, Vs + (p * NTHR + wave_ * 64) * 16);
      }
    }
  };
  stage_k(0, tid);
  const int q0 = ((QS > 1 ? (int)blockIdx.z * NW : 0) + wave) * 32 * JQ;
  const int hsel = lane >> 5, l31 = lane & 31;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  half8_t qf[JQ][KS];
#pragma unroll
  for (int jq = 0; jq < JQ; ++jq)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half_t* qp = Q + ((long)img * TT + qblk * T + q0 + jq * 32 + l31) * D + head * HD;
      if (ks * 16 + 16 <= HD) qf[jq][ks] = *(const half8_t*)(qp + ks * 16 + hsel * 8);
      else qf[jq][ks] = hsel ? zero8 : *(const half8_t*)(qp + ks * 16);
    }
  stage_v(0, tid);

  if constexpr (MODE == 1) {  // everything has landed -> one output row per query, straight from the Q registers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int jq = 0; jq < JQ; ++jq)
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks)
        *(half8_t*)(O + ((long)img * TT + qblk * T + q0 + jq * 32 + l31) * D + head * HD + ks * 16 + hsel * 8) = qf[jq][ks];
    return;
  }
  stamp(1);
  f32x16 Oa[JQ][NDB];  // [jq][db]
#pragma unroll
  for (int jq = 0; jq < JQ; ++jq)
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) Oa[jq][db][e] = 0.f;
  float mrun[JQ], lrun[JQ];
#pragma unroll
  for (int jq = 0; jq < JQ; ++jq) {
    mrun[jq] = -3.0e38f;
    lrun[jq] = 0.f;
  }

  if (VMIN == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // raw barrier: __syncthreads() would drain the in-flight V^T DMAs
  asm volatile("" ::: "memory");
  stamp(2);
  wg_stamp(1);

  f32x16 zero16;
#pragma unroll
  for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
  // S^T block kb: 32 keys x 32 JQ queries (the first MFMA takes a shared all-zero C: no per-block accumulator clears)
  auto qk = [&](f32x16 (&S)[JQ], int kb) {
    const int row = kb * 32 + l31;
    const int key = KSWZ ? ((row >> 1) & 7) : 0;
    const char* kp = Ks + row * KROW;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      half8_t kf;
      if (ks * 16 + 16 <= HD) kf = *(const half8_t*)(kp + (((ks * 2 + hsel) ^ key) << 4));
      else {  // half slot: the upper 8 dims do not exist
        kf = *(const half8_t*)(kp + (ks * 2 << 4));
        kf = hsel ? zero8 : kf;
      }
#pragma unroll
      for (int jq = 0; jq < JQ; ++jq) S[jq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[jq][ks], ks == 0 ? zero16 : S[jq], 0, 0, 0);
    }
  };
  // online softmax update for the queries this lane owns, then O^T[d][q] += sum_key V^T[d][key] P[q][key].
  // VALU diet (the kernel is VALU-issue-bound: ~1.9k VALU per wave vs 128 MFMAs): 3-input max, packed fp32 FMA / ADD on
  // register pairs, one v_permlane32_swap instead of a ds_bpermute round trip for the lane^32 exchange.
  bool first_chunk = true;
  auto softmax_pv = [&](f32x16 (&S)[JQ], int kb) {
    half8_t P[JQ][2];
#pragma unroll
    for (int jq = 0; jq < JQ; ++jq)  // attention_stream_kernel.h: optimistic exponentials against the running reference, full path for the first block / on overflow risk
      att_softmax_block<NDB>(S[jq], kb == 0 && first_chunk, mrun[jq], lrun[jq], Oa[jq], scale_log2e, P[jq]);
    if (kb == 0 && first_chunk) {  // V^T was issued after K and Q: only now must it have landed (every wave's share)
      stamp(21);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stamp(22);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        // rows past HD (third block of hd 72) re-read row HD-1: finite values into accumulator rows nobody stores
        const int d = (db * 32 + 32 <= HD) ? db * 32 + l31 : (db * 32 + l31 < HD ? db * 32 + l31 : HD - 1);
        const int vkey = d & VKEY;
        // keys {4 h + r} and {8 + 4 h + r} of the 16-key k-slot are ONE 16-byte chunk of the permuted V^T row (gemm_kernel.h: vt_pos): chunk 2 (2 kb + s) + h
        const int c0 = kb * 4 + 2 * s + hsel;
        const half8_t vf = *(const half8_t*)(Vs + d * (2 * T) + ((c0 ^ vkey) << 4));
#pragma unroll
        for (int jq = 0; jq < JQ; ++jq) Oa[jq][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, P[jq][s], Oa[jq][db], 0, 0, 0);
      }
    }
    if (kb == 0) stamp(23);
  };
  // software pipeline over the key blocks: the S MFMAs of block kb+1 are issued BEFORE the softmax VALU of block kb, so
  // the matrix pipe works underneath the VALU-heavy part instead of the wave idling on the MFMA -> max -> exp -> MFMA chain
  f32x16 Sa[JQ], Sb[JQ];
#pragma unroll 1
  for (int chunk = 0; chunk < NCH; ++chunk) {
    if (NCH > 1 && chunk > 0) {  // every wave is done with the previous chunk of keys: stage the next one, wait for all of it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      unsigned tid_l = threadIdx.x;
      if constexpr (HD == 64) asm volatile("" : "+v"(tid_l));  // hd 72 runs one workgroup per CU under 256 VGPRs: hoisted offsets cost nothing there
      stage_k(chunk, tid_l);
      stage_v(chunk, tid_l);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      first_chunk = false;
    }
    qk(Sa, 0);
#pragma unroll 1
    for (int kb = 0; kb < NKB; kb += 2) {
      qk(Sb, kb + 1);
      stamp(3 + 2 * kb);
      softmax_pv(Sa, kb);
      stamp(4 + 2 * kb);
      if (kb + 2 < NKB) qk(Sa, kb + 2);
      stamp(5 + 2 * kb);
      softmax_pv(Sb, kb + 1);
      stamp(6 + 2 * kb);
    }
  }
  // ---- normalise and store: lane owns query q, d = db*32 + 8g + 4*hsel + r
  if constexpr (HD == 64 && MODE == 0) {
    // Round 4: through the LDS (K / V^T are dead) so that a store instruction covers eight whole 128-byte output rows instead of 64 scattered 8-byte
    // pieces (the store drain was ~5k of a workgroup's ~33k cycles, profiles/r03_attention_wg_timeline.txt; MI355X guide T21).  Per-wave region of
    // 32 JQ rows x 144 B (128 B of data; the 36-dword stride keeps the 16 lanes of a ds_write_b64 group on distinct banks).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave has finished reading K / V^T
    asm volatile("" ::: "memory");
    char* ob = smem + wave * (32 * JQ * 144);
#pragma unroll
    for (int jq = 0; jq < JQ; ++jq) {
      const float inv = 1.0f / (lrun[jq] + xhalf(lrun[jq]));
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          half4_t h = {(half_t)(Oa[jq][db][4 * g] * inv), (half_t)(Oa[jq][db][4 * g + 1] * inv), (half_t)(Oa[jq][db][4 * g + 2] * inv),
                       (half_t)(Oa[jq][db][4 * g + 3] * inv)};
          *(half4_t*)(ob + (jq * 32 + l31) * 144 + (db * 32 + 8 * g + 4 * hsel) * 2) = h;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: no barrier
    half_t* obase = O + ((long)img * TT + qblk * T + q0) * D + head * HD;
#pragma unroll
    for (int i = 0; i < 4 * JQ; ++i) {
      const int row = i * 8 + (lane >> 3), ch = lane & 7;
      const half8_t v = *(const half8_t*)(ob + row * 144 + ch * 16);
      *(half8_t*)(obase + (long)row * D + ch * 8) = v;
    }
    stamp(19);
    return;
  }
#pragma unroll
  for (int jq = 0; jq < JQ; ++jq) {
    const float inv = 1.0f / (lrun[jq] + xhalf(lrun[jq]));
    half_t* orow = O + ((long)img * TT + qblk * T + q0 + jq * 32 + l31) * D + head * HD;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (db * 32 + 8 * g >= HD) continue;  // HD % 8 == 0: an 8-row group is live or dead as a whole
        half4_t h = {(half_t)(Oa[jq][db][4 * g] * inv), (half_t)(Oa[jq][db][4 * g + 1] * inv), (half_t)(Oa[jq][db][4 * g + 2] * inv),
                     (half_t)(Oa[jq][db][4 * g + 3] * inv)};
        *(half4_t*)(orow + db * 32 + 8 * g + 4 * hsel) = h;
      }
  }
  stamp(19);
  if constexpr (MODE == 3) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(20);
    wg_stamp(2);
  }
}

// 16 tokens (the DiT-x/8 family on 32x32 latents, models/DiT.py:362-415): 256 scores per (image, head) -- no MFMA tile to fill, and
// 16 rows of hd halves are served by the L1/L2 as they are.  One lane per query, four (image, head) items per wave; the lanes of an item
// read the same K row / V^T row at the same time (one broadcast fetch).  Same arithmetic as the tiled kernel: fp32 scores, exp2 of the
// scaled difference to the row maximum, P rounded to fp16 before P V, the row sum taken from the unrounded values, fp32 accumulation.
template <int HD>
__global__ __launch_bounds__(64) void dit_attention_t16_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ Vt,
                                                               half_t* __restrict__ O, int D, int heads, int items, float scale_log2e) {
  constexpr int T = 16, NC = HD / 8;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 4), q = threadIdx.x & 15;
  if (item >= items) return;
  const int img = item / heads, head = item - img * heads;
  const half_t* qp = Q + ((long)img * T + q) * D + head * HD;
  const half_t* kp = K + (long)img * T * D + head * HD;
  const half_t* vp = Vt + ((long)img * heads + head) * HD * T;
  half8_t qf[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) qf[c] = *(const half8_t*)(qp + c * 8);
  float s[T];
  float mx = -3.0e38f;
#pragma unroll
  for (int k = 0; k < T; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const half8_t kf = *(const half8_t*)(kp + (long)k * D + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += (float)qf[c][e] * (float)kf[e];
    }
    s[k] = acc;
    mx = fmaxf(mx, acc);
  }
  float l = 0.f;
  half_t ph[T];
#pragma unroll
  for (int k = 0; k < T; ++k) {
    const float p = __builtin_amdgcn_exp2f((s[k] - mx) * scale_log2e);
    l += p;
    ph[k] = (half_t)p;
  }
  const float inv = 1.0f / l;
  half_t* op = O + ((long)img * T + q) * D + head * HD;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    half8_t o8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const half8_t v0 = *(const half8_t*)(vp + (c * 8 + e) * T);
      const half8_t v1 = *(const half8_t*)(vp + (c * 8 + e) * T + 8);
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += (float)ph[vt_pos(k)] * (float)v0[k] + (float)ph[vt_pos(k + 8)] * (float)v1[k];  // memory position -> token (vt_pos)
      o8[e] = (half_t)(acc * inv);
    }
    *(half8_t*)(op + c * 8) = o8;
  }
}

// Q, K: [batch*T, heads*hd] token-major; Vt: [batch][heads*hd][T]; O: [batch*T, heads*hd].  hd 64 / 72; T in {16, 64, 128, 256, 1024}.
static int attention_launch(const half_t* Q, const half_t* K, const half_t* Vt, half_t* O, int batch, int heads, int hd, int T, hipStream_t st) {
  if (hd != 64 && hd != 72) return LFM_ERR_SHAPE;
  const int D = heads * hd;
  const float sl2 = (hd == 64 ? 0.125f : 0.11785113019775793f) * 1.4426950408889634f;  // hd^-0.5 * log2(e)
  const size_t lds = (size_t)T * hd * 4;  // K + V^T, 2 bytes each
  dim3 grid(heads, batch);
  // Rounds 1-3: 8 waves x 32 queries (4 waves/SIMD) measured 44.3 us vs 40.2 us for 4 waves x 64 queries (two 8-byte V^T reads per fragment then).
  // Round 4: with the V^T operand a single conflict-free ds_read_b128 (vt_pos) the balance flipped -- 8 waves x 32 queries (126 VGPRs: four waves per
  // SIMD) 35.7 us, 4 waves x 64 queries (228 VGPRs: two) 38.9 us -- so the narrow shape is the default; flag 256 selects the wide one (A/B)
  const bool narrow = T == 256 && !(lfm_gemm_debug_flags() & 256);
  [[maybe_unused]] const int mode = (lfm_gemm_debug_flags() >> 25) & 3;  // flags 33554432 / 67108864 / both: the measurement-only variants MODE 1 / 2 / 3 (hd 64, 256 tokens)
#ifdef LFM_MEASURE
  if (mode && hd == 64 && T == 256 && narrow && lfm_attention_stream_enabled() && batch * heads > 64) {  // the streamed kernel's phase split
    const int rc = attention_stream_launch(Q, K, Vt, O, batch, heads, st, mode);
    if (rc <= 0) return rc;
  }
  if (mode && hd == 64 && T == 256 && narrow) {  // the shipped shape (8 waves x 32 queries); flag 256 + mode = the wide one below
    static bool set = false;
    if (!set) {
      (void)hipFuncSetAttribute((const void*)dit_attention_kernel<256, 1, 64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 4);
      (void)hipFuncSetAttribute((const void*)dit_attention_kernel<256, 1, 64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 4);
      (void)hipFuncSetAttribute((const void*)dit_attention_kernel<256, 1, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 4);
      set = true;
    }
    if (mode == 3) hipLaunchKernelGGL((dit_attention_kernel<256, 1, 64, 3>), grid, dim3(512), lds, st, Q, K, Vt, O, D, heads, sl2, lfm_stagger_ticks());
    else if (mode == 1) hipLaunchKernelGGL((dit_attention_kernel<256, 1, 64, 1>), grid, dim3(512), lds, st, Q, K, Vt, O, D, heads, sl2, lfm_stagger_ticks());
    else hipLaunchKernelGGL((dit_attention_kernel<256, 1, 64, 2>), grid, dim3(512), lds, st, Q, K, Vt, O, D, heads, sl2, lfm_stagger_ticks());
    LFM_CHECK_LAUNCH();
    return LFM_OK;
  }
  if (mode && hd == 64 && T == 256) {
    static bool set = false;
    if (!set) {
      (void)hipFuncSetAttribute((const void*)dit_attention_kernel<256, 2, 64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 4);
      (void)hipFuncSetAttribute((const void*)dit_attention_kernel<256, 2, 64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 4);
      (void)hipFuncSetAttribute((const void*)dit_attention_kernel<256, 2, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 4);
      set = true;
    }
    if (mode == 3) hipLaunchKernelGGL((dit_attention_kernel<256, 2, 64, 3>), grid, dim3(256), lds, st, Q, K, Vt, O, D, heads, sl2, lfm_stagger_ticks());
    else if (mode == 1) hipLaunchKernelGGL((dit_attention_kernel<256, 2, 64, 1>), grid, dim3(256), lds, st, Q, K, Vt, O, D, heads, sl2, lfm_stagger_ticks());
    else hipLaunchKernelGGL((dit_attention_kernel<256, 2, 64, 2>), grid, dim3(256), lds, st, Q, K, Vt, O, D, heads, sl2, lfm_stagger_ticks());
    LFM_CHECK_LAUNCH();
    return LFM_OK;
  }
#endif
  if (T == 16) {
    const int items = batch * heads;
    if (hd == 64) hipLaunchKernelGGL(dit_attention_t16_kernel<64>, dim3((items + 3) / 4), dim3(64), 0, st, Q, K, Vt, O, D, heads, items, sl2);
    else hipLaunchKernelGGL(dit_attention_t16_kernel<72>, dim3((items + 3) / 4), dim3(64), 0, st, Q, K, Vt, O, D, heads, items, sl2);
    LFM_CHECK_LAUNCH();
    return LFM_OK;
  }
  // the dynamic-LDS attribute is per (function, device): one bit per device and instantiation
  const unsigned long long dbit = lfm_device_bit();
#define ATT_CASE(TT, JQ, HD)                                                                                                             \
  {                                                                                                                                     \
    static lfm_device_mask set{0};                                                                                                      \
    if (lfm_device_todo(set, dbit)) {                                                                                                   \
      (void)hipFuncSetAttribute((const void*)dit_attention_kernel<TT, JQ, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, TT * HD * 4); \
      lfm_device_done(set, dbit);                                                                                                       \
    }                                                                                                                                   \
    hipLaunchKernelGGL((dit_attention_kernel<TT, JQ, HD>), grid, dim3((TT / (32 * JQ)) * 64), lds, st, Q, K, Vt, O, D, heads, sl2, lfm_stagger_ticks()); \
  }
  if (T == 1024) {  // four key chunks of 256 through the LDS, one workgroup per 256 queries
    const dim3 grid4(heads, batch, 4);
    if (hd == 64) {
      static lfm_device_mask set{0};
      if (lfm_device_todo(set, dbit)) (void)hipFuncSetAttribute((const void*)dit_attention_kernel<256, 1, 64, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 4);
      lfm_device_done(set, dbit);
      hipLaunchKernelGGL((dit_attention_kernel<256, 1, 64, 0, 4>), grid4, dim3(512), (size_t)256 * 64 * 4, st, Q, K, Vt, O, D, heads, sl2, lfm_stagger_ticks());
    } else {
      static lfm_device_mask set{0};  // its own flag: the hd-72 kernel needs 72 KiB, above the 64-KiB default, whatever the hd-64 one did before
      if (lfm_device_todo(set, dbit)) (void)hipFuncSetAttribute((const void*)dit_attention_kernel<256, 1, 72, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 72 * 4);
      lfm_device_done(set, dbit);
      hipLaunchKernelGGL((dit_attention_kernel<256, 1, 72, 0, 4>), grid4, dim3(512), (size_t)256 * 72 * 4, st, Q, K, Vt, O, D, heads, sl2, lfm_stagger_ticks());
    }
    LFM_CHECK_LAUNCH();
    return LFM_OK;
  }
  if (hd == 64 && T == 256 && narrow && batch * heads <= 64) {  // latency mode: two workgroups of four waves per (image, head)
    static lfm_device_mask set{0};
    if (lfm_device_todo(set, dbit)) (void)hipFuncSetAttribute((const void*)dit_attention_kernel<256, 1, 64, 0, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 4);
    lfm_device_done(set, dbit);
    hipLaunchKernelGGL((dit_attention_kernel<256, 1, 64, 0, 1, 2>), dim3(heads, batch, 2), dim3(256), lds, st, Q, K, Vt, O, D, heads, sl2, 0);
    LFM_CHECK_LAUNCH();
    return LFM_OK;
  }
  if (hd == 64 && T == 256 && narrow && lfm_attention_stream_enabled()) {  // round 6: persistent workgroups, K / V^T streamed through an LDS ring (attention_stream_kernel.h)
    const int rc = attention_stream_launch(Q, K, Vt, O, batch, heads, st);
    if (rc <= 0) return rc;  // 1: not for that kernel (tensors of 2 GiB and more) -> the per-item kernel below
  }
  if (hd == 64) {
    if (T == 64) ATT_CASE(64, 2, 64)
    else if (T == 128) ATT_CASE(128, 2, 64)
    else if (T == 256 && narrow) ATT_CASE(256, 1, 64)
    else if (T == 256) ATT_CASE(256, 2, 64)
    else return LFM_ERR_SHAPE;
  } else {  // hd 72: one query block per wave (48 accumulator + 20 Q registers per block)
    if (T == 64) ATT_CASE(64, 1, 72)
    else if (T == 128) ATT_CASE(128, 1, 72)
    else if (T == 256) ATT_CASE(256, 1, 72)
    else return LFM_ERR_SHAPE;
  }
#undef ATT_CASE
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
