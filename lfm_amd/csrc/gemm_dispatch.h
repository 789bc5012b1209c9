// Kernel selection for every GEMM of the library (all generations share the operand / A-source / epilogue interfaces).
#pragma once
#include <type_traits>
#include "gemm256w_kernel.h"

// v6 (gemm256w_kernel.h, one wave per SIMD) serves row-major A operands and epilogues without per-lane tile accumulators
template <class ASrc, class Epi>
struct gemm_v6_ok {
  static constexpr bool value = std::is_same<ASrc, ASrcRowMajor>::value && !epi_has_finish_tile<Epi>::value;
};
int lfm_gemm_v6_default();  // 1: chip-filling row-major GEMMs with K % 64 == 0 take v6 instead of v5 (lfm_set_option key 2)

// Dispatcher: the 256x256 kernel (16x16x32 MFMAs, gemm256h_kernel.h) when the problem fills the chip with such tiles and K % 64 == 0, the 256x128
// two-per-CU kernel (gemm256n_kernel.h) for chip-filling problems that are only 128 columns wide (or where it measured faster, see
// lfm_gemm_prefers_v4), the 128x128 kernel otherwise.  lfm_gemm_select() (0 auto, 1 / 4 / 5 force a kernel) exists for A/B measurements and
// for parity tests of all kernels.
// which kernel launch_gemm_auto takes for a shape: 1 = 128x128, 4 = 256x128, 5 = 256x256 (callers that depend on the epilogue's lane mapping ask)
static inline int gemm_auto_choice(int M, int N, int K, int batch = 1) {
  const long tiles256 = (long)cdiv(M, 256) * cdiv(N, 256) * batch;
  const int sel = lfm_gemm_selected();
  const bool big = tiles256 >= 192 && N >= 256 && M >= 256;
  if ((K % G256N_BK) == 0) {
    const long tiles128 = (long)cdiv(M, 256) * cdiv(N, G256N_BN) * batch;
    const bool narrow = N > 64 && N < 256 && tiles128 >= 256;
    const bool mid = !big && N >= 256 && tiles128 >= 192;  // see launch_gemm_auto
    if (sel == 4 || (sel == 0 && (narrow || mid || (big && lfm_gemm_prefers_v4(M, N, K))))) return 4;
  }
  if (sel == 6 && (K % G256Q_BK) == 0) return 6;
  if ((sel == 5 || (sel == 0 && big)) && (K % G256Q_BK) == 0) return 5;
  return 1;
}

template <class ASrc, class Epi>
static inline int launch_gemm_auto(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t stream,
                                   int batch = 1, long bsA = 0, long bsW = 0, long bsC = 0) {
  const long tiles256 = (long)cdiv(M, 256) * cdiv(N, 256) * batch;
  const int sel = lfm_gemm_selected();
  const bool big = tiles256 >= 192 && N >= 256 && M >= 256;
  if ((K % G256N_BK) == 0) {
    const long tiles128 = (long)cdiv(M, 256) * cdiv(N, G256N_BN) * batch;
    const bool narrow = N > 64 && N < 256 && tiles128 >= 256;  // e.g. the 128-channel convolutions at 256^2 / 512^2
    // Round 6: problems too small for the 256x256 tiling (< 192 such tiles) but with >= 192 tiles of 256x128 -- the UNets' 1x1 convolutions / attention
    // projections at 16 384 x 384 .. 512, 4096 x 1536, 32 768 x 256 -- ran on the 128x128 kernel: 14.4 vs 17.2, 18.0 vs 20.2, 16.8 vs 19.3, 18.6 vs 20.4 us here
    // (tools/linear_shapes_probe.py, profiles/r06_linear_shapes_probe.txt; below 192 tiles the 128x128 kernel wins: 4096 x 512 11.0 vs 13.8 us).  Bit-identical.
    const bool mid = !big && N >= 256 && tiles128 >= 192;
    if (sel == 4 || (sel == 0 && (narrow || mid || (big && lfm_gemm_prefers_v4(M, N, K)))))
      return launch_gemm256n_tn(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
  }
  // the 256x256 kernels address row-major operands through buffer resources (unsigned 32-bit byte offsets: operands below 2^31 elements, like the 32-bit row offsets of every kernel)
  bool fits256 = (long)N * ldw < (1L << 31);
  if constexpr (asrc_has_buffer<ASrc>::value) fits256 = fits256 && asrc_fits_buffer(asrc, 0);
  if constexpr (gemm_v6_ok<ASrc, Epi>::value) {
    if ((sel == 6 || (sel == 0 && big && lfm_gemm_v6_default())) && (K % G256Q_BK) == 0 && fits256)
      return launch_gemm256w_tn(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
  }
  if ((sel == 5 || sel == 6 || (sel == 0 && big)) && (K % G256Q_BK) == 0 && fits256)
    return launch_gemm256h_tn(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
  return launch_gemm_tn(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
}

// slices of a deep small-map problem on the 256x256 kernel (declared in gemm_kernel.h, which the 256x256 kernels include): slice bz = K range
// [bz ks, (bz + 1) ks) through the batch index, fp32 partial tile into slab[bz].  Returns 1 when the A source cannot take the kernel.
template <class ASrc, class Epi>
static inline int launch_gemm_splitk256(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int ks, const EpiSlabF32& e, hipStream_t stream, int S) {
  if ((long)N * ldw >= (1L << 31)) return 1;
  if constexpr (asrc_has_buffer<ASrc>::value) {
    if (!asrc_fits_buffer(asrc, 0)) return 1;
  }
  return launch_gemm256h_tn(asrc, W, ldw, M, N, ks, e, stream, S, ks, ks, 0);
}
