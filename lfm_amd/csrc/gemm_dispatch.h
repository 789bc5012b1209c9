// Kernel selection for every GEMM of the library (all generations share the operand / A-source / epilogue interfaces).
#pragma once
#include "gemm256h_kernel.h"

// Dispatcher: a 256x256 kernel when the problem fills the chip with such tiles, the 256x128 two-per-CU kernel (v4) for chip-filling
// problems that are only 128 columns wide (or where it measured faster, see lfm_gemm_v4_shapes), v1 otherwise.
// Chip-filling 256x256 problems with K % 64 == 0 take the 16x16x32-MFMA generation (v5, gemm256h_kernel.h); flag 1048576 = v3 instead (A/B).
// lfm_gemm_select() (0 auto, 1 force v1, 2 force v2, 3 force v3 -- v2 when K % 64 != 0, 4 force v4, 5 force v5) exists for A/B measurements
// and parity tests of all kernels.
template <class ASrc, class Epi>
static inline int launch_gemm_auto(const ASrc& asrc, const half_t* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t stream,
                                   int batch = 1, long bsA = 0, long bsW = 0, long bsC = 0) {
  const long tiles256 = (long)cdiv(M, 256) * cdiv(N, 256) * batch;
  const int sel = lfm_gemm_selected();
  const bool big = tiles256 >= 192 && N >= 256 && M >= 256;
  if ((K % G256N_BK) == 0) {
    const long tiles128 = (long)cdiv(M, 256) * cdiv(N, G256N_BN) * batch;
    const bool narrow = N > 64 && N < 256 && tiles128 >= 256;  // e.g. the 128-channel convolutions at 256^2 / 512^2
    if (sel == 4 || (sel == 0 && (narrow || (big && lfm_gemm_prefers_v4(M, N, K)))))
      return launch_gemm256n_tn(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
  }
  if ((sel == 5 || (sel == 0 && big && !(lfm_gemm_debug_flags() & 1048576))) && (K % G256Q_BK) == 0)
    return launch_gemm256h_tn(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
  if ((sel == 3 || (sel == 0 && big)) && (K % G256Q_BK) == 0) {
    if (lfm_gemm_debug_flags() & 1)  // A/B switch: the two-barriers-per-phase schedule
      return launch_gemm256q_tn<ASrc, Epi, false, 0>(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
    return launch_gemm256q_tn(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
  }
  if (sel == 2 || sel == 3 || sel == 4 || sel == 5 || (sel == 0 && big)) return launch_gemm256_tn(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
  return launch_gemm_tn(asrc, W, ldw, M, N, K, epi, stream, batch, bsA, bsW, bsC);
}
