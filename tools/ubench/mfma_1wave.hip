// Microbenchmark (round 4): how fast can ONE wave per SIMD issue v_mfma_f32_16x16x32_f16?  The four-wave GEMM (csrc/gemm256w_kernel.h) ran its bare
// MFMA stream (no reads, no DMA) at 1.4 PFLOP/s where the eight-wave kernel's two waves per SIMD reach 2.0-2.3.  Variants: accumulators in the
// VGPR half (builtin / inline asm) or the AGPR half (inline asm "+a"), 32 or 64 accumulator tiles, one or two waves per SIMD, operand order.
// 128 KiB of LDS per workgroup pins one workgroup per CU.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_1wave mfma_1wave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: builtin (compiler picks the register half); 1: inline asm, accumulators "+v"; 2: inline asm, accumulators "+a"
// ORDER 0: consecutive MFMAs share the B operand (i-major, the GEMM's order); 1: consecutive MFMAs share nothing (diagonal walk)
template <int THREADS, int NT, int MODE, int ORDER, int NOPS>
__global__ __launch_bounds__(THREADS) void k(const half8* __restrict__ in, float* out, int iters) {
  extern __shared__ char smem[];
  half8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = in[(threadIdx.x * 8 + i) % 4096];
    b[i] = in[(threadIdx.x * 8 + 8 + i) % 4096];
  }
  f32x4 c[NT];
  int dummy = threadIdx.x;
  half8 spare[4];
  const int ldsaddr = (threadIdx.x & 63) * 16;
  for (int j = 0; j < NT; ++j) c[j] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ia = ORDER == 0 ? (j % 8) : (j % 8), ib = ORDER == 0 ? (j / 8) % 8 : (j + j / 8) % 8;
      if constexpr (MODE == 0) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ia], b[ib], c[j], 0, 0, 0);
      else if constexpr (MODE == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a[ia]), "v"(b[ib]));
      else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[j]) : "v"(a[ia]), "v"(b[ib]));
      if constexpr (NOPS == 1) asm volatile("s_nop 0");
      if constexpr (NOPS == 2) asm volatile("v_mov_b32 %0, %0" : "+v"(dummy));
      if constexpr (NOPS == 3) {  // the GEMM's rendezvous: one s_barrier (behind a full s_waitcnt) per 128 MFMAs
        if (j == NT - 1 && (it & 1)) {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
      }
      if constexpr (NOPS == 4) {  // a fragment read (into a spare register) after every fourth MFMA
        if ((j & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(spare[(j >> 2) & 3]) : "v"(ldsaddr) : "memory");
        if (j == NT - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
  if constexpr (MODE != 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  float s = 0;
  for (int j = 0; j < NT; ++j) s += c[j][0] + c[j][3];
  if (smem[threadIdx.x] == 77) s += 1 + dummy;
  if constexpr (NOPS == 4) s += (float)spare[0][0] + (float)spare[1][1] + (float)spare[2][2] + (float)spare[3][3];
  out[(blockIdx.x % 256) * THREADS + threadIdx.x] = s;
}

template <int THREADS, int NT, int MODE, int ORDER, int NOPS>
static void bench(const char* name, const half8* din, float* dout, const char* fill) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto kern = k<THREADS, NT, MODE, ORDER, NOPS>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int iters = 16384 * 32 / NT / (THREADS / 256);
  hipLaunchKernelGGL(kern, dim3(256), dim3(THREADS), 131072, 0, din, dout, 64);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(THREADS), 131072, 0, din, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)NT * 16384 * iters * (THREADS / 64) * 256;
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)NT * iters * (THREADS / 256));  // cycles per MFMA per SIMD at 2.4 GHz
  printf("fill=%-8s %-58s: %8.3f ms  %5.0f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", fill, name, ms, flop / (ms * 1e-3) / 1e12, cyc);
  fflush(stdout);
}

int main() {
  std::vector<_Float16> h(4096 * 8);
  half8* din;
  float* dout;
  hipMalloc(&din, h.size() * 2);
  hipMalloc(&dout, 256 * 512 * 4);
  for (int fill = 0; fill < 2; ++fill) {
    srand(1);
    for (auto& v : h) v = fill == 0 ? (_Float16)0.f : (_Float16)((float)rand() / RAND_MAX * 2 - 1);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const char* f = fill ? "uniform" : "zeros";
    bench<512, 32, 0, 0, 0>("2 waves/SIMD, builtin, 32 tiles", din, dout, f);
    bench<256, 32, 0, 0, 0>("1 wave/SIMD, builtin, 32 tiles", din, dout, f);
    bench<256, 32, 1, 0, 0>("1 wave/SIMD, asm +v, 32 tiles", din, dout, f);
    bench<256, 32, 2, 0, 0>("1 wave/SIMD, asm +a, 32 tiles", din, dout, f);
    bench<256, 64, 2, 0, 0>("1 wave/SIMD, asm +a, 64 tiles (the GEMM)", din, dout, f);
    bench<256, 64, 2, 1, 0>("1 wave/SIMD, asm +a, 64 tiles, no shared operand", din, dout, f);
    bench<256, 64, 2, 0, 1>("1 wave/SIMD, asm +a, 64 tiles, s_nop 0 after each", din, dout, f);
    bench<256, 64, 2, 0, 2>("1 wave/SIMD, asm +a, 64 tiles, v_mov after each", din, dout, f);
    bench<256, 64, 2, 0, 3>("1 wave/SIMD, asm +a, 64 tiles, barrier per 128 MFMAs", din, dout, f);
    bench<256, 64, 2, 0, 4>("1 wave/SIMD, asm +a, 64 tiles, ds_read_b128 per 4 MFMAs", din, dout, f);
    bench<512, 32, 2, 0, 0>("2 waves/SIMD, asm +a, 32 tiles", din, dout, f);
    bench<256, 64, 0, 0, 0>("1 wave/SIMD, builtin, 64 tiles", din, dout, f);
  }
  return 0;
}
