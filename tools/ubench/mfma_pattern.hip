// Microbenchmark (round 3): why does the GEMM's bare MFMA stream (all loads and barriers ablated, tools/mainloop_ablation.py) stop at ~1.38 PFLOP/s
// when mfma_shape.hip sustains 1.94 on random data?  Same 16x16x32 stream, 2 waves per SIMD, 32 accumulator tiles, varying (a) how many distinct
// A / B fragments feed it (the GEMM: 8 + 8 per phase), (b) the launch shape: 256 long workgroups vs 1024 workgroups of one "tile" (1024 MFMAs per wave).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_pattern mfma_pattern.hip ; run: ./mfma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NF>
__global__ __launch_bounds__(512) void k(const half8* __restrict__ in, float* out, int iters) {
  half8 a[NF], b[NF];
  for (int i = 0; i < NF; ++i) {
    a[i] = in[(threadIdx.x * 8 + i) % 4096];
    b[i] = in[(threadIdx.x * 8 + NF + i) % 4096];
  }
  f32x4 c[32];
  for (int j = 0; j < 32; ++j) c[j] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j % NF], b[(j / 4) % NF], c[j], 0, 0, 0);
  }
  float s = 0;
  for (int j = 0; j < 32; ++j) s += c[j][0] + c[j][3];
  out[(blockIdx.x % 256) * 512 + threadIdx.x] = s;
}

int main() {
  std::vector<_Float16> h(4096 * 8);
  half8* din;
  float* dout;
  hipMalloc(&din, h.size() * 2);
  hipMalloc(&dout, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int fill = 0; fill < 2; ++fill) {
    srand(1);
    for (auto& v : h) v = fill == 0 ? (_Float16)0.f : (_Float16)((float)rand() / RAND_MAX * 2 - 1);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int nf : {4, 8}) {
      for (int shape = 0; shape < 3; ++shape) {
        const int blocks = shape == 0 ? 256 : (shape == 1 ? 1024 : 4096);
        const int iters = shape == 0 ? 16384 : 32;  // 32 iterations x 32 MFMAs = the 1024 MFMAs a wave issues per 256x256x1024 GEMM tile
        const int reps = shape == 0 ? 1 : 128;
        auto run = [&]() {
          if (nf == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(512), 0, 0, din, dout, iters);
          else hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(512), 0, 0, din, dout, iters);
        };
        run();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) run();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 32.0 * 16384 * iters * 8.0 * blocks * reps;
        printf("fill=%s fragments=%d+%d blocks=%4d iters=%5d (%3d launches): %8.3f ms  %5.0f TFLOP/s\n", fill ? "uniform[-1,1)" : "zeros", nf, nf, blocks,
               iters, reps, ms, flop / (ms * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
