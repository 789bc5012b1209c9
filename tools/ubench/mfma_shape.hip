// Microbenchmark: sustained MFMA rate under the power cap, 32x32x16 vs 16x16x32 f16, register-resident random operands, whole chip.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip ; run: ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const half8* __restrict__ in, float* out, int iters) {
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(threadIdx.x * 8 + i) % 4096];
    b[i] = in[(threadIdx.x * 8 + 4 + i) % 4096];
  }
  if constexpr (SHAPE == 32) {
    f32x16 c[8];
    for (int j = 0; j < 8; ++j)
      for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 3], b[(j >> 1) & 3], c[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][7];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    f32x4 c[32];  // same 128 accumulator registers
    for (int j = 0; j < 32; ++j) c[j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j & 3], b[(j >> 2) & 3], c[j], 0, 0, 0);
#pragma unroll
      for (int j = 16; j < 32; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j & 3], b[(j >> 2) & 3], c[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 32; ++j) s += c[j][0] + c[j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

int main() {
  std::vector<_Float16> h(4096 * 8);
  half8* din;
  float* dout;
  hipMalloc(&din, h.size() * 2);
  hipMalloc(&dout, 256 * 8 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int fill = 0; fill < 3; ++fill) {
    srand(1);
    for (auto& v : h) {
      float r = (float)rand() / RAND_MAX * 2 - 1;
      v = fill == 0 ? (_Float16)0.f : (fill == 1 ? (_Float16)r : (_Float16)(r * 0.03f));
    }
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int shape : {32, 16, 32, 16}) {
      const int iters = 20000, blocks = 256;
      auto run = [&]() {
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(512), 0, 0, din, dout, iters);
        else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(512), 0, 0, din, dout, iters);
      };
      run();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 3; ++r) run();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 3;
      // flops: per wave per iter: SHAPE 32: 8 x 32768; SHAPE 16: 32 x 16384 -> 524288 vs 262144*... (32 x 16384 = 524288; 8 x 32768 = 262144)
      const double per_iter = shape == 32 ? 8.0 * 32768 : 32.0 * 16384;
      const double tf = per_iter * iters * 8 * blocks / (ms * 1e-3) / 1e12;
      printf("fill=%s mfma %s: %.2f ms  %.0f TFLOP/s\n", fill == 0 ? "zeros" : (fill == 1 ? "uniform[-1,1)" : "uniform*0.03"), shape == 32 ? "32x32x16" : "16x16x32", ms, tf);
    }
  }
  return 0;
}
