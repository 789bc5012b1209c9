// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of liblfm_hip.so.
// wave = 64 lanes; MFMA fragments follow the CDNA4 32x32x16 / 16x16x32 f16 maps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef half_t half2_t __attribute__((ext_vector_type(2)));
typedef half_t half4_t __attribute__((ext_vector_type(4)));
typedef half_t half8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LFM_OK 0
#define LFM_ERR_SHAPE (-1)      // unsupported / inconsistent shape
#define LFM_ERR_ALIGN (-2)      // pointer or leading dimension not aligned as required
#define LFM_ERR_WORKSPACE (-3)  // caller-provided workspace too small
#define LFM_ERR_LAUNCH (-4)     // hipGetLastError() != hipSuccess after a launch
#define LFM_ERR_ARG (-5)        // null pointer / bad enum

#define LFM_CHECK_LAUNCH()                                  \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return LFM_ERR_LAUNCH; \
  } while (0)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16-byte async global->LDS copy: LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// GELU(tanh) as torch: 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715 x^3)))
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  // tanh(u) = 1 - 2/(exp(2u)+1); clamp to keep exp finite
  float e = __expf(2.0f * fminf(fmaxf(u, -15.0f), 15.0f));
  float th = 1.0f - 2.0f / (e + 1.0f);
  return 0.5f * x * (1.0f + th);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
