// Microbenchmark: issue cost (cycles per wave64 instruction) of the VALU operations the GELU epilogue is made of, one and two waves per SIMD,
// one CU.  s_memtime around an unrolled stream of independent instructions (16 register chains), lane 0 of wave 0 reports.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x(0) x(1) x(2) x(3) x(4) x(5) x(6) x(7) x(8) x(9) x(10) x(11) x(12) x(13) x(14) x(15)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(512) void k(const float* in, float* out, unsigned long long* cyc, int iters) {
  float v[16];
  f32x2 p[16];
  for (int i = 0; i < 16; ++i) {
    v[i] = in[threadIdx.x + 64 * i];
    p[i] = (f32x2){v[i], v[i] * 0.5f};
  }
  const float c = in[threadIdx.x];
  const f32x2 c2 = {c, c};
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
#define EXPH(i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
#define RCPH(i) asm volatile("v_rcp_f16 %0, %0" : "+v"(v[i]));
#define CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
#define PKMULH(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
#define PKFMAH(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
    if constexpr (OP == 0) { REP16(MUL) }
    if constexpr (OP == 1) { REP16(FMA) }
    if constexpr (OP == 2) { REP16(PKMUL) }
    if constexpr (OP == 3) { REP16(PKFMA) }
    if constexpr (OP == 4) { REP16(EXP) }
    if constexpr (OP == 5) { REP16(RCP) }
    if constexpr (OP == 6) { REP16(EXPH) }
    if constexpr (OP == 7) { REP16(RCPH) }
    if constexpr (OP == 8) { REP16(CVTPK) }
    if constexpr (OP == 9) { REP16(PKMULH) }
    if constexpr (OP == 10) { REP16(PKFMAH) }
    if constexpr (OP == 11) { REP16(SQRT) }
#define EXP_PK(i) EXP(i) PKMUL(i)
#define EXP_PK2(i) EXP(i) PKMUL(i) PKFMA(i)
#define EXP_PK3(i) EXP(i) PKMUL(i) PKFMA(i) PKMUL(i)
    if constexpr (OP == 12) { REP16(EXP_PK) }
    if constexpr (OP == 13) { REP16(EXP_PK2) }
    if constexpr (OP == 14) { REP16(EXP_PK3) }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i] + p[i].x + p[i].y;
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float *din, *dout;
  unsigned long long* dc;
  hipMalloc(&din, 4096 * 4);
  hipMalloc(&dout, 4096 * 4);
  hipMalloc(&dc, 8);
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = 1.0f + 1e-4f * (i % 7);
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  const char* names[] = {"v_mul_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_exp_f16", "v_rcp_f16",
                         "v_cvt_pk_f16_f32", "v_pk_mul_f16", "v_pk_fma_f16", "v_sqrt_f32", "v_exp_f32 + 1 pk (per group)", "v_exp_f32 + 2 pk (per group)",
                         "v_exp_f32 + 3 pk (per group)"};
  const int iters = 2000;
  for (int threads : {256, 512}) {  // one / two waves per SIMD
    for (int op = 0; op < 15; ++op) {
      for (int rep = 0; rep < 2; ++rep) {
#define L(o) case o: hipLaunchKernelGGL(k<o>, dim3(1), dim3(threads), 0, 0, din, dout, dc, iters); break;
        switch (op) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) }
        hipDeviceSynchronize();
      }
      unsigned long long c;
      hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
      printf("%d waves/SIMD  %-18s %6.2f cycles per instruction per wave (%.2f per SIMD issue slot)\n", threads / 256, names[op],
             (double)c / (iters * 16.0), (double)c / (iters * 16.0) / (threads / 256));
    }
  }
  return 0;
}
