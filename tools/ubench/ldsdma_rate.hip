// Microbenchmark (round 4): the LDS-DMA (global_load_lds_dwordx4) ceiling of one CU on L2-resident data -- the GEMMs stage 64 KiB per 64-deep K-tile
// per CU, and removing the DMAs from the main loop is worth 15 us of ~95 in BOTH the eight-wave and the four-wave kernel (profiles/r04_w6_ablation.txt).
// Every workgroup walks the same 2 MiB region (so it sits in each XCD's L2) the way a GEMM walks an operand panel: a wave-instruction covers
// ROWS rows x (1024 / ROWS) bytes (8 rows x 128 B = the 128-byte-row kernels, 16 rows x 64 B = the 64-byte-row ones).  No MFMA, no reads: the pure rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_rate ldsdma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half_t;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int THREADS, int ROWS, int INFLIGHT>
__global__ __launch_bounds__(THREADS) void k(const char* __restrict__ src, float* out, int iters, long ld, int region_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int LPR = 64 / ROWS;                 // lanes per row
  const int row0 = wave * ROWS + lane / LPR;     // this lane's row inside a pass of (THREADS / 64) * ROWS rows
  const int coff = (lane % LPR) * 16;
  constexpr int PASS_ROWS = (THREADS / 64) * ROWS;
  int r = (blockIdx.x * 64) % region_rows;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {  // 16 wave-instructions = 16 KiB per wave
      const char* g = src + (long)((r + p * PASS_ROWS + row0) % region_rows) * ld + coff;
      __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(smem + ((it & 1) * 16 + p) * (THREADS / 64) * 1024 + wave * 1024), 16, 0, 0);
      if constexpr (INFLIGHT < 16) {
        if (p % INFLIGHT == INFLIGHT - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    r = (r + 16 * PASS_ROWS) % region_rows;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[blockIdx.x * THREADS + tid] = smem[tid * 4];
}

// the same walk with the address as SGPR base + 32-bit VGPR offset (MODE 0: global_load_lds_dwordx4 v, s[..]; MODE 1: buffer_load_dwordx4 ... offen lds)
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void ksaddr(const char* __restrict__ src, float* out, int iters, long ld, int region_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = wave * 8 + lane / 8, coff = (lane % 8) * 16;
  constexpr int PASS_ROWS = (THREADS / 64) * 8;
  int r = (blockIdx.x * 64) % region_rows;
  i32x4 rsrc;
  rsrc.x = (int)(unsigned)(unsigned long long)src;
  rsrc.y = (int)((unsigned long long)src >> 32);
  rsrc.z = 0x7fffffff;
  rsrc.w = 0x00020000;  // raw buffer, dword format irrelevant for raw loads
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const unsigned off = (unsigned)(((r + p * PASS_ROWS + row0) % region_rows) * (int)ld + coff);
      const unsigned ldsb = (unsigned)(unsigned long long)(smem + ((it & 1) * 16 + p) * (THREADS / 64) * 1024 + wave * 1024);
      const unsigned m0v = __builtin_amdgcn_readfirstlane(ldsb);
      if constexpr (MODE == 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(src), "s"(m0v) : "memory");
      else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off), "s"(rsrc), "s"(m0v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    r = (r + 16 * PASS_ROWS) % region_rows;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[blockIdx.x * THREADS + tid] = smem[tid * 4];
}
template <int THREADS, int MODE>
static void bench_saddr(const char* name, const char* src, float* out, long ld, int region_rows) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto kern = ksaddr<THREADS, MODE>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int iters = 2048;
  hipLaunchKernelGGL(kern, dim3(256), dim3(THREADS), 131072, 0, src, out, 64, ld, region_rows);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(THREADS), 131072, 0, src, out, iters, ld, region_rows);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)iters * 16 * 1024 * (THREADS / 64) * 256;
  printf("%-72s: %7.3f ms  %6.2f TB/s chip  %5.1f B/clk/CU at 2.4 GHz  %5.1f clk per wave-instruction per CU\n", name, ms, bytes / (ms * 1e-3) / 1e12,
         bytes / 256 / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (THREADS / 64)));
  fflush(stdout);
}

// the same walk with plain global_load_dwordx4 into registers (16 in flight per wave), optionally followed by the ds_write_b128 into the LDS image
typedef float f4 __attribute__((ext_vector_type(4)));
template <int THREADS, int WRITE>
__global__ __launch_bounds__(THREADS) void kreg(const char* __restrict__ src, float* out, int iters, long ld, int region_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = wave * 8 + lane / 8, coff = (lane % 8) * 16;
  constexpr int PASS_ROWS = (THREADS / 64) * 8;
  int r = (blockIdx.x * 64) % region_rows;
  f4 acc = {0, 0, 0, 0};
  f4 v[16];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 16; ++p) v[p] = *(const f4*)(src + (long)((r + p * PASS_ROWS + row0) % region_rows) * ld + coff);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      if constexpr (WRITE) *(f4*)(smem + ((it & 1) * 16 + p) * (THREADS / 64) * 1024 + wave * 1024 + lane * 16) = v[p];
      else acc += v[p];
    }
    r = (r + 16 * PASS_ROWS) % region_rows;
  }
  __syncthreads();
  out[blockIdx.x * THREADS + tid] = acc.x + acc.y + smem[tid * 4];
}
template <int THREADS, int WRITE>
static void bench_reg(const char* name, const char* src, float* out, long ld, int region_rows) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto kern = kreg<THREADS, WRITE>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int iters = 2048;
  hipLaunchKernelGGL(kern, dim3(256), dim3(THREADS), 131072, 0, src, out, 64, ld, region_rows);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(THREADS), 131072, 0, src, out, iters, ld, region_rows);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)iters * 16 * 1024 * (THREADS / 64) * 256;
  printf("%-72s: %7.3f ms  %6.2f TB/s chip  %5.1f B/clk/CU at 2.4 GHz  %5.1f clk per wave-instruction per CU\n", name, ms, bytes / (ms * 1e-3) / 1e12,
         bytes / 256 / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (THREADS / 64)));
  fflush(stdout);
}

template <int THREADS, int ROWS, int INFLIGHT>
static void bench(const char* name, const char* src, float* out, long ld, int region_rows) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto kern = k<THREADS, ROWS, INFLIGHT>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int iters = 2048;
  hipLaunchKernelGGL(kern, dim3(256), dim3(THREADS), 131072, 0, src, out, 64, ld, region_rows);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(THREADS), 131072, 0, src, out, iters, ld, region_rows);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)iters * 16 * 1024 * (THREADS / 64) * 256;
  printf("%-72s: %7.3f ms  %6.2f TB/s chip  %5.1f B/clk/CU at 2.4 GHz  %5.1f clk per wave-instruction per CU\n", name, ms, bytes / (ms * 1e-3) / 1e12,
         bytes / 256 / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (THREADS / 64)));
  fflush(stdout);
}

int main() {
  const int region_rows = 16384;  // x 128 B = 2 MiB
  char* src;
  float* out;
  hipMalloc(&src, (size_t)region_rows * 128 + 4096);
  hipMemset(src, 1, (size_t)region_rows * 128 + 4096);
  hipMalloc(&out, 256 * 512 * 4);
  bench<256, 8, 16>("4 waves/CU, 8 rows x 128 B per instruction, 16+ in flight per wave", src, out, 128, region_rows);
  bench<512, 8, 16>("8 waves/CU, 8 rows x 128 B per instruction, 16+ in flight per wave", src, out, 128, region_rows);
  bench<256, 16, 16>("4 waves/CU, 16 rows x 64 B per instruction, 16+ in flight per wave", src, out, 128, region_rows);
  bench<512, 16, 16>("8 waves/CU, 16 rows x 64 B per instruction, 16+ in flight per wave", src, out, 128, region_rows);
  bench<256, 8, 4>("4 waves/CU, 8 rows x 128 B per instruction, <= 8 in flight per wave", src, out, 128, region_rows);
  bench<256, 8, 2>("4 waves/CU, 8 rows x 128 B per instruction, <= 4 in flight per wave", src, out, 128, region_rows);
  bench<64, 8, 16>("1 wave/CU, 8 rows x 128 B per instruction, 16+ in flight", src, out, 128, region_rows);
  bench<128, 8, 16>("2 waves/CU, 8 rows x 128 B per instruction, 16+ in flight per wave", src, out, 128, region_rows);
  bench_saddr<64, 0>("1 wave/CU, global_load_lds_dwordx4 SGPR base + VGPR offset", src, out, 128, region_rows);
  bench_saddr<256, 0>("4 waves/CU, global_load_lds_dwordx4 SGPR base + VGPR offset", src, out, 128, region_rows);
  bench_saddr<512, 0>("8 waves/CU, global_load_lds_dwordx4 SGPR base + VGPR offset", src, out, 128, region_rows);
  bench_saddr<64, 1>("1 wave/CU, buffer_load_dwordx4 offen lds", src, out, 128, region_rows);
  bench_saddr<256, 1>("4 waves/CU, buffer_load_dwordx4 offen lds", src, out, 128, region_rows);
  bench_saddr<512, 1>("8 waves/CU, buffer_load_dwordx4 offen lds", src, out, 128, region_rows);
  bench_reg<256, 0>("4 waves/CU, global_load_dwordx4 into registers, 16 in flight per wave", src, out, 128, region_rows);
  bench_reg<512, 0>("8 waves/CU, global_load_dwordx4 into registers, 16 in flight per wave", src, out, 128, region_rows);
  bench_reg<64, 0>("1 wave/CU, global_load_dwordx4 into registers, 16 in flight", src, out, 128, region_rows);
  bench_reg<256, 1>("4 waves/CU, global_load_dwordx4 + ds_write_b128 (register staging)", src, out, 128, region_rows);
  bench_reg<512, 1>("8 waves/CU, global_load_dwordx4 + ds_write_b128 (register staging)", src, out, 128, region_rows);
  return 0;
}
