// Reproducer attempt for the co-scheduling event of the folded fc1 epilogue (profiles/r05_cosched_root_cause.txt): does a
// v_pk_fma_f32 whose LOW half takes src1 from the HIGH register of the pair (op_sel:[0,1,0], the instruction the compiler emits for
// `ab.y * u + v` with (a, b) in a register pair) ever disagree with two plain v_fma_f32 on the SAME registers -- alone, and while another
// stream keeps the chip busy?  Every lane checks itself, iteration by iteration; mismatches are counted per (form, half, lane quarter).
//   form 0: op_sel:[0,1,0]        lo = u.lo * ab.HI + v.lo,  hi = u.hi * ab.hi + v.hi   (t = b u + v of EpiModGeluF16::store8r)
//   form 1: op_sel_hi:[1,0,1]     lo = x.lo * ab.lo + t.lo,  hi = x.hi * ab.LO + t.hi   (a acc + t)
//   form 2: natural halves        lo = u.lo * bb.lo + v.lo,  hi = u.hi * bb.hi + v.hi   with bb = (b, b) built by two v_mov
// Shape of the victim: 512 threads, 130 KiB of LDS (one workgroup per CU, like the 256x256 GEMM), (a, b) fetched by ds_read_b64 from a table the
// workgroup wrote, GELU-like transcendentals between the packed operations.  Built as a shared library: tools/pkfma_opsel_probe.py drives it.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ubench/_bin/libpkfma_opsel.so tools/ubench/pkfma_opsel.hip
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void pk_victim(const float* __restrict__ in, unsigned* __restrict__ counters, float* __restrict__ sink, int iters, int trans) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* rs = (float*)(smem + 131072);  // [256][2] behind 128 KiB, as the GEMM's rs[]
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < 256) {
    rs[2 * tid] = 1.5f + 0.001f * in[(blockIdx.x * 256 + tid) & 65535];
    rs[2 * tid + 1] = 0.001f * in[(blockIdx.x * 256 + tid + 7) & 65535];
  }
  __syncthreads();
  f32x2 u[4], v[4], x[4];
  for (int i = 0; i < 4; ++i) {
    u[i] = (f32x2){in[(tid + 512 * i) & 65535], in[(tid + 512 * i + 1) & 65535]};
    v[i] = (f32x2){in[(tid + 512 * i + 2) & 65535] * 0.05f, in[(tid + 512 * i + 3) & 65535] * 0.05f};
    x[i] = (f32x2){in[(tid + 512 * i + 4) & 65535] * 8.f, in[(tid + 512 * i + 5) & 65535] * 8.f};
  }
  unsigned bad[3][2] = {{0, 0}, {0, 0}, {0, 0}};
  float acc = 0.f;
  const int rrow = lane >> 3;
  for (int it = 0; it < iters; ++it) {
    const int row = (it * 8 + rrow) & 255;
    f32x2 ab[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) asm volatile("ds_read_b64 %0, %1" : "=v"(ab[ps]) : "v"((int)(131072 + 8 * ((row + 8 * ps) & 255))) : "memory");
    asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");  // as the compiled epilogue: the last read may still be in flight
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      if (ps == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f32x2 t0, x1, t2, bb;
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(t0) : "v"(u[ps]), "v"(ab[ps]), "v"(v[ps]));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(x1) : "v"(x[ps]), "v"(ab[ps]), "v"(t0));
      asm volatile("v_mov_b32 %0, %1" : "=v"(bb.x) : "v"(ab[ps].y));
      asm volatile("v_mov_b32 %0, %1" : "=v"(bb.y) : "v"(ab[ps].y));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t2) : "v"(u[ps]), "v"(bb), "v"(v[ps]));
      float r0, r1, q0, q1;
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(u[ps].x), "v"(ab[ps].y), "v"(v[ps].x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(u[ps].y), "v"(ab[ps].y), "v"(v[ps].y));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(x[ps].x), "v"(ab[ps].x), "v"(t0.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(x[ps].y), "v"(ab[ps].x), "v"(t0.y));
      bad[0][0] += __builtin_bit_cast(unsigned, t0.x) != __builtin_bit_cast(unsigned, r0);
      bad[0][1] += __builtin_bit_cast(unsigned, t0.y) != __builtin_bit_cast(unsigned, r1);
      bad[1][0] += __builtin_bit_cast(unsigned, x1.x) != __builtin_bit_cast(unsigned, q0);
      bad[1][1] += __builtin_bit_cast(unsigned, x1.y) != __builtin_bit_cast(unsigned, q1);
      bad[2][0] += __builtin_bit_cast(unsigned, t2.x) != __builtin_bit_cast(unsigned, r0);
      bad[2][1] += __builtin_bit_cast(unsigned, t2.y) != __builtin_bit_cast(unsigned, r1);
      if (trans) {  // the GELU tail of the real epilogue: packed ops feeding v_exp / v_rcp, whose results feed packed ops
        f32x2 z = x1 * (x1 * x1 * 0.1029432397f + 2.3022081985f);
        f32x2 e = {__builtin_amdgcn_exp2f(-z.x), __builtin_amdgcn_exp2f(-z.y)};
        e += 1.0f;
        f32x2 r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
        r *= x1;
        acc += r.x + r.y;
      } else
        acc += x1.x + x1.y + t2.x;
      u[ps].x += 1e-3f;  // the operands move: a stale result would not pass
      x[ps].y -= 1e-3f;
    }
  }
  for (int f = 0; f < 3; ++f)
    for (int h = 0; h < 2; ++h)
      if (bad[f][h]) atomicAdd(&counters[(f * 2 + h) * 4 + (lane >> 4)], bad[f][h]);
  if (acc == 123.456f) sink[0] = acc;
}

extern "C" int pk_victim_launch(const float* in, unsigned* counters, float* sink, int blocks, int iters, int trans, int lds_bytes, void* stream) {
  static bool set = false;
  if (!set) {
    if (hipFuncSetAttribute((const void*)pk_victim, hipFuncAttributeMaxDynamicSharedMemorySize, 133120) != hipSuccess) return -1;
    set = true;
  }
  hipLaunchKernelGGL(pk_victim, dim3(blocks), dim3(512), lds_bytes, (hipStream_t)stream, in, counters, sink, iters, trans);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
