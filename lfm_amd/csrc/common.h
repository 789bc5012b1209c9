// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of liblfm_hip.so.
// wave = 64 lanes; MFMA fragments follow the CDNA4 32x32x16 / 16x16x32 f16 maps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>

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
// Full-wave sum on the DPP cross-lane network (quad swaps, two row rotations, row_bcast15 / row_bcast31), total read back from lane 63
// through an SGPR: six VALU adds with DPP operands instead of six ds_bpermute round trips through the LDS (~100+ cycles each), which is
// what __shfl_xor compiles to.  Every lane receives the same value.  Summation order differs from wave_sum (a tree over lanes either way).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
  const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, m);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]
  v = dpp_add<0x124>(v);       // row_ror:4
  v = dpp_add<0x128>(v);       // row_ror:8   -> every lane of a 16-lane row holds the row total
  v = dpp_add<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3 -> lane 63 holds the wave total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// value held by lane ^ 32 (the other half-wave): one v_permlane32_swap instead of a ds_bpermute round trip through the LDS
__device__ __forceinline__ float xhalf(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r[0] = [lo|lo], r[1] = [hi|hi]
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }  // explicit v_rcp_f32 (no -ffast-math: a plain division is ~10 instructions)

// GELU(tanh) as torch: 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715 x^3)  ==  x * sigmoid(2u)  ==  x / (1 + 2^(-z)),
// z = 2u*log2(e) = x*(c0 + c1*x^2).  Two transcendentals (v_exp_f32, v_rcp_f32) and four plain VALU per element; saturates
// correctly without clamps (2^+inf -> rcp(inf) = 0, 2^-inf = 0 -> x).
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float c0 = 2.3022081985f, c1 = 0.1029432397f;  // 2*sqrt(2/pi)*log2(e), c0*0.044715
  const float z = x * (c0 + c1 * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-z));
}
// Two elements at a time on the packed-fp32 VALU (v_pk_mul / v_pk_fma / v_pk_add_f32: two lanes' worth of fp32 per issue slot).  The GELU
// epilogue of fc1 is VALU-bound (profiles/r02_epilogue_trace.txt: ~1900 of the ~2900 cycles per 32-row block); per pair this is 5 packed
// ops + 4 transcendentals instead of 12 plain + 4 (1900 -> 1500 cycles per block).  Measured and not kept: the 16 pairs of a read-back pass
// software-pipelined by hand so that every v_exp / v_rcp is followed by packed ops of other pairs (1550-1590 cycles, fc1 127.0 vs 125.5 us).  Same operation order per element as gelu_tanh_f.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_tanh_pk(f32x2_t x) {
  const f32x2_t c0 = {2.3022081985f, 2.3022081985f}, c1 = {0.1029432397f, 0.1029432397f}, one = {1.0f, 1.0f};
  f32x2_t z = x * __builtin_elementwise_fma(x * x, c1, c0);
#ifdef LFM_EXP_NOP_TRANS  // (experiment build) idle cycles between the packed chain and the transcendental that reads its result, and again before v_rcp
  asm volatile("s_nop 7" : "+v"(z));
#endif
  const f32x2_t e = {__builtin_amdgcn_exp2f(-z.x), __builtin_amdgcn_exp2f(-z.y)};
  f32x2_t d = one + e;
#ifdef LFM_EXP_NOP_TRANS
  asm volatile("s_nop 7" : "+v"(d));
#endif
  const f32x2_t r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  return x * r;
}

// a * b + c as ONE plain v_fma_f32 per element.  The row-affine epilogues of the folded LayerNorm path (gemm_kernel.h: a[m] * acc + (b[m] * u[n] + v[n])
// with (a, b) in a register pair) are written element by element with it INSTEAD of the vector expression `ab.x * acc + (ab.y * u + v)`, for which the
// compiler selects v_pk_fma_f32 with op_sel:[0,1,0] -- the LOW half of the packed operation takes src1 from the HIGH register of the pair.  On MI355X
// that form sporadically evaluates with the operand read as 0.0 in lanes 48-63 when a foreign wave shares the SIMD (two HIP streams in flight): the
// product term b * u vanished from ~500 elements per affected fc1 launch, 1-2 fp16 ulp in H (profiles/r05_cosched_root_cause.txt:
// operand dump of the epilogue solo vs co-scheduled; 6 of 11-19 co-scheduled evaluations differ with the packed form in four builds, 0 of 24 and
// 0 of the stress suite with scalar FMAs).  Scalar code stays scalar because the library is built with -fno-slp-vectorize (lfm_amd/_build.py);
// tests/test_host_logic.py::test_no_packed_fp32_op_sel disassembles the built code objects and keeps every op_sel'd packed-fp32 form out of the library.
// (Plain C, not inline asm: sixteen asm statements per pass pushed the one-wave-per-SIMD QKV kernel's accumulator view into scratch memory.)
__device__ __forceinline__ float fma_v(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// a * acc + (b * u + v), four columns of one row.  Packed again, but only in the SAFE form: a and b are first pinned into registers of their own (the empty
// asm), so that the compiler broadcasts each as the LOW register of a pair -- v_pk_fma_f32 .. op_sel_hi:[0,1,1], both halves read the low register -- and
// never reaches into the high register of the (a, b) pair for the low half (op_sel:[1,..]).  In round 4's events the op_sel_hi-broadcast FMA of the same
// chain (a * acc + t) never produced a wrong half; only the op_sel one did.  One v_mov + eight packed FMAs per eight outputs (sixteen scalar FMAs cost the
// VALU-bound fc1 / qkv epilogues ~2 us per launch).  LFM_EXP_AFFINE_SCALAR: the all-scalar form (A/B).
__device__ __forceinline__ f32x4 row_affine4(float a, float b, f32x4 acc, f32x4 u, f32x4 v) {
#ifdef LFM_EXP_AFFINE_SCALAR
  return (f32x4){fma_v(a, acc.x, fma_v(b, u.x, v.x)), fma_v(a, acc.y, fma_v(b, u.y, v.y)), fma_v(a, acc.z, fma_v(b, u.z, v.z)), fma_v(a, acc.w, fma_v(b, u.w, v.w))};
#else
  asm("" : "+v"(a));
  asm("" : "+v"(b));
  const f32x2 aa = {a, a}, bb = {b, b};
  const f32x2 t0 = __builtin_elementwise_fma(bb, (f32x2){u.x, u.y}, (f32x2){v.x, v.y}), t1 = __builtin_elementwise_fma(bb, (f32x2){u.z, u.w}, (f32x2){v.z, v.w});
  const f32x2 x0 = __builtin_elementwise_fma(aa, (f32x2){acc.x, acc.y}, t0), x1 = __builtin_elementwise_fma(aa, (f32x2){acc.z, acc.w}, t1);
  return (f32x4){x0.x, x0.y, x1.x, x1.y};
#endif
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// One-time set-up per (call site, DEVICE): the dynamic-LDS attribute of a kernel is per (function, device), a symbol address is per device -- and the library may be
// driven from several host threads (per-call settings are thread-scoped, ABI 4), or for several devices from one process.  A bit per device in an atomic mask:
// two threads racing through the first call both run the (idempotent) set-up, nobody skips it, nobody reads a half-written flag.  (Rounds 1-5: plain
// `static bool set` / `static unsigned long long` -- per process, unsynchronised: round-5 review, weak item 1.)
typedef std::atomic<unsigned long long> lfm_device_mask;
static inline unsigned long long lfm_device_bit() {
  int d = 0;
  (void)hipGetDevice(&d);
  return 1ull << (d & 63);
}
static inline bool lfm_device_todo(const lfm_device_mask& m, unsigned long long bit) { return !(m.load(std::memory_order_acquire) & bit); }
static inline void lfm_device_done(lfm_device_mask& m, unsigned long long bit) { m.fetch_or(bit, std::memory_order_release); }

// Zero a small device buffer with a KERNEL instead of hipMemsetAsync: inside a captured hipGraph a memset becomes a memset node, and
// graphs of the host-sequenced UNets (GroupNorm statistics are zeroed before every accumulation) produced NaN on the first replay
// after the device had gone idle -- the accumulation kernel ran against a scratch that was not zero yet.  Kernel nodes only.
static __global__ void lfm_zero_u32_kernel(unsigned int* __restrict__ p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
static inline int lfm_zero_async(void* p, size_t bytes, hipStream_t st) {  // bytes % 4 == 0
  const long n = (long)(bytes / 4);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(lfm_zero_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned int*)p, n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
