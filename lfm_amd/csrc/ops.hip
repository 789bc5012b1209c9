// Generic NHWC-fp16 building blocks behind the C ABI, used by the origin-ADM UNet (reference
// models/guided_diffusion/unet.py) whose layer sequence is data-dependent and therefore driven from the host side:
// implicit-GEMM 3x3 convolution (same / nearest-2x-upsampled input / stride 2), 1x1 convolution / linear with residual,
// general GroupNorm(32) with optional FiLM scale-shift and SiLU, channel concat, small-T attention, timestep embedding.
#include "../../include/lfm_hip.h"
#include "gemm_dispatch.h"
#include "conv_halo_kernel.h"

// ------------------------------------------------------------------ implicit-GEMM A source, NHWC fp16, 3x3, pad 1
// MODE 0: same size.  MODE 1: input is nearest-2x upsampled on the fly (Upsample, unet.py:73-100).
// MODE 2: stride 2 (Downsample, unet.py:103-128): output (oy,ox) reads input (2oy+dy-1, 2ox+dx-1).
template <int MODE>
struct ASrcConv {
  const half_t* in;
  const half_t* zeros;
  int H, W, Cin, M;  // OUTPUT spatial size; M = N*H*W
  int tap, ci0;
  int tap_begin, ci_begin;  // split-K: slice bz starts at k = bz * bs (init), a multiple of 64 <= Cin granularity
  __device__ __forceinline__ void init(int bz, long bs) {
    const long k = (long)bz * bs;
    tap_begin = (int)(k / Cin);
    ci_begin = (int)(k - (long)tap_begin * Cin);
  }
  struct Row {
    int n, y, x;
  };
  __device__ __forceinline__ Row row(int m) const {
    if (m >= M) m = M - 1;
    Row r;
    r.x = m % W;
    const int t = m / W;
    r.y = t % H;
    r.n = t / H;
    return r;
  }
  __device__ __forceinline__ void begin_tile(int kt, int bk) {
    if (kt == 0) {
      tap = tap_begin;
      ci0 = ci_begin;
    } else {
      ci0 += bk;
      if (ci0 >= Cin) {
        ci0 = 0;
        ++tap;
      }
    }
  }
  __device__ __forceinline__ const half_t* ptr(const Row& r, int koff) const {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    if (MODE == 2) {
      const int Hi = H * 2, Wi = W * 2;
      const int iy = 2 * r.y + dy, ix = 2 * r.x + dx;
      if ((unsigned)iy >= (unsigned)Hi || (unsigned)ix >= (unsigned)Wi) return zeros + koff;
      return in + (((long)r.n * Hi + iy) * Wi + ix) * Cin + ci0 + koff;
    }
    const int iy = r.y + dy, ix = r.x + dx;
    if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return zeros + koff;
    const int Hs = H >> (MODE == 1), Ws = W >> (MODE == 1);
    return in + (((long)r.n * Hs + (iy >> (MODE == 1))) * Ws + (ix >> (MODE == 1))) * Cin + ci0 + koff;
  }
};

struct EpiResidF16 {  // out = acc + bias (+ residual) -> fp16
  half_t* C;
  long ldc;
  const float* bias;
  const half_t* resid;
  struct Aux {
    f32x4 b;
    half4_t r;
  };
  __device__ __forceinline__ Aux load(int m, int n) const {
    Aux a;
    a.b = bias ? *(const f32x4*)(bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    a.r = resid ? *(const half4_t*)(resid + (long)m * ldc + n) : (half4_t){0, 0, 0, 0};
    return a;
  }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& a) const {
    v += a.b;
    half4_t h = {(half_t)(v.x + (float)a.r.x), (half_t)(v.y + (float)a.r.y), (half_t)(v.z + (float)a.r.z), (half_t)(v.w + (float)a.r.w)};
    *(half4_t*)(C + (long)m * ldc + n) = h;
  }
  __device__ __forceinline__ bool wide_ok() const { return (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 && (!resid || ((uintptr_t)resid & 15) == 0); }
  __device__ __forceinline__ void store8(int m, int n, f32x4 lo, f32x4 hi, const Aux& al, const Aux& ah) const {
    lo += al.b;
    hi += ah.b;
    half8_t h = {(half_t)(lo.x + (float)al.r.x), (half_t)(lo.y + (float)al.r.y), (half_t)(lo.z + (float)al.r.z), (half_t)(lo.w + (float)al.r.w),
                 (half_t)(hi.x + (float)ah.r.x), (half_t)(hi.y + (float)ah.r.y), (half_t)(hi.z + (float)ah.r.z), (half_t)(hi.w + (float)ah.r.w)};
    *(half8_t*)(C + (long)m * ldc + n) = h;
  }
};

struct EpiNCHWF32 {  // Cout <= 4 output conv: fp32 NCHW, channels beyond nch are padding
  float* out;
  const float* bias;  // [4]
  int HW, nch;
  typedef f32x4 Aux;
  __device__ __forceinline__ Aux load(int, int n) const { return *(const f32x4*)(bias + n); }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& b) const {
    if (n != 0) return;
    v += b;
    const int img = m / HW, pix = m - img * HW;
    float* o = out + (long)img * nch * HW + pix;
    o[0] = v.x;
    if (nch > 1) o[HW] = v.y;
    if (nch > 2) o[2 * HW] = v.z;
    if (nch > 3) o[3 * HW] = v.w;
  }
};

static __device__ half_t g_zero_page[64];  // zero padding rows for the conv gathers (zero-initialised device global)

static const half_t* zero_page() {  // the symbol's address is per DEVICE (one slot per device, written once; racing threads write the same value)
  static std::atomic<const half_t*> p[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  const half_t* v = p[dev & 63].load(std::memory_order_acquire);
  if (!v) {
    void* d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_zero_page)) != hipSuccess) return nullptr;
    v = (const half_t*)d;
    p[dev & 63].store(v, std::memory_order_release);
  }
  return v;
}

// Low-resolution levels of a UNet are small-M, huge-K problems (celeb512 at batch 32: 4x4 maps = 512 rows x 1024 columns x K 9216..18432):
// 32 tiles of 128x128 on 256 CUs, hundreds of K-tiles in sequence.  With a caller-provided workspace the K range is sliced over
// blockIdx.y (fp32 slabs, summed in a fixed order by a finish kernel that applies the real epilogue -- deterministic), as for the
// batch-1 DiT GEMMs.  workspace may be NULL (no split-K).
#define CONV_SPLITK_MAX_TILES 256  // 16x16 maps at batch 32 x 512 channels = 256 tiles of 128x128: two slices = two workgroups per CU
#define CONV_SPLITK_MAX_WG 512
extern "C" size_t lfm_conv3x3_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  const long M = (long)N * H * W;
  if (M <= 0 || Cout <= 0 || Cin <= 0) return 0;
  const long tiles = (long)cdiv(M, 128) * cdiv(Cout, 128);
  const int s256 = M < (1L << 30) ? splitk256_slices((int)M, Cout, 9 * Cin, (size_t)-1) : 0;  // slices on the 256x256 kernel (gemm_kernel.h), if the shape takes it
  int s = 1;
  if (tiles <= CONV_SPLITK_MAX_TILES)
    while (tiles * (s * 2) <= CONV_SPLITK_MAX_WG && (9L * Cin) / (s * 2) >= 128 && ((9L * Cin) / (s * 2)) % 64 == 0) s *= 2;
  if (s256 > s) s = s256;
  return s < 2 ? 0 : (size_t)s * M * Cout * 4;
}

template <int MODE>
static int conv3x3_mode(const half_t* xi, const half_t* z, const half_t* wi, const EpiResidF16& epi, int H, int W, int Cin, int Cout, int M, float* ws,
                        size_t ws_bytes, hipStream_t st) {
  ASrcConv<MODE> a{xi, z, H, W, Cin, M, 0, 0, 0, 0};
  const int rc = launch_gemm_splitk_src(a, wi, 9L * Cin, M, Cout, 9 * Cin, epi, ws, ws_bytes, st, CONV_SPLITK_MAX_TILES, CONV_SPLITK_MAX_WG);
  if (rc != 1) return rc;
  return launch_gemm_auto(a, wi, 9L * Cin, M, Cout, 9 * Cin, epi, st);
}

extern "C" int lfm_conv3x3_f16_ws(const void* in, const void* w, const float* bias, const void* resid, void* out, int N, int H, int W, int Cin,
                                  int Cout, int mode, void* workspace, size_t workspace_bytes, lfm_stream_t stream) {
  if (!in || !w || !out) return LFM_ERR_ARG;
  if (N <= 0 || H <= 0 || W <= 0 || Cin % 64 || Cout % 4 || mode < 0 || mode > 2) return LFM_ERR_SHAPE;
  if (mode == 1 && ((H | W) & 1)) return LFM_ERR_SHAPE;
  if (workspace && ((uintptr_t)workspace & 15)) return LFM_ERR_ALIGN;
  const half_t* z = zero_page();
  if (!z) return LFM_ERR_LAUNCH;
  const int M = N * H * W;
  EpiResidF16 epi{(half_t*)out, Cout, bias, (const half_t*)resid};
  hipStream_t st = (hipStream_t)stream;
  const half_t* wi = (const half_t*)w;
  const half_t* xi = (const half_t*)in;
  float* ws = (float*)workspace;
  // plain and upsample-fused 3x3 convolutions on 16-aligned maps that fill the chip: the halo-tiled direct kernel (conv_halo_kernel.h);
  // flag 8388608: the implicit GEMM instead (A/B and parity tests)
  if (mode != 2 && lfm_gemm_selected() == 0 && !(lfm_gemm_debug_flags() & 8388608) && !(((uintptr_t)out | (uintptr_t)resid) & 15)) {
    const int rc = mode == 1 ? launch_conv3x3_halo<1>(xi, z, wi, N, H, W, Cin, Cout, epi, st) : launch_conv3x3_halo<0>(xi, z, wi, N, H, W, Cin, Cout, epi, st);
    if (rc != 1) return rc;
  }
  if (mode == 0) return conv3x3_mode<0>(xi, z, wi, epi, H, W, Cin, Cout, M, ws, workspace_bytes, st);
  if (mode == 1) return conv3x3_mode<1>(xi, z, wi, epi, H, W, Cin, Cout, M, ws, workspace_bytes, st);
  return conv3x3_mode<2>(xi, z, wi, epi, H, W, Cin, Cout, M, ws, workspace_bytes, st);
}

extern "C" int lfm_conv3x3_f16(const void* in, const void* w, const float* bias, const void* resid, void* out, int N, int H, int W, int Cin,
                               int Cout, int mode, lfm_stream_t stream) {
  return lfm_conv3x3_f16_ws(in, w, bias, resid, out, N, H, W, Cin, Cout, mode, nullptr, 0, stream);
}

extern "C" int lfm_conv3x3_out_f32(const void* in, const void* w4, const float* bias4, float* out_nchw, int N, int H, int W, int Cin, int nch,
                                   lfm_stream_t stream) {
  if (!in || !w4 || !bias4 || !out_nchw) return LFM_ERR_ARG;
  if (N <= 0 || Cin % 64 || nch < 1 || nch > 4) return LFM_ERR_SHAPE;
  const half_t* z = zero_page();
  if (!z) return LFM_ERR_LAUNCH;
  const int M = N * H * W;
  const EpiNCHWF32 eo{out_nchw, bias4, H * W, nch};
  if (lfm_gemm_selected() == 0 && !(lfm_gemm_debug_flags() & 8388608)) {  // flag 8388608: the implicit GEMM (A/B)
    const int rc = launch_conv3x3_halo_out((const half_t*)in, z, (const half_t*)w4, N, H, W, Cin, eo, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  return launch_gemm_tn(ASrcConv<0>{(const half_t*)in, z, H, W, Cin, M, 0, 0, 0, 0}, (const half_t*)w4, 9L * Cin, M, 4, 9 * Cin, eo, (hipStream_t)stream);
}

extern "C" int lfm_linear_f16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias,
                              const void* resid, lfm_stream_t stream) {
  if (!A || !W || !C) return LFM_ERR_ARG;
  if ((lda % 8) || ((uintptr_t)A & 15)) return LFM_ERR_ALIGN;
  return launch_gemm_auto(ASrcRowMajor{(const half_t*)A, lda, M, 0}, (const half_t*)W, ldw, M, N, K,
                          EpiResidF16{(half_t*)C, ldc, bias, (const half_t*)resid}, (hipStream_t)stream);
}

// C = [A1 | A2] W^T + bias (+ resid): the 1x1 skip convolution of a ResBlock whose input is the channel concat of two tensors (unet.py:649 + :236),
// read in place: the K range [0, K1) comes from A1 [M, K1], [K1, K1 + K2) from A2 [M, K2].  K1 % 64 == 0 (a K-tile never straddles the seam).
extern "C" int lfm_linear2_f16(const void* A1, int K1, const void* A2, int K2, const void* W, long ldw, void* C, long ldc, int M, int N,
                               const float* bias, const void* resid, lfm_stream_t stream) {
  if (!A1 || !A2 || !W || !C) return LFM_ERR_ARG;
  if (K1 <= 0 || K2 <= 0 || (K1 % 64) || (K2 % 8)) return LFM_ERR_SHAPE;
  if (((uintptr_t)A1 | (uintptr_t)A2) & 15) return LFM_ERR_ALIGN;
  return launch_gemm_auto(ASrcRowMajor2{(const half_t*)A1, (const half_t*)A2, K1, K2, M, 0}, (const half_t*)W, ldw, M, N, K1 + K2,
                          EpiResidF16{(half_t*)C, ldc, bias, (const half_t*)resid}, (hipStream_t)stream);
}

// ------------------------------------------------------------------ first conv: fp32 NCHW (Cin <= 16) -> fp16 NHWC
// Weights are transposed into LDS as [k = (c,ky,kx)][Cout] so the 8 output channels of a thread are two float4 reads per tap;
// a block walks CI_PIX pixels (threads = Cout/8 channel-octets x pixel rows), input taps are L1-served broadcast loads.
#define CI_PIX 256
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                      half_t* __restrict__ out, int N, int H, int W, int Cin, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [Cin*9][Cout]
  const int KK = Cin * 9, c8n = Cout / 8;
  for (int e = threadIdx.x; e < KK * Cout; e += 256) {
    const int co = e / KK, k = e - co * KK;  // w is [Cout][Cin][3][3] = [Cout][KK]
    wl[k * Cout + co] = w[e];
  }
  __syncthreads();
  const int oct = threadIdx.x % c8n, prow = threadIdx.x / c8n, rows = 256 / c8n;
  if (prow >= rows) return;
  const int co = oct * 8;
  const long total = (long)N * H * W;
  const long p0 = (long)blockIdx.x * CI_PIX;
  for (long pix = p0 + prow; pix < p0 + CI_PIX && pix < total; pix += rows) {
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H), n = (int)(pix / ((long)H * W));
    f32x4 a0 = *(const f32x4*)(b + co), a1 = *(const f32x4*)(b + co + 4);
    for (int c = 0; c < Cin; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = yy + t / 3 - 1, ix = xx + t % 3 - 1;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
        const float v = x[(((long)n * Cin + c) * H + iy) * W + ix];
        const float* wr = wl + (c * 9 + t) * Cout + co;
        a0 += v * *(const f32x4*)wr;
        a1 += v * *(const f32x4*)(wr + 4);
      }
    half8_t h = {(half_t)a0.x, (half_t)a0.y, (half_t)a0.z, (half_t)a0.w, (half_t)a1.x, (half_t)a1.y, (half_t)a1.z, (half_t)a1.w};
    *(half8_t*)(out + pix * Cout + co) = h;
  }
}

// The same on the matrix cores (Cout % 64 == 0): the scalar kernel above runs at 16 TFLOP/s and had become 1.5 % of a celeb512 UNet evaluation.
// A workgroup walks chunks of 64 pixels: its 256 threads gather the im2col rows (k = c * 9 + tap, zero beyond the border and beyond Cin * 9) once into
// the LDS, split hi / lo into two fp16 planes (x = hi + lo to 2^-22: three MFMAs -- w_hi x_hi + w_hi x_lo + w_lo x_hi -- are the fp32 dot product to
// ~2^-21, the fp32 arithmetic of the scalar kernel and of the reference's conv at TF32-or-better); the fp32 weights are split the same way once per
// workgroup.  Wave w owns output channels [w Cout / 4, (w + 1) Cout / 4): D[channel][pixel] on v_mfma_f32_16x16x16_f16 with the weights as the A
// operand, so a lane ends up with four consecutive channels of one pixel: 8-byte NHWC stores.  Bound: the fp16 output write.
#define CIM_PIX 64
template <int NT>  // 16-channel tiles per wave = Cout / 64
__global__ __launch_bounds__(256) void conv_in_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                           half_t* __restrict__ out, int N, int H, int W, int Cin, int Cout, int KS) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int Kp = KS * 16, LD = Kp + 8, KK = Cin * 9;  // LDS row stride (halves): + 8 keeps 16 consecutive rows on distinct banks for the 8-byte reads
  half_t* wh = (half_t*)smraw;                  // [Cout][LD]
  half_t* wl = wh + (size_t)Cout * LD;
  half_t* xh = wl + (size_t)Cout * LD;          // [CIM_PIX][LD]
  half_t* xl = xh + CIM_PIX * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  for (int e = tid; e < Cout * Kp; e += 256) {
    const int co = e / Kp, k = e - co * Kp;
    const float v = k < KK ? w[(long)co * KK + k] : 0.f;
    const half_t h = (half_t)v;
    wh[co * LD + k] = h;
    wl[co * LD + k] = (half_t)(v - (float)h);
  }
  const long total = (long)N * H * W;
  const int c0 = wave * NT * 16;
  f32x4 bias4[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias4[nt] = *(const f32x4*)(b + c0 + nt * 16 + q * 4);
  for (long p0 = (long)blockIdx.x * CIM_PIX; p0 < total; p0 += (long)gridDim.x * CIM_PIX) {
    __syncthreads();  // the previous chunk's fragment reads are done (and, first time round, the weights are in place)
    {
      const long pix = p0 + (tid & 63);
      const bool live = pix < total;
      const int xx = live ? (int)(pix % W) : 0, yy = live ? (int)((pix / W) % H) : 0, n = live ? (int)(pix / ((long)H * W)) : 0;
      for (int k = tid >> 6; k < Kp; k += 4) {
        float v = 0.f;
        if (live && k < KK) {
          const int c = k / 9, t = k - c * 9, iy = yy + t / 3 - 1, ix = xx + t % 3 - 1;
          if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[(((long)n * Cin + c) * H + iy) * W + ix];
        }
        const half_t h = (half_t)v;
        xh[(tid & 63) * LD + k] = h;
        xl[(tid & 63) * LD + k] = (half_t)(v - (float)h);
      }
    }
    __syncthreads();
#pragma unroll
    for (int pt = 0; pt < CIM_PIX / 16; ++pt) {
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = bias4[nt];
      for (int s = 0; s < KS; ++s) {
        const half4_t bh = *(const half4_t*)(xh + (pt * 16 + r) * LD + s * 16 + q * 4), bl = *(const half4_t*)(xl + (pt * 16 + r) * LD + s * 16 + q * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const half4_t ah = *(const half4_t*)(wh + (c0 + nt * 16 + r) * LD + s * 16 + q * 4), al = *(const half4_t*)(wl + (c0 + nt * 16 + r) * LD + s * 16 + q * 4);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, acc[nt], 0, 0, 0);
        }
      }
      const long pix = p0 + pt * 16 + r;  // lane holds channels c0 + 16 nt + 4 q .. + 3 of this pixel
      if (pix < total) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          *(half4_t*)(out + pix * Cout + c0 + nt * 16 + q * 4) = (half4_t){(half_t)acc[nt].x, (half_t)acc[nt].y, (half_t)acc[nt].z, (half_t)acc[nt].w};
      }
    }
  }
}

template <int NT>
static int launch_conv_in_mfma(const float* x, const float* w, const float* b, half_t* out, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
  const int KS = cdiv(Cin * 9, 16), LD = KS * 16 + 8;
  const size_t lds = ((size_t)2 * Cout + 2 * CIM_PIX) * LD * 2;
  if (lds > 160 * 1024) return 1;
  static unsigned long long attr_set = 0;
  int devid = 0;
  (void)hipGetDevice(&devid);
  if (!((attr_set >> (devid & 63)) & 1)) {
    if (hipFuncSetAttribute((const void*)conv_in_mfma_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return LFM_ERR_LAUNCH;
    attr_set |= 1ull << (devid & 63);
  }
  const long chunks = cdiv((long)N * H * W, CIM_PIX);
  const int grid = (int)(chunks < 1024 ? chunks : 1024);  // a workgroup pays the weight split once and then walks its chunks
  hipLaunchKernelGGL(conv_in_mfma_kernel<NT>, dim3(grid), dim3(256), lds, st, x, w, b, out, N, H, W, Cin, Cout, KS);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

extern "C" int lfm_conv3x3_in_f32(const float* x_nchw, const float* w, const float* bias, void* out_nhwc, int N, int H, int W, int Cin,
                                  int Cout, lfm_stream_t stream) {
  if (!x_nchw || !w || !bias || !out_nhwc) return LFM_ERR_ARG;
  if (N <= 0 || Cin <= 0 || Cin > 16 || Cout % 8 || Cout / 8 > 256) return LFM_ERR_SHAPE;
  if ((Cout == 64 || Cout == 128 || Cout == 192 || Cout == 256) && !((uintptr_t)out_nhwc & 7) && !((uintptr_t)bias & 15) &&
      !(lfm_gemm_debug_flags() & 1)) {  // flag 1: the scalar kernel (A/B)
    int rc;
    if (Cout == 64) rc = launch_conv_in_mfma<1>(x_nchw, w, bias, (half_t*)out_nhwc, N, H, W, Cin, Cout, (hipStream_t)stream);
    else if (Cout == 128) rc = launch_conv_in_mfma<2>(x_nchw, w, bias, (half_t*)out_nhwc, N, H, W, Cin, Cout, (hipStream_t)stream);
    else if (Cout == 192) rc = launch_conv_in_mfma<3>(x_nchw, w, bias, (half_t*)out_nhwc, N, H, W, Cin, Cout, (hipStream_t)stream);
    else rc = launch_conv_in_mfma<4>(x_nchw, w, bias, (half_t*)out_nhwc, N, H, W, Cin, Cout, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  const size_t lds = (size_t)Cin * 9 * Cout * 4;
  if (lds > 160 * 1024) return LFM_ERR_SHAPE;
  static lfm_device_mask set{0};
  const unsigned long long dbit = lfm_device_bit();
  if (lfm_device_todo(set, dbit)) {
    if (hipFuncSetAttribute((const void*)conv_in_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return LFM_ERR_LAUNCH;
    lfm_device_done(set, dbit);
  }
  hipLaunchKernelGGL(conv_in_kernel, dim3(cdiv((long)N * H * W, CI_PIX)), dim3(256), lds, (hipStream_t)stream, x_nchw, w, bias,
                     (half_t*)out_nhwc, N, H, W, Cin, Cout);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

// ------------------------------------------------------------------ general GroupNorm(32 groups) on NHWC fp16
// 1) stats: one block per (image, pixel-slab) [fast path] or (group, image, pixel-slab) -> per-block PARTIAL {sum, sumsq} slots
// 2) coef : per (n, c): folds the partials of its group in a FIXED order (deterministic: the reference is; atomics are not), then
//           a = rstd*gamma*(1+scale),  b = (beta - mean*rstd*gamma)*(1+scale) + shift     (FiLM optional)
// 3) apply: y = silu?(x*a + b), 8 channels per thread
// GroupNorm input = the channel concat [a | b] of two NHWC tensors read in place (th.cat([h, hs.pop()], dim=1) feeding a ResBlock's first
// GroupNorm, unet.py:649 + :171: the concatenated tensor is never materialised); b == nullptr: a alone (Ca == C).  Ca % 8 == 0.
struct GnIn {
  const half_t* a;
  const half_t* b;
  int Ca, Cb;
  __device__ __forceinline__ const half_t* at(long pix, int c) const { return c < Ca ? a + pix * Ca + c : b + pix * Cb + (c - Ca); }
  // a thread that owns channel c of every pixel of image n: first pixel's address and the row stride of the tensor that holds c
  __device__ __forceinline__ const half_t* column(long pix0, int c, long& stride) const {
    const bool fa = c < Ca;
    stride = fa ? Ca : Cb;
    return fa ? a + pix0 * Ca + c : b + pix0 * Cb + (c - Ca);
  }
};

template <int VEC>
__global__ __launch_bounds__(256) void gn_stats_general_kernel(GnIn in, float* __restrict__ part, int HW, int C, int cpg,
                                                               int pix_per_block, int G) {
  const int g = blockIdx.x, n = blockIdx.y;
  const int p0 = blockIdx.z * pix_per_block, p1 = min(p0 + pix_per_block, HW);
  const int vpp = cpg / VEC;  // vectors per pixel in this group
  const long total = (long)(p1 - p0) * vpp;
  float s = 0.f, q = 0.f;
  for (long e = threadIdx.x; e < total; e += 256) {
    const int p = p0 + (int)(e / vpp), v = (int)(e % vpp);
    const half_t* ptr = in.at((long)n * HW + p, g * cpg + v * VEC);
    if (VEC == 4) {
      const half4_t h = *(const half4_t*)ptr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float f = (float)h[j];
        s += f;
        q += f * f;
      }
    } else {
      const float f = (float)ptr[0];
      s += f;
      q += f * f;
    }
  }
  __shared__ float rs[4], rq[4];
  s = wave_sum(s);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) {
    rs[threadIdx.x >> 6] = s;
    rq[threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // slot [n][slab][g]
    float* o = part + (((long)n * gridDim.z + blockIdx.z) * G + g) * 2;
    o[0] = rs[0] + rs[1] + rs[2] + rs[3];
    o[1] = rq[0] + rq[1] + rq[2] + rq[3];
  }
}

// fast path (cpg % 4 == 0, C/8 <= 256): a block reads a slab of pixels with FULL rows (coalesced); thread = channel octet x pixel
// row; per half-octet partial sums are folded through LDS (fixed order) and leave as slot [n][slab][half-octet] of the partial buffer.
__global__ __launch_bounds__(256) void gn_stats_rows_kernel(GnIn in, float* __restrict__ part, int HW, int C, int pix_per_block) {
  __shared__ float red[4][256];
  const int n = blockIdx.y, c8n = C / 8, tid = threadIdx.x;
  const int rows = 256 / c8n;
  const int oct = tid % c8n, prow = tid / c8n;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, HW);
  float s[2] = {0.f, 0.f}, q[2] = {0.f, 0.f};
  if (prow < rows) {
    long XS;  // row stride of the tensor that holds this thread's octet
    const half_t* base = in.column((long)n * HW, oct * 8, XS);
    auto add = [&](const half8_t& v) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[j];
        s[j >> 2] += f;
        q[j >> 2] += f * f;
      }
    };
    int p = p0 + prow;
    for (; p + 3 * rows < p1; p += 4 * rows) {  // four independent loads in flight (one per iteration ran at 1.5 TB/s), summed in pixel order
      const half8_t v0 = *(const half8_t*)(base + (long)p * XS), v1 = *(const half8_t*)(base + (long)(p + rows) * XS);
      const half8_t v2 = *(const half8_t*)(base + (long)(p + 2 * rows) * XS), v3 = *(const half8_t*)(base + (long)(p + 3 * rows) * XS);
      add(v0);
      add(v1);
      add(v2);
      add(v3);
    }
    for (; p < p1; p += rows) add(*(const half8_t*)(base + (long)p * XS));
  }
  red[0][tid] = s[0];
  red[1][tid] = s[1];
  red[2][tid] = q[0];
  red[3][tid] = q[1];
  __syncthreads();
  if (tid < c8n) {
    for (int r = 1; r < rows; ++r) {
      s[0] += red[0][tid + r * c8n];
      s[1] += red[1][tid + r * c8n];
      q[0] += red[2][tid + r * c8n];
      q[1] += red[3][tid + r * c8n];
    }
    float* o = part + (((long)n * gridDim.x + blockIdx.x) * (C / 4) + tid * 2) * 2;
    o[0] = s[0];
    o[1] = q[0];
    o[2] = s[1];
    o[3] = q[1];
  }
}

// rows != 0: partials are [n][slab][C/4 half-octets] (fast path);  rows == 0: [n][slab][G]
__global__ void gn_coef_kernel(const float* __restrict__ part, int slabs, int rows, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ film, long film_stride, float* __restrict__ ab, int N, int C, int cpg, float cnt,
                               float eps, int G) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C, g = c / cpg;
  float sum = 0.f, sq = 0.f;  // every channel of a group folds the same slots in the same order: identical, deterministic statistics
  if (rows) {
    const int h0 = g * (cpg / 4), h1 = h0 + cpg / 4, Q = C / 4;
#pragma unroll 4
    for (int b = 0; b < slabs; ++b) {
      const float* p = part + ((long)n * slabs + b) * Q * 2;
      for (int h = h0; h < h1; ++h) {
        sum += p[2 * h];
        sq += p[2 * h + 1];
      }
    }
  } else {
    for (int b = 0; b < slabs; ++b) {
      const float* p = part + (((long)n * slabs + b) * G + g) * 2;
      sum += p[0];
      sq += p[1];
    }
  }
  const float mean = sum / cnt;
  const float var = fmaxf(sq / cnt - mean * mean, 0.f);
  const float rstd = rsqrtf(var + eps);
  float a = rstd * gamma[c], b = beta[c] - mean * rstd * gamma[c];
  if (film) {  // h = norm(h) * (1 + scale) + shift   (unet.py:229-232; film row = [scale(C) | shift(C)])
    const float sc = 1.0f + film[(long)n * film_stride + c], sh = film[(long)n * film_stride + C + c];
    a *= sc;
    b = b * sc + sh;
  }
  ab[(long)i * 2] = a;
  ab[(long)i * 2 + 1] = b;
}

template <bool SILU>
__global__ __launch_bounds__(256) void gn_affine_kernel(GnIn in, half_t* __restrict__ y, const float* __restrict__ ab,
                                                        int HW, int C, long total8) {
  // grid (chunks of one image / 256, images): 32-bit index arithmetic only (HW * C / 8 < 2^31 per image)
  const int c8n = C / 8, n = blockIdx.y;
  const unsigned j = blockIdx.x * 256u + threadIdx.x;
  if (j >= (unsigned)HW * (unsigned)c8n) return;
  const unsigned pix = j / (unsigned)c8n;
  const int c0 = (int)(j - pix * (unsigned)c8n) * 8;
  const long i = (long)n * HW * c8n + j;
  (void)total8;
  const half8_t v = *(const half8_t*)in.at((long)n * HW + pix, c0);
  const f32x4* p = (const f32x4*)(ab + ((long)n * C + c0) * 2);
  half8_t o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 t = p[j];  // a0 b0 a1 b1
    float f0 = (float)v[2 * j] * t.x + t.y, f1 = (float)v[2 * j + 1] * t.z + t.w;
    if (SILU) {
      f0 = silu_f(f0);
      f1 = silu_f(f1);
    }
    o[2 * j] = (half_t)f0;
    o[2 * j + 1] = (half_t)f1;
  }
  ((half8_t*)y)[i] = o;
}
// The same apply for channel counts whose octets divide the block (256 % (C / 8) == 0: 256 / 512 / 1024 / 2048 channels, every large map of the
// UNets): a thread owns ONE channel octet -- its 16 coefficients stay in registers -- and walks the pixels of its block's slab, four 16-byte
// chunks in flight; no per-chunk 64-bit div / mod, no 64 bytes of coefficient loads per 16 bytes of data (round 4; same arithmetic per element).
template <bool SILU>
__global__ __launch_bounds__(256) void gn_affine_rows_kernel(GnIn in, half_t* __restrict__ y, const float* __restrict__ ab, int HW, int C,
                                                             int pix_per_block) {
  const int n = blockIdx.y, c8n = C / 8, tid = threadIdx.x;
  const int oct = tid % c8n, prow = tid / c8n, pstride = 256 / c8n;
  const f32x4* cp = (const f32x4*)(ab + ((long)n * C + oct * 8) * 2);
  f32x4 t[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = cp[j];  // a0 b0 a1 b1
  long XS;
  const half_t* xb = in.column((long)n * HW, oct * 8, XS);
  half_t* yb = y + ((long)n * HW) * C + oct * 8;
  auto one = [&](const half8_t v) {
    half8_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f0 = (float)v[2 * j] * t[j].x + t[j].y, f1 = (float)v[2 * j + 1] * t[j].z + t[j].w;
      if (SILU) {
        f0 = silu_f(f0);
        f1 = silu_f(f1);
      }
      o[2 * j] = (half_t)f0;
      o[2 * j + 1] = (half_t)f1;
    }
    return o;
  };
  const int p0 = blockIdx.x * pix_per_block, p1 = min(p0 + pix_per_block, HW);
  int p = p0 + prow;
  for (; p + 3 * pstride < p1; p += 4 * pstride) {
    half8_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const half8_t*)(xb + (long)(p + u * pstride) * XS);
#pragma unroll
    for (int u = 0; u < 4; ++u) *(half8_t*)(yb + (long)(p + u * pstride) * C) = one(v[u]);
  }
  for (; p < p1; p += pstride) *(half8_t*)(yb + (long)p * C) = one(*(const half8_t*)(xb + (long)p * XS));
}

// Fused small-tensor path (one launch instead of three): one block per (image, chunk of GPB groups) streams its channels of every pixel twice
// (the second pass hits the L2): pass 1 per-group {sum, sumsq} with a fixed-order block reduction, pass 2 y = silu?(x*a + b) with the
// per-channel a, b (FiLM folded) held in registers.  cpg % 8 == 0, HW <= 1024: the GroupNorms of the UNets' lower resolutions, where the
// three-kernel path is launch-bound.
template <bool SILU>
__global__ __launch_bounds__(256) void gn_fused_kernel(GnIn in, half_t* __restrict__ y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ film, long film_stride, int HW,
                                                       int C, int cpg, int gpb, float eps) {
  __shared__ float red[2][256];
  __shared__ float mr[2][32];
  const int n = blockIdx.y, tid = threadIdx.x;
  const int CW = gpb * cpg, o8 = CW >> 3;          // channels / octets of this block
  const int oct = tid % o8, prow = tid / o8, rows = 256 / o8;
  const int c0 = blockIdx.x * CW + oct * 8;         // first channel of this thread's octet
  long XS;  // row stride of the tensor that holds this thread's octet
  const half_t* xb = in.column((long)n * HW, c0, XS);
  float s = 0.f, q = 0.f;
  if (prow < rows) {
    auto add = [&](const half8_t& v) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[j];
        s += f;
        q += f * f;
      }
    };
    int p = prow;
    for (; p + 3 * rows < HW; p += 4 * rows) {  // four loads in flight per thread (the block is alone on its CU: latency, not bandwidth, bound it)
      const half8_t v0 = *(const half8_t*)(xb + (long)p * XS), v1 = *(const half8_t*)(xb + (long)(p + rows) * XS);
      const half8_t v2 = *(const half8_t*)(xb + (long)(p + 2 * rows) * XS), v3 = *(const half8_t*)(xb + (long)(p + 3 * rows) * XS);
      add(v0);
      add(v1);
      add(v2);
      add(v3);
    }
    for (; p < HW; p += rows) add(*(const half8_t*)(xb + (long)p * XS));
  }
  red[0][tid] = s;
  red[1][tid] = q;
  __syncthreads();
  if (tid < gpb) {  // group tid of this block: octets [tid*cpg/8, (tid+1)*cpg/8) of every pixel row, folded in a fixed order
    const int ob = tid * (cpg >> 3), oe = ob + (cpg >> 3);
    float ss = 0.f, qq = 0.f;
    for (int r = 0; r < rows; ++r)
      for (int o = ob; o < oe; ++o) {
        ss += red[0][r * o8 + o];
        qq += red[1][r * o8 + o];
      }
    const float cnt = (float)HW * (float)cpg, mean = ss / cnt;
    mr[0][tid] = mean;
    mr[1][tid] = rsqrtf(fmaxf(qq / cnt - mean * mean, 0.f) + eps);
  }
  __syncthreads();
  if (prow >= rows) return;
  const int gl = (oct * 8) / cpg;
  const float mean = mr[0][gl], rstd = mr[1][gl];
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    a[j] = rstd * gamma[c];
    b[j] = beta[c] - mean * a[j];
    if (film) {
      const float sc = 1.0f + film[(long)n * film_stride + c], sh = film[(long)n * film_stride + C + c];
      a[j] *= sc;
      b[j] = b[j] * sc + sh;
    }
  }
  half_t* yb = y + (long)n * HW * C + c0;
  auto apply = [&](const half8_t& v, int p) {
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = (float)v[j] * a[j] + b[j];
      if (SILU) f = silu_f(f);
      o[j] = (half_t)f;
    }
    *(half8_t*)(yb + (long)p * C) = o;
  };
  int p = prow;
  for (; p + 3 * rows < HW; p += 4 * rows) {
    const half8_t v0 = *(const half8_t*)(xb + (long)p * XS), v1 = *(const half8_t*)(xb + (long)(p + rows) * XS);
    const half8_t v2 = *(const half8_t*)(xb + (long)(p + 2 * rows) * XS), v3 = *(const half8_t*)(xb + (long)(p + 3 * rows) * XS);
    apply(v0, p);
    apply(v1, p + rows);
    apply(v2, p + 2 * rows);
    apply(v3, p + 3 * rows);
  }
  for (; p < HW; p += rows) apply(*(const half8_t*)(xb + (long)p * XS), p);
}

#define GN_MAX_SLABS 64  // pixel slabs per image: bounds the partial buffer independently of HW
static inline size_t gn_part_bytes(int N, int C) {
  const size_t per_slab = (size_t)(C / 4 > 32 ? C / 4 : 32) * 2 * 4;  // max of the two partial layouts
  return (((size_t)N * GN_MAX_SLABS * per_slab) + 255) / 256 * 256;
}
extern "C" size_t lfm_groupnorm_scratch_bytes(int N, int C) { return gn_part_bytes(N, C) + (size_t)N * C * 8 + 256; }

static int groupnorm_impl(const GnIn& in, void* y, const float* gamma, const float* beta, const float* film, long film_stride, void* scratch, int N, int HW,
                          int C, int groups, float eps, int silu, lfm_stream_t stream) {
  if (!in.a || !y || !gamma || !beta || !scratch) return LFM_ERR_ARG;
  if (N <= 0 || HW <= 0 || groups <= 0 || groups > 32 || C % groups || C % 8) return LFM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int G = groups, cpg = C / G;
  // (measured on the celeb512 UNet: at 64x64 maps the fused kernel's 256 blocks are too few -- 300 us vs ~35 us for the three kernels)
  // fused path: one block's 256 threads cover the gpb * cpg / 8 eight-channel columns of its groups, so a single group may be at most 2048
  // channels wide (wider groups -- e.g. groups = 1 on 4096 channels -- take the three-kernel path below instead of shrinking gpb to zero)
  if (cpg % 8 == 0 && cpg <= 2048 && HW <= 1024 && !(lfm_gemm_debug_flags() & 16384)) {  // flag 16384: the three-kernel path (A/B)
    int gpb = 1;
    while (gpb * 2 <= G && G % (gpb * 2) == 0 && (long)N * (G / (gpb * 2)) >= 256 && gpb * 2 * cpg <= 2048) gpb *= 2;
    dim3 grid(G / gpb, N);
    if (silu) hipLaunchKernelGGL(gn_fused_kernel<true>, grid, dim3(256), 0, st, in, (half_t*)y, gamma, beta, film, film_stride, HW, C, cpg, gpb, eps);
    else hipLaunchKernelGGL(gn_fused_kernel<false>, grid, dim3(256), 0, st, in, (half_t*)y, gamma, beta, film, film_stride, HW, C, cpg, gpb, eps);
    LFM_CHECK_LAUNCH();
    return LFM_OK;
  }
  float* part = (float*)scratch;
  float* ab = (float*)((char*)scratch + gn_part_bytes(N, C));
  int slabs, rows;
  if (cpg % 4 == 0 && C / 8 <= 256) {
    int ppb = HW >= 4096 ? 256 : (HW >= 256 ? 64 : HW);  // (512 pixels per block at 64x64 maps left 256 blocks: one per CU; 128 made gn_coef's serial slab walk the longer kernel)
    if (cdiv(HW, ppb) > GN_MAX_SLABS) ppb = cdiv(HW, GN_MAX_SLABS);
    slabs = cdiv(HW, ppb);
    rows = 1;
    hipLaunchKernelGGL(gn_stats_rows_kernel, dim3(slabs, N), dim3(256), 0, st, in, part, HW, C, ppb);
  } else {
    int ppb = 2048;
    if (cdiv(HW, ppb) > GN_MAX_SLABS) ppb = cdiv(HW, GN_MAX_SLABS);
    slabs = cdiv(HW, ppb);
    rows = 0;
    dim3 grid(G, N, slabs);
    if (cpg % 4 == 0) hipLaunchKernelGGL(gn_stats_general_kernel<4>, grid, dim3(256), 0, st, in, part, HW, C, cpg, ppb, G);
    else hipLaunchKernelGGL(gn_stats_general_kernel<1>, grid, dim3(256), 0, st, in, part, HW, C, cpg, ppb, G);
  }
  LFM_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_coef_kernel, dim3(cdiv((long)N * C, 256)), dim3(256), 0, st, part, slabs, rows, gamma, beta, film, film_stride, ab, N, C,
                     cpg, (float)HW * (float)cpg, eps, G);
  LFM_CHECK_LAUNCH();
  if (C / 8 <= 256 && 256 % (C / 8) == 0) {  // a thread per channel octet
    const int pstride = 256 / (C / 8);
    int app = HW >= 4096 ? 256 : (HW >= 256 ? 64 : HW);
    if (app < 4 * pstride) app = 4 * pstride < HW ? 4 * pstride : HW;
    if (silu) hipLaunchKernelGGL(gn_affine_rows_kernel<true>, dim3(cdiv(HW, app), N), dim3(256), 0, st, in, (half_t*)y, ab, HW, C, app);
    else hipLaunchKernelGGL(gn_affine_rows_kernel<false>, dim3(cdiv(HW, app), N), dim3(256), 0, st, in, (half_t*)y, ab, HW, C, app);
    LFM_CHECK_LAUNCH();
    return LFM_OK;
  }
  const long total8 = (long)N * HW * C / 8;
  if ((long)HW * (C / 8) >= (1L << 31)) return LFM_ERR_SHAPE;
  const dim3 ggrid(cdiv((long)HW * (C / 8), 256), N);
  if (silu) hipLaunchKernelGGL(gn_affine_kernel<true>, ggrid, dim3(256), 0, st, in, (half_t*)y, ab, HW, C, total8);
  else hipLaunchKernelGGL(gn_affine_kernel<false>, ggrid, dim3(256), 0, st, in, (half_t*)y, ab, HW, C, total8);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

extern "C" int lfm_groupnorm_f16(const void* x, void* y, const float* gamma, const float* beta, const float* film, long film_stride, void* scratch,
                                 int N, int HW, int C, int groups, float eps, int silu, lfm_stream_t stream) {
  return groupnorm_impl(GnIn{(const half_t*)x, nullptr, C, 0}, y, gamma, beta, film, film_stride, scratch, N, HW, C, groups, eps, silu, stream);
}
// GroupNorm of the channel concat [xa (Ca channels) | xb (Cb channels)] without materialising it; y is dense [N*HW, Ca + Cb]
extern "C" int lfm_groupnorm2_f16(const void* xa, int Ca, const void* xb, int Cb, void* y, const float* gamma, const float* beta, const float* film,
                                  long film_stride, void* scratch, int N, int HW, int groups, float eps, int silu, lfm_stream_t stream) {
  if (!xa || !xb) return LFM_ERR_ARG;
  if (Ca <= 0 || Cb <= 0 || (Ca % 8) || (Cb % 8)) return LFM_ERR_SHAPE;
  if (((uintptr_t)xa | (uintptr_t)xb) & 15) return LFM_ERR_ALIGN;
  return groupnorm_impl(GnIn{(const half_t*)xa, (const half_t*)xb, Ca, Cb}, y, gamma, beta, film, film_stride, scratch, N, HW, Ca + Cb, groups, eps, silu,
                        stream);
}

// ------------------------------------------------------------------ 2x2 average pooling, stride 2 (EDM Conv2d(down=True) with
// resample_filter [1,1]: conv2d with the 2x2 box filter / 4, models/EDM.py:96-98,122-125)
__global__ void avgpool2_kernel(const half8_t* __restrict__ x, half8_t* __restrict__ y, int Ho, int Wo, int C8, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C8);
  const long p = i / C8;
  const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho);
  const long n = p / ((long)Wo * Ho);
  const long Wi = 2L * Wo;
  const half8_t* b = x + ((n * 2 * Ho + 2 * oy) * Wi + 2 * ox) * C8 + c;
  const half8_t a0 = b[0], a1 = b[C8], a2 = b[Wi * C8], a3 = b[Wi * C8 + C8];
  half8_t o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)(0.25f * ((float)a0[j] + (float)a1[j] + (float)a2[j] + (float)a3[j]));
  y[i] = o;
}
extern "C" int lfm_avgpool2_f16(const void* x, void* y, int N, int Ho, int Wo, int C, lfm_stream_t stream) {
  if (!x || !y) return LFM_ERR_ARG;
  if (N <= 0 || Ho <= 0 || Wo <= 0 || C % 8) return LFM_ERR_SHAPE;
  const long total = (long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(avgpool2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const half8_t*)x, (half8_t*)y, Ho, Wo, C / 8, total);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

// ------------------------------------------------------------------ nearest-neighbour 2x upsample (EDM Conv2d(up=True, kernel=0) skip path:
// conv_transpose2d with the all-ones 2x2 filter, models/EDM.py:117-121)
__global__ void upsample2_kernel(const half8_t* __restrict__ x, half8_t* __restrict__ y, int Ho, int Wo, int C8, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C8);
  const long p = i / C8;
  const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho);
  const long n = p / ((long)Wo * Ho);
  y[i] = x[((n * (Ho / 2) + oy / 2) * (Wo / 2) + ox / 2) * C8 + c];
}
extern "C" int lfm_upsample2_f16(const void* x, void* y, int N, int Ho, int Wo, int C, lfm_stream_t stream) {
  if (!x || !y) return LFM_ERR_ARG;
  if (N <= 0 || Ho <= 0 || Wo <= 0 || ((Ho | Wo) & 1) || C % 8) return LFM_ERR_SHAPE;
  const long total = (long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(upsample2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const half8_t*)x, (half8_t*)y, Ho, Wo, C / 8, total);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

// ------------------------------------------------------------------ h + emb_out[..., None, None] (ResBlock without scale-shift norm, unet.py:233-235)
__global__ void add_image_vec_kernel(const half8_t* __restrict__ x, const float* __restrict__ e, long e_stride, half8_t* __restrict__ y, int HW,
                                     int C8, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C8);
  const long n = i / ((long)C8 * HW);
  const half8_t v = x[i];
  const float* ev = e + n * e_stride + c * 8;
  half8_t o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)v[j] + ev[j]);
  y[i] = o;
}
extern "C" int lfm_add_image_vec_f16(const void* x, const float* e, long e_stride, void* y, int N, int HW, int C, lfm_stream_t stream) {
  if (!x || !e || !y) return LFM_ERR_ARG;
  if (N <= 0 || HW <= 0 || C % 8) return LFM_ERR_SHAPE;
  const long total = (long)N * HW * (C / 8);
  hipLaunchKernelGGL(add_image_vec_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const half8_t*)x, e, e_stride, (half8_t*)y, HW,
                     C / 8, total);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

// ------------------------------------------------------------------ channel concat (th.cat([h, skip], dim=1), unet.py:649)
__global__ void concat_c_kernel(const half8_t* __restrict__ a, const half8_t* __restrict__ b, half8_t* __restrict__ o, long pixels, int Ca8, int Cb8) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Co8 = Ca8 + Cb8;
  if (i >= pixels * Co8) return;
  const long p = i / Co8;
  const int c = (int)(i - p * Co8);
  o[i] = c < Ca8 ? a[p * Ca8 + c] : b[p * Cb8 + (c - Ca8)];
}
extern "C" int lfm_concat_channels_f16(const void* a, const void* b, void* out, long pixels, int Ca, int Cb, lfm_stream_t stream) {
  if (!a || !b || !out) return LFM_ERR_ARG;
  if (Ca % 8 || Cb % 8 || pixels <= 0) return LFM_ERR_SHAPE;
  hipLaunchKernelGGL(concat_c_kernel, dim3(cdiv(pixels * ((Ca + Cb) / 8), 256)), dim3(256), 0, (hipStream_t)stream, (const half8_t*)a,
                     (const half8_t*)b, (half8_t*)out, pixels, Ca / 8, Cb / 8);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

// ------------------------------------------------------------------ small-T attention (QKVAttentionLegacy, unet.py:310-334)
// qkv: fp16 [N*T, 3*C], column layout [head][q | k | v][ch] (what reshape(bs*heads, 3*ch, T).split(ch) means for a
// token-major tensor); out: fp16 [N*T, C] with columns [head][ch].  softmax((q*s)(k*s)^T) v with s = ch^-1/4, fp32 softmax.
// One workgroup per (head, image); T <= 256, ch <= 256.  FLOPs are negligible (T <= 64 in every reference config), so this
// is a plain VALU kernel: K and V rows in LDS as fp32, one query per thread-group.
__global__ __launch_bounds__(256) void attention_small_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int T, int heads, int ch,
                                                              int QB) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];  // S fp32 [QB][T+1], then K, V fp16 [T][ch+2]
  const int head = blockIdx.x, n = blockIdx.y, q0 = blockIdx.z * QB, tid = threadIdx.x;
  const int nq = min(QB, T - q0);
  const int C = heads * ch, ldq = 3 * C, ks = ch + 2;
  float* S = (float*)smraw;
  half_t* Ks = (half_t*)(S + QB * (T + 1));
  half_t* Vs = Ks + T * ks;
  const half_t* base = qkv + (long)n * T * ldq + head * 3 * ch;
  for (int e = tid; e < T * ch; e += 256) {
    const int t = e / ch, c = e - t * ch;
    Ks[t * ks + c] = base[(long)t * ldq + ch + c];
    Vs[t * ks + c] = base[(long)t * ldq + 2 * ch + c];
  }
  __syncthreads();
  const float scale = rsqrtf((float)ch);  // (ch^-1/4)^2
  for (int e = tid; e < nq * T; e += 256) {
    const int t = e / T, s = e - t * T;
    const half_t* q = base + (long)(q0 + t) * ldq;
    float a = 0.f;
    for (int c = 0; c < ch; ++c) a += (float)q[c] * (float)Ks[s * ks + c];
    S[t * (T + 1) + s] = a * scale;
  }
  __syncthreads();
  for (int t = tid; t < nq; t += 256) {
    float* r = S + t * (T + 1);
    float mx = r[0];
    for (int s = 1; s < T; ++s) mx = fmaxf(mx, r[s]);
    float sum = 0.f;
    for (int s = 0; s < T; ++s) {
      r[s] = __expf(r[s] - mx);
      sum += r[s];
    }
    const float inv = 1.0f / sum;
    for (int s = 0; s < T; ++s) r[s] *= inv;
  }
  __syncthreads();
  half_t* ob = out + ((long)n * T + q0) * C + head * ch;
  for (int e = tid; e < nq * ch; e += 256) {
    const int t = e / ch, c = e - t * ch;
    const float* r = S + t * (T + 1);
    float a = 0.f;
    for (int s = 0; s < T; ++s) a += r[s] * (float)Vs[s * ks + c];
    ob[(long)t * C + c] = (half_t)a;
  }
}

// MFMA path for the shapes the big UNets use (T = 64 / 256 tokens = 8x8 / 16x16 maps, ch = 64 / 128 per head).  The VALU kernel above was written for
// "T <= 64, FLOPs negligible"; the celeb512 UNet attends at 16x16 (T = 256, ch = 128: 4.3 GFLOP per call at batch 32) and spent 5 % of an evaluation there
// (85-104 us per call = 45 TFLOP/s).  One workgroup = 64 queries of one (image, head), a wave = 16 queries:
//   K [T][ch] and V^T [ch][T] of the head in LDS (padded rows; V is transposed while it is staged, two keys per dword);
//   S^T = K Q^T on v_mfma_f32_16x16x32_f16 with K as the A operand: lane (q = lane >> 4, j = lane & 15) ends up with the scores of ONE query j
//   at keys 16 tile + 4 q + r -- the whole softmax row of a query lives in four lanes (two xor-shuffles per reduction), and
//   O^T = V^T P^T takes P straight from those registers as the B operand: the 8 k-slots of lane q in key step kt are keys 32 kt + 4 q + r and
//   32 kt + 16 + 4 q + r, which is what the two 8-byte V^T reads of the A operand fetch.  No transposition of P, no cross-lane traffic for O;
//   fp32 scores / softmax, P and V in fp16 (as the DiT attention kernel), 1 / sum applied to the fp32 accumulators.
template <int CH, int T>
__global__ __launch_bounds__(256) void attention_unet_mfma_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int heads, float scale) {
  constexpr int KS = CH * 2 + 16, VS = T * 2 + 16;  // LDS row strides in bytes (16 B of padding: consecutive rows start 4 banks apart)
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  char* Ks = smraw;            // [T][KS]
  char* Vt = smraw + T * KS;   // [CH][VS]
  const int head = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = heads * CH, ldq = 3 * C;
  const half_t* base = qkv + (long)n * T * ldq + head * 3 * CH;
  // ---- stage K (16-byte chunks) and V^T (two keys x 8 channels per item -> eight dword writes)
  for (int e = tid; e < T * (CH / 8); e += 256) {
    const int t = e / (CH / 8), c8 = e - t * (CH / 8);
    *(half8_t*)(Ks + t * KS + c8 * 16) = *(const half8_t*)(base + (long)t * ldq + CH + c8 * 8);
  }
  for (int e = tid; e < (T / 2) * (CH / 8); e += 256) {
    const int tp = e % (T / 2), c8 = e / (T / 2);  // consecutive lanes: consecutive key pairs of one channel octet (conflict-free dword writes)
    const half8_t v0 = *(const half8_t*)(base + (long)(2 * tp) * ldq + 2 * CH + c8 * 8);
    const half8_t v1 = *(const half8_t*)(base + (long)(2 * tp + 1) * ldq + 2 * CH + c8 * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) *(half2_t*)(Vt + (c8 * 8 + i) * VS + tp * 4) = (half2_t){v0[i], v1[i]};
  }
  // ---- this wave's 16 queries as the B operand of S^T: lane (q, j) holds Q[q0 + j][32 ks + 8 q .. + 7]
  const int j = lane & 15, q = lane >> 4;
  const int q0 = blockIdx.z * 64 + wave * 16;
  half8_t qf[CH / 32];
#pragma unroll
  for (int ks = 0; ks < CH / 32; ++ks) qf[ks] = *(const half8_t*)(base + (long)(q0 + j) * ldq + ks * 32 + q * 8);
  __syncthreads();
  // ---- S^T tiles: st[tile][r] = score of query j at key 16 tile + 4 q + r
  f32x4 st[T / 16];
#pragma unroll
  for (int tile = 0; tile < T / 16; ++tile) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < CH / 32; ++ks) {
      const half8_t kf = *(const half8_t*)(Ks + (tile * 16 + j) * KS + (ks * 4 + q) * 16);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], a, 0, 0, 0);
    }
    st[tile] = a;
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int tile = 0; tile < T / 16; ++tile) mx = fmaxf(fmaxf(fmaxf(st[tile].x, st[tile].y), fmaxf(st[tile].z, st[tile].w)), mx);
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float sl = scale * 1.4426950408889634f, mo = mx * sl;
  float sum = 0.f;
  half4_t pf[T / 16];
#pragma unroll
  for (int tile = 0; tile < T / 16; ++tile) {
    const float e0 = __builtin_amdgcn_exp2f(st[tile].x * sl - mo), e1 = __builtin_amdgcn_exp2f(st[tile].y * sl - mo);
    const float e2 = __builtin_amdgcn_exp2f(st[tile].z * sl - mo), e3 = __builtin_amdgcn_exp2f(st[tile].w * sl - mo);
    sum += (e0 + e1) + (e2 + e3);
    pf[tile] = (half4_t){(half_t)e0, (half_t)e1, (half_t)e2, (half_t)e3};
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  // ---- O^T = V^T P^T: channel tile ct, key step kt (32 keys = score tiles 2 kt, 2 kt + 1)
  half_t* ob = out + ((long)n * T + q0 + j) * C + head * CH;
#pragma unroll
  for (int ct = 0; ct < CH / 16; ++ct) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < T / 32; ++kt) {
      const char* vr = Vt + (ct * 16 + j) * VS + (kt * 32 + q * 4) * 2;
      const half4_t va = *(const half4_t*)vr, vb = *(const half4_t*)(vr + 32);
      const half8_t vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
      const half4_t pa = pf[2 * kt], pb = pf[2 * kt + 1];
      const half8_t pp = {pa[0], pa[1], pa[2], pa[3], pb[0], pb[1], pb[2], pb[3]};
      o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pp, o, 0, 0, 0);
    }
    // lane holds O[query j][channels 16 ct + 4 q .. + 3]
    *(half4_t*)(ob + ct * 16 + q * 4) = (half4_t){(half_t)(o.x * inv), (half_t)(o.y * inv), (half_t)(o.z * inv), (half_t)(o.w * inv)};
  }
}

template <int CH, int T>
static int launch_attention_unet_mfma(const half_t* qkv, half_t* out, int N, int heads, hipStream_t st) {
  constexpr int LDS = T * (CH * 2 + 16) + CH * (T * 2 + 16);
  static unsigned long long attr_set = 0;
  int devid = 0;
  (void)hipGetDevice(&devid);
  if (!((attr_set >> (devid & 63)) & 1)) {
    if (hipFuncSetAttribute((const void*)attention_unet_mfma_kernel<CH, T>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LFM_ERR_LAUNCH;
    attr_set |= 1ull << (devid & 63);
  }
  hipLaunchKernelGGL((attention_unet_mfma_kernel<CH, T>), dim3(heads, N, T / 64), dim3(256), LDS, st, qkv, out, heads, 1.0f / sqrtf((float)CH));
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

extern "C" int lfm_attention_small_f16(const void* qkv, void* out, int N, int T, int heads, int ch, lfm_stream_t stream) {
  if (!qkv || !out) return LFM_ERR_ARG;
  if (N <= 0 || T <= 0 || heads <= 0 || ch <= 0) return LFM_ERR_SHAPE;
  if (!(((uintptr_t)qkv | (uintptr_t)out) & 15) && !(lfm_gemm_debug_flags() & 16)) {  // flag 16: the VALU kernel (A/B)
    const half_t* qi = (const half_t*)qkv;
    half_t* oi = (half_t*)out;
    hipStream_t st = (hipStream_t)stream;
    if (T == 256 && ch == 128) return launch_attention_unet_mfma<128, 256>(qi, oi, N, heads, st);
    if (T == 256 && ch == 64) return launch_attention_unet_mfma<64, 256>(qi, oi, N, heads, st);
    if (T == 64 && ch == 128) return launch_attention_unet_mfma<128, 64>(qi, oi, N, heads, st);
    if (T == 64 && ch == 64) return launch_attention_unet_mfma<64, 64>(qi, oi, N, heads, st);
  }
  const int QB = T < 64 ? T : 64;  // queries per workgroup
  const size_t lds = (size_t)QB * (T + 1) * 4 + (size_t)2 * T * (ch + 2) * 2;
  if (lds > 160 * 1024) return LFM_ERR_SHAPE;  // every reference config has T <= 64 (8x8 / 4x4 feature maps)
  static lfm_device_mask set{0};
  const unsigned long long dbit = lfm_device_bit();
  if (lfm_device_todo(set, dbit)) {
    if (hipFuncSetAttribute((const void*)attention_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return LFM_ERR_LAUNCH;
    lfm_device_done(set, dbit);
  }
  hipLaunchKernelGGL(attention_small_kernel, dim3(heads, N, cdiv(T, QB)), dim3(256), lds, (hipStream_t)stream, (const half_t*)qkv, (half_t*)out,
                     T, heads, ch, QB);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

// ------------------------------------------------------------------ timestep embedding MLP (nn.py:103-121 + unet.py:633-641)
// emb[r] = W2 silu(W0 [cos(t f) | sin(t f)] + b0) + b2 (+ label_emb[y[r]]);  also emits fp16 silu(emb) for the ResBlock emb_layers
__global__ __launch_bounds__(256) void time_embed1_kernel(const float* __restrict__ t, int t_len, const float* __restrict__ w0,
                                                          const float* __restrict__ b0, float* __restrict__ h1, int F, int E) {
  const int r = blockIdx.y, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= E) return;
  const float tv = t[t_len == 1 ? 0 : r];
  const int half = F / 2;
  const float* w = w0 + (long)j * F;
  float s = 0.f;
  for (int k = lane; k < 2 * half; k += 64) {
    const int i = k < half ? k : k - half;
    const float a = tv * expf(-9.210340371976184f * (float)i / (float)half);
    s += w[k] * (k < half ? cosf(a) : sinf(a));
  }
  s = wave_sum(s);
  if (lane == 0) h1[(long)r * E + j] = silu_f(s + b0[j]);
}
__global__ __launch_bounds__(256) void time_embed2_kernel(const float* __restrict__ h1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                          const float* __restrict__ label_table, const int64_t* __restrict__ y,
                                                          int label_rows, float* __restrict__ emb, half_t* __restrict__ emb_silu, int E) {
  const int r = blockIdx.y, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= E) return;
  const float* w = w2 + (long)j * E;
  const float* h = h1 + (long)r * E;
  float s = 0.f;
  for (int k = lane; k < E; k += 64) s += w[k] * h[k];
  s = wave_sum(s);
  if (lane == 0) {
    float v = s + b2[j];
    if (label_table) {  // out-of-range label: nn.Embedding raises in the reference; poison instead of reading out of bounds
      const long yr = (long)y[r];
      v = (yr >= 0 && yr < label_rows) ? v + label_table[yr * E + j] : __builtin_nanf("");
    }
    emb[(long)r * E + j] = v;
    emb_silu[(long)r * E + j] = (half_t)silu_f(v);
  }
}

extern "C" int lfm_time_embed(const float* t, int t_len, const float* w0, const float* b0, const float* w2, const float* b2,
                              const float* label_table, const int64_t* y, int label_rows, float* scratch_h1, float* emb, void* emb_silu_f16, int N, int F,
                              int E, lfm_stream_t stream) {
  if (!t || !w0 || !b0 || !w2 || !b2 || !scratch_h1 || !emb || !emb_silu_f16) return LFM_ERR_ARG;
  if ((label_table != nullptr) != (y != nullptr)) return LFM_ERR_ARG;
  if (label_table && label_rows <= 0) return LFM_ERR_SHAPE;
  if (N <= 0 || F <= 0 || (F & 1) || E <= 0 || (t_len != 1 && t_len != N)) return LFM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(time_embed1_kernel, dim3(cdiv(E, 4), N), dim3(256), 0, st, t, t_len, w0, b0, scratch_h1, F, E);
  LFM_CHECK_LAUNCH();
  hipLaunchKernelGGL(time_embed2_kernel, dim3(cdiv(E, 4), N), dim3(256), 0, st, scratch_h1, w2, b2, label_table, y, label_rows, emb, (half_t*)emb_silu_f16, E);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
