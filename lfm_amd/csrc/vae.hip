// SD f8 KL-VAE decoder on gfx950 (decode half of diffusers AutoencoderKL, sd-vae-ft-mse config).
// Reference call site: /root/reference/test_flow_latent.py:193  first_stage_model.decode(z / scale_factor).sample
// Activations are NHWC fp16 ([pixels, C] row-major == the GEMM "A" operand), so every 3x3 convolution is an
// implicit GEMM on the same MFMA kernel as the DiT linears: M = N*H*W pixels, N = Cout, K = 9*Cin with
// k = tap*Cin + ci; the A-tile gather (shifted pixel rows, zero padding, optional nearest-2x upsample) is done
// by the per-lane LDS-DMA source address.  GroupNorm statistics / apply+SiLU are bandwidth-bound side kernels.
#include "../../include/lfm_hip.h"
#include "gemm_dispatch.h"
#include "conv_halo_kernel.h"

// ------------------------------------------------------------------ implicit-GEMM A source: 3x3 conv, pad 1, NHWC
// UPS=1: the convolution runs on the nearest-2x upsampled image (diffusers Upsample2D) without materialising it.
template <int UPS>
struct ASrcConv3x3 {
  const half_t* in;    // [N, Hs, Ws, Cin] with Hs = H >> UPS
  const half_t* zeros; // >= 64 halves of zeros (padding rows)
  int H, W, Cin, M;    // output (= conv input after upsample) spatial size; M = N*H*W
  int tap, ci0;        // k-tile state: k0 = tap*Cin + ci0 (a 64-wide k tile never straddles a tap: Cin % 64 == 0)
  __device__ __forceinline__ void init(int, long) {}
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
  __device__ __forceinline__ void begin_tile(int kt, int bk) {  // called with kt = 0, 1, 2, ... in order
    if (kt == 0) {
      tap = 0;
      ci0 = 0;
    } else {
      ci0 += bk;
      if (ci0 >= Cin) {
        ci0 = 0;
        ++tap;
      }
    }
  }
  __device__ __forceinline__ const half_t* ptr(const Row& r, int koff) const {
    const int iy = r.y + tap / 3 - 1, ix = r.x + tap % 3 - 1;
    if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return zeros + koff;
    const int Hs = H >> UPS, Ws = W >> UPS;
    return in + (((long)r.n * Hs + (iy >> UPS)) * Ws + (ix >> UPS)) * Cin + ci0 + koff;
  }
};

// Encoder downsampling (diffusers Downsample2D with padding=0: F.pad(x, (0,1,0,1)) then conv3x3, stride 2): output (oy, ox) reads
// input (2*oy + ky, 2*ox + kx), zero beyond the bottom / right edge.  H, W are the OUTPUT size; the input is [N, 2H, 2W, Cin].
struct ASrcConvDown {
  const half_t* in;
  const half_t* zeros;
  int H, W, Cin, M;
  int tap, ci0;
  __device__ __forceinline__ void init(int, long) {}
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
      tap = 0;
      ci0 = 0;
    } else {
      ci0 += bk;
      if (ci0 >= Cin) {
        ci0 = 0;
        ++tap;
      }
    }
  }
  __device__ __forceinline__ const half_t* ptr(const Row& r, int koff) const {
    const int iy = 2 * r.y + tap / 3, ix = 2 * r.x + tap % 3;
    if (iy >= 2 * H || ix >= 2 * W) return zeros + koff;
    return in + (((long)r.n * (2 * H) + iy) * (2 * W) + ix) * Cin + ci0 + koff;
  }
};

// ------------------------------------------------------------------ epilogues
struct EpiConvF16 {  // out = acc + bias (+ residual)  -> fp16 NHWC
  half_t* C;
  long ldc;
  const float* bias;
  const half_t* resid;  // may be null; same layout as C
  struct Aux {
    f32x4 b;
    half4_t r;
  };
  __device__ __forceinline__ Aux load(int m, int n) const {
    Aux a;
    a.b = *(const f32x4*)(bias + n);
    if (resid) a.r = *(const half4_t*)(resid + (long)m * ldc + n);
    else a.r = (half4_t){0, 0, 0, 0};
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

// EpiConvF16 that also leaves the GroupNorm statistics of what it stores (round 3): the next layer of every resnet is a GroupNorm over exactly this
// tensor, and its statistics pass (gn_stats_kernel) re-read it from HBM just to add it up.  In the row-major hand-over of the 256-row kernels a lane
// owns the same eight output columns for all of its rows, so it keeps (sum, sum of squares) of the ROUNDED fp16 values per half-octet (a group is
// >= 4 channels wide) in four registers, folds the eight lanes that share its columns at the end of the tile, and writes one fixed slot per
// (image, 128-row slab, half-octet): part[n][slab][C / 4][2], the layout gn_finish_kernel folds in a fixed order -- deterministic, no atomics.
// Host-side preconditions (conv3): HW % 256 == 0 (a tile lies in one image), the 256x128 or 256x256 kernel, 16-byte-store path.
struct EpiConvStatsF16 {
  half_t* C;
  long ldc;
  const float* bias;
  const half_t* resid;
  float* part;  // [n][slabs][ldc / 4][2]
  int HW, slabs;
  mutable float s0, q0, s1, q1;
  typedef EpiConvF16::Aux Aux;
  __device__ __forceinline__ Aux load(int m, int n) const {
    Aux a;
    a.b = *(const f32x4*)(bias + n);
    if (resid) a.r = *(const half4_t*)(resid + (long)m * ldc + n);
    else a.r = (half4_t){0, 0, 0, 0};
    return a;
  }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& a) const {  // (4-column path: not used with statistics, kept for the interface)
    v += a.b;
    half4_t h = {(half_t)(v.x + (float)a.r.x), (half_t)(v.y + (float)a.r.y), (half_t)(v.z + (float)a.r.z), (half_t)(v.w + (float)a.r.w)};
    *(half4_t*)(C + (long)m * ldc + n) = h;
  }
  __device__ __forceinline__ bool wide_ok() const { return true; }  // checked on the host
  __device__ __forceinline__ void store8(int m, int n, f32x4 lo, f32x4 hi, const Aux& al, const Aux& ah) const {
    lo += al.b;
    hi += ah.b;
    const half8_t h = {(half_t)(lo.x + (float)al.r.x), (half_t)(lo.y + (float)al.r.y), (half_t)(lo.z + (float)al.r.z), (half_t)(lo.w + (float)al.r.w),
                       (half_t)(hi.x + (float)ah.r.x), (half_t)(hi.y + (float)ah.r.y), (half_t)(hi.z + (float)ah.r.z), (half_t)(hi.w + (float)ah.r.w)};
    *(half8_t*)(C + (long)m * ldc + n) = h;
    const float f0 = (float)h[0], f1 = (float)h[1], f2 = (float)h[2], f3 = (float)h[3], f4 = (float)h[4], f5 = (float)h[5], f6 = (float)h[6], f7 = (float)h[7];
    s0 += (f0 + f1) + (f2 + f3);
    q0 += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
    s1 += (f4 + f5) + (f6 + f7);
    q1 += (f4 * f4 + f5 * f5) + (f6 * f6 + f7 * f7);
  }
  // fold the eight lanes that own the same columns and write the wave's slot: columns ncol0 + 8 (lane & 7) .. + 7 of slab `slab` of image `img`
  __device__ __forceinline__ void finish_slab(int img, int slab, int ncol0, int lane) const {
    float a = s0, b = q0, c = s1, d = q1;
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {  // the eight lanes lane & 7, + 8, .., + 56 own the same columns
      a += __shfl_xor(a, o, 64);
      b += __shfl_xor(b, o, 64);
      c += __shfl_xor(c, o, 64);
      d += __shfl_xor(d, o, 64);
    }
    if (lane < 8) {
      float* o = part + (((long)img * slabs + slab) * (ldc >> 2) + ((ncol0 + lane * 8) >> 2)) * 2;
      *(f32x4*)o = (f32x4){a, b, c, d};
    }
  }
  __device__ __forceinline__ void finish_tile(int m0, int n0, int g, int wn, int lane) const {
    const int img = m0 / HW;
    finish_slab(img, ((m0 - img * HW) >> 8) * 2 + g, n0 + wn * 64, lane);
  }
};

struct EpiTransposeF16 {  // per image: Ct[img][n][m % T] = acc + bias[n]   (V^T for the mid attention)
  half_t* Ct;
  const float* bias;
  int T, N;
  typedef f32x4 Aux;
  __device__ __forceinline__ bool direct(int) const { return true; }
  __device__ __forceinline__ Aux load(int, int n) const { return *(const f32x4*)(bias + n); }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& b) const {
    v += b;
    const int img = m / T, tok = m - img * T;
    half_t* d = Ct + ((long)img * N + n) * T + tok;
    d[0] = (half_t)v.x;
    d[T] = (half_t)v.y;
    d[2 * T] = (half_t)v.z;
    d[3 * T] = (half_t)v.w;
  }
};

struct EpiBatchF32 {  // batched scores: S[bz][m][n] = acc
  float* C;
  long ldc;
  typedef int Aux;
  __device__ __forceinline__ void batch(int bz, long bs) { C += (long)bz * bs; }
  __device__ __forceinline__ Aux load(int, int) const { return 0; }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux&) const { *(f32x4*)(C + (long)m * ldc + n) = v; }
};

struct EpiBatchF16 {  // batched: O[bz][m][n] = acc -> fp16
  half_t* C;
  long ldc;
  typedef int Aux;
  __device__ __forceinline__ void batch(int bz, long bs) { C += (long)bz * bs; }
  __device__ __forceinline__ Aux load(int, int) const { return 0; }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux&) const {
    half4_t h = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
    *(half4_t*)(C + (long)m * ldc + n) = h;
  }
};

struct EpiConvOutNCHW {  // final conv (Cout=3, padded to 4): fp32 NCHW image, the `.sample` tensor
  float* out;
  const float* bias;  // [4]
  int HW;             // H*W
  typedef f32x4 Aux;
  __device__ __forceinline__ Aux load(int, int n) const { return *(const f32x4*)(bias + n); }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& b) const {
    if (n != 0) return;
    v += b;
    const int img = m / HW, pix = m - img * HW;
    float* o = out + (long)img * 3 * HW + pix;
    o[0] = v.x;
    o[HW] = v.y;
    o[2 * HW] = v.z;
  }
};

struct EpiMomentsNCHW {  // encoder conv_out (+ folded quant_conv): fp32 NCHW [img][nch][pix], nch a multiple of 4
  float* out;
  const float* bias;
  int HW, nch;
  typedef f32x4 Aux;
  __device__ __forceinline__ Aux load(int, int n) const { return n < nch ? *(const f32x4*)(bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ void store(int m, int n, f32x4 v, const Aux& b) const {
    if (n >= nch) return;
    v += b;
    const int img = m / HW, pix = m - img * HW;
    float* o = out + ((long)img * nch + n) * HW + pix;
    o[0] = v.x;
    o[HW] = v.y;
    o[2 * HW] = v.z;
    o[3 * HW] = v.w;
  }
};

// ------------------------------------------------------------------ post_quant_conv (1x1, 4->4) + conv_in (3x3, 4->Cout)
// z fp32 NCHW [N,4,R,R] -> fp16 NHWC [N,R,R,Cout].  conv_in sees post_quant(z) zero-padded, so the 1x1 is
// evaluated per tap and skipped (=0) outside the image.  One thread = one pixel x 8 output channels.
#define VCI_PIX 256
__global__ __launch_bounds__(256) void vae_conv_in_kernel(const float* __restrict__ z, const float* __restrict__ pq_w,
                                                          const float* __restrict__ pq_b, const float* __restrict__ w,
                                                          const float* __restrict__ b, half_t* __restrict__ out, int N, int R, int Cout, int ppb) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // conv_in weights transposed to [k = (c,ky,kx)][Cout]
  for (int e = threadIdx.x; e < 36 * Cout; e += 256) {
    const int co = e / 36, k = e - co * 36;
    wl[k * Cout + co] = w[e];
  }
  __syncthreads();
  const int c8n = Cout / 8, rows = 256 / c8n;
  const int oct = threadIdx.x % c8n, prow = threadIdx.x / c8n;
  if (prow >= rows) return;
  const int co = oct * 8;
  const long total = (long)N * R * R, p0 = (long)blockIdx.x * ppb;  // ppb = VCI_PIX pixels per block, fewer when the batch alone cannot fill the chip
  for (long pix = p0 + prow; pix < p0 + ppb && pix < total; pix += rows) {
    const int x = (int)(pix % R), y = (int)((pix / R) % R), n = (int)(pix / ((long)R * R));
    f32x4 a0 = *(const f32x4*)(b + co), a1 = *(const f32x4*)(b + co + 4);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
      if ((unsigned)iy >= (unsigned)R || (unsigned)ix >= (unsigned)R) continue;
      float zi[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) zi[c] = z[(((long)n * 4 + c) * R + iy) * R + ix];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float pq = pq_b[c] + pq_w[c * 4 + 0] * zi[0] + pq_w[c * 4 + 1] * zi[1] + pq_w[c * 4 + 2] * zi[2] + pq_w[c * 4 + 3] * zi[3];
        const float* wr = wl + (c * 9 + t) * Cout + co;
        a0 += pq * *(const f32x4*)wr;
        a1 += pq * *(const f32x4*)(wr + 4);
      }
    }
    half8_t h = {(half_t)a0.x, (half_t)a0.y, (half_t)a0.z, (half_t)a0.w, (half_t)a1.x, (half_t)a1.y, (half_t)a1.z, (half_t)a1.w};
    *(half8_t*)(out + pix * Cout + co) = h;
  }
}

// ------------------------------------------------------------------ GroupNorm(32 groups, eps 1e-6) on NHWC fp16
// Two-stage, deterministic statistics (the reference is deterministic; float atomics are not): every block folds its slab of pixels
// into per-half-octet partial {sum, sumsq} slots part[n][slab][C/4], a small second kernel folds the slabs and the half-octets of a
// group in a fixed order into stats[n][g]; apply fuses SiLU.
// Block = 256 threads over a slab of pixels; thread t owns channel-octet (t % (C/8)) and strides over pixels.
#define VGN_MAX_SLABS 64
__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                       int pix_per_block) {
  __shared__ float red[4][256];
  const int n = blockIdx.y, c8n = C / 8, tid = threadIdx.x;
  const int oct = tid % c8n, prow = tid / c8n, pstride = 256 / c8n;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, HW);
  float s[2] = {0.f, 0.f}, q[2] = {0.f, 0.f};  // per half-octet (4 channels): a group is >= 4 channels wide
  const half_t* base = x + (long)n * HW * C + oct * 8;
  for (int p = p0 + prow; p < p1; p += pstride) {
    const half8_t v = *(const half8_t*)(base + (long)p * C);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)v[j];
      s[j >> 2] += f;
      q[j >> 2] += f * f;
    }
  }
  red[0][tid] = s[0];
  red[1][tid] = s[1];
  red[2][tid] = q[0];
  red[3][tid] = q[1];
  __syncthreads();
  if (tid < c8n) {  // fold the pixel-rows of this channel octet, then one atomic pair per half-octet
    for (int r = 1; r < pstride; ++r) {
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
// one WAVE per (image, group): lane l folds slabs l, l + 64, ... in order, then a fixed-tree wave sum (deterministic).  (One thread per
// (image, group) walking all slabs serially took 75 us once the convolution epilogues started to deliver 512 slabs per image.)
__global__ __launch_bounds__(256) void gn_finish_kernel(const float* __restrict__ part, float* __restrict__ stats, int slabs, int C, int total) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // (n, g)
  if (i >= total) return;
  const int n = i >> 5, g = i & 31, hpg = C / 128;  // half-octets per group = (C/32)/4
  float sum = 0.f, sq = 0.f;
  for (int b = lane; b < slabs; b += 64) {
    const float* p = part + (((long)n * slabs + b) * (C / 4) + g * hpg) * 2;
    for (int h = 0; h < hpg; ++h) {
      sum += p[2 * h];
      sq += p[2 * h + 1];
    }
  }
  sum = wave_sum(sum);
  sq = wave_sum(sq);
  if (lane == 0) {
    stats[(long)i * 2] = sum;
    stats[(long)i * 2 + 1] = sq;
  }
}

// apply: y = silu?((x - mean) rstd gamma + beta).  Round 4: a thread OWNS one channel octet (256 % (C / 8) == 0) and walks the pixels of its block's
// slab, so the statistics, gamma and beta are read once per thread and there is no per-element 64-bit index arithmetic (the first version -- one
// 16-byte chunk per thread, four divisions and two 64-bit div / mod per chunk -- ran at 2.3 TB/s on the 2.15 GB full-resolution tensors:
// profiles/r03_final_bench_kernel_stats.csv, 6.2 ms of a 39.6 ms decode).  Same arithmetic per element.
template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int HW, int C,
                                                       int pix_per_block) {
  const int n = blockIdx.y, c8n = C / 8, cpg = C / 32, tid = threadIdx.x;
  const int oct = tid % c8n, prow = tid / c8n, pstride = 256 / c8n;
  const float cnt = (float)HW * (float)cpg;
  float mean[2], rstd[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int g = (oct * 8 + hh * 4) / cpg;
    mean[hh] = stats[((long)n * 32 + g) * 2] / cnt;
    const float var = fmaxf(stats[((long)n * 32 + g) * 2 + 1] / cnt - mean[hh] * mean[hh], 0.f);
    rstd[hh] = rsqrtf(var + 1e-6f);
  }
  const f32x4 g0 = *(const f32x4*)(gamma + oct * 8), g1 = *(const f32x4*)(gamma + oct * 8 + 4);
  const f32x4 b0 = *(const f32x4*)(beta + oct * 8), b1 = *(const f32x4*)(beta + oct * 8 + 4);
  const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  const int p0 = blockIdx.x * pix_per_block, p1 = min(p0 + pix_per_block, HW);
  const long base = ((long)n * HW) * C + oct * 8;
  auto one = [&](const half8_t v) {
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = ((float)v[j] - mean[j >> 2]) * rstd[j >> 2] * gm[j] + bt[j];
      if (SILU) f = silu_f(f);
      o[j] = (half_t)f;
    }
    return o;
  };
  int p = p0 + prow;
  for (; p + 3 * pstride < p1; p += 4 * pstride) {  // four chunks in flight per thread
    half8_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const half8_t*)(x + base + (long)(p + u * pstride) * C);
#pragma unroll
    for (int u = 0; u < 4; ++u) *(half8_t*)(y + base + (long)(p + u * pstride) * C) = one(v[u]);
  }
  for (; p < p1; p += pstride) *(half8_t*)(y + base + (long)p * C) = one(*(const half8_t*)(x + base + (long)p * C));
}

// rows of S [rows, T] fp32 -> P fp16 = softmax(S * scale); one wave per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, half_t* __restrict__ P, long rows, int T, float scale_log2e) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* s = S + r * T;
  float mx = -3.0e38f;
  for (int i = lane; i < T; i += 64) mx = fmaxf(mx, s[i]);
  mx = wave_max(mx) * scale_log2e;
  float sum = 0.f;
  for (int i = lane; i < T; i += 64) sum += __builtin_amdgcn_exp2f(s[i] * scale_log2e - mx);
  const float inv = 1.0f / wave_sum(sum);
  half_t* p = P + r * T;
  for (int i = lane; i < T; i += 64) p[i] = (half_t)(__builtin_amdgcn_exp2f(s[i] * scale_log2e - mx) * inv);
}

// ------------------------------------------------------------------ host side
static inline size_t a256(size_t v) { return (v + 255) / 256 * 256; }

struct VaeWs {
  half_t *b0, *b1, *b2, *b3;  // four ping-pong activation buffers of the largest size
  float* stats;               // [chunk, 32, 2]
  float* part;                // [chunk, VGN_MAX_SLABS, 128, 2] partial sums of the two-stage GroupNorm statistics
  half_t* zeros;              // 256 B
  float* S;                   // [chunk, T, T] scores
  size_t part_pairs;          // capacity of `part` in (sum, sum of squares) pairs
  size_t total;
};

static VaeWs vae_carve(int R, int chunk, void* ws) {
  // largest activation: 128 ch at 8R x 8R  ==  256 ch at 4R x 4R x 2 ... = chunk * (8R)^2 * 128 halves;
  // the upsample conv of block 2 writes 256 ch at 8R x 8R: chunk * (8R)^2 * 256 halves -> size for that.
  const size_t act = (size_t)chunk * (8 * R) * (8 * R) * 256 * 2;
  const size_t T = (size_t)R * R;
  size_t off = 0;
  char* base = (char*)ws;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += a256(bytes);
    return p;
  };
  VaeWs w;
  w.b0 = (half_t*)take(act);
  w.b1 = (half_t*)take(act);
  w.b2 = (half_t*)take(act);
  w.b3 = (half_t*)take(act);
  w.stats = (float*)take((size_t)chunk * 64 * 4);
  // the larger of: the statistics kernel's slabs (VGN_MAX_SLABS x C / 4 <= 128 half-octets) and the convolution epilogues' 128-row slabs
  // (2 HW / 256 per image x C / 4 half-octets: at most (8R)^2 * 256 / 512 for the 256-channel tensor at full resolution)
  const size_t part_conv = (size_t)(8 * R) * (8 * R) * 256 / 512, part_stats = (size_t)VGN_MAX_SLABS * 128;
  w.part_pairs = (size_t)chunk * (part_conv > part_stats ? part_conv : part_stats);
  w.part = (float*)take(w.part_pairs * 2 * 4);
  w.zeros = (half_t*)take(256);
  w.S = (float*)take((size_t)chunk * T * T * 4);
  w.total = off;
  return w;
}

extern "C" size_t lfm_vae_workspace_bytes(int R, int chunk) {
  if (R <= 0 || chunk <= 0 || (R % 8)) return 0;
  return vae_carve(R, chunk, nullptr).total;
}

#define RC(x)            \
  do {                   \
    int _rc = (x);       \
    if (_rc) return _rc; \
  } while (0)

// ready_slabs > 0: the convolution that produced x already left its partial sums in `part` (EpiConvStatsF16), in that many slabs per image
static int gn(const half_t* x, half_t* y, float* stats, float* part, const float* g, const float* b, int n, int HW, int C, bool silu,
              hipStream_t st, int ready_slabs = 0, size_t part_pairs = 0) {
  if (C % 128 || 256 % (C / 8) || C > 512) return LFM_ERR_SHAPE;  // groups of >= 4 channels, octet-per-thread mapping
  int slabs = ready_slabs;
  if (!ready_slabs) {
    // slabs per image: VGN_MAX_SLABS when the images fill the chip; a FEW images (--measure_time decodes ONE) get up to 512 -- 64 blocks walked a 256x256x128 map in
    // 64 dependent 16-byte loads per thread, 68 us per GroupNorm and 38 % of the batch-1 decode (profiles/r06_latency_mode.txt) -- as far as `part` has room
    // (part_pairs = its capacity in pairs, 0 = unknown: the old cap)
    int cap = VGN_MAX_SLABS;
    if (part_pairs && n * VGN_MAX_SLABS < 1024) {
      const long room = (long)(part_pairs / ((size_t)n * (C / 4)));
      const long want = 1024 / n;
      cap = (int)(want < room ? want : room);
      if (cap > 512) cap = 512;
      if (cap < VGN_MAX_SLABS) cap = VGN_MAX_SLABS;
    }
    int ppb = 1024;
    if (cap > VGN_MAX_SLABS) ppb = cdiv(HW, cap) > 64 ? cdiv(HW, cap) : 64;  // >= 64 pixels per block: at least a few loads per thread
    if (cdiv(HW, ppb) > cap) ppb = cdiv(HW, cap);
    slabs = cdiv(HW, ppb);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(slabs, n), dim3(256), 0, st, x, part, HW, C, ppb);
    LFM_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(gn_finish_kernel, dim3(cdiv(n * 32, 4)), dim3(256), 0, st, part, stats, slabs, C, n * 32);
  LFM_CHECK_LAUNCH();
  const int app = HW >= 4096 ? 256 : (HW >= 256 ? 64 : HW);  // pixels per block: 16 .. 4 chunks per thread at 128 .. 512 channels
  if (silu) hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(cdiv(HW, app), n), dim3(256), 0, st, x, y, stats, g, b, HW, C, app);
  else hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(cdiv(HW, app), n), dim3(256), 0, st, x, y, stats, g, b, HW, C, app);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

// out[n,H,W,Cout] = conv3x3(in (optionally nearest-2x upsampled)) + bias (+ resid)
// *stat_slabs (optional): set to the number of partial-sum slabs per image this convolution left in `part` for the GroupNorm that follows, or 0
static int conv3(const half_t* in, const half_t* w, const float* b, const half_t* resid, half_t* out, const half_t* zeros, int n, int H, int W,
                 int Cin, int Cout, bool ups, hipStream_t st, float* part = nullptr, int* stat_slabs = nullptr) {
  if (Cin % 64 || Cout % 4) return LFM_ERR_SHAPE;
  const int M = n * H * W;
  if (stat_slabs) *stat_slabs = 0;
  const int HW = H * W, kern = gemm_auto_choice(M, Cout, 9 * Cin);
  // every 3x3 convolution on a 16-aligned map: the halo-tiled direct kernel (conv_halo_kernel.h; 1.1-1.2 PFLOP/s where the implicit GEMM reaches
  // 0.65-1.05, profiles/r03_halo_conv_probe.txt); flag 8388608: the implicit GEMM instead (A/B)
  if (lfm_gemm_selected() == 0 && !(((uintptr_t)out | (uintptr_t)resid) & 15) && !(lfm_gemm_debug_flags() & 8388608)) {
    int rc;
    if (part && stat_slabs && (HW % 256) == 0 && !(lfm_gemm_debug_flags() & 4194304)) {
      EpiConvStatsF16 es{out, Cout, b, resid, part, HW, 2 * (HW / 256), 0.f, 0.f, 0.f, 0.f};
      rc = ups ? launch_conv3x3_halo<1>(in, zeros, w, n, H, W, Cin, Cout, es, st) : launch_conv3x3_halo<0>(in, zeros, w, n, H, W, Cin, Cout, es, st);
      if (rc == 0) *stat_slabs = es.slabs;
    } else {
      EpiConvF16 ep{out, Cout, b, resid};
      rc = ups ? launch_conv3x3_halo<1>(in, zeros, w, n, H, W, Cin, Cout, ep, st) : launch_conv3x3_halo<0>(in, zeros, w, n, H, W, Cin, Cout, ep, st);
    }
    if (rc != 1) return rc;
  }
  if (part && stat_slabs && (HW % 256) == 0 && (kern == 4 || kern == 5) && (Cout % (kern == 4 ? 128 : 256)) == 0 && (Cout % 128) == 0 &&
      !(((uintptr_t)out | (uintptr_t)resid) & 15) && !(lfm_gemm_debug_flags() & (1024 | 4194304))) {  // flag 4194304: the separate statistics pass (A/B)
    *stat_slabs = 2 * (HW / 256);
    EpiConvStatsF16 es{out, Cout, b, resid, part, HW, *stat_slabs, 0.f, 0.f, 0.f, 0.f};
    if (ups) return launch_gemm_auto(ASrcConv3x3<1>{in, zeros, H, W, Cin, M, 0, 0}, w, 9L * Cin, M, Cout, 9 * Cin, es, st);
    return launch_gemm_auto(ASrcConv3x3<0>{in, zeros, H, W, Cin, M, 0, 0}, w, 9L * Cin, M, Cout, 9 * Cin, es, st);
  }
  EpiConvF16 epi{out, Cout, b, resid};
  if (ups) return launch_gemm_auto(ASrcConv3x3<1>{in, zeros, H, W, Cin, M, 0, 0}, w, 9L * Cin, M, Cout, 9 * Cin, epi, st);
  return launch_gemm_auto(ASrcConv3x3<0>{in, zeros, H, W, Cin, M, 0, 0}, w, 9L * Cin, M, Cout, 9 * Cin, epi, st);
}

// x_slabs: in = partial-sum slabs per image already in ws.part for x (0 = none), out = the same for the result
static int resnet(const lfm_vae_resnet* r, half_t*& x, half_t*& t1, half_t*& t2, half_t*& t3, const VaeWs& ws, int n, int H, int W, hipStream_t st,
                  int* x_slabs = nullptr) {
  const int HW = H * W, M = n * HW;
  int mid_slabs = 0, out_slabs = 0;
  RC(gn(x, t1, ws.stats, ws.part, r->n1_g, r->n1_b, n, HW, r->cin, true, st, x_slabs ? *x_slabs : 0, ws.part_pairs));
  RC(conv3(t1, (const half_t*)r->c1_w, r->c1_b, nullptr, t2, ws.zeros, n, H, W, r->cin, r->cout, false, st, ws.part, &mid_slabs));
  RC(gn(t2, t1, ws.stats, ws.part, r->n2_g, r->n2_b, n, HW, r->cout, true, st, mid_slabs, ws.part_pairs));
  const half_t* skip = x;
  if (r->sc_w) {  // 1x1 conv shortcut
    RC(launch_gemm_auto(ASrcRowMajor{x, r->cin, M, 0}, (const half_t*)r->sc_w, r->cin, M, r->cout, r->cin, EpiConvF16{t3, r->cout, r->sc_b, nullptr}, st));
    skip = t3;
  }
  RC(conv3(t1, (const half_t*)r->c2_w, r->c2_b, skip, t2, ws.zeros, n, H, W, r->cout, r->cout, false, st, x_slabs ? ws.part : nullptr,
           x_slabs ? &out_slabs : nullptr));
  if (x_slabs) *x_slabs = out_slabs;
  half_t* o = t2;  // result in t2; rotate buffers so x is the result
  t2 = x;
  x = o;
  return LFM_OK;
}

// mid-block attention (diffusers Attention, 1 head of 512 channels over T tokens, residual): GroupNorm -> q, k, v -> softmax(q k^T / sqrt(C)) v
// -> to_out + x, all on the GEMM kernel.  x is replaced by the result (buffers rotate).
static int mid_attention(const float* at_g, const float* at_b, const void* q_w, const float* q_b, const void* k_w, const float* k_b, const void* v_w,
                         const float* v_b, const void* o_w, const float* o_b, half_t*& x, half_t*& t1, half_t*& t2, half_t*& t3, const VaeWs& ws,
                         int n, int T, hipStream_t st, int x_slabs = 0) {
  const int M = n * T, C = 512;
  RC(gn(x, t1, ws.stats, ws.part, at_g, at_b, n, T, C, false, st, x_slabs, ws.part_pairs));
  half_t* Qb = t2;                 // [M, C]
  half_t* Kb = t2 + (size_t)M * C;  // [M, C]
  half_t* Vt = t3;                 // [n, C, T]
  half_t* Pb = t3 + (size_t)M * C;  // [n, T, T]
  RC(launch_gemm_tn(ASrcRowMajor{t1, C, M, 0}, (const half_t*)q_w, C, M, C, C, EpiConvF16{Qb, C, q_b, nullptr}, st));
  RC(launch_gemm_tn(ASrcRowMajor{t1, C, M, 0}, (const half_t*)k_w, C, M, C, C, EpiConvF16{Kb, C, k_b, nullptr}, st));
  RC(launch_gemm_tn(ASrcRowMajor{t1, C, M, 0}, (const half_t*)v_w, C, M, C, C, EpiTransposeF16{Vt, v_b, T, C}, st));
  RC(launch_gemm_tn(ASrcRowMajor{Qb, C, T, 0}, Kb, C, T, T, C, EpiBatchF32{ws.S, T}, st, n, (long)T * C, (long)T * C, (long)T * T));
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv((long)n * T, 4)), dim3(256), 0, st, ws.S, Pb, (long)n * T, T, 1.4426950408889634f / sqrtf((float)C));
  LFM_CHECK_LAUNCH();
  half_t* Ob = t1;  // GN output is dead now
  RC(launch_gemm_tn(ASrcRowMajor{Pb, T, T, 0}, Vt, T, T, C, T, EpiBatchF16{Ob, C}, st, n, (long)T * T, (long)C * T, (long)T * C));
  RC(launch_gemm_tn(ASrcRowMajor{Ob, C, M, 0}, (const half_t*)o_w, C, M, C, C, EpiConvF16{t2, C, o_b, x}, st));
  half_t* o = t2;
  t2 = x;
  x = o;
  return LFM_OK;
}

extern "C" int lfm_vae_decode(const lfm_vae_weights* w, void* workspace, size_t workspace_bytes, const float* z, float* out, int N, int R,
                              int chunk, lfm_stream_t stream) {
  if (!w || !workspace || !z || !out) return LFM_ERR_ARG;
  if (N <= 0 || R <= 0 || (R % 8) || chunk <= 0) return LFM_ERR_SHAPE;
  const VaeWs ws = vae_carve(R, chunk, workspace);
  if (ws.total > workspace_bytes) return LFM_ERR_WORKSPACE;
  if ((uintptr_t)workspace & 255) return LFM_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (lfm_zero_async(ws.zeros, 256, st)) return LFM_ERR_LAUNCH;
  const int T = R * R;
  for (int n0 = 0; n0 < N; n0 += chunk) {
    const int n = (N - n0 < chunk) ? N - n0 : chunk;
    half_t *x = ws.b0, *t1 = ws.b1, *t2 = ws.b2, *t3 = ws.b3;
    {
      static lfm_device_mask set{0};
      const unsigned long long dbit = lfm_device_bit();
      if (lfm_device_todo(set, dbit)) {
        if (hipFuncSetAttribute((const void*)vae_conv_in_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 512 * 4) != hipSuccess)
          return LFM_ERR_LAUNCH;
        lfm_device_done(set, dbit);
      }
    }
    // batch 1 (--measure_time: run_sampling(1, ..), reference test_flow_latent.py:223-246): 1024 pixels are FOUR blocks of 256 pixels -- 183 us for 38 MFLOP
    // (profiles/r06_latency_mode.txt); 16 pixels per block put them on 64 CUs
    const int vci_ppb = (long)n * T >= 256L * VCI_PIX ? VCI_PIX : ((long)n * T >= 64L * VCI_PIX ? 64 : 16);
    hipLaunchKernelGGL(vae_conv_in_kernel, dim3(cdiv((long)n * T, vci_ppb)), dim3(256), 36 * 512 * 4, st, z + (long)n0 * 4 * T, w->pq_w, w->pq_b,
                       w->cin_w, w->cin_b, x, n, R, 512, vci_ppb);
    LFM_CHECK_LAUNCH();
    int H = R;
    int xs = 0;  // partial-sum slabs per image that the producer of x left in ws.part (0 = none: the GroupNorm runs its own statistics pass)
    RC(resnet(&w->mid[0], x, t1, t2, t3, ws, n, H, H, st, &xs));
    RC(mid_attention(w->at_g, w->at_b, w->q_w, w->q_b, w->k_w, w->k_b, w->v_w, w->v_b, w->o_w, w->o_b, x, t1, t2, t3, ws, n, T, st, xs));
    xs = 0;
    RC(resnet(&w->mid[1], x, t1, t2, t3, ws, n, H, H, st, &xs));
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 3; ++j) RC(resnet(&w->up[i][j], x, t1, t2, t3, ws, n, H, H, st, &xs));
      if (i < 3) {
        const int C = w->up[i][2].cout;
        H *= 2;
        RC(conv3(x, (const half_t*)w->ups_w[i], w->ups_b[i], nullptr, t1, ws.zeros, n, H, H, C, C, true, st, ws.part, &xs));
        half_t* o = t1;
        t1 = x;
        x = o;
      }
    }
    RC(gn(x, t1, ws.stats, ws.part, w->no_g, w->no_b, n, H * H, 128, true, st, xs, ws.part_pairs));
    const int M = n * H * H;
    {
      const EpiConvOutNCHW eo{out + (long)n0 * 3 * H * H, w->cout_b, H * H};
      int rc = 1;  // flag 8388608: the implicit GEMM (A/B)
      if (lfm_gemm_selected() == 0 && !(lfm_gemm_debug_flags() & 8388608)) rc = launch_conv3x3_halo_out(t1, ws.zeros, (const half_t*)w->cout_w, n, H, H, 128, eo, st);
      if (rc == 1) rc = launch_gemm_tn(ASrcConv3x3<0>{t1, ws.zeros, H, H, 128, M, 0, 0}, (const half_t*)w->cout_w, 9L * 128, M, 4, 9 * 128, eo, st);
      RC(rc);
    }
  }
  return LFM_OK;
}


// ------------------------------------------------------------------ encoder (diffusers AutoencoderKL.encode, sd-vae-ft-mse config)
// Reference call sites: train_flow_latent.py:143 and downstream_tasks/test_flow_latent_inpainting.py:146
//   first_stage_model.encode(x).latent_dist.sample().mul_(scale_factor)
// moments[N,8,R,R] fp32 NCHW = quant_conv(Encoder(x)) for x[N,3,8R,8R] fp32 NCHW: channels 0..3 the mean, 4..7 the log-variance of the
// DiagonalGaussianDistribution (the sampling / clamping of the distribution is host code).  quant_conv (1x1, 8 -> 8) is folded into
// conv_out's weights by the caller (exact algebra).  Same workspace as the decoder (lfm_vae_workspace_bytes(R, chunk)).
extern "C" int lfm_conv3x3_in_f32(const float* x_nchw, const float* w, const float* bias, void* out_nhwc, int N, int H, int W, int Cin, int Cout,
                                  lfm_stream_t stream);

static int conv_down(const half_t* in, const half_t* w, const float* b, half_t* out, const half_t* zeros, int n, int Ho, int Wo, int C, hipStream_t st) {
  if (C % 64) return LFM_ERR_SHAPE;
  const int M = n * Ho * Wo;
  return launch_gemm_auto(ASrcConvDown{in, zeros, Ho, Wo, C, M, 0, 0}, w, 9L * C, M, C, 9 * C, EpiConvF16{out, C, b, nullptr}, st);
}

extern "C" int lfm_vae_encode(const lfm_vae_enc_weights* w, void* workspace, size_t workspace_bytes, const float* x, float* moments, int N, int R,
                              int chunk, lfm_stream_t stream) {
  if (!w || !workspace || !x || !moments) return LFM_ERR_ARG;
  if (N <= 0 || R <= 0 || (R % 8) || chunk <= 0) return LFM_ERR_SHAPE;
  const VaeWs ws = vae_carve(R, chunk, workspace);
  if (ws.total > workspace_bytes) return LFM_ERR_WORKSPACE;
  if ((uintptr_t)workspace & 255) return LFM_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (lfm_zero_async(ws.zeros, 256, st)) return LFM_ERR_LAUNCH;
  const int S = 8 * R;
  for (int n0 = 0; n0 < N; n0 += chunk) {
    const int n = (N - n0 < chunk) ? N - n0 : chunk;
    half_t *h = ws.b0, *t1 = ws.b1, *t2 = ws.b2, *t3 = ws.b3;
    RC(lfm_conv3x3_in_f32(x + (long)n0 * 3 * S * S, w->cin_w, w->cin_b, h, n, S, S, 3, 128, stream));
    int H = S;
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 2; ++j) RC(resnet(&w->down[i][j], h, t1, t2, t3, ws, n, H, H, st));
      if (i < 3) {
        const int C = w->down[i][1].cout;
        H /= 2;
        RC(conv_down(h, (const half_t*)w->ds_w[i], w->ds_b[i], t1, ws.zeros, n, H, H, C, st));
        half_t* o = t1;
        t1 = h;
        h = o;
      }
    }
    RC(resnet(&w->mid[0], h, t1, t2, t3, ws, n, H, H, st));
    RC(mid_attention(w->at_g, w->at_b, w->q_w, w->q_b, w->k_w, w->k_b, w->v_w, w->v_b, w->o_w, w->o_b, h, t1, t2, t3, ws, n, H * H, st));
    RC(resnet(&w->mid[1], h, t1, t2, t3, ws, n, H, H, st));
    RC(gn(h, t1, ws.stats, ws.part, w->no_g, w->no_b, n, H * H, 512, true, st, 0, ws.part_pairs));
    const int M = n * H * H;
    RC(launch_gemm_tn(ASrcConv3x3<0>{t1, ws.zeros, H, H, 512, M, 0, 0}, (const half_t*)w->cout_w, 9L * 512, M, 8, 9 * 512,
                      EpiMomentsNCHW{moments + (long)n0 * 8 * H * H, w->cout_b, H * H, 8}, st));
  }
  return LFM_OK;
}

// images: u8 NHWC = trunc(clamp((x + 1) / 2, 0, 1) * 255)   (test_flow_latent_ddp.py:131-135), or with ROUND the single-process
// script's torchvision.utils.save_image conversion  trunc(clamp(.., 0, 1) * 255 + 0.5)  (test_flow_latent.py:264-269,297)
template <bool ROUND>
__global__ void to_uint8_nhwc_kernel(const float* __restrict__ x, uint8_t* __restrict__ o, int HW, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over N*HW pixels
  if (i >= total) return;
  const long n = i / HW, p = i - n * HW;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = (x[(n * 3 + c) * HW + p] + 1.0f) * 0.5f;
    v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;
    if (ROUND) v = fminf(v + 0.5f, 255.0f);
    o[i * 3 + c] = (uint8_t)v;
  }
}

extern "C" int lfm_images_to_uint8_mode(const float* x, uint8_t* out, int N, int H, int W, int rounding, lfm_stream_t stream) {
  if (!x || !out) return LFM_ERR_ARG;
  if (N <= 0 || H <= 0 || W <= 0) return LFM_ERR_SHAPE;
  const long total = (long)N * H * W;
  if (rounding) hipLaunchKernelGGL(to_uint8_nhwc_kernel<true>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, H * W, total);
  else hipLaunchKernelGGL(to_uint8_nhwc_kernel<false>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, H * W, total);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
extern "C" int lfm_images_to_uint8(const float* x, uint8_t* out, int N, int H, int W, lfm_stream_t stream) {
  return lfm_images_to_uint8_mode(x, out, N, H, W, 0, stream);
}
