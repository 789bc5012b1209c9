// Halo-tiled direct 3x3 convolution (pad 1, NHWC fp16) for 128-wide output-channel blocks on v_mfma_f32_16x16x32_f16.
//
// Why (profiles/r03_final_bench_kernel_stats.csv): as an implicit GEMM the six 128-channel convolutions of the VAE decoder at full resolution
// run on the 256x128 kernel at 630-750 TFLOP/s while the 256-channel ones reach 1500 on the 256x256 kernel: with N = 128 every staged byte of
// the A operand meets only 128 output columns, and the implicit-GEMM A operand stages every input pixel NINE times (once per tap) -- 85 flop
// per byte staged into the LDS.  The convolution itself needs each input pixel once per tile plus a one-pixel rim:
//
//   tile        16 x 16 output pixels x 128 output channels per workgroup (4 waves, 2 x 2: a wave owns 8 tile rows x 64 channels = 8 x 4
//               accumulator tiles of 16 pixels x 16 channels; the 16 pixels of an MFMA tile are one row segment of the image);
//   K order     input channels in quarters of 32 (outer), the nine taps (inner): one K-step of the 16x16x32 MFMA per (quarter, tap) "slab";
//   halo        the 18 x 18 input pixels of a quarter = 324 x 64 B, staged ONCE per quarter by LDS-DMA (zero page beyond the image border) and
//               double-buffered over quarters: quarter h+1 lands while the nine taps of quarter h multiply.  A tap is a shift of the fragment
//               read address by (dy * 18 + dx) pixels -- an immediate offset, no data movement;
//   weights     one slab = 128 output channels x 32 input channels = 8 KiB, three-deep LDS-DMA ring (two slabs in flight);
//   LDS         2 x 21 KiB + 3 x 8 KiB = 66 KiB: TWO workgroups per CU, each SIMD hosting one wave of either, so one workgroup's prologue
//               and epilogue run under the other's MFMAs (the reason for the 256x128 kernel, gemm256n_kernel.h);
//   traffic     83 KiB of halo + 288 KiB of weights per 75.5 MFLOP tile = 200 flop per staged byte (2.35 x the implicit GEMM), and 0.08
//               LDS-DMA instructions per MFMA instead of 0.19 (256x128 implicit GEMM) / 0.125 (256x256).  The main-loop rates of the three kernels
//               follow that ratio (~1.5, ~1.2, ~0.9 PFLOP/s): staging each input byte once is what buys the rate.
// Pixel rows and weight rows are 64 B (4 chunks of 16 B); chunk c of row p sits at c ^ ((p >> 1) & 2) -- conflict-free for ds_read_b128 of
// 16 consecutive rows starting at ANY row (searched exhaustively over the instruction's lane groups: the tap shift moves the start row).
// The accumulator map and the row-major LDS-transposed hand-over to the epilogue are those of gemm256h_kernel.h (shared Epi interface:
// load / store8, and finish_slab for the GroupNorm partial sums).
// Measured (tools/conv_probe.py, profiles/r03_halo_conv_probe.txt): 1.0-1.4 PFLOP/s on every 3x3 convolution of the VAE decoder and the ADM UNet
// against 0.65-1.2 for the implicit GEMM.  Measured and not kept: s_setprio around the MFMA block (+-1 %).
#pragma once
#include "gemm256h_kernel.h"

#define CH_HALO_PIX_PAD 336                      // 18 * 18 = 324 pixels, padded to 21 wave-wide DMA instructions (16 pixels each)
#define CH_HALO_BYTES (CH_HALO_PIX_PAD * 64)     // 21504
#define CH_W_BYTES (128 * 64)                    // 8192
#define CH_RING_OFF (2 * CH_HALO_BYTES)          // 43008
#define CH_LDS_BYTES (CH_RING_OFF + 3 * CH_W_BYTES)  // 67584

template <int V>
struct ch_ic {
  static constexpr int value = V;
};

template <class Epi, class = void>
struct epi_has_finish_slab {
  static constexpr bool value = false;
};
template <class Epi>
struct epi_has_finish_slab<Epi, decltype((void)&Epi::finish_slab)> {
  static constexpr bool value = true;
};

// UPS = 1: the convolution runs on the nearest-2x upsampled image (H, W are the upsampled = output size) without materialising it: only the
// halo's source addresses change.  nb = Cout / 128 output-channel blocks per tile.
template <class Epi, int UPS>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(const half_t* __restrict__ in, const half_t* __restrict__ zeros,
                                                               const half_t* __restrict__ Wt, long ldw, int H, int W, int Cin, int tiles_x,
                                                               int tiles_per_img, int total_wg, int nb, Epi epi, int stag) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
#ifdef LFM_MEASURE
  lfm_stagger_start(stag);
#endif
  // consecutive workgroup ids go round the 8 XCDs: give every XCD a contiguous range of (tile, channel block) pairs, channel block fastest, so
  // that the channel blocks of a tile (same halo) and neighbouring tiles (shared rims) meet in one L2
  int l = blockIdx.x;
  if ((total_wg & 7) == 0) l = (l & 7) * (total_wg >> 3) + (l >> 3);
  const int t = l / nb, n0 = (l - t * nb) * 128;
  const int img = t / tiles_per_img, rem = t - img * tiles_per_img, ty = rem / tiles_x, tx = rem - ty * tiles_x;
  const int y0 = ty * 16, x0 = tx * 16;
  const int nh = Cin >> 5;
  const int Hs = H >> UPS, Ws = W >> UPS;
  const half_t* inimg = in + (size_t)img * Hs * Ws * Cin;

  // ---- halo DMA sources: slot = widx * 64 + lane -> pixel p = slot >> 2 of the 18-wide halo, physical chunk slot & 3
  int hoff[6];  // element offset inside the image (chunk swizzle included), -1 = zero page
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int slot = (k * 4 + wave) * 64 + lane, p = slot >> 2, ch = slot & 3;
    const int hy = p / 18, hx = p - hy * 18, iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = p < 324 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    hoff[k] = ok ? ((iy >> UPS) * Ws + (ix >> UPS)) * Cin + ((ch ^ ((p >> 1) & 2)) << 3) : -1;
  }
  auto issue_halo = [&](int h, int hb) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int widx = k * 4 + wave;
      if (widx < 21) {
        int o = hoff[k];
        asm volatile("" : "+v"(o));  // keep the six 32-bit offsets, not six hoisted 64-bit pointers (register budget)
        glds16(o >= 0 ? inimg + o + h * 32 : zeros, smem + hb * CH_HALO_BYTES + widx * 1024);
      }
    }
  };
  // ---- weight slab DMA sources: slot -> row = slot >> 2 (output channel), physical chunk slot & 3; (row >> 1) & 2 = (lane >> 3) & 2
  const half_t* wsrc[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) wsrc[k] = Wt + (long)(n0 + k * 64 + wave * 16 + (lane >> 2)) * ldw + (((lane & 3) ^ ((lane >> 3) & 2)) << 3);
  auto issue_w = [&](int h, int tap, int rs) {
    const int koff = tap * Cin + h * 32;
#pragma unroll
    for (int k = 0; k < 2; ++k) glds16(wsrc[k] + koff, smem + CH_RING_OFF + rs * CH_W_BYTES + (k * 4 + wave) * 1024);
  };

  // ---- fragment read addresses.  Pixel fragment of accumulator row-tile i at tap (dy, dx): halo pixel p = pl + c, pl = wm * 144 + (lane & 15),
  // c = (i + dy) * 18 + dx (compile time); byte address p * 64 + ((q ^ key(p)) << 4), q = lane >> 4.  key(p) is bit 2 of p = bit 2 of
  // (pl & 7) + (c & 7): eight per-lane bases, one per c & 7, and c * 64 goes into the instruction's immediate offset.
  const int r = lane & 15, q = lane >> 4, pl = wm * 144 + r;
  int abase[8];
#pragma unroll
  for (int c7 = 0; c7 < 8; ++c7) abase[c7] = pl * 64 + ((q ^ ((((pl & 7) + c7) >> 1) & 2)) << 4);
  const int wbase = (wn * 64 + r) * 64 + ((q ^ ((r >> 1) & 2)) << 4);

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  half8_t af[8], bf[4];
  auto lds_read = [&](half8_t& dst, int addr, auto OFFC) {
    constexpr int OFF = decltype(OFFC)::value;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
  };

#define CH_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
  // one slab = (quarter h in halo buffer HB, tap TAP, weights in ring slot TAP % 3): wait, barrier, issue the slab two ahead, 12 reads, 32 MFMAs
  auto slab = [&](auto HBC, auto TAPC, int h) {
    constexpr int HB = decltype(HBC)::value, TAP = decltype(TAPC)::value, RS = TAP % 3;
    const bool more_h = h + 1 < nh;
    // my DMAs of this slab's weights (and, at tap 0, of this quarter's halo) have landed; younger ones may fly: the next slab's weights (2)
    // and, for two slabs after a halo issue, the halo (5 or 6 per wave; 5 is the safe count)
    if (TAP == 8 && !more_h) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if ((TAP == 1 || TAP == 2) && more_h) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    G256_BARRIER();
    {
      constexpr int T2 = (TAP + 2) % 9;
      const int h2 = h + (TAP + 2 >= 9 ? 1 : 0);
      if (h2 < nh) issue_w(h2, T2, (RS + 2) % 3);
      if (TAP == 0 && more_h) issue_halo(h + 1, HB ^ 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int WOFF = CH_RING_OFF + RS * CH_W_BYTES;
    lds_read(bf[0], wbase, ch_ic<WOFF>{});
    lds_read(bf[1], wbase, ch_ic<WOFF + 1024>{});
    lds_read(bf[2], wbase, ch_ic<WOFF + 2048>{});
    lds_read(bf[3], wbase, ch_ic<WOFF + 3072>{});
    constexpr int DY = TAP / 3, DX = TAP % 3;
#define CH_AREAD(i)                                                               \
  {                                                                               \
    constexpr int C = ((i) + DY) * 18 + DX;                                       \
    lds_read(af[i], abase[C & 7], ch_ic<HB * CH_HALO_BYTES + C * 64>{});          \
  }
    CH_AREAD(0) CH_AREAD(1) CH_AREAD(2) CH_AREAD(3) CH_AREAD(4) CH_AREAD(5) CH_AREAD(6) CH_AREAD(7)
#undef CH_AREAD
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i == 0) CH_LGKM(7);
      else if (i == 1) CH_LGKM(6);
      else if (i == 2) CH_LGKM(5);
      else if (i == 3) CH_LGKM(4);
      else if (i == 4) CH_LGKM(3);
      else if (i == 5) CH_LGKM(2);
      else if (i == 6) CH_LGKM(1);
      else CH_LGKM(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#undef CH_LGKM

  // ---- prologue: halo of quarter 0, weight slabs 0 and 1
  issue_halo(0, 0);
  issue_w(0, 0, 0);
  issue_w(0, 1, 1);
  for (int h = 0; h < nh; h += 2) {  // Cin % 64 == 0: quarters come in pairs (halo buffer 0, 1)
    slab(ch_ic<0>{}, ch_ic<0>{}, h);
    slab(ch_ic<0>{}, ch_ic<1>{}, h);
    slab(ch_ic<0>{}, ch_ic<2>{}, h);
    slab(ch_ic<0>{}, ch_ic<3>{}, h);
    slab(ch_ic<0>{}, ch_ic<4>{}, h);
    slab(ch_ic<0>{}, ch_ic<5>{}, h);
    slab(ch_ic<0>{}, ch_ic<6>{}, h);
    slab(ch_ic<0>{}, ch_ic<7>{}, h);
    slab(ch_ic<0>{}, ch_ic<8>{}, h);
    slab(ch_ic<1>{}, ch_ic<0>{}, h + 1);
    slab(ch_ic<1>{}, ch_ic<1>{}, h + 1);
    slab(ch_ic<1>{}, ch_ic<2>{}, h + 1);
    slab(ch_ic<1>{}, ch_ic<3>{}, h + 1);
    slab(ch_ic<1>{}, ch_ic<4>{}, h + 1);
    slab(ch_ic<1>{}, ch_ic<5>{}, h + 1);
    slab(ch_ic<1>{}, ch_ic<6>{}, h + 1);
    slab(ch_ic<1>{}, ch_ic<7>{}, h + 1);
    slab(ch_ic<1>{}, ch_ic<8>{}, h + 1);
  }
  G256_BARRIER();  // every wave's last fragment reads are retired: the LDS is free for the epilogue scratch (4 x 8.5 KiB)

  // ---- epilogue: 32 pixels (two row segments) x 64 channels at a time through a per-wave scratch [32][64 + pad] fp32; then lane (rrow = lane >> 3,
  // rcol = lane & 7) owns 8 consecutive channels of pixel rows rrow, rrow + 8, ..: 16-byte stores, full 128-byte runs per pixel
  char* scr = smem + wave * (32 * 272);
  const int rrow = lane >> 3, rcol = lane & 7, n = n0 + wn * 64 + rcol * 8;
#pragma unroll
  for (int I = 0; I < 4; ++I) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4_t*)(scr + (h2 * 16 + r) * 272 + (j * 16 + q * 4) * 4) = acc[2 * I + h2][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 lo[4], hi[4];
    int m[4];
    typename Epi::Aux al[4], ah[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int rb = ps * 8 + rrow;  // row of the 32-pixel block: segment rb >> 4, pixel rb & 15
      lo[ps] = *(const f32x4*)(scr + rb * 272 + rcol * 32);
      hi[ps] = *(const f32x4*)(scr + rb * 272 + rcol * 32 + 16);
      m[ps] = (img * H + y0 + wm * 8 + 2 * I + (rb >> 4)) * W + x0 + (rb & 15);
      al[ps] = epi.load(m[ps], n);
      ah[ps] = epi.load(m[ps], n + 4);
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) epi.store8(m[ps], n, lo[ps], hi[ps], al[ps], ah[ps]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if constexpr (epi_has_finish_slab<Epi>::value) epi.finish_slab(img, rem * 2 + wm, n0 + wn * 64, lane);
}

// out[n, H, W, Cout] = conv3x3(in) through `epi` (bias, residual, statistics); in = [n, H, W, Cin], or [n, H/2, W/2, Cin] with UPS = 1 (nearest-2x
// upsample fused).  Returns 1 when the shape is not this kernel's (H, W multiples of 16; Cin a multiple of 64; Cout a multiple of 128; at least
// one workgroup per CU), so that the caller takes the implicit GEMM.
template <int UPS, class Epi>
static inline int launch_conv3x3_halo(const half_t* in, const half_t* zeros, const half_t* Wt, int n, int H, int W, int Cin, int Cout, const Epi& epi,
                                      hipStream_t st) {
  if (n <= 0 || (H & 15) || (W & 15) || (Cin & 63) || (Cout & 127)) return 1;
  if ((long)n * H * W >= (1L << 31) || (long)H * W * Cin >= (1L << 31)) return 1;
  const int tiles_x = W / 16, tiles_per_img = tiles_x * (H / 16), nb = Cout / 128;
  const long total = (long)n * tiles_per_img * nb;
  if ((total < 256 && !(lfm_gemm_debug_flags() & 16777216)) || total >= (1L << 31)) return 1;  // flag 16777216: small problems too (parity tests)
  if (((uintptr_t)in | (uintptr_t)Wt | (uintptr_t)zeros) & 15) return LFM_ERR_ALIGN;
  static lfm_device_mask attr_set{0};  // one bit per device: the attribute is per (function, device)
  const unsigned long long dbit = lfm_device_bit();
  if (lfm_device_todo(attr_set, dbit)) {
    if (hipFuncSetAttribute((const void*)conv3x3_halo_kernel<Epi, UPS>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES) != hipSuccess)
      return LFM_ERR_LAUNCH;
    lfm_device_done(attr_set, dbit);
  }
  hipLaunchKernelGGL((conv3x3_halo_kernel<Epi, UPS>), dim3((unsigned)total), dim3(256), CH_LDS_BYTES, st, in, zeros, Wt, 9L * Cin, H, W, Cin, tiles_x,
                     tiles_per_img, (int)total, nb, epi, lfm_stagger_ticks());
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}

// ---- the same halo tiling for an OUTPUT convolution with at most four channels (the VAE decoder's conv_out, the UNets' out conv): fp16 NHWC in,
// `epi.store(m, 0, f32x4 of channels 0..3, aux)` out.  As an implicit GEMM on the 128x128 kernel these layers compute 128 columns to keep 4 and
// stage every input pixel nine times (1.2 ms per VAE decode of 64 images, 80 us per UNet evaluation); the work is one read of the input:
//   tile     16 x 16 pixels per 4-wave workgroup, a wave owns 4 rows; ONE accumulator tile column (16 channels, 4 real) -- D[channel][pixel], so the lanes
//            of k-group 0 end up with the four channels of one pixel each and store straight from registers;
//   K order  32-channel quarters; per quarter the halo (21 KiB) AND all nine taps of the weights (9 x 16 rows x 64 B, rows 4..15 from the zero page)
//            are staged, double-buffered over quarters: one barrier per quarter, 36 MFMAs per wave between barriers;
//   LDS      2 x (21 KiB + 9 KiB) = 60 KiB: two workgroups per CU.  Bound: the HBM read of the input.
// w4: fp16 [4][9 * Cin] (k = tap * Cin + ci), rows beyond the real channel count zero.  Cin % 32 == 0, H % 16 == W % 16 == 0.
#define CHO_W_BYTES (9 * 16 * 64)                      // 9216
#define CHO_BUF_BYTES (CH_HALO_BYTES + CHO_W_BYTES)    // 30720
#define CHO_LDS_BYTES (2 * CHO_BUF_BYTES)              // 61440

template <class Epi>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_out_kernel(const half_t* __restrict__ in, const half_t* __restrict__ zeros,
                                                                   const half_t* __restrict__ w4, int H, int W, int Cin, int tiles_x, int tiles_per_img,
                                                                   int total_tiles, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x;
  if ((total_tiles & 7) == 0) t = (t & 7) * (total_tiles >> 3) + (t >> 3);
  const int img = t / tiles_per_img, rem = t - img * tiles_per_img, ty = rem / tiles_x, tx = rem - ty * tiles_x;
  const int y0 = ty * 16, x0 = tx * 16;
  const int nh = Cin >> 5;
  const half_t* inimg = in + (size_t)img * H * W * Cin;
  int hoff[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int slot = (k * 4 + wave) * 64 + lane, p = slot >> 2, ch = slot & 3;
    const int hy = p / 18, hx = p - hy * 18, iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = p < 324 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    hoff[k] = ok ? (iy * W + ix) * Cin + ((ch ^ ((p >> 1) & 2)) << 3) : -1;
  }
  // weights of a quarter: slot = widx * 64 + lane -> (tap, row) = slot >> 2, physical chunk slot & 3; 576 slots = 9 wave-wide DMAs
  int woff[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int slot = (k * 4 + wave) * 64 + lane, tr = slot >> 2, ch = slot & 3, tap = tr >> 4, row = tr & 15;
    woff[k] = (row < 4 && tap < 9) ? row * 9 * Cin + tap * Cin + ((ch ^ ((row >> 1) & 2)) << 3) : -1;
  }
  auto issue = [&](int h, int b) {
    char* base = smem + b * CHO_BUF_BYTES;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int widx = k * 4 + wave;
      if (widx < 21) {
        int o = hoff[k];
        asm volatile("" : "+v"(o));
        glds16(o >= 0 ? inimg + o + h * 32 : zeros, base + widx * 1024);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int widx = k * 4 + wave;
      if (widx < 9) {
        int o = woff[k];
        asm volatile("" : "+v"(o));
        glds16(o >= 0 ? w4 + o + h * 32 : zeros, base + CH_HALO_BYTES + widx * 1024);
      }
    }
  };
  const int r = lane & 15, q = lane >> 4, pl = wave * 72 + r;  // wave w owns tile rows 4 w .. 4 w + 3: halo pixel (4 w + i + dy) * 18 + dx + r
  const int wfrag = r * 64 + ((q ^ ((r >> 1) & 2)) << 4);
  f32x4_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  issue(0, 0);
  for (int h = 0; h < nh; ++h) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my share of quarter h has landed
    G256_BARRIER();                                   // everyone's has; and every wave is done reading the other buffer (quarter h - 1)
    if (h + 1 < nh) issue(h + 1, (h + 1) & 1);
    const char* hb = smem + (h & 1) * CHO_BUF_BYTES;
    const char* wb = hb + CH_HALO_BYTES;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const half8_t wf = *(const half8_t*)(wb + tap * 1024 + wfrag);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = pl + (i + tap / 3) * 18 + tap % 3;
        const half8_t af = *(const half8_t*)(hb + p * 64 + ((q ^ ((p >> 1) & 2)) << 4));
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, af, acc[i], 0, 0, 0);
      }
    }
  }
  if (q == 0) {  // D[channel 4 q + e][pixel r]: k-group 0 holds channels 0..3
    const typename Epi::Aux aux = epi.load(0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = (img * H + y0 + wave * 4 + i) * W + x0 + r;
      epi.store(m, 0, (f32x4){acc[i].x, acc[i].y, acc[i].z, acc[i].w}, aux);
    }
  }
}

// returns 1 when the shape is not this kernel's (the caller takes the implicit GEMM)
template <class Epi>
static inline int launch_conv3x3_halo_out(const half_t* in, const half_t* zeros, const half_t* w4, int n, int H, int W, int Cin, const Epi& epi,
                                          hipStream_t st) {
  if (n <= 0 || (H & 15) || (W & 15) || (Cin & 31)) return 1;
  if ((long)n * H * W >= (1L << 31) || (long)H * W * Cin >= (1L << 31)) return 1;
  const int tiles_x = W / 16, tiles_per_img = tiles_x * (H / 16);
  const long total = (long)n * tiles_per_img;
  if ((total < 256 && !(lfm_gemm_debug_flags() & 16777216)) || total >= (1L << 31)) return 1;
  if (((uintptr_t)in | (uintptr_t)w4 | (uintptr_t)zeros) & 15) return LFM_ERR_ALIGN;
  static lfm_device_mask attr_set{0};
  const unsigned long long dbit = lfm_device_bit();
  if (lfm_device_todo(attr_set, dbit)) {
    if (hipFuncSetAttribute((const void*)conv3x3_halo_out_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, CHO_LDS_BYTES) != hipSuccess)
      return LFM_ERR_LAUNCH;
    lfm_device_done(attr_set, dbit);
  }
  hipLaunchKernelGGL((conv3x3_halo_out_kernel<Epi>), dim3((unsigned)total), dim3(256), CHO_LDS_BYTES, st, in, zeros, w4, H, W, Cin, tiles_x, tiles_per_img,
                     (int)total, epi);
  LFM_CHECK_LAUNCH();
  return LFM_OK;
}
