// Shared pieces of the f16x3 kernels (conv3x3_f16x3.hip, conv3x3_f16x3_v3.hip): operand scales, the fp32 -> 2 x fp16
// split, the maxima format and the argument block of the forward / input-gradient convolution.
#pragma once
#include "common.h"

namespace f16x3 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kW = 32, kPW = 34, CK = 16, TROWS = 4;
constexpr int PIXB = 80;                        // bytes per patch pixel: 2 planes x 32 B + 16 pad ((PIXB/16) odd)
constexpr int NB = 80;                          // bytes per cout row of a weight tile
constexpr int PATCH_B = (TROWS + 2) * kPW * PIXB;   // 16320
constexpr int BN = 128;
constexpr int WT_B = BN * NB;                   // 10240
constexpr int SMEM_B = PATCH_B + 3 * WT_B;      // 47040 (dynamic shared memory): 1 patch + 3-deep weight ring

// ---- power-of-two scales from an absolute maximum given as fp32 bits
// s = 2^(140 - e) with e the (clamped) biased exponent of the maximum: max * s in [2^13, 2^14).  inv = 1 / s.
__device__ __forceinline__ void scale_of(unsigned maxbits, float& s, float& inv) {
  int e = (int)((maxbits >> 23) & 255u);
  e = e < 14 ? 14 : (e > 254 ? 254 : e);
  s = __uint_as_float((unsigned)(267 - e) << 23);
  inv = __uint_as_float((unsigned)(e - 13) << 23);
}

// Maxima arrays hold kMaxParts partial maxima per row (fp32 bit patterns; producers write their partials without
// atomics or a zero-fill pass, unused entries are 0); consumers take the maximum of the 16 entries of a row.
constexpr int kMaxParts = 16;
__device__ __forceinline__ unsigned row_max16(const unsigned* __restrict__ m, int row) {
  const unsigned* r = m + (size_t)row * kMaxParts;
  unsigned v = 0;
#pragma unroll
  for (int i = 0; i < kMaxParts; ++i) v = max(v, r[i]);
  return v;
}

__device__ __forceinline__ void split2(float vs, _Float16& h, _Float16& l) {
  h = (_Float16)vs;
  l = (_Float16)(vs - (float)h);
}

struct ConvArgsH {
  const float* x;            // [B,H,32,C] fp32
  const unsigned* xmax;      // [B] fp32 bits of max|x[b]|
  const unsigned char* wp;   // packed weights [9][C/16][N][2][16] fp16 (scaled)
  const unsigned* wmax;      // [1] fp32 bits of max|w|
  const float* bias; const float* cbias; const float* res;
  float* y;
  int B, H, C, N, cbias_mode;
  unsigned long long* stamps;   // dev-only (mulan_set_debug_buffer)
  unsigned char* xs;            // optional by-product: the split planes of x, [B][C/16][H*W][plane][16] fp16
  unsigned* ymax;               // optional by-product: [B][16] partial maxima of |y| (mulan_absmax_rows format)
  const unsigned char* xplanes; // optional: the input as split planes (format of xs, scaled by xmax) instead of x
  // GroupNorm-fed instantiation (conv3x3_f16x3_v3.hip, GNF): x is the fp32 INPUT of the GroupNorm (+ SiLU) in front of the
  // convolution, x2 (optional) the second half of a virtual channel concat [x | x2] of equal widths; the patch fill
  // normalises with gn_mean / gn_rstd [B, gn_groups] and gn_gamma / gn_beta [C]; xmax is the bound mulan_groupnorm_stats left
  const float* x2; const float* gn_mean; const float* gn_rstd; const float* gn_gamma; const float* gn_beta;
  int gn_act, gn_groups;
  // statistics handed from convolution to convolution (forward-only chains): ystats (optional output) receives this
  // launch's partial sums of y and y^2 per (image, 8-row tile, channel quad): [B][H / 8][N / 4][2]; xstats / xstats2
  // (optional inputs, same layout for x / x2) replace gn_mean / gn_rstd as the source of the statistics -- the block
  // forms mean / rstd (and the bound) itself, and block (row tile 0, cout block 0) writes them to gn_mean_out /
  // gn_rstd_out / xmax_out for a later backward pass
  float* ystats; const float* xstats; const float* xstats2; float* gn_mean_out; float* gn_rstd_out; unsigned* xmax_out;
  float gn_eps;
  int pair_cols;                // (set by the launcher) the cout blocks of a pixel tile run on one XCD, see the kernel
  int xstats_tiles;             // row tiles per image in xstats / xstats2 (the PRODUCER's tile height: 4, 8 or 16; 0 = 4)
  int alone;                    // the caller vouches that no other stream's kernels share the chip with this launch
                                // (forward passes, evaluators): small launches may then take a whole CU's LDS per block
};

// sigmoid on the hardware exp2 and reciprocal: the expression of groupnorm.hip's sigmoid_fast, bit for bit
__device__ __forceinline__ float sigmoid_hw(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int kBufWord3 = 0x00020000;           // raw buffer resource (no swizzle); out-of-range lanes are dropped / read 0


}  // namespace f16x3

// conv3x3_f16x3_v3.hip: 16x16x32 MFMA, two co-resident blocks per CU; returns a hipError_t as int
int mulan_launch_conv3x3_f16x3_v3(const f16x3::ConvArgsH& a, hipStream_t stream);
bool mulan_conv3x3_f16x3_v3_eligible(int H, int C, int N);
// image rows per block the launcher will use for this launch (8, 4 or 2): ystats has H / rows row tiles per image
int mulan_conv3x3_f16x3_v3_tile_rows(int B, int H, int N, bool with_ymax);
// 1 or 2: k-split groups per block the launcher will use (conv3x3_f16x3_v3.hip)
int mulan_conv3x3_f16x3_v3_ksplit(int B, int H, int C, int N, bool with_ymax, int alone);
