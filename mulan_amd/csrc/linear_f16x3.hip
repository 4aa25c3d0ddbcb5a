// Per-pixel dense layers (1x1 convolutions) with fp32-equivalent products on the fp16 matrix cores ("f16x3", the
// scheme of conv3x3_f16x3.hip: two power-of-two-scaled fp16 pieces per fp32 operand, three MFMA passes).
//     y[M, N1 | N2] = [x1 | x2][M, K1 + K2] @ W[K1 + K2, N1 + N2] + bias + res
// over a virtual channel concat on the input side (nin_shortcut on concat[h, skip], ldm/model_vdm.py:369,652-653)
// and a split on the output side (its input gradient), so neither concat nor split is ever materialised, and for
// q / k / v / proj_out of AttnBlock (ldm/model_vdm.py:676-685).  These layers are memory bound (K = 128 .. 256):
// one pass over the inputs, one over the output.
//
// Block = 128 rows x 128 output columns, wave = 64 x 64 (2 x 2 MFMA tiles, 64 accumulator registers, two to three
// blocks per CU so that one block's loads overlap another's MFMAs).  A stage is 32 input channels: the activations
// are split while they are staged into LDS (double buffered, one barrier per stage), the weight fragments are read
// straight from the packed global tensor [K/16][N][plane][16] (L2 resident) one stage ahead.
// Rows are pixels of images of `rows_per_img` rows; the activation scale is per image (maxima in the 16-partials
// format of mulan_absmax_rows).
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxParts = 16;
constexpr int TM = 128, TN = 128, SK = 32;
constexpr int ROWB = 144;                      // LDS bytes per row of a stage: 2 planes x 64 B + 16 pad
constexpr int STAGE_B = TM * ROWB;             // 18432
constexpr int LIN_SMEM = 2 * STAGE_B;          // 36864 (the epilogue staging tile, 34816 B, reuses it)

__device__ __forceinline__ unsigned row_max16(const unsigned* __restrict__ m, int row) {
  const unsigned* r = m + (size_t)row * kMaxParts;
  unsigned v = 0;
#pragma unroll
  for (int i = 0; i < kMaxParts; ++i) v = max(v, r[i]);
  return v;
}

__device__ __forceinline__ void scale_of(unsigned maxbits, float& s, float& inv) {   // see conv3x3_f16x3.hip
  int e = (int)((maxbits >> 23) & 255u);
  e = e < 14 ? 14 : (e > 254 ? 254 : e);
  s = __uint_as_float((unsigned)(267 - e) << 23);
  inv = __uint_as_float((unsigned)(e - 13) << 23);
}

__device__ __forceinline__ void split2(float vs, _Float16& h, _Float16& l) {
  h = (_Float16)vs;
  l = (_Float16)(vs - (float)h);
}

struct LinArgs {
  const float* x1; const float* x2;            // [M, K1], [M, K2] (x2 may be null, K2 = 0)
  const unsigned* x1max; const unsigned* x2max;
  const unsigned char* wp; const unsigned* wmax;   // packed weights [K/16][N][2][16] fp16 (scaled), N = N1 + N2
  const float* bias; const float* res;         // bias [N]; res like y1 (single-output calls only)
  float* y1; float* y2;                        // [M, N1], [M, N2]
  int M, K1, K2, N1, N2, rows_per_img;
  unsigned char* xs;                           // optional by-product: split planes of [x1 | x2], [B][K/16][rows][2][16] fp16
  size_t wp_img_stride;                        // bytes between the packed weights of consecutive images (0: shared)
  int wmax_per_img;                            // wmax is [B][16] (one maximum per image) instead of [1][16]
};

typedef int i32x2 __attribute__((ext_vector_type(2)));
constexpr int kBufWord3 = 0x00020000;          // raw buffer resource; out-of-range lanes are dropped

// NT = 2: a block is 128 rows x 128 columns (wave: 64 x 64).  NT = 4: 128 rows x 256 columns (wave: 64 x 128) for layers
// with at least 256 output columns -- the rows are read, split and staged once for 256 columns instead of twice (E = 256:
// the kernel is bound by that work, not by HBM; E = 128: the two-output input gradient of nin_shortcut reads dy once).
// The weight fragments of k step t + 1 are fetched while step t is multiplied (a ring of two: 64 registers at NT = 4).
// MT (round 5): row tiles per wave.  MT = 2: 128 rows per block; MT = 1: 64 rows per block, for launches that would
// otherwise leave CUs without a block (M = 16 K rows at a 16-image sampling batch: 128 blocks of 128 rows).  Every output
// element is the same sum in the same order at either value.
template <int NT, int MT = 2>
__global__ __launch_bounds__(256, 2) void linear_f16x3_kernel(LinArgs p) {
  constexpr int TMB = 64 * MT;                   // rows per block
  constexpr int TNB = NT * 64;                   // columns per block
  __shared__ __attribute__((aligned(16))) unsigned char smem[LIN_SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int r0 = blockIdx.x * TMB, n0 = blockIdx.y * TNB;
  const int K = p.K1 + p.K2, N = p.N1 + p.N2;
  const int nst = K / SK;
  const int b = r0 / p.rows_per_img;
  unsigned mb = row_max16(p.x1max, b);
  if (p.x2) mb = max(mb, row_max16(p.x2max, b));
  float sx, inv_x, sw, inv_w;
  scale_of(mb, sx, inv_x);
  scale_of(row_max16(p.wmax, p.wmax_per_img ? b : 0), sw, inv_w);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // activation staging: TMB rows x 8 float4 per stage, 2 MT slots per thread (slot = tid + 256 i: quad = slot & 7)
  const int aq = tid & 7, arow = tid >> 3;                     // + 32 i rows
  f32x4 areg[2 * MT];
  auto gload_a = [&](int s) {
    const int c = s * SK;
    const float* src; int ld, cl;
    if (c < p.K1) { src = p.x1; ld = p.K1; cl = c; } else { src = p.x2; ld = p.K2; cl = c - p.K1; }
    const float* base = src + (size_t)(r0 + arow) * ld + cl + aq * 4;
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i) areg[i] = ld_stream4(base + (size_t)(32 * i) * ld);
  };
  // plane output window: this block's image (only the first column block writes; nothing when not requested)
  const __amdgpu_buffer_rsrc_t xs_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.xs + (size_t)b * (K / 16) * p.rows_per_img * 64, 0, (p.xs && blockIdx.y == 0) ? (K / 16) * p.rows_per_img * 64 : 0,
      kBufWord3);
  const int pix0 = r0 - b * p.rows_per_img + arow;
  auto store_a = [&](unsigned char* buf, int s) {
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i) {
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 h, l;
        split2(areg[i][e] * sx, h, l);
        hi[e] = h; lo[e] = l;
      }
      unsigned char* d = buf + (arow + 32 * i) * ROWB + aq * 8;
      *reinterpret_cast<f16x4*>(d) = hi;
      *reinterpret_cast<f16x4*>(d + 64) = lo;
      // the same quads go to the plane tensor (consumed by the weight gradient of this layer)
      const unsigned eo = (unsigned)((((2 * s + (aq >> 2)) * p.rows_per_img + pix0 + 32 * i) * 2) * 32 + (aq & 3) * 8);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, hi), xs_rsrc, eo, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, lo), xs_rsrc, eo + 32, 0, 0);
    }
  };
  // weight fragments of a stage: [k step j][n tile][plane]
  const unsigned char* bbase = p.wp + (size_t)b * p.wp_img_stride + (size_t)(n0 + wn * (NT * 32) + li) * 64 + lh * 16;
  const size_t chunk_stride = (size_t)N * 64;
  // weight fragments of k step t (16 input channels): [n tile][plane]
  auto gload_b = [&](f16x8 (&bs)[NT][2], int t) {
    const unsigned char* q = bbase + (size_t)t * chunk_stride;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) bs[nt][pl] = *reinterpret_cast<const f16x8*>(q + nt * 2048 + pl * 32);
  };

  f16x8 bring[2][NT][2];
  gload_a(0);
  gload_b(bring[0], 0);
  for (int s = 0; s < nst; ++s) {
    unsigned char* buf = smem + (s & 1) * STAGE_B;
    store_a(buf, s);
    __syncthreads();
    const bool more = s + 1 < nst;
    if (more) gload_a(s + 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = 2 * s + j;
      if (j == 0 || more) gload_b(bring[(j + 1) & 1], t + 1);    // the fragments of the next k step, while this one runs
      f16x8 af[MT][2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          af[mt][pl] = *reinterpret_cast<const f16x8*>(buf + (wm * (32 * MT) + mt * 32 + li) * ROWB + pl * 64 + j * 32 + lh * 16);
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        constexpr int PA[3] = {1, 0, 0};
        constexpr int PB[3] = {0, 1, 0};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][PA[term]], bring[j][nt][PB[term]], acc[mt][nt], 0,
                                                                 0, 0);
      }
    }
  }

  // epilogue: transposed through LDS so every lane moves float4s; scales divided out here
  const float* __restrict__ res = p.res;
  constexpr int TS = 64 + 4;
  float* stage = reinterpret_cast<float*>(smem) + wave * 32 * TS;
  const int c4 = lane & 15, prl = lane >> 4;
  __syncthreads();
#pragma unroll
  for (int nh = 0; nh < NT / 2; ++nh) {            // 64 columns of the wave's tile at a time (N1 % 128 == 0: one output each)
  const int ncol = n0 + wn * (NT * 32) + nh * 64;  // first column of this half in [y1 | y2]
  float* out; int ldo, col0;
  if (ncol < p.N1) { out = p.y1; ldo = p.N1; col0 = ncol; } else { out = p.y2; ldo = p.N2; col0 = ncol - p.N1; }
  float* __restrict__ yout = out;
  const int nb = c4 * 4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + ncol + nb);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[mfma32_row(r, lane) * TS + nt * 32 + li] = (acc[mt][nh * 2 + nt][r] * inv_x) * inv_w;
    const size_t rowbase = (size_t)(r0 + wm * (32 * MT) + mt * 32) * ldo + col0 + nb;
    f32x4 add[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) add[it] = bias4;
    if (res) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const f32x4 c = ld_stream4(res + rowbase + (size_t)(it * 4 + prl) * ldo);
        add[it][0] += c[0]; add[it][1] += c[1]; add[it][2] += c[2]; add[it][3] += c[3];
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's ds_writes have landed (tile is wave private)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(stage + (it * 4 + prl) * TS + c4 * 4);
      const f32x4 o = {a[0] + add[it][0], a[1] + add[it][1], a[2] + add[it][2], a[3] + add[it][3]};
      *reinterpret_cast<f32x4*>(yout + rowbase + (size_t)(it * 4 + prl) * ldo) = o;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  }
}

// wp[cc][o][plane][k] = split2( s_w * Wl[cc*16 + k][o] ),  Wl[kin][o] = w[kin][o] (w row-major [Kin, Nout]) or, with
// transpose, w[o][kin] (w row-major [Nout, Kin]: the weight of the input-gradient product dy @ w^T)
// blockIdx.y = image of a batched call (one [Kin, Nout] operand and one maximum per image)
__global__ void linear_pack_f16x3_kernel(const float* __restrict__ w, _Float16* __restrict__ wp,
                                         const unsigned* __restrict__ wmax, int Kin, int Nout, int transpose) {
  const size_t total = (size_t)Kin * Nout;
  w += (size_t)blockIdx.y * total;
  wp += (size_t)blockIdx.y * total * 2;
  float sw, inv_w;
  scale_of(row_max16(wmax, blockIdx.y), sw, inv_w);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % 16);
    const size_t r = i / 16;
    const int o = (int)(r % Nout);
    const int cc = (int)(r / Nout);
    const int kin = cc * 16 + k;
    const float v = transpose ? w[(size_t)o * Kin + kin] : w[(size_t)kin * Nout + o];
    _Float16 h, l;
    split2(v * sw, h, l);
    _Float16* dst = wp + (((size_t)cc * Nout + o) * 2) * 16 + k;
    dst[0] = h; dst[16] = l;
  }
}

}  // namespace

MULAN_API size_t mulan_linear_pack_f16x3_bytes(int K, int N) { return (size_t)K * N * 4; }

// Packs w for mulan_linear_f16x3.  transpose = 0: w is [K, N] (y = x @ w); transpose = 1: w is [N, K] and the packed
// operand is its transpose (dx = dy @ w^T with K = w's columns).  wmax[16] = mulan_absmax_rows(w, 1 row).
MULAN_API int mulan_linear_pack_f16x3(const float* w, void* wp, const unsigned* wmax, int K, int N, int transpose,
                                      hipStream_t stream) {
  if (K <= 0 || N <= 0 || K % 16 != 0 || !wmax) return (int)hipErrorInvalidValue;
  const size_t total = (size_t)K * N;
  const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  hipLaunchKernelGGL(linear_pack_f16x3_kernel, dim3(blocks), dim3(256), 0, stream, w, static_cast<_Float16*>(wp), wmax, K,
                     N, transpose);
  MULAN_CHECK_LAUNCH();
}

// y[M, N1 | N2] = [x1 | x2][M, K1 + K2] @ W + bias + res.  Rows are pixels, rows_per_img per image; x1max / x2max are
// the per-image maxima ([M / rows_per_img][16], mulan_absmax_rows format).  Needs M % 128 == 0, rows_per_img % 128 == 0,
// K1 % 32 == 0, K2 % 32 == 0, N1 % 128 == 0, N2 % 128 == 0 (x2 / y2 optional: K2 = 0 / N2 = 0); res only with N2 = 0.
// xs (optional, M * (K1 + K2) * 4 bytes): receives the split planes of [x1 | x2] (scaled with max(x1max, x2max) per
// image), the input format of mulan_linear_wgrad_f16x3_planes.
static int linear_f16x3_launch(const float* x1, const unsigned* x1max, const float* x2, const unsigned* x2max, int K1,
                               int K2, const void* wp, const unsigned* wmax, const float* bias, const float* res,
                               float* y1, float* y2, void* xs, int N1, int N2, int M, int rows_per_img,
                               size_t wp_img_stride, int wmax_per_img, hipStream_t stream) {
  if (M <= 0 || M % TM != 0 || rows_per_img <= 0 || rows_per_img % TM != 0 || M % rows_per_img != 0 || K1 <= 0 ||
      K1 % SK != 0 || K2 < 0 || K2 % SK != 0 || N1 <= 0 || N1 % TN != 0 || N2 < 0 || N2 % TN != 0 || !x1 || !x1max ||
      !wp || !wmax || !y1 || (K2 > 0 && (!x2 || !x2max)) || (N2 > 0 && (!y2 || res)))
    return (int)hipErrorInvalidValue;
  if (xs && (size_t)M * (K1 + K2) * 4 >= 0x80000000ull) return (int)hipErrorInvalidValue;
  LinArgs a{x1, K2 > 0 ? x2 : nullptr, x1max, x2max, static_cast<const unsigned char*>(wp), wmax, bias, res, y1, y2,
            M, K1, K2, N1, N2, rows_per_img, static_cast<unsigned char*>(xs), wp_img_stride, wmax_per_img};
  const bool wide = (N1 + N2) % 256 == 0 && g_mulan_tune[11] != 1;   // tune[11] = 1: dev A/B, 128-column blocks everywhere
  const int colblocks = wide ? (N1 + N2) / 256 : (N1 + N2) / TN;
  // fewer 128-row blocks than CUs: 64-row blocks (tune[28] = 1: dev A/B, 128 rows everywhere)
  const bool shortb = (M / TM) * colblocks < 256 && g_mulan_tune[28] != 1;
  if (wide) {
    if (shortb) hipLaunchKernelGGL((linear_f16x3_kernel<4, 1>), dim3(M / 64, colblocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((linear_f16x3_kernel<4, 2>), dim3(M / TM, colblocks), dim3(256), 0, stream, a);
  } else {
    if (shortb) hipLaunchKernelGGL((linear_f16x3_kernel<2, 1>), dim3(M / 64, colblocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((linear_f16x3_kernel<2, 2>), dim3(M / TM, colblocks), dim3(256), 0, stream, a);
  }
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_linear_f16x3(const float* x1, const unsigned* x1max, const float* x2, const unsigned* x2max, int K1,
                                 int K2, const void* wp, const unsigned* wmax, const float* bias, const float* res,
                                 float* y1, float* y2, void* xs, int N1, int N2, int M, int rows_per_img,
                                 hipStream_t stream) {
  return linear_f16x3_launch(x1, x1max, x2, x2max, K1, K2, wp, wmax, bias, res, y1, y2, xs, N1, N2, M, rows_per_img, 0, 0,
                             stream);
}

// Batched form for the attention products (ldm/model_vdm.py:775-796): image b (rows_per_img rows of x) is multiplied
// by its own operand, packed by mulan_linear_pack_f16x3_batched at wp + b * K * N * 4 bytes, with its own maximum
// wmax[b][16]:  y[b] = x[b] @ W[b]  (+ res).  xs as in mulan_linear_f16x3.
MULAN_API int mulan_linear_f16x3_batched(const float* x, const unsigned* xmax, int K, const void* wp,
                                         const unsigned* wmax, const float* res, float* y, void* xs, int N, int M,
                                         int rows_per_img, hipStream_t stream) {
  return linear_f16x3_launch(x, xmax, nullptr, nullptr, K, 0, wp, wmax, nullptr, res, y, nullptr, xs, N, 0, M,
                             rows_per_img, (size_t)K * N * 4, 1, stream);
}

// batch operands w[b] ([K, N], or [N, K] with transpose) -> packed operands of mulan_linear_f16x3_batched;
// wmax [batch][16] = mulan_absmax_rows(w, batch rows)
MULAN_API int mulan_linear_pack_f16x3_batched(const float* w, void* wp, const unsigned* wmax, int K, int N, int transpose,
                                              int batch, hipStream_t stream) {
  if (K <= 0 || N <= 0 || K % 16 != 0 || !wmax || batch <= 0 || batch > 65535) return (int)hipErrorInvalidValue;
  const size_t total = (size_t)K * N;
  const int blocks = (int)((total + 255) / 256 > 256 ? 256 : (total + 255) / 256);
  hipLaunchKernelGGL(linear_pack_f16x3_kernel, dim3(blocks, batch), dim3(256), 0, stream, w, static_cast<_Float16*>(wp),
                     wmax, K, N, transpose);
  MULAN_CHECK_LAUNCH();
}
