// 3x3 convolution with fp32-equivalent products on the bf16 matrix cores ("bf16x6" = the 6-pass scheme XLA uses
// for jax_default_matmul_precision=float32 / Precision.HIGHEST, which is what the reference requests: ldm/main.py:39).
//
// Every fp32 operand is split exactly into three bf16 pieces  v = v1 + v2 + v3  (8 + 8 + 8 mantissa bits) and
//     a * b  ~=  a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1          (terms below 2^-24 |a||b| dropped)
// is accumulated in fp32 by six v_mfma_f32_32x32x16_bf16 per 16-deep k step: 6 x 32 cycles instead of the
// 8 x 64 cycles of v_mfma_f32_32x32x2_f32 for the same k, i.e. 2.67x fewer matrix-pipe cycles at fp32 accuracy
// (small integers stay exact; error bound ~2^-23 sum|a||b|, the same order as an fp32 FMA chain).
//
// Same tiling as conv3x3_fwd_kernel (block = 4 image rows x 128 couts, wave = 2 rows x 64 couts, halo patch in
// LDS, one stage per (16-channel chunk, tap)); the activations are split while they are staged into LDS, the
// weights are pre-split by mulan_conv3x3_pack_bf16x6 into exactly the LDS tile layout [cout][plane][16 k].
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kW = 32, kPW = 34, CK = 16, TROWS = 4;
constexpr int PIXB = 112;                       // bytes per patch pixel: 3 planes x 32 B + 16 pad ((PIXB/16) odd)
constexpr int NB = 112;                         // bytes per cout row of a weight tile
constexpr int PATCH_B = (TROWS + 2) * kPW * PIXB;   // 22848
constexpr int BN = 128;
constexpr int WT_B = BN * NB;                   // 14336
constexpr int SMEM_B = PATCH_B + 3 * WT_B;      // 65856 (dynamic shared memory): 1 patch + 3-deep weight ring

struct ConvArgsB {
  const float* x;            // [B,H,32,C] fp32
  const unsigned char* wp;   // packed weights [9][C/16][N][3][16] bf16
  const float* bias; const float* cbias; const float* res;
  float* y;
  int B, H, C, N, cbias_mode;
  unsigned long long* stamps;   // dev-only (mulan_set_debug_buffer)
};

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)v;
  const float r1 = v - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

__global__ __launch_bounds__(256) void conv3x3_bf16x6_kernel(ConvArgsB p) {
  constexpr int MT = 2, NT = 2, WN = 2;
  constexpr int PV = 4;                          // float4 patch slots per thread (816 slots)
  constexpr int WV = 3;                          // 16-byte weight pieces per thread (768 pieces)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* pbuf0 = smem;
  unsigned char* wbuf0 = smem + PATCH_B;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_per_img = p.H / TROWS;
  const int b = blockIdx.x / tiles_per_img;
  const int h0 = (blockIdx.x % tiles_per_img) * TROWS;
  const int n0 = blockIdx.y * BN;
  const int C = p.C, N = p.N;
  const int nchunks = C / CK;
  const float* xb = p.x + (size_t)b * p.H * kW * C;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // chunk / stage invariant prefetch addressing
  const float* pptr[PV];
  int pdst[PV];
  unsigned phalo = 0;
#pragma unroll
  for (int s = 0; s < PV; ++s) {
    const int slot = tid + s * 256;
    const int q = slot & 3, pix = slot >> 2;
    const int prow = pix / kPW, pcol = pix - prow * kPW;
    const int hh = h0 + prow - 1, ww = pcol - 1;
    const bool inb = slot < (TROWS + 2) * kPW * 4;
    const bool ok = inb && hh >= 0 && hh < p.H && ww >= 0 && ww < kW;
    pptr[s] = ok ? xb + ((size_t)hh * kW + ww) * C + q * 4 : p.x;
    pdst[s] = inb ? pix * PIXB + q * 8 : -1;
    phalo |= (ok ? 1u : 0u) << s;
  }
  int wsrc[WV], wdst[WV];
#pragma unroll
  for (int s = 0; s < WV; ++s) {
    const int part = tid + s * 256;              // 16-byte piece of the 128 x 96 B tile
    const int n = part / 6, piece = part - n * 6;
    wsrc[s] = part * 16;
    wdst[s] = n * NB + piece * 16;
  }
  const size_t tile_stride = (size_t)N * 96;     // bytes between (tap, chunk) tiles of the packed weights
  const unsigned char* wtile0 = p.wp + (size_t)n0 * 96;

  f32x4 preg[PV];
  f32x4 wreg[WV];
  auto gload_patch = [&](int cc) {
#pragma unroll
    for (int s = 0; s < PV; ++s) preg[s] = *reinterpret_cast<const f32x4*>(pptr[s] + (((phalo >> s) & 1u) ? cc * CK : 0));
  };
  auto store_patch = [&](unsigned char* pb) {
#pragma unroll
    for (int s = 0; s < PV; ++s) {
      if (pdst[s] < 0) continue;
      f32x4 v = preg[s];
      if (!((phalo >> s) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x4 hi, mi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 h, m, l;
        split3(v[e], h, m, l);
        hi[e] = h; mi[e] = m; lo[e] = l;
      }
      *reinterpret_cast<bf16x4*>(pb + pdst[s]) = hi;
      *reinterpret_cast<bf16x4*>(pb + pdst[s] + 32) = mi;
      *reinterpret_cast<bf16x4*>(pb + pdst[s] + 64) = lo;
    }
  };
  auto gload_w = [&](int cc, int tap) {
    const unsigned char* t = wtile0 + ((size_t)tap * nchunks + cc) * tile_stride;
#pragma unroll
    for (int s = 0; s < WV; ++s) wreg[s] = *reinterpret_cast<const f32x4*>(t + wsrc[s]);
  };
  auto store_w = [&](unsigned char* wb) {
#pragma unroll
    for (int s = 0; s < WV; ++s) *reinterpret_cast<f32x4*>(wb + wdst[s]) = wreg[s];
  };

  // Software pipeline.  Weight tiles live in a 3-deep LDS ring: tile s+1 is complete (barrier of stage s-1) while
  // stage s computes, so stage s+1's A and B fragments are read from LDS *during* stage s's 24 MFMAs and nothing
  // waits on an LDS read in front of an MFMA cluster; tile s+2 is fetched from L2 meanwhile and stored into the slot
  // that tile s-1 vacated.  The activation patch is single buffered: one extra barrier per 9 stages.
  const int nsteps = nchunks * 9;
  auto tile_of = [&](int s2, int& cc2, int& tap2) { cc2 = s2 / 9; tap2 = s2 - cc2 * 9; };
  auto read_frags = [&](bf16x8 (&af)[MT][3], bf16x8 (&bfr)[NT][3], int tap, const unsigned char* wb) {
    const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int prow = wm * MT + mt + kh, pcol = li + kw;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        af[mt][pl] = *reinterpret_cast<const bf16x8*>(pbuf0 + (prow * kPW + pcol) * PIXB + pl * 32 + lh * 16);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        bfr[nt][pl] = *reinterpret_cast<const bf16x8*>(wb + ((wn * NT + nt) * 32 + li) * NB + pl * 32 + lh * 16);
  };

  const bool stamp = p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  if (stamp) { p.stamps[0] = __builtin_amdgcn_s_memtime(); p.stamps[30] = __builtin_amdgcn_s_memrealtime(); }
  gload_patch(0);
  gload_w(0, 0);
  store_patch(pbuf0);
  store_w(wbuf0);
  if (nsteps > 1) {
    gload_w(0, 1);
    store_w(wbuf0 + WT_B);
  }
  __syncthreads();

  bf16x8 afc[MT][3], bfc[NT][3], afn[MT][3], bfn[NT][3];
  read_frags(afc, bfc, 0, wbuf0);
  int step = 0, slot_next = 1, slot_fill = 2;          // ring slots of tile step+1 / step+2
  if (stamp) p.stamps[1] = __builtin_amdgcn_s_memtime();
  for (int cc = 0; cc < nchunks; ++cc) {
    if (stamp && cc < 20) p.stamps[2 + cc] = __builtin_amdgcn_s_memtime();
    const bool more_chunks = cc + 1 < nchunks;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap, ++step) {
      const bool has_next = step + 1 < nsteps, has_fill = step + 2 < nsteps;
      if (tap < 8) read_frags(afn, bfn, tap + 1, wbuf0 + slot_next * WT_B);   // next stage's operands, same patch
      if (has_fill) {
        int c2, t2;
        tile_of(step + 2, c2, t2);
        gload_w(c2, t2);
      }
      if (tap == 0 && more_chunks) gload_patch(cc + 1);
      __builtin_amdgcn_sched_barrier(0);

      // small terms first: a1 b3, a3 b1, a2 b2, a1 b2, a2 b1, a1 b1
#pragma unroll
      for (int term = 0; term < 6; ++term) {
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0};
        constexpr int PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[mt][PA[term]], bfc[nt][PB[term]], acc[mt][nt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (has_fill) store_w(wbuf0 + slot_fill * WT_B);
      __syncthreads();
      if (tap == 8 && more_chunks) {       // every wave has left this chunk's patch: replace it
        store_patch(pbuf0);
        __syncthreads();
      }
      if (tap == 8 && has_next) read_frags(afn, bfn, 0, wbuf0 + slot_next * WT_B);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) afc[mt][pl] = afn[mt][pl];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bfc[nt][pl] = bfn[nt][pl];
      slot_next = slot_next == 2 ? 0 : slot_next + 1;
      slot_fill = slot_fill == 2 ? 0 : slot_fill + 1;
    }
  }

  if (stamp) p.stamps[22] = __builtin_amdgcn_s_memtime();
  // epilogue: transposed through LDS so every lane moves float4s (see conv3x3_fwd_kernel)
  const float* __restrict__ res = p.res;
  const float* __restrict__ cbp = p.cbias;
  float* __restrict__ yout = p.y;
  constexpr int TS = 64 + 4;
  float* stage = reinterpret_cast<float*>(smem) + wave * 32 * TS;
  const int c4 = lane & 15, prl = lane >> 4;
  const int nb = n0 + wn * 64 + c4 * 4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + nb);
  if (p.cbias_mode == 1) {
    const f32x4 c = *reinterpret_cast<const f32x4*>(cbp + (size_t)b * N + nb);
    bias4[0] += c[0]; bias4[1] += c[1]; bias4[2] += c[2]; bias4[3] += c[3];
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[mfma32_row(r, lane) * TS + nt * 32 + li] = acc[mt][nt][r];
    __syncthreads();
    const int hh = h0 + wm * MT + mt;
    const size_t rowbase = (((size_t)b * p.H + hh) * kW) * N + nb;
    f32x4 add[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) add[it] = bias4;
    if (p.cbias_mode == 2) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(cbp + rowbase + (size_t)(it * 4 + prl) * N);
        add[it][0] += c[0]; add[it][1] += c[1]; add[it][2] += c[2]; add[it][3] += c[3];
      }
    }
    if (res) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(res + rowbase + (size_t)(it * 4 + prl) * N);
        add[it][0] += c[0]; add[it][1] += c[1]; add[it][2] += c[2]; add[it][3] += c[3];
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(stage + (it * 4 + prl) * TS + c4 * 4);
      const f32x4 o = {a[0] + add[it][0], a[1] + add[it][1], a[2] + add[it][2], a[3] + add[it][3]};
      *reinterpret_cast<f32x4*>(yout + rowbase + (size_t)(it * 4 + prl) * N) = o;
    }
  }
  if (stamp) { p.stamps[23] = __builtin_amdgcn_s_memtime(); p.stamps[31] = __builtin_amdgcn_s_memrealtime(); }
}

// wp[t][cc][o][plane][k] = split3( Wl[t][cc*16 + k][o] ),  Wl = w (flip = 0) or the tap-flipped, channel-transposed
// weights of the input-gradient convolution (flip = 1: Wl[t][k][o] = w[8-t][o][k], k over N, o over C).
__global__ void conv3x3_pack_bf16x6_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int C, int N, int flip) {
  const int Kin = flip ? N : C, Nout = flip ? C : N;
  const size_t total = (size_t)9 * Kin * Nout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % 16);
    size_t r = i / 16;
    const int o = (int)(r % Nout); r /= Nout;
    const int cc = (int)(r % (Kin / 16));
    const int t = (int)(r / (Kin / 16));
    const int kin = cc * 16 + k;
    const float v = flip ? w[((size_t)(8 - t) * C + o) * N + kin] : w[((size_t)t * C + kin) * N + o];
    __bf16 h, m, l;
    split3(v, h, m, l);
    __bf16* dst = wp + (((size_t)(t * (Kin / 16) + cc) * Nout + o) * 3) * 16 + k;
    dst[0] = h; dst[16] = m; dst[32] = l;
  }
}


// ------------------------------------------------------------------------------------------------ wgrad
// dW[t][ci][co] = sum_pixels x[p + shift_t][ci] * dy[p][co] with the same 6-pass split.  The contraction index is
// the pixel, while both tensors are stored channel-contiguous, so the operands are fetched with the transposing LDS
// read ds_read_b64_tr_b16 (a 4 k x 16 channel block per 16-lane group, delivered k-major): a tap shift is then a
// whole-row address offset and every read stays 8-byte aligned.  LDS rows are 64 B (32 channels of one split plane),
// which makes the two 16-lane groups of a half-wave cover all 64 banks exactly once.
constexpr int WG_T = 64, WG_ROWS = 2;
constexpr int XPIX = (WG_ROWS + 2) * kPW;                 // 136 halo-patch pixels
constexpr int X_HALF = XPIX * 64, X_PLANE = 2 * X_HALF;   // bytes
constexpr int DPIX = WG_ROWS * kW;                        // 64
constexpr int D_HALF = DPIX * 64, D_PLANE = 2 * D_HALF;
constexpr int WG_SMEM = 3 * X_PLANE + 3 * D_PLANE;        // 52224 + 24576 = 76800

struct WgradArgsB {
  const float* x; const float* dy; float* slab;
  int B, H, C, N, S;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* base) {
  // two 4-k blocks (k .. k+3 and k+4 .. k+7) -> the 8 k values of this lane's MFMA operand
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + 4 * 64));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256) void conv3x3_wgrad_bf16x6_kernel(WgradArgsB p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;
  unsigned char* ds = smem + 3 * X_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wci = wave >> 1, wco = wave & 1;
  const int C = p.C, N = p.N;
  const int c0 = blockIdx.y * WG_T, n0 = blockIdx.z * WG_T;
  const int pairs_per_img = p.H / WG_ROWS;
  const int total_pairs = p.B * pairs_per_img;
  const int per_split = (total_pairs + p.S - 1) / p.S;
  const int pair_begin = blockIdx.x * per_split;
  const int pair_end = min(total_pairs, pair_begin + per_split);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  constexpr int XV = (XPIX * 16 + 255) / 256;   // 9 float4 slots / thread
  constexpr int DV = (DPIX * 16) / 256;         // 4
  f32x4 xreg[XV], dreg[DV];
  int xoff[XV], doff[DV], xdst[XV], ddst[DV];
  unsigned xstat = 0, xtop = 0, xbot = 0, dstat = 0, xmask = 0;
#pragma unroll
  for (int i = 0; i < XV; ++i) {
    const int slot = tid + i * 256;
    const int q = slot & 15, pix = slot >> 4;
    const int prow = pix / kPW, pcol = pix - prow * kPW;
    const int ww = pcol - 1, c = c0 + q * 4;
    const bool inb = slot < XPIX * 16;
    const bool ok = inb && ww >= 0 && ww < kW && c < C;
    xoff[i] = ((prow - 1) * kW + ww) * C + c;
    xdst[i] = inb ? (q >> 3) * X_HALF + pix * 64 + (q & 7) * 8 : -1;
    xstat |= (ok ? 1u : 0u) << i;
    xtop |= (prow == 0 ? 1u : 0u) << i;
    xbot |= (prow == WG_ROWS + 1 ? 1u : 0u) << i;
  }
#pragma unroll
  for (int i = 0; i < DV; ++i) {
    const int slot = tid + i * 256;
    const int q = slot & 15, pix = slot >> 4;
    const int n = n0 + q * 4;
    doff[i] = pix * N + n;
    ddst[i] = (q >> 3) * D_HALF + pix * 64 + (q & 7) * 8;
    dstat |= (n < N ? 1u : 0u) << i;
  }
  auto gload = [&](int pr) {
    const int b = pr / pairs_per_img, h0 = (pr - b * pairs_per_img) * WG_ROWS;
    const float* xrow = p.x + ((size_t)b * p.H + h0) * kW * C;
    const float* dyb = p.dy + ((size_t)b * p.H + h0) * kW * N;
    xmask = xstat & ~(h0 == 0 ? xtop : 0u) & ~(h0 + WG_ROWS >= p.H ? xbot : 0u);
#pragma unroll
    for (int i = 0; i < XV; ++i) xreg[i] = *reinterpret_cast<const f32x4*>(((xmask >> i) & 1u) ? xrow + xoff[i] : p.x);
#pragma unroll
    for (int i = 0; i < DV; ++i) dreg[i] = *reinterpret_cast<const f32x4*>(((dstat >> i) & 1u) ? dyb + doff[i] : p.dy);
  };
  auto split_store = [&](unsigned char* base, int plane_stride, int dst, f32x4 v) {
    bf16x4 hi, mi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      __bf16 h, m, l;
      split3(v[e], h, m, l);
      hi[e] = h; mi[e] = m; lo[e] = l;
    }
    *reinterpret_cast<bf16x4*>(base + dst) = hi;
    *reinterpret_cast<bf16x4*>(base + plane_stride + dst) = mi;
    *reinterpret_cast<bf16x4*>(base + 2 * plane_stride + dst) = lo;
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      if (xdst[i] < 0) continue;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      split_store(xs, X_PLANE, xdst[i], ((xmask >> i) & 1u) ? xreg[i] : z);
    }
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      split_store(ds, D_PLANE, ddst[i], ((dstat >> i) & 1u) ? dreg[i] : z);
    }
  };

  // lane part of every transposing read: row q of the 4-k block, columns 4p..4p+3 of this group's 16 channels,
  // k half = lh  (MFMA operand element j <-> k = 8 lh + j)
  const int grp_q = (lane & 15) >> 2, grp_p = lane & 3, cb16 = ((lane >> 4) & 1) * 16;
  const int lane_off = (8 * lh + grp_q) * 64 + (cb16 + 4 * grp_p) * 2;
  const unsigned char* xa = xs + wci * X_HALF + lane_off;
  const unsigned char* db = ds + wco * D_HALF + lane_off;

  if (pair_begin < pair_end) {
    gload(pair_begin);
    lstore();
  }
  __syncthreads();
  for (int pr = pair_begin; pr < pair_end; ++pr) {
    const bool has_next = pr + 1 < pair_end;
    if (has_next) gload(pr + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int ks = 0; ks < DPIX / 16; ++ks) {
      const int rr = ks >> 1, w0 = (ks & 1) * 16;
      bf16x8 bfr[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bfr[pl] = tr_read8(db + pl * D_PLANE + (rr * kW + w0) * 64);
      bf16x8 af[2][3];
      const unsigned char* xk = xa + (rr * kPW + w0) * 64;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) af[0][pl] = tr_read8(xk + pl * X_PLANE);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t + 1 < 9) {
          const int kh = (t + 1) / 3, kw = (t + 1) % 3;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) af[(t + 1) & 1][pl] = tr_read8(xk + pl * X_PLANE + (kh * kPW + kw) * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int term = 0; term < 6; ++term) {
          constexpr int PA[6] = {0, 2, 1, 0, 1, 0};
          constexpr int PB[6] = {2, 0, 1, 1, 0, 0};
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t & 1][PA[term]], bfr[PB[term]], acc[t], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (has_next) {
      lstore();
      __syncthreads();
    }
  }

  float* slab = p.slab + (size_t)blockIdx.x * 9 * C * N;
  const int n = n0 + wco * 32 + li;
  if (n < N) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wci * 32 + mfma32_row(r, lane);
        if (c < C) slab[((size_t)t * C + c) * N + n] = acc[t][r];
      }
  }
}

__global__ void slab_reduce_b_kernel(const float* __restrict__ slab, float* __restrict__ out, int S, int E,
                                     int accumulate) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  float s = 0.f;
  int i = 0;
  for (; i + 8 <= S; i += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(i + u) * E + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < S; ++i) s += slab[(size_t)i * E + e];
  out[e] = accumulate ? out[e] + s : s;
}

int wgrad_splits_b(int B, int H, int C, int N) {
  const int tiles = ((C + WG_T - 1) / WG_T) * ((N + WG_T - 1) / WG_T);
  const int pairs = B * (H / WG_ROWS);
  const int target = g_mulan_tune[1] > 0 ? g_mulan_tune[1] : 256;
  int S = target / tiles;
  if (S < 1) S = 1;
  if (S > pairs) S = pairs;
  while (S > 1 && pairs / S < 4) --S;
  return S;
}

}  // namespace

MULAN_API size_t mulan_conv3x3_pack_bf16x6_bytes(int C, int N) { return (size_t)9 * C * N * 3 * 2; }

MULAN_API int mulan_conv3x3_pack_bf16x6(const float* w, void* wp, int C, int N, int flip, hipStream_t stream) {
  const int Kin = flip ? N : C;
  if (Kin % 16 != 0 || C <= 0 || N <= 0) return (int)hipErrorInvalidValue;
  const size_t total = (size_t)9 * C * N;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(conv3x3_pack_bf16x6_kernel, dim3(blocks), dim3(256), 0, stream, w, static_cast<__bf16*>(wp), C, N,
                     flip);
  MULAN_CHECK_LAUNCH();
}

// Eligibility: W == 32, H % 4 == 0, C % 16 == 0, N % 128 == 0 (the ResBlock convolutions); everything else goes
// through mulan_conv3x3_fwd.  `wp` is the packed weight of mulan_conv3x3_pack_bf16x6 for this direction.
MULAN_API int mulan_conv3x3_fwd_bf16x6(const float* x, const void* wp, const float* bias, const float* cbias,
                                       int cbias_mode, const float* res, float* y, int B, int H, int W, int C, int N,
                                       hipStream_t stream) {
  if (W != kW || H % TROWS != 0 || B <= 0 || C % CK != 0 || C <= 0 || N % BN != 0 || N <= 0)
    return (int)hipErrorInvalidValue;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16x6_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  ConvArgsB a{x, static_cast<const unsigned char*>(wp), bias, cbias, res, y, B, H, C, N, cbias ? cbias_mode : 0,
              g_mulan_debug_buffer};
  dim3 grid(B * (H / TROWS), N / BN);
  hipLaunchKernelGGL(conv3x3_bf16x6_kernel, grid, dim3(256), SMEM_B, stream, a);
  MULAN_CHECK_LAUNCH();
}

MULAN_API size_t mulan_conv3x3_wgrad_bf16x6_workspace(int B, int H, int W, int C, int N) {
  if (W != kW || H % WG_ROWS != 0) return 0;
  return (size_t)wgrad_splits_b(B, H, C, N) * 9 * C * N * sizeof(float);
}

// dw[3,3,C,N] (+)= sum x (x) dy with the 6-pass split; needs C % 4 == 0, N % 4 == 0 and 16-byte aligned tensors.
MULAN_API int mulan_conv3x3_wgrad_bf16x6(const float* x, const float* dy, float* dw, float* workspace, int B, int H,
                                         int W, int C, int N, int accumulate, hipStream_t stream) {
  if (W != kW || H % WG_ROWS != 0 || B <= 0 || C % 4 != 0 || N % 4 != 0) return (int)hipErrorInvalidValue;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_bf16x6_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WG_SMEM);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  const int S = wgrad_splits_b(B, H, C, N);
  WgradArgsB a{x, dy, workspace, B, H, C, N, S};
  dim3 grid(S, (C + WG_T - 1) / WG_T, (N + WG_T - 1) / WG_T);
  hipLaunchKernelGGL(conv3x3_wgrad_bf16x6_kernel, grid, dim3(256), WG_SMEM, stream, a);
  const int E = 9 * C * N;
  hipLaunchKernelGGL(slab_reduce_b_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, workspace, dw, S, E, accumulate);
  MULAN_CHECK_LAUNCH();
}
