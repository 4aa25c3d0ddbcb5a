// 3x3 / stride 1 / SAME convolution on NHWC fp32 images with W == 32, as an implicit GEMM on the
// exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32).  Replaces the XLA-lowered flax nn.Conv calls
// of the reference ResnetBlock (ldm/model_vdm.py:633-634,645-650; ldm/ldm_unet.py:33-34,49-54),
// conv_in/conv_out (model_vdm.py:348-349,378-383) and their autodiff (experiment.py:339).
//
//   forward   y[b,h,w,n]  = sum_{kh,kw,c} x[b,h+kh-1,w+kw-1,c] * wgt[kh,kw,c,n]  (+bias +cond bias +res)
//   dgrad     = the same kernel on dy with wflip(wgt)[t][n][c] = wgt[8-t][c][n]
//   wgrad     dW[t,c,n]   = sum_{b,h,w} x[b,h+kh-1,w+kw-1,c] * dy[b,h,w,n]        (split over images)
//
// One MFMA M-tile (32 rows) is one image row, so the im2col gather is just an LDS address shift
// into a halo patch [rows+2][34][channel chunk]; nothing is materialised in HBM.
#include "common.h"

namespace {

constexpr int kW = 32;          // image width the tiling is built around
constexpr int kPW = 34;         // patch width with the zero halo
constexpr int CK = 16;          // input-channel chunk per pipeline stage
constexpr int PS = CK + 4;      // patch pixel stride in floats: (PS/4) odd => ds_read_b128 conflict free
constexpr int TROWS = 4;        // image rows per block tile (BM = 128 pixels)
constexpr int PATCH_F = (TROWS + 2) * kPW * PS;   // floats per patch buffer (4080)

struct ConvArgs {
  const float* x;      // [B,H,32,C]
  const float* w;      // [9,C,N]   (HWIO)
  const float* bias;   // [N] or null
  const float* cbias;  // cond bias: mode 1 [B,N], mode 2 [B,H,32,N]
  const float* res;    // [B,H,32,N] or null
  float* y;            // [B,H,32,N]
  int B, H, C, N, cbias_mode;
  unsigned long long* stamps;   // dev-only: s_memtime stamps of block 0 / wave 0 (mulan_set_debug_buffer), else null
};

#define MULAN_STAMP(i)                                                                   \
  do {                                                                                   \
    if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)              \
      p.stamps[i] = __builtin_amdgcn_s_memtime();                                        \
  } while (0)

// VEC: C % 4 == 0 and N % 4 == 0.  The VEC loaders are branch free (clamped address + validity mask applied at
// the LDS store): a divergent branch around a prefetch load makes hipcc wait vmcnt(0) right after issuing it,
// which serialises the global-load latency in front of every MFMA cluster.
// TPS: taps per pipeline stage (1 or 3): one __syncthreads() per TPS*32 MFMAs per wave.
template <int BN, int WM, int WN, bool VEC, int TPS>
__global__ __launch_bounds__(256) void conv3x3_fwd_kernel(ConvArgs p) {
  constexpr int MT = TROWS / WM;        // image rows per wave
  constexpr int NT = BN / 32 / WN;      // 32-wide cout tiles per wave
  constexpr int WT1 = CK * BN;          // floats per tap tile
  constexpr int WT_F = TPS * WT1;       // floats per weight buffer
  constexpr int NSG = 9 / TPS;          // stages per channel chunk
  constexpr int NPB = TPS == 1 ? 2 : 1; // patch buffers (single buffer when the weight stage is large)
  constexpr int WV = (WT_F / 4 + 255) / 256;  // float4 weight slots per thread
  constexpr int PV = (PATCH_F / PS * (CK / 4) + 255) / 256;  // float4 patch slots per thread (4)
  __shared__ __attribute__((aligned(16))) float smem[NPB * PATCH_F + 2 * WT_F];
  float* pbuf0 = smem;
  float* wbuf0 = smem + NPB * PATCH_F;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_per_img = p.H / TROWS;
  const int b = blockIdx.x / tiles_per_img;
  const int h0 = (blockIdx.x % tiles_per_img) * TROWS;
  const int n0 = blockIdx.y * BN;
  const int C = p.C, N = p.N;
  const bool vec_in = VEC || (C & 3) == 0;
  const bool vec_w = VEC || (N & 3) == 0;
  const int nchunks = (C + CK - 1) / CK;
  const float* xb = p.x + (size_t)b * p.H * kW * C;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 preg[PV];
  f32x4 wreg[WV];
  unsigned pmask = 0, wmask = 0;   // validity bits of the prefetched slots (VEC path)

  // VEC path: everything about a thread's prefetch slots that does not change from stage to stage is computed
  // once here, so a stage's prefetch costs a handful of instructions (address arithmetic issued in front of an
  // MFMA cluster is directly exposed: the probe in tools/mfma_probe.hip runs this loop shape at ~90 % of peak).
  const float* pptr[PV];
  const float* wptr[WV];
  unsigned phalo = 0, wnok = 0;
  int wk[WV];
  if (VEC) {
#pragma unroll
    for (int s = 0; s < PV; ++s) {
      const int slot = tid + s * 256;
      const int q = slot & 3, pix = slot >> 2;
      const int prow = pix / kPW, pcol = pix - prow * kPW;
      const int hh = h0 + prow - 1, ww = pcol - 1;
      const bool ok = slot < (TROWS + 2) * kPW * (CK / 4) && hh >= 0 && hh < p.H && ww >= 0 && ww < kW;
      pptr[s] = ok ? xb + ((size_t)hh * kW + ww) * C + q * 4 : p.x;
      phalo |= (ok ? 1u : 0u) << s;
    }
#pragma unroll
    for (int s = 0; s < WV; ++s) {
      const int slot = tid + s * 256;
      const int tl = slot / (WT1 / 4), rs = slot - tl * (WT1 / 4);
      const int k = rs / (BN / 4), nq = rs - k * (BN / 4);
      const int n = n0 + nq * 4;
      const bool ok = slot < WT_F / 4 && n < N;
      wptr[s] = ok ? p.w + ((size_t)tl * C + k) * N + n : p.w;
      wnok |= (ok ? 1u : 0u) << s;
      wk[s] = k;
    }
  }

  auto gload_patch = [&](int cc) {
    if (VEC) {
      pmask = 0;
      const int c0 = cc * CK;
#pragma unroll
      for (int s = 0; s < PV; ++s) {
        const bool ok = ((phalo >> s) & 1u) && (c0 + (int)(((tid + s * 256) & 3) * 4) < C);
        preg[s] = *reinterpret_cast<const f32x4*>(ok ? pptr[s] + c0 : p.x);
        pmask |= (ok ? 1u : 0u) << s;
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < PV; ++s) {
      const int slot = tid + s * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (slot < (TROWS + 2) * kPW * (CK / 4)) {
        const int q = slot & 3, pix = slot >> 2;
        const int prow = pix / kPW, pcol = pix - prow * kPW;
        const int hh = h0 + prow - 1, ww = pcol - 1;
        const int c = cc * CK + q * 4;
        if (hh >= 0 && hh < p.H && ww >= 0 && ww < kW && c < C) {
          const float* src = xb + ((size_t)hh * kW + ww) * C + c;
          if (vec_in) {
            v = *reinterpret_cast<const f32x4*>(src);
          } else {
            v[0] = src[0];
            if (c + 1 < C) v[1] = src[1];
            if (c + 2 < C) v[2] = src[2];
            if (c + 3 < C) v[3] = src[3];
          }
        }
      }
      preg[s] = v;
    }
  };
  auto store_patch = [&](float* pb) {
#pragma unroll
    for (int s = 0; s < PV; ++s) {
      const int slot = tid + s * 256;
      if (slot < (TROWS + 2) * kPW * (CK / 4)) {
        const int q = slot & 3, pix = slot >> 2;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(pb + pix * PS + q * 4) = (!VEC || ((pmask >> s) & 1u)) ? preg[s] : z;
      }
    }
  };
  auto gload_w = [&](int cc, int sg) {
    if (VEC) {
      wmask = 0;
      const int c0 = cc * CK;
      const size_t off = ((size_t)(sg * TPS) * C + c0) * N;   // wave uniform
#pragma unroll
      for (int s = 0; s < WV; ++s) {
        const bool ok = ((wnok >> s) & 1u) && (c0 + wk[s] < C);
        wreg[s] = *reinterpret_cast<const f32x4*>(ok ? wptr[s] + off : p.w);
        wmask |= (ok ? 1u : 0u) << s;
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < WV; ++s) {
      const int slot = tid + s * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (slot < WT_F / 4) {
        const int tl = slot / (WT1 / 4), rs = slot - tl * (WT1 / 4);
        const int tap = sg * TPS + tl;
        const int k = rs / (BN / 4), nq = rs - k * (BN / 4);
        const int c = cc * CK + k, n = n0 + nq * 4;
        if (c < C && n < N) {
          const float* src = p.w + ((size_t)tap * C + c) * N + n;
          if (vec_w) {
            v = *reinterpret_cast<const f32x4*>(src);
          } else {
            v[0] = src[0];
            if (n + 1 < N) v[1] = src[1];
            if (n + 2 < N) v[2] = src[2];
            if (n + 3 < N) v[3] = src[3];
          }
        }
      }
      wreg[s] = v;
    }
  };
  auto store_w = [&](float* wb) {
#pragma unroll
    for (int s = 0; s < WV; ++s) {
      const int slot = tid + s * 256;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      if (slot < WT_F / 4) *reinterpret_cast<f32x4*>(wb + slot * 4) = (!VEC || ((wmask >> s) & 1u)) ? wreg[s] : z;
    }
  };

  MULAN_STAMP(0);
  if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.stamps[30] = __builtin_amdgcn_s_memrealtime();
  // prologue: stage chunk 0 / stage 0
  gload_patch(0);
  gload_w(0, 0);
  store_patch(pbuf0);
  store_w(wbuf0);
  __syncthreads();

  MULAN_STAMP(1);
  int step = 0;
  for (int cc = 0; cc < nchunks; ++cc) {
    if (cc < 20) MULAN_STAMP(2 + cc);
    const float* pb = pbuf0 + (NPB == 2 ? (cc & 1) * PATCH_F : 0);
    const bool more_chunks = cc + 1 < nchunks;
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg, ++step) {
      const float* wbs = wbuf0 + (step & 1) * WT_F;
      float* wb_next = wbuf0 + ((step + 1) & 1) * WT_F;
      const bool last_sg = sg == NSG - 1;
      const bool has_next = !last_sg || more_chunks;

#pragma unroll
      for (int tl = 0; tl < TPS; ++tl) {
        const int tap = sg * TPS + tl;
        const int kh = tap / 3, kw = tap - kh * 3;
        const float* wb = wbs + tl * WT1;
        // A fragments of the whole tap up front (ds_read_b128), B fragments double-buffered one k-step
        // ahead (ds_read_b32) so no MFMA group waits on an LDS read issued right in front of it.
        f32x4 a4[CK / 8][MT];
#pragma unroll
        for (int k8 = 0; k8 < CK / 8; ++k8)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int prow = wm * MT + mt + kh, pcol = li + kw;
            a4[k8][mt] = *reinterpret_cast<const f32x4*>(pb + (prow * kPW + pcol) * PS + k8 * 8 + 4 * lh);
          }
        float bcur[NT], bnext[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bcur[nt] = wb[(4 * lh) * BN + (wn * NT + nt) * 32 + li];
#pragma unroll
        for (int st = 0; st < CK / 2; ++st) {
          const int k8 = st >> 2, j = st & 3;
          if (st + 1 < CK / 2) {
            const int kk = ((st + 1) >> 2) * 8 + 4 * lh + ((st + 1) & 3);   // k permutation shared by A and B
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bnext[nt] = wb[kk * BN + (wn * NT + nt) * 32 + li];
          }
          if (tl == 0 && st == 1) {
            // next stage's prefetch is issued behind the first MFMA group so its address arithmetic runs in the
            // shadow of the matrix pipe; pinned here (hipcc otherwise sinks a load past the cluster and exposes
            // its latency in front of the barrier)
            if (has_next) {
              if (!last_sg) gload_w(cc, sg + 1); else gload_w(cc + 1, 0);
            }
            // the next chunk's activation patch comes from HBM (several us): request it a whole chunk ahead
            if (sg == 0 && more_chunks) gload_patch(cc + 1);
          }
          __builtin_amdgcn_sched_barrier(0);   // the read of step st+1 stays in front of step st's MFMAs
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma32(a4[k8][mt][j], bcur[nt], acc[mt][nt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bcur[nt] = bnext[nt];
        }
      }

      __builtin_amdgcn_sched_barrier(0);
      if (has_next) store_w(wb_next);
      if (NPB == 2) {
        if (last_sg && more_chunks) store_patch(pbuf0 + ((cc + 1) & 1) * PATCH_F);
        __syncthreads();
      } else {
        __syncthreads();
        if (last_sg && more_chunks) {     // single patch buffer: overwrite only after every wave left this chunk
          store_patch(pbuf0);
          __syncthreads();
        }
      }
    }
  }

  MULAN_STAMP(22);
  // epilogue
  const float* __restrict__ res = p.res;
  const float* __restrict__ cbp = p.cbias;
  float* __restrict__ yout = p.y;
  if (VEC && NT == 2 && MT >= 1) {
    // Transposed through LDS so every lane moves float4s: a wave instruction covers 4 pixels x 256 contiguous
    // bytes (the accumulator layout itself would need 64 dword loads + 64 dword stores per lane).
    constexpr int TS = 64 + 4;                       // staging row stride (floats), 16-byte aligned rows
    float* stage = smem + wave * 32 * TS;            // one 32 px x 64 cout tile per wave
    const int c4 = lane & 15, prl = lane >> 4;       // float4 column, pixel sub-row
    const int nb = n0 + wn * 64 + c4 * 4;
    const bool nok = nb < N;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (nok) {
      if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + nb);
      if (p.cbias_mode == 1) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(cbp + (size_t)b * N + nb);
        bias4[0] += c[0]; bias4[1] += c[1]; bias4[2] += c[2]; bias4[3] += c[3];
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      __syncthreads();                               // previous users of this LDS region are done
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
      if (nok && p.cbias_mode == 2) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const f32x4 c = *reinterpret_cast<const f32x4*>(cbp + rowbase + (size_t)(it * 4 + prl) * N);
          add[it][0] += c[0]; add[it][1] += c[1]; add[it][2] += c[2]; add[it][3] += c[3];
        }
      }
      if (nok && res) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const f32x4 c = *reinterpret_cast<const f32x4*>(res + rowbase + (size_t)(it * 4 + prl) * N);
          add[it][0] += c[0]; add[it][1] += c[1]; add[it][2] += c[2]; add[it][3] += c[3];
        }
      }
      if (nok) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(stage + (it * 4 + prl) * TS + c4 * 4);
          const f32x4 o = {a[0] + add[it][0], a[1] + add[it][1], a[2] + add[it][2], a[3] + add[it][3]};
          *reinterpret_cast<f32x4*>(yout + rowbase + (size_t)(it * 4 + prl) * N) = o;
        }
      }
    }
    MULAN_STAMP(23);
    if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.stamps[31] = __builtin_amdgcn_s_memrealtime();
    return;
  }
  // generic epilogue: the loads of a whole 32x32 tile are issued before any store (the __restrict__ copies let
  // hipcc do that; otherwise it serialises load -> wait -> store per element, 64 dependent round trips per block)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + (wn * NT + nt) * 32 + li;
    if (n >= N) continue;
    const float bv = p.bias ? p.bias[n] : 0.f;
    const float cb1 = (p.cbias_mode == 1) ? cbp[(size_t)b * N + n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int hh = h0 + wm * MT + mt;
      const size_t rowbase = (((size_t)b * p.H + hh) * kW) * N + n;
      float add[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) add[r] = bv + cb1;
      if (p.cbias_mode == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] += cbp[rowbase + (size_t)mfma32_row(r, lane) * N];
      }
      if (res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] += res[rowbase + (size_t)mfma32_row(r, lane) * N];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) yout[rowbase + (size_t)mfma32_row(r, lane) * N] = acc[mt][nt][r] + add[r];
    }
  }
}

// wT[t][n][c] = w[8-t][c][n]  (tap flip + channel transpose) so dgrad reuses the forward kernel.
__global__ void conv3x3_wflip_kernel(const float* __restrict__ w, float* __restrict__ wT, int C, int N) {
  const int total = 9 * C * N;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int t = i / (C * N), rem = i - t * C * N;
    const int n = rem / C, c = rem - n * C;           // i indexes wT[t][n][c]
    wT[i] = w[((size_t)(8 - t) * C + c) * N + n];
  }
}

// ---------------------------------------------------------------------------------- wgrad
constexpr int WG_T = 64;       // ci tile == co tile
constexpr int WG_ROWS = 2;     // image rows per K chunk (64 pixels)
constexpr int WG_XP = (WG_ROWS + 2) * kPW * WG_T;   // 8704 floats
constexpr int WG_DY = WG_ROWS * kW * WG_T;          // 4096 floats

struct WgradArgs {
  const float* x;    // [B,H,32,C]
  const float* dy;   // [B,H,32,N]
  float* slab;       // [S,9,C,N]
  int B, H, C, N, S;
};

template <bool VEC>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(WgradArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[WG_XP + WG_DY];
  float* xp = smem;
  float* dyt = smem + WG_XP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wci = wave >> 1, wco = wave & 1;
  const int C = p.C, N = p.N;
  const int c0 = blockIdx.y * WG_T, n0 = blockIdx.z * WG_T;
  const bool vec_x = VEC || (C & 3) == 0, vec_dy = VEC || (N & 3) == 0;
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

  constexpr int XV = ((WG_ROWS + 2) * kPW * (WG_T / 4) + 255) / 256;   // 9 float4 slots / thread
  constexpr int DV = (WG_ROWS * kW * (WG_T / 4) + 255) / 256;           // 4
  f32x4 xreg[XV], dreg[DV];
  unsigned xmask = 0, dmask = 0;
  // chunk-invariant part of the prefetch addressing (VEC path), see conv3x3_fwd_kernel
  int xoff[XV], doff[DV];
  unsigned xstat = 0, xtop = 0, xbot = 0, dstat = 0;
  if (VEC) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int slot = tid + i * 256;
      const int q = slot & 15, pix = slot >> 4;
      const int prow = pix / kPW, pcol = pix - prow * kPW;
      const int ww = pcol - 1, c = c0 + q * 4;
      const bool ok = slot < (WG_ROWS + 2) * kPW * (WG_T / 4) && ww >= 0 && ww < kW && c < C;
      xoff[i] = ((prow - 1) * kW + ww) * C + c;
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
      dstat |= (n < N ? 1u : 0u) << i;
    }
  }
  // global -> registers for row pair `pr` (x halo patch [4][34][64], dy tile [64 px][64]), zero filled.
  // VEC path: branch-free clamped loads + validity mask applied at the LDS store (see conv3x3_fwd_kernel).
  auto gload = [&](int pr) {
    const int b = pr / pairs_per_img, h0 = (pr - b * pairs_per_img) * WG_ROWS;
    const float* xb = p.x + (size_t)b * p.H * kW * C;
    const float* dyb = p.dy + ((size_t)b * p.H + h0) * kW * N;
    if (VEC) {
      // per-slot offsets / static validity were computed once (xoff, doff, xstat, xtop, xbot, dstat)
      const float* xrow = xb + (size_t)h0 * kW * C;
      xmask = xstat & ~(h0 == 0 ? xtop : 0u) & ~(h0 + WG_ROWS >= p.H ? xbot : 0u);
      dmask = dstat;
#pragma unroll
      for (int i = 0; i < XV; ++i)
        xreg[i] = *reinterpret_cast<const f32x4*>(((xmask >> i) & 1u) ? xrow + xoff[i] : p.x);
#pragma unroll
      for (int i = 0; i < DV; ++i)
        dreg[i] = *reinterpret_cast<const f32x4*>(((dmask >> i) & 1u) ? dyb + doff[i] : p.dy);
      return;
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int slot = tid + i * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (slot < (WG_ROWS + 2) * kPW * (WG_T / 4)) {
        const int q = slot & 15, pix = slot >> 4;
        const int prow = pix / kPW, pcol = pix - prow * kPW;
        const int hh = h0 + prow - 1, ww = pcol - 1;
        const int c = c0 + q * 4;
        if (hh >= 0 && hh < p.H && ww >= 0 && ww < kW && c < C) {
          const float* src = xb + ((size_t)hh * kW + ww) * C + c;
          if (vec_x) {
            v = *reinterpret_cast<const f32x4*>(src);
          } else {
            v[0] = src[0];
            if (c + 1 < C) v[1] = src[1];
            if (c + 2 < C) v[2] = src[2];
            if (c + 3 < C) v[3] = src[3];
          }
        }
      }
      xreg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int slot = tid + i * 256;
      const int q = slot & 15, pix = slot >> 4;
      const int n = n0 + q * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < N) {
        const float* src = dyb + (size_t)pix * N + n;
        if (vec_dy) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
          v[0] = src[0];
          if (n + 1 < N) v[1] = src[1];
          if (n + 2 < N) v[2] = src[2];
          if (n + 3 < N) v[3] = src[3];
        }
      }
      dreg[i] = v;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int slot = tid + i * 256;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      if (slot < (WG_ROWS + 2) * kPW * (WG_T / 4))
        *reinterpret_cast<f32x4*>(xp + slot * 4) = (!VEC || ((xmask >> i) & 1u)) ? xreg[i] : z;
    }
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(dyt + (tid + i * 256) * 4) = (!VEC || ((dmask >> i) & 1u)) ? dreg[i] : z;
    }
  };

  if (pair_begin < pair_end) {
    gload(pair_begin);
    lstore();
  }
  __syncthreads();
  for (int pr = pair_begin; pr < pair_end; ++pr) {
    const bool has_next = pr + 1 < pair_end;
    if (has_next) gload(pr + 1);          // in flight during this chunk's 288 MFMAs per wave
    __builtin_amdgcn_sched_barrier(0);
    {
      // operands of k-step s+1 are read while the 9 MFMAs of k-step s issue
      float acur[9], anext[9], bcur, bnext;
      auto lds_read = [&](int s, float* av, float& bv) {
        const int pix = 2 * s + lh;          // K index = pixel inside the 2-row chunk
        const int r = pix >> 5, ww = pix & 31;
        bv = dyt[pix * WG_T + wco * 32 + li];
        const float* xa = xp + (r * kPW + ww) * WG_T + wci * 32 + li;
#pragma unroll
        for (int t = 0; t < 9; ++t) av[t] = xa[((t / 3) * kPW + (t % 3)) * WG_T];
      };
      lds_read(0, acur, bcur);
      // explicit ping-pong (no register copies): the reads of k-step s+1 are in flight during step s's 9 MFMAs
      for (int s = 0; s < WG_ROWS * kW / 2; s += 2) {
        lds_read(s + 1, anext, bnext);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = mfma32(acur[t], bcur, acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < WG_ROWS * kW / 2) lds_read(s + 2, acur, bcur);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = mfma32(anext[t], bnext, acc[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                       // chunk fully consumed
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

// dw[e] (+)= sum_s slab[s][e]   -- fixed summation order => bitwise reproducible
__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int S, int E,
                                   int accumulate) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  float s = 0.f;
  int i = 0;
  for (; i + 8 <= S; i += 8) {           // 8 independent loads in flight, summed in index order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(i + u) * E + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < S; ++i) s += slab[(size_t)i * E + e];
  out[e] = accumulate ? out[e] + s : s;
}

// ---------------------------------------------------------------------------------------------------------------
// Thin ends of the U-Nets: conv_out (E -> 3 for the score network, E -> 1 for the encoder, ldm/model_vdm.py:378-383,
// model_mulan_epsilon.py:146-149), its input gradient (3 / 1 -> E) and its weight gradient.  On the MFMA kernels above
// these pad the thin dimension to a 32-wide tile (113 / 52 / 184 us per launch at B = 128 for 0.9 GFLOP each); they are
// reads / writes of one [B, 1024, E] tensor, so here they run on the vector ALUs in exact fp32 (v_fma_f32), lanes over
// the channel quads of a pixel, a 3 x 3 window of float4s sliding along the image row: one pass over the tensor.
constexpr int THIN_ROWS = 8;      // image rows per 256-thread block (two per wave)

// y[b, p, n] = sum_{tap, c} x[b, p + tap, c] w[tap, c, n] + bias[n] + res[b, p, n]   for NT = N <= 4, C = 128 or 256
template <int NT>
__global__ __launch_bounds__(256) void conv3x3_thin_n_fwd_kernel(ConvArgs p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = p.C, G = C >> 2;                     // lanes per pixel (32 or 64)
  const int ppw = 64 / G;                            // pixels a wave works on at a time (2 or 1)
  const int c4 = lane % G, slot = lane / G;
  const int rows_per_img = p.H / THIN_ROWS;
  const int b = blockIdx.x / rows_per_img, h0 = (blockIdx.x % rows_per_img) * THIN_ROWS;
  float wr[9][4][NT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n) wr[t][j][n] = p.w[((size_t)t * C + c4 * 4 + j) * NT + n];
  const int cols = kW / ppw;                         // consecutive columns per slot
  for (int r = wave; r < THIN_ROWS; r += 4) {
    const int h = h0 + r;
    const float* rowp[3];
    bool rok[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = h + kh - 1;
      rok[kh] = hh >= 0 && hh < p.H;
      rowp[kh] = p.x + ((size_t)(b * p.H + (rok[kh] ? hh : h)) * kW) * C + c4 * 4;
    }
    const int col0 = slot * cols;
    f32x4 win[3][3];                                 // [kh][kw]: columns col - 1, col, col + 1
    auto ld = [&](int kh, int col) {
      return (rok[kh] && col >= 0 && col < kW) ? *reinterpret_cast<const f32x4*>(rowp[kh] + (size_t)col * C)
                                               : f32x4{0.f, 0.f, 0.f, 0.f};
    };
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) { win[kh][1] = ld(kh, col0 - 1); win[kh][2] = ld(kh, col0); }
    for (int i = 0; i < cols; ++i) {
      const int col = col0 + i;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) { win[kh][0] = win[kh][1]; win[kh][1] = win[kh][2]; win[kh][2] = ld(kh, col + 1); }
      float acc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = __builtin_fmaf(win[kh][kw][j], wr[kh * 3 + kw][j][n], acc[n]);
      for (int o = 1; o < G; o <<= 1)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] += __shfl_xor(acc[n], o, 64);
      if (c4 == 0) {
        const size_t px = (size_t)(b * p.H + h) * kW + col;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          float v = acc[n];
          if (p.bias) v += p.bias[n];
          if (p.res) v += p.res[px * NT + n];
          p.y[px * NT + n] = v;
        }
      }
    }
  }
}

// y[b, p, n] = sum_{tap, c} x[b, p + tap, c] w[tap, c, n] + bias[n] + res[b, p, n]   for CT = C <= 4, N = 128 or 256:
// the input gradient of conv_out (x = dy, w = the flipped kernel)
template <int CT>
__global__ __launch_bounds__(256) void conv3x3_thin_c_fwd_kernel(ConvArgs p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = p.N, G = N >> 2;
  const int ppw = 64 / G;
  const int n4 = lane % G, slot = lane / G;
  const int rows_per_img = p.H / THIN_ROWS;
  const int b = blockIdx.x / rows_per_img, h0 = (blockIdx.x % rows_per_img) * THIN_ROWS;
  f32x4 wr[9][CT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < CT; ++c) wr[t][c] = *reinterpret_cast<const f32x4*>(p.w + ((size_t)t * CT + c) * N + n4 * 4);
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n4 * 4);
  const int cols = kW / ppw;
  for (int r = wave; r < THIN_ROWS; r += 4) {
    const int h = h0 + r;
    const float* rowp[3];
    bool rok[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = h + kh - 1;
      rok[kh] = hh >= 0 && hh < p.H;
      rowp[kh] = p.x + ((size_t)(b * p.H + (rok[kh] ? hh : h)) * kW) * CT;
    }
    const int col0 = slot * cols;
    float win[3][3][CT];
    auto ld = [&](int kh, int col, float (&dst)[CT]) {
      const bool ok = rok[kh] && col >= 0 && col < kW;
#pragma unroll
      for (int c = 0; c < CT; ++c) dst[c] = ok ? rowp[kh][(size_t)col * CT + c] : 0.f;
    };
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) { ld(kh, col0 - 1, win[kh][1]); ld(kh, col0, win[kh][2]); }
    for (int i = 0; i < cols; ++i) {
      const int col = col0 + i;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int c = 0; c < CT; ++c) { win[kh][0][c] = win[kh][1][c]; win[kh][1][c] = win[kh][2][c]; }
        ld(kh, col + 1, win[kh][2]);
      }
      f32x4 acc = bias4;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            const f32x4 w4 = wr[kh * 3 + kw][c];
            const float xv = win[kh][kw][c];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(xv, w4[e], acc[e]);
          }
      const size_t o = ((size_t)(b * p.H + h) * kW + col) * N + n4 * 4;
      if (p.res) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + o);
        acc[0] += rv[0]; acc[1] += rv[1]; acc[2] += rv[2]; acc[3] += rv[3];
      }
      *reinterpret_cast<f32x4*>(p.y + o) = acc;
    }
  }
}

// slab[block][tap][c][n] = sum over the block's rows of x[b, p + tap, c] dy[b, p, n]   for NT = N <= 4, C = 128 or 256
template <int NT>
__global__ __launch_bounds__(256) void conv3x3_thin_n_wgrad_kernel(WgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];          // [9][C][NT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = p.C, G = C >> 2;
  const int ppw = 64 / G;
  const int c4 = lane % G, slot = lane / G;
  const int rows_per_img = p.H / THIN_ROWS;
  const int b = blockIdx.x / rows_per_img, h0 = (blockIdx.x % rows_per_img) * THIN_ROWS;
  float acc[9][4][NT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[t][j][n] = 0.f;
  const int cols = kW / ppw;
  for (int r = wave; r < THIN_ROWS; r += 4) {
    const int h = h0 + r;
    const float* rowp[3];
    bool rok[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = h + kh - 1;
      rok[kh] = hh >= 0 && hh < p.H;
      rowp[kh] = p.x + ((size_t)(b * p.H + (rok[kh] ? hh : h)) * kW) * C + c4 * 4;
    }
    const float* dyrow = p.dy + ((size_t)(b * p.H + h) * kW) * NT;
    const int col0 = slot * cols;
    f32x4 win[3][3];
    auto ld = [&](int kh, int col) {
      return (rok[kh] && col >= 0 && col < kW) ? *reinterpret_cast<const f32x4*>(rowp[kh] + (size_t)col * C)
                                               : f32x4{0.f, 0.f, 0.f, 0.f};
    };
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) { win[kh][1] = ld(kh, col0 - 1); win[kh][2] = ld(kh, col0); }
    for (int i = 0; i < cols; ++i) {
      const int col = col0 + i;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) { win[kh][0] = win[kh][1]; win[kh][1] = win[kh][2]; win[kh][2] = ld(kh, col + 1); }
      float g[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) g[n] = dyrow[(size_t)col * NT + n];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[kh * 3 + kw][j][n] = __builtin_fmaf(win[kh][kw][j], g[n], acc[kh * 3 + kw][j][n]);
    }
  }
  // the pixel slots of a wave, then the four waves in a fixed order through LDS
  for (int o = G; o < 64; o <<= 1)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][j][n] += __shfl_xor(acc[t][j][n], o, 64);
  for (int w = 0; w < 4; ++w) {
    if (wave == w && slot == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            float* d = tsm + ((size_t)t * C + c4 * 4 + j) * NT + n;
            *d = w == 0 ? acc[t][j][n] : *d + acc[t][j][n];
          }
    }
    __syncthreads();
  }
  const int E = 9 * C * NT;
  float* slab = p.slab + (size_t)blockIdx.x * E;
  for (int e = tid; e < E; e += 256) slab[e] = tsm[e];
}

bool thin_n_ok(int H, int C, int N) { return N >= 1 && N <= 4 && (C == 128 || C == 256) && H % THIN_ROWS == 0; }
bool thin_c_ok(int H, int C, int N) { return C >= 1 && C <= 4 && (N == 128 || N == 256) && H % THIN_ROWS == 0; }

int wgrad_splits(int B, int H, int C, int N) {
  const int tiles = ((C + WG_T - 1) / WG_T) * ((N + WG_T - 1) / WG_T);
  const int pairs = B * (H / WG_ROWS);
  const int target = g_mulan_tune[1] > 0 ? g_mulan_tune[1] : 256;   // one fully pipelined block per CU
  int S = target / tiles;
  if (S < 1) S = 1;
  if (S > pairs) S = pairs;
  // keep at least 4 row pairs per split so the slab traffic stays small next to the MFMA work
  while (S > 1 && pairs / S < 4) --S;
  return S;
}

}  // namespace

MULAN_API int mulan_conv3x3_fwd(const float* x, const float* w, const float* bias, const float* cbias,
                                int cbias_mode, const float* res, float* y, int B, int H, int W, int C, int N,
                                hipStream_t stream) {
  if (W != kW || H % TROWS != 0 || B <= 0 || C <= 0 || N <= 0) return (int)hipErrorInvalidValue;
  ConvArgs a{x, w, bias, cbias, res, y, B, H, C, N, cbias ? cbias_mode : 0, g_mulan_debug_buffer};
  if (!cbias && g_mulan_tune[8] != 1) {   // thin ends (conv_out and its input gradient): vector-ALU kernels; tune[8] = 1: dev A/B
    const dim3 tgrid(B * (H / THIN_ROWS));
    if (thin_n_ok(H, C, N)) {
      switch (N) {
        case 1: hipLaunchKernelGGL(conv3x3_thin_n_fwd_kernel<1>, tgrid, dim3(256), 0, stream, a); break;
        case 2: hipLaunchKernelGGL(conv3x3_thin_n_fwd_kernel<2>, tgrid, dim3(256), 0, stream, a); break;
        case 3: hipLaunchKernelGGL(conv3x3_thin_n_fwd_kernel<3>, tgrid, dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL(conv3x3_thin_n_fwd_kernel<4>, tgrid, dim3(256), 0, stream, a); break;
      }
      MULAN_CHECK_LAUNCH();
    }
    if (thin_c_ok(H, C, N) && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
      switch (C) {
        case 1: hipLaunchKernelGGL(conv3x3_thin_c_fwd_kernel<1>, tgrid, dim3(256), 0, stream, a); break;
        case 2: hipLaunchKernelGGL(conv3x3_thin_c_fwd_kernel<2>, tgrid, dim3(256), 0, stream, a); break;
        case 3: hipLaunchKernelGGL(conv3x3_thin_c_fwd_kernel<3>, tgrid, dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL(conv3x3_thin_c_fwd_kernel<4>, tgrid, dim3(256), 0, stream, a); break;
      }
      MULAN_CHECK_LAUNCH();
    }
  }
  const int mtiles = B * (H / TROWS);
  const auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool vec = (C % 4 == 0) && (N % 4 == 0) && al(x) && al(w);
#define MULAN_CONV_LAUNCH(BN_, WM_, WN_, GY)                                                              \
  do {                                                                                                    \
    dim3 grid(mtiles, GY);                                                                                \
    if (vec) hipLaunchKernelGGL((conv3x3_fwd_kernel<BN_, WM_, WN_, true, 1>), grid, dim3(256), 0, stream, a); \
    else hipLaunchKernelGGL((conv3x3_fwd_kernel<BN_, WM_, WN_, false, 1>), grid, dim3(256), 0, stream, a);    \
  } while (0)
  if (N > 64 && vec && g_mulan_tune[0] == 3) {
    dim3 grid(mtiles, (N + 127) / 128);
    hipLaunchKernelGGL((conv3x3_fwd_kernel<128, 2, 2, true, 3>), grid, dim3(256), 0, stream, a);
  } else if (N > 64) MULAN_CONV_LAUNCH(128, 2, 2, (N + 127) / 128);
  else if (N > 32) MULAN_CONV_LAUNCH(64, 2, 2, 1);
  else MULAN_CONV_LAUNCH(32, 4, 1, 1);
#undef MULAN_CONV_LAUNCH
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_conv3x3_wflip(const float* w, float* wT, int C, int N, hipStream_t stream) {
  const int total = 9 * C * N;
  const int blocks = min(2048, (total + 255) / 256);
  hipLaunchKernelGGL(conv3x3_wflip_kernel, dim3(blocks), dim3(256), 0, stream, w, wT, C, N);
  MULAN_CHECK_LAUNCH();
}

MULAN_API size_t mulan_conv3x3_wgrad_workspace(int B, int H, int W, int C, int N) {
  if (W != kW || H % WG_ROWS != 0) return 0;
  if (thin_n_ok(H, C, N) && g_mulan_tune[8] != 1) return (size_t)B * (H / THIN_ROWS) * 9 * C * N * sizeof(float);
  return (size_t)wgrad_splits(B, H, C, N) * 9 * C * N * sizeof(float);
}

MULAN_API int mulan_conv3x3_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int H, int W,
                                  int C, int N, int accumulate, hipStream_t stream) {
  if (W != kW || H % WG_ROWS != 0 || B <= 0) return (int)hipErrorInvalidValue;
  if (thin_n_ok(H, C, N) && g_mulan_tune[8] != 1) {   // conv_out: vector-ALU kernel, one slab per block of 8 image rows
    const int S = B * (H / THIN_ROWS), E = 9 * C * N;
    WgradArgs a{x, dy, workspace, B, H, C, N, S};
    const size_t lds = (size_t)E * sizeof(float);
    switch (N) {
      case 1: hipLaunchKernelGGL(conv3x3_thin_n_wgrad_kernel<1>, dim3(S), dim3(256), lds, stream, a); break;
      case 2: hipLaunchKernelGGL(conv3x3_thin_n_wgrad_kernel<2>, dim3(S), dim3(256), lds, stream, a); break;
      case 3: hipLaunchKernelGGL(conv3x3_thin_n_wgrad_kernel<3>, dim3(S), dim3(256), lds, stream, a); break;
      default: hipLaunchKernelGGL(conv3x3_thin_n_wgrad_kernel<4>, dim3(S), dim3(256), lds, stream, a); break;
    }
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, workspace, dw, S, E, accumulate);
    MULAN_CHECK_LAUNCH();
  }
  const int S = wgrad_splits(B, H, C, N);
  WgradArgs a{x, dy, workspace, B, H, C, N, S};
  dim3 grid(S, (C + WG_T - 1) / WG_T, (N + WG_T - 1) / WG_T);
  const auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if ((C % 4 == 0) && (N % 4 == 0) && al(x) && al(dy))
    hipLaunchKernelGGL(conv3x3_wgrad_kernel<true>, grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(conv3x3_wgrad_kernel<false>, grid, dim3(256), 0, stream, a);
  const int E = 9 * C * N;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, workspace, dw, S, E,
                     accumulate);
  MULAN_CHECK_LAUNCH();
}
