// Fused single-head self-attention over 1024 positions (AttnBlock core, ldm/model_vdm.py:679-683,704-802:
// softmax((q / sqrt(C)) k^T) v and its gradients) with fp32-equivalent products on the fp16 matrix cores ("f16x3", see
// conv3x3_f16x3.hip).  The [1024 x 1024] score / probability matrices never reach HBM: forward keeps the online-softmax
// state and writes o and the per-query log-sum-exp; backward recomputes the probabilities from q, k and that.
//
// One skeleton, three kernels.  A wave owns 32 "columns" (queries, or keys) that sit on its lanes for the whole kernel
// and sweeps the other axis in tiles of 32 "rows":
//     T1 = X Y^T                    [rows x columns]   scores                       (3 MFMA passes per 16-deep k step)
//     T2 = X2 Y2^T                  [rows x columns]   dP = dO v^T                   (backward only)
//     E  = f(T1, T2)                elementwise, in the accumulator layout (column on the lane, 16 rows in registers)
//     Out^T += Z^T E                [d x columns]      E is used as the B operand straight from its registers
// MODE 0 (forward):  columns = queries; X = k, Y = q; E = P (online softmax);               Z = v      -> o, lse
// MODE 1 (dq):       columns = queries; X = k, Y = q, X2 = v, Y2 = do; E = dS = P (dP - delta); Z = k  -> dq
// MODE 2 (dk, dv):   columns = keys;    X = q, Y = k, X2 = do, Y2 = v; E = P -> Z = do -> dv;  E2 = dS, Z2 = q -> dk
// Operands arrive pre-split (mulan_linear_pack_f16x3_batched, two layouts of the same [1024, C] tensor):
//   "T" pack [C/16][1024][plane][16 channels]  (contraction over channels: X, Y, X2, Y2)
//   "N" pack [1024/16][C][plane][16 rows]      (contraction over rows: Z, Z2)
// with one power-of-two scale per image and tensor.  P (<= 1) is split with the fixed scale 2^13; dS with a scale from
// the a-priori bound C max|do| max|v| + max|delta| (the split is accurate relative to each element: a loose bound only
// raises the absolute error floor, 2^-38 of the bound).
// Block = 4 waves = 128 columns; X / X2 / Z / Z2 tiles of the current 32 rows go through LDS (next tile's global loads
// are in flight in registers meanwhile), the column-side fragments Y / Y2 stay in registers.  C = 128: the three kernels
// above.  C = 256 (the ImageNet-32 width): the forward kernel as it is (one block per CU at ~370 registers); in the
// backward pass Y, Y2 and the output accumulators of MODE 1 / 2 would need 128 + 128 + 128 (+ 128) registers, so two
// waves share a set of 32 columns (KSPLIT = 2; a block = 2 x 2 waves = 64 columns): each holds the Y / Y2 fragments of
// HALF the channels (64 + 64 registers), forms its half of the contraction T1 / T2, the two partial tiles are exchanged
// through LDS and added (a + b == b + a: both waves hold the same sums), both evaluate E, and each accumulates HALF the
// output channels (64 registers).  No product is computed twice.  MODE 2 is split by product to fit the LDS:
//   MODE 3 (dv):  columns = keys; X = q, Y = k;                    E = P;  Z = do  -> dv   (no T2)
//   MODE 4 (dk):  columns = keys; X = q, Y = k, X2 = do, Y2 = v;   E = dS; Z = q   -> dk
// -- three launches instead of two (the scores are formed once more than an unsplit kernel would), and no
// [B, 1024, 1024] tensor in HBM (the unfused path a training step at E = 256 used until round 4 materialised S, P, dP
// and dS: 16 GiB at B = 1024).  OSPLIT (output channels split over blockIdx.z, the scores recomputed by every part)
// is the simpler variant measured first: 3.6 ms per backward call at B = 128 against 2.4 unfused
// (profiles/r04_attention_probe_c256.log); kept as a template parameter.
#include "common.h"
#include "f16x3_common.h"

namespace {

using namespace f16x3;

constexpr int AS = 1024;                 // positions
constexpr int XP = 80;                   // LDS pitch of a T-packed row (64 B + 16: conflict-free ds_read_b128, see PIXB)
constexpr int ZP = 72;                   // LDS pitch of an N-packed row (conflict-free ds_read_b64 over 32 lanes)
constexpr float kPScale = 8192.f, kPInv = 1.f / 8192.f;

struct AttnArgs {
  const unsigned char* qt; const unsigned char* kt; const unsigned char* vt; const unsigned char* dot;   // T packs
  const unsigned char* qn; const unsigned char* kn; const unsigned char* vn; const unsigned char* don;   // N packs
  const unsigned* qmax; const unsigned* kmax; const unsigned* vmax; const unsigned* domax; const unsigned* dmax;
  const float* lse; const float* delta;      // [B, 1024] (backward inputs)
  float* out; float* out2; float* lse_out;   // MODE 0: o, -, lse; MODE 1: dq; MODE 2: dv (out), dk (out2)
  float alpha;                               // 1 / sqrt(C)
  int B;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int MODE, int AC = 128, int OSPLIT = 1, int KSPLIT = 1>
__global__ __launch_bounds__(256, (MODE == 0 && AC == 128) ? 2 : 1) void attn_f16x3_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(AC == 128 || AC == 256, "C = 128 or 256");
  static_assert(MODE <= 2 || AC == 256, "MODE 3 / 4: the split form of MODE 2 for C = 256");
  static_assert(KSPLIT == 1 || (KSPLIT == 2 && OSPLIT == 1 && MODE != 0 && MODE != 2), "KSPLIT = 2: MODE 1 / 3 / 4");
  constexpr int NCH = AC / 16;             // channel chunks (8 / 16)
  constexpr int NCW = NCH / KSPLIT;        // ... contracted by one wave
  constexpr int OC = AC / OSPLIT;          // output channels of this block (blockIdx.z-th part of the AC)
  constexpr int NDT = OC / 32 / KSPLIT;    // output-channel tiles of one wave
  constexpr int X_BYTES = NCH * 32 * XP;   // 20480 / 40960
  constexpr int Z_BYTES = 2 * OC * ZP;     // 18432 / 36864
  constexpr int NPX = NCH / 2;             // 16-byte pieces per thread and T-packed tile (NCH * 128 pieces)
  constexpr int NPZ = OC / 32;             // ... and N-packed tile (8 OC pieces)
  constexpr bool BWD = MODE != 0;          // (the staging / fragment code below reads BWD as "there is a T2")
  constexpr bool KEYS = MODE >= 2;         // the columns are keys
  constexpr bool T2ON = MODE == 1 || MODE == 2 || MODE == 4;
  constexpr int NZ = MODE == 2 ? 2 : 1;
  unsigned char* xs = smem;                               // X tile
  unsigned char* x2s = smem + X_BYTES;                    // X2 tile (backward)
  unsigned char* zs = smem + (T2ON ? 2 : 1) * X_BYTES;    // Z tile(s)
  float* rowvals = reinterpret_cast<float*>(zs + NZ * Z_BYTES);   // KEYS: lse / delta of the tile's 32 rows
  unsigned char* exch = zs + NZ * Z_BYTES + 256;                  // KSPLIT = 2: partial T1 / T2 tiles, 4 (8) KB per wave

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int kpart = KSPLIT == 2 ? (wave & 1) : 0;          // which half of the channels this wave contracts / accumulates
  const int cb = kpart * NCW;                              // its first channel chunk
  const int oc0 = blockIdx.z * OC + kpart * (OC / KSPLIT); // its first output channel
  const int col0 = blockIdx.x * (128 / KSPLIT) + (wave / KSPLIT) * 32;
  const size_t img = (size_t)AS * AC * 4;                 // bytes of one packed image
  // operands by role
  const unsigned char* X = (KEYS ? p.qt : p.kt) + b * img;
  const unsigned char* Y = (KEYS ? p.kt : p.qt) + b * img;
  const unsigned char* X2 = (KEYS ? p.dot : p.vt) + b * img;
  const unsigned char* Y2 = (KEYS ? p.vt : p.dot) + b * img;
  const unsigned char* Z = (MODE == 0 ? p.vn : (MODE == 1 ? p.kn : (MODE == 4 ? p.qn : p.don))) + b * img;
  const unsigned char* Z2 = p.qn + b * img;

  float s_, inv_q, inv_k, inv_v, inv_do = 0.f, s_ds = 0.f, inv_ds = 0.f;
  scale_of(row_max16(p.qmax, b), s_, inv_q);
  scale_of(row_max16(p.kmax, b), s_, inv_k);
  scale_of(row_max16(p.vmax, b), s_, inv_v);
  if (BWD) {
    scale_of(row_max16(p.domax, b), s_, inv_do);
    const float bound = (float)AC * __uint_as_float(row_max16(p.domax, b)) * __uint_as_float(row_max16(p.vmax, b)) +
                        __uint_as_float(row_max16(p.dmax, b));
    scale_of(__float_as_uint(bound), s_ds, inv_ds);
  }
  const float c1 = p.alpha * inv_q * inv_k;               // accumulator of T1 -> alpha-scaled score
  const float c2 = inv_v * inv_do;                        // accumulator of T2 -> dP

  // ---- column-side fragments (B operand of T1 / T2): lane = column r, channels 8 h .. 8 h + 7 of chunk c
  f16x8 yf[NCW][2], y2f[T2ON ? NCW : 1][2];
#pragma unroll
  for (int c = 0; c < NCW; ++c)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const size_t off = (((size_t)(cb + c) * AS + col0 + r) * 2 + pl) * 32 + h * 16;
      yf[c][pl] = *reinterpret_cast<const f16x8*>(Y + off);
      if (T2ON) y2f[c][pl] = *reinterpret_cast<const f16x8*>(Y2 + off);
    }
  float col_l = 0.f, col_d = 0.f;                         // MODE 1: lse / delta of this lane's column
  if (MODE == 1) {
    col_l = p.lse[(size_t)b * AS + col0 + r];
    col_d = p.delta[(size_t)b * AS + col0 + r];
  }

  f32x16 out[NDT], out2[MODE == 2 ? NDT : 1];
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) { out[i][e] = 0.f; if (MODE == 2) out2[i][e] = 0.f; }
  float m_run = -3.0e38f, l_run = 0.f;                    // MODE 0: online softmax state of this lane's column (own half of the rows)

  // ---- staging: 16-byte pieces, NPX / NPZ per thread and tile (C = 128: 1024 pieces = 16 KB per tile)
  constexpr int OX2 = NPX, OZ = (T2ON ? 2 : 1) * NPX, OZ2 = OZ + NPZ;
  constexpr int NSTG = OZ + NZ * NPZ;
  i32x4 stg[NSTG];
  auto gload = [&](int t) {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int idx = tid + 256 * i;
      // T pack: chunk c = idx >> 7, row = (idx >> 2) & 31, piece = idx & 3
      const size_t xo = (((size_t)(idx >> 7) * AS + t * 32 + ((idx >> 2) & 31)) * 64) + (idx & 3) * 16;
      stg[i] = *reinterpret_cast<const i32x4*>(X + xo);
      if (T2ON) stg[OX2 + i] = *reinterpret_cast<const i32x4*>(X2 + xo);
    }
#pragma unroll
    for (int i = 0; i < NPZ; ++i) {
      const int idx = tid + 256 * i;
      // N pack: the two 16-row chunks of the tile are [2][C][64 B]; this block's OC channels start at oc0
      const int rc2 = idx / (OC * 4), rem = idx - rc2 * (OC * 4);
      const size_t zo = ((size_t)(t * 2 + rc2) * AC + blockIdx.z * OC) * 64 + (size_t)rem * 16;
      stg[OZ + i] = *reinterpret_cast<const i32x4*>(Z + zo);
      if (MODE == 2) stg[OZ2 + i] = *reinterpret_cast<const i32x4*>(Z2 + zo);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int idx = tid + 256 * i;
      const int xd = ((idx >> 7) * 32 + ((idx >> 2) & 31)) * XP + (idx & 3) * 16;
      *reinterpret_cast<i32x4*>(xs + xd) = stg[i];
      if (T2ON) *reinterpret_cast<i32x4*>(x2s + xd) = stg[OX2 + i];
    }
#pragma unroll
    for (int i = 0; i < NPZ; ++i) {
      const int idx = tid + 256 * i;
      const int zd = (idx >> 2) * ZP + (idx & 3) * 16;       // 72-byte pitch: 8-byte aligned stores
      const i32x4 z = stg[OZ + i];
      *reinterpret_cast<i32x2*>(zs + zd) = i32x2{z[0], z[1]};
      *reinterpret_cast<i32x2*>(zs + zd + 8) = i32x2{z[2], z[3]};
      if (MODE == 2) {
        const i32x4 z2 = stg[OZ2 + i];
        *reinterpret_cast<i32x2*>(zs + Z_BYTES + zd) = i32x2{z2[0], z2[1]};
        *reinterpret_cast<i32x2*>(zs + Z_BYTES + zd + 8) = i32x2{z2[2], z2[3]};
      }
    }
  };
  // Z fragment (A operand of the output product): lane = channel row r of tile dt, k step s of the 32-row tile; the
  // accumulator-as-operand k order (cdna guide, "An accumulator tile as the next MFMA's operand"): element j is row
  // 16 s + 8 (j >> 2) + 4 h + (j & 3)  ->  two 8-byte pieces of the 16 packed rows
  auto zfrag = [&](const unsigned char* zb, int s, int dt, int pl) {
    const unsigned char* a = zb + (s * OC + kpart * (OC / KSPLIT) + dt * 32 + r) * ZP + pl * 32 + h * 8;
    const s16x4 lo = *reinterpret_cast<const s16x4*>(a);
    const s16x4 hi = *reinterpret_cast<const s16x4*>(a + 16);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, v);
  };
  // registers 8 s .. 8 s + 7 of an fp32 tile -> the two fp16 pieces of k step s (B operand)
  auto esplit = [&](const f32x16& e, float scale, int s, f16x8& eh, f16x8& el) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 hh, ll;
      split2(e[8 * s + j] * scale, hh, ll);
      eh[j] = hh; el[j] = ll;
    }
  };

  gload(0);
  lstore();
  if (KEYS && tid < 64) rowvals[tid] = (tid < 32 ? p.lse : p.delta)[(size_t)b * AS + (tid & 31)];
  __syncthreads();

  for (int t = 0; t < AS / 32; ++t) {
    if (t + 1 < AS / 32) gload(t + 1);
    float nxt_rv = 0.f;
    if (KEYS && tid < 64 && t + 1 < AS / 32) nxt_rv = (tid < 32 ? p.lse : p.delta)[(size_t)b * AS + (t + 1) * 32 + (tid & 31)];

    // ---- T1 (and T2): contraction over the channels
    f32x16 t1, t2;
#pragma unroll
    for (int e = 0; e < 16; ++e) { t1[e] = 0.f; t2[e] = 0.f; }
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
      const unsigned char* xa = xs + ((cb + c) * 32 + r) * XP + h * 16;
      const f16x8 xh = *reinterpret_cast<const f16x8*>(xa), xl = *reinterpret_cast<const f16x8*>(xa + 32);
      t1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yf[c][0], t1, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yf[c][1], t1, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yf[c][0], t1, 0, 0, 0);
      if (T2ON) {
        const unsigned char* x2a = x2s + ((cb + c) * 32 + r) * XP + h * 16;
        const f16x8 x2h = *reinterpret_cast<const f16x8*>(x2a), x2l = *reinterpret_cast<const f16x8*>(x2a + 32);
        t2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x2l, y2f[c][0], t2, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x2h, y2f[c][1], t2, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x2h, y2f[c][0], t2, 0, 0, 0);
      }
    }

    if constexpr (KSPLIT == 2) {
      // the partner wave (wave ^ 1) contracted the other half of the channels: exchange the partial tiles and add
      constexpr int NQ = T2ON ? 8 : 4;                        // float4 quads per lane
      f32x4* mine = reinterpret_cast<f32x4*>(exch) + (size_t)wave * NQ * 64 + lane;
      const f32x4* theirs = reinterpret_cast<const f32x4*>(exch) + (size_t)(wave ^ 1) * NQ * 64 + lane;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        mine[qd * 64] = f32x4{t1[4 * qd], t1[4 * qd + 1], t1[4 * qd + 2], t1[4 * qd + 3]};
        if (T2ON) mine[(4 + qd) * 64] = f32x4{t2[4 * qd], t2[4 * qd + 1], t2[4 * qd + 2], t2[4 * qd + 3]};
      }
      __syncthreads();
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 a = theirs[qd * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) t1[4 * qd + e] += a[e];
        if (T2ON) {
          const f32x4 a2 = theirs[(4 + qd) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) t2[4 * qd + e] += a2[e];
        }
      }
    }

    // ---- elementwise: register e of a tile is row (e & 3) + 8 (e >> 2) + 4 h, column r
    f32x16 E, E2;
    float escale, e2scale = 0.f;
    if (MODE == 0) {
      float tmax = -3.0e38f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { t1[e] *= c1; tmax = fmaxf(tmax, t1[e]); }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));           // the other half of this column's rows
      const float m_new = fmaxf(m_run, tmax);
      const float corr = __expf(m_run - m_new);
      float lsum = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { E[e] = __expf(t1[e] - m_new); lsum += E[e]; }
      l_run = l_run * corr + lsum;
      m_run = m_new;
      if (__builtin_amdgcn_readfirstlane(__any(corr != 1.f))) {
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) out[i][e] *= corr;
      }
      escale = kPScale;
    } else if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) E[e] = __expf(t1[e] * c1 - col_l) * (t2[e] * c2 - col_d);
      escale = s_ds;
    } else {
      f32x4 lr[4], dr[4];                                       // lse / delta of this lane's 16 rows
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        lr[g] = *reinterpret_cast<const f32x4*>(rowvals + 8 * g + 4 * h);
        dr[g] = *reinterpret_cast<const f32x4*>(rowvals + 32 + 8 * g + 4 * h);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        E[e] = __expf(t1[e] * c1 - lr[e >> 2][e & 3]);
        if (MODE == 2) E2[e] = E[e] * (t2[e] * c2 - dr[e >> 2][e & 3]);
        if (MODE == 4) E[e] = E[e] * (t2[e] * c2 - dr[e >> 2][e & 3]);
      }
      escale = MODE == 4 ? s_ds : kPScale;
      e2scale = s_ds;
    }

    // ---- output product(s): contraction over the 32 rows of the tile = 2 k steps
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 eh, el, e2h, e2l;
      esplit(E, escale, s, eh, el);
      if (MODE == 2) esplit(E2, e2scale, s, e2h, e2l);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const f16x8 zh = zfrag(zs, s, dt, 0), zl = zfrag(zs, s, dt, 1);
        out[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zl, eh, out[dt], 0, 0, 0);
        out[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, el, out[dt], 0, 0, 0);
        out[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, eh, out[dt], 0, 0, 0);
        if (MODE == 2) {
          const f16x8 z2h = zfrag(zs + Z_BYTES, s, dt, 0), z2l = zfrag(zs + Z_BYTES, s, dt, 1);
          out2[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(z2l, e2h, out2[dt], 0, 0, 0);
          out2[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(z2h, e2l, out2[dt], 0, 0, 0);
          out2[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(z2h, e2h, out2[dt], 0, 0, 0);
        }
      }
    }

    __syncthreads();                                          // every wave is done with this tile's LDS image
    if (t + 1 < AS / 32) {
      lstore();
      if (KEYS && tid < 64) rowvals[tid] = nxt_rv;
    }
    __syncthreads();
  }

  // ---- epilogue: Out^T is [channel rows (registers) x column (lane)]; register e of tile dt is channel
  // dt * 32 + (e & 3) + 8 (e >> 2) + 4 h: four consecutive channels per register quad -> float4 stores
  float oscale, o2scale = 0.f;
  if (MODE == 0) {
    l_run += __shfl_xor(l_run, 32, 64);
    oscale = kPInv * inv_v / l_run;
    if (h == 0) p.lse_out[(size_t)b * AS + col0 + r] = m_run + __logf(l_run);
  } else if (MODE == 1) {
    oscale = p.alpha * inv_ds * inv_k;
  } else if (MODE == 4) {
    oscale = p.alpha * inv_ds * inv_q;                        // dk = alpha dS^T q
  } else {
    oscale = kPInv * inv_do;                                  // dv = P^T do
    o2scale = p.alpha * inv_ds * inv_q;                       // dk = alpha dS^T q
  }
  float* orow = p.out + ((size_t)b * AS + col0 + r) * AC + oc0;
  float* orow2 = MODE == 2 ? p.out2 + ((size_t)b * AS + col0 + r) * AC + oc0 : nullptr;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = dt * 32 + 8 * g + 4 * h;
      f32x4 v = {out[dt][4 * g] * oscale, out[dt][4 * g + 1] * oscale, out[dt][4 * g + 2] * oscale, out[dt][4 * g + 3] * oscale};
      *reinterpret_cast<f32x4*>(orow + ch) = v;
      if (MODE == 2) {
        f32x4 w = {out2[dt][4 * g] * o2scale, out2[dt][4 * g + 1] * o2scale, out2[dt][4 * g + 2] * o2scale,
                   out2[dt][4 * g + 3] * o2scale};
        *reinterpret_cast<f32x4*>(orow2 + ch) = w;
      }
    }
}

// delta[b, i] = sum_c do[b, i, c] * o[b, i, c]   (the row term of the softmax gradient; C = 128 or 256: 32 lanes per
// row, one or two float4 per lane)
template <int AC>
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ dout, const float* __restrict__ o,
                                                         float* __restrict__ delta, size_t rows) {
  const size_t row = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int l = threadIdx.x & 31;
  if (row >= rows) return;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < AC / 128; ++i) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(dout + row * AC + i * 128 + l * 4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(o + row * AC + i * 128 + l * 4);
    s += (a[0] * c[0] + a[1] * c[1]) + (a[2] * c[2] + a[3] * c[3]);
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (l == 0) delta[row] = s;
}

// Both packs of x [B, 1024, C] in one pass over x: "T" [C/16][1024][plane][16 channels] and "N"
// [1024/16][C][plane][16 rows] (either may be omitted).  A block takes 16 rows x C channels: the split values go
// through LDS once as fp16 [plane][row][channel] and leave as 16-byte pieces in both orders.
template <int AC>
__global__ __launch_bounds__(256) void attn_pack_kernel(const float* __restrict__ x, const unsigned* __restrict__ xmax,
                                                        unsigned char* __restrict__ xt, unsigned char* __restrict__ xn) {
  constexpr int NV = AC / 64;                                             // float4s / pieces per thread (2 or 4)
  __shared__ __attribute__((aligned(16))) _Float16 sp[2][16][AC + 8];     // +8 halves: rows 16 B apart in banks
  const int tid = threadIdx.x;
  const int b = blockIdx.y, rc = blockIdx.x;                // row chunk of 16 rows
  float sx, inv;
  scale_of(row_max16(xmax, b), sx, inv);
  const float* src = x + ((size_t)b * AS + rc * 16) * AC;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + 256 * i;                          // float4 number idx of the 16 x C slab
    const int row = idx / (AC / 4), c4 = (idx % (AC / 4)) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + row * AC + c4);
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 h, l;
      split2(v[e] * sx, h, l);
      hi[e] = h; lo[e] = l;
    }
    *reinterpret_cast<f16x4*>(&sp[0][row][c4]) = hi;
    *reinterpret_cast<f16x4*>(&sp[1][row][c4]) = lo;
  }
  __syncthreads();
  const size_t img = (size_t)AS * AC * 4;
  if (xt) {       // pieces (chunk c, row, plane, half): 16 B = 8 channels; for a chunk the 16 rows are 1 KB contiguous
    unsigned char* dst = xt + b * img;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int pc = tid + 256 * i;
      const int c = pc >> 6, row = (pc >> 2) & 15, pl = (pc >> 1) & 1, half = pc & 1;
      const i32x4 v = *reinterpret_cast<const i32x4*>(&sp[pl][row][c * 16 + half * 8]);
      *reinterpret_cast<i32x4*>(dst + (((size_t)c * AS + rc * 16 + row) * 2 + pl) * 32 + half * 16) = v;
    }
  }
  if (xn) {       // pieces (channel, plane, half): 8 rows of one channel; the whole row chunk is 8 KB contiguous
    unsigned char* dst = xn + b * img + (size_t)rc * AC * 64;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int pc = tid + 256 * i;
      const int ch = pc >> 2, pl = (pc >> 1) & 1, half = pc & 1;
      f16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = sp[pl][half * 8 + j][ch];
      *reinterpret_cast<f16x8*>(dst + (size_t)pc * 16) = v;
    }
  }
}

template <int MODE, int AC = 128, int OSPLIT = 1, int KSPLIT = 1>
int launch(const AttnArgs& a, hipStream_t stream) {
  constexpr bool t2on = MODE == 1 || MODE == 2 || MODE == 4;
  constexpr int smem = (t2on ? 2 : 1) * (AC / 16) * 32 * XP + (MODE == 2 ? 2 : 1) * 2 * (AC / OSPLIT) * ZP + 256 +
                       (KSPLIT == 2 ? 4 * (t2on ? 8192 : 4096) : 0);
  static_assert(smem <= 160 * 1024, "LDS");
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f16x3_kernel<MODE, AC, OSPLIT, KSPLIT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  hipLaunchKernelGGL((attn_f16x3_kernel<MODE, AC, OSPLIT, KSPLIT>), dim3(AS / (128 / KSPLIT), a.B, OSPLIT), dim3(256), smem,
                     stream, a);
  return (int)hipGetLastError();
}

}  // namespace

// Forward: o[B,1024,C] = softmax(alpha q k^T) v, lse[B,1024] = log sum_j exp(alpha q_i k_j); C = 128 or 256.
// qt / kt: "T" packs of q / k (mulan_linear_pack_f16x3_batched(x, K = C, N = 1024, transpose = 1)); vn: "N" pack of v
// (K = 1024, N = C, transpose = 0); *max: the per-image maxima the packs were scaled with ([B][16]).
MULAN_API int mulan_attention_fwd_f16x3(const void* qt, const void* kt, const void* vn, const unsigned* qmax,
                                        const unsigned* kmax, const unsigned* vmax, float* o, float* lse, int B, int S,
                                        int C, float alpha, hipStream_t stream) {
  if (S != AS || (C != 128 && C != 256) || B <= 0 || B > 65535 || !qt || !kt || !vn || !qmax || !kmax || !vmax || !o || !lse)
    return (int)hipErrorInvalidValue;
  AttnArgs a{};
  a.qt = static_cast<const unsigned char*>(qt); a.kt = static_cast<const unsigned char*>(kt);
  a.vn = static_cast<const unsigned char*>(vn);
  a.qmax = qmax; a.kmax = kmax; a.vmax = vmax;
  a.out = o; a.lse_out = lse; a.alpha = alpha; a.B = B;
  return C == 128 ? launch<0, 128>(a, stream) : launch<0, 256>(a, stream);
}

// Both packs of x [B,1024,C] (C = 128 or 256) for the attention kernels in one pass (xt and / or xn; each 4 B per element),
// equal bit for bit to mulan_linear_pack_f16x3_batched(x, K = C, N = 1024, transpose = 1) and (K = 1024, N = C, transpose = 0).
MULAN_API int mulan_attention_pack_f16x3(const float* x, const unsigned* xmax, void* xt, void* xn, int B, int S, int C,
                                         hipStream_t stream) {
  if (S != AS || (C != 128 && C != 256) || B <= 0 || B > 65535 || !x || !xmax || (!xt && !xn)) return (int)hipErrorInvalidValue;
  if (C == 128)
    hipLaunchKernelGGL(attn_pack_kernel<128>, dim3(AS / 16, B), dim3(256), 0, stream, x, xmax,
                       static_cast<unsigned char*>(xt), static_cast<unsigned char*>(xn));
  else
    hipLaunchKernelGGL(attn_pack_kernel<256>, dim3(AS / 16, B), dim3(256), 0, stream, x, xmax,
                       static_cast<unsigned char*>(xt), static_cast<unsigned char*>(xn));
  MULAN_CHECK_LAUNCH();
}

// delta[B,1024] = rowsum(do * o): input of mulan_attention_bwd_f16x3 (and its maxima, mulan_absmax_rows(delta, B rows))
MULAN_API int mulan_attention_delta(const float* dout, const float* o, float* delta, int B, int S, int C,
                                    hipStream_t stream) {
  if (S != AS || (C != 128 && C != 256) || B <= 0) return (int)hipErrorInvalidValue;
  const size_t rows = (size_t)B * S;
  if (C == 128)
    hipLaunchKernelGGL(attn_delta_kernel<128>, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, stream, dout, o, delta, rows);
  else
    hipLaunchKernelGGL(attn_delta_kernel<256>, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, stream, dout, o, delta, rows);
  MULAN_CHECK_LAUNCH();
}

// Backward: dq, dk, dv from the packs of q, k, v, do (both layouts each where listed), lse of the forward pass and
// delta.  C = 128: two launches, the query-major kernel (dq) and the key-major kernel (dk, dv); the probabilities are
// recomputed in both.  C = 256: three launches -- dq and dk with the output channels split over two blocks per column
// tile, dv alone (see the header of this file).
MULAN_API int mulan_attention_bwd_f16x3(const void* qt, const void* qn, const void* kt, const void* kn, const void* vt,
                                        const void* dot, const void* don, const unsigned* qmax, const unsigned* kmax,
                                        const unsigned* vmax, const unsigned* domax, const unsigned* dmax,
                                        const float* lse, const float* delta, float* dq, float* dk, float* dv, int B,
                                        int S, int C, float alpha, hipStream_t stream) {
  if (S != AS || (C != 128 && C != 256) || B <= 0 || B > 65535 || !qt || !qn || !kt || !kn || !vt || !dot || !don || !qmax ||
      !kmax || !vmax || !domax || !dmax || !lse || !delta || !dq || !dk || !dv)
    return (int)hipErrorInvalidValue;
  AttnArgs a{};
  a.qt = static_cast<const unsigned char*>(qt); a.qn = static_cast<const unsigned char*>(qn);
  a.kt = static_cast<const unsigned char*>(kt); a.kn = static_cast<const unsigned char*>(kn);
  a.vt = static_cast<const unsigned char*>(vt);
  a.dot = static_cast<const unsigned char*>(dot); a.don = static_cast<const unsigned char*>(don);
  a.qmax = qmax; a.kmax = kmax; a.vmax = vmax; a.domax = domax; a.dmax = dmax;
  a.lse = lse; a.delta = delta; a.alpha = alpha; a.B = B;
  a.out = dq;
  if (C == 256) {
    if (g_mulan_tune[17] == 1) {      // dev A/B: the output-split variant (scores recomputed per output half)
      int e = launch<1, 256, 2>(a, stream);
      if (e) return e;
      a.out = dv;
      e = launch<3, 256, 1>(a, stream);
      if (e) return e;
      a.out = dk;
      return launch<4, 256, 2>(a, stream);
    }
    int e = launch<1, 256, 1, 2>(a, stream);
    if (e) return e;
    a.out = dv;
    // dv needs T1 only: one wave holds all of Y (128 registers) + the 256 output channels (128) without a partner
    e = g_mulan_tune[17] == 2 ? launch<3, 256, 1, 2>(a, stream) : launch<3, 256, 1, 1>(a, stream);
    if (e) return e;
    a.out = dk;
    return launch<4, 256, 1, 2>(a, stream);
  }
  int e = launch<1>(a, stream);
  if (e) return e;
  a.out = dv; a.out2 = dk;
  return launch<2>(a, stream);
}
