// Fused GroupNorm (+ SiLU) (+ dropout) forward / backward on NHWC fp32, HW = 1024.
// Replaces flax nn.GroupNorm() [32 groups, eps 1e-6, E[x^2]-E[x]^2 variance] followed by nn.swish and
// nn.Dropout in the reference ResnetBlock (ldm/model_vdm.py:622-623,632,643-644), the final
// normalisation (model_vdm.py:376-377) and the activation-free one of AttnBlock (model_vdm.py:672-674).
//
// The input may be a virtual channel concat [x1 | x2] (up blocks: model_vdm.py:369); the output is
// written as one tensor with C1+C2 channels, so the concat is never materialised on its own.
// One 256-thread block owns (sample, 32-channel slab): 1024 px x 128 B, kept in registers.
#include "common.h"
#include "f16x3_common.h"

namespace {

constexpr int HW = 1024;
constexpr int NP = HW / 32;   // pixels per thread

struct GnArgs {
  const float* x1; const float* x2; int C1, C2;
  const float* gamma; const float* beta;     // [C1+C2]
  float* y;                                   // [B,HW,C1+C2]
  float* mean; float* rstd;                   // [B,G]
  int B, G; float eps; int act;               // act: 0 none, 1 silu
  float keep; unsigned long long seed, offset;  // dropout: keep == 1 -> off
  unsigned* ymax;                             // optional [B][16]: partial maxima of |y| (mulan_absmax_rows format)
  const unsigned long long* seed_dev;         // optional: dropout seed = seed ^ seed_dev[0] (stream-ordered: graph replay)
  unsigned char* yplanes;                     // optional: y as the split fp16 planes of the f16x3 kernels instead of fp32
                                              // ([B][Ct/16][HW][plane][16], scaled by the bound below); y is not written
  unsigned* maskbits;                         // optional output (dropout on): the keep-bits as drawn, for the backward
                                              // kernel: [B][Ct/32][256 threads][4 words] (thread t = prow * 8 + quad of
                                              // this kernel; bit 4 i + e: element e of pixel prow + 32 i)
  int stats_only;                             // 1: mean / rstd and the bound (into ymax) only, nothing is normalised: the
                                              // consumer (mulan_conv3x3_fwd_f16x3_gn_in) applies them while it fills its patches
  // streaming kernel only: the statistics arrive as the partial sums the convolutions that PRODUCED x1 (, x2) left
  // (their `ystats`: [B][HW / 256][C1 / 4][2] = sum and sum of squares per image, 8-row tile and channel quad);
  // mean / rstd are then OUTPUTS (for the backward pass).  NULL: mean / rstd are inputs.
  const float* xstats1; const float* xstats2;
  int xstats_tiles;                           // row tiles per image in xstats1 / xstats2 (the producer's tile height: 4, 8 or 16)
};

__device__ __forceinline__ void drop4(f32x4& v, float keep, unsigned long long seed, unsigned long long ctr) {
  // keep-mask = (u32 < floor(keep * 2^32)); jax.random.bernoulli(keep) semantics, scaled by 1/keep
  // (flax nn.Dropout).  Integer compare so the numpy oracle reproduces the mask bit for bit.
  const Philox4 r = philox4x32_10(seed, ctr, 0ull);
  const uint32_t thr = (uint32_t)((double)keep * 4294967296.0);
  const float inv = 1.f / keep;
  v[0] = (r.x < thr) ? v[0] * inv : 0.f;
  v[1] = (r.y < thr) ? v[1] * inv : 0.f;
  v[2] = (r.z < thr) ? v[2] * inv : 0.f;
  v[3] = (r.w < thr) ? v[3] * inv : 0.f;
}

// 4 keep-bits (bit e = element e kept) of one float4: same Philox draw and integer compare as drop4
__device__ __forceinline__ unsigned drop_bits4(uint32_t thr, unsigned long long seed, unsigned long long ctr) {
  const Philox4 r = philox4x32_10(seed, ctr, 0ull);
  return (r.x < thr ? 1u : 0u) | (r.y < thr ? 2u : 0u) | (r.z < thr ? 4u : 0u) | (r.w < thr ? 8u : 0u);
}
// v = kept ? v * inv : 0 from precomputed keep-bits (bit position `pos` is a compile-time constant after unrolling)
__device__ __forceinline__ float apply_bit(float v, float inv, unsigned bits, int pos) {
  const int m = (int)(bits << (31 - pos)) >> 31;                 // all ones / zero
  return __uint_as_float(__float_as_uint(v * inv) & (unsigned)m);
}
// sigmoid / SiLU / SiLU' on the hardware exp2 and reciprocal (1 ulp each): the accurate expf + IEEE division cost
// ~25 VALU slots per element, which made these HBM-shaped kernels VALU-bound (19 us of memory time + 26 us of
// arithmetic per launch, measured); relative error ~3e-7
__device__ __forceinline__ float sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float silu_fast(float x) { return x * sigmoid_fast(x); }
__device__ __forceinline__ float silu_grad_fast(float x) {
  const float sg = sigmoid_fast(x);
  return sg * (1.f + x * (1.f - sg));
}

// two-wide arithmetic: clang lowers <2 x float> mul / add / fma to v_pk_*_f32 (one issue slot for two elements)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 lo2(const f32x4& v) { return f32x2{v[0], v[1]}; }
__device__ __forceinline__ f32x2 hi2(const f32x4& v) { return f32x2{v[2], v[3]}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 sigmoid_fast2(f32x2 x) {
  const f32x2 t = x * -1.4426950408889634f;
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.f;
  return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

// Sums `a`,`b` over all threads of the block that share quad-group id gq = quad / qpg.
// red: 2 * 4 * 8 floats.  Returns the group totals for this thread's group.
__device__ __forceinline__ void group_reduce2(float& a, float& b, int quad, int qpg, float* red) {
  // lanes: tid = prow*8 + quad; reduce over prow bits inside the wave (lane bits 3..5)
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane < 8) { red[wave * 8 + lane] = a; red[32 + wave * 8 + lane] = b; }
  __syncthreads();
  const int g0 = (quad / qpg) * qpg;
  float sa = 0.f, sb = 0.f;
  for (int q = g0; q < g0 + qpg; ++q)
#pragma unroll
    for (int w = 0; w < 4; ++w) { sa += red[w * 8 + q]; sb += red[32 + w * 8 + q]; }
  a = sa; b = sb;
}

// NT: the slab is loaded with the non-temporal policy.  The input is read once here (its next reader is the backward
// pass), and tools/mall_probe.hip / profiles/r04_mall_probe.log show a streaming consumer reads faster that way both
// when its input still sits in the Infinity Cache behind the producer's plain stores (27.2 vs 30.8 us for 67 MB) and
// when it comes from HBM (36.6 vs 46.8 us).  tune[14] = 1: dev A/B, plain loads.
template <bool NT>
__global__ __launch_bounds__(256) void gn_fwd_kernel(GnArgs p) {
  if (p.seed_dev) p.seed ^= p.seed_dev[0];
  __shared__ float red[64];
  const int tid = threadIdx.x, quad = tid & 7, prow = tid >> 3;
  const int b = blockIdx.x, Ct = p.C1 + p.C2;
  const int c0 = blockIdx.y * 32;
  const int cpg = Ct / p.G, qpg = cpg >> 2;       // channels / float4-quads per group
  const float* src; int ld, cs;
  if (c0 < p.C1) { src = p.x1; ld = p.C1; cs = c0; } else { src = p.x2; ld = p.C2; cs = c0 - p.C1; }
  src += (size_t)b * HW * ld + cs + quad * 4;

  f32x4 v[NP];
  float s1 = 0.f, s2 = 0.f;
#pragma clang loop unroll(full)
  for (int i = 0; i < NP; ++i) {
    const f32x4* a = reinterpret_cast<const f32x4*>(src + (size_t)(prow + 32 * i) * ld);
    v[i] = NT ? __builtin_nontemporal_load(a) : *a;
  }
  // Planes mode: the operand scale must be known before the first element is written, so it comes from a bound instead
  // of the maximum: |xhat| < sqrt(n - 1) for the n = HW * cpg elements of a group, |silu(z)| <= |z|, dropout scales by
  // 1 / keep, hence |y| <= (sqrt(n) max|gamma| + max|beta|) / keep for every image of the launch.  Every block takes
  // the two maxima over all Ct channels itself (same value in all blocks; the loads hide behind the slab's).
  float bound = 0.f;
  if (p.yplanes || p.stats_only) {
    float gm = 0.f, bm = 0.f;
    for (int cc = tid; cc < Ct; cc += 256) { gm = fmaxf(gm, fabsf(p.gamma[cc])); bm = fmaxf(bm, fabsf(p.beta[cc])); }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { gm = fmaxf(gm, __shfl_xor(gm, o, 64)); bm = fmaxf(bm, __shfl_xor(bm, o, 64)); }
    __shared__ float bred[8];
    if ((tid & 63) == 0) { bred[tid >> 6] = gm; bred[4 + (tid >> 6)] = bm; }
    __syncthreads();
    gm = fmaxf(fmaxf(bred[0], bred[1]), fmaxf(bred[2], bred[3]));
    bm = fmaxf(fmaxf(bred[4], bred[5]), fmaxf(bred[6], bred[7]));
    bound = (sqrtf((float)(HW * cpg)) * gm + bm) / p.keep;
  }
  // the dropout mask does not depend on the data: draw it (10 Philox rounds per float4, quarter-rate integer
  // multiplies) while the slab is still in flight, 4 bits per float4
  unsigned mb[NP / 8];
  const bool dropping = p.keep < 1.f;
  if (dropping) {
    const uint32_t thr = (uint32_t)((double)p.keep * 4294967296.0);
#pragma clang loop unroll(full)
    for (int i = 0; i < NP; ++i) {
      const unsigned long long idx4 = (((unsigned long long)b * HW + (prow + 32 * i)) * Ct + (c0 + quad * 4)) >> 2;
      const unsigned bits = drop_bits4(thr, p.seed, p.offset + idx4) << ((i & 7) * 4);
      mb[i >> 3] = (i & 7) ? (mb[i >> 3] | bits) : bits;
    }
    if (p.maskbits)
      *reinterpret_cast<uint4*>(p.maskbits + (((size_t)b * gridDim.y + blockIdx.y) * 256 + tid) * 4) =
          make_uint4(mb[0], mb[1], mb[2], mb[3]);
  }
#pragma clang loop unroll(full)
  for (int i = 0; i < NP; ++i) {
    s1 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    s2 += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
  }
  group_reduce2(s1, s2, quad, qpg, red);
  const float inv_n = 1.f / (float)(HW * cpg);
  const float mean = s1 * inv_n;
  const float var = fmaxf(0.f, s2 * inv_n - mean * mean);
  const float rstd = rsqrtf(var + p.eps);
  const int c = c0 + quad * 4;
  const int g = c / cpg;
  if (prow == 0 && (quad % qpg) == 0) { p.mean[b * p.G + g] = mean; p.rstd[b * p.G + g] = rstd; }
  const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
  const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
  float* dst = p.y + (size_t)b * HW * Ct + c;
  unsigned amax = 0;
  const float inv_keep = 1.f / p.keep;
  float psc, pinv;
  f16x3::scale_of(__float_as_uint(bound), psc, pinv);
  // plane record of pixel px, channels c .. c + 3: [b][c / 16][px][plane][c % 16]
  unsigned char* pdst = p.yplanes ? p.yplanes + ((size_t)(b * (Ct >> 4) + (c >> 4)) * HW) * 64 + (c & 15) * 2 : nullptr;
  const f32x2 sc_lo = lo2(ga) * rstd, sc_hi = hi2(ga) * rstd;
#pragma clang loop unroll(full)
  for (int i = 0; i < (p.stats_only ? 0 : NP); ++i) {
    const int px = prow + 32 * i;
    f32x4 o;
    {
      f32x2 u0 = fma2(lo2(v[i]) - mean, sc_lo, lo2(be)), u1 = fma2(hi2(v[i]) - mean, sc_hi, hi2(be));
      if (p.act) {
        u0 = u0 * sigmoid_fast2(u0);
        u1 = u1 * sigmoid_fast2(u1);
      }
      o = f32x4{u0[0], u0[1], u1[0], u1[1]};
    }
    if (dropping) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = apply_bit(o[e], inv_keep, mb[i >> 3], (i & 7) * 4 + e);
    }
    if (pdst) {
      // (the clamp only acts where cancellation in E[x^2] - E[x]^2 left a variance far below the true one: the fp32
      // path would carry on with those values, fp16 planes must not overflow)
      // two-wide: v_pk_mul / v_cvt_pk_f16_f32 / v_pk_add, the clamp is one v_med3 per element (same values as split2)
      typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
      f16x3::f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const f32x2 vs = f32x2{__builtin_amdgcn_fmed3f(o[e], -bound, bound), __builtin_amdgcn_fmed3f(o[e + 1], -bound, bound)} * psc;
        const f16x2v h = __builtin_convertvector(vs, f16x2v);
        const f16x2v l = __builtin_convertvector(vs - __builtin_convertvector(h, f32x2), f16x2v);
        hi[e] = h[0]; hi[e + 1] = h[1]; lo[e] = l[0]; lo[e + 1] = l[1];
      }
      *reinterpret_cast<f16x3::f16x4*>(pdst + (size_t)px * 64) = hi;
      *reinterpret_cast<f16x3::f16x4*>(pdst + (size_t)px * 64 + 32) = lo;
    } else {
      *reinterpret_cast<f32x4*>(dst + (size_t)px * Ct) = o;
#pragma unroll
      for (int e = 0; e < 4; ++e) amax = max(amax, __float_as_uint(o[e]) & 0x7fffffffu);
    }
  }
  if (p.yplanes || p.stats_only) amax = __float_as_uint(bound);   // what the planes are scaled with: the consumers' "maximum"
  if (p.ymax) {   // this block's slab is partial maximum number blockIdx.y of image b (unused entries zeroed)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o, 64));
    __syncthreads();
    unsigned* ured = reinterpret_cast<unsigned*>(red);
    if ((tid & 63) == 0) ured[tid >> 6] = amax;
    __syncthreads();
    if (tid == 0) p.ymax[b * 16 + blockIdx.y] = max(max(ured[0], ured[1]), max(ured[2], ured[3]));
    if (blockIdx.y == 0 && tid >= (int)gridDim.y && tid < 16) p.ymax[b * 16 + tid] = 0u;
  }
}

struct GnBwdArgs {
  const float* dy;                             // [B,HW,C1+C2]
  const float* x1; const float* x2; int C1, C2;
  const float* gamma; const float* beta; const float* mean; const float* rstd;
  float* dx1; float* dx2;                      // [B,HW,C1], [B,HW,C2]
  float* dgamma_part; float* dbeta_part;       // [B,C1+C2] per-sample partials
  int B, G; int act; float keep; unsigned long long seed, offset;
  int accumulate;                              // dx += instead of dx =
  unsigned* dx1max; unsigned* dx2max;          // optional [B][16]: partial maxima of |dx1|, |dx2| (single-pass kernel)
  const float* add1; const float* add2;        // optional: dx1 += add1, dx2 += add2 (gradient of a skip path of x)
  float* dxsum_part;                           // optional [B, C1+C2]: per-sample channel sums of the written dx
  const unsigned long long* seed_dev;          // optional: dropout seed = seed ^ seed_dev[0] (as in the forward pass)
  // optional in-kernel final reduction over the samples (fused entry point): the block that finishes last among the B
  // blocks of a 32-channel slab sums the per-sample partials in a fixed order and writes the totals
  unsigned* tickets;                           // [Ct / 32] arrival counters, zero before the first launch (re-armed here)
  float* dgamma; float* dbeta;                 // [Ct] totals
  float* dxsum; float* dxsum2;                 // optional [C1] each: sum over samples of dxsum_part's x1 columns (the bias
                                               // gradient of the convolution in front, and of a shortcut layer sharing it)
  // optional (single-pass kernel, C2 == 0, no add1, no accumulate): dx1 is written as the split fp16 operand planes of
  // the input-gradient convolution / weight-gradient kernel in front ([B][C1/16][HW][plane][16]) INSTEAD of as fp32,
  // scaled by an a-priori bound (below) that dx1max receives; dymax: [B][16] maxima of dy (the bound needs max|dy[b]|)
  unsigned char* dx1planes; const unsigned* dymax;
  // optional (single-pass kernel): the keep-bits the forward kernel stored (GnArgs::maskbits) instead of re-drawing them
  // (10 Philox rounds per float4: ~45 % of this kernel's arithmetic in the dropout layers)
  const unsigned* maskbits;
  // optional (single-pass kernel): a second gradient that reaches x1 from outside -- dx1 = (dx1 + add1) + add1b: the
  // gradient a block output receives through its U-Net skip connection, added here instead of by a kernel of its own
  const float* add1b;
  // streaming kernel only: the partial sums of g gamma and g gamma xhat (g = dy mask / keep act'(u)) per image, 8-row
  // tile and channel quad, [B][HW / 256][(C1 + C2) / 4][2], as the input-gradient convolution that produced dy left them
  const float* gstats;
};

__global__ __launch_bounds__(256) void gn_bwd_kernel(GnBwdArgs p) {
  if (p.seed_dev) p.seed ^= p.seed_dev[0];
  __shared__ float red[64];
  __shared__ float cred[2 * 4 * 32];
  const int tid = threadIdx.x, quad = tid & 7, prow = tid >> 3;
  const int b = blockIdx.x, Ct = p.C1 + p.C2;
  const int c0 = blockIdx.y * 32;
  const int cpg = Ct / p.G, qpg = cpg >> 2;
  const float* src; float* dxp; int ld, cs;
  if (c0 < p.C1) { src = p.x1; dxp = p.dx1; ld = p.C1; cs = c0; }
  else { src = p.x2; dxp = p.dx2; ld = p.C2; cs = c0 - p.C1; }
  src += (size_t)b * HW * ld + cs + quad * 4;
  dxp += (size_t)b * HW * ld + cs + quad * 4;
  const int c = c0 + quad * 4, g = c / cpg;
  const float mean = p.mean[b * p.G + g], rstd = p.rstd[b * p.G + g];
  const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
  const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
  const float* dyp = p.dy + (size_t)b * HW * Ct + c;

  // pass 1: accumulate group sums and per-channel partials (x, dy are re-read in pass 2: L2/MALL)
  float s1 = 0.f, s2 = 0.f;
  f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int px = prow + 32 * i;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(src + (size_t)px * ld);
    f32x4 d = *reinterpret_cast<const f32x4*>(dyp + (size_t)px * Ct);
    if (p.keep < 1.f) {
      const unsigned long long idx4 = (((unsigned long long)b * HW + px) * Ct + c) >> 2;
      drop4(d, p.keep, p.seed, p.offset + idx4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xhat = (xv[e] - mean) * rstd;
      const float u = xhat * ga[e] + be[e];
      const float gu = p.act ? d[e] * silu_grad_f(u) : d[e];
      dg[e] += gu * xhat;
      db[e] += gu;
      const float dxh = gu * ga[e];
      s1 += dxh;
      s2 += dxh * xhat;
    }
  }
  group_reduce2(s1, s2, quad, qpg, red);
  // per-channel partial sums over the 32 prow lanes
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) { dg[e] += __shfl_xor(dg[e], o, 64); db[e] += __shfl_xor(db[e], o, 64); }
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { cred[wave * 32 + lane * 4 + e] = dg[e]; cred[128 + wave * 32 + lane * 4 + e] = db[e]; }
  }
  __syncthreads();
  if (tid < 32) {
    const float a = cred[tid] + cred[32 + tid] + cred[64 + tid] + cred[96 + tid];
    const float bb = cred[128 + tid] + cred[160 + tid] + cred[192 + tid] + cred[224 + tid];
    p.dgamma_part[(size_t)b * Ct + c0 + tid] = a;
    p.dbeta_part[(size_t)b * Ct + c0 + tid] = bb;
  }
  const float inv_n = 1.f / (float)(HW * cpg);
  const float m1 = s1 * inv_n, m2 = s2 * inv_n;
  // pass 2: dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat))
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int px = prow + 32 * i;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(src + (size_t)px * ld);
    f32x4 d = *reinterpret_cast<const f32x4*>(dyp + (size_t)px * Ct);
    if (p.keep < 1.f) {
      const unsigned long long idx4 = (((unsigned long long)b * HW + px) * Ct + c) >> 2;
      drop4(d, p.keep, p.seed, p.offset + idx4);
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xhat = (xv[e] - mean) * rstd;
      const float u = xhat * ga[e] + be[e];
      const float gu = p.act ? d[e] * silu_grad_f(u) : d[e];
      o[e] = rstd * (gu * ga[e] - m1 - xhat * m2);
    }
    float* dp = dxp + (size_t)px * ld;
    if (p.accumulate) {
      const f32x4 old = *reinterpret_cast<const f32x4*>(dp);
      o[0] += old[0]; o[1] += old[1]; o[2] += old[2]; o[3] += old[3];
    }
    *reinterpret_cast<f32x4*>(dp) = o;
  }
}

// Single-pass backward: 512 threads own (sample, 32-channel slab); each thread keeps its 16 pixels' xhat and
// activation-gradient float4s in registers (128 VGPRs), so x and dy are read exactly once: 2 reads + 1 write.
__global__ __launch_bounds__(512) void gn_bwd_kernel_1pass(GnBwdArgs p) {
  if (p.seed_dev) p.seed ^= p.seed_dev[0];
  constexpr int NPB = HW / 64;   // pixels per thread
  __shared__ float red[2 * 8 * 8];
  __shared__ float cred[2 * 8 * 32];
  const int tid = threadIdx.x, quad = tid & 7, prow = tid >> 3;   // prow 0..63
  const int b = blockIdx.x, Ct = p.C1 + p.C2;
  const int c0 = blockIdx.y * 32;
  const int cpg = Ct / p.G, qpg = cpg >> 2;
  const float* src; float* dxp; int ld, cs;
  if (c0 < p.C1) { src = p.x1; dxp = p.dx1; ld = p.C1; cs = c0; }
  else { src = p.x2; dxp = p.dx2; ld = p.C2; cs = c0 - p.C1; }
  src += (size_t)b * HW * ld + cs + quad * 4;
  dxp += (size_t)b * HW * ld + cs + quad * 4;
  const float* addp = c0 < p.C1 ? p.add1 : p.add2;
  if (addp) addp += (size_t)b * HW * ld + cs + quad * 4;
  const float* addq = c0 < p.C1 ? p.add1b : nullptr;
  if (addq) addq += (size_t)b * HW * ld + cs + quad * 4;
  const int c = c0 + quad * 4, g = c / cpg;
  const float mean = p.mean[b * p.G + g], rstd = p.rstd[b * p.G + g];
  const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
  const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
  const float* dyp = p.dy + (size_t)b * HW * Ct + c;

  f32x4 xh[NPB], gq[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int px = prow + 64 * i;
    xh[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)px * ld));
    gq[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dyp + (size_t)px * Ct));
  }
  // Planes mode: the operand scale must be known before the first element is written and must be the same in every
  // block of an image, so it comes from a bound instead of the maximum.  With g = dy mask / keep act'(u) gamma:
  //   dx = rstd (g - mean(g) - xhat mean(g xhat)),  |g| <= G0 = max|dy[b]| 1.1 max|gamma| / keep  (|silu'| < 1.0999),
  //   |mean(g)| <= G0,  |mean(g xhat)| <= G0 sqrt(mean(xhat^2)) <= G0,  |xhat| < sqrt(n)  (n elements per group)
  //   => |dx| <= max_g rstd[b, g] G0 (2 + sqrt(n)).
  // Every wave takes the three maxima itself (identical in all waves and blocks; the loads hide behind the slabs').
  float bound = 0.f;
  if (p.dx1planes) {
    const int ln = tid & 63;
    float rm = 0.f, gm = 0.f, dm = ln < 16 ? __uint_as_float(p.dymax[b * 16 + ln]) : 0.f;
    for (int gg = ln; gg < p.G; gg += 64) rm = fmaxf(rm, p.rstd[b * p.G + gg]);
    for (int cc = ln; cc < Ct; cc += 64) gm = fmaxf(gm, fabsf(p.gamma[cc]));
    rm = wave_max(rm); gm = wave_max(gm); dm = wave_max(dm);
    bound = ((sqrtf((float)(HW * cpg)) + 2.f) * (p.act ? 1.1f : 1.f)) * rm * gm * (dm / p.keep);
  }
  // the forward pass's dropout mask: as stored by the forward kernel, or re-drawn while the two slabs are in flight
  // (see gn_fwd_kernel).  Stored form: the forward thread (prow & 31, quad) holds pixels (prow & 31) + 32 j at bit 4 j;
  // this thread's pixel prow + 64 i is its j = (prow >> 5) + 2 i -- every second nibble, from bit 4 (prow >> 5) on.
  unsigned mb[NPB / 8];
  const bool dropping = p.keep < 1.f;
  const float inv_keep = 1.f / p.keep;
  if (dropping && p.maskbits) {
    const uint4 w = *reinterpret_cast<const uint4*>(p.maskbits + (((size_t)b * gridDim.y + blockIdx.y) * 256 + (prow & 31) * 8 + quad) * 4);
    const unsigned sh = (unsigned)(prow >> 5) * 4u;
    const unsigned ws[4] = {w.x >> sh, w.y >> sh, w.z >> sh, w.w >> sh};
    mb[0] = mb[1] = 0u;
#pragma unroll
    for (int i = 0; i < NPB; ++i)       // nibble 2 i of the shifted words -> nibble i of this thread's 16
      mb[i >> 3] |= ((ws[i >> 2] >> ((2 * i & 7) * 4)) & 15u) << ((i & 7) * 4);
  } else if (dropping) {
    const uint32_t thr = (uint32_t)((double)p.keep * 4294967296.0);
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const unsigned long long idx4 = (((unsigned long long)b * HW + (prow + 64 * i)) * Ct + c) >> 2;
      const unsigned bits = drop_bits4(thr, p.seed, p.offset + idx4) << ((i & 7) * 4);
      mb[i >> 3] = (i & 7) ? (mb[i >> 3] | bits) : bits;
    }
  }
  f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f}, dg_lo = {0.f, 0.f}, dg_hi = {0.f, 0.f}, db_lo = {0.f, 0.f}, db_hi = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    if (dropping) {
#pragma unroll
      for (int e = 0; e < 4; ++e) gq[i][e] = apply_bit(gq[i][e], inv_keep, mb[i >> 3], (i & 7) * 4 + e);
    }
    {
      f32x2 xa = (lo2(xh[i]) - mean) * rstd, xb = (hi2(xh[i]) - mean) * rstd;
      f32x2 ga_ = lo2(gq[i]), gb_ = hi2(gq[i]);
      if (p.act) {
        const f32x2 ua = fma2(xa, lo2(ga), lo2(be)), ub = fma2(xb, hi2(ga), hi2(be));
        const f32x2 sa = sigmoid_fast2(ua), sb = sigmoid_fast2(ub);
        ga_ = ga_ * (sa * fma2(ua, 1.f - sa, f32x2{1.f, 1.f}));      // g * silu'(u)
        gb_ = gb_ * (sb * fma2(ub, 1.f - sb, f32x2{1.f, 1.f}));
      }
      dg_lo = fma2(ga_, xa, dg_lo); dg_hi = fma2(gb_, xb, dg_hi);
      db_lo += ga_; db_hi += gb_;
      const f32x2 da = ga_ * lo2(ga), dbv = gb_ * hi2(ga);
      s1v += da + dbv;
      s2v = fma2(da, xa, fma2(dbv, xb, s2v));
      xh[i] = f32x4{xa[0], xa[1], xb[0], xb[1]};
      gq[i] = f32x4{da[0], da[1], dbv[0], dbv[1]};
    }
  }
  float s1 = s1v[0] + s1v[1], s2 = s2v[0] + s2v[1];
  f32x4 dg = {dg_lo[0], dg_lo[1], dg_hi[0], dg_hi[1]}, db = {db_lo[0], db_lo[1], db_hi[0], db_hi[1]};
  // reduce over the 64 prow lanes: inside the wave (lane bits 3..5), then across the 8 waves through LDS
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[e] += __shfl_xor(dg[e], o, 64); db[e] += __shfl_xor(db[e], o, 64); }
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < 8) {
    red[wave * 8 + lane] = s1;
    red[64 + wave * 8 + lane] = s2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { cred[wave * 32 + lane * 4 + e] = dg[e]; cred[256 + wave * 32 + lane * 4 + e] = db[e]; }
  }
  __syncthreads();
  if (tid < 32) {
    float a = 0.f, bb = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { a += cred[w * 32 + tid]; bb += cred[256 + w * 32 + tid]; }
    if (p.tickets) {   // handed to another block inside this launch: write-through stores (guide G16, R1)
      __hip_atomic_store(p.dgamma_part + (size_t)b * Ct + c0 + tid, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.dbeta_part + (size_t)b * Ct + c0 + tid, bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      p.dgamma_part[(size_t)b * Ct + c0 + tid] = a;
      p.dbeta_part[(size_t)b * Ct + c0 + tid] = bb;
    }
  }
  const int g0 = (quad / qpg) * qpg;
  float t1 = 0.f, t2 = 0.f;
  unsigned amax = 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  for (int q = g0; q < g0 + qpg; ++q)
#pragma unroll
    for (int w = 0; w < 8; ++w) { t1 += red[w * 8 + q]; t2 += red[64 + w * 8 + q]; }
  const float inv_n = 1.f / (float)(HW * cpg);
  const float m1 = t1 * inv_n, m2 = t2 * inv_n;
  float psc, pinv;
  f16x3::scale_of(__float_as_uint(bound), psc, pinv);
  // plane record of pixel px, channels c .. c + 3: [b][c / 16][px][plane][c % 16]
  unsigned char* pdst = p.dx1planes ? p.dx1planes + ((size_t)(b * (Ct >> 4) + (c >> 4)) * HW) * 64 + (c & 15) * 2 : nullptr;
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int px = prow + 64 * i;
    f32x4 o;
    {
      const f32x2 oa = (lo2(gq[i]) - m1 - lo2(xh[i]) * m2) * rstd, ob = (hi2(gq[i]) - m1 - hi2(xh[i]) * m2) * rstd;
      o = f32x4{oa[0], oa[1], ob[0], ob[1]};
    }
    if (pdst) {     // (the clamp never acts while the bound holds; fp16 planes must not overflow whatever the input)
      typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
      f16x3::f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const f32x2 vs = f32x2{__builtin_amdgcn_fmed3f(o[e], -bound, bound), __builtin_amdgcn_fmed3f(o[e + 1], -bound, bound)} * psc;
        const f16x2v h = __builtin_convertvector(vs, f16x2v);
        const f16x2v l = __builtin_convertvector(vs - __builtin_convertvector(h, f32x2), f16x2v);
        hi[e] = h[0]; hi[e + 1] = h[1]; lo[e] = l[0]; lo[e + 1] = l[1];
      }
      *reinterpret_cast<f16x3::f16x4*>(pdst + (size_t)px * 64) = hi;
      *reinterpret_cast<f16x3::f16x4*>(pdst + (size_t)px * 64 + 32) = lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) csum[e] += o[e];
      continue;
    }
    float* dp = dxp + (size_t)px * ld;
    if (p.accumulate) {
      const f32x4 old = *reinterpret_cast<const f32x4*>(dp);
      o[0] += old[0]; o[1] += old[1]; o[2] += old[2]; o[3] += old[3];
    }
    if (addp) {
      const f32x4 ad = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(addp + (size_t)px * ld));
      o[0] += ad[0]; o[1] += ad[1]; o[2] += ad[2]; o[3] += ad[3];
    }
    if (addq) {
      const f32x4 ad = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(addq + (size_t)px * ld));
      o[0] += ad[0]; o[1] += ad[1]; o[2] += ad[2]; o[3] += ad[3];
    }
    *reinterpret_cast<f32x4*>(dp) = o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      amax = max(amax, __float_as_uint(o[e]) & 0x7fffffffu);
      csum[e] += o[e];
    }
  }
  if (p.dxsum_part) {   // per-sample channel sums of what was written (the bias gradient of the convolution in front)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) csum[e] += __shfl_xor(csum[e], o, 64);
    __syncthreads();
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cred[wave * 32 + lane * 4 + e] = csum[e];
    }
    __syncthreads();
    if (tid < 32) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += cred[w * 32 + tid];
      if (p.tickets) __hip_atomic_store(p.dxsum_part + (size_t)b * Ct + c0 + tid, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else p.dxsum_part[(size_t)b * Ct + c0 + tid] = a;
    }
  }
  unsigned* mout = c0 < p.C1 ? p.dx1max : p.dx2max;
  if (mout) {   // this block's slab is one partial maximum of its image in its input tensor (unused entries zeroed)
    const int slab = (c0 < p.C1 ? c0 : c0 - p.C1) / 32, nslab = (c0 < p.C1 ? p.C1 : p.C2) / 32;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o, 64));
    __syncthreads();
    unsigned* ured = reinterpret_cast<unsigned*>(red);
    if ((tid & 63) == 0) ured[tid >> 6] = amax;
    __syncthreads();
    if (tid == 0) {
      unsigned m = 0;
#pragma unroll
      for (int w = 0; w < 8; ++w) m = max(m, ured[w]);
      // planes: what they were scaled with is the consumers' "maximum" (the same value from every block of the image)
      mout[b * 16 + slab] = p.dx1planes ? __float_as_uint(bound) : m;
    }
    if (slab == 0 && tid >= nslab && tid < 16) mout[b * 16 + tid] = 0u;
  }
  if (p.tickets) {
    // Final reduction over the samples, inside the launch (replaces one mulan_colsum launch per GroupNorm and one per
    // convolution bias).  Hand-off as in the guide's G16 / R1: the partials above are write-through stores; every wave
    // drains its stores, the block's barrier, ONE lane takes a ticket (agent-scope atomic add); the block whose add
    // returns B - 1 came last: its lanes load after the barrier that follows, with sc1 loads (no acquire needed for
    // this row of the table: 4-byte sc1 stores, 4-byte sc1 loads, hipMalloc memory).  The sum runs in a fixed order
    // (16 interleaved sample lanes, then the 16 partial sums in order): deterministic, whichever block comes last.
    __shared__ unsigned s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
#ifdef MULAN_GN_TICKET_FENCE
      // memory-model form of the hand-off (release on the ticket, acquire in the last block): measured as an A/B
      // against the sc1 write-through / sc1 load recipe the kernel ships with (profiles/r03_gn_ticket_fence.log)
      const unsigned t = __hip_atomic_fetch_add(p.tickets + blockIdx.y, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
      const unsigned t = __hip_atomic_fetch_add(p.tickets + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      s_last = (t == (unsigned)p.B - 1u) ? 1u : 0u;
      if (s_last) __hip_atomic_store(p.tickets + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    }
    __syncthreads();
    if (s_last) {
#ifdef MULAN_GN_TICKET_FENCE
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
      const int ch = tid & 31, sl = tid >> 5;                  // 16 sample lanes x 32 channels
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      const bool want_x = p.dxsum && c0 < p.C1;
      // this block is alone on the critical path of the launch: all loads of a batch of 8 samples per lane (128 per
      // block) are issued before the first one is consumed -- one memory round trip instead of one per sample
      for (int s0 = sl; s0 < p.B; s0 += 128) {
        float g[8], bt[8], xs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int sm = s0 + 16 * u;
          const size_t o = (size_t)(sm < p.B ? sm : s0) * Ct + c0 + ch;
          g[u] = __hip_atomic_load(p.dgamma_part + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bt[u] = __hip_atomic_load(p.dbeta_part + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          xs[u] = want_x ? __hip_atomic_load(p.dxsum_part + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (s0 + 16 * u < p.B) { a0 += g[u]; a1 += bt[u]; a2 += xs[u]; }
      }
      float* outs[3] = {p.dgamma, p.dbeta, c0 < p.C1 ? p.dxsum : nullptr};
      const float vals[3] = {a0, a1, a2};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        __syncthreads();
        cred[tid] = vals[k];
        __syncthreads();
        if (tid < 32 && outs[k]) {
          float t = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) t += cred[j * 32 + tid];
          outs[k][c0 + tid] = t;
          if (k == 2 && p.dxsum2) p.dxsum2[c0 + tid] = t;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Streaming forms (round 5): the same arithmetic without the register-resident slab.  The slab kernels above read a
// whole (sample, 32-channel slab) before the first byte is written, because the statistics (forward) / the two group
// sums (backward) are reductions over it: every CU alternates between a read phase and a write phase, and with 512
// blocks of one dispatch round the whole chip does so in step (4.0-4.3 TB/s).  Here the reductions arrive from the
// PRODUCER of the tensor -- the convolution whose epilogue wrote x (forward: sum x, sum x^2) or dy (backward: sum g gamma,
// sum g gamma xhat), per image, 8-row tile and channel quad -- so every element is final the moment it is loaded: 16 waves
// per block, 8 pixels per lane, loads and stores of different waves overlap all the time.
// Block = (sample, 32-channel slab), 1024 threads: thread (sp, prow, quad) = (tid >> 8, (tid >> 3) & 31, tid & 7) owns the
// float4 `quad` of pixels prow + 32 (8 sp + i), i = 0..7 -- the pixels of keep-bit word `sp` of slab-kernel thread
// (prow, quad): both kernel families read and write the same keep-bit layout.
constexpr int SU = 8;

// mean / rstd of group g of image b from the producers' partial sums (fixed order: quads of the group, then row tiles)
__device__ __forceinline__ void stats_from_partials(const GnArgs& p, int b, int g, int cpg, float& mean, float& rstd) {
  const int qpg = cpg >> 2, nq1 = p.C1 >> 2, nq2 = p.C2 >> 2;
  float s1 = 0.f, s2 = 0.f;
  for (int qq = g * qpg; qq < (g + 1) * qpg; ++qq) {
    const bool first = qq < nq1;
    const int nq = first ? nq1 : nq2;
    const float* st = (first ? p.xstats1 : p.xstats2) + ((size_t)b * p.xstats_tiles * nq + (first ? qq : qq - nq1)) * 2;
    for (int t = 0; t < p.xstats_tiles; ++t) { s1 += st[(size_t)t * nq * 2]; s2 += st[(size_t)t * nq * 2 + 1]; }
  }
  const float inv_n = 1.f / (float)(HW * cpg);
  mean = s1 * inv_n;
  rstd = rsqrtf(fmaxf(0.f, s2 * inv_n - mean * mean) + p.eps);
}

// Addressing: one buffer resource per tensor and image (wave-uniform, scalar registers), one 32-bit lane offset per
// tensor, the pixel step as the instruction's scalar offset -- no 64-bit address arithmetic, no address registers per
// pixel (with flat pointers the compiler keeps all of a thread's 8-32 addresses live and spills).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t gn_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, f16x3::kBufWord3);
}
__device__ __forceinline__ f32x4 gn_ld4(rsrc_t r, unsigned voff, unsigned soff) {      // non-temporal: read once
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2));
}
// A 16-byte buffer store reads its data registers a few cycles AFTER it issues.  hipcc (ROCm 7.2) pads a following VALU
// write of those registers with wait states only for the immediate-offset form, not when the scalar offset is a register
// (LLVM's model of the hazard) -- and gfx950 then stored the NEW value of the first data register: `buffer_store_dwordx4
// v[26:29], ..., s3 offen` followed at once by `v_and_b32 v26, 0x7fffffff, v26` (the maxima update) wrote |y| for element
// 0 (seen as sign flips in tests/test_gpu_gn_stream.py, 256 NSP >= 512 threads).  The data registers are therefore kept
// alive across three wait states behind the store; 8-byte stores (the plane pieces) have no such hazard.
__device__ __forceinline__ void gn_st4(rsrc_t r, f32x4 v, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(f16x3::i32x4, v), r, voff, soff, 0);
  asm volatile("s_nop 2" ::: "memory");
  asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
}
__device__ __forceinline__ void gn_st_planes(rsrc_t r, f32x4 o, float bound, float psc, unsigned voff, unsigned soff) {
  // the split of gn_fwd_kernel's planes mode (clamp, scale, hi = fp16(v), lo = fp16(v - hi)), two 8-byte pieces
  typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
  f16x3::f16x4 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    const f32x2 vs = f32x2{__builtin_amdgcn_fmed3f(o[e], -bound, bound), __builtin_amdgcn_fmed3f(o[e + 1], -bound, bound)} * psc;
    const f16x2v h = __builtin_convertvector(vs, f16x2v);
    const f16x2v l = __builtin_convertvector(vs - __builtin_convertvector(h, f32x2), f16x2v);
    hi[e] = h[0]; hi[e + 1] = h[1]; lo[e] = l[0]; lo[e + 1] = l[1];
  }
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(f16x3::i32x2, hi), r, voff, soff, 0);
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(f16x3::i32x2, lo), r, voff + 32, soff, 0);
}

template <int NSP>     // NSP quarters (256 pixels each) of the slab per block: 256 NSP threads, grid.z = 4 / NSP
__global__ __launch_bounds__(256 * NSP) void gn_fwd_stream_kernel(GnArgs p) {
  if (p.seed_dev) p.seed ^= p.seed_dev[0];
  constexpr int NW = 4 * NSP;
  __shared__ unsigned ured[NW];
  const int tid = threadIdx.x, quad = tid & 7, prow = (tid >> 3) & 31, sp = (tid >> 8) + blockIdx.z * NSP;
  const int b = blockIdx.x, Ct = p.C1 + p.C2;
  const int c0 = blockIdx.y * 32;
  const int cpg = Ct / p.G, qpg = cpg >> 2, csh = 31 - __builtin_clz(cpg);   // (cpg is 4, 8, 16 or 32)
  const bool first = c0 < p.C1;
  const int ld = first ? p.C1 : p.C2, cs = first ? c0 : c0 - p.C1;
  const rsrc_t xr = gn_rsrc((first ? p.x1 : p.x2) + (size_t)b * HW * ld, (size_t)HW * ld * 4);
  const int px0 = prow + 256 * sp;            // pixel of i = 0; the others follow 32 apart
  const unsigned xoff = (unsigned)((px0 * ld + cs + quad * 4) * 4);
  const unsigned xstep = (unsigned)(32 * ld * 4);

  f32x4 v[SU];
#pragma unroll
  for (int i = 0; i < SU; ++i) v[i] = gn_ld4(xr, xoff, i * xstep);
  const int c = c0 + quad * 4, g = c >> csh;
  float mean, rstd;
  if (p.xstats1) {
    stats_from_partials(p, b, g, cpg, mean, rstd);
    if (sp == 0 && prow == 0 && (quad & (qpg - 1)) == 0) { p.mean[b * p.G + g] = mean; p.rstd[b * p.G + g] = rstd; }
  } else {
    mean = p.mean[b * p.G + g]; rstd = p.rstd[b * p.G + g];
  }
  // the bound of the planes (see gn_fwd_kernel): every wave forms it by itself (no barrier between the loads and the
  // stores of a wave: a __syncthreads here would also wait for the loads in flight)
  float bound = 0.f;
  if (p.yplanes) {
    float gm = 0.f, bm = 0.f;
    for (int cc = tid & 63; cc < Ct; cc += 64) { gm = fmaxf(gm, fabsf(p.gamma[cc])); bm = fmaxf(bm, fabsf(p.beta[cc])); }
    gm = wave_max(gm); bm = wave_max(bm);
    bound = (sqrtf((float)(HW * cpg)) * gm + bm) / p.keep;
  }
  unsigned mb = 0u;
  const bool dropping = p.keep < 1.f;
  if (dropping) {
    const uint32_t thr = (uint32_t)((double)p.keep * 4294967296.0);
#pragma unroll
    for (int i = 0; i < SU; ++i) {
      const unsigned long long idx4 = (((unsigned long long)b * HW + (px0 + 32 * i)) * Ct + (c0 + quad * 4)) >> 2;
      mb |= drop_bits4(thr, p.seed, p.offset + idx4) << (i * 4);
    }
    if (p.maskbits) p.maskbits[(((size_t)b * gridDim.y + blockIdx.y) * 256 + (tid & 255)) * 4 + sp] = mb;
  }
  const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
  const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
  unsigned amax = 0;
  const float inv_keep = 1.f / p.keep;
  float psc, pinv;
  f16x3::scale_of(__float_as_uint(bound), psc, pinv);
  // output: fp32 [HW][Ct] of image b, or the plane records [Ct / 16][HW][64 B] of image b
  const rsrc_t yr = p.yplanes ? gn_rsrc(p.yplanes + (size_t)b * HW * Ct * 4, (size_t)HW * Ct * 4)
                              : gn_rsrc(p.y + (size_t)b * HW * Ct, (size_t)HW * Ct * 4);
  const unsigned yoff = p.yplanes ? (unsigned)(((c >> 4) * HW + px0) * 64 + (c & 15) * 2) : (unsigned)((px0 * Ct + c) * 4);
  const unsigned ystep = p.yplanes ? 32u * 64u : (unsigned)(32 * Ct * 4);
  const f32x2 sc_lo = lo2(ga) * rstd, sc_hi = hi2(ga) * rstd;
#pragma unroll
  for (int i = 0; i < SU; ++i) {
    f32x4 o;
    {
      f32x2 u0 = fma2(lo2(v[i]) - mean, sc_lo, lo2(be)), u1 = fma2(hi2(v[i]) - mean, sc_hi, hi2(be));
      if (p.act) {
        u0 = u0 * sigmoid_fast2(u0);
        u1 = u1 * sigmoid_fast2(u1);
      }
      o = f32x4{u0[0], u0[1], u1[0], u1[1]};
    }
    if (dropping) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = apply_bit(o[e], inv_keep, mb, i * 4 + e);
    }
    if (p.yplanes) {
      gn_st_planes(yr, o, bound, psc, yoff, i * ystep);
    } else {
      gn_st4(yr, o, yoff, i * ystep);
#pragma unroll
      for (int e = 0; e < 4; ++e) amax = max(amax, __float_as_uint(o[e]) & 0x7fffffffu);
    }
  }
  if (p.yplanes) amax = __float_as_uint(bound);
  if (p.ymax) {     // partial maximum number (slab, z block) of image b; unused entries zeroed
    const int part = blockIdx.y * gridDim.z + blockIdx.z, nparts = gridDim.y * gridDim.z;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o, 64));
    if ((tid & 63) == 0) ured[tid >> 6] = amax;
    __syncthreads();
    if (tid == 0) {
      unsigned m = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) m = max(m, ured[w]);
      p.ymax[b * 16 + part] = m;
    }
    if (part == 0 && tid >= nparts && tid < 16) p.ymax[b * 16 + tid] = 0u;
  }
}

// UB: pixels per load batch (8: the whole keep-bit word at once; 4, 2: in halves / quarters, each batch fenced from the
// next so that the register count follows the batch -- UB = 2 fits 64 registers: eight waves per SIMD, or one wave beside
// a matrix-core block that owns the CU's LDS and seven eighths of its registers)
template <int NSP, int UB>
__global__ __launch_bounds__(256 * NSP) __attribute__((amdgpu_waves_per_eu(UB == 2 ? 8 : 4, 8)))
void gn_bwd_stream_kernel(GnBwdArgs p) {
  if (p.seed_dev) p.seed ^= p.seed_dev[0];
  constexpr int NT = 256 * NSP, NW = 4 * NSP, NZ = 4 / NSP;
  __shared__ float cred[3 * 16 * 32 > NT ? 3 * 16 * 32 : NT];
  __shared__ unsigned ured[NW];
  const int tid = threadIdx.x, quad = tid & 7, prow = (tid >> 3) & 31, sp = (tid >> 8) + blockIdx.z * NSP;
  const int b = blockIdx.x, Ct = p.C1 + p.C2;
  const int c0 = blockIdx.y * 32;
  const int cpg = Ct / p.G, qpg = cpg >> 2, csh = 31 - __builtin_clz(cpg);   // (cpg is 4, 8, 16 or 32)
  const bool first = c0 < p.C1;
  const int ld = first ? p.C1 : p.C2, cs = first ? c0 : c0 - p.C1;
  const size_t xbytes = (size_t)HW * ld * 4;
  const rsrc_t xr = gn_rsrc((first ? p.x1 : p.x2) + (size_t)b * HW * ld, xbytes);
  const rsrc_t dyr = gn_rsrc(p.dy + (size_t)b * HW * Ct, (size_t)HW * Ct * 4);
  const float* addp = first ? p.add1 : p.add2;
  const float* addq = first ? p.add1b : nullptr;
  const rsrc_t a1r = gn_rsrc(addp ? addp + (size_t)b * HW * ld : nullptr, addp ? xbytes : 0);
  const rsrc_t a2r = gn_rsrc(addq ? addq + (size_t)b * HW * ld : nullptr, addq ? xbytes : 0);
  const int c = c0 + quad * 4, g = c >> csh;
  const int px0 = prow + 256 * sp;
  const unsigned xoff = (unsigned)((px0 * ld + cs + quad * 4) * 4), xstep = (unsigned)(32 * ld * 4);
  const unsigned dyoff = (unsigned)((px0 * Ct + c) * 4), dystep = (unsigned)(32 * Ct * 4);

  // (UB = 8) the loads go out before anything else: the statistics / bound / keep-bit loads below overlap them.  The
  // batched forms load inside the loop: nothing is live across the prologue
  f32x4 xh[UB], gq[UB];
  if (UB == SU) {
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      xh[i] = gn_ld4(xr, xoff, i * xstep);
      gq[i] = gn_ld4(dyr, dyoff, i * dystep);
    }
  }
  const float mean = p.mean[b * p.G + g], rstd = p.rstd[b * p.G + g];
  const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
  const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
  // the two group sums, from the partial sums the producer of dy left (fixed order: quads of the group, then row tiles)
  float m1, m2;
  {
    const int nq = Ct >> 2;
    float t1 = 0.f, t2 = 0.f;
    for (int qq = g * qpg; qq < (g + 1) * qpg; ++qq) {
      const float* st = p.gstats + ((size_t)b * (HW / 256) * nq + qq) * 2;
#pragma unroll
      for (int t = 0; t < HW / 256; ++t) { t1 += st[(size_t)t * nq * 2]; t2 += st[(size_t)t * nq * 2 + 1]; }
    }
    const float inv_n = 1.f / (float)(HW * cpg);
    m1 = t1 * inv_n; m2 = t2 * inv_n;
  }
  float bound = 0.f;     // planes mode: the a-priori bound of |dx[b]| (see gn_bwd_kernel_1pass)
  if (p.dx1planes) {
    const int ln = tid & 63;
    float rm = 0.f, gm = 0.f, dm = ln < 16 ? __uint_as_float(p.dymax[b * 16 + ln]) : 0.f;
    for (int gg = ln; gg < p.G; gg += 64) rm = fmaxf(rm, p.rstd[b * p.G + gg]);
    for (int cc = ln; cc < Ct; cc += 64) gm = fmaxf(gm, fabsf(p.gamma[cc]));
    rm = wave_max(rm); gm = wave_max(gm); dm = wave_max(dm);
    bound = ((sqrtf((float)(HW * cpg)) + 2.f) * (p.act ? 1.1f : 1.f)) * rm * gm * (dm / p.keep);
  }
  unsigned mb = 0u;
  const bool dropping = p.keep < 1.f;
  const float inv_keep = 1.f / p.keep;
  if (dropping && p.maskbits) {
    mb = p.maskbits[(((size_t)b * gridDim.y + blockIdx.y) * 256 + (tid & 255)) * 4 + sp];
  } else if (dropping) {
    const uint32_t thr = (uint32_t)((double)p.keep * 4294967296.0);
#pragma unroll
    for (int i = 0; i < SU; ++i) {
      const unsigned long long idx4 = (((unsigned long long)b * HW + (px0 + 32 * i)) * Ct + c) >> 2;
      mb |= drop_bits4(thr, p.seed, p.offset + idx4) << (i * 4);
    }
  }
  float psc, pinv;
  f16x3::scale_of(__float_as_uint(bound), psc, pinv);
  // output: fp32 [HW][ld] of image b (dx1 or dx2), or the plane records [C1 / 16][HW][64 B] of image b
  const rsrc_t or_ = p.dx1planes ? gn_rsrc(p.dx1planes + (size_t)b * HW * Ct * 4, (size_t)HW * Ct * 4)
                                 : gn_rsrc((first ? p.dx1 : p.dx2) + (size_t)b * HW * ld, xbytes);
  const unsigned ooff = p.dx1planes ? (unsigned)(((c >> 4) * HW + px0) * 64 + (c & 15) * 2) : xoff;
  const unsigned ostep = p.dx1planes ? 32u * 64u : xstep;
  f32x2 dg_lo = {0.f, 0.f}, dg_hi = {0.f, 0.f}, db_lo = {0.f, 0.f}, db_hi = {0.f, 0.f};
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  unsigned amax = 0;
  // (rolled over the batches: one batch's registers; the scalar offsets advance by a batch per iteration)
#pragma unroll 1
  for (int h = 0; h < SU / UB; ++h) {
    const unsigned hx = (unsigned)(h * UB) * xstep, hd = (unsigned)(h * UB) * dystep, ho = (unsigned)(h * UB) * ostep;
    if (UB < SU) {
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        xh[i] = gn_ld4(xr, xoff, hx + i * xstep);
        gq[i] = gn_ld4(dyr, dyoff, hd + i * dystep);
      }
    }
    f32x4 ad1[UB], ad2[UB];
    if (addp) {
#pragma unroll
      for (int i = 0; i < UB; ++i) ad1[i] = gn_ld4(a1r, xoff, hx + i * xstep);
    }
    if (addq) {
#pragma unroll
      for (int i = 0; i < UB; ++i) ad2[i] = gn_ld4(a2r, xoff, hx + i * xstep);
    }
    const unsigned mbh = mb >> (h * UB * 4);
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      if (dropping) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gq[i][e] = apply_bit(gq[i][e], inv_keep, mbh, i * 4 + e);
      }
      f32x4 o;
      {
        const f32x2 xa = (lo2(xh[i]) - mean) * rstd, xb = (hi2(xh[i]) - mean) * rstd;
        f32x2 ga_ = lo2(gq[i]), gb_ = hi2(gq[i]);
        if (p.act) {
          const f32x2 ua = fma2(xa, lo2(ga), lo2(be)), ub = fma2(xb, hi2(ga), hi2(be));
          const f32x2 sa = sigmoid_fast2(ua), sb = sigmoid_fast2(ub);
          ga_ = ga_ * (sa * fma2(ua, 1.f - sa, f32x2{1.f, 1.f}));      // g * silu'(u)
          gb_ = gb_ * (sb * fma2(ub, 1.f - sb, f32x2{1.f, 1.f}));
        }
        dg_lo = fma2(ga_, xa, dg_lo); dg_hi = fma2(gb_, xb, dg_hi);
        db_lo += ga_; db_hi += gb_;
        const f32x2 da = ga_ * lo2(ga), dbv = gb_ * hi2(ga);
        const f32x2 oa = (da - m1 - xa * m2) * rstd, ob = (dbv - m1 - xb * m2) * rstd;
        o = f32x4{oa[0], oa[1], ob[0], ob[1]};
      }
      if (p.dx1planes) {
        gn_st_planes(or_, o, bound, psc, ooff, ho + i * ostep);
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[e] += o[e];
        continue;
      }
      if (addp) { o[0] += ad1[i][0]; o[1] += ad1[i][1]; o[2] += ad1[i][2]; o[3] += ad1[i][3]; }
      if (addq) { o[0] += ad2[i][0]; o[1] += ad2[i][1]; o[2] += ad2[i][2]; o[3] += ad2[i][3]; }
      gn_st4(or_, o, ooff, ho + i * ostep);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        amax = max(amax, __float_as_uint(o[e]) & 0x7fffffffu);
        csum[e] += o[e];
      }
    }
  }
  // per-channel partial sums of this block: over the 8 pixel rows of a wave, then over the waves
  f32x4 dg = {dg_lo[0], dg_lo[1], dg_hi[0], dg_hi[1]}, db = {db_lo[0], db_lo[1], db_hi[0], db_hi[1]};
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dg[e] += __shfl_xor(dg[e], o, 64); db[e] += __shfl_xor(db[e], o, 64); csum[e] += __shfl_xor(csum[e], o, 64);
    }
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cred[wave * 32 + lane * 4 + e] = dg[e];
      cred[512 + wave * 32 + lane * 4 + e] = db[e];
      cred[1024 + wave * 32 + lane * 4 + e] = csum[e];
    }
  }
  __syncthreads();
  const size_t prow_out = (size_t)b * NZ + blockIdx.z;      // this block's row of the partial arrays [B * NZ][Ct]
  if (tid < 96) {
    const int k = tid >> 5, ch = tid & 31;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) a += cred[k * 512 + w * 32 + ch];
    float* dstp = k == 0 ? p.dgamma_part : (k == 1 ? p.dbeta_part : p.dxsum_part);
    if (dstp) __hip_atomic_store(dstp + prow_out * Ct + c0 + ch, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned* mout = first ? p.dx1max : p.dx2max;
  if (mout) {
    const int slab = cs / 32, nslab = ld / 32;
    const int part = slab * NZ + blockIdx.z, nparts = nslab * NZ;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o, 64));
    if (lane == 0) ured[wave] = amax;
    __syncthreads();
    if (tid == 0) {
      unsigned m = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) m = max(m, ured[w]);
      mout[b * 16 + part] = p.dx1planes ? __float_as_uint(bound) : m;
    }
    if (part == 0 && tid >= nparts && tid < 16) mout[b * 16 + tid] = 0u;
  }
  {
    // final reduction over the samples inside the launch: the hand-off of gn_bwd_kernel_1pass (write-through partials,
    // every wave drains, barrier, one ticket per block; the block whose ticket comes last sums with sc1 loads, fixed order)
    __shared__ unsigned s_last;
    const int R = p.B * NZ;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(p.tickets + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == (unsigned)R - 1u) ? 1u : 0u;
      if (s_last) __hip_atomic_store(p.tickets + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    }
    __syncthreads();
    if (s_last) {
      constexpr int LN = NT / 32;                              // row lanes x 32 channels
      const int ch = tid & 31, sl = tid >> 5;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      const bool want_x = p.dxsum && first;
      for (int s0 = sl; s0 < R; s0 += LN * 8) {
        float gsm[8], bt[8], xs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int sm = s0 + LN * u;
          const size_t o = (size_t)(sm < R ? sm : s0) * Ct + c0 + ch;
          gsm[u] = __hip_atomic_load(p.dgamma_part + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bt[u] = __hip_atomic_load(p.dbeta_part + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          xs[u] = want_x ? __hip_atomic_load(p.dxsum_part + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (s0 + LN * u < R) { a0 += gsm[u]; a1 += bt[u]; a2 += xs[u]; }
      }
      float* outs[3] = {p.dgamma, p.dbeta, first ? p.dxsum : nullptr};
      const float vals[3] = {a0, a1, a2};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        __syncthreads();
        cred[tid] = vals[k];
        __syncthreads();
        if (tid < 32 && outs[k]) {
          float t = 0.f;
#pragma unroll
          for (int j = 0; j < LN; ++j) t += cred[j * 32 + tid];
          outs[k][c0 + tid] = t;
          if (k == 2 && p.dxsum2) p.dxsum2[c0 + tid] = t;
        }
      }
    }
  }
}

// The backward streaming kernel as a THIN kernel (round 5 probe): 256 threads (one wave per SIMD), <= 64 registers, 3 KB of
// LDS, so that its blocks fit on a CU whose other resources a matrix-core block of the weight-gradient stream owns
// (conv3x3_wgrad_f16x3_planes_kernel: 448 registers, 137 KB).  Compile-time variants instead of run-time flags (every
// run-time branch of gn_bwd_stream_kernel costs live registers), batches of 2 pixels, and NO final reduction over the
// samples: the per-block partial rows [B * 4][Ct] are summed by the caller (mulan_colsum_pair or the next launch).
template <bool PLANES, int NADD, bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void gn_bwd_thin_kernel(GnBwdArgs p) {
  if (p.seed_dev) p.seed ^= p.seed_dev[0];
  constexpr int UB = 2;
  __shared__ float cred[3 * 4 * 32];
  __shared__ unsigned ured[4];
  const int tid = threadIdx.x, quad = tid & 7, prow = tid >> 3, sp = blockIdx.z;
  const int b = blockIdx.x, Ct = p.C1 + p.C2;
  const int c0 = blockIdx.y * 32;
  const int cpg = Ct / p.G, qpg = cpg >> 2, csh = 31 - __builtin_clz(cpg);
  const bool first = c0 < p.C1;
  const int ld = first ? p.C1 : p.C2, cs = first ? c0 : c0 - p.C1;
  const size_t xbytes = (size_t)HW * ld * 4;
  const int c = c0 + quad * 4, g = c >> csh;
  const int px0 = prow + 256 * sp;
  const float mean = p.mean[b * p.G + g], rstd = p.rstd[b * p.G + g];
  const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
  const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
  float m1, m2;
  {
    const int nq = Ct >> 2;
    float t1 = 0.f, t2 = 0.f;
    for (int qq = g * qpg; qq < (g + 1) * qpg; ++qq) {
      const float* st = p.gstats + ((size_t)b * (HW / 256) * nq + qq) * 2;
#pragma unroll
      for (int t = 0; t < HW / 256; ++t) { t1 += st[(size_t)t * nq * 2]; t2 += st[(size_t)t * nq * 2 + 1]; }
    }
    const float inv_n = 1.f / (float)(HW * cpg);
    m1 = t1 * inv_n; m2 = t2 * inv_n;
  }
  float bound = 0.f;
  if (PLANES) {
    const int ln = tid & 63;
    float rm = 0.f, gm = 0.f, dm = ln < 16 ? __uint_as_float(p.dymax[b * 16 + ln]) : 0.f;
    for (int gg = ln; gg < p.G; gg += 64) rm = fmaxf(rm, p.rstd[b * p.G + gg]);
    for (int cc = ln; cc < Ct; cc += 64) gm = fmaxf(gm, fabsf(p.gamma[cc]));
    rm = wave_max(rm); gm = wave_max(gm); dm = wave_max(dm);
    bound = ((sqrtf((float)(HW * cpg)) + 2.f) * (p.act ? 1.1f : 1.f)) * rm * gm * (dm / p.keep);
  }
  unsigned mb = 0u;
  const float inv_keep = 1.f / p.keep;
  if (DROP) {
    if (p.maskbits) {
      mb = p.maskbits[(((size_t)b * gridDim.y + blockIdx.y) * 256 + tid) * 4 + sp];
    } else {
      const uint32_t thr = (uint32_t)((double)p.keep * 4294967296.0);
#pragma unroll 1
      for (int i = 0; i < SU; ++i) {
        const unsigned long long idx4 = (((unsigned long long)b * HW + (px0 + 32 * i)) * Ct + c) >> 2;
        mb |= drop_bits4(thr, p.seed, p.offset + idx4) << (i * 4);
      }
    }
  }
  float psc, pinv;
  f16x3::scale_of(__float_as_uint(bound), psc, pinv);
  const rsrc_t xr = gn_rsrc((first ? p.x1 : p.x2) + (size_t)b * HW * ld, xbytes);
  const rsrc_t dyr = gn_rsrc(p.dy + (size_t)b * HW * Ct, (size_t)HW * Ct * 4);
  const float* addp = first ? p.add1 : p.add2;
  const float* addq = first ? p.add1b : nullptr;
  const rsrc_t a1r = gn_rsrc(NADD >= 1 && addp ? addp + (size_t)b * HW * ld : nullptr, NADD >= 1 && addp ? xbytes : 0);
  const rsrc_t a2r = gn_rsrc(NADD >= 2 && addq ? addq + (size_t)b * HW * ld : nullptr, NADD >= 2 && addq ? xbytes : 0);
  const rsrc_t or_ = PLANES ? gn_rsrc(p.dx1planes + (size_t)b * HW * Ct * 4, (size_t)HW * Ct * 4)
                            : gn_rsrc((first ? p.dx1 : p.dx2) + (size_t)b * HW * ld, xbytes);
  const unsigned xoff = (unsigned)((px0 * ld + cs + quad * 4) * 4), xstep = (unsigned)(32 * ld * 4);
  const unsigned dyoff = (unsigned)((px0 * Ct + c) * 4), dystep = (unsigned)(32 * Ct * 4);
  const unsigned ooff = PLANES ? (unsigned)(((c >> 4) * HW + px0) * 64 + (c & 15) * 2) : xoff;
  const unsigned ostep = PLANES ? 32u * 64u : xstep;
  f32x2 dg_lo = {0.f, 0.f}, dg_hi = {0.f, 0.f}, db_lo = {0.f, 0.f}, db_hi = {0.f, 0.f};
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  unsigned amax = 0;
#pragma unroll 1
  for (int h = 0; h < SU / UB; ++h) {
    const unsigned hx = (unsigned)(h * UB) * xstep, hd = (unsigned)(h * UB) * dystep, ho = (unsigned)(h * UB) * ostep;
    f32x4 xh[UB], gq[UB], ad1[UB], ad2[UB];
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      xh[i] = gn_ld4(xr, xoff, hx + i * xstep);
      gq[i] = gn_ld4(dyr, dyoff, hd + i * dystep);
      if (NADD >= 1) ad1[i] = gn_ld4(a1r, xoff, hx + i * xstep);     // (NULL tensor: zero-sized resource, reads 0)
      if (NADD >= 2) ad2[i] = gn_ld4(a2r, xoff, hx + i * xstep);
    }
    const unsigned mbh = mb >> (h * UB * 4);
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      if (DROP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gq[i][e] = apply_bit(gq[i][e], inv_keep, mbh, i * 4 + e);
      }
      f32x4 o;
      {
        const f32x2 xa = (lo2(xh[i]) - mean) * rstd, xb = (hi2(xh[i]) - mean) * rstd;
        f32x2 ga_ = lo2(gq[i]), gb_ = hi2(gq[i]);
        if (p.act) {
          const f32x2 ua = fma2(xa, lo2(ga), lo2(be)), ub = fma2(xb, hi2(ga), hi2(be));
          const f32x2 sa = sigmoid_fast2(ua), sb = sigmoid_fast2(ub);
          ga_ = ga_ * (sa * fma2(ua, 1.f - sa, f32x2{1.f, 1.f}));
          gb_ = gb_ * (sb * fma2(ub, 1.f - sb, f32x2{1.f, 1.f}));
        }
        dg_lo = fma2(ga_, xa, dg_lo); dg_hi = fma2(gb_, xb, dg_hi);
        db_lo += ga_; db_hi += gb_;
        const f32x2 da = ga_ * lo2(ga), dbv = gb_ * hi2(ga);
        const f32x2 oa = (da - m1 - xa * m2) * rstd, ob = (dbv - m1 - xb * m2) * rstd;
        o = f32x4{oa[0], oa[1], ob[0], ob[1]};
      }
      if (PLANES) {
        gn_st_planes(or_, o, bound, psc, ooff, ho + i * ostep);
      } else {
        if (NADD >= 1) { o[0] += ad1[i][0]; o[1] += ad1[i][1]; o[2] += ad1[i][2]; o[3] += ad1[i][3]; }
        if (NADD >= 2) { o[0] += ad2[i][0]; o[1] += ad2[i][1]; o[2] += ad2[i][2]; o[3] += ad2[i][3]; }
        gn_st4(or_, o, ooff, ho + i * ostep);
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = max(amax, __float_as_uint(o[e]) & 0x7fffffffu);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) csum[e] += o[e];
    }
  }
  f32x4 dg = {dg_lo[0], dg_lo[1], dg_hi[0], dg_hi[1]}, db = {db_lo[0], db_lo[1], db_hi[0], db_hi[1]};
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dg[e] += __shfl_xor(dg[e], o, 64); db[e] += __shfl_xor(db[e], o, 64); csum[e] += __shfl_xor(csum[e], o, 64);
    }
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cred[wave * 32 + lane * 4 + e] = dg[e];
      cred[128 + wave * 32 + lane * 4 + e] = db[e];
      cred[256 + wave * 32 + lane * 4 + e] = csum[e];
    }
  }
  if (!PLANES) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o, 64));
    if (lane == 0) ured[wave] = amax;
  }
  __syncthreads();
  if (tid < 96) {
    const int k = tid >> 5, ch = tid & 31;
    const float a = (cred[k * 128 + ch] + cred[k * 128 + 32 + ch]) + (cred[k * 128 + 64 + ch] + cred[k * 128 + 96 + ch]);
    float* dstp = k == 0 ? p.dgamma_part : (k == 1 ? p.dbeta_part : p.dxsum_part);
    if (dstp) dstp[((size_t)b * 4 + sp) * Ct + c0 + ch] = a;
  }
  if (b == 0 && sp == 0 && tid < 32) {     // (probe) the totals this kernel does not form: zeroed, so that a timing run stays finite
    if (p.dgamma) p.dgamma[c0 + tid] = 0.f;
    if (p.dbeta) p.dbeta[c0 + tid] = 0.f;
    if (first && p.dxsum) { p.dxsum[c0 + tid] = 0.f; if (p.dxsum2) p.dxsum2[c0 + tid] = 0.f; }
  }
  unsigned* mout = first ? p.dx1max : p.dx2max;
  if (mout) {
    const int part = (cs / 32) * 4 + sp, nparts = (ld / 32) * 4;
    if (tid == 0) mout[b * 16 + part] = PLANES ? __float_as_uint(bound) : max(max(ured[0], ured[1]), max(ured[2], ured[3]));
    if (part == 0 && tid >= nparts && tid < 16) mout[b * 16 + tid] = 0u;
  }
}

}  // namespace

// seed_dev (optional, device memory): the dropout seed is `seed ^ seed_dev[0]`, read by the kernel when it runs -- a
// stream-ordered parameter, so that a captured HIP graph replays with a fresh seed per step (seed = 0 there).
MULAN_API int mulan_groupnorm_fwd_dyn(const float* x1, const float* x2, int C1, int C2, const float* gamma,
                                      const float* beta, float* y, float* mean, float* rstd, int B, int hw, int G,
                                      float eps, int act, float keep, unsigned long long seed,
                                      unsigned long long offset, const unsigned long long* seed_dev, unsigned* ymax,
                                      hipStream_t stream) {
  const int Ct = C1 + C2;
  if (hw != HW || B <= 0 || G <= 0 || Ct % G != 0) return (int)hipErrorInvalidValue;
  const int cpg = Ct / G;
  if (cpg % 4 != 0 || 32 % cpg != 0 || C1 % 32 != 0 || C2 % 32 != 0) return (int)hipErrorInvalidValue;
  if (ymax && Ct / 32 > 16) return (int)hipErrorInvalidValue;
  GnArgs a{x1, x2, C1, C2, gamma, beta, y, mean, rstd, B, G, eps, act, keep, seed, offset, ymax, seed_dev, nullptr, nullptr};
  if (g_mulan_tune[14] == 1) hipLaunchKernelGGL(gn_fwd_kernel<false>, dim3(B, Ct / 32), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(gn_fwd_kernel<true>, dim3(B, Ct / 32), dim3(256), 0, stream, a);
  MULAN_CHECK_LAUNCH();
}

// The same with the output written as the split fp16 operand planes of the f16x3 convolution that consumes it
// (mulan_conv3x3_fwd_f16x3_planes_in, and later its weight-gradient kernel) instead of as fp32: the convolution then
// neither splits its input nor stores planes.  yplanes: mulan_conv3x3_planes_bytes(B, 32, 32, C1 + C2) bytes; ymax
// [B][16] receives the a-priori bound the planes are scaled with, in the maxima format (see gn_fwd_kernel).
MULAN_API int mulan_groupnorm_fwd_planes(const float* x1, const float* x2, int C1, int C2, const float* gamma,
                                         const float* beta, void* yplanes, float* mean, float* rstd, int B, int hw, int G,
                                         float eps, int act, float keep, unsigned long long seed,
                                         unsigned long long offset, const unsigned long long* seed_dev, unsigned* ymax,
                                         hipStream_t stream) {
  const int Ct = C1 + C2;
  if (hw != HW || B <= 0 || G <= 0 || Ct % G != 0 || !yplanes || !ymax || !(keep > 0.f)) return (int)hipErrorInvalidValue;
  const int cpg = Ct / G;
  if (cpg % 4 != 0 || 32 % cpg != 0 || C1 % 32 != 0 || C2 % 32 != 0 || Ct / 32 > 16) return (int)hipErrorInvalidValue;
  GnArgs a{x1, x2, C1, C2, gamma, beta, nullptr, mean, rstd, B, G, eps, act, keep, seed, offset, ymax, seed_dev,
           static_cast<unsigned char*>(yplanes)};
  if (g_mulan_tune[14] == 1) hipLaunchKernelGGL(gn_fwd_kernel<false>, dim3(B, Ct / 32), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(gn_fwd_kernel<true>, dim3(B, Ct / 32), dim3(256), 0, stream, a);
  MULAN_CHECK_LAUNCH();
}

// mulan_groupnorm_fwd_planes that also stores the dropout keep-bits it drew (keepbits: B * (C1 + C2) / 32 * 1024 unsigned,
// see GnArgs::maskbits), for mulan_groupnorm_bwd_fused_planes: the backward kernel then skips its 10 Philox rounds per
// float4.  keep must be < 1.
MULAN_API int mulan_groupnorm_fwd_planes_keepbits(const float* x1, const float* x2, int C1, int C2, const float* gamma,
                                                  const float* beta, void* yplanes, float* mean, float* rstd, int B, int hw,
                                                  int G, float eps, int act, float keep, unsigned long long seed,
                                                  unsigned long long offset, const unsigned long long* seed_dev,
                                                  unsigned* ymax, unsigned* keepbits, hipStream_t stream) {
  const int Ct = C1 + C2;
  if (hw != HW || B <= 0 || G <= 0 || Ct % G != 0 || !yplanes || !ymax || !keepbits || !(keep > 0.f) || !(keep < 1.f))
    return (int)hipErrorInvalidValue;
  const int cpg = Ct / G;
  if (cpg % 4 != 0 || 32 % cpg != 0 || C1 % 32 != 0 || C2 % 32 != 0 || Ct / 32 > 16) return (int)hipErrorInvalidValue;
  GnArgs a{x1, x2, C1, C2, gamma, beta, nullptr, mean, rstd, B, G, eps, act, keep, seed, offset, ymax, seed_dev,
           static_cast<unsigned char*>(yplanes), keepbits};
  if (g_mulan_tune[14] == 1) hipLaunchKernelGGL(gn_fwd_kernel<false>, dim3(B, Ct / 32), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(gn_fwd_kernel<true>, dim3(B, Ct / 32), dim3(256), 0, stream, a);
  MULAN_CHECK_LAUNCH();
}

// Statistics pass of a GroupNorm whose normalisation happens inside its consumer (mulan_conv3x3_fwd_f16x3_gn_in: the
// convolution normalises, activates and splits the fp32 tensor while it fills its LDS patches, so the normalised tensor
// never reaches HBM -- forward-only paths: evaluators, sampler).  mean / rstd [B, G] and bound [B][16] (the a-priori bound
// of |y| in the maxima format) are exactly what mulan_groupnorm_fwd_planes computes (same summation order); x is read
// once, nothing else is written.  No dropout (keep = 1).
MULAN_API int mulan_groupnorm_stats(const float* x1, const float* x2, int C1, int C2, const float* gamma, const float* beta,
                                    float* mean, float* rstd, unsigned* bound, int B, int hw, int G, float eps,
                                    hipStream_t stream) {
  const int Ct = C1 + C2;
  if (hw != HW || B <= 0 || G <= 0 || Ct % G != 0 || !bound || !mean || !rstd) return (int)hipErrorInvalidValue;
  const int cpg = Ct / G;
  if (cpg % 4 != 0 || 32 % cpg != 0 || C1 % 32 != 0 || C2 % 32 != 0 || Ct / 32 > 16) return (int)hipErrorInvalidValue;
  GnArgs a{x1, x2, C1, C2, gamma, beta, nullptr, mean, rstd, B, G, eps, 0, 1.f, 0ull, 0ull, bound, nullptr, nullptr, nullptr, 1};
  if (g_mulan_tune[14] == 1) hipLaunchKernelGGL(gn_fwd_kernel<false>, dim3(B, Ct / 32), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(gn_fwd_kernel<true>, dim3(B, Ct / 32), dim3(256), 0, stream, a);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_groupnorm_fwd(const float* x1, const float* x2, int C1, int C2, const float* gamma,
                                  const float* beta, float* y, float* mean, float* rstd, int B, int hw, int G,
                                  float eps, int act, float keep, unsigned long long seed,
                                  unsigned long long offset, unsigned* ymax, hipStream_t stream) {
  return mulan_groupnorm_fwd_dyn(x1, x2, C1, C2, gamma, beta, y, mean, rstd, B, hw, G, eps, act, keep, seed, offset,
                                 nullptr, ymax, stream);
}

MULAN_API int mulan_groupnorm_bwd_dyn(const float* dy, const float* x1, const float* x2, int C1, int C2,
                                      const float* gamma, const float* beta, const float* mean, const float* rstd,
                                      float* dx1, float* dx2, float* dgamma_part, float* dbeta_part, int B, int hw,
                                      int G, int act, float keep, unsigned long long seed, unsigned long long offset,
                                      const unsigned long long* seed_dev, int accumulate, unsigned* dx1max,
                                      unsigned* dx2max, const float* add1, const float* add2, float* dxsum_part,
                                      hipStream_t stream) {
  const int Ct = C1 + C2;
  if (hw != HW || B <= 0 || G <= 0 || Ct % G != 0) return (int)hipErrorInvalidValue;
  const int cpg = Ct / G;
  if (cpg % 4 != 0 || 32 % cpg != 0 || C1 % 32 != 0 || C2 % 32 != 0) return (int)hipErrorInvalidValue;
  if ((dx1max && C1 / 32 > 16) || (dx2max && C2 / 32 > 16)) return (int)hipErrorInvalidValue;
  GnBwdArgs a{dy, x1, x2, C1, C2, gamma, beta, mean, rstd, dx1, dx2, dgamma_part, dbeta_part,
              B, G, act, keep, seed, offset, accumulate, dx1max, dx2max, add1, add2, dxsum_part, seed_dev, nullptr, nullptr,
              nullptr, nullptr, nullptr, nullptr, nullptr};
  if (g_mulan_tune[2] == 1 && !dx1max && !dx2max && !add1 && !add2 && !dxsum_part)   // dev A/B: two-pass 256-thread variant
    hipLaunchKernelGGL(gn_bwd_kernel, dim3(B, Ct / 32), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(gn_bwd_kernel_1pass, dim3(B, Ct / 32), dim3(512), 0, stream, a);
  MULAN_CHECK_LAUNCH();
}

// mulan_groupnorm_bwd_dyn + the final reduction over the samples inside the launch: dgamma / dbeta [C1 + C2] receive the
// totals (dgamma_part / dbeta_part [B, C1 + C2] stay the scratch for the per-sample partials), dxsum (optional, [C1]) the
// sum over samples of dxsum_part's x1 columns -- the bias gradient of the convolution whose output gradient dx1 is --
// and dxsum2 (optional) a second copy of it (the bias of a shortcut layer that sees the same gradient).
// tickets: [16] unsigned, zero before the first launch on a stream (every launch leaves them zero again).
// add1b (optional, like x1): a second outside gradient of x1, dx1 = (dx1 + add1) + add1b (the skip-connection gradient of a
// U-Net block output whose other consumer is this GroupNorm).
MULAN_API int mulan_groupnorm_bwd_fused(const float* dy, const float* x1, const float* x2, int C1, int C2,
                                        const float* gamma, const float* beta, const float* mean, const float* rstd,
                                        float* dx1, float* dx2, float* dgamma_part, float* dbeta_part, int B, int hw,
                                        int G, int act, float keep, unsigned long long seed, unsigned long long offset,
                                        const unsigned long long* seed_dev, unsigned* dx1max, unsigned* dx2max,
                                        const float* add1, const float* add2, const float* add1b, float* dxsum_part,
                                        float* dgamma, float* dbeta, float* dxsum, float* dxsum2, unsigned* tickets,
                                        hipStream_t stream) {
  const int Ct = C1 + C2;
  if (hw != HW || B <= 0 || G <= 0 || Ct % G != 0 || !tickets || !dgamma || !dbeta || !dgamma_part || !dbeta_part ||
      (dxsum && !dxsum_part) || (dxsum2 && !dxsum) || Ct / 32 > 16)
    return (int)hipErrorInvalidValue;
  const int cpg = Ct / G;
  if (cpg % 4 != 0 || 32 % cpg != 0 || C1 % 32 != 0 || C2 % 32 != 0) return (int)hipErrorInvalidValue;
  GnBwdArgs a{dy, x1, x2, C1, C2, gamma, beta, mean, rstd, dx1, dx2, dgamma_part, dbeta_part,
              B, G, act, keep, seed, offset, 0, dx1max, dx2max, add1, add2, dxsum_part, seed_dev, tickets, dgamma, dbeta,
              dxsum, dxsum2, nullptr, nullptr, nullptr, add1b};
  hipLaunchKernelGGL(gn_bwd_kernel_1pass, dim3(B, Ct / 32), dim3(512), 0, stream, a);
  MULAN_CHECK_LAUNCH();
}

// mulan_groupnorm_bwd_fused for a single input whose gradient feeds ONLY the f16x3 kernels of the convolution in front
// (its input-gradient convolution and its weight gradient; the bias / FiLM gradients come from dxsum_part / dxsum): dx is
// written as their split fp16 operand planes (dxplanes: mulan_conv3x3_planes_bytes(B, 32, 32, C) bytes) instead of as
// fp32, so that the convolution neither splits its input nor stores planes out of the MFMA kernel.  The planes are
// scaled with an a-priori bound of |dx[b]| (see gn_bwd_kernel_1pass) formed from dymax ([B][16] maxima of dy, as the
// convolution that produced dy leaves them), rstd and max|gamma|; dxmax [B][16] receives that bound in the maxima format.
// keepbits (optional): the dropout keep-bits mulan_groupnorm_fwd_planes_keepbits stored in the forward pass (else the
// kernel re-draws them from seed / offset: same bits).
MULAN_API int mulan_groupnorm_bwd_fused_planes(const float* dy, const unsigned* dymax, const float* x, int C,
                                               const float* gamma, const float* beta, const float* mean,
                                               const float* rstd, void* dxplanes, float* dgamma_part, float* dbeta_part,
                                               int B, int hw, int G, int act, float keep, unsigned long long seed,
                                               unsigned long long offset, const unsigned long long* seed_dev,
                                               unsigned* dxmax, float* dxsum_part, float* dgamma, float* dbeta,
                                               float* dxsum, float* dxsum2, unsigned* tickets,
                                               const unsigned* keepbits, hipStream_t stream) {
  if (hw != HW || B <= 0 || G <= 0 || C % G != 0 || !tickets || !dgamma || !dbeta || !dgamma_part || !dbeta_part ||
      !dxplanes || !dymax || !dxmax || !(keep > 0.f) || (dxsum && !dxsum_part) || (dxsum2 && !dxsum) || C / 32 > 16 ||
      C % 32 != 0 || (size_t)B * HW * C * 4 >= 0x80000000ull)
    return (int)hipErrorInvalidValue;
  const int cpg = C / G;
  if (cpg % 4 != 0 || 32 % cpg != 0) return (int)hipErrorInvalidValue;
  GnBwdArgs a{dy, x, nullptr, C, 0, gamma, beta, mean, rstd, nullptr, nullptr, dgamma_part, dbeta_part,
              B, G, act, keep, seed, offset, 0, dxmax, nullptr, nullptr, nullptr, dxsum_part, seed_dev, tickets, dgamma,
              dbeta, dxsum, dxsum2, static_cast<unsigned char*>(dxplanes), dymax, keep < 1.f ? keepbits : nullptr};
  hipLaunchKernelGGL(gn_bwd_kernel_1pass, dim3(B, C / 32), dim3(512), 0, stream, a);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_groupnorm_bwd(const float* dy, const float* x1, const float* x2, int C1, int C2,
                                  const float* gamma, const float* beta, const float* mean, const float* rstd,
                                  float* dx1, float* dx2, float* dgamma_part, float* dbeta_part, int B, int hw,
                                  int G, int act, float keep, unsigned long long seed, unsigned long long offset,
                                  int accumulate, unsigned* dx1max, unsigned* dx2max, const float* add1,
                                  const float* add2, float* dxsum_part, hipStream_t stream) {
  return mulan_groupnorm_bwd_dyn(dy, x1, x2, C1, C2, gamma, beta, mean, rstd, dx1, dx2, dgamma_part, dbeta_part, B, hw,
                                 G, act, keep, seed, offset, nullptr, accumulate, dx1max, dx2max, add1, add2,
                                 dxsum_part, stream);
}

// ---- streaming forms (round 5): statistics / group sums handed in by the producers, see gn_fwd_stream_kernel.
// Forward.  Exactly one of y (fp32, + ymax: the true maxima) and yplanes (split planes, + ymax: the bound) is written.
// xstats1 (, xstats2 when C2 > 0): the partial sums the convolutions that produced x1 (, x2) left in their `ystats`
// ([B][xstats_tiles][C / 4][2]: sum and sum of squares per image, row tile of the producing launch (4, 8 or 16 per image:
// mulan_conv3x3_f16x3_tile_rows) and channel quad); mean / rstd [B, G] are then outputs.
// xstats1 == NULL: mean / rstd are inputs (e.g. from mulan_groupnorm_stats).  keepbits (optional, keep < 1): the
// keep-bits as drawn, in the layout of mulan_groupnorm_fwd_planes_keepbits.
MULAN_API int mulan_groupnorm_fwd_stream(const float* x1, const float* x2, int C1, int C2, const float* gamma,
                                         const float* beta, float* y, void* yplanes, float* mean, float* rstd,
                                         const float* xstats1, const float* xstats2, int xstats_tiles, int B, int hw,
                                         int G, float eps, int act, float keep, unsigned long long seed,
                                         unsigned long long offset,
                                         const unsigned long long* seed_dev, unsigned* ymax, unsigned* keepbits,
                                         hipStream_t stream) {
  const int Ct = C1 + C2;
  if (hw != HW || B <= 0 || G <= 0 || Ct % G != 0 || !mean || !rstd || (!y == !yplanes) || !(keep > 0.f) ||
      (yplanes && !ymax) || (xstats1 && C2 > 0 && !xstats2) || (keepbits && !(keep < 1.f)) ||
      (xstats1 && xstats_tiles != 4 && xstats_tiles != 8 && xstats_tiles != 16))
    return (int)hipErrorInvalidValue;
  const int cpg = Ct / G;
  if (cpg % 4 != 0 || 32 % cpg != 0 || C1 % 32 != 0 || C2 % 32 != 0 || C1 % cpg != 0 || (ymax && Ct / 32 > 16))
    return (int)hipErrorInvalidValue;
  GnArgs a{x1, x2, C1, C2, gamma, beta, y, mean, rstd, B, G, eps, act, keep, seed, offset, ymax, seed_dev,
           static_cast<unsigned char*>(yplanes), keepbits, 0, xstats1, xstats2, xstats_tiles};
  // tune[20] (dev A/B): quarters of a slab per block -- 0 / 4: 1024 threads, 2: 512, 1: 256 (the maxima array has 16
  // entries per image: (slabs) x (z blocks) must fit)
  int nsp = g_mulan_tune[20] == 1 ? 1 : (g_mulan_tune[20] == 2 ? 2 : 4);
  while (ymax && !yplanes && (Ct / 32) * (4 / nsp) > 16) nsp *= 2;
  const dim3 grid(B, Ct / 32, 4 / nsp);
  if (nsp == 1) hipLaunchKernelGGL(gn_fwd_stream_kernel<1>, grid, dim3(256), 0, stream, a);
  else if (nsp == 2) hipLaunchKernelGGL(gn_fwd_stream_kernel<2>, grid, dim3(512), 0, stream, a);
  else hipLaunchKernelGGL(gn_fwd_stream_kernel<4>, grid, dim3(1024), 0, stream, a);
  MULAN_CHECK_LAUNCH();
}

// Backward (the arguments of mulan_groupnorm_bwd_fused / _fused_planes in one entry point).  gstats [B][4][(C1 + C2) / 4][2]:
// the partial sums of g gamma and g gamma xhat (g = dy mask / keep act'(u)) per image, 8-row tile and channel quad, as the
// producer of dy would leave them (no shipped kernel does: DESIGN.md section 3.3; tests and tools form them with torch).  dx1planes
// (optional; then C2 == 0, no add*, dymax required): dx1 as split planes instead of fp32, dx1max receives the bound.
MULAN_API int mulan_groupnorm_bwd_stream(const float* dy, const unsigned* dymax, const float* x1, const float* x2, int C1,
                                         int C2, const float* gamma, const float* beta, const float* mean,
                                         const float* rstd, const float* gstats, float* dx1, float* dx2, void* dx1planes,
                                         float* dgamma_part, float* dbeta_part, int B, int hw, int G, int act, float keep,
                                         unsigned long long seed, unsigned long long offset,
                                         const unsigned long long* seed_dev, unsigned* dx1max, unsigned* dx2max,
                                         const float* add1, const float* add2, const float* add1b, float* dxsum_part,
                                         float* dgamma, float* dbeta, float* dxsum, float* dxsum2, unsigned* tickets,
                                         const unsigned* keepbits, hipStream_t stream) {
  const int Ct = C1 + C2;
  if (hw != HW || B <= 0 || G <= 0 || Ct % G != 0 || !gstats || !tickets || !dgamma || !dbeta || !dgamma_part ||
      !dbeta_part || (dxsum && !dxsum_part) || (dxsum2 && !dxsum) || Ct / 32 > 16 || !(keep > 0.f) ||
      (!dx1 == !dx1planes) || (C2 > 0 && !dx2) ||
      (dx1planes && (C2 != 0 || add1 || add1b || !dymax || !dx1max || (size_t)B * HW * C1 * 4 >= 0x80000000ull)))
    return (int)hipErrorInvalidValue;
  const int cpg = Ct / G;
  if (cpg % 4 != 0 || 32 % cpg != 0 || C1 % 32 != 0 || C2 % 32 != 0) return (int)hipErrorInvalidValue;
  GnBwdArgs a{dy, x1, x2, C1, C2, gamma, beta, mean, rstd, dx1, dx2, dgamma_part, dbeta_part,
              B, G, act, keep, seed, offset, 0, dx1max, dx2max, add1, add2, dxsum_part, seed_dev, tickets, dgamma, dbeta,
              dxsum, dxsum2, static_cast<unsigned char*>(dx1planes), dymax, keep < 1.f ? keepbits : nullptr, add1b, gstats};
  // tune[21] (dev A/B): quarters of a slab per block (as tune[20]); tune[22] = 1 / 2: load batches of 4 / 2 pixels instead of 8.
  // The per-block partial rows are [B * 4 / nsp][Ct]: dgamma_part / dbeta_part / dxsum_part must hold 4 B rows.
  int nsp = g_mulan_tune[21] == 1 ? 1 : (g_mulan_tune[21] == 2 ? 2 : 4);
  const int C12 = C1 > C2 ? C1 : C2;
  if (g_mulan_tune[21] == 3 && (C12 / 32) * 4 <= 16) {
    // (dev probe) the thin kernel: no reduction over the samples -- dgamma / dbeta / dxsum are NOT written
    const dim3 grid(B, Ct / 32, 4);
    const int nadd = add1b ? 2 : ((add1 || add2) ? 1 : 0);
    const bool drop = keep < 1.f;
#define MULAN_GN_THIN(PL, NA)                                                                                     \
  if (drop) hipLaunchKernelGGL((gn_bwd_thin_kernel<PL, NA, true>), grid, dim3(256), 0, stream, a);                \
  else hipLaunchKernelGGL((gn_bwd_thin_kernel<PL, NA, false>), grid, dim3(256), 0, stream, a);
    if (dx1planes) { MULAN_GN_THIN(true, 0) }
    else if (nadd == 2) { MULAN_GN_THIN(false, 2) }
    else if (nadd == 1) { MULAN_GN_THIN(false, 1) }
    else { MULAN_GN_THIN(false, 0) }
#undef MULAN_GN_THIN
    MULAN_CHECK_LAUNCH();
  }
  while ((dx1max || dx2max) && (C12 / 32) * (4 / nsp) > 16) nsp *= 2;
  const dim3 grid(B, Ct / 32, 4 / nsp);
  const int ub = g_mulan_tune[22];
#define MULAN_GN_BWD_STREAM(NSP)                                                                                  \
  if (ub == 1) hipLaunchKernelGGL((gn_bwd_stream_kernel<NSP, 4>), grid, dim3(256 * NSP), 0, stream, a);           \
  else if (ub == 2) hipLaunchKernelGGL((gn_bwd_stream_kernel<NSP, 2>), grid, dim3(256 * NSP), 0, stream, a);      \
  else hipLaunchKernelGGL((gn_bwd_stream_kernel<NSP, 8>), grid, dim3(256 * NSP), 0, stream, a);
  if (nsp == 1) { MULAN_GN_BWD_STREAM(1) } else if (nsp == 2) { MULAN_GN_BWD_STREAM(2) } else { MULAN_GN_BWD_STREAM(4) }
#undef MULAN_GN_BWD_STREAM
  MULAN_CHECK_LAUNCH();
}
