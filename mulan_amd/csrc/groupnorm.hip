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
