// Shared device helpers for the MuLAN gfx950 kernels (wave64, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MULAN_API extern "C" __attribute__((visibility("default")))

// the public header, seen by every translation unit of the library: a definition whose parameter list differs from
// its declaration there does not compile
#define MULAN_STREAM_T
typedef hipStream_t mulan_stream_t;
#include "../../include/mulan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MULAN_CHECK_LAUNCH() return (int)hipGetLastError()

// A 16-byte load of data this kernel reads once (the rows of the per-pixel dense kernel, its residual, optimizer state).
// tools/mall_probe.hip (profiles/r04_mall_probe.log): a plain streaming consumer reads 13-25 % faster with non-temporal
// loads, and the GroupNorm forward kernel gains 0.5 ms per step from them -- but in these kernels the A/B went the other
// way (profiles/r04_nt_loads_ab.log: the step 0.3 ms SLOWER with nt loads in the dense kernel + AdamW, and the
// convolution's residual loads cost it 0.527 vs 0.538 of its roofline), so the default is the plain load;
// -DMULAN_NT_MORE=1: dev A/B build.  nt STORES bypass the Infinity Cache (the next reader falls to the cold rate): never.
#ifndef MULAN_NT_MORE
#define MULAN_NT_MORE 0
#endif
__device__ __forceinline__ f32x4 ld_stream4(const float* p) {
#if MULAN_NT_MORE
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
  return *reinterpret_cast<const f32x4*>(p);
#endif
}

// developer tuning knobs (mulan_set_tuning): [0] conv fwd variant, [1] wgrad resident-block target
extern int g_mulan_tune[32];
extern unsigned long long* g_mulan_debug_buffer;   // dev-only stamp buffer (>= 64 u64), normally null

// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31] lives in accumulator register r (0..15).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma32_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// d/dx [x * sigmoid(x)]
__device__ __forceinline__ float silu_grad_f(float x) {
  const float s = sigmoid_f(x);
  return s * (1.f + x * (1.f - s));
}
__device__ __forceinline__ float softplus_f(float x) {
  // log1p(exp(x)) written the numerically safe way (jax.nn.softplus = logaddexp(x, 0))
  return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based RNG: (seed, counter) -> 4 x u32.  The numpy oracle implements the same
// function, so dropout masks are bit-reproducible on the host.
struct Philox4 { uint32_t x, y, z, w; };
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
  uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Block-wide sum for 256-thread blocks; `red` is >= 4 floats of LDS.  All threads get the result.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
