// Stand-alone probe (dev tool, not part of the library): what stops a v_mfma_f32_32x32x2_f32 stream from issuing
// back to back?  hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

// MODE 0: operands in registers.  MODE 1: 10 ds_read_b32 per 9 MFMAs (wgrad shape), ping-pong.
// MODE 2: conv shape: 4 accumulators, A from 4 b128 reads per 32 MFMAs, B 2x b32 per 4 MFMAs, one-step lookahead.
// MODE 3: as 2 plus a __syncthreads() every 32 MFMAs.   MODE 4: as 1 without the lookahead (read then use).
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 16384; i += 256) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc[9];
  for (int t = 0; t < 9; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  if (MODE == 0) {
    float a = lds[tid], b = lds[tid + 256];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = MFMA(a, b, acc[t]);
    }
  } else if (MODE == 1 || MODE == 4) {
    float a0[9], a1[9], b0, b1;
    auto rd = [&](int s, float* av, float& bv) {
      const float* base = lds + ((s & 31) * 64 + li + lh * 2048);
      bv = base[8192 + (s & 15)];
#pragma unroll
      for (int t = 0; t < 9; ++t) av[t] = base[(t / 3) * 2176 + (t % 3) * 64];
    };
    if (MODE == 1) rd(0, a0, b0);
    for (int it = 0; it < iters; it += 2) {
      if (MODE == 1) {
        rd(it + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = MFMA(a0[t], b0, acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        rd(it + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = MFMA(a1[t], b1, acc[t]);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        rd(it, a0, b0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = MFMA(a0[t], b0, acc[t]);
        rd(it + 1, a1, b1);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = MFMA(a1[t], b1, acc[t]);
      }
    }
  } else {
    // iters counts 32-MFMA "taps" here
    for (int it = 0; it < iters; ++it) {
      const float* pb = lds + (it & 7) * 20;
      const float* wb = lds + 8192 + (it & 1) * 2048;
      f32x4 a4[2][2];
#pragma unroll
      for (int k8 = 0; k8 < 2; ++k8)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          a4[k8][mt] = *reinterpret_cast<const f32x4*>(pb + ((mt + 1) * 34 + li) * 20 + k8 * 8 + 4 * lh);
      float bc[2], bn[2];
      bc[0] = wb[(4 * lh) * 128 + li];
      bc[1] = wb[(4 * lh) * 128 + 32 + li];
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        if (st + 1 < 8) {
          const int kk = ((st + 1) >> 2) * 8 + 4 * lh + ((st + 1) & 3);
          bn[0] = wb[kk * 128 + li];
          bn[1] = wb[kk * 128 + 32 + li];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt * 2 + nt] = MFMA(a4[st >> 2][mt][st & 3], bc[nt], acc[mt * 2 + nt]);
        bc[0] = bn[0]; bc[1] = bn[1];
      }
      if (MODE == 3) __syncthreads();
    }
  }
  float s = 0.f;
  for (int t = 0; t < 9; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters, double mfma_per_iter) {
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(s);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double t = ms * 1e-3 / 5;
  const double flops = (double)blocks * 4 * iters * mfma_per_iter * 4096.0;
  printf("%-46s blocks=%4d: %8.1f us  %7.2f TFLOP/s (%5.1f%% of 157.3)\n", name, blocks, t * 1e6, flops / t / 1e12,
         flops / t / 1e12 / 157.3 * 100);
  hipFree(out);
}

int main() {
  for (int blocks : {256, 512}) {
    run<0>("0: registers only, 9 acc", blocks, 2048, 9);
    run<1>("1: 10 ds_read_b32 / 9 MFMA, lookahead", blocks, 2048, 9);
    run<4>("4: 10 ds_read_b32 / 9 MFMA, no lookahead", blocks, 2048, 9);
    run<2>("2: conv tap (b128 A, b32 B lookahead), 4 acc", blocks, 576, 32);
    run<3>("3: conv tap + __syncthreads per tap", blocks, 576, 32);
  }
  return 0;
}
