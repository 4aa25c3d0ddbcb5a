// Dev probe (round 3, for the round-4 plan): how long does a GroupNorm-backward-shaped kernel (read x and dy slabs into
// registers, block reduction, write dx: 2 reads + 1 write of [B, 1024, C] fp32) take when 120 of the 256 CUs are held
// by another stream's blocks (the weight-gradient kernel: one block per CU, 137 KB of LDS), for two block shapes:
//   slab512: 512 threads own (sample, 32 channels), 16 pixels x 2 float4 per thread -> one block per CU (the shipped form)
//   slab256: 256 threads own (sample, 16 channels), the same per-thread state       -> two blocks per CU
// Build / run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/gn_share_probe tools/gn_share_probe.hip && /tmp/gn_share_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int HW = 1024;

template <int THREADS, int QUADS>     // QUADS float4 columns per block (8: 32 channels, 4: 16 channels)
__global__ __launch_bounds__(THREADS) void slab_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ dx, int C) {
  constexpr int PROWS = THREADS / QUADS, NPB = HW / PROWS;      // 64 pixel rows, 16 pixels per thread
  __shared__ float red[64];
  const int tid = threadIdx.x, quad = tid % QUADS, prow = tid / QUADS;
  const size_t base = (size_t)blockIdx.x * HW * C + blockIdx.y * (QUADS * 4) + quad * 4;
  f32x4 xv[NPB], gv[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    xv[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + base + (size_t)(prow + PROWS * i) * C));
    gv[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dy + base + (size_t)(prow + PROWS * i) * C));
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NPB; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1 += gv[i][e]; s2 += gv[i][e] * xv[i][e]; }
  for (int o = QUADS; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < QUADS) { red[wave * 8 + lane] = s1; red[32 + wave * 8 + lane] = s2; }
  __syncthreads();
  float t1 = 0.f, t2 = 0.f;
  for (int w = 0; w < THREADS / 64; ++w) { t1 += red[(w & 3) * 8 + quad]; t2 += red[32 + (w & 3) * 8 + quad]; }
  t1 *= 1.f / 4096.f; t2 *= 1.f / 4096.f;
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = gv[i][e] - t1 - xv[i][e] * t2;
    *reinterpret_cast<f32x4*>(dx + base + (size_t)(prow + PROWS * i) * C) = o;
  }
}

// holds one CU per block (137 KB of LDS) for `cycles` core cycles, busy on the matrix cores
__global__ __launch_bounds__(256) void hog_kernel(long long cycles, float* sink) {
  extern __shared__ float lds[];
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  f32x16 acc[20];                       // 320 accumulator registers: occupancy one wave per SIMD, i.e. the whole register file of
  for (int i = 0; i < 20; ++i) acc[i] = f32x16{};   // the SIMDs (one wave each), like the weight-gradient kernel: nothing co-resides
  f16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = a;
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 20; ++i) t += acc[i][0];
  if (t == 12345.f) sink[0] = t + lds[threadIdx.x];
}

int main() {
  const int B = 128;
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 137472);
  float* sink; hipMalloc(&sink, 64);
  for (int C : {128, 256}) {
    const size_t n = (size_t)B * HW * C;
    float *x, *dy, *dx;
    hipMalloc(&x, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dx, n * 4);
    hipMemset(x, 0, n * 4); hipMemset(dy, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int hog : {0, 120, 96, 144}) {
      for (int shape = 0; shape < 2; ++shape) {
        std::vector<float> ts;
        for (int rep = 0; rep < 9; ++rep) {
          if (hog) hipLaunchKernelGGL(hog_kernel, dim3(hog), dim3(256), 137472, s1, 2000000LL, sink);   // ~1 ms
          hipStreamSynchronize(s2);
          // let the hog blocks settle on their CUs
          for (volatile int spin = 0; spin < 200000; ++spin) {}
          hipEventRecord(e0, s2);
          for (int k = 0; k < 4; ++k) {
            if (shape == 0) hipLaunchKernelGGL((slab_kernel<512, 8>), dim3(B, C / 32), dim3(512), 0, s2, x, dy, dx, C);
            else hipLaunchKernelGGL((slab_kernel<256, 4>), dim3(B, C / 16), dim3(256), 0, s2, x, dy, dx, C);
          }
          hipEventRecord(e1, s2);
          hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          ts.push_back(ms * 1e3f / 4);
          hipDeviceSynchronize();
        }
        std::sort(ts.begin(), ts.end());
        printf("C=%d  %3d CUs held  %s: %7.1f us per launch (median of 9 x 4), %.2f TB/s\n", C, hog,
               shape == 0 ? "512 threads x 32 channels" : "256 threads x 16 channels", ts[4], 3.0 * n * 4 / ts[4] / 1e6);
      }
    }
    hipFree(x); hipFree(dy); hipFree(dx);
  }
  return 0;
}
