// Dev probe: HBM rate of a "read slab into registers, then write it back" kernel (the GroupNorm access shape) for
// two block->memory mappings on [B, 1024, C] fp32:
//   strided: block (b, j) owns channels [32 j, 32 j + 32) of all 1024 pixels  (128 B out of every 4 C bytes)
//   contig : block (b, j) owns all C channels of pixels [P j, P j + P), P = 1024 * 32 / C  (one contiguous 128 KB run)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int HW = 1024, NP = 32;

__global__ __launch_bounds__(256) void strided(const float* __restrict__ x, float* __restrict__ y, int C) {
  const int tid = threadIdx.x, quad = tid & 7, prow = tid >> 3;
  const size_t base = (size_t)blockIdx.x * HW * C + blockIdx.y * 32 + quad * 4;
  f32x4 v[NP];
#pragma clang loop unroll(full)
  for (int i = 0; i < NP; ++i) v[i] = *reinterpret_cast<const f32x4*>(x + base + (size_t)(prow + 32 * i) * C);
  float s = 0.f;
#pragma clang loop unroll(full)
  for (int i = 0; i < NP; ++i) s += v[i][0];
  s += __shfl_xor(s, 8, 64);
  __syncthreads();
#pragma clang loop unroll(full)
  for (int i = 0; i < NP; ++i) {
    f32x4 o = v[i] * s;
    *reinterpret_cast<f32x4*>(y + base + (size_t)(prow + 32 * i) * C) = o;
  }
}

__global__ __launch_bounds__(256) void contig(const float* __restrict__ x, float* __restrict__ y, int C) {
  // block owns 32 * 1024 floats = 128 KB contiguous; thread t reads float4 number t + 256 i
  const size_t base = ((size_t)blockIdx.x * (C / 32) + blockIdx.y) * 32 * HW + threadIdx.x * 4;
  f32x4 v[NP];
#pragma clang loop unroll(full)
  for (int i = 0; i < NP; ++i) v[i] = *reinterpret_cast<const f32x4*>(x + base + (size_t)i * 1024);
  float s = 0.f;
#pragma clang loop unroll(full)
  for (int i = 0; i < NP; ++i) s += v[i][0];
  s += __shfl_xor(s, 8, 64);
  __syncthreads();
#pragma clang loop unroll(full)
  for (int i = 0; i < NP; ++i) {
    f32x4 o = v[i] * s;
    *reinterpret_cast<f32x4*>(y + base + (size_t)i * 1024) = o;
  }
}

template <class F> float time_it(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  for (int C : {128, 256}) {
    const int B = 128;
    const size_t n = (size_t)B * HW * C;
    float *x, *y;
    hipMalloc((void**)&x, n * 4); hipMalloc((void**)&y, n * 4);
    hipMemset(x, 0, n * 4);
    const dim3 g(B, C / 32);
    const float t1 = time_it([&] { strided<<<g, 256>>>(x, y, C); }, 20);
    const float t2 = time_it([&] { contig<<<g, 256>>>(x, y, C); }, 20);
    printf("C=%d  strided %.1f us = %.2f TB/s   contig %.1f us = %.2f TB/s\n", C, t1 * 1e3, 2 * n * 4 / t1 * 1e-9,
           t2 * 1e3, 2 * n * 4 / t2 * 1e-9);
    hipFree(x); hipFree(y);
  }
  return 0;
}
