// Dev probe (round 6): a bare fp16 MFMA loop on random operands held in registers (no LDS, no memory traffic), every CU busy,
// 1 or 2 waves per SIMD, ~3 s per variant -- tools/mfma_energy_probe.py samples the socket power beside it: joules per executed
// FLOP of the matrix cores alone, the floor the convolution / weight-gradient kernels are priced against.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_energy_probe.hip -o tools/bin/mfma_energy_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <bool M16, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void loop(const _Float16* __restrict__ src, float* sink, int iters) {
  const int tid = threadIdx.x + blockIdx.x * WAVES * 64;
  h8 a[4], b[4];
  for (int k = 0; k < 4; ++k)
    for (int j = 0; j < 8; ++j) { a[k][j] = src[(tid * 64 + k * 8 + j) & 0xfffff]; b[k][j] = src[(tid * 64 + 32 + k * 8 + j) & 0xfffff]; }
  f16v acc32[4];
  f4 acc16[16];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc32[t][r] = 0.f;
  for (int t = 0; t < 16; ++t) acc16[t] = f4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (M16) {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc16[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 3], b[(u >> 2) & 3], acc16[u], 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc32[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u >> 1) & 3], acc32[u & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc32[t][r];
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) s += acc16[t][r];
  sink[tid] = s;
}

template <bool M16, int WAVES>
void run(const _Float16* src, float* sink, double seconds) {
  const int iters = 20000;
  const double flop_per_launch = 256.0 * WAVES * iters * (M16 ? 16 * 2.0 * 16 * 16 * 32 : 8 * 2.0 * 32 * 32 * 16);
  loop<M16, WAVES><<<256, WAVES * 64>>>(src, sink, 100);
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  int n = 0;
  double dt = 0;
  while (dt < seconds) {
    for (int k = 0; k < 4; ++k) loop<M16, WAVES><<<256, WAVES * 64>>>(src, sink, iters);
    hipDeviceSynchronize();
    n += 4;
    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  printf("RESULT %s waves_per_simd=%d seconds=%.3f tflops=%.1f\n", M16 ? "16x16x32" : "32x32x16", WAVES / 4, dt, flop_per_launch * n / dt / 1e12);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  _Float16* src; float* sink;
  hipMalloc((void**)&src, (1 << 20) * 2);
  hipMalloc((void**)&sink, 256 * 512 * 4);
  _Float16* h = (_Float16*)malloc((1 << 20) * 2);
  srand(1);
  for (int i = 0; i < (1 << 20); ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
  hipMemcpy(src, h, (1 << 20) * 2, hipMemcpyHostToDevice);
  run<false, 4>(src, sink, seconds);
  run<true, 4>(src, sink, seconds);
  run<false, 8>(src, sink, seconds);
  run<true, 8>(src, sink, seconds);
  return 0;
}
