// Dev probe (round 5): a stand-in for a ring all-reduce with RCCL's FOOTPRINT -- `blocks` workgroups of
// 512 threads (RCCL runs one workgroup per channel) that, for the ring's duration, stream the bucket through HBM twice
// (read + write of 2 x the bucket's bytes, paced evenly over `ns`), optionally holding `lds` bytes of LDS each so that no
// matrix-core block can share their CUs.  Loaded by tools/overlap_timing_probe.py --footprint K.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libfootprint.so tools/footprint_kernel.hip
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void footprint_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n16,
                                                         unsigned long long ns, int passes) {
  extern __shared__ unsigned char lds[];
  if (threadIdx.x == 0 && n16 == ~(size_t)0) lds[0] = 1;            // (keeps the allocation)
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();    // 100 MHz
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
  constexpr int CH = 512 * 8;                                        // 64 KB per block and chunk
  const size_t nchunk = (hi > lo ? (hi - lo + CH - 1) / CH : 0) * (size_t)passes;
  size_t done = 0;
  for (int p = 0; p < passes; ++p)
    for (size_t c = lo; c < hi; c += CH, ++done) {
      // pace: chunk k of n starts no earlier than k / n of the duration
      const unsigned long long due = t0 + (unsigned long long)((double)ns * 0.1 * (double)done / (double)(nchunk ? nchunk : 1));
      while (__builtin_amdgcn_s_memrealtime() < due) __builtin_amdgcn_s_sleep(8);
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const size_t i = c + u * 512 + threadIdx.x; v[u] = i < hi ? __builtin_nontemporal_load(src + i) : f32x4{0, 0, 0, 0}; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const size_t i = c + u * 512 + threadIdx.x; if (i < hi) dst[i] = v[u]; }
    }
  while (__builtin_amdgcn_s_memrealtime() < t0 + ns / 10) __builtin_amdgcn_s_sleep(8);
}

extern "C" int footprint_copy(const void* src, void* dst, size_t bytes, int blocks, unsigned long long ns, int lds, void* stream) {
  static int configured = -1;
  if (lds > 0 && configured != lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(footprint_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    configured = lds;
  }
  // an all-reduce reads and writes the bucket (N - 1) / N times each way per step of the ring; 2 passes = 2 x the bucket
  hipLaunchKernelGGL(footprint_kernel, dim3(blocks), dim3(512), lds, static_cast<hipStream_t>(stream),
                     static_cast<const f32x4*>(src), static_cast<f32x4*>(dst), bytes / 16, ns, 2);
  return (int)hipGetLastError();
}
