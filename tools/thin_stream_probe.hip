// Dev probe (round 4, for DESIGN section 7 item 2): can an HBM-streaming kernel that needs at most 64 vector registers per
// lane run ON THE SAME CUs as the 3x3 weight-gradient kernel's blocks (one wave per SIMD at 448 registers, 137 KB of LDS:
// 64 registers and 23 KB are left), at what bandwidth, and what does it cost the matrix-core kernel?  If it can, the
// GroupNorm backward of the main chain (a fat kernel today: 512 threads x 238 registers, it cannot share a CU with anything)
// could hide under a weight-gradient launch that owns all 240 CUs instead of splitting the chip 120 / 136.
//   mfma_hog : 240 blocks x 256 threads, 16 accumulator tiles + fragments, 137 KB of LDS, a fixed number of
//              iterations of 48 MFMAs fed by LDS fragment reads (like the real kernel) -- timed
//   thin     : 2 reads + 1 write over [128, 1024, 128] fp32 (201 MB: the GroupNorm-backward mix), grid-stride, capped at
//              64 VGPRs (amdgpu_waves_per_eu(8, 8)), 4 float4 per stream in flight per lane -- timed
// Build / run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/thin_stream_probe tools/thin_stream_probe.hip && /tmp/thin_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int HOG_LDS = 137472;
constexpr int NACC = 16;        // 256 accumulator registers + fragments (hipcc spills above that with the MFMA builtin); what limits
                                // co-residency like the real kernel's 448 registers is the LDS: see THIN_LDS
constexpr int THIN_LDS = 20480; // a thin block asks for 20 KB of (unused) LDS: beside a hog block's 137 KB only ONE thin block =
                                // one thin wave per SIMD fits on the CU, which is what 64 free registers would allow

__global__ __launch_bounds__(256) void mfma_hog(int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  // varied operand bits (a power-limited part draws less on constant / zero operands)
  for (int i = tid; i < HOG_LDS / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003800u ^ (i * 2654435761u & 0x03ff03ffu);
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x16{};
  const unsigned char* base = lds + (tid & 63) * 16 + (tid >> 6) * 8192;
  for (int it = 0; it < iters; ++it) {
    const unsigned char* p = base + (it & 7) * 1024;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = *reinterpret_cast<const f16x8*>(p + i * 32768);
      b[i] = *reinterpret_cast<const f16x8*>(p + i * 32768 + 16384);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][7];
  if (t == 12345.f) sink[0] = t;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void thin(const f32x4* __restrict__ x, const f32x4* __restrict__ dy, f32x4* __restrict__ dx, size_t n4, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);     // a memory wave beside a matrix-core wave of the same SIMD: issue first
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n4; i += 4 * stride) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = __builtin_nontemporal_load(x + i + u * stride);
      b[u] = __builtin_nontemporal_load(dy + i + u * stride);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) dx[i + u * stride] = a[u] * 0.5f + b[u];
  }
}

int main() {
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_hog), hipFuncAttributeMaxDynamicSharedMemorySize, HOG_LDS);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(mfma_hog));
  printf("mfma_hog: %d registers, thin: ", fa.numRegs);
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(thin));
  printf("%d registers\n", fa.numRegs);
  float* sink; hipMalloc(&sink, 64);
  const size_t n4 = (size_t)128 * 1024 * 128 / 4;
  f32x4 *x, *dy, *dx;
  hipMalloc(&x, n4 * 16); hipMalloc(&dy, n4 * 16); hipMalloc(&dx, n4 * 16);
  hipMemset(x, 0, n4 * 16); hipMemset(dy, 0, n4 * 16);
  hipEvent_t h0, h1, t0, t1;
  hipEventCreate(&h0); hipEventCreate(&h1); hipEventCreate(&t0); hipEventCreate(&t1);
  const int iters = 1500;                  // ~ 400 us of matrix-core work per block
  const int thin_launches = 6;
  for (int prio : {0, 1})
  for (int thin_lds : {THIN_LDS})
  for (int hog_blocks : {0, 256, 240}) {
    for (int thin_blocks : {0, 1024, 2048}) {
      if (!hog_blocks && !thin_blocks) continue;
      std::vector<float> th, tt;
      for (int rep = 0; rep < 7; ++rep) {
        hipDeviceSynchronize();
        if (hog_blocks) {
          hipEventRecord(h0, s1);
          hipLaunchKernelGGL(mfma_hog, dim3(hog_blocks), dim3(256), HOG_LDS, s1, iters, sink);
          hipEventRecord(h1, s1);
        }
        if (thin_blocks) {
          for (volatile int spin = 0; spin < 100000; ++spin) {}       // let the hog blocks land first
          hipEventRecord(t0, s2);
          for (int k = 0; k < thin_launches; ++k)
            hipLaunchKernelGGL(thin, dim3(thin_blocks), dim3(256), thin_lds, s2, x, dy, dx, n4, prio);
          hipEventRecord(t1, s2);
        }
        hipDeviceSynchronize();
        float ms;
        if (hog_blocks) { hipEventElapsedTime(&ms, h0, h1); th.push_back(ms * 1e3f); }
        if (thin_blocks) { hipEventElapsedTime(&ms, t0, t1); tt.push_back(ms * 1e3f / thin_launches); }
      }
      std::sort(th.begin(), th.end()); std::sort(tt.begin(), tt.end());
      printf("prio %d  thin LDS %5d  hog blocks %3d  thin blocks %4d :", prio, thin_lds, hog_blocks, thin_blocks);
      if (hog_blocks) printf("  hog %7.1f us", th[3]);
      if (thin_blocks) printf("  thin %6.1f us per launch = %.2f TB/s", tt[3], 3.0 * n4 * 16 / tt[3] / 1e6);
      printf("\n");
      fflush(stdout);
    }
  }
  return 0;
}
