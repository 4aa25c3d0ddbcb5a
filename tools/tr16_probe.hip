#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  // block: rows (k) kbase..kbase+3, cols cbase..cbase+15 ; row stride 64 shorts
  const int kbase = 8 * (g >> 1), cbase = 16 * (g & 1);
  __attribute__((address_space(3))) s16x4* ptr = (__attribute__((address_space(3))) s16x4*)(lds + (kbase + q) * 64 + cbase + 4 * p);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
#include <cstdio>
int main() {
  short* d; hipMalloc(&d, 256 * sizeof(short));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    int g = lane >> 4, i = lane & 15, kbase = 8 * (g >> 1), cbase = 16 * (g & 1);
    for (int e = 0; e < 4; ++e) { int exp = (kbase + e) * 64 + cbase + i; if (h[lane * 4 + e] != exp) ++bad; }
  }
  printf("tr16 probe: %d mismatches; lane0 = %d %d %d %d, lane17 = %d %d %d %d, lane35 = %d %d %d %d\n", bad, h[0], h[1], h[2], h[3],
         h[68], h[69], h[70], h[71], h[140], h[141], h[142], h[143]);
  return 0;
}
