// Dev probe: raw buffer load/store semantics (b128 load, out-of-range behaviour) with the resource word used here.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* in, int* out, int nbytes) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
  const unsigned off = threadIdx.x == 5 ? 0xffffffffu : threadIdx.x * 16u;
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  int h[64 * 4], *din, *dout;
  for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
  hipMalloc((void**)&din, 1024); hipMalloc((void**)&dout, 1024);
  hipMemcpy(din, h, 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>(din, dout, 1024 - 64);   // last 4 threads out of range
  int o[256]; hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
  for (int t : {0, 1, 2, 5, 58, 59, 60, 63}) printf("t%d: %d %d %d %d\n", t, o[t * 4], o[t * 4 + 1], o[t * 4 + 2], o[t * 4 + 3]);
  return 0;
}
