// Dev probe: issue throughput of ds_read_b64_tr_b16 vs ds_read_b64 / ds_read_b128 (4 waves per CU, all CUs busy).
// hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o tools/bin/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 256) reinterpret_cast<int*>(lds)[i] = i;
  __syncthreads();
  const int lh = lane >> 5, q = (lane & 15) >> 2, pp = lane & 3, cb = ((lane >> 4) & 1) * 16;
  const int off_tr = (8 * lh + q) * 64 + (cb + 4 * pp) * 2 + (tid >> 6) * 8192;
  const int off_b64 = lane * 8 + (tid >> 6) * 8192;
  const int off_b128 = lane * 16 + (tid >> 6) * 8192;
  int acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) {
        typedef __attribute__((address_space(3))) s16x4* lp;
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + off_tr + u * 1024 % 8192));
        acc += v[0] + v[3];
      } else if (MODE == 1) {
        const i32x2 v = *reinterpret_cast<const i32x2*>(lds + off_b64 + (u * 512) % 8192);
        acc += v[0] + v[1];
      } else {
        const i32x4 v = *reinterpret_cast<const i32x4*>(lds + off_b128 + (u * 1024) % 8192);
        acc += v[0] + v[3];
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc == 0x7fffffff) out[1] = acc;
  if (blockIdx.x == 0 && tid == 0) out[0] = t1 - t0;
}

int main() {
  unsigned long long* d; hipMalloc((void**)&d, 64);
  const int iters = 1000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) probe<0><<<256, 256>>>(d, iters);
      if (mode == 1) probe<1><<<256, 256>>>(d, iters);
      if (mode == 2) probe<2><<<256, 256>>>(d, iters);
      hipDeviceSynchronize();
    }
    unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / (iters * 16.0);
    const int bytes = mode == 2 ? 1024 : 512;
    printf("mode %d (%s): %.2f cycles per wave-instruction with 4 waves/CU -> %.1f B/clk/CU\n", mode,
           mode == 0 ? "ds_read_b64_tr_b16" : mode == 1 ? "ds_read_b64" : "ds_read_b128", per, 4.0 * bytes / per);
  }
  return 0;
}
