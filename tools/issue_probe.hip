// Dev probe: how many VALU / LDS instructions of the same wave fit into the shadow of one v_mfma_f32_32x32x16_f16
// (1 wave per SIMD, 4 waves per CU, all CUs busy)?  Prints cycles per MFMA for n filler instructions per MFMA.
// hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/bin/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int N, int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, float* sink, int iters) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(tid * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  f16v acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float f[8];
  for (int j = 0; j < 8; ++j) f[j] = tid + j;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[n & 7]) : "v"(f[(n + 1) & 7]));
        if (KIND == 1) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(f[n & 7]) : "v"(f[(n + 1) & 7]));
        if (KIND == 2) asm volatile("ds_write_b64 %0, %1" :: "v"(tid * 8), "v"(*(double*)&f[(n & 3) * 2]) : "memory");
        if (KIND == 3) asm volatile("v_fma_mixlo_f16 %0, %1, %1, 0" : "+v"(f[n & 7]) : "v"(f[(n + 1) & 7]));
        if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[n & 7]) : "v"(f[(n + 1) & 7]));
        if (KIND == 5) asm volatile("s_nop 0");
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int j = 0; j < 8; ++j) s += f[j];
  sink[blockIdx.x * 256 + tid] = s + lds[tid];
  if (blockIdx.x == 0 && tid == 0) out[0] = t1 - t0;
}

template <int N, int KIND>
void run(unsigned long long* d, float* sink, const char* name) {
  const int iters = 2000;
  probe<N, KIND><<<256, 256>>>(d, sink, iters);
  probe<N, KIND><<<256, 256>>>(d, sink, iters);
  hipDeviceSynchronize();
  unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-18s n=%2d: %.1f cycles per MFMA\n", name, N, (double)h / (iters * 8.0));
}

int main() {
  unsigned long long* d; hipMalloc((void**)&d, 64);
  float* sink; hipMalloc((void**)&sink, 256 * 256 * 4);
  run<0, 0>(d, sink, "none");
  run<2, 0>(d, sink, "v_fma_f32"); run<4, 0>(d, sink, "v_fma_f32"); run<6, 0>(d, sink, "v_fma_f32"); run<8, 0>(d, sink, "v_fma_f32");
  run<12, 0>(d, sink, "v_fma_f32"); run<16, 0>(d, sink, "v_fma_f32");
  run<2, 1>(d, sink, "v_cvt_pk_f16_f32"); run<4, 1>(d, sink, "v_cvt_pk_f16_f32"); run<8, 1>(d, sink, "v_cvt_pk_f16_f32");
  run<1, 2>(d, sink, "ds_write_b64"); run<2, 2>(d, sink, "ds_write_b64"); run<4, 2>(d, sink, "ds_write_b64");
  run<2, 3>(d, sink, "v_fma_mixlo_f16"); run<4, 3>(d, sink, "v_fma_mixlo_f16"); run<8, 3>(d, sink, "v_fma_mixlo_f16");
  run<4, 4>(d, sink, "v_cndmask_b32"); run<8, 4>(d, sink, "v_cndmask_b32");
  run<4, 5>(d, sink, "s_nop"); run<8, 5>(d, sink, "s_nop");
  return 0;
}
