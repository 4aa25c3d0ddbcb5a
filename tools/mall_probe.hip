// Dev probe (round 4): does a tensor that the PREVIOUS kernel wrote come out of the 256 MB Infinity
// Cache when the next kernel reads it, and what do the store / load policy and the order in which the consumer walks the
// tensor change?  Producer: reads A (and optionally A2), writes T.  Consumer: reads T, writes U (1 read + 1 write, the
// GroupNorm forward mix).  Everything else is flushed out of the caches (1 GiB rewrite) before every producer launch, so
// only what the producer leaves behind counts.
// Build / run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe tools/mall_probe.hip && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CHUNK = 16384;   // float4 per block: 256 KB, 64 float4 per thread

template <int ST, int NREAD>   // ST: 0 plain stores, 1 nt stores
__global__ __launch_bounds__(256) void prod(const f32x4* __restrict__ a, const f32x4* __restrict__ a2, f32x4* __restrict__ t) {
  const size_t base = (size_t)blockIdx.x * CHUNK + threadIdx.x;
#pragma unroll 8
  for (int i = 0; i < CHUNK / 256; ++i) {
    f32x4 v = __builtin_nontemporal_load(a + base + i * 256);
    if (NREAD == 2) v += __builtin_nontemporal_load(a2 + base + i * 256);
    v *= 1.0001f;
    if (ST == 1) __builtin_nontemporal_store(v, t + base + i * 256);
    else t[base + i * 256] = v;
  }
}

template <int LD, int STU>     // LD: 0 plain loads, 1 nt loads; STU: policy of the consumer's own output
__global__ __launch_bounds__(256) void cons(const f32x4* __restrict__ t, f32x4* __restrict__ u, int rev) {
  const int blk = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const size_t base = (size_t)blk * CHUNK + threadIdx.x;
#pragma unroll 8
  for (int i = 0; i < CHUNK / 256; ++i) {
    f32x4 v = LD == 1 ? __builtin_nontemporal_load(t + base + i * 256) : t[base + i * 256];
    v = v * v + 1.f;
    if (STU == 1) __builtin_nontemporal_store(v, u + base + i * 256);
    else u[base + i * 256] = v;
  }
}

__global__ void flush_kernel(f32x4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.f;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const size_t flush_n = (size_t)1 << 26;                 // 1 GiB of float4
  f32x4* fl; CK(hipMalloc(&fl, flush_n * 16)); CK(hipMemset(fl, 0, flush_n * 16));
  for (int mb : {67, 134}) {
    const size_t n4 = (size_t)mb * 1024 * 1024 / 16 / CHUNK * CHUNK;
    const int blocks = (int)(n4 / CHUNK);
    f32x4 *a, *a2, *t, *u;
    CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&a2, n4 * 16)); CK(hipMalloc(&t, n4 * 16)); CK(hipMalloc(&u, n4 * 16));
    CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(a2, 0, n4 * 16)); CK(hipMemset(t, 0, n4 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int pst, int nread, int ld, int stu, int rev, int mode) {
      // mode 0: flush -> producer -> consumer (timed); 1: flush -> consumer (cold); 2: consumer -> consumer (read-warm)
      std::vector<float> ts;
      for (int it = 0; it < 9; ++it) {
        flush_kernel<<<2048, 256>>>(fl, flush_n);
        if (mode == 0) {
          if (pst == 0 && nread == 1) prod<0, 1><<<blocks, 256>>>(a, a2, t);
          if (pst == 1 && nread == 1) prod<1, 1><<<blocks, 256>>>(a, a2, t);
          if (pst == 0 && nread == 2) prod<0, 2><<<blocks, 256>>>(a, a2, t);
          if (pst == 1 && nread == 2) prod<1, 2><<<blocks, 256>>>(a, a2, t);
        }
        if (mode == 2) cons<0, 0><<<blocks, 256>>>(t, u, 0);
        hipEventRecord(e0);
        if (ld == 0 && stu == 0) cons<0, 0><<<blocks, 256>>>(t, u, rev);
        if (ld == 1 && stu == 0) cons<1, 0><<<blocks, 256>>>(t, u, rev);
        if (ld == 0 && stu == 1) cons<0, 1><<<blocks, 256>>>(t, u, rev);
        if (ld == 1 && stu == 1) cons<1, 1><<<blocks, 256>>>(t, u, rev);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f);
      }
      std::sort(ts.begin(), ts.end());
      const float us = ts[ts.size() / 2];
      printf("T=%3d MB  %-58s %7.1f us  %5.2f TB/s\n", mb, name, us, 2.0 * n4 * 16 / us * 1e-6);
      fflush(stdout);
    };
    run("consumer cold (flush in front)", 0, 1, 0, 0, 0, 1);
    run("consumer cold, nt loads", 0, 1, 1, 0, 0, 1);
    run("consumer cold, nt loads + nt stores", 0, 1, 1, 1, 0, 1);
    run("consumer read-warm (the same launch in front)", 0, 1, 0, 0, 0, 2);
    run("producer(plain st) -> consumer", 0, 1, 0, 0, 0, 0);
    run("producer(plain st) -> consumer reversed", 0, 1, 0, 0, 1, 0);
    run("producer(nt st)    -> consumer", 1, 1, 0, 0, 0, 0);
    run("producer(nt st)    -> consumer reversed", 1, 1, 0, 0, 1, 0);
    run("producer(plain st) -> consumer nt loads", 0, 1, 1, 0, 0, 0);
    run("producer(plain st) -> consumer nt loads + nt stores", 0, 1, 1, 1, 0, 0);
    run("producer(plain st) -> consumer reversed, nt ld + nt st", 0, 1, 1, 1, 1, 0);
    run("producer(2 reads, plain st) -> consumer", 0, 2, 0, 0, 0, 0);
    run("producer(2 reads, plain st) -> consumer reversed", 0, 2, 0, 0, 1, 0);
    run("producer(2 reads, nt st)    -> consumer reversed", 1, 2, 0, 0, 1, 0);
    hipFree(a); hipFree(a2); hipFree(t); hipFree(u);
  }
  return 0;
}
