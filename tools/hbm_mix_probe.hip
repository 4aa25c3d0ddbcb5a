// Dev probe (round 5): what do read + write streams of the GroupNorm shapes sustain on this part, cold and warm?
// VERDICT r04 item 1 prices the GroupNorm kernels against tools/thin_stream_probe.hip (6.1-6.7 TB/s for 2 reads + 1 write
// of 201 MB).  That probe runs on ONE set of zero-filled buffers, back to back: 201 MB fit the 256 MB Infinity Cache.  Here
// the same streams run on `sets` rotating buffer sets of random data (sets x 201 MB >> 256 MB: every byte from / to HBM)
// and on one set (cache-warm), for four access shapes:
//   contig : grid-stride, every wave instruction = 1 KB contiguous (thin_stream_probe's shape)
//   slab   : the GroupNorm mapping -- block (b, 32-channel slab), wave instruction = 8 pixels x 128 B, 512 B apart
//   planes : reads as slab, writes as the split planes ([b][c/16][px][plane][16] fp16: two 8-byte pieces per lane)
// Build / run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_mix_probe tools/hbm_mix_probe.hip && /tmp/hbm_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int HW = 1024, C = 128, B = 128;

template <int NR>
__global__ __launch_bounds__(256) void contig(const f32x4* __restrict__ x, const f32x4* __restrict__ dy, f32x4* __restrict__ dx, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n4; i += 4 * stride) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = __builtin_nontemporal_load(x + i + u * stride);
      if (NR == 2) b[u] = __builtin_nontemporal_load(dy + i + u * stride);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) dx[i + u * stride] = NR == 2 ? a[u] * 0.5f + b[u] : a[u] * 0.5f;
  }
}

// block (b, slab of 32 channels, quarter sp): thread (prow, quad) owns pixels prow + 32 (8 sp + i), i = 0..7
template <int NR, bool PLANES>
__global__ __launch_bounds__(256) void slab(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx) {
  const int tid = threadIdx.x, quad = tid & 7, prow = tid >> 3;
  const int b = blockIdx.x, c0 = blockIdx.y * 32, sp = blockIdx.z;
  const size_t base = (size_t)b * HW * C + c0 + quad * 4;
  const int px0 = prow + 256 * sp;
  f32x4 a[8], g[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + base + (size_t)(px0 + 32 * i) * C));
    if (NR == 2) g[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dy + base + (size_t)(px0 + 32 * i) * C));
  }
  const int c = c0 + quad * 4;
  unsigned char* pdst = reinterpret_cast<unsigned char*>(dx) + ((size_t)(b * (C >> 4) + (c >> 4)) * HW) * 64 + (c & 15) * 2;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 o = NR == 2 ? a[i] * 0.5f + g[i] : a[i] * 0.5f;
    if (PLANES) {
      *reinterpret_cast<f32x2*>(pdst + (size_t)(px0 + 32 * i) * 64) = f32x2{o[0], o[1]};
      *reinterpret_cast<f32x2*>(pdst + (size_t)(px0 + 32 * i) * 64 + 32) = f32x2{o[2], o[3]};
    } else {
      *reinterpret_cast<f32x4*>(dx + base + (size_t)(px0 + 32 * i) * C) = o;
    }
  }
}

// the same work per thread, but lanes run along the channels: wave instruction = 2 pixels x 512 B = 1 KB contiguous
template <int NR, bool PLANES>
__global__ __launch_bounds__(256) void rows(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx) {
  const int tid = threadIdx.x, q = tid & 31, pr = tid >> 5;       // 32 quads x 8 pixels
  const int b = blockIdx.x, blk = blockIdx.y;                      // 16 blocks of 64 pixels per image
  const size_t base = (size_t)b * HW * C + q * 4;
  const int px0 = blk * 64 + pr;
  f32x4 a[8], g[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + base + (size_t)(px0 + 8 * i) * C));
    if (NR == 2) g[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dy + base + (size_t)(px0 + 8 * i) * C));
  }
  const int c = q * 4;
  unsigned char* pdst = reinterpret_cast<unsigned char*>(dx) + ((size_t)(b * (C >> 4) + (c >> 4)) * HW) * 64 + (c & 15) * 2;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 o = NR == 2 ? a[i] * 0.5f + g[i] : a[i] * 0.5f;
    if (PLANES) {
      *reinterpret_cast<f32x2*>(pdst + (size_t)(px0 + 8 * i) * 64) = f32x2{o[0], o[1]};
      *reinterpret_cast<f32x2*>(pdst + (size_t)(px0 + 8 * i) * 64 + 32) = f32x2{o[2], o[3]};
    } else {
      *reinterpret_cast<f32x4*>(dx + base + (size_t)(px0 + 8 * i) * C) = o;
    }
  }
}

int main() {
  const size_t n = (size_t)B * HW * C, n4 = n / 4;
  const int NSET = 8;
  std::vector<float*> xs(NSET), dys(NSET), dxs(NSET);
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((int)(rand() % 2001) - 1000) * 1e-3f;
  for (int s = 0; s < NSET; ++s) {
    hipMalloc(&xs[s], n * 4); hipMalloc(&dys[s], n * 4); hipMalloc(&dxs[s], n * 4);
    hipMemcpy(xs[s], h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dys[s], h.data(), n * 4, hipMemcpyHostToDevice);
  }
  hipEvent_t t0, t1;
  hipEventCreate(&t0); hipEventCreate(&t1);
  auto run = [&](const char* name, int nr, auto launch) {
    for (int sets : {NSET, 1}) {
      std::vector<float> ts;
      for (int rep = 0; rep < 5; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(t0, 0);
        for (int k = 0; k < 24; ++k) launch(xs[k % sets], dys[k % sets], dxs[k % sets]);
        hipEventRecord(t1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, t0, t1);
        ts.push_back(ms * 1e3f / 24);
      }
      std::sort(ts.begin(), ts.end());
      printf("%-34s %s : %6.1f us per launch = %.2f TB/s\n", name, sets > 1 ? "hbm (8 sets)" : "warm (1 set)", ts[2],
             (nr + 1.0) * n * 4 / ts[2] / 1e6);
    }
  };
  for (int blocks : {1024, 2048, 4096}) {
    char nm[64];
    snprintf(nm, sizeof nm, "contig 1R+1W, %d blocks", blocks);
    run(nm, 1, [&](float* x, float* dy, float* dx) { hipLaunchKernelGGL(contig<1>, dim3(blocks), dim3(256), 0, 0, (const f32x4*)x, (const f32x4*)dy, (f32x4*)dx, n4); });
    snprintf(nm, sizeof nm, "contig 2R+1W, %d blocks", blocks);
    run(nm, 2, [&](float* x, float* dy, float* dx) { hipLaunchKernelGGL(contig<2>, dim3(blocks), dim3(256), 0, 0, (const f32x4*)x, (const f32x4*)dy, (f32x4*)dx, n4); });
  }
  run("slab 1R+1W fp32", 1, [&](float* x, float* dy, float* dx) { hipLaunchKernelGGL((slab<1, false>), dim3(B, C / 32, 4), dim3(256), 0, 0, x, dy, dx); });
  run("slab 1R+1W planes", 1, [&](float* x, float* dy, float* dx) { hipLaunchKernelGGL((slab<1, true>), dim3(B, C / 32, 4), dim3(256), 0, 0, x, dy, dx); });
  run("slab 2R+1W fp32", 2, [&](float* x, float* dy, float* dx) { hipLaunchKernelGGL((slab<2, false>), dim3(B, C / 32, 4), dim3(256), 0, 0, x, dy, dx); });
  run("slab 2R+1W planes", 2, [&](float* x, float* dy, float* dx) { hipLaunchKernelGGL((slab<2, true>), dim3(B, C / 32, 4), dim3(256), 0, 0, x, dy, dx); });
  run("rows 1R+1W fp32", 1, [&](float* x, float* dy, float* dx) { hipLaunchKernelGGL((rows<1, false>), dim3(B, 16), dim3(256), 0, 0, x, dy, dx); });
  run("rows 1R+1W planes", 1, [&](float* x, float* dy, float* dx) { hipLaunchKernelGGL((rows<1, true>), dim3(B, 16), dim3(256), 0, 0, x, dy, dx); });
  run("rows 2R+1W fp32", 2, [&](float* x, float* dy, float* dx) { hipLaunchKernelGGL((rows<2, false>), dim3(B, 16), dim3(256), 0, 0, x, dy, dx); });
  run("rows 2R+1W planes", 2, [&](float* x, float* dy, float* dx) { hipLaunchKernelGGL((rows<2, true>), dim3(B, 16), dim3(256), 0, 0, x, dy, dx); });
  // zero-filled buffers (what thin_stream_probe measured)
  for (int s = 0; s < NSET; ++s) { hipMemset(xs[s], 0, n * 4); hipMemset(dys[s], 0, n * 4); }
  run("contig 2R+1W, 1024 blocks, ZEROS", 2, [&](float* x, float* dy, float* dx) { hipLaunchKernelGGL(contig<2>, dim3(1024), dim3(256), 0, 0, (const f32x4*)x, (const f32x4*)dy, (f32x4*)dx, n4); });
  return 0;
}
