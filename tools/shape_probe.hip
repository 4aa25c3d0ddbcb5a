// Stand-alone probe (dev tool, not part of the library).  Two questions for the f16x3 convolution kernels:
//  (1) MI355X_MICROARCH.md "DVFS give-back" item 7: does the chip hold a higher clock on v_mfma_f32_16x16x32_f16 than on
//      v_mfma_f32_32x32x16_f16 at equal FLOP per cycle (random operands, 96 accumulator registers per wave)?
//  (2) two co-resident 256-thread blocks per CU (two waves per SIMD) that both issue MFMAs back to back: does the matrix
//      pipe alternate between them (both finish together) or serve the older one first; what does s_setprio change;
//      which blocks share a CU (HW_ID)?
// hipcc --offload-arch=gfx950 -O3 tools/shape_probe.hip -o /tmp/shape_probe && /tmp/shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// One "round" = 12 x 32x32x16 (SHAPE 0, 6 accumulators used twice) or 24 x 16x16x32 (SHAPE 1, 24 accumulators): the
// same FLOP (12 x 32768 MAC) and the same nominal matrix-pipe cycles (384).  LDSR 1: all 8 operand fragments are
// re-read from LDS every round, software pipelined (the next round's reads are issued before this round's MFMAs).
// PRIO 1: blocks whose s_memrealtime-ordered arrival on their CU is second (tracked through a per-CU counter) raise
// their priority; PRIO 2: the first arrivals do.
template <int SHAPE, int LDSR, int PRIO, int WPS>
__global__ __launch_bounds__(256, WPS) void probe(const _Float16* __restrict__ src, float* out, unsigned long long* tl,
                                                  unsigned* cu_ctr, int rounds) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 256) lds[i] = src[(blockIdx.x * 8192 + i) & 0xfffff];
  __syncthreads();
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned cu = ((xcc & 15u) << 8) | (((hwid >> 13) & 7u) << 5) | (((hwid >> 12) & 1u) << 4) | ((hwid >> 8) & 15u);
  __shared__ unsigned arrival;
  if (tid == 0) arrival = atomicAdd(cu_ctr + cu, 1u);
  __syncthreads();
  const unsigned arr = __builtin_amdgcn_readfirstlane(arrival);
  if (PRIO == 1 && (arr & 1u)) __builtin_amdgcn_s_setprio(2);
  if (PRIO == 2 && !(arr & 1u)) __builtin_amdgcn_s_setprio(2);
  f16x8 a[4], b[4], an[4], bn[4];
  auto rd = [&](f16x8 (&av)[4], f16x8 (&bv)[4], int it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      av[i] = *reinterpret_cast<const f16x8*>(lds + ((tid * 8 + i * 2048 + it * 64) & 8191));
      bv[i] = *reinterpret_cast<const f16x8*>(lds + ((tid * 8 + i * 2048 + 1024 + it * 64) & 8191));
    }
  };
  rd(a, b, 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < rounds; ++it) {
      if (LDSR) rd(an, bn, it + 1);
#pragma unroll
      for (int t = 0; t < 12; ++t)
        acc[t % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t & 3], b[(t >> 2) + (t & 1)], acc[t % 6], 0, 0, 0);
      if (LDSR) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = an[i]; b[i] = bn[i]; }
      }
    }
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[t][r];
  } else {
    f32x4 acc[24];
#pragma unroll
    for (int t = 0; t < 24; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < rounds; ++it) {
      if (LDSR) rd(an, bn, it + 1);
#pragma unroll
      for (int t = 0; t < 24; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t & 3], b[((t >> 2) + (t & 1)) & 3], acc[t], 0, 0, 0);
      if (LDSR) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = an[i]; b[i] = bn[i]; }
      }
    }
#pragma unroll
    for (int t = 0; t < 24; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[t][r];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) {
    tl[blockIdx.x * 4] = t0; tl[blockIdx.x * 4 + 1] = t1; tl[blockIdx.x * 4 + 2] = c1 - c0;
    tl[blockIdx.x * 4 + 3] = ((unsigned long long)cu << 8) | arr;
  }
}

static _Float16* g_src;
static float* g_out;
static unsigned long long* g_tl;
static unsigned* g_ctr;

template <int SHAPE, int LDSR, int PRIO, int WPS>
void run(const char* name, int blocks, int rounds) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  auto launch = [&]() {
    hipMemsetAsync(g_ctr, 0, 4096 * 4, 0);
    hipLaunchKernelGGL((probe<SHAPE, LDSR, PRIO, WPS>), dim3(blocks), dim3(256), 0, 0, g_src, g_out, g_tl, g_ctr, rounds);
  };
  for (int r = 0; r < 60; ++r) launch();
  hipDeviceSynchronize();
  const int reps = 60;
  hipEventRecord(s);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double t = ms * 1e-3 / reps;
  const double flops = (double)blocks * 4 * rounds * 12.0 * 32 * 32 * 16 * 2;
  std::vector<unsigned long long> tl(blocks * 4);
  hipMemcpy(tl.data(), g_tl, blocks * 4 * 8, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull;
  for (int b = 0; b < blocks; ++b) tmin = std::min(tmin, tl[b * 4]);
  std::vector<double> dur, clk, d_first, d_second;
  std::map<unsigned, int> per_cu;
  int same_parity_pairs = 0, idx_delta_256 = 0;
  std::map<unsigned, std::vector<int>> members;
  for (int b = 0; b < blocks; ++b) {
    const double d = (tl[b * 4 + 1] - tl[b * 4]) * 0.01;
    dur.push_back(d);
    clk.push_back((double)tl[b * 4 + 2] / ((tl[b * 4 + 1] - tl[b * 4]) * 10.0));
    const unsigned cu = (unsigned)(tl[b * 4 + 3] >> 8), arr = (unsigned)(tl[b * 4 + 3] & 255);
    per_cu[cu]++;
    members[cu].push_back(b);
    ((arr & 1) ? d_second : d_first).push_back(d);
  }
  for (auto& kv : members)
    if (kv.second.size() >= 2) {
      if (((kv.second[0] / 256) & 1) == ((kv.second[1] / 256) & 1)) same_parity_pairs++;
      if (std::abs(kv.second[1] - kv.second[0]) == 256) idx_delta_256++;
    }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  std::sort(dur.begin(), dur.end()); std::sort(clk.begin(), clk.end());
  printf("%-40s blk=%4d: %7.1f us %7.1f TF | blk us min/med/max %6.1f %6.1f %6.1f | first/second arrival med %6.1f %6.1f | "
         "CUs %3zu, pairs idx+256: %3d | clk %.3f GHz\n",
         name, blocks, t * 1e6, flops / t / 1e12, dur[0], dur[blocks / 2], dur[blocks - 1], med(d_first), med(d_second),
         per_cu.size(), idx_delta_256, clk[blocks / 2]);
}

int main() {
  std::vector<_Float16> h(1 << 20);
  srand(1);
  for (auto& v : h) {   // roughly gaussian, full sign / exponent spread like split activations
    float u = 0.f;
    for (int i = 0; i < 6; ++i) u += (float)rand() / RAND_MAX - 0.5f;
    v = (_Float16)(u * 300.f);
  }
  hipMalloc(&g_src, h.size() * 2);
  hipMemcpy(g_src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&g_out, 2048 * 256 * 4);
  hipMalloc(&g_tl, 2048 * 4 * 8);
  hipMalloc(&g_ctr, 4096 * 4);
  const int R = 1500;
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0, 0, 1>("32x32x16 regs, 1 wave/SIMD", 256, R);
    run<1, 0, 0, 1>("16x16x32 regs, 1 wave/SIMD", 256, R);
    run<0, 1, 0, 1>("32x32x16 LDS piped, 1 wave/SIMD", 256, R);
    run<1, 1, 0, 1>("16x16x32 LDS piped, 1 wave/SIMD", 256, R);
    run<0, 1, 0, 2>("32x32x16 LDS piped, 2 blocks/CU", 512, R);
    run<1, 1, 0, 2>("16x16x32 LDS piped, 2 blocks/CU", 512, R);
    run<0, 1, 1, 2>("32x32x16 LDS piped, 2 blk/CU, 2nd prio", 512, R);
    run<0, 1, 2, 2>("32x32x16 LDS piped, 2 blk/CU, 1st prio", 512, R);
    run<0, 0, 1, 2>("32x32x16 regs, 2 blk/CU, 2nd prio", 512, R);
  }
  return 0;
}
