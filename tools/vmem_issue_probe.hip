// Dev probe (round 6): what does one vector-memory instruction cost a wave whose stream is otherwise MFMAs?
// The eight-wave weight-gradient block lost ~600 of 5650 cycles per row pair to 16 loads per SIMD -- and as much with 4 bytes
// per lane as with 16, and with every load hitting L2 (profiles/r06_w8_ablation.log): the price is per INSTRUCTION.  This
// probe measures it in isolation: per iteration G MFMAs (independent accumulators) + one load of the given form from a
// 256 KB L2-resident window, 1 or 2 waves per SIMD, every CU busy; cycles per iteration minus the load-free loop.
//   hipcc --offload-arch=gfx950 -O3 tools/vmem_issue_probe.hip -o tools/bin/vmem_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i4 __attribute__((ext_vector_type(4)));

// KIND 0 none, 1 buffer_load_dwordx4 offen, 2 buffer_load_dwordx4 offen + soffset, 3 global_load_dwordx4 (64-bit vaddr),
// 4 global_load_dwordx4 saddr + 32-bit voffset, 5 buffer_load_dword, 6 global_load_lds_dwordx4 (LDS-DMA),
// 7 buffer_load_dwordx4 ... lds (LDS-DMA), 8 ds_read_b128 (for scale), 9 ds_write_b128
template <int KIND, int G, bool M16, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(unsigned long long* out, float* sink, const float* src, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(tid * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  f16v acc32[4];
  f4 acc16[8];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc32[t][r] = 0.f;
  for (int t = 0; t < 8; ++t) acc16[t] = f4{0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 20, 0x00020000);
  i4 v[4];
  for (int k = 0; k < 4; ++k) v[k] = i4{0, 0, 0, 0};
  unsigned voff = (unsigned)tid * 16u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it4 = 0; it4 < iters; it4 += 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int it = it4 + q;
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (M16) acc16[u & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc16[u & 7], 0, 0, 0);
        else acc32[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc32[u & 3], 0, 0, 0);
      }
      const unsigned o = (voff + (unsigned)(it & 15) * 16384u) & 0x3fff0u;
      const unsigned so = (unsigned)(it & 7) * 1024u;
      if (KIND == 1) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
      if (KIND == 2) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0);
      if (KIND == 3) {
        const f4* p = reinterpret_cast<const f4*>(reinterpret_cast<const unsigned char*>(src) + o);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[q]) : "v"(p) : "memory");
      }
      if (KIND == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[q]) : "v"(o), "s"(src) : "memory");
      if (KIND == 5) v[q][0] = __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0);
      if (KIND == 6) {
        typedef __attribute__((address_space(3))) void* lds_p;
        typedef const __attribute__((address_space(1))) void* gbl_p;
        __builtin_amdgcn_global_load_lds((gbl_p)(reinterpret_cast<const unsigned char*>(src) + o),
                                         (lds_p)(lds + (tid >> 6) * 4096 + q * 1024), 16, 0, 0);
      }
      if (KIND == 7) {
        typedef __attribute__((address_space(3))) void* lds_p;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_p)(lds + (tid >> 6) * 4096 + q * 1024), 16, o, 0, 0, 0);
      }
      if (KIND == 8) v[q] = *reinterpret_cast<const i4*>(lds + ((tid * 16 + q * 1024) & 0x7ff0));
      if (KIND == 9) *reinterpret_cast<i4*>(lds + ((tid * 16 + q * 1024) & 0x7ff0)) = v[q];
      if ((KIND == 3 || KIND == 4) && q == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (asm loads: the compiler does not count them)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc32[t][r];
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) s += acc16[t][r];
  for (int k = 0; k < 4; ++k) s += (float)(v[k][0] + v[k][1] + v[k][2] + v[k][3]);
  sink[blockIdx.x * WAVES * 64 + tid] = s + lds[tid];
  if (blockIdx.x == 0 && tid == 0) out[0] = t1 - t0;
}

template <int KIND, int G, bool M16, int WAVES>
double run(unsigned long long* d, float* sink, const float* src) {
  const int iters = 4000;
  for (int r = 0; r < 2; ++r) probe<KIND, G, M16, WAVES><<<256, WAVES * 64, 40960>>>(d, sink, src, iters);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  return (double)h / iters;
}

template <int G, bool M16, int WAVES>
void table(unsigned long long* d, float* sink, const float* src) {
  const double base = run<0, G, M16, WAVES>(d, sink, src);
  printf("%s, %d MFMAs + 1 load per iteration, %d waves per SIMD: load-free loop %.0f cycles (%.1f per MFMA per wave)\n",
         M16 ? "16x16x32" : "32x32x16", G, WAVES / 4, base, base / G);
  const char* names[] = {"", "buffer_load_dwordx4 offen", "buffer_load_dwordx4 offen+soffset", "global_load_dwordx4 vaddr64",
                         "global_load_dwordx4 saddr", "buffer_load_dword offen", "global_load_lds_dwordx4", "buffer_load_dwordx4 lds",
                         "ds_read_b128", "ds_write_b128"};
  const double r[] = {0, run<1, G, M16, WAVES>(d, sink, src), run<2, G, M16, WAVES>(d, sink, src), run<3, G, M16, WAVES>(d, sink, src),
                      run<4, G, M16, WAVES>(d, sink, src), run<5, G, M16, WAVES>(d, sink, src), run<6, G, M16, WAVES>(d, sink, src),
                      run<7, G, M16, WAVES>(d, sink, src), run<8, G, M16, WAVES>(d, sink, src), run<9, G, M16, WAVES>(d, sink, src)};
  for (int k = 1; k <= 9; ++k) printf("    %-36s +%6.1f cycles per instruction\n", names[k], r[k] - base);
  fflush(stdout);
}

int main() {
  unsigned long long* d;
  hipMalloc((void**)&d, 64);
  float *sink, *src;
  hipMalloc((void**)&sink, 256 * 512 * 4);
  hipMalloc((void**)&src, 1 << 20);
  hipMemset(src, 0, 1 << 20);
  table<8, false, 4>(d, sink, src);
  table<8, false, 8>(d, sink, src);
  table<16, true, 4>(d, sink, src);
  table<16, true, 8>(d, sink, src);
  table<4, true, 8>(d, sink, src);
  return 0;
}
