// Dev probe (round 4): does a buffer load with an LDS destination (buffer_load_dwordx4 ... lds) write ZEROS to LDS for lanes
// whose offset is out of the buffer's range (as a register load returns zeros), or does it leave the LDS bytes as they were?
// The answer decides how the weight-gradient kernel's LDS-DMA staging pads the rows above / below the image.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_oob_probe tools/dma_oob_probe.hip && /tmp/dma_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned char* __restrict__ g, unsigned* out, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_p;
  for (int i = threadIdx.x; i < 2048 / 4; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(g), 0, n, 0x00020000);
  const int lane = threadIdx.x & 63;
  // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_p)(smem + 1024), 16, (lane & 1) ? 0xfffffff0u : lane * 16, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 2048 / 4; i += 64) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}
int main() {
  unsigned char* g; unsigned* out;
  hipMalloc(&g, 4096); hipMalloc(&out, 2048);
  hipMemset(g, 0x11, 4096);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 2048, 0, g, out, 4096);
  unsigned h[512];
  hipMemcpy(h, out, 2048, hipMemcpyDeviceToHost);
  printf("before the DMA area: %08x; lane 0 (in range): %08x %08x; lane 1 (out of range): %08x %08x; lane 2: %08x; lane 3 (oob): %08x\n",
         h[255], h[256], h[259], h[260], h[263], h[264], h[268]);
  printf("out-of-range lanes write %s\n", h[260] == 0 ? "ZEROS" : (h[260] == 0xdeadbeefu ? "NOTHING (LDS keeps its bytes)" : "something else"));
  return 0;
}
