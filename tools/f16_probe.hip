// Dev probe: is a 2-piece fp16 split (3 MFMA passes) an fp32-equivalent product on gfx950?
//   (1) does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs,
//   (2) error of  a1 b1 + a1 b2 + a2 b1  (fp16 pieces, per-tensor power-of-two scale) against fp64, next to the
//       6-pass bf16 split and the exact-fp32 MFMA, for plain and wide-dynamic-range data.
// hipcc --offload-arch=gfx950 -O3 tools/f16_probe.hip -o /tmp/f16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ inline int row_of(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// C[32x32] = A[32xK] * B[Kx32]; A row-major [32][K], B given transposed Bt[32][K]
__global__ void probe(const float* A, const float* Bt, int K, float sa, float sb, float* Cf16, float* Cbf, float* Cf32,
                      int four) {
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  f16v acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) acc1[r] = acc2[r] = acc3[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    h8 a1, a2, b1, b2;
    b8 x1, x2, x3, y1, y2, y3;
    for (int j = 0; j < 8; ++j) {
      const float a = A[li * K + k0 + 8 * lh + j], b = Bt[li * K + k0 + 8 * lh + j];
      const float as = a * sa, bs = b * sb;
      a1[j] = (_Float16)as; a2[j] = (_Float16)(as - (float)a1[j]);
      b1[j] = (_Float16)bs; b2[j] = (_Float16)(bs - (float)b1[j]);
      x1[j] = (__bf16)a; float r = a - (float)x1[j]; x2[j] = (__bf16)r; x3[j] = (__bf16)(r - (float)x2[j]);
      y1[j] = (__bf16)b; r = b - (float)y1[j]; y2[j] = (__bf16)r; y3[j] = (__bf16)(r - (float)y2[j]);
    }
    if (four) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, y3, acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, y1, acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, y2, acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, y2, acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, y1, acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, y1, acc2, 0, 0, 0);
    for (int kk = 0; kk < 16; kk += 2) {
      const float a = A[li * K + k0 + kk + lh], b = Bt[li * K + k0 + kk + lh];
      acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
    }
  }
  const float inv = 1.f / (sa * sb);
  for (int r = 0; r < 16; ++r) {
    const int row = row_of(r, lane);
    Cf16[row * 32 + li] = acc1[r] * inv;
    Cbf[row * 32 + li] = acc2[r];
    Cf32[row * 32 + li] = acc3[r];
  }
}

// raw subnormal test: A = fp16 subnormal 2^-20, B = 1 -> expect K * 2^-20 if subnormals are honoured
__global__ void denorm(float* out) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1.f; }
  f16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
  // subnormal x large: 2^-24 (smallest subnormal) * 2^10
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)5.9604644775390625e-08f; b[j] = (_Float16)1024.f; }
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[2] = acc[0];
}

static double frand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(frand())) * cos(6.283185307179586 * frand()); }

static float pow2_scale(float mx) {   // max|v * s| in [2^13, 2^14)
  if (!(mx > 0.f)) return 1.f;
  int e;
  frexpf(mx, &e);                      // mx = m * 2^e, m in [0.5,1)
  return ldexpf(1.f, 14 - e);
}

int main() {
  float* dout; hipMalloc(&dout, 16);
  denorm<<<1, 64>>>(dout);
  float h[3]; hipMemcpy(h, dout, 12, hipMemcpyDeviceToHost);
  printf("subnormal: 16*2^-20 = %.9g  got %.9g  (a as float %.9g);  16*2^-24*1024 = %.9g got %.9g\n", 16 * 9.5367431640625e-07,
         h[0], h[1], 16 * 5.9604644775390625e-08 * 1024, h[2]);
  for (int mode = 0; mode < 4; ++mode) {
    for (int K : {144, 1152, 2304}) {
      std::vector<float> A(32 * K), Bt(32 * K);
      float ma = 0, mb = 0;
      for (int i = 0; i < 32 * K; ++i) {
        double a = nrand(), b = nrand() * 0.05;
        if (mode == 1) { a *= exp(-16.0 * frand()); b *= exp(-16.0 * frand()); }           // element-wise 7 decades
        if (mode == 2) { a *= exp(-14.0 * ((i / K) % 4)); }                                  // row-wise ("per image") range
        if (mode == 3) { a = fabs(a) + 3.0; b = fabs(b) + 0.1; }                             // no cancellation
        A[i] = (float)a; Bt[i] = (float)b;
        ma = fmaxf(ma, fabsf(A[i])); mb = fmaxf(mb, fabsf(Bt[i]));
      }
      float *dA, *dB, *c1, *c2, *c3;
      hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, A.size() * 4);
      hipMalloc(&c1, 4096); hipMalloc(&c2, 4096); hipMalloc(&c3, 4096);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(dB, Bt.data(), A.size() * 4, hipMemcpyHostToDevice);
      for (int four = 0; four < 2; ++four) {
        probe<<<1, 64>>>(dA, dB, K, pow2_scale(ma), pow2_scale(mb), c1, c2, c3, four);
        std::vector<float> r1(1024), r2(1024), r3(1024);
        hipMemcpy(r1.data(), c1, 4096, hipMemcpyDeviceToHost);
        hipMemcpy(r2.data(), c2, 4096, hipMemcpyDeviceToHost);
        hipMemcpy(r3.data(), c3, 4096, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0, e3 = 0, w1 = 0, w2 = 0, w3 = 0;
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 32; ++j) {
            double ref = 0, mag = 0;
            for (int k = 0; k < K; ++k) { const double t = (double)A[i * K + k] * Bt[j * K + k]; ref += t; mag += fabs(t); }
            const double d1 = fabs(r1[i * 32 + j] - ref) / mag, d2 = fabs(r2[i * 32 + j] - ref) / mag,
                         d3 = fabs(r3[i * 32 + j] - ref) / mag;
            e1 += d1; e2 += d2; e3 += d3;
            w1 = fmax(w1, d1); w2 = fmax(w2, d2); w3 = fmax(w3, d3);
          }
        printf("mode %d K %4d f16x%d: mean/max err / sum|ab|   f16 %.3e %.3e   bf16x6 %.3e %.3e   f32 %.3e %.3e\n", mode, K,
               3 + four, e1 / 1024, w1, e2 / 1024, w2, e3 / 1024, w3);
      }
    }
  }
  return 0;
}
