// Exact-likelihood ODE evaluator: the probability-flow drift of VDM.reverse_ode
// (ldm/model_mulan_velocity.py:393-421, ldm/model_mulan_epsilon.py:459-478, ldm/model_vdm.py:243-260), the
// Hutchinson divergence term of notebook_utils._get_value_div_fn (:203-215), the dequantisation noise and prior of
// get_ode_likelihood_fn (:316-371) and a Dormand-Prince 5(4) integrator whose state never leaves the device
// (the reference hands every function evaluation to scipy.integrate.solve_ivp on the host, :358).
//
// State layout of the integrator: one double vector  y = [ x (B * 3072) | delta_logp (B) ],  stage derivatives
// K[s] as fp32 vectors of the same length (they are fp32 network outputs; the reference widens them to float64 the
// same way, :193-195).
#include "common.h"

namespace {

// mode & 3: 0 velocity, 1 velocity_from_epsilon, 2 epsilon (MuLAN epsilon model and the plain VDM)
// c = 0.5 alpha sigma gamma'   (velocity):  drift = v c,  v = net | -e^{g/2} x + sqrt(1 + e^g) net
// epsilon:                                  drift = 0.5 (net - sigma x) sigma gamma'
// cot = d(sum drift * hutch) / d net   (the cotangent handed to the U-Net's input-gradient pass)
// mode & 4: reverse_ode(high_precision=True) -- the two selects of ldm/model_mulan_velocity.py:410-417 (alpha and sigma)
// and ldm/model_mulan_epsilon.py:472-475 (sigma): where 1 - sigmoid(g) <= 1e-3 alpha = exp(-g / 2), where
// sigmoid(g) <= 1e-3 sigma = exp(g / 2); the square roots elsewhere
__device__ __forceinline__ void ode_alpha_sigma(float g, bool hp, float& alpha, float& sigma) {
  const float s = sigmoid_f(g);
  sigma = sqrtf(s);
  alpha = sqrtf(sigmoid_f(-g));
  if (hp) {
    if (s <= 1e-3f) sigma = expf(0.5f * g);
    if (1.f - s <= 1e-3f) alpha = expf(-0.5f * g);
  }
}

__global__ void ode_drift_kernel(const float* __restrict__ net, const float* __restrict__ x,
                                 const float* __restrict__ gt, const float* __restrict__ gp,
                                 const float* __restrict__ hutch, float* __restrict__ drift, float* __restrict__ cot,
                                 size_t n, int mode_hp, int g_per_sample) {
  const int mode = mode_hp & 3;
  const bool hp = (mode_hp & 4) != 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t gi = g_per_sample ? i / (size_t)g_per_sample : i;
    const float g = gt[gi], dg = gp[gi];
    float sigma, alpha;
    ode_alpha_sigma(g, hp, alpha, sigma);
    float d, k;      // drift = d, d drift / d net = k
    if (mode == 2) {
      k = 0.5f * sigma * dg;
      d = (net[i] - sigma * x[i]) * k;
    } else {
      const float c = 0.5f * alpha * sigma * dg;
      if (mode == 1) {
        const float s = sqrtf(1.f + expf(g));
        d = (-expf(0.5f * g) * x[i] + s * net[i]) * c;
        k = s * c;
      } else {
        d = net[i] * c;
        k = c;
      }
    }
    drift[i] = d;
    if (cot) cot[i] = hutch[i] * k;
  }
}

// div[b] = sum_i (gx_i + diag_i hutch_i) hutch_i,  diag = explicit d drift_i / d x_i of the closed form around the
// network (0 for mode 0); gx = the U-Net's input gradient for the cotangent above.  One 256-thread block per sample.
__global__ __launch_bounds__(256) void ode_div_kernel(const float* __restrict__ gx, const float* __restrict__ gt,
                                                      const float* __restrict__ gp, const float* __restrict__ hutch,
                                                      float* __restrict__ div, int d, int mode_hp, int g_per_sample) {
  __shared__ float red[4];
  const int mode = mode_hp & 3;
  const bool hp = (mode_hp & 4) != 0;
  const int b = blockIdx.x;
  float s = 0.f;
  for (int j = threadIdx.x; j < d; j += 256) {
    const size_t i = (size_t)b * d + j;
    const float h = hutch[i];
    float diag = 0.f;
    if (mode != 0) {
      const size_t gi = g_per_sample ? (size_t)b : i;
      const float g = gt[gi], dg = gp[gi];
      float sigma, alpha;
      ode_alpha_sigma(g, hp, alpha, sigma);
      // (the plain forms keep the expressions of rounds 3-5: sigma^2 is taken as sigmoid(g) itself)
      if (mode == 2) diag = -0.5f * (hp ? sigma * sigma : sigmoid_f(g)) * dg;
      else diag = -expf(0.5f * g) * 0.5f * alpha * sigma * dg;
    }
    s += (gx[i] + diag * h) * h;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) div[b] = s;
}

struct Coef { double c[7]; };

// out = y + h sum_j coef_j K[j]   (double), optionally also / only as fp32 (the network's input)
__global__ void rk_combine_kernel(const double* __restrict__ y, const float* __restrict__ K, size_t kstride, Coef cf,
                                  int ncoef, double h, double* __restrict__ out, float* __restrict__ out32, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int j = 0; j < ncoef; ++j) acc += cf.c[j] * (double)K[(size_t)j * kstride + i];
    const double v = y[i] + h * acc;
    if (out) out[i] = v;
    if (out32) out32[i] = (float)v;
  }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

constexpr int kNormBlocks = 256;

// partial[blk] = sum_i (h sum_j E_j K[j][i] / (atol + rtol max(|y_i|, |ynew_i|)))^2    (scipy RK45 _estimate_error_norm)
__global__ __launch_bounds__(256) void rk_error_kernel(const double* __restrict__ y, const double* __restrict__ ynew,
                                                       const float* __restrict__ K, size_t kstride, Coef e, double h,
                                                       double rtol, double atol, double* __restrict__ partial,
                                                       size_t n) {
  __shared__ double red[4];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 7; ++j) acc += e.c[j] * (double)K[(size_t)j * kstride + i];
    const double scale = atol + rtol * fmax(fabs(y[i]), fabs(ynew[i]));
    const double r = h * acc / scale;
    s += r * r;
  }
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// the three sums of scipy's select_initial_step: (y0 / scale)^2, (f0 / scale)^2, ((f1 - f0) / scale)^2,
// scale = atol + rtol |y0|;  partial: [3][gridDim.x]
__global__ __launch_bounds__(256) void rk_init_norms_kernel(const double* __restrict__ y0, const float* __restrict__ f0,
                                                            const float* __restrict__ f1, double rtol, double atol,
                                                            double* __restrict__ partial, size_t n) {
  __shared__ double red[4];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double scale = atol + rtol * fabs(y0[i]);
    const double a = y0[i] / scale, b = (double)f0[i] / scale;
    s0 += a * a;
    s1 += b * b;
    if (f1) {
      const double c = ((double)f1[i] - (double)f0[i]) / scale;
      s2 += c * c;
    }
  }
  s0 = block_sum_d(s0, red);
  s1 = block_sum_d(s1, red);
  s2 = block_sum_d(s2, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = s0;
    partial[gridDim.x + blockIdx.x] = s1;
    partial[2 * gridDim.x + blockIdx.x] = s2;
  }
}

// out[r] = sum of partial[r][0:m]  (deterministic order)
__global__ __launch_bounds__(64) void sum_partials_kernel(const double* __restrict__ partial, double* __restrict__ out,
                                                          int m) {
  double s = 0.0;
  for (int j = threadIdx.x; j < m; j += 64) s += partial[(size_t)blockIdx.x * m + j];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// out[r] = -0.5 cols log(2 pi) - 0.5 sum_j x[r][j]^2      (notebook_utils._prior_logp, :218-221)
__global__ __launch_bounds__(256) void normal_logp_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          int cols) {
  __shared__ float red[4];
  float s = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) {
    const float v = x[(size_t)blockIdx.x * cols + j];
    s += v * v;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = -0.5f * (float)cols * 1.8378770664093453f - 0.5f * s;
}

// kind 0: U[0,1);  1: Rademacher +-1;  2: standard normal truncated to [lo, hi] by inverse CDF;  3: standard Gumbel
__global__ void noise_kernel(float* __restrict__ out, size_t n, unsigned long long seed, unsigned long long offset,
                             int kind, float lo, float hi) {
  const size_t n4 = (n + 3) >> 2;
  const float plo = 0.5f * (1.f + erff(lo * 0.70710678118654752f)), phi = 0.5f * (1.f + erff(hi * 0.70710678118654752f));
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
    const Philox4 r = philox4x32_10(seed, offset + q, 0ull);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t i = (q << 2) + e;
      if (i >= n) break;
      float v;
      if (kind == 1) {
        v = (w[e] >> 31) ? 1.f : -1.f;
      } else {
        const float u = (float)(w[e] >> 8) * 5.9604644775390625e-08f;      // 24 bits, [0, 1)
        if (kind == 0) {
          v = u;
        } else if (kind == 3) {
          v = -logf(-logf(u + 2.98023223876953125e-08f));                   // u in (0, 1)
        } else {
          const float p = plo + (phi - plo) * (u + 2.98023223876953125e-08f);
          v = fminf(fmaxf(1.41421356237309505f * erfinvf(2.f * p - 1.f), lo), hi);
        }
      }
      out[i] = v;
    }
  }
}

// data = encode(x) + u s  (x u8, encode: 2 (x + .5) / 256 - 1);  requant = round(clip(128 (data + 1) - 0.5, 0, 255))
// (get_ode_likelihood_fn, notebook_utils.py:316-337): uniform: u in [0,1) -> 2 (u - 0.5) / 256, s = 1; tn: s = e^{gt/2}
__global__ void dequantize_kernel(const unsigned char* __restrict__ x, const float* __restrict__ u,
                                  float* __restrict__ data, unsigned char* __restrict__ requant, size_t n, int uniform,
                                  float s) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float f = 2.f * (((float)x[i] + 0.5f) / 256.f) - 1.f;
    const float noise = uniform ? 2.f * (u[i] - 0.5f) / 256.f : u[i] * s;
    const float d = f + noise;
    data[i] = d;
    requant[i] = (unsigned char)rintf(fminf(fmaxf(128.f * (d + 1.f) - 0.5f, 0.f), 255.f));
  }
}

int grid_for(size_t n) { return (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256); }

}  // namespace

MULAN_API int mulan_ode_drift(const float* net, const float* x, const float* gt, const float* gp, const float* hutch,
                              float* drift, float* cot, size_t n, int mode, int g_per_sample, hipStream_t stream) {
  if (n == 0 || mode < 0 || mode > 6 || (mode & 3) == 3 || g_per_sample < 0 || (cot && !hutch)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ode_drift_kernel, dim3(grid_for(n)), dim3(256), 0, stream, net, x, gt, gp, hutch, drift, cot, n,
                     mode, g_per_sample);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_ode_div(const float* gx, const float* gt, const float* gp, const float* hutch, float* div, int B,
                            int d, int mode, int g_per_sample, hipStream_t stream) {
  if (B <= 0 || d <= 0 || mode < 0 || mode > 6 || (mode & 3) == 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ode_div_kernel, dim3(B), dim3(256), 0, stream, gx, gt, gp, hutch, div, d, mode, g_per_sample);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_rk_combine(const double* y, const float* K, size_t kstride, const double* coef, int ncoef, double h,
                               double* out, float* out32, size_t n, hipStream_t stream) {
  if (n == 0 || ncoef < 0 || ncoef > 7 || (!out && !out32)) return (int)hipErrorInvalidValue;
  Coef cf{};
  for (int j = 0; j < ncoef; ++j) cf.c[j] = coef[j];
  hipLaunchKernelGGL(rk_combine_kernel, dim3(grid_for(n)), dim3(256), 0, stream, y, K, kstride, cf, ncoef, h, out, out32,
                     n);
  MULAN_CHECK_LAUNCH();
}

MULAN_API size_t mulan_rk_workspace_bytes(void) { return (size_t)3 * kNormBlocks * sizeof(double); }

MULAN_API int mulan_rk_error_norm(const double* y, const double* ynew, const float* K, size_t kstride, const double* e,
                                  double h, double rtol, double atol, double* workspace, double* out, size_t n,
                                  hipStream_t stream) {
  if (n == 0) return (int)hipErrorInvalidValue;
  Coef cf{};
  for (int j = 0; j < 7; ++j) cf.c[j] = e[j];
  hipLaunchKernelGGL(rk_error_kernel, dim3(kNormBlocks), dim3(256), 0, stream, y, ynew, K, kstride, cf, h, rtol, atol,
                     workspace, n);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, stream, workspace, out, kNormBlocks);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_rk_init_norms(const double* y0, const float* f0, const float* f1, double rtol, double atol,
                                  double* workspace, double* out3, size_t n, hipStream_t stream) {
  if (n == 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(rk_init_norms_kernel, dim3(kNormBlocks), dim3(256), 0, stream, y0, f0, f1, rtol, atol, workspace,
                     n);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(3), dim3(64), 0, stream, workspace, out3, kNormBlocks);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_normal_logp(const float* x, float* out, int rows, int cols, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(normal_logp_kernel, dim3(rows), dim3(256), 0, stream, x, out, cols);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_noise(float* out, size_t n, unsigned long long seed, unsigned long long offset, int kind, float lo,
                          float hi, hipStream_t stream) {
  if (n == 0 || kind < 0 || kind > 3 || (kind == 2 && !(lo < hi))) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(noise_kernel, dim3(grid_for((n + 3) >> 2)), dim3(256), 0, stream, out, n, seed, offset, kind, lo, hi);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_dequantize(const unsigned char* x, const float* u, float* data, unsigned char* requant, size_t n,
                               int uniform, float scale, hipStream_t stream) {
  if (n == 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(dequantize_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, u, data, requant, n, uniform, scale);
  MULAN_CHECK_LAUNCH();
}
