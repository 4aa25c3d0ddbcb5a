// Ancestral sampler steps: the elementwise part of VDM.sample / conditional_sample and
// VDM.generate_x of the reference (ldm/model_mulan_velocity.py:281-368, ldm/model_mulan_epsilon.py:377-460,
// ldm/model_vdm.py:182-227).  The 1000 U-Net evaluations in between are the forward kernels of the train path.
#include "common.h"

namespace {

// z_s = sqrt(a/b) (z_t - sigma_t c eps_hat) + sqrt((1-a) c) eps,   a = sigmoid(-g_s), b = sigmoid(-g_t),
// c = -expm1(g_s - g_t), sigma_t = sqrt(sigmoid(g_t)), alpha_t = sqrt(sigmoid(-g_t));
// mode 0 (velocity): eps_hat = net alpha_t + sigma_t z_t;  mode 1 (epsilon): eps_hat = net;
// mode 2 (plain VDM, reparam_type 'input'): eps_hat = (z_t - alpha_t net) / sigma_t.
// gamma is per element (g_per_sample = 0, [n]) or per sample (g_per_sample = d: [n / d], the plain VDM).
__global__ void ancestral_step_kernel(const float* __restrict__ zt, const float* __restrict__ net,
                                      const float* __restrict__ gt, const float* __restrict__ gs,
                                      const float* __restrict__ eps, float* __restrict__ zs, size_t n, int mode,
                                      int g_per_sample) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t gi = g_per_sample ? i / (size_t)g_per_sample : i;
    const float g_t = gt[gi], g_s = gs[gi];
    const float a = sigmoid_f(-g_s), b = sigmoid_f(-g_t), c = -expm1f(g_s - g_t);
    const float sigma_t = sqrtf(sigmoid_f(g_t));
    const float z = zt[i];
    float eh = net[i];
    if (mode == 0) eh = eh * sqrtf(b) + sigma_t * z;
    if (mode == 2) eh = (z - sqrtf(b) * eh) / sigma_t;
    zs[i] = sqrtf(a / b) * (z - sigma_t * c * eh) + sqrtf(sigmoid_f(g_s) * c) * eps[i];   // 1 - a = sigmoid(g_s), without the cancellation
  }
}

// x = argmax_j of the 256-bin decoder logits of EncDec.decode (ldm/model_vdm.py:282-296) at
// z_0 / sqrt(1 - sigmoid(g_0)):  logits_j = -0.5 ((z - v_j) exp(-0.5 g_0))^2,  v_j = 2 (j + 0.5) / 256 - 1
// (first maximum wins, like jnp.argmax)
// sample != 0: jax.random.categorical(logits) instead of the argmax (sample_softmax = True), drawn by the Gumbel-max
// trick with Philox noise: counter (offset + 64 i + j / 4) so every (element, bin) has its own draw
__global__ void decode_argmax_kernel(const float* __restrict__ z0, const float* __restrict__ g0,
                                     unsigned char* __restrict__ out, size_t n, int g_per_sample, int sample,
                                     unsigned long long seed, unsigned long long offset) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float g = g0[g_per_sample ? i / (size_t)g_per_sample : i];
    const float z = z0[i] / sqrtf(1.f - sigmoid_f(g));
    const float inv_stdev = expf(-0.5f * g);
    float best = -INFINITY;
    int arg = 0;
    for (int j4 = 0; j4 < 64; ++j4) {
      float gum[4] = {0.f, 0.f, 0.f, 0.f};
      if (sample) {
        const Philox4 r = philox4x32_10(seed, offset + (unsigned long long)i * 64ull + j4, 0ull);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          gum[e] = -logf(-logf(((float)(w[e] >> 8) + 0.5f) * 5.9604644775390625e-08f));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j4 * 4 + e;
        const float v = 2.f * (((float)j + 0.5f) / 256.f) - 1.f;
        const float d = (z - v) * inv_stdev;
        const float l = -0.5f * d * d + gum[e];
        if (l > best) { best = l; arg = j; }
      }
    }
    out[i] = (unsigned char)arg;
  }
}

// EncDec.decode (ldm/model_vdm.py:282-296) as a table: out[i, j] = log_softmax_j(-0.5 ((z_i - v_j) exp(-0.5 g_i))^2), the 256
// decoder log-probabilities of every sub-pixel.  One wave per element (4 bins per lane, one 1 KB row written per wave);
// the train / eval path never materialises this table (mulan_qsample_fwd evaluates the bin of x in registers): this
// entry point backs the reference's module-level EncDec.decode / EncDec.__call__.
__global__ __launch_bounds__(256) void decode_logprobs_kernel(const float* __restrict__ z, const float* __restrict__ g0,
                                                              float* __restrict__ out, size_t n, int g_per_sample) {
  const int lane = threadIdx.x & 63;
  for (size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (size_t)gridDim.x * 4) {
    const float g = g0[g_per_sample ? i / (size_t)g_per_sample : i];
    const float zi = z[i], istd = expf(-0.5f * g);
    float l[4], mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = 2.f * (((float)(lane * 4 + e) + 0.5f) / 256.f) - 1.f;
      const float u = (zi - v) * istd;
      l[e] = -0.5f * u * u;
      mx = fmaxf(mx, l[e]);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = expf(l[0] - mx) + expf(l[1] - mx) + expf(l[2] - mx) + expf(l[3] - mx);
    se = wave_sum(se);
    // (l - mx) first: both are O(1e4 .. 1e6), their difference is exact; adding log(se) to mx would round it away
    const float ls = logf(se);
    float4 o4 = make_float4((l[0] - mx) - ls, (l[1] - mx) - ls, (l[2] - mx) - ls, (l[3] - mx) - ls);
    *reinterpret_cast<float4*>(out + i * 256 + lane * 4) = o4;
  }
}

// out[r] = mean of x[r, 0:cols]  (VDM._get_score_model_gt, ldm/model_mulan_velocity.py:141-146); one wave per row
__global__ __launch_bounds__(256) void rowmean_kernel(const float* __restrict__ x, float* __restrict__ out, int rows,
                                                      int cols) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += x[(size_t)row * cols + c];
  s = wave_sum(s);
  if (lane == 0) out[row] = s / (float)cols;
}

int grid_for(size_t n) { return (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256); }

}  // namespace

MULAN_API int mulan_ancestral_step(const float* zt, const float* net, const float* gt, const float* gs, const float* eps,
                                   float* zs, size_t n, int mode, int g_per_sample, hipStream_t stream) {
  if (n == 0 || mode < 0 || mode > 2 || g_per_sample < 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ancestral_step_kernel, dim3(grid_for(n)), dim3(256), 0, stream, zt, net, gt, gs, eps, zs, n, mode,
                     g_per_sample);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_decode_argmax(const float* z0, const float* g0, unsigned char* out, size_t n, int g_per_sample,
                                  hipStream_t stream) {
  if (n == 0 || g_per_sample < 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(decode_argmax_kernel, dim3(grid_for(n)), dim3(256), 0, stream, z0, g0, out, n, g_per_sample, 0, 0ull,
                     0ull);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_decode_logprobs(const float* z, const float* g0, float* out, size_t n, int g_per_sample,
                                    hipStream_t stream) {
  if (n == 0 || g_per_sample < 0 || !z || !g0 || !out) return (int)hipErrorInvalidValue;
  const size_t blocks = (n + 3) / 4;
  hipLaunchKernelGGL(decode_logprobs_kernel, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, stream, z, g0,
                     out, n, g_per_sample);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_decode_sample(const float* z0, const float* g0, unsigned char* out, size_t n, int g_per_sample,
                                  unsigned long long seed, unsigned long long offset, hipStream_t stream) {
  if (n == 0 || g_per_sample < 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(decode_argmax_kernel, dim3(grid_for(n)), dim3(256), 0, stream, z0, g0, out, n, g_per_sample, 1, seed,
                     offset);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_rowmean(const float* x, float* out, int rows, int cols, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(rowmean_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, out, rows, cols);
  MULAN_CHECK_LAUNCH();
}
