// Small HBM-bound kernels around the contractions: activations, segmented column sums (bias /
// cond-bias gradients), row softmax for the single-head attention (ldm/model_vdm.py:773-786),
// Base-2 Fourier features (model_vdm.py:812-829) and the sinusoidal timestep embedding
// (model_vdm.py:391-413).  sin/cos use the accurate libm forms: arguments reach ~1000-3000 rad.
#include "common.h"

namespace {

// kind: 1 silu, 2 softplus + shift  (c = shift + softplus(u): model_mulan_epsilon.py:537)
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int kind, float shift) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float u = x[i];
    y[i] = kind == 1 ? silu_f(u) : shift + softplus_f(u);
  }
}
__global__ void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                               size_t n, int kind) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float u = x[i];
    dx[i] = dy[i] * (kind == 1 ? silu_grad_f(u) : sigmoid_f(u));
  }
}

// out[s][c] = sum_{r < seg} x[(s*seg + r)][c];  x is [nseg*seg, C] with row stride ld.
// block = 4 row-lanes x 64 columns.  This scalar form serves the narrow tensors (C = 1 / 3: the bias gradients of the
// conv_out layers), whose column sums cancel almost completely (measured: |sum| ~ 1e-5 of sum |x|): it accumulates in
// float64, so the result carries the rounding of its inputs only, not that of 1024 sequential fp32 additions.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int seg,
                                                     int C, int ld, int accumulate) {
  __shared__ double red[256];
  const int col = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int s = blockIdx.x;
  double acc = 0.0;
  if (col < C) {
    const float* base = x + (size_t)s * seg * ld + col;
    for (int r = rl; r < seg; r += 4) acc += (double)base[(size_t)r * ld];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0 && col < C) {
    const double v = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
    float* o = out + (size_t)s * C + col;
    *o = accumulate ? (float)((double)*o + v) : (float)v;
  }
}

// Same reduction for C % 4 == 0 and 16-byte aligned rows: block = 16 row-lanes x 16 float4 columns,
// 8 independent 16-byte loads in flight per thread.
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ x, float* __restrict__ out, int seg,
                                                         int C, int ld, int accumulate, float* __restrict__ out1) {
  __shared__ f32x4 red[256];
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int col = blockIdx.y * 64 + cq * 4;
  const int s = blockIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (col < C) {
    const float* base = x + (size_t)s * seg * ld + col;
    int r = rl;
    for (; r + 7 * 16 < seg; r += 8 * 16) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + (size_t)(r + u * 16) * ld);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
    }
    for (; r < seg; r += 16) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)r * ld);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    if (rl < o) {
      const f32x4 a = red[threadIdx.x], b = red[threadIdx.x + o * 16];
      red[threadIdx.x] = f32x4{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
    }
    __syncthreads();
  }
  if (rl == 0 && col < C) {
    f32x4 v = red[threadIdx.x];
    float* o = (out1 && s == 1 ? out1 : out + (size_t)s * C) + col;   // out1: segment 1 has its own destination
    if (accumulate) { v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

// y[row] = softmax(x[row]) over `cols` (<= 4096, multiple of 4); one 256-thread block per row.
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int cols,
                                                          float alpha) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  const float* xr = x + row * cols;
  float* yr = y + row * cols;
  f32x4 v[4];
  float mx = -INFINITY;
  const int nv = cols >> 2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = threadIdx.x + i * 256;
    if (q < nv) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + q * 4) * alpha;
      mx = fmaxf(mx, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
    }
  }
  mx = block_max_256(mx, red);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = threadIdx.x + i * 256;
    if (q < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[i][e] = expf(v[i][e] - mx); sum += v[i][e]; }
    }
  }
  sum = block_sum_256(sum, red);
  const float inv = 1.f / sum;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = threadIdx.x + i * 256;
    if (q < nv) {
      f32x4 o = {v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv};
      *reinterpret_cast<f32x4*>(yr + q * 4) = o;
    }
  }
}
// ds = p * (dp - sum_j dp_j p_j)
// ds = alpha p (dp - sum p dp)  (gradient w.r.t. the unscaled logits of softmax(alpha x)); rowmax (optional):
// [rows] maximum of |ds| of each row (mulan_absmax_rows over it gives the per-image maxima; atomics on a handful of
// addresses cost 0.5 ms here, measured)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                          float* __restrict__ ds, int cols, float alpha,
                                                          float* __restrict__ rowmax) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  const float* pr = p + row * cols;
  const float* dr = dp + row * cols;
  float* sr = ds + row * cols;
  f32x4 pv[4], dv[4];
  float dot = 0.f;
  const int nv = cols >> 2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = threadIdx.x + i * 256;
    if (q < nv) {
      pv[i] = *reinterpret_cast<const f32x4*>(pr + q * 4);
      dv[i] = *reinterpret_cast<const f32x4*>(dr + q * 4);
      dot += (pv[i][0] * dv[i][0] + pv[i][1] * dv[i][1]) + (pv[i][2] * dv[i][2] + pv[i][3] * dv[i][3]);
    }
  }
  dot = block_sum_256(dot, red);
  unsigned amax = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = threadIdx.x + i * 256;
    if (q < nv) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = alpha * (pv[i][e] * (dv[i][e] - dot));
        amax = max(amax, __float_as_uint(o[e]) & 0x7fffffffu);
      }
      *reinterpret_cast<f32x4*>(sr + q * 4) = o;
    }
  }
  if (rowmax) {
    const float m = block_max_256(__uint_as_float(amax), red);
    if (threadIdx.x == 0) rowmax[row] = m;
  }
}

// Fourier features: out[p][0..2] = z, out[p][3+2c+j] = sin(w_j z_c), out[p][9+2c+j] = cos(w_j z_c),
// out[p][15] = 0 (pad to 16 channels), w_j = 2^(6+j) * 2pi.   (model_vdm.py:341-343,812-829)
__device__ __forceinline__ float fourier_w(int j) { return (j == 0 ? 64.f : 128.f) * 6.283185307179586f; }

__global__ void fourier_fwd_kernel(const float* __restrict__ z, float* __restrict__ out, size_t npix) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
    float o[16];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float zc = z[p * 3 + c];
      o[c] = zc;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float a = fourier_w(j) * zc;
        o[3 + 2 * c + j] = sinf(a);
        o[9 + 2 * c + j] = cosf(a);
      }
    }
    o[15] = 0.f;
    f32x4* dst = reinterpret_cast<f32x4*>(out + p * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
  }
}
__global__ void fourier_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dout, float* __restrict__ dz,
                                   size_t npix, int accumulate) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
    float d[16];
    const f32x4* src = reinterpret_cast<const f32x4*>(dout + p * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x4 v = src[q]; d[4*q] = v[0]; d[4*q+1] = v[1]; d[4*q+2] = v[2]; d[4*q+3] = v[3]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float zc = z[p * 3 + c];
      float g = d[c];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float w = fourier_w(j), a = w * zc;
        g += w * (d[3 + 2 * c + j] * cosf(a) - d[9 + 2 * c + j] * sinf(a));
      }
      dz[p * 3 + c] = accumulate ? dz[p * 3 + c] + g : g;
    }
  }
}

// Timestep embedding: t[n] -> out[n][ld] at column offset col0: [sin(1000 t w_k), cos(1000 t w_k)],
// w_k = exp(-k ln(1e4)/(E/2-1)).  Optional copy of `extra[n or n/rep][nextra]` behind it (the concat
// with the conditioning vector: model_vdm.py:336; ldm/ldm_unet.py:85-88 with rep = 3072/3... see host).
__global__ void temb_fwd_kernel(const float* __restrict__ t, float* __restrict__ out, int n, int E, int ld, int col0) {
  const int half = E >> 1;
  const float nlw = (float)(-log(10000.0) / (double)(half - 1));  // f32(-ln(1e4)/(half-1)) as in the reference
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n * half; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / half), k = (int)(i - (size_t)r * half);
    const float w = expf((float)k * nlw);
    const float a = (t[r] * 1000.f) * w;
    out[(size_t)r * ld + col0 + k] = sinf(a);
    out[(size_t)r * ld + col0 + half + k] = cosf(a);
  }
}
// dt[r] = sum_k 1000 w_k (dsin_k cos(a) - dcos_k sin(a));  one wave per row.
__global__ void temb_bwd_kernel(const float* __restrict__ t, const float* __restrict__ dout, float* __restrict__ dt,
                                int n, int E, int ld, int col0) {
  const int half = E >> 1;
  const float nlw = (float)(-log(10000.0) / (double)(half - 1));  // f32(-ln(1e4)/(half-1)) as in the reference
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n) return;
  const float tv = t[r] * 1000.f;
  float acc = 0.f;
  for (int k = lane; k < half; k += 64) {
    const float w = expf((float)k * nlw), a = tv * w;
    acc += 1000.f * w * (dout[(size_t)r * ld + col0 + k] * cosf(a) - dout[(size_t)r * ld + col0 + half + k] * sinf(a));
  }
  acc = wave_sum(acc);
  if (lane == 0) dt[r] = acc;
}

// y[r][c] = x[r / rep][c]  (row broadcast, used for the per-pixel conditioning of ldm_unet.py:85-87)
__global__ void rowbcast_kernel(const float* __restrict__ x, float* __restrict__ y, size_t rows, int cols, int rep,
                                int ld, int col0) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cols; const int c = (int)(i - r * cols);
    y[r * ld + col0 + c] = x[(r / rep) * cols + c];
  }
}

// f = 2 * ((x + .5) / 256) - 1   (EncDec.encode, ldm/model_vdm.py:274-280)
__global__ void encode_u8_kernel(const unsigned char* __restrict__ x, float* __restrict__ f, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    f[i] = 2.f * (((float)x[i] + 0.5f) / 256.f) - 1.f;
}

__global__ void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float a, float b) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = a * x[i] + b * y[i];
}

inline int nblocks(size_t n, int per = 256) { size_t b = (n + per - 1) / per; return (int)(b > 4096 ? 4096 : (b ? b : 1)); }

}  // namespace

MULAN_API int mulan_act_fwd(const float* x, float* y, size_t n, int kind, float shift, hipStream_t stream) {
  hipLaunchKernelGGL(act_fwd_kernel, dim3(nblocks(n)), dim3(256), 0, stream, x, y, n, kind, shift);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_act_bwd(const float* x, const float* dy, float* dx, size_t n, int kind, hipStream_t stream) {
  hipLaunchKernelGGL(act_bwd_kernel, dim3(nblocks(n)), dim3(256), 0, stream, x, dy, dx, n, kind);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_colsum(const float* x, float* out, int nseg, int seg, int C, int ld, int accumulate,
                           hipStream_t stream) {
  if (nseg <= 0 || seg <= 0 || C <= 0) return (int)hipErrorInvalidValue;
  const bool vec = (C % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  if (vec)
    hipLaunchKernelGGL(colsum_vec_kernel, dim3(nseg, (C + 63) / 64), dim3(256), 0, stream, x, out, seg, C, ld,
                       accumulate, static_cast<float*>(nullptr));
  else
    hipLaunchKernelGGL(colsum_kernel, dim3(nseg, (C + 63) / 64), dim3(256), 0, stream, x, out, seg, C, ld, accumulate);
  MULAN_CHECK_LAUNCH();
}
// Two column sums in one launch: out0[c] = sum_r x[0][r][c], out1[c] = sum_r x[1][r][c] for x [2, seg, C]
// (the dgamma / dbeta partials of a GroupNorm backward, whose destinations are far apart in the flat gradient buffer).
MULAN_API int mulan_colsum_pair(const float* x, float* out0, float* out1, int seg, int C, hipStream_t stream) {
  if (seg <= 0 || C <= 0 || C % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out0) & 15) ||
      (reinterpret_cast<uintptr_t>(out1) & 15))
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(colsum_vec_kernel, dim3(2, (C + 63) / 64), dim3(256), 0, stream, x, out0, seg, C, C, 0, out1);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_softmax_fwd(const float* x, float* y, size_t rows, int cols, hipStream_t stream) {
  if (cols % 4 != 0 || cols > 4096 || rows == 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, y, cols, 1.f);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_softmax_bwd(const float* p, const float* dp, float* ds, size_t rows, int cols,
                                hipStream_t stream) {
  if (cols % 4 != 0 || cols > 4096 || rows == 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, stream, p, dp, ds, cols, 1.f,
                     static_cast<float*>(nullptr));
  MULAN_CHECK_LAUNCH();
}
// softmax(alpha x) and its gradient w.r.t. x (the 1/sqrt(C) of dot_product_attention, ldm/model_vdm.py:775-779, folded
// in); rowmax: optional [rows] receiving max |ds| of each row
MULAN_API int mulan_softmax_scaled_fwd(const float* x, float* y, size_t rows, int cols, float alpha,
                                       hipStream_t stream) {
  if (cols % 4 != 0 || cols > 4096 || rows == 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, y, cols, alpha);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_softmax_scaled_bwd(const float* p, const float* dp, float* ds, size_t rows, int cols, float alpha,
                                       float* rowmax, hipStream_t stream) {
  if (cols % 4 != 0 || cols > 4096 || rows == 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, stream, p, dp, ds, cols, alpha, rowmax);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_fourier_fwd(const float* z, float* out, size_t npix, hipStream_t stream) {
  hipLaunchKernelGGL(fourier_fwd_kernel, dim3(nblocks(npix)), dim3(256), 0, stream, z, out, npix);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_fourier_bwd(const float* z, const float* dout, float* dz, size_t npix, int accumulate,
                                hipStream_t stream) {
  hipLaunchKernelGGL(fourier_bwd_kernel, dim3(nblocks(npix)), dim3(256), 0, stream, z, dout, dz, npix, accumulate);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_temb_fwd(const float* t, float* out, int n, int E, int ld, int col0, hipStream_t stream) {
  if (E < 4 || (E & 1)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(temb_fwd_kernel, dim3(nblocks((size_t)n * (E / 2))), dim3(256), 0, stream, t, out, n, E, ld, col0);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_temb_bwd(const float* t, const float* dout, float* dt, int n, int E, int ld, int col0,
                             hipStream_t stream) {
  if (E < 4 || (E & 1)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(temb_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, t, dout, dt, n, E, ld, col0);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_rowbcast(const float* x, float* y, size_t rows, int cols, int rep, int ld, int col0,
                             hipStream_t stream) {
  hipLaunchKernelGGL(rowbcast_kernel, dim3(nblocks(rows * cols)), dim3(256), 0, stream, x, y, rows, cols, rep, ld, col0);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_encode_u8(const unsigned char* x, float* f, size_t n, hipStream_t stream) {
  hipLaunchKernelGGL(encode_u8_kernel, dim3(nblocks(n)), dim3(256), 0, stream, x, f, n);
  MULAN_CHECK_LAUNCH();
}
MULAN_API int mulan_axpby(const float* x, float* y, size_t n, float a, float b, hipStream_t stream) {
  hipLaunchKernelGGL(axpby_kernel, dim3(nblocks(n)), dim3(256), 0, stream, x, y, n, a, b);
  MULAN_CHECK_LAUNCH();
}
