// A caller of libmulan_hip.so that is neither Python nor torch: plain C++ over include/mulan_hip.h and the HIP runtime API
// (hipMalloc / hipMemcpy / a stream of its own), the way SURVEY 8(b) names "C++ unit-test driver" as the second caller of
// the C ABI.  It runs the 3x3 convolution of the ResnetBlock (ldm/model_vdm.py:633-656) through the boundary in both
// arithmetic modes, its weight gradient and its input gradient, on small-integer data -- where fp32 AND the f16x3 split
// are exact, so the comparison with the loops below is bit for bit --, GroupNorm + swish and the AdamW / EMA step against
// double-precision loops (tolerances at the checks), and checks the error convention (a hipError_t as
// int, no exception, nothing launched) on arguments the entry points must refuse.
//   hipcc -O1 -I include tests/abi_driver.cpp -o /tmp/abi_driver -L mulan_amd -lmulan_hip -Wl,-rpath,$PWD/mulan_amd
// tests/test_abi_and_host.py compiles it (CPU suite); tests/test_gpu_kernels.py runs it on the GPU.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mulan_hip.h"

namespace {

int g_checks = 0, g_failed = 0;

void expect(bool ok, const char* what) {
  ++g_checks;
  if (!ok) { ++g_failed; std::fprintf(stderr, "abi_driver: FAILED: %s\n", what); }
}

void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) { std::fprintf(stderr, "abi_driver: %s: %s\n", what, hipGetErrorString(e)); std::exit(2); }
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  explicit DevBuf(size_t count) : n(count) { hip_ok(hipMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T)), "hipMalloc"); }
  DevBuf(const std::vector<T>& h) : DevBuf(h.size()) { hip_ok(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy h2d"); }
  ~DevBuf() { (void)hipFree(p); }
  std::vector<T> host() const {
    std::vector<T> h(n);
    hip_ok(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy d2h");
    return h;
  }
};

// small integers from a 64-bit LCG: |v| <= amp
std::vector<float> ints(size_t n, int amp, uint64_t seed) {
  std::vector<float> v(n);
  uint64_t s = seed * 6364136223846793005ull + 1442695040888963407ull;
  for (size_t i = 0; i < n; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    v[i] = (float)((int)((s >> 33) % (2 * amp + 1)) - amp);
  }
  return v;
}

constexpr int H = 32, W = 32;

// y[b,h,w,n] = sum_{kh,kw,c} x[b,h+kh-1,w+kw-1,c] w[kh,kw,c,n] + bias[n] + res[b,h,w,n]   (SAME padding, NHWC / HWIO)
std::vector<float> conv_host(const std::vector<float>& x, const std::vector<float>& w, const std::vector<float>* bias,
                             const std::vector<float>* res, int B, int C, int N) {
  std::vector<float> y((size_t)B * H * W * N);
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h)
      for (int ww = 0; ww < W; ++ww)
        for (int n = 0; n < N; ++n) {
          double s = bias ? (*bias)[n] : 0.0;
          for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
              const int hh = h + kh - 1, wc = ww + kw - 1;
              if (hh < 0 || hh >= H || wc < 0 || wc >= W) continue;
              const float* xp = &x[(((size_t)b * H + hh) * W + wc) * C];
              const float* wp = &w[((size_t)(kh * 3 + kw) * C) * N + n];
              for (int c = 0; c < C; ++c) s += (double)xp[c] * wp[(size_t)c * N];
            }
          const size_t o = (((size_t)b * H + h) * W + ww) * N + n;
          if (res) s += (*res)[o];
          y[o] = (float)s;
        }
  return y;
}

// dw[kh,kw,c,n] = sum_{b,h,w} x[b,h+kh-1,w+kw-1,c] dy[b,h,w,n]
std::vector<float> wgrad_host(const std::vector<float>& x, const std::vector<float>& dy, int B, int C, int N) {
  std::vector<double> acc((size_t)9 * C * N, 0.0);
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h)
      for (int ww = 0; ww < W; ++ww)
        for (int kh = 0; kh < 3; ++kh)
          for (int kw = 0; kw < 3; ++kw) {
            const int hh = h + kh - 1, wc = ww + kw - 1;
            if (hh < 0 || hh >= H || wc < 0 || wc >= W) continue;
            const float* xp = &x[(((size_t)b * H + hh) * W + wc) * C];
            const float* gp = &dy[(((size_t)b * H + h) * W + ww) * N];
            double* a = &acc[(size_t)(kh * 3 + kw) * C * N];
            for (int c = 0; c < C; ++c)
              for (int n = 0; n < N; ++n) a[(size_t)c * N + n] += (double)xp[c] * gp[n];
          }
  std::vector<float> dw(acc.size());
  for (size_t i = 0; i < acc.size(); ++i) dw[i] = (float)acc[i];
  return dw;
}

// wT[t][n][c] = w[8 - t][c][n]: the weights of the input-gradient convolution (what mulan_conv3x3_wflip / flip = 1 form)
std::vector<float> flip_host(const std::vector<float>& w, int C, int N) {
  std::vector<float> t((size_t)9 * C * N);
  for (int k = 0; k < 9; ++k)
    for (int c = 0; c < C; ++c)
      for (int n = 0; n < N; ++n) t[((size_t)k * N + n) * C + c] = w[((size_t)(8 - k) * C + c) * N + n];
  return t;
}

bool same(const std::vector<float>& a, const std::vector<float>& b) {
  return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0;
}

void exact_fp32_path(hipStream_t stream) {
  const int B = 2, C = 8, N = 16;
  const auto x = ints((size_t)B * H * W * C, 3, 1), w = ints((size_t)9 * C * N, 2, 2), bias = ints(N, 5, 3),
             res = ints((size_t)B * H * W * N, 7, 4), dy = ints((size_t)B * H * W * N, 2, 5);
  DevBuf<float> dx(x), dw(w), dbias(bias), dres(res), ddy(dy), y((size_t)B * H * W * N);
  expect(mulan_conv3x3_fwd(dx.p, dw.p, dbias.p, nullptr, 0, dres.p, y.p, B, H, W, C, N, stream) == 0, "mulan_conv3x3_fwd returns 0");
  hip_ok(hipStreamSynchronize(stream), "sync");
  expect(same(y.host(), conv_host(x, w, &bias, &res, B, C, N)), "mulan_conv3x3_fwd == host loops, bit for bit (integer data)");

  // input gradient = the same entry point on dy with the flipped weights, C and N swapped
  DevBuf<float> wT((size_t)9 * C * N), gx((size_t)B * H * W * C);
  expect(mulan_conv3x3_wflip(dw.p, wT.p, C, N, stream) == 0, "mulan_conv3x3_wflip returns 0");
  expect(mulan_conv3x3_fwd(ddy.p, wT.p, nullptr, nullptr, 0, nullptr, gx.p, B, H, W, N, C, stream) == 0, "input-gradient launch returns 0");
  hip_ok(hipStreamSynchronize(stream), "sync");
  expect(same(wT.host(), flip_host(w, C, N)), "mulan_conv3x3_wflip == host transpose");
  expect(same(gx.host(), conv_host(dy, flip_host(w, C, N), nullptr, nullptr, B, N, C)), "input gradient == host loops");

  const size_t wsb = mulan_conv3x3_wgrad_workspace(B, H, W, C, N);
  DevBuf<float> ws(wsb / 4 + 1), gw((size_t)9 * C * N);
  expect(mulan_conv3x3_wgrad(dx.p, ddy.p, gw.p, ws.p, B, H, W, C, N, 0, stream) == 0, "mulan_conv3x3_wgrad returns 0");
  hip_ok(hipStreamSynchronize(stream), "sync");
  expect(same(gw.host(), wgrad_host(x, dy, B, C, N)), "mulan_conv3x3_wgrad == host loops");
}

void f16x3_path(hipStream_t stream) {
  // the fp16-matrix-core path: maxima -> packed weights -> convolution; C % 32 == 0 and N % 128 == 0 select the kernel
  // the train step runs (conv3x3_f16x3_v3); integers below 2^11 split exactly into the two fp16 pieces
  const int B = 2, C = 32, N = 128;
  const auto x = ints((size_t)B * H * W * C, 3, 11), w = ints((size_t)9 * C * N, 2, 12), bias = ints(N, 5, 13),
             res = ints((size_t)B * H * W * N, 7, 14), dy = ints((size_t)B * H * W * N, 2, 15);
  DevBuf<float> dx(x), dw(w), dbias(bias), dres(res), ddy(dy), y((size_t)B * H * W * N);
  DevBuf<unsigned> xmax((size_t)B * 16), wmax(16), ymax((size_t)B * 16), dymax((size_t)B * 16);
  expect(mulan_absmax_rows(dx.p, xmax.p, B, (size_t)H * W * C, stream) == 0, "mulan_absmax_rows(x) returns 0");
  expect(mulan_absmax_rows(dw.p, wmax.p, 1, (size_t)9 * C * N, stream) == 0, "mulan_absmax_rows(w) returns 0");
  const size_t pb = mulan_conv3x3_pack_f16x3_bytes(C, N);
  expect(pb >= (size_t)9 * C * N * 4, "mulan_conv3x3_pack_f16x3_bytes covers two fp16 planes");
  DevBuf<unsigned char> wp(pb), wpT(pb);
  expect(mulan_conv3x3_pack_f16x3(dw.p, wp.p, wmax.p, C, N, 0, stream) == 0, "mulan_conv3x3_pack_f16x3 returns 0");
  expect(mulan_conv3x3_fwd_f16x3(dx.p, xmax.p, wp.p, wmax.p, dbias.p, nullptr, 0, dres.p, y.p, nullptr, ymax.p, B, H, W, C, N,
                                 stream) == 0, "mulan_conv3x3_fwd_f16x3 returns 0");
  hip_ok(hipStreamSynchronize(stream), "sync");
  const auto want = conv_host(x, w, &bias, &res, B, C, N);
  expect(same(y.host(), want), "mulan_conv3x3_fwd_f16x3 == host loops, bit for bit (integer data)");
  // the maxima by-product: the largest |y| of each image among its 16 partial maxima (fp32 bit patterns)
  {
    const auto m = ymax.host();
    bool ok = true;
    for (int b = 0; b < B; ++b) {
      float hm = 0.f, dm = 0.f;
      for (size_t i = 0; i < (size_t)H * W * N; ++i) { const float a = want[(size_t)b * H * W * N + i]; hm = a < 0 ? (-a > hm ? -a : hm) : (a > hm ? a : hm); }
      for (int k = 0; k < 16; ++k) { float f; std::memcpy(&f, &m[b * 16 + k], 4); dm = f > dm ? f : dm; }
      ok = ok && hm == dm;
    }
    expect(ok, "ymax of mulan_conv3x3_fwd_f16x3 == per-image max |y|");
  }

  // input gradient: flip = 1 packs the tap-flipped, channel-transposed weights ([3,3,N,C]: N % 32, C % 128 -> swap roles
  // by using a square layer for this leg)
  const int E = 128;
  const auto xs = ints((size_t)B * H * W * E, 2, 21), wsq = ints((size_t)9 * E * E, 1, 22), gy = ints((size_t)B * H * W * E, 2, 23);
  DevBuf<float> dxs(xs), dwsq(wsq), dgy(gy), gx((size_t)B * H * W * E), gw((size_t)9 * E * E);
  DevBuf<unsigned> xsmax((size_t)B * 16), gymax((size_t)B * 16), wsqmax(16);
  const size_t pbs = mulan_conv3x3_pack_f16x3_bytes(E, E);
  DevBuf<unsigned char> wpf(pbs);
  expect(mulan_absmax_rows(dxs.p, xsmax.p, B, (size_t)H * W * E, stream) == 0 &&
         mulan_absmax_rows(dgy.p, gymax.p, B, (size_t)H * W * E, stream) == 0 &&
         mulan_absmax_rows(dwsq.p, wsqmax.p, 1, (size_t)9 * E * E, stream) == 0, "maxima of the square layer");
  expect(mulan_conv3x3_pack_f16x3(dwsq.p, wpf.p, wsqmax.p, E, E, 1, stream) == 0, "pack with flip = 1 returns 0");
  expect(mulan_conv3x3_fwd_f16x3(dgy.p, gymax.p, wpf.p, wsqmax.p, nullptr, nullptr, 0, nullptr, gx.p, nullptr, nullptr, B, H, W, E,
                                 E, stream) == 0, "input-gradient launch (f16x3) returns 0");
  const size_t wsb = mulan_conv3x3_wgrad_f16x3_workspace(B, H, W, E, E);
  DevBuf<float> ws(wsb / 4 + 1);
  expect(mulan_conv3x3_wgrad_f16x3(dxs.p, xsmax.p, dgy.p, gymax.p, gw.p, ws.p, B, H, W, E, E, 0, stream) == 0,
         "mulan_conv3x3_wgrad_f16x3 returns 0");
  hip_ok(hipStreamSynchronize(stream), "sync");
  expect(same(gx.host(), conv_host(gy, flip_host(wsq, E, E), nullptr, nullptr, B, E, E)), "f16x3 input gradient == host loops");
  expect(same(gw.host(), wgrad_host(xs, gy, B, E, E)), "mulan_conv3x3_wgrad_f16x3 == host loops");
}

// floats in (-1, 1) from the same LCG
std::vector<float> reals(size_t n, float amp, uint64_t seed) {
  std::vector<float> v(n);
  uint64_t s = seed * 6364136223846793005ull + 1442695040888963407ull;
  for (size_t i = 0; i < n; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    v[i] = amp * ((float)((s >> 40) & 0xffffff) / 8388608.f - 1.f);
  }
  return v;
}

double max_abs_diff(const std::vector<float>& a, const std::vector<double>& b) {
  double m = 0;
  for (size_t i = 0; i < a.size(); ++i) { const double d = std::fabs((double)a[i] - b[i]); m = d > m ? d : m; }
  return m;
}

void groupnorm_and_optimizer(hipStream_t stream) {
  // nn.GroupNorm(32 groups, eps 1e-6, fast variance E[x^2] - E[x]^2) + swish over the channel concat [x1 | x2]
  // (ldm/model_vdm.py:622-623): against double-precision loops, tolerance 2e-5 absolute on O(1) outputs (fp32 sums over
  // 4096 elements per group)
  const int B = 2, C1 = 64, C2 = 64, Ct = C1 + C2, G = 32, cpg = Ct / G, HWp = H * W;
  const auto x1 = reals((size_t)B * HWp * C1, 2.f, 31), x2 = reals((size_t)B * HWp * C2, 2.f, 32);
  const auto gamma = reals(Ct, 1.5f, 33), beta = reals(Ct, 0.5f, 34);
  DevBuf<float> dx1(x1), dx2(x2), dg(gamma), db(beta), y((size_t)B * HWp * Ct), mean((size_t)B * G), rstd((size_t)B * G);
  expect(mulan_groupnorm_fwd(dx1.p, dx2.p, C1, C2, dg.p, db.p, y.p, mean.p, rstd.p, B, HWp, G, 1e-6f, 1, 1.f, 0ull, 0ull, nullptr,
                             stream) == 0, "mulan_groupnorm_fwd returns 0");
  hip_ok(hipStreamSynchronize(stream), "sync");
  std::vector<double> want((size_t)B * HWp * Ct);
  for (int b = 0; b < B; ++b)
    for (int g = 0; g < G; ++g) {
      double s1 = 0, s2 = 0;
      for (int px = 0; px < HWp; ++px)
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
          const double v = c < C1 ? x1[((size_t)b * HWp + px) * C1 + c] : x2[((size_t)b * HWp + px) * C2 + c - C1];
          s1 += v; s2 += v * v;
        }
      const double n = (double)HWp * cpg, m = s1 / n, var = s2 / n - m * m, r = 1.0 / std::sqrt((var > 0 ? var : 0) + 1e-6);
      for (int px = 0; px < HWp; ++px)
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
          const double v = c < C1 ? x1[((size_t)b * HWp + px) * C1 + c] : x2[((size_t)b * HWp + px) * C2 + c - C1];
          const double u = (v - m) * r * gamma[c] + beta[c];
          want[((size_t)b * HWp + px) * Ct + c] = u / (1.0 + std::exp(-u));
        }
    }
  expect(max_abs_diff(y.host(), want) < 2e-5, "mulan_groupnorm_fwd (+swish, concat input) == double loops to 2e-5");
  expect(mulan_groupnorm_fwd(dx1.p, dx2.p, C1, C2, dg.p, db.p, y.p, mean.p, rstd.p, B, 1000, G, 1e-6f, 1, 1.f, 0ull, 0ull, nullptr,
                             stream) != 0, "hw != 1024 is refused");

  // optax.adamw (b1 0.9, b2 0.99, eps 1e-8, wd 0.01 on the first n_decay elements) + EMA, ldm/train_state.py:70-102
  const size_t n = 10000, n_decay = 6000;
  const auto p0 = reals(n, 1.f, 41), g0 = reals(n, 0.1f, 42), m0 = reals(n, 0.05f, 43), e0 = reals(n, 1.f, 45);
  auto v0 = reals(n, 0.01f, 44);
  for (auto& v : v0) v = std::fabs(v);
  DevBuf<float> dp(p0), dgr(g0), dm(m0), dv(v0), de(e0);
  const float lr = 2e-4f, b1 = 0.9f, b2 = 0.99f, eps = 1e-8f, wd = 0.01f, ema_rate = 0.9999f, gscale = 0.5f;
  const int step = 7;
  expect(mulan_adamw_ema_step(dp.p, dgr.p, dm.p, dv.p, de.p, n, n_decay, lr, b1, b2, eps, wd, step, ema_rate, gscale, stream) == 0,
         "mulan_adamw_ema_step returns 0");
  hip_ok(hipStreamSynchronize(stream), "sync");
  std::vector<double> wp(n), wm(n), wv(n), we(n);
  const double bc1 = 1.0 - std::pow((double)b1, step), bc2 = 1.0 - std::pow((double)b2, step);
  for (size_t i = 0; i < n; ++i) {
    const double g = (double)g0[i] * gscale;
    wm[i] = b1 * (double)m0[i] + (1.0 - b1) * g;
    wv[i] = b2 * (double)v0[i] + (1.0 - b2) * g * g;
    double u = (wm[i] / bc1) / (std::sqrt(wv[i] / bc2) + eps);
    if (i < n_decay) u += wd * (double)p0[i];
    wp[i] = (double)p0[i] - lr * u;
    we[i] = ema_rate * (double)e0[i] + (1.0 - ema_rate) * wp[i];
  }
  expect(max_abs_diff(dp.host(), wp) < 2e-7 && max_abs_diff(dm.host(), wm) < 1e-7 && max_abs_diff(dv.host(), wv) < 1e-8 &&
         max_abs_diff(de.host(), we) < 2e-7, "mulan_adamw_ema_step == optax.adamw + EMA in double (fp32 rounding)");
}

void error_convention(hipStream_t stream) {
  // refused arguments: a non-zero hipError_t comes back, nothing is launched, nothing throws, the stream stays usable
  DevBuf<float> a(1024), b(1024);
  DevBuf<unsigned> m(64);
  DevBuf<unsigned char> wp(1024);
  expect(mulan_conv3x3_fwd_f16x3(a.p, m.p, wp.p, m.p, nullptr, nullptr, 0, nullptr, b.p, nullptr, nullptr, 1, 32, 31, 32, 128, stream) != 0,
         "W != 32 is refused");
  expect(mulan_conv3x3_fwd_f16x3(a.p, nullptr, wp.p, m.p, nullptr, nullptr, 0, nullptr, b.p, nullptr, nullptr, 1, 32, 32, 32, 128, stream) != 0,
         "missing maxima are refused");
  expect(mulan_conv3x3_wgrad_f16x3(a.p, m.p, b.p, m.p, a.p, b.p, 0, 32, 32, 32, 128, 0, stream) != 0, "B == 0 is refused");
  expect(hipStreamSynchronize(stream) == hipSuccess && hipGetLastError() == hipSuccess, "no sticky device error after refused calls");
  void* sig = nullptr;
  const int rc = mulan_signal_create(&sig);           // (hipErrorNotSupported where the device cannot wait on values)
  expect(rc == 0 ? sig != nullptr : sig == nullptr, "mulan_signal_create: handle iff success");
  if (rc == 0) {
    DevBuf<unsigned> tick(std::vector<unsigned>{7u});
    unsigned out[2] = {0, 0};
    expect(mulan_signal_set(sig, tick.p, stream) == 0 && mulan_stream_wait_signal(stream, sig, 7u) == 0 &&
           hipStreamSynchronize(stream) == hipSuccess && mulan_signal_read(sig, out) == 0 && out[0] == 7u,
           "signal word: set by a kernel, waited for by the stream, read back");
    expect(mulan_signal_destroy(sig) == 0, "mulan_signal_destroy returns 0");
  }
}

}  // namespace

int main() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::fprintf(stderr, "abi_driver: no HIP device\n"); return 3; }
  hip_ok(hipSetDevice(0), "hipSetDevice");
  const char* v = mulan_version();
  expect(v != nullptr && v[0] != 0, "mulan_version");
  hipStream_t stream;
  hip_ok(hipStreamCreate(&stream), "hipStreamCreate");
  exact_fp32_path(stream);
  f16x3_path(stream);
  groupnorm_and_optimizer(stream);
  error_convention(stream);
  hip_ok(hipStreamDestroy(stream), "hipStreamDestroy");
  std::printf("abi_driver: %d checks, %d failed (library %s)\n", g_checks, g_failed, v ? v : "?");
  return g_failed ? 1 : 0;
}
