// Flat-buffer AdamW + EMA step and the Philox normal generator.
// AdamW semantics follow optax.adamw as configured by the reference (ldm/experiment.py:132-182:
// scale_by_adam(b1,b2,eps) -> add_decayed_weights(wd, mask) -> scale(-lr)) and the EMA of
// ldm/train_state.py:88-95: ema += (1 - rate) * (p_new - ema).  All parameters live in one
// contiguous fp32 buffer whose first `n_decay` elements are weight-decayed (kernels and GroupNorm
// scales) and the rest are not (leaves named `bias`).  HBM traffic: 5 reads + 4 writes = 36 B/param.
#include "common.h"

namespace {

struct AdamArgs {
  float* p; const float* g; float* m; float* v; float* ema;
  size_t n, n_decay;
  float lr, b1, b2, eps, wd, bc1, bc2, ema_rate, gscale;
  const float* gscale_dev;   // optional extra factor computed on the device (global-norm clipping)
  const float* dyn;          // optional [3] on the device: lr, 1 - b1^t, 1 - b2^t (stream-ordered: graph replay)
};

// U float4 per array and thread in flight (U = 2: 10 loads of 16 B before the first use); NT: non-temporal loads and
// stores (every byte is touched once per step; the arrays are 4 x 142 MB + the gradient, nothing of it is reused
// before the next step's weight packing reads p)
template <int U, bool NT>
__global__ __launch_bounds__(256) void adamw_ema_kernel(AdamArgs a) {
  const size_t n4 = a.n >> 2;
  const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2, ome = 1.f - a.ema_rate;
  if (a.gscale_dev) a.gscale *= a.gscale_dev[0];
  if (a.dyn) { a.lr = a.dyn[0]; a.bc1 = a.dyn[1]; a.bc2 = a.dyn[2]; }
  const auto ld = [](const float* q) -> f32x4 {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
    else return ld_stream4(q);
  };
  const auto st = [](float* q, f32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(q));
    else *reinterpret_cast<f32x4*>(q) = v;
  };
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t q0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q0 < n4; q0 += stride * U) {
    f32x4 p[U], g[U], m[U], v[U], e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t q = q0 + stride * u;
      if (q < n4) {
        const size_t i = q << 2;
        p[u] = ld(a.p + i); g[u] = ld(a.g + i); m[u] = ld(a.m + i); v[u] = ld(a.v + i); e[u] = ld(a.ema + i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t q = q0 + stride * u;
      if (q >= n4) continue;
      const size_t i = q << 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gk = g[u][k] * a.gscale;
        m[u][k] = a.b1 * m[u][k] + omb1 * gk;
        v[u][k] = a.b2 * v[u][k] + omb2 * gk * gk;
        float up = (m[u][k] / a.bc1) / (sqrtf(v[u][k] / a.bc2) + a.eps);
        if (i + k < a.n_decay) up += a.wd * p[u][k];
        p[u][k] -= a.lr * up;
        e[u][k] += ome * (p[u][k] - e[u][k]);
      }
      st(a.p + i, p[u]); st(a.m + i, m[u]); st(a.v + i, v[u]); st(a.ema + i, e[u]);
    }
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0) {
    for (size_t i = (n4 << 2) + threadIdx.x; i < a.n; i += blockDim.x) {
      const float gk = a.g[i] * a.gscale;
      const float m = a.b1 * a.m[i] + omb1 * gk;
      const float v = a.b2 * a.v[i] + omb2 * gk * gk;
      float u = (m / a.bc1) / (sqrtf(v / a.bc2) + a.eps);
      float p = a.p[i];
      if (i < a.n_decay) u += a.wd * p;
      p -= a.lr * u;
      a.p[i] = p; a.m[i] = m; a.v[i] = v;
      a.ema[i] += ome * (p - a.ema[i]);
    }
  }
}

// optax.clip_by_global_norm (ldm/experiment.py:176-178): partial sums of squares of the flat gradient, then
// scale = min(1, clip / (pre * sqrt(sum)))  (pre = 1 / world: the norm is that of the rank-averaged gradient)
constexpr int kNormParts = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, double* __restrict__ part) {
  __shared__ double red[4];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double v = (double)g[i];
    s += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void clip_scale_kernel(const double* __restrict__ part, float clip, float pre,
                                                        float* __restrict__ out) {
  double s = 0.0;
  for (int j = threadIdx.x; j < kNormParts; j += 64) s += part[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) {
    const double norm = (double)pre * sqrt(s);
    out[0] = (float)fmin(1.0, (double)clip / norm);       // clip * min(1 / norm, 1 / clip)
    out[1] = (float)norm;
  }
}

// out[i] ~ N(0,1): Philox4x32-10(seed, counter = offset + i/4) + Box-Muller.
__global__ void randn_kernel(float* __restrict__ out, size_t n, unsigned long long seed, unsigned long long offset) {
  const size_t n4 = (n + 3) >> 2;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
    const Philox4 r = philox4x32_10(seed, offset + q, 0ull);
    const float k = 2.3283064365386963e-10f;   // 2^-32
    const float u0 = ((float)r.x + 0.5f) * k, u1 = ((float)r.y + 0.5f) * k;
    const float u2 = ((float)r.z + 0.5f) * k, u3 = ((float)r.w + 0.5f) * k;
    const float r0 = sqrtf(-2.f * logf(fminf(u0, 1.f))), r1 = sqrtf(-2.f * logf(fminf(u2, 1.f)));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u1, &s0, &c0);
    sincosf(6.283185307179586f * u3, &s1, &c1);
    const float z[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
    const size_t i = q << 2;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < n) out[i + e] = z[e];
  }
}

}  // namespace

static int adamw_launch(float* p, const float* g, float* m, float* v, float* ema, size_t n, size_t n_decay, float lr,
                        float b1, float b2, float eps, float weight_decay, int step, float ema_rate, float grad_scale,
                        const float* grad_scale_dev, hipStream_t stream, const float* dyn = nullptr) {
  if (step < 1 && !dyn) return (int)hipErrorInvalidValue;
  const auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!(al(p) && al(g) && al(m) && al(v) && al(ema))) return (int)hipErrorInvalidValue;
  const int st = step < 1 ? 1 : step;
  AdamArgs a{p, g, m, v, ema, n, n_decay, lr, b1, b2, eps, weight_decay,
             (float)(1.0 - pow((double)b1, st)), (float)(1.0 - pow((double)b2, st)), ema_rate, grad_scale,
             grad_scale_dev, dyn};
  // tune[25] (dev A/B): 0 = as shipped (one float4 per array in flight, non-temporal); 1: plain loads / stores (the form of
  // rounds 1-4); 2: two float4 in flight, plain; 3: two, non-temporal.  tune[26]: block cap (0 = 8192).  Alone at 35.6 M
  // elements (tools/adamw_bench.py, profiles/r05_adamw_bench.log): 264 us plain / 2048 blocks -> 227 us (5.6 TB/s).  In
  // the train step (71 M parameters, 2.56 GB per launch) the kernel was at the copy ceiling before: 438 -> 433 us.
  const int var = g_mulan_tune[25];
  const int U = (var == 2 || var == 3) ? 2 : 1;
  size_t blocks = ((n >> 2) + 256 * U - 1) / (256 * U);
  const size_t cap = g_mulan_tune[26] > 0 ? (size_t)g_mulan_tune[26] : 8192;
  if (blocks > cap) blocks = cap;
  if (blocks == 0) blocks = 1;
  const dim3 grid((unsigned)blocks);
  if (var == 1) hipLaunchKernelGGL((adamw_ema_kernel<1, false>), grid, dim3(256), 0, stream, a);
  else if (var == 2) hipLaunchKernelGGL((adamw_ema_kernel<2, false>), grid, dim3(256), 0, stream, a);
  else if (var == 3) hipLaunchKernelGGL((adamw_ema_kernel<2, true>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((adamw_ema_kernel<1, true>), grid, dim3(256), 0, stream, a);
  MULAN_CHECK_LAUNCH();
}

// The step with its per-step scalars on the device: dyn[0] = learning rate, dyn[1] = 1 - b1^t, dyn[2] = 1 - b2^t (t = the
// Adam count), read by the kernel when it runs, so that a captured HIP graph replays with this step's values;
// grad_scale_dev as in mulan_adamw_ema_step_scaled (may be NULL).
MULAN_API int mulan_adamw_ema_step_dyn(float* p, const float* g, float* m, float* v, float* ema, size_t n,
                                       size_t n_decay, float b1, float b2, float eps, float weight_decay,
                                       float ema_rate, float grad_scale, const float* grad_scale_dev,
                                       const float* dyn, hipStream_t stream) {
  if (!dyn) return (int)hipErrorInvalidValue;
  return adamw_launch(p, g, m, v, ema, n, n_decay, 0.f, b1, b2, eps, weight_decay, 0, ema_rate, grad_scale,
                      grad_scale_dev, stream, dyn);
}

MULAN_API int mulan_adamw_ema_step(float* p, const float* g, float* m, float* v, float* ema, size_t n,
                                   size_t n_decay, float lr, float b1, float b2, float eps, float weight_decay,
                                   int step, float ema_rate, float grad_scale, hipStream_t stream) {
  return adamw_launch(p, g, m, v, ema, n, n_decay, lr, b1, b2, eps, weight_decay, step, ema_rate, grad_scale, nullptr,
                      stream);
}

// The same step with the gradient additionally multiplied by grad_scale_dev[0], a factor that lives on the device
// (mulan_global_norm_clip): optax.chain(clip_by_global_norm, adamw) of ldm/experiment.py:176-178 without a host sync.
MULAN_API int mulan_adamw_ema_step_scaled(float* p, const float* g, float* m, float* v, float* ema, size_t n,
                                          size_t n_decay, float lr, float b1, float b2, float eps, float weight_decay,
                                          int step, float ema_rate, float grad_scale, const float* grad_scale_dev,
                                          hipStream_t stream) {
  if (!grad_scale_dev) return (int)hipErrorInvalidValue;
  return adamw_launch(p, g, m, v, ema, n, n_decay, lr, b1, b2, eps, weight_decay, step, ema_rate, grad_scale,
                      grad_scale_dev, stream);
}

MULAN_API size_t mulan_global_norm_clip_workspace(void) { return (size_t)kNormParts * sizeof(double); }

// out[0] = min(1, clip / norm), out[1] = norm, norm = pre_scale * ||g||_2  (optax.clip_by_global_norm)
MULAN_API int mulan_global_norm_clip(const float* g, size_t n, float clip, float pre_scale, void* workspace, float* out,
                                     hipStream_t stream) {
  if (n == 0 || !(clip > 0.f) || !workspace || !out) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(sumsq_kernel, dim3(kNormParts), dim3(256), 0, stream, g, n, static_cast<double*>(workspace));
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(64), 0, stream, static_cast<const double*>(workspace), clip,
                     pre_scale, out);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_randn(float* out, size_t n, unsigned long long seed, unsigned long long offset,
                          hipStream_t stream) {
  size_t blocks = (((n + 3) >> 2) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, n, seed, offset);
  MULAN_CHECK_LAUNCH();
}

int g_mulan_tune[32] = {0};
unsigned long long* g_mulan_debug_buffer = nullptr;

MULAN_API int mulan_set_debug_buffer(void* dev_ptr) {
  g_mulan_debug_buffer = static_cast<unsigned long long*>(dev_ptr);
  return 0;
}

MULAN_API int mulan_set_tuning(int key, int value) {
  if (key < 0 || key >= 32) return (int)hipErrorInvalidValue;
  g_mulan_tune[key] = value;
  return 0;
}

MULAN_API int mulan_event_create(void** event) {
  if (!event) return (int)hipErrorInvalidValue;
  hipEvent_t e;
  const hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  *event = r == hipSuccess ? static_cast<void*>(e) : nullptr;
  return (int)r;
}

MULAN_API int mulan_event_destroy(void* event) { return (int)hipEventDestroy(static_cast<hipEvent_t>(event)); }

// On a capturing stream: an event-record NODE of the graph being captured, depending on everything `stream` has
// captured so far (every replay records the event when that node runs; work outside the graph waits for it with
// mulan_stream_wait_event after the launch).  Added through the capture's own graph handle (hipStreamGetCaptureInfo_v2 +
// hipGraphAddEventRecordNode): hipEventRecordWithFlags(..., hipEventRecordExternal) returns hipErrorInvalidValue in the
// HIP runtime torch 2.10+rocm7.0 ships.
// `chain` (optional, a stream of the same capture): the node also becomes a dependency of whatever `chain` captures next
// (hipStreamUpdateCaptureDependencies, add).  A leaf node is placed by HIP's graph executor wherever its scheduler likes --
// measured: five such leaves planted along the backward pass all fired together, when the main branch had drained, 4.7 ms
// before the end of a 78 ms graph -- whereas a node that is a link of a branch's own chain runs at its place in that
// branch.  The caller names the branch that reaches the point LAST (the weight-gradient stream), so nothing waits for the
// node that would not have waited anyway.  On a stream that is not capturing: a plain hipEventRecord.
MULAN_API int mulan_event_record_external(void* event, hipStream_t stream, hipStream_t chain) {
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  hipError_t e = hipStreamGetCaptureInfo_v2(stream, &status, &id, &graph, &deps, &ndeps);
  if (e != hipSuccess) return (int)e;
  if (status != hipStreamCaptureStatusActive) return (int)hipEventRecord(static_cast<hipEvent_t>(event), stream);
  hipGraphNode_t node;
  e = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, static_cast<hipEvent_t>(event));
  if (e != hipSuccess || !chain) return (int)e;
  hipStreamCaptureStatus cstatus = hipStreamCaptureStatusNone;
  e = hipStreamIsCapturing(chain, &cstatus);
  if (e != hipSuccess || cstatus != hipStreamCaptureStatusActive) return (int)e;
  return (int)hipStreamUpdateCaptureDependencies(chain, &node, 1, hipStreamAddCaptureDependencies);
}

// ---- signals: a word of signal memory that a KERNEL NODE of a captured graph sets and a stream outside the graph waits
// for (hipStreamWaitValue32: the command processor polls the word, no CU is held).  Event-record nodes turned out not to
// be a usable hand-off from a multi-branch graph on this runtime: planted along the backward pass of the train step they
// all fire together near the end of the graph, whichever branch they are chained into (tools/overlap_timing_probe.py);
// a kernel node runs at its place in its branch like every other kernel of the step.
__global__ void signal_set_kernel(unsigned* sig, const unsigned* __restrict__ value) {
  // everything this branch wrote before is visible device-wide at the kernel boundary in front of this launch; the word
  // itself is written through to memory for the command processor
  sig[1] = (unsigned)__builtin_amdgcn_s_memrealtime();     // (dev: when it ran, 10 ns ticks; read by tools/overlap_timing_probe.py)
  __hip_atomic_store(sig, value[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

MULAN_API int mulan_signal_create(void** sig) {
  if (!sig) return (int)hipErrorInvalidValue;
  int dev = 0, ok = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&ok, hipDeviceAttributeCanUseStreamWaitValue, dev);
  if (e != hipSuccess) return (int)e;
  if (!ok) return (int)hipErrorNotSupported;
  void* p = nullptr;
  e = hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory);
  if (e != hipSuccess) return (int)e;
  e = hipMemset(p, 0, 8);
  *sig = e == hipSuccess ? p : nullptr;
  return (int)e;
}

MULAN_API int mulan_signal_destroy(void* sig) { return (int)hipFree(sig); }

// sig <- value_dev[0] by a one-thread kernel on `stream` (inside a capture: a kernel node); value_dev: device memory the
// caller writes before every replay (a step counter), so that a replayed graph signals a fresh value each time
MULAN_API int mulan_signal_set(void* sig, const unsigned* value_dev, hipStream_t stream) {
  if (!sig || !value_dev) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(signal_set_kernel, dim3(1), dim3(1), 0, stream, static_cast<unsigned*>(sig), value_dev);
  MULAN_CHECK_LAUNCH();
}

// (diagnostic, synchronous) the two words of a signal: [0] the last value stored, [1] the 10 ns clock stamp of that store
MULAN_API int mulan_signal_read(void* sig, unsigned* out2) {
  return (int)hipMemcpy(out2, sig, 8, hipMemcpyDeviceToHost);
}

// everything enqueued on `stream` after this call waits until *sig >= value
MULAN_API int mulan_stream_wait_signal(hipStream_t stream, void* sig, unsigned value) {
  return (int)hipStreamWaitValue32(stream, sig, value, hipStreamWaitValueGte, 0xffffffffu);
}

MULAN_API int mulan_stream_wait_event(hipStream_t stream, void* event) {
  return (int)hipStreamWaitEvent(stream, static_cast<hipEvent_t>(event), 0);
}

MULAN_API const char* mulan_version(void) { return "mulan_hip 0.1 (gfx950, fp32 MFMA)"; }
