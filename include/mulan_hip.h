/* libmulan_hip.so -- C ABI of the MI355X (gfx950) MuLAN training / eval-BPD hot path.
 *
 * The reference (s-sahoo/MuLAN) has no native layer: every device op is whatever XLA lowers from
 * JAX/Flax.  This header is therefore the boundary a maintainer would bind (ctypes / XLA custom-call)
 * to run the path on MI355X; each entry point names the reference code it replaces (file:line are
 * relative to the reference checkout).  See INTEGRATION.md for the binding stubs.
 *
 * Conventions
 *   - all tensors are contiguous fp32 device buffers unless stated; images are NHWC with W == 32 and
 *     H*W == 1024 (the model never resamples: ldm/model_vdm.py:353-371), matrices are row-major;
 *   - the caller owns every buffer including workspaces; the compute entry points never allocate, free or
 *     synchronise (the only exceptions are the handle calls of the last section: mulan_event_create / _destroy and
 *     mulan_signal_create / _destroy own an event / an 8-byte signal word, mulan_signal_read is a synchronous diagnostic);
 *     every call is asynchronous on `stream` and re-entrant; the production entry points read no
 *     process-global state (mulan_set_tuning / mulan_set_debug_buffer are developer switches for kernel-variant A/B
 *     runs and timing probes: all zero by default, never set by the product path);
 *   - the return value is a hipError_t as int (0 == hipSuccess); no exceptions cross the boundary;
 *   - `hipStream_t` is passed as an opaque pointer so plain C callers need no HIP headers.
 */
#ifndef MULAN_HIP_H_
#define MULAN_HIP_H_

#include <stddef.h>

/* (the library's own sources define MULAN_STREAM_T and typedef mulan_stream_t as hipStream_t before they include
 * this header, so that the compiler checks every definition against the declaration below) */
#ifndef MULAN_STREAM_T
typedef void* mulan_stream_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

const char* mulan_version(void);
/* developer knob for kernel-variant A/B runs (tools/kbench.py); never needed in production */
int mulan_set_tuning(int key, int value);
/* dev-only: device buffer (>= 64 x u64) that receives s_memtime stamps of block 0; NULL (default) disables */
int mulan_set_debug_buffer(void* dev_ptr);

/* ---- a point INSIDE a captured HIP graph that work OUTSIDE it can wait for ----------------------------------------
 * The reference's pmap(scan(train_step)) overlaps lax.pmean of the gradients with the rest of the backward pass inside
 * one compiled program (ldm/experiment.py:89-95,341).  Here the train step is a replayed HIP graph and the RCCL
 * all-reduce of gradient bucket k is issued outside it (torch.distributed): it may start as soon as the captured backward
 * pass has produced that bucket.  mulan_event_record_external on a capturing stream becomes an event-record NODE
 * (hipGraphAddEventRecordNode on the capture's own graph, behind everything the stream has captured so far) instead of an
 * internal fork / join edge; `chain` (optional): a stream of the same capture whose next captured work shall depend on the
 * node, which makes the node a link of that branch instead of a leaf the executor may run late; after hipGraphLaunch the
 * collective's stream waits for it with mulan_stream_wait_event.
 * Outside a capture the pair is an ordinary record / wait.  (torch.cuda.Event(external=True) is refused on ROCm builds
 * of torch 2.10 and hipEventRecordWithFlags(hipEventRecordExternal) returns hipErrorInvalidValue in the runtime it ships,
 * hence these four; tools/ext_event_probe.py, profiles/r04_ext_event_probe.log.) */
int mulan_event_create(void** event);
int mulan_event_destroy(void* event);
int mulan_event_record_external(void* event, mulan_stream_t stream, mulan_stream_t chain /* nullable */);
int mulan_stream_wait_event(mulan_stream_t stream, void* event);
/* The hand-off the replayed multi-rank train step uses (round 4; the event-record nodes above fire late in a multi-branch
 * graph on this runtime): a word of signal memory (hipMallocSignalMemory) set by a one-thread KERNEL NODE at its place in
 * the captured backward pass -- mulan_signal_set stores value_dev[0], a step counter the caller writes before each
 * replay -- and waited for by the collective's stream with hipStreamWaitValue32 (*sig >= value; the command processor
 * polls, no CU is held).  mulan_signal_create returns hipErrorNotSupported where the device cannot wait on values. */
int mulan_signal_create(void** sig);
int mulan_signal_destroy(void* sig);
int mulan_signal_set(void* sig, const unsigned* value_dev, mulan_stream_t stream);
int mulan_stream_wait_signal(mulan_stream_t stream, void* sig, unsigned value);
/* diagnostic, synchronous: out2[0] = the value last stored, out2[1] = the 10 ns clock stamp of that store */
int mulan_signal_read(void* sig, unsigned* out2);

/* ---- 3x3 SAME convolution, NHWC, HWIO weights [3,3,C,N] --------------------------------------
 * flax nn.Conv(kernel_size=(3,3)) in ResnetBlock conv1/conv2 (ldm/model_vdm.py:633-634,645-650;
 * ldm/ldm_unet.py:33-34,49-54), conv_in / conv_out (model_vdm.py:348-349,378-383;
 * ldm/model_mulan_epsilon.py:125-126,146-151).
 * y = conv(x, w) + bias + cbias + res.  cbias_mode 1: cbias is [B,N] (per-sample FiLM bias,
 * model_vdm.py:639-641); 2: cbias is [B,H,W,N] (per-pixel, ldm_unet.py:38-45); 0/NULL: none.
 * res (nullable) is the residual/shortcut branch added in the epilogue (model_vdm.py:656, :386). */
int mulan_conv3x3_fwd(const float* x, const float* w, const float* bias, const float* cbias, int cbias_mode,
                      const float* res, float* y, int B, int H, int W, int C, int N, mulan_stream_t stream);
/* wT[t][n][c] = w[8-t][c][n]: input gradient = mulan_conv3x3_fwd(dy, wT) with C and N swapped
 * (autodiff of the above under jax.value_and_grad, ldm/experiment.py:339). */
int mulan_conv3x3_wflip(const float* w, float* wT, int C, int N, mulan_stream_t stream);
/* dw[3,3,C,N] (+)= sum_{b,h,w} x (x) dy.  workspace: mulan_conv3x3_wgrad_workspace() bytes. */
size_t mulan_conv3x3_wgrad_workspace(int B, int H, int W, int C, int N);
int mulan_conv3x3_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int H, int W, int C,
                        int N, int accumulate, mulan_stream_t stream);

/* fp32-equivalent fast path on the bf16 matrix cores (6-pass split, = XLA's float32 / HIGHEST matmul precision that
 * the reference requests in ldm/main.py:39).  wp = weights pre-split by mulan_conv3x3_pack_bf16x6 (flip = 1 packs the
 * tap-flipped, channel-transposed weights of the input-gradient convolution).  Needs C % 16 == 0 and N % 128 == 0. */
size_t mulan_conv3x3_pack_bf16x6_bytes(int C, int N);
int mulan_conv3x3_pack_bf16x6(const float* w, void* wp, int C, int N, int flip, mulan_stream_t stream);
int mulan_conv3x3_fwd_bf16x6(const float* x, const void* wp, const float* bias, const float* cbias, int cbias_mode,
                             const float* res, float* y, int B, int H, int W, int C, int N, mulan_stream_t stream);

size_t mulan_conv3x3_wgrad_bf16x6_workspace(int B, int H, int W, int C, int N);
int mulan_conv3x3_wgrad_bf16x6(const float* x, const float* dy, float* dw, float* workspace, int B, int H, int W,
                               int C, int N, int accumulate, mulan_stream_t stream);

/* fp32-equivalent fast path on the fp16 matrix cores (3-pass split of power-of-two-scaled operands into two fp16
 * pieces each; same contract as above at half the matrix-core cycles).  mulan_absmax_rows gives the per-image
 * maxima from which the kernels derive the operand scales: out[r][0..15] = fp32 bit patterns of 16 partial maxima of
 * |x[r, 0:row_len]| (consumers take the maximum of a row's 16 entries; no atomics, no zero-fill pass).
 * wmax[16] = mulan_absmax_rows(w, 1 row); wp from mulan_conv3x3_pack_f16x3 (flip as above).
 * Needs C % 16 == 0 and N % 128 == 0. */
int mulan_absmax_rows(const float* x, unsigned* out, int rows, size_t row_len, mulan_stream_t stream);
/* c = a + b (c may alias a or b) with out = mulan_absmax_rows(c) in the same pass: the gradient sum autograd forms for
 * a tensor with two consumers (the U-Net skip connections, model_vdm.py:351-372) fused with the maxima pass of the
 * convolution that receives it */
int mulan_add_absmax_rows(const float* a, const float* b, float* c, unsigned* out, int rows, size_t row_len,
                          mulan_stream_t stream);
/* ... and with the column sums of c as a second by-product: each row is a [row_len / ncols, ncols] matrix (pixels x
 * channels), colpart [rows][16][ncols] receives 16 partial column sums per row; the sum of the rows * 16 vectors
 * (mulan_colsum) is the bias gradient of the convolution whose output gradient c is (autodiff of the bias add of
 * model_vdm.py:633-656 for a block output with two consumers), without a second pass over c.  256 % (ncols / 4) == 0. */
int mulan_add_absmax_rows_colsum(const float* a, const float* b, float* c, unsigned* out, float* colpart, int rows,
                                 size_t row_len, int ncols, mulan_stream_t stream);
size_t mulan_conv3x3_pack_f16x3_bytes(int C, int N);
int mulan_conv3x3_pack_f16x3(const float* w, void* wp, const unsigned* wmax, int C, int N, int flip,
                             mulan_stream_t stream);
int mulan_conv3x3_fwd_f16x3(const float* x, const unsigned* xmax, const void* wp, const unsigned* wmax,
                            const float* bias, const float* cbias, int cbias_mode, const float* res, float* y, void* xs,
                            unsigned* ymax, int B, int H, int W, int C, int N, mulan_stream_t stream);
/* ... with the caller's word on what else runs (round 5): alone != 0 = no other stream's kernels share the chip with this
 * launch (a forward pass, an evaluator, the vector-Jacobian product of the ODE likelihood, notebook_utils.py:193-373);
 * launches of at most 256 short-tile blocks then run as k-split blocks of eight waves.  mulan_conv3x3_fwd_f16x3 = alone 0. */
int mulan_conv3x3_fwd_f16x3_alone(const float* x, const unsigned* xmax, const void* wp, const unsigned* wmax,
                                  const float* bias, const float* cbias, int cbias_mode, const float* res, float* y,
                                  void* xs, unsigned* ymax, int alone, int B, int H, int W, int C, int N,
                                  mulan_stream_t stream);
/* The forward convolution fed with the split planes of its input (written by mulan_groupnorm_fwd_planes; xmax = the
 * [B][16] array that call filled).  H % 8 == 0, C % 32 == 0, N % 128 == 0.  ldm/model_vdm.py:633-656. */
int mulan_conv3x3_fwd_f16x3_planes_in(const void* xplanes, const unsigned* xmax, const void* wp, const unsigned* wmax,
                                      const float* bias, const float* cbias, int cbias_mode, const float* res, float* y,
                                      unsigned* ymax, int B, int H, int W, int C, int N, mulan_stream_t stream);
/* ... and (round 5) with the by-product the NEXT GroupNorm needs: ystats (optional) receives the partial sums of y and
 * y^2 per (image, row tile of this launch, channel quad), [B][H / mulan_conv3x3_f16x3_tile_rows(B, H, N, ymax != NULL)]
 * [N / 4][2] floats -- mulan_groupnorm_fwd_stream forms mean / rstd from them (norm2 behind conv1, the next block's
 * norm1 behind conv2: ldm/model_vdm.py:622-644), so that pass needs no statistics phase.  alone != 0: the caller
 * vouches that no other stream's kernels share the chip with this launch (a forward pass, an evaluator); launches of at
 * most 256 blocks then run as k-split blocks of eight waves that take a whole CU's LDS (two waves per SIMD where a
 * 4-wave block would sit alone on its CU).  The GroupNorm-fed entry point below always may. */
int mulan_conv3x3_fwd_f16x3_planes_in_stats(const void* xplanes, const unsigned* xmax, const void* wp, const unsigned* wmax,
                                            const float* bias, const float* cbias, int cbias_mode, const float* res,
                                            float* y, unsigned* ymax, float* ystats, int alone, int B, int H, int W,
                                            int C, int N, mulan_stream_t stream);
/* GroupNorm (+ swish) normalised inside the convolution that consumes it (round 3; ResnetBlock norm1 + swish -> conv1,
 * norm2 + swish -> conv2 wherever no dropout is drawn: ldm/model_vdm.py:622-623,632-633,643-650 under eval_step /
 * sample / the likelihood evaluators).  mulan_groupnorm_stats reads x once and leaves mean / rstd [B, G] and the a-priori
 * bound of |y| ([B][16], maxima format) -- what mulan_groupnorm_fwd_planes computes, same summation order -- and
 * mulan_conv3x3_fwd_f16x3_gn_in normalises, activates and splits x1 (, x2: virtual channel concat, C2 == C1) while it
 * fills its patches: the result is bit for bit that of mulan_groupnorm_fwd_planes + ..._planes_in, the normalised tensor
 * never reaches HBM (yplanes_out, optional: store it as the weight-gradient operand after all).  C1 + C2 <= 512.
 * Statistics handed from convolution to convolution: ystats (optional output, [B][H / rows][N / 4][2], rows =
 * mulan_conv3x3_f16x3_tile_rows) receives this launch's partial sums of y and y^2 per image, row tile and channel quad;
 * given as xstats1 (, xstats2; xstats_tiles = their H / rows) to the next
 * launch they replace the statistics pass -- every block forms mean / rstd / bound itself and mean, rstd, bound become
 * outputs (same formulas, another summation order: agreement to fp32 rounding instead of bit for bit). */
int mulan_groupnorm_stats(const float* x1, const float* x2, int C1, int C2, const float* gamma, const float* beta,
                          float* mean, float* rstd, unsigned* bound, int B, int hw, int G, float eps,
                          mulan_stream_t stream);
int mulan_conv3x3_fwd_f16x3_gn_in(const float* x1, const float* x2, int C1, int C2, const float* gamma,
                                  const float* beta, float* mean, float* rstd, int G, int act, float eps,
                                  unsigned* bound, const float* xstats1, const float* xstats2, int xstats_tiles,
                                  const void* wp, const unsigned* wmax, const float* bias, const float* cbias,
                                  int cbias_mode, const float* res, float* y, unsigned* ymax, float* ystats,
                                  void* yplanes_out, int B, int H, int W, int N, mulan_stream_t stream);
/* Image rows per block (8, 4 or 2) the two-blocks-per-CU convolution kernel uses for a launch of B images and N output
 * channels (round 5: launches of fewer than 128 images get shorter tiles so that every CU still holds two blocks -- the
 * 64 images per GPU of BASELINE configs[2] at 8 GPUs, the 16 of a sampling batch).  The statistics arrays follow the
 * tile: ystats is [B][H / rows][N / 4][2], and the consumer is told the producer's count through xstats_tiles. */
int mulan_conv3x3_f16x3_tile_rows(int B, int H, int N, int with_ymax);

size_t mulan_conv3x3_wgrad_f16x3_workspace(int B, int H, int W, int C, int N);
int mulan_conv3x3_wgrad_f16x3(const float* x, const unsigned* xmax, const float* dy, const unsigned* dymax, float* dw,
                              float* workspace, int B, int H, int W, int C, int N, int accumulate,
                              mulan_stream_t stream);

/* The convolution above can hand its split operand on: xs (optional, mulan_conv3x3_planes_bytes) receives the two
 * scaled fp16 planes of x as [B][C/16][H*W][plane][16].  The weight gradient below consumes the planes of the forward
 * input (xs, from the forward call) and of the output gradient (dys, from the input-gradient call, i.e. the same
 * kernel run on dy with flip = 1 weights), so the fp32 -> 2 x fp16 split is done once per tensor.  xmax / dymax are
 * the per-image maxima the planes were scaled with.  Needs C % 128 == 0 and N % 128 == 0.
 * ymax (optional, [B][16]): the maxima of the convolution's own output, for whichever f16x3 kernel reads it next. */
size_t mulan_conv3x3_planes_bytes(int B, int H, int W, int C);
/* share_chip (0 / 1; the same value for the workspace query and the launch): 1 = the caller runs this launch beside
 * another stream's matrix-core kernels (the train step's weight-gradient stream beside the input-gradient chain): the
 * launch then uses about half as many blocks, so that both streams keep running side by side (profiles/DESIGN_r04.md 3.2). */
size_t mulan_conv3x3_wgrad_f16x3_planes_workspace(int B, int H, int W, int C, int N, int share_chip);
int mulan_conv3x3_wgrad_f16x3_planes(const void* xs, const unsigned* xmax, const void* dys, const unsigned* dymax,
                                     float* dw, float* workspace, int B, int H, int W, int C, int N, int accumulate,
                                     int share_chip, mulan_stream_t stream);
/* Slab reductions folded into the next weight-gradient launch (round 6).  The plane-fed weight gradient splits the
 * pixels over S blocks per output tile and sums their partial results ("slabs", workspace [S][9][C][N]) in a launch of
 * its own.  In the backward pass of a train step the weight gradients follow one another on one stream, so
 * mulan_conv3x3_wgrad_f16x3_planes_fold writes its slabs WITHOUT summing them and instead sums the slabs of up to two
 * EARLIER launches (`pending`, host array; complete by stream order) in the prologue of its blocks, in slab order -- the
 * very bits of the separate reduction -- while their first operand loads are in flight.  The caller keeps the record
 * {workspace, dw, S = _splits(...), E = 9 C N, accumulate} of each _fold launch and hands it to a later _fold launch
 * or, for the last ones of a pass, to mulan_slab_reduce.  (autodiff of ldm/model_vdm.py:633-650 under
 * ldm/experiment.py:339; same arguments otherwise as mulan_conv3x3_wgrad_f16x3_planes) */
typedef struct mulan_slab_reduction {
  const float* slab;   /* [S][E] partial results */
  float* out;          /* [E]; 16-byte aligned like slab */
  int S, E;            /* E % 4 == 0 */
  int accumulate;      /* out += sum instead of out = sum */
  int reserved;
} mulan_slab_reduction;
int mulan_conv3x3_wgrad_f16x3_planes_splits(int B, int H, int W, int C, int N, int share_chip);
int mulan_conv3x3_wgrad_f16x3_planes_fold(const void* xs, const unsigned* xmax, const void* dys, const unsigned* dymax,
                                          float* workspace, int B, int H, int W, int C, int N, int share_chip,
                                          const mulan_slab_reduction* pending, int n_pending, mulan_stream_t stream);
int mulan_slab_reduce(const float* slab, float* out, int S, int E, int accumulate, mulan_stream_t stream);


/* Once-per-step weight preparation for all eligible parameter leaves of the flat parameter buffer in two launches
 * (instead of maxima + pack per layer and direction).  Leaf record = 8 x int64: element offset in `flat`, kind (0: 3x3
 * conv [3,3,C,N], 1: dense [C,N]), C, N, byte offset of the forward operand in `packed` or -1, byte offset of the
 * gradient operand (tap-flipped / transposed) or -1, number of elements, 0.  Formats as produced by
 * mulan_absmax_rows / mulan_conv3x3_pack_f16x3 / mulan_linear_pack_f16x3. */
int mulan_param_maxima(const float* flat, const long long* leaves, int n, unsigned* out, mulan_stream_t stream);
int mulan_param_pack_f16x3(const float* flat, const long long* leaves, int n, const unsigned* maxima, void* packed,
                           mulan_stream_t stream);

/* ---- per-pixel dense layers on the fp16 matrix cores (f16x3 scheme) -------------------------------
 *   y[M, N1 | N2] = [x1 | x2][M, K1 + K2] @ W + bias + res
 * with a virtual channel concat on the input side and a split on the output side: nin_shortcut on
 * concat[h, skip] (model_vdm.py:369,652-653) and its input gradient, q / k / v / proj_out of AttnBlock
 * (model_vdm.py:676-685).  Rows are pixels (rows_per_img per image); x1max / x2max are per-image maxima in the
 * mulan_absmax_rows format; wp from mulan_linear_pack_f16x3 (transpose = 1 packs w^T for dx = dy @ w^T);
 * wmax[16] = mulan_absmax_rows(w, 1 row).  Needs M % 128 == 0, rows_per_img % 128 == 0, K1, K2 % 32 == 0,
 * N1, N2 % 128 == 0 (K2 = 0 / N2 = 0 without x2 / y2); res only with N2 = 0. */
size_t mulan_linear_pack_f16x3_bytes(int K, int N);
int mulan_linear_pack_f16x3(const float* w, void* wp, const unsigned* wmax, int K, int N, int transpose,
                            mulan_stream_t stream);
int mulan_linear_f16x3(const float* x1, const unsigned* x1max, const float* x2, const unsigned* x2max, int K1, int K2,
                       const void* wp, const unsigned* wmax, const float* bias, const float* res, float* y1, float* y2,
                       void* xs, int N1, int N2, int M, int rows_per_img, mulan_stream_t stream);
/* xs (optional, M (K1+K2) 4 bytes): the layer hands the split planes of [x1 | x2] on (scaled per image with the larger
 * of the two maxima).  Its weight gradient dw[K1+K2, N] (+)= [x1|x2]^T dy then comes from those planes and the planes of
 * dy handed on by the convolution that consumed the same dy (nin_shortcut and conv2 of a ResnetBlock share dy):
 * H x W = 32 x 32 pixels per image; C = K1 + K2 and N multiples of 128; xmax = elementwise max of x1max, x2max. */
size_t mulan_linear_wgrad_f16x3_planes_workspace(int B, int H, int W, int C, int N, int share_chip);
int mulan_linear_wgrad_f16x3_planes(const void* xs, const unsigned* xmax, const void* dys, const unsigned* dymax,
                                    float* dw, float* workspace, int B, int H, int W, int C, int N, int accumulate,
                                    int share_chip, mulan_stream_t stream);
/* The same weight gradient with the layer's input in fp32 (round 3): [x1 | x2] ([B,1024,C1], [B,1024,C2]; x2 / C2 may be
 * NULL / 0) is read as it is and split while staged -- the kernel is memory bound, the split is free --, so the
 * forward call need not hand planes on (xs = NULL above: 134 MB less written per nin_shortcut at E = 128).  xmax, xmax2:
 * the maxima of x1 and x2 as mulan_linear_f16x3 took them (the concat is scaled with their elementwise max; xmax2 unused
 * without x2).  Bit-identical to the planes form.  C1, C2, N % 128 == 0. */
size_t mulan_linear_wgrad_f16x3_x32_workspace(int B, int H, int W, int C, int N, int share_chip);
int mulan_linear_wgrad_f16x3_x32(const float* x1, const float* x2, int C1, int C2, const unsigned* xmax,
                                 const unsigned* xmax2, const void* dys, const unsigned* dymax, float* dw,
                                 float* workspace, int B, int H, int W, int N, int accumulate, int share_chip,
                                 mulan_stream_t stream);
/* The attention products (lax.dot_general in dot_product_attention, model_vdm.py:775-796, and their autodiff) on the
 * same kernels, one operand per image:  y[b] = x[b] @ W[b] (+ res) with W[b] packed by the batched pack (w: batch
 * operands [K, N], or [N, K] with transpose = 1; wmax [batch][16] = mulan_absmax_rows(w, batch rows)) at
 * wp + b * K * N * 4 bytes;  out[b] = x[b]^T @ dy[b] ([C, N] per image) from the planes the forward products hand on
 * (S = Q K^T, O = P V, dP = dO V^T, dQ = dS K by the first form; dV = P^T dO, dK = dS^T Q by the second). */
int mulan_linear_pack_f16x3_batched(const float* w, void* wp, const unsigned* wmax, int K, int N, int transpose,
                                    int batch, mulan_stream_t stream);
int mulan_linear_f16x3_batched(const float* x, const unsigned* xmax, int K, const void* wp, const unsigned* wmax,
                               const float* res, float* y, void* xs, int N, int M, int rows_per_img,
                               mulan_stream_t stream);
int mulan_bmm_tn_f16x3_planes(const void* xs, const unsigned* xmax, const void* dys, const unsigned* dymax, float* out,
                              int B, int H, int W, int C, int N, mulan_stream_t stream);

/* ---- batched GEMM:  C[b] = alpha * op(A[b]) op(B[b]) + bias[n] + beta * R[b] ------------------
 * nn.Dense / nn.DenseGeneral and lax.dot_general call sites: nin_shortcut (model_vdm.py:652-653),
 * q,k,v,proj_out and the attention products (model_vdm.py:676-685,775-796), dense0/dense1/cond_proj
 * (model_vdm.py:337-338,639-641), gamma MLP (model_mulan_epsilon.py:531-538), encoder head
 * (model_mulan_epsilon.py:153-154).  transA: A stored [K][lda]; transB: B stored [N][ldb]. */
int mulan_gemm(const float* A, const float* B, float* C, const float* bias, const float* R, int M, int N, int K,
               int lda, int ldb, int ldc, int ldr, int transA, int transB, int batch, long long strideA,
               long long strideB, long long strideC, long long strideR, float alpha, float beta, float* workspace,
               mulan_stream_t stream);
/* bytes of split-K workspace mulan_gemm can use for this shape (0: none needed; workspace may be NULL) */
size_t mulan_gemm_workspace(int M, int N, int K, int batch);

/* ---- fused attention core (f16x3), S = 1024 positions, one head ----------------------------------------------
 * C = 128 (the CIFAR width, configs/cifar10-conditioned.py:70) and C = 256 (configs/imagenet32.py:70): forward and
 * backward; B <= 65535.  (C = 256 backward: dq and dk with the output channels split over two blocks per column tile, dv
 * in a launch of its own -- attention_f16x3.hip.)
 * AttnBlock core softmax((q / sqrt(C)) k^T) v and its gradients (model_vdm.py:679-683, 704-802) without the
 * [1024 x 1024] score / probability matrices in HBM.  Operands are the packs of mulan_linear_pack_f16x3_batched:
 * "T" pack of x [B,1024,C]: (K = C, N = 1024, transpose = 1); "N" pack: (K = 1024, N = C, transpose = 0); *max: the
 * per-image maxima ([B][16], mulan_absmax_rows) the packs were scaled with.  alpha = 1 / sqrt(C).
 * forward: o [B,1024,C] and lse [B,1024] = log sum_j exp(alpha q_i k_j).
 * backward: delta = rowsum(do * o) (mulan_attention_delta), dmax = its maxima; dq, dk, dv [B,1024,C]. */
int mulan_attention_fwd_f16x3(const void* qt, const void* kt, const void* vn, const unsigned* qmax, const unsigned* kmax,
                              const unsigned* vmax, float* o, float* lse, int B, int S, int C, float alpha,
                              mulan_stream_t stream);
/* both packs of x [B,1024,C] in one pass (xt and / or xn may be NULL) */
int mulan_attention_pack_f16x3(const float* x, const unsigned* xmax, void* xt, void* xn, int B, int S, int C,
                               mulan_stream_t stream);
int mulan_attention_delta(const float* dout, const float* o, float* delta, int B, int S, int C, mulan_stream_t stream);
int mulan_attention_bwd_f16x3(const void* qt, const void* qn, const void* kt, const void* kn, const void* vt,
                              const void* dot, const void* don, const unsigned* qmax, const unsigned* kmax,
                              const unsigned* vmax, const unsigned* domax, const unsigned* dmax, const float* lse,
                              const float* delta, float* dq, float* dk, float* dv, int B, int S, int C, float alpha,
                              mulan_stream_t stream);

/* ---- GroupNorm (+SiLU) (+dropout), input = virtual channel concat [x1|x2] --------------------
 * nn.GroupNorm() + nn.swish + nn.Dropout in ResnetBlock (model_vdm.py:622-623,632,643-644), final
 * norm (model_vdm.py:376-377), AttnBlock norm (model_vdm.py:672-674).  hw must be 1024.
 * act: 0 none, 1 SiLU.  keep < 1 enables dropout with Philox4x32-10(seed, offset + element/4). */
int mulan_groupnorm_fwd(const float* x1, const float* x2, int C1, int C2, const float* gamma, const float* beta,
                        float* y, float* mean, float* rstd, int B, int hw, int G, float eps, int act, float keep,
                        unsigned long long seed, unsigned long long offset, unsigned* ymax, mulan_stream_t stream);
/* ymax (optional, [B][16]): partial maxima of |y| in the format of mulan_absmax_rows, a by-product that saves the
 * following convolution its own pass over y. */
/* _dyn: seed_dev (optional, device memory) makes the dropout seed `seed ^ seed_dev[0]`, read when the kernel runs: a
 * stream-ordered parameter, so a captured HIP graph (the lax.scan of ldm/experiment.py:89-91: many sub-steps per host
 * dispatch) replays with a fresh seed per step. */
int mulan_groupnorm_fwd_dyn(const float* x1, const float* x2, int C1, int C2, const float* gamma, const float* beta,
                            float* y, float* mean, float* rstd, int B, int hw, int G, float eps, int act, float keep,
                            unsigned long long seed, unsigned long long offset, const unsigned long long* seed_dev,
                            unsigned* ymax, mulan_stream_t stream);
/* The same with the output written as the split fp16 operand planes of the f16x3 convolution that consumes it instead
 * of as fp32 ([B][C/16][hw][plane][16], mulan_conv3x3_planes_bytes): the convolution (mulan_conv3x3_fwd_f16x3_planes_in)
 * then neither splits its input nor stores planes, and its weight-gradient kernel takes the same tensor.  The planes are
 * scaled by an a-priori bound of |y| -- (sqrt(hw * C / G) max|gamma| + max|beta|) / keep -- which ymax [B][16] receives in
 * the maxima format.  ldm/model_vdm.py:622-623,632,643-644 (norm + swish + dropout in front of conv1 / conv2). */
int mulan_groupnorm_fwd_planes(const float* x1, const float* x2, int C1, int C2, const float* gamma, const float* beta,
                               void* yplanes, float* mean, float* rstd, int B, int hw, int G, float eps, int act,
                               float keep, unsigned long long seed, unsigned long long offset,
                               const unsigned long long* seed_dev, unsigned* ymax, mulan_stream_t stream);
int mulan_groupnorm_bwd_dyn(const float* dy, const float* x1, const float* x2, int C1, int C2, const float* gamma,
                            const float* beta, const float* mean, const float* rstd, float* dx1, float* dx2,
                            float* dgamma_part, float* dbeta_part, int B, int hw, int G, int act, float keep,
                            unsigned long long seed, unsigned long long offset, const unsigned long long* seed_dev,
                            int accumulate, unsigned* dx1max, unsigned* dx2max, const float* add1, const float* add2,
                            float* dxsum_part, mulan_stream_t stream);
/* dgamma_part / dbeta_part are [B, C1+C2] per-sample partials (reduce with mulan_colsum). */
int mulan_groupnorm_bwd(const float* dy, const float* x1, const float* x2, int C1, int C2, const float* gamma,
                        const float* beta, const float* mean, const float* rstd, float* dx1, float* dx2,
                        float* dgamma_part, float* dbeta_part, int B, int hw, int G, int act, float keep,
                        unsigned long long seed, unsigned long long offset, int accumulate, unsigned* dx1max,
                        unsigned* dx2max, const float* add1, const float* add2, float* dxsum_part,
                        mulan_stream_t stream);
/* _fused: the reduction of the per-sample partials over the samples happens inside the launch (the last block of each
 * 32-channel slab sums them in a fixed order): dgamma / dbeta [C1+C2] receive the totals, dxsum (optional) the sum over
 * samples of dxsum_part's x1 columns [C1] = the bias gradient of the convolution whose output gradient dx1 is, dxsum2
 * (optional) a second copy of it (a shortcut layer's bias that sees the same gradient).  tickets: [16] unsigned, zero
 * before the first launch on a stream (each launch leaves them zero).  add1b (optional, round 3): a second outside
 * gradient of x1, dx1 = (dx1 + add1) + add1b -- the gradient a block output receives through its U-Net skip connection
 * (autodiff of model_vdm.py:351-372), added here instead of by a kernel of its own. */
int mulan_groupnorm_bwd_fused(const float* dy, const float* x1, const float* x2, int C1, int C2, const float* gamma,
                              const float* beta, const float* mean, const float* rstd, float* dx1, float* dx2,
                              float* dgamma_part, float* dbeta_part, int B, int hw, int G, int act, float keep,
                              unsigned long long seed, unsigned long long offset, const unsigned long long* seed_dev,
                              unsigned* dx1max, unsigned* dx2max, const float* add1, const float* add2,
                              const float* add1b, float* dxsum_part, float* dgamma, float* dbeta, float* dxsum,
                              float* dxsum2, unsigned* tickets, mulan_stream_t stream);
/* _fused_planes (round 3; autodiff of ldm/model_vdm.py:643-650, the gradient norm2 hands to conv1): single input, no
 * skip-path gradient; dx is written ONLY as the split fp16 operand planes of the f16x3 kernels of the convolution in
 * front ([B][C/16][1024][plane][16], mulan_conv3x3_planes_bytes bytes) -- its input-gradient convolution
 * (mulan_conv3x3_fwd_f16x3_planes_in on the tap-flipped weights) and its weight gradient
 * (mulan_conv3x3_wgrad_f16x3_planes) -- scaled per image with an a-priori bound of |dx[b]| derived from dymax ([B][16]
 * maxima of dy, a by-product of the convolution kernel that wrote dy), rstd and max|gamma|; dxmax [B][16] receives the
 * bound in the maxima format.  Bias / FiLM gradients of that convolution come from dxsum_part / dxsum as before. */
int mulan_groupnorm_bwd_fused_planes(const float* dy, const unsigned* dymax, const float* x, int C, const float* gamma,
                                     const float* beta, const float* mean, const float* rstd, void* dxplanes,
                                     float* dgamma_part, float* dbeta_part, int B, int hw, int G, int act, float keep,
                                     unsigned long long seed, unsigned long long offset,
                                     const unsigned long long* seed_dev, unsigned* dxmax, float* dxsum_part,
                                     float* dgamma, float* dbeta, float* dxsum, float* dxsum2, unsigned* tickets,
                                     const unsigned* keepbits, mulan_stream_t stream);
/* keepbits (optional): the dropout keep-bits as the forward pass drew them -- written by
 * mulan_groupnorm_fwd_planes_keepbits (mulan_groupnorm_fwd_planes + one more output of B * C / 32 * 1024 unsigned) --, so
 * that the backward kernel does not repeat the 10 Philox rounds per float4 (same bits either way: nn.Dropout's mask of
 * model_vdm.py:644 is a function of (seed, element) only). */
int mulan_groupnorm_fwd_planes_keepbits(const float* x1, const float* x2, int C1, int C2, const float* gamma,
                                        const float* beta, void* yplanes, float* mean, float* rstd, int B, int hw, int G,
                                        float eps, int act, float keep, unsigned long long seed, unsigned long long offset,
                                        const unsigned long long* seed_dev, unsigned* ymax, unsigned* keepbits,
                                        mulan_stream_t stream);
/* Streaming forms (round 5; the same layers, model_vdm.py:622-623,632,643-644 and their autodiff): the reductions that
 * force the kernels above to hold a whole (sample, 32-channel slab) before they write -- mean / variance in the forward
 * pass, the two group sums of the backward pass -- arrive from the PRODUCER of the tensor instead: the convolution whose
 * epilogue wrote x leaves sum x, sum x^2 (`ystats` of mulan_conv3x3_fwd_f16x3_planes_in_stats / _gn_in: per image, row
 * tile of that launch and channel quad, [B][xstats_tiles][C / 4][2] floats), the producer of dy would leave sum g gamma,
 * sum g gamma xhat (`gstats`, [B][4][C / 4][2]; no shipped kernel writes them: measured not to pay, DESIGN.md section
 * 3.3).  The GroupNorm kernels are then plain streaming passes (forward 1 read + 1 write, backward 2 reads + 1 write).
 * The training step runs _fwd_stream where it is the faster kernel (dropout layers, the 2 x 128-channel concat).
 * _fwd_stream: exactly one of y (fp32; ymax optional: true maxima) and yplanes (split planes; ymax required: the bound).
 * xstats1 (, xstats2 for C2 > 0) given: mean / rstd [B, G] are OUTPUTS; NULL: they are inputs.  keepbits (optional,
 * keep < 1): as mulan_groupnorm_fwd_planes_keepbits.
 * _bwd_stream: arguments of mulan_groupnorm_bwd_fused plus gstats; dx1planes (optional, then C2 == 0 and no add*):
 * dx1 as split planes (mulan_groupnorm_bwd_fused_planes), dymax required, dx1max receives the bound. */
int mulan_groupnorm_fwd_stream(const float* x1, const float* x2, int C1, int C2, const float* gamma, const float* beta,
                               float* y, void* yplanes, float* mean, float* rstd, const float* xstats1,
                               const float* xstats2, int xstats_tiles, int B, int hw, int G, float eps, int act, float keep,
                               unsigned long long seed, unsigned long long offset, const unsigned long long* seed_dev,
                               unsigned* ymax, unsigned* keepbits, mulan_stream_t stream);
int mulan_groupnorm_bwd_stream(const float* dy, const unsigned* dymax, const float* x1, const float* x2, int C1, int C2,
                               const float* gamma, const float* beta, const float* mean, const float* rstd,
                               const float* gstats, float* dx1, float* dx2, void* dx1planes, float* dgamma_part,
                               float* dbeta_part, int B, int hw, int G, int act, float keep, unsigned long long seed,
                               unsigned long long offset, const unsigned long long* seed_dev, unsigned* dx1max,
                               unsigned* dx2max, const float* add1, const float* add2, const float* add1b,
                               float* dxsum_part, float* dgamma, float* dbeta, float* dxsum, float* dxsum2,
                               unsigned* tickets, const unsigned* keepbits, mulan_stream_t stream);
/* add1 / add2 (optional): gradients arriving through a skip path of x1 / x2 (the ResnetBlock residual, nin_shortcut),
 * added while dx is written, so that no separate accumulation pass exists.  By-products of the written gradients (the
 * dy of the convolution in front): dx1max / dx2max (optional, [B][16], mulan_absmax_rows format) and dxsum_part
 * (optional, [B, C1+C2]: per-sample channel sums = that convolution's per-sample bias gradient). */

/* ---- small fused elementwise / reduction kernels ---------------------------------------------- */
/* kind 1: SiLU (nn.swish); kind 2: shift + softplus (model_mulan_epsilon.py:537). */
int mulan_act_fwd(const float* x, float* y, size_t n, int kind, float shift, mulan_stream_t stream);
int mulan_act_bwd(const float* x, const float* dy, float* dx, size_t n, int kind, mulan_stream_t stream);
/* out[s][c] (+)= sum_{r<seg} x[s*seg + r][c]  (bias and per-sample cond-bias gradients) */
int mulan_colsum(const float* x, float* out, int nseg, int seg, int C, int ld, int accumulate,
                 mulan_stream_t stream);
/* the two column sums of x [2, seg, C] in one launch, each to its own destination (GroupNorm dgamma / dbeta) */
int mulan_colsum_pair(const float* x, float* out0, float* out1, int seg, int C, mulan_stream_t stream);
/* row softmax of the attention weights (model_vdm.py:786) and its backward */
int mulan_softmax_fwd(const float* x, float* y, size_t rows, int cols, mulan_stream_t stream);
int mulan_softmax_bwd(const float* p, const float* dp, float* ds, size_t rows, int cols, mulan_stream_t stream);
/* softmax(alpha x) and its gradient w.r.t. x (the 1/sqrt(C) of dot_product_attention, model_vdm.py:775-779, folded in);
 * rowmax: optional [rows] receiving max |ds| of each row (mulan_absmax_rows over it yields per-image maxima) */
int mulan_softmax_scaled_fwd(const float* x, float* y, size_t rows, int cols, float alpha, mulan_stream_t stream);
int mulan_softmax_scaled_bwd(const float* p, const float* dp, float* ds, size_t rows, int cols, float alpha,
                             float* rowmax, mulan_stream_t stream);
/* Base2FourierFeatures(start=6, stop=8) + concat, padded to 16 channels (model_vdm.py:341-343,812-829) */
int mulan_fourier_fwd(const float* z, float* out, size_t npix, mulan_stream_t stream);
int mulan_fourier_bwd(const float* z, const float* dout, float* dz, size_t npix, int accumulate,
                      mulan_stream_t stream);
/* get_timestep_embedding (model_vdm.py:391-413): out[r][col0 .. col0+E) = [sin, cos](1000 t_r w_k) */
int mulan_temb_fwd(const float* t, float* out, int n, int E, int ld, int col0, mulan_stream_t stream);
int mulan_temb_bwd(const float* t, const float* dout, float* dt, int n, int E, int ld, int col0,
                   mulan_stream_t stream);
/* y[r][col0 + c] = x[r / rep][c]  (conditioning broadcast of ldm_unet.py:85-87 / concat of model_vdm.py:336) */
int mulan_rowbcast(const float* x, float* y, size_t rows, int cols, int rep, int ld, int col0,
                   mulan_stream_t stream);
/* EncDec.encode (model_vdm.py:274-280): f = 2 ((x + .5) / 256) - 1 */
int mulan_encode_u8(const unsigned char* x, float* f, size_t n, mulan_stream_t stream);
int mulan_axpby(const float* x, float* y, size_t n, float a, float b, mulan_stream_t stream);

/* ---- ancestral sampler ---------------------------------------------------------
 * One reverse step of VDM.sample / conditional_sample (model_mulan_velocity.py:281-350, model_mulan_epsilon.py:377-437,
 * model_vdm.py:182-210):  z_s = sqrt(a/b) (z_t - sigma_t c eps_hat) + sqrt((1-a) c) eps  with a = sigmoid(-g_s),
 * b = sigmoid(-g_t), c = -expm1(g_s - g_t); mode 0: net is the velocity (eps_hat = net alpha_t + sigma_t z_t), mode 1:
 * net is eps_hat, mode 2: net is x_hat (plain VDM reparam_type 'input': eps_hat = (z_t - alpha_t net) / sigma_t).
 * gamma per element (g_per_sample = 0) or one value per g_per_sample consecutive elements. */
int mulan_ancestral_step(const float* zt, const float* net, const float* gt, const float* gs, const float* eps,
                         float* zs, size_t n, int mode, int g_per_sample, mulan_stream_t stream);
/* VDM.generate_x with sample_softmax = False (model_mulan_velocity.py:352-368, model_vdm.py:212-227): argmax over the
 * 256 decoder bins of EncDec.decode (model_vdm.py:282-296) at z_0 / sqrt(1 - sigmoid(g_0)). */
int mulan_decode_argmax(const float* z0, const float* g0, unsigned char* out, size_t n, int g_per_sample,
                        mulan_stream_t stream);
/* sample_softmax = True: jax.random.categorical over the same logits (Gumbel-max with Philox4x32-10 draws) */
int mulan_decode_sample(const float* z0, const float* g0, unsigned char* out, size_t n, int g_per_sample,
                        unsigned long long seed, unsigned long long offset, mulan_stream_t stream);
/* EncDec.decode as a table (model_vdm.py:282-296): out[n][256] = log_softmax over the 256 bins of
 * -0.5 ((z - v_j) exp(-0.5 g_0))^2 per sub-pixel (z as given: no rescaling).  Module-surface only: the train / eval
 * path evaluates the bin of x inside mulan_qsample_fwd and never materialises the table. */
int mulan_decode_logprobs(const float* z, const float* g0, float* out, size_t n, int g_per_sample,
                          mulan_stream_t stream);
/* out[r] = mean(x[r, :])  (VDM._get_score_model_gt, model_mulan_velocity.py:141-146) */
int mulan_rowmean(const float* x, float* out, int rows, int cols, mulan_stream_t stream);

/* ---- exact-likelihood ODE evaluator ------------------------------------------------
 * Probability-flow drift of VDM.reverse_ode (model_mulan_velocity.py:393-421 modes 0 / 1 = velocity /
 * velocity_from_epsilon, model_mulan_epsilon.py:459-478 and model_vdm.py:243-260 mode 2) from the network output, and
 * the cotangent d(sum drift * hutch)/d net that the U-Net's input-gradient pass is fed for the Hutchinson estimator
 * (notebook_utils._get_value_div_fn, :203-215).  cot / hutch may be NULL (drift only).  mode | 4 (both entry points):
 * reverse_ode(high_precision=True), the selects alpha = exp(-g/2) where 1 - sigmoid(g) <= 1e-3 and sigma = exp(g/2)
 * where sigmoid(g) <= 1e-3 of model_mulan_velocity.py:410-417 / model_mulan_epsilon.py:472-475. */
int mulan_ode_drift(const float* net, const float* x, const float* gt, const float* gp, const float* hutch, float* drift,
                    float* cot, size_t n, int mode, int g_per_sample, mulan_stream_t stream);
/* div[b] = sum_i (gx_i + diag_i hutch_i) hutch_i: gx = U-Net input gradient for `cot`, diag = the closed-form
 * d drift_i / d x_i around the network. */
int mulan_ode_div(const float* gx, const float* gt, const float* gp, const float* hutch, float* div, int B, int d,
                  int mode, int g_per_sample, mulan_stream_t stream);
/* Dormand-Prince RK45 pieces on a device-resident float64 state (replaces the host round trip per function
 * evaluation of scipy.integrate.solve_ivp, notebook_utils.py:345-358).  K: up to 7 fp32 stage vectors, kstride
 * elements apart; coef / e: host arrays.  out = y + h sum_j coef_j K_j (float64 and / or fp32). */
int mulan_rk_combine(const double* y, const float* K, size_t kstride, const double* coef, int ncoef, double h,
                     double* out, float* out32, size_t n, mulan_stream_t stream);
size_t mulan_rk_workspace_bytes(void);
/* out[0] = sum_i (h sum_j e_j K_j[i] / (atol + rtol max(|y_i|, |ynew_i|)))^2   (scipy's RK45 error norm, squared,
 * not yet divided by n) */
int mulan_rk_error_norm(const double* y, const double* ynew, const float* K, size_t kstride, const double* e, double h,
                        double rtol, double atol, double* workspace, double* out, size_t n, mulan_stream_t stream);
/* out3 = sums of (y0/scale)^2, (f0/scale)^2, ((f1-f0)/scale)^2 with scale = atol + rtol |y0| (scipy's
 * select_initial_step); f1 may be NULL */
int mulan_rk_init_norms(const double* y0, const float* f0, const float* f1, double rtol, double atol, double* workspace,
                        double* out3, size_t n, mulan_stream_t stream);
/* out[r] = log N(x[r, :]; 0, I)   (notebook_utils._prior_logp, :218-221) */
int mulan_normal_logp(const float* x, float* out, int rows, int cols, mulan_stream_t stream);
/* Philox4x32-10 draws: kind 0 U[0,1), 1 Rademacher +-1, 2 standard normal truncated to [lo, hi], 3 Gumbel (stand in for
 * jax.random.uniform / randint / truncated_normal, notebook_utils.py:243-260, 318-330) */
int mulan_noise(float* out, size_t n, unsigned long long seed, unsigned long long offset, int kind, float lo, float hi,
                mulan_stream_t stream);
/* data = encode(x) + noise (uniform: 2 (u - .5) / 256; else u * scale); requant = round(clip(128 (data + 1) - .5, 0,
 * 255)) as the encoder's integer input (notebook_utils.py:316-337) */
int mulan_dequantize(const unsigned char* x, const float* u, float* data, unsigned char* requant, size_t n, int uniform,
                     float scale, mulan_stream_t stream);

/* ---- MuLAN closed-form terms; d must be 3072 ------------------------------------------------- */
/* NoiseSchedule_polynomial_fixedend._eval_polynomial / _grad_t (model_mulan_epsilon.py:514-555).
 * g0, g1, gprime may be NULL. */
int mulan_poly_gamma_fwd(const float* a, const float* b, const float* c, const float* t, float* g0, float* g1,
                         float* gt, float* gprime, int B, int d, float gamma_min, float gamma_max,
                         mulan_stream_t stream);
int mulan_poly_gamma_bwd(const float* a, const float* b, const float* c, const float* t, const float* dgt,
                         const float* dgprime, float* da, float* db, float* dc, int B, int d, float gamma_min,
                         float gamma_max, mulan_stream_t stream);
/* discrete-time (T > 0) loss weight w = T * expm1(gamma_t - gamma_s) (model_mulan_epsilon.py:348-355) */
int mulan_expm1_weight_fwd(const float* gt, const float* gs, float* w, size_t n, float T, mulan_stream_t stream);
int mulan_expm1_weight_bwd(const float* gt, const float* gs, const float* dw, float* dgt, float* dgs, size_t n,
                           float T, mulan_stream_t stream);
/* VDM.__call__ up to the score-model call (model_mulan_velocity.py:208-236, model_mulan_epsilon.py:300-328,
 * model_vdm.py:119-151) with EncDec.encode/decode/logprob (model_vdm.py:274-303) fused in.
 * per_element_gamma: g* are [B,d] (MuLAN) or [B] (scalar schedule of model_vdm.VDM). */
int mulan_qsample_fwd(const unsigned char* x, const float* g0, const float* g1, const float* gt,
                      int per_element_gamma, const float* eps0, const float* eps, float* zt, float* gbar,
                      float* loss_recon, float* loss_klz, float* var0, float* var1, int B, int d,
                      mulan_stream_t stream);
int mulan_qsample_bwd(const unsigned char* x, const float* g0, const float* g1, const float* gt,
                      int per_element_gamma, const float* eps0, const float* eps, const float* dzt,
                      const float* dgbar, const float* drecon, const float* dklz, float* dgt, float* dg0,
                      float* dg1, int B, int d, mulan_stream_t stream);
/* mode 0: velocity (model_mulan_velocity.py:250-260); 1: velocity_from_epsilon (:246-249);
 * 2: epsilon with weight gprime (model_mulan_epsilon.py:338-355, model_vdm.py:156-170). */
int mulan_diffloss_fwd(int mode, const unsigned char* x, const float* gt, const float* gprime,
                       int per_element_gamma, const float* eps, const float* zt, const float* net,
                       float* loss_diff, int B, int d, mulan_stream_t stream);
int mulan_diffloss_bwd(int mode, const unsigned char* x, const float* gt, const float* gprime,
                       int per_element_gamma, const float* eps, const float* zt, const float* net,
                       const float* dloss, float* dnet, float* dgt, float* dgprime, float* dzt, int B, int d,
                       mulan_stream_t stream);
/* _topk_embedding_and_loss + _gamma_noise + _gumbel_kl_loss (model_mulan_velocity.py:78-120).
 * gnoise: [10,B,L] raw Gamma(1/k,1) draws; tau < 0: gnoise is [B,L] additive noise (topk_noise_type 'gumbel',
 * model_mulan_epsilon.py:236-239); gnoise = NULL: hard k-hot of the plain logits (soft / nrm may be NULL). */
int mulan_topk_fwd(const float* logits, const float* gnoise, float* emb, float* kl, float* soft, float* nrm,
                   int B, int L, int k, float tau, mulan_stream_t stream);
int mulan_topk_bwd(const float* logits, const float* soft, const float* nrm, const float* demb, const float* dkl,
                   float* dlogits, int B, int L, mulan_stream_t stream);

/* ---- optimiser + RNG ------------------------------------------------------------------------- */
/* TrainState.apply_gradients (ldm/train_state.py:70-102) with optax.adamw of ldm/experiment.py:132-182
 * on one flat buffer; elements [0, n_decay) are weight-decayed.  step >= 1 is the Adam count. */
int mulan_adamw_ema_step(float* p, const float* g, float* m, float* v, float* ema, size_t n, size_t n_decay,
                         float lr, float b1, float b2, float eps, float weight_decay, int step, float ema_rate,
                         float grad_scale, mulan_stream_t stream);
/* optax.chain(clip_by_global_norm(c), adamw) (experiment.py:176-178, optional `gradient_clip_norm`):
 * mulan_global_norm_clip writes out[0] = min(1, c / norm), out[1] = norm with norm = pre_scale * ||g||_2 (pre_scale =
 * 1 / world: the norm of the rank-averaged gradient); mulan_adamw_ema_step_scaled multiplies the gradient by that
 * device-resident factor on top of grad_scale, so no host synchronisation is needed. */
size_t mulan_global_norm_clip_workspace(void);
int mulan_global_norm_clip(const float* g, size_t n, float clip, float pre_scale, void* workspace, float* out,
                           mulan_stream_t stream);
int mulan_adamw_ema_step_scaled(float* p, const float* g, float* m, float* v, float* ema, size_t n, size_t n_decay,
                                float lr, float b1, float b2, float eps, float weight_decay, int step, float ema_rate,
                                float grad_scale, const float* grad_scale_dev, mulan_stream_t stream);
/* the same step with lr and the Adam bias corrections on the device: dyn[0] = lr(step) (ldm/experiment.py:343),
 * dyn[1] = 1 - b1^t, dyn[2] = 1 - b2^t; stream-ordered parameters for HIP-graph replay of the train step. */
int mulan_adamw_ema_step_dyn(float* p, const float* g, float* m, float* v, float* ema, size_t n, size_t n_decay,
                             float b1, float b2, float eps, float weight_decay, float ema_rate, float grad_scale,
                             const float* grad_scale_dev, const float* dyn, mulan_stream_t stream);
/* N(0,1) draws: Philox4x32-10 + Box-Muller (stands in for jax.random.normal, model_mulan_velocity.py:223,235) */
int mulan_randn(float* out, size_t n, unsigned long long seed, unsigned long long offset, mulan_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MULAN_HIP_H_ */
