// 3x3 convolution with fp32-equivalent products on the fp16 matrix cores ("f16x3").
//
// An fp32 operand v, scaled by a power of two s so that max|v s| lies in [2^13, 2^14), splits into two fp16 pieces
//     v s = h + l,   h = fp16(v s),  l = fp16(v s - h)          (11 + 11 mantissa bits + the two rounding signs:
// |v s - h - l| <= 2^-24 |v s| as long as l is a normal fp16 number, and <= 2^-25 absolute below that: fp16 subnormals
// are honoured by the matrix cores, tools/f16_probe.hip), and
//     a * b  ~=  a_h b_h + a_h b_l + a_l b_h                    (the dropped a_l b_l is <= 2^-24 |a b|)
// is accumulated in fp32 by three v_mfma_f32_32x32x16_f16 per 16-deep k step: half the matrix-pipe cycles of the
// 6-pass bf16 split (conv3x3_bf16x6.hip) and 5.3x fewer than v_mfma_f32_32x32x2_f32, at the same measured error
// against fp64 as either (tools/f16_probe.hip, tests/test_gpu_f16x3.py).  The power-of-two scales are exact and
// are divided out of the fp32 accumulators in the epilogue.  Activations / output gradients carry one scale per
// image (every image is normalised on its own, so images of very different magnitude in one batch keep full
// precision); weights carry one scale per tensor; the weight-gradient kernel, which sums over images, uses the
// per-tensor maxima.  The reference asks XLA for float32 matmul precision (ldm/main.py:39); this is that contract.
//
// Tiling is the one of conv3x3_bf16x6_kernel: block = 4 image rows x 128 couts, wave = 2 rows x 64 couts, halo patch
// in LDS, one stage per (16-channel chunk, tap), weights pre-split by mulan_conv3x3_pack_f16x3 into the LDS tile
// layout [cout][plane][16 k], 3-deep weight ring.
#include "common.h"
#include "f16x3_common.h"

namespace {

using namespace f16x3;

__global__ __launch_bounds__(256) void conv3x3_f16x3_kernel(ConvArgsH p) {
  constexpr int MT = 2, NT = 2, WN = 2;
  constexpr int PV = 4;                          // float4 patch slots per thread (816 slots)
  constexpr int WV = 2;                          // 16-byte weight pieces per thread (512 pieces)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* pbuf0 = smem;
  unsigned char* wbuf0 = smem + PATCH_B;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_per_img = p.H / TROWS;
  const int b = blockIdx.x / tiles_per_img;
  const int h0 = (blockIdx.x % tiles_per_img) * TROWS;
  const int n0 = blockIdx.y * BN;
  const int C = p.C, N = p.N;
  const int nchunks = C / CK;
  const float* xb = p.x + (size_t)b * p.H * kW * C;
  float sx, inv_x, sw, inv_w;
  scale_of(row_max16(p.xmax, b), sx, inv_x);
  scale_of(row_max16(p.wmax, 0), sw, inv_w);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // chunk / stage invariant prefetch addressing
  const float* pptr[PV];
  int pdst[PV];
  unsigned phalo = 0;
#pragma unroll
  for (int s = 0; s < PV; ++s) {
    const int slot = tid + s * 256;
    const int q = slot & 3, pix = slot >> 2;
    const int prow = pix / kPW, pcol = pix - prow * kPW;
    const int hh = h0 + prow - 1, ww = pcol - 1;
    const bool inb = slot < (TROWS + 2) * kPW * 4;
    const bool ok = inb && hh >= 0 && hh < p.H && ww >= 0 && ww < kW;
    pptr[s] = ok ? xb + ((size_t)hh * kW + ww) * C + q * 4 : p.x;
    pdst[s] = inb ? pix * PIXB + q * 8 : -1;
    phalo |= (ok ? 1u : 0u) << s;
  }
  int wsrc[WV], wdst[WV];
#pragma unroll
  for (int s = 0; s < WV; ++s) {
    const int part = tid + s * 256;              // 16-byte piece of the 128 x 64 B tile
    const int n = part >> 2, piece = part & 3;
    wsrc[s] = part * 16;
    wdst[s] = n * NB + piece * 16;
  }
  const size_t tile_stride = (size_t)N * 64;     // bytes between (tap, chunk) tiles of the packed weights
  const unsigned char* wtile0 = p.wp + (size_t)n0 * 64;

  f32x4 preg[PV];
  f32x4 wreg[WV];
  auto gload_patch = [&](int cc) {
#pragma unroll
    for (int s = 0; s < PV; ++s) preg[s] = *reinterpret_cast<const f32x4*>(pptr[s] + (((phalo >> s) & 1u) ? cc * CK : 0));
  };
  auto store_patch = [&](unsigned char* pb) {
#pragma unroll
    for (int s = 0; s < PV; ++s) {
      if (pdst[s] < 0) continue;
      f32x4 v = preg[s];
      if (!((phalo >> s) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 h, l;
        split2(v[e] * sx, h, l);
        hi[e] = h; lo[e] = l;
      }
      *reinterpret_cast<f16x4*>(pb + pdst[s]) = hi;
      *reinterpret_cast<f16x4*>(pb + pdst[s] + 32) = lo;
    }
  };
  auto gload_w = [&](int cc, int tap) {
    const unsigned char* t = wtile0 + ((size_t)tap * nchunks + cc) * tile_stride;
#pragma unroll
    for (int s = 0; s < WV; ++s) wreg[s] = *reinterpret_cast<const f32x4*>(t + wsrc[s]);
  };
  auto store_w = [&](unsigned char* wb) {
#pragma unroll
    for (int s = 0; s < WV; ++s) *reinterpret_cast<f32x4*>(wb + wdst[s]) = wreg[s];
  };

  // Software pipeline: see conv3x3_bf16x6_kernel.  Tile s+1 is complete while stage s computes, so stage s+1's
  // fragments are read from LDS during stage s's 12 MFMAs; tile s+2 is fetched from L2 meanwhile.
  const int nsteps = nchunks * 9;
  auto tile_of = [&](int s2, int& cc2, int& tap2) { cc2 = s2 / 9; tap2 = s2 - cc2 * 9; };
  auto read_frags = [&](f16x8 (&af)[MT][2], f16x8 (&bfr)[NT][2], int tap, const unsigned char* wb) {
    const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int prow = wm * MT + mt + kh, pcol = li + kw;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        af[mt][pl] = *reinterpret_cast<const f16x8*>(pbuf0 + (prow * kPW + pcol) * PIXB + pl * 32 + lh * 16);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        bfr[nt][pl] = *reinterpret_cast<const f16x8*>(wb + ((wn * NT + nt) * 32 + li) * NB + pl * 32 + lh * 16);
  };

  const bool stamp = p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  if (stamp) { p.stamps[0] = __builtin_amdgcn_s_memtime(); p.stamps[30] = __builtin_amdgcn_s_memrealtime(); }
  gload_patch(0);
  gload_w(0, 0);
  store_patch(pbuf0);
  store_w(wbuf0);
  if (nsteps > 1) {
    gload_w(0, 1);
    store_w(wbuf0 + WT_B);
  }
  __syncthreads();

  f16x8 afc[MT][2], bfc[NT][2], afn[MT][2], bfn[NT][2];
  read_frags(afc, bfc, 0, wbuf0);
  int step = 0, slot_next = 1, slot_fill = 2;          // ring slots of tile step+1 / step+2
  if (stamp) p.stamps[1] = __builtin_amdgcn_s_memtime();
  for (int cc = 0; cc < nchunks; ++cc) {
    if (stamp && cc < 20) p.stamps[2 + cc] = __builtin_amdgcn_s_memtime();
    const bool more_chunks = cc + 1 < nchunks;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap, ++step) {
      const bool has_next = step + 1 < nsteps, has_fill = step + 2 < nsteps;
      if (tap < 8) read_frags(afn, bfn, tap + 1, wbuf0 + slot_next * WT_B);   // next stage's operands, same patch
      if (has_fill) {
        int c2, t2;
        tile_of(step + 2, c2, t2);
        gload_w(c2, t2);
      }
      if (tap == 0 && more_chunks) gload_patch(cc + 1);
      __builtin_amdgcn_sched_barrier(0);

      // small terms first: a_l b_h, a_h b_l, a_h b_h
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        constexpr int PA[3] = {1, 0, 0};
        constexpr int PB[3] = {0, 1, 0};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afc[mt][PA[term]], bfc[nt][PB[term]], acc[mt][nt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (has_fill) store_w(wbuf0 + slot_fill * WT_B);
      __syncthreads();
      if (tap == 8 && more_chunks) {       // every wave has left this chunk's patch: replace it
        store_patch(pbuf0);
        __syncthreads();
      }
      if (tap == 8 && has_next) read_frags(afn, bfn, 0, wbuf0 + slot_next * WT_B);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) afc[mt][pl] = afn[mt][pl];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) bfc[nt][pl] = bfn[nt][pl];
      slot_next = slot_next == 2 ? 0 : slot_next + 1;
      slot_fill = slot_fill == 2 ? 0 : slot_fill + 1;
    }
  }

  if (stamp) p.stamps[22] = __builtin_amdgcn_s_memtime();
  // epilogue: transposed through LDS so every lane moves float4s (see conv3x3_fwd_kernel); scales divided out here
  const float* __restrict__ res = p.res;
  const float* __restrict__ cbp = p.cbias;
  float* __restrict__ yout = p.y;
  constexpr int TS = 64 + 4;
  float* stage = reinterpret_cast<float*>(smem) + wave * 32 * TS;
  const int c4 = lane & 15, prl = lane >> 4;
  const int nb = n0 + wn * 64 + c4 * 4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + nb);
  if (p.cbias_mode == 1) {
    const f32x4 c = *reinterpret_cast<const f32x4*>(cbp + (size_t)b * N + nb);
    bias4[0] += c[0]; bias4[1] += c[1]; bias4[2] += c[2]; bias4[3] += c[3];
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[mfma32_row(r, lane) * TS + nt * 32 + li] = (acc[mt][nt][r] * inv_x) * inv_w;
    __syncthreads();
    const int hh = h0 + wm * MT + mt;
    const size_t rowbase = (((size_t)b * p.H + hh) * kW) * N + nb;
    f32x4 add[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) add[it] = bias4;
    if (p.cbias_mode == 2) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(cbp + rowbase + (size_t)(it * 4 + prl) * N);
        add[it][0] += c[0]; add[it][1] += c[1]; add[it][2] += c[2]; add[it][3] += c[3];
      }
    }
    if (res) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(res + rowbase + (size_t)(it * 4 + prl) * N);
        add[it][0] += c[0]; add[it][1] += c[1]; add[it][2] += c[2]; add[it][3] += c[3];
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(stage + (it * 4 + prl) * TS + c4 * 4);
      const f32x4 o = {a[0] + add[it][0], a[1] + add[it][1], a[2] + add[it][2], a[3] + add[it][3]};
      *reinterpret_cast<f32x4*>(yout + rowbase + (size_t)(it * 4 + prl) * N) = o;
    }
  }
  if (stamp) { p.stamps[23] = __builtin_amdgcn_s_memtime(); p.stamps[31] = __builtin_amdgcn_s_memrealtime(); }
}

// ---- "big tile" variant: block = 8 image rows x 128 couts, wave = 4 rows x 64 couts (4 x 2 MFMA tiles, 128
// accumulator registers, one wave per SIMD).  The weight fragments are read straight from the packed global tensor
// (L2/L1 resident, two stages ahead, no LDS copy and no per-stage barrier); only the activation patch goes through
// LDS, double buffered, with one barrier per 16-channel chunk.  LDS traffic per MFMA is half that of the 2 x 2
// variant above, which is LDS-bandwidth bound with three passes per product.
constexpr int TR2 = 8;
constexpr int PATCH2_B = (TR2 + 2) * kPW * PIXB;     // 27200
constexpr int SMEM2_B = 2 * PATCH2_B + 64;           // 54464 (+ dummy store target)

__global__ __launch_bounds__(256) void conv3x3_f16x3_v2_kernel(ConvArgsH p) {
  constexpr int MT = 4, NT = 2;
  constexpr int PV = 6;                          // float4 patch slots per thread (1360 slots)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_per_img = p.H / TR2;
  const int b = blockIdx.x / tiles_per_img;
  const int h0 = (blockIdx.x % tiles_per_img) * TR2;
  const int n0 = blockIdx.y * BN;
  const int C = p.C, N = p.N;
  const int nchunks = C / CK;
  float sx, inv_x, sw, inv_w;
  scale_of(row_max16(p.xmax, b), sx, inv_x);
  scale_of(row_max16(p.wmax, 0), sw, inv_w);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Global operands go through buffer descriptors: per-stage offsets are formed on the scalar unit (soffset) instead
  // of 64-bit vector address arithmetic, and halo / out-of-image slots simply read zeros (offset 2^31 is out of range).
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)((size_t)p.B * p.H * kW * C * 4), kBufWord3);
  const __amdgpu_buffer_rsrc_t wp_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.wp), 0, 9 * C * N * 4, kBufWord3);
  unsigned poff[PV];
  int pdst[PV];
  unsigned pemit[PV];
#pragma unroll
  for (int s = 0; s < PV; ++s) {
    const int slot = tid + s * 256;
    const int q = slot & 3, pix = slot >> 2;
    const int prow = pix / kPW, pcol = pix - prow * kPW;
    const int hh = h0 + prow - 1, ww = pcol - 1;
    const bool inb = slot < (TR2 + 2) * kPW * 4;
    const bool ok = inb && hh >= 0 && hh < p.H && ww >= 0 && ww < kW;
    poff[s] = ok ? (unsigned)((((b * p.H + hh) * kW + ww) * C + q * 4) * 4) : 0x80000000u;
    pdst[s] = inb ? pix * PIXB + q * 8 : -1;
    // byte offset of this slot's hi quad inside (image b, chunk 0) of the plane tensor; interior pixels only
    const bool interior = inb && prow >= 1 && prow <= TR2 && ww >= 0 && ww < kW;
    pemit[s] = interior ? (unsigned)(((hh * kW + ww) * 2) * 32 + q * 8) : 0xffffffffu;
  }
  // plane output window of this block: image b (nothing for the other cout blocks, or when not requested)
  const __amdgpu_buffer_rsrc_t xs_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.xs + (size_t)b * nchunks * 65536, 0, (p.xs && blockIdx.y == 0) ? nchunks * 65536 : 0, kBufWord3);
  i32x4 preg[PV];
  auto gload_patch = [&](int cc) {
#pragma unroll
    for (int s = 0; s < PV; ++s) preg[s] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, poff[s], cc * CK * 4, 0);
  };
  // branch free (the loop body must stay one basic block so that this work can be scheduled between the MFMAs;
  // even a uniform condition in here makes the compiler branch): slots beyond the patch land in a dummy area behind
  // the two patches.  (During the last chunk the "next" patch is a harmless re-load of the current one: same data,
  // stored into the idle buffer and re-emitted to the same place.)
  auto store_slot = [&](unsigned char* pb, int s, int cc) {
    const f32x4 v = __builtin_bit_cast(f32x4, preg[s]);
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 h, l;
      split2(v[e] * sx, h, l);
      hi[e] = h; lo[e] = l;
    }
    unsigned char* d = pdst[s] >= 0 ? pb + pdst[s] : smem + 2 * PATCH2_B;
    *reinterpret_cast<f16x4*>(d) = hi;
    *reinterpret_cast<f16x4*>(d + 32) = lo;
    // the same two quads go to the plane tensor (consumed by the weight-gradient kernel)
    const unsigned eo = pemit[s] != 0xffffffffu ? pemit[s] + (unsigned)cc * 65536u : 0xffffffffu;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, hi), xs_rsrc, eo, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, lo), xs_rsrc, eo == 0xffffffffu ? eo : eo + 32, 0, 0);
  };
  auto store_patch = [&](unsigned char* pb, int cc) {
#pragma unroll
    for (int s = 0; s < PV; ++s) store_slot(pb, s, cc);
  };
  // weight fragments: packed [tap][chunk][cout][plane][16 k] fp16; this lane's 8 k of cout (n0 + wn*64 + nt*32 + li)
  const unsigned bvoff = (unsigned)((n0 + wn * 64 + li) * 64 + lh * 16);
  auto gload_b = [&](f16x8 (&bs)[NT][2], int cc, int tap) {
    const int so = (tap * nchunks + cc) * N * 64;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        bs[nt][pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wp_rsrc, bvoff + nt * 2048 + pl * 32, so, 0));
  };
  auto read_a = [&](f16x8 (&af)[MT][2], const unsigned char* pb, int tap) {
    const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int prow = wm * MT + mt + kh, pcol = li + kw;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        af[mt][pl] = *reinterpret_cast<const f16x8*>(pb + (prow * kPW + pcol) * PIXB + pl * 32 + lh * 16);
    }
  };

  const bool stamp = p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  if (stamp) { p.stamps[0] = __builtin_amdgcn_s_memtime(); p.stamps[30] = __builtin_amdgcn_s_memrealtime(); }
  // dev-only timeline: [64 + 2 blk] = start, [65 + 2 blk] = end of every block in 100 MHz ticks (tools/kbench.py timeline)
  const unsigned tl_blk = blockIdx.y * gridDim.x + blockIdx.x;
  const bool tl = p.stamps && threadIdx.x == 0 && tl_blk < 1024;
  if (tl) p.stamps[64 + 2 * tl_blk] = __builtin_amdgcn_s_memrealtime();
  f16x8 bq[3][NT][2];
  f16x8 afc[MT][2], afn[MT][2];
  gload_patch(0);
  gload_b(bq[0], 0, 0);
  gload_b(bq[1], 0, 1);                            // nchunks * 9 >= 9 stages
  store_patch(smem, 0);
  __syncthreads();
  read_a(afc, smem, 0);
  if (stamp) p.stamps[1] = __builtin_amdgcn_s_memtime();

  for (int cc = 0; cc < nchunks; ++cc) {
    if (stamp && cc < 20) p.stamps[2 + cc] = __builtin_amdgcn_s_memtime();
    const unsigned char* pcur = smem + (cc & 1) * PATCH2_B;
    unsigned char* pnxt = smem + ((cc + 1) & 1) * PATCH2_B;
    const bool more = cc + 1 < nchunks;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap == 8) __syncthreads();               // next chunk's patch (stored at taps 2..7) is complete
      read_a(afn, tap < 8 ? pcur : pnxt, tap < 8 ? tap + 1 : 0);
      {                                            // past the last tile: a harmless reload of the last chunk
        const int t2 = tap + 2 >= 9 ? tap + 2 - 9 : tap + 2;
        const int c2 = tap + 2 >= 9 ? cc + 1 : cc;
        gload_b(bq[(tap + 2) % 3], c2 < nchunks ? c2 : nchunks - 1, t2);
      }
      if (tap == 0) gload_patch(more ? cc + 1 : cc);
      // the next patch is split and stored one float4 slot per stage (taps 2..7), inside the MFMA shadow
      if (tap >= 2 && tap < 2 + PV) store_slot(pnxt, tap - 2, more ? cc + 1 : cc);
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        constexpr int PA[3] = {1, 0, 0};
        constexpr int PB[3] = {0, 1, 0};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afc[mt][PA[term]], bq[tap % 3][nt][PB[term]], acc[mt][nt],
                                                                 0, 0, 0);
      }
      // issue order inside the stage: one MFMA at a time, with the next stage's LDS reads / global loads and the
      // VALU + LDS-write work of the patch slot spread over the MFMA shadows
#pragma unroll
      for (int g = 0; g < 24; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (g < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else if (g < 18) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        if (g >= 12) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) afc[mt][pl] = afn[mt][pl];
    }
  }

  if (stamp) p.stamps[22] = __builtin_amdgcn_s_memtime();
  // epilogue: per image row, transposed through LDS so every lane moves float4s; scales divided out here
  const float* __restrict__ res = p.res;
  const float* __restrict__ cbp = p.cbias;
  float* __restrict__ yout = p.y;
  constexpr int TS = 64 + 4;
  float* stage = reinterpret_cast<float*>(smem) + wave * 32 * TS;
  const int c4 = lane & 15, prl = lane >> 4;
  const int nb = n0 + wn * 64 + c4 * 4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + nb);
  if (p.cbias_mode == 1) {
    const f32x4 c = *reinterpret_cast<const f32x4*>(cbp + (size_t)b * N + nb);
    bias4[0] += c[0]; bias4[1] += c[1]; bias4[2] += c[2]; bias4[3] += c[3];
  }
  // (issuing the residual / per-pixel bias loads of all four rows up front, or of row mt + 1 before row mt is stored,
  // were both measured slower: with a residual the launch moves 285 MB and the epilogue is bound by the memory
  // system's throughput, not by load latency; staggering the blocks' start does not help either)
  unsigned omax = 0;
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[mfma32_row(r, lane) * TS + nt * 32 + li] = (acc[mt][nt][r] * inv_x) * inv_w;
    const size_t rowbase = (((size_t)b * p.H + h0 + wm * MT + mt) * kW) * N + nb;
    f32x4 add[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) add[it] = bias4;
    if (p.cbias_mode == 2) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(cbp + rowbase + (size_t)(it * 4 + prl) * N);
        add[it][0] += c[0]; add[it][1] += c[1]; add[it][2] += c[2]; add[it][3] += c[3];
      }
    }
    if (res) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(res + rowbase + (size_t)(it * 4 + prl) * N);
        add[it][0] += c[0]; add[it][1] += c[1]; add[it][2] += c[2]; add[it][3] += c[3];
      }
    }
    // the staging tile is private to this wave: only wave-level ordering is needed
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's ds_writes have landed
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(stage + (it * 4 + prl) * TS + c4 * 4);
      const f32x4 o = {a[0] + add[it][0], a[1] + add[it][1], a[2] + add[it][2], a[3] + add[it][3]};
      *reinterpret_cast<f32x4*>(yout + rowbase + (size_t)(it * 4 + prl) * N) = o;
#pragma unroll
      for (int e = 0; e < 4; ++e) omax = max(omax, __float_as_uint(o[e]) & 0x7fffffffu);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  if (p.ymax) {   // this block is partial maximum number (row tile, cout block) of image b; unused entries zeroed
    const int part = (h0 / TR2) * gridDim.y + blockIdx.y, nparts = tiles_per_img * gridDim.y;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o, 64));
    __syncthreads();
    unsigned* ured = reinterpret_cast<unsigned*>(smem);
    if (lane == 0) ured[wave] = omax;
    __syncthreads();
    if (tid == 0) p.ymax[b * 16 + part] = max(max(ured[0], ured[1]), max(ured[2], ured[3]));
    if (part == 0 && tid >= nparts && tid < 16) p.ymax[b * 16 + tid] = 0u;
  }
  if (stamp) { p.stamps[23] = __builtin_amdgcn_s_memtime(); p.stamps[31] = __builtin_amdgcn_s_memrealtime(); }
  if (tl) p.stamps[65 + 2 * tl_blk] = __builtin_amdgcn_s_memrealtime();
}

// out[r][c] = fp32 bits of the maximum |x| over part c of row r (16 parts per row; non-negative floats order like
// their bit patterns, so integer max works); no atomics, no zero-fill pass
__global__ __launch_bounds__(256) void absmax_rows_kernel(const float* __restrict__ x, unsigned* __restrict__ out,
                                                          size_t row_len4) {
  __shared__ unsigned red[4];
  const f32x4* row = reinterpret_cast<const f32x4*>(x) + (size_t)blockIdx.x * row_len4;
  unsigned m = 0;
  const size_t stride = (size_t)kMaxParts * 256;
  size_t i = (size_t)blockIdx.y * 256 + threadIdx.x;
  for (; i + 3 * stride < row_len4; i += 4 * stride) {
    const f32x4 a = row[i], b = row[i + stride], c = row[i + 2 * stride], d = row[i + 3 * stride];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      m = max(m, __float_as_uint(a[e]) & 0x7fffffffu);
      m = max(m, __float_as_uint(b[e]) & 0x7fffffffu);
      m = max(m, __float_as_uint(c[e]) & 0x7fffffffu);
      m = max(m, __float_as_uint(d[e]) & 0x7fffffffu);
    }
  }
  for (; i < row_len4; i += stride) {
    const f32x4 a = row[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(a[e]) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[(size_t)blockIdx.x * kMaxParts + blockIdx.y] = max(max(red[0], red[1]), max(red[2], red[3]));
}

// c = a + b with the maxima of c as above: the sum autograd would form for a tensor with two consumers (the U-Net skip
// connections) and the maxima pass of the convolution that takes c as its dy, in one pass over the data
// colpart (optional, [rows][kMaxParts][ncols]): the column sums of c's rows seen as [row_len / ncols, ncols] matrices,
// one partial per block -- the bias gradient of the convolution that receives c as its output gradient is the column
// sum of these rows * 16 small vectors instead of a second pass over c.  (ncols / 4) divides 256: a thread's float4s
// all belong to one column quad.
__global__ __launch_bounds__(256) void add_absmax_rows_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ c, unsigned* __restrict__ out,
                                                              size_t row_len4, float* __restrict__ colpart, int ncols) {
  __shared__ unsigned red[4];
  __shared__ f32x4 cred[256];
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  const size_t base = (size_t)blockIdx.x * row_len4;
  const f32x4* ra = reinterpret_cast<const f32x4*>(a) + base;
  const f32x4* rb = reinterpret_cast<const f32x4*>(b) + base;
  f32x4* rc = reinterpret_cast<f32x4*>(c) + base;
  unsigned m = 0;
  const size_t stride = (size_t)kMaxParts * 256;
  for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < row_len4; i += stride) {
    const f32x4 v = ra[i] + rb[i];
    rc[i] = v;
    cs += v;
#pragma unroll
    for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(v[e]) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  if (colpart) cred[threadIdx.x] = cs;
  __syncthreads();
  if (threadIdx.x == 0) out[(size_t)blockIdx.x * kMaxParts + blockIdx.y] = max(max(red[0], red[1]), max(red[2], red[3]));
  if (colpart) {     // threads t, t + q, t + 2 q, ... (q = ncols / 4 column quads) hold the same columns: fixed order
    const int q = ncols >> 2;
    if ((int)threadIdx.x < q) {
      f32x4 t = cred[threadIdx.x];
      for (int j = threadIdx.x + q; j < 256; j += q) t += cred[j];
      reinterpret_cast<f32x4*>(colpart)[((size_t)blockIdx.x * kMaxParts + blockIdx.y) * q + threadIdx.x] = t;
    }
  }
}

// wp[t][cc][o][plane][k] = split2( s_w * Wl[t][cc*16 + k][o] ),  Wl = w (flip = 0) or the tap-flipped,
// channel-transposed weights of the input-gradient convolution (flip = 1: Wl[t][k][o] = w[8-t][o][k]).
__global__ void conv3x3_pack_f16x3_kernel(const float* __restrict__ w, _Float16* __restrict__ wp,
                                          const unsigned* __restrict__ wmax, int C, int N, int flip) {
  const int Kin = flip ? N : C, Nout = flip ? C : N;
  const size_t total = (size_t)9 * Kin * Nout;
  float sw, inv_w;
  scale_of(row_max16(wmax, 0), sw, inv_w);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % 16);
    size_t r = i / 16;
    const int o = (int)(r % Nout); r /= Nout;
    const int cc = (int)(r % (Kin / 16));
    const int t = (int)(r / (Kin / 16));
    const int kin = cc * 16 + k;
    const float v = flip ? w[((size_t)(8 - t) * C + o) * N + kin] : w[((size_t)t * C + kin) * N + o];
    _Float16 h, l;
    split2(v * sw, h, l);
    _Float16* dst = wp + (((size_t)(t * (Kin / 16) + cc) * Nout + o) * 2) * 16 + k;
    dst[0] = h; dst[16] = l;
  }
}


// ------------------------------------------------------------------------------------------------ wgrad
// dW[t][ci][co] = sum_pixels x[p + shift_t][ci] * dy[p][co] with the same 3-pass split; operands fetched with the
// transposing LDS read ds_read_b64_tr_b16 exactly as in conv3x3_wgrad_bf16x6_kernel (two planes instead of three).
constexpr int WG_T = 64, WG_ROWS = 2;
constexpr int XPIX = (WG_ROWS + 2) * kPW;                 // 136 halo-patch pixels
constexpr int X_HALF = XPIX * 64, X_PLANE = 2 * X_HALF;   // bytes
constexpr int DPIX = WG_ROWS * kW;                        // 64
constexpr int D_HALF = DPIX * 64, D_PLANE = 2 * D_HALF;
constexpr int WG_SMEM = 2 * X_PLANE + 2 * D_PLANE;        // 34816 + 16384 = 51200

struct WgradArgsH {
  const float* x; const float* dy; float* slab;
  const unsigned* xmax; const unsigned* dymax;    // [B] per-image maxima (the kernel takes the tensor maximum)
  int B, H, C, N, S;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f16x8 tr_read8h(const unsigned char* base) {
  // two 4-k blocks (k .. k+3 and k+4 .. k+7) -> the 8 k values of this lane's MFMA operand
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + 4 * 64));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(f16x8, v);
}

__global__ __launch_bounds__(256) void conv3x3_wgrad_f16x3_kernel(WgradArgsH p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;
  unsigned char* ds = smem + 2 * X_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wci = wave >> 1, wco = wave & 1;
  const int C = p.C, N = p.N;
  const int c0 = blockIdx.y * WG_T, n0 = blockIdx.z * WG_T;
  const int pairs_per_img = p.H / WG_ROWS;
  const int total_pairs = p.B * pairs_per_img;
  const int per_split = (total_pairs + p.S - 1) / p.S;
  const int pair_begin = blockIdx.x * per_split;
  const int pair_end = min(total_pairs, pair_begin + per_split);

  // tensor maxima -> scales
  float sx, inv_x, sg, inv_g;
  {
    unsigned mx = 0, mg = 0;
    for (int i = tid; i < p.B * kMaxParts; i += 256) { mx = max(mx, p.xmax[i]); mg = max(mg, p.dymax[i]); }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
      mg = max(mg, (unsigned)__shfl_xor((int)mg, o, 64));
    }
    unsigned* red = reinterpret_cast<unsigned*>(smem);
    if (lane == 0) { red[wave] = mx; red[4 + wave] = mg; }
    __syncthreads();
    mx = max(max(red[0], red[1]), max(red[2], red[3]));
    mg = max(max(red[4], red[5]), max(red[6], red[7]));
    __syncthreads();
    scale_of(mx, sx, inv_x);
    scale_of(mg, sg, inv_g);
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  constexpr int XV = (XPIX * 16 + 255) / 256;   // 9 float4 slots / thread
  constexpr int DV = (DPIX * 16) / 256;         // 4
  f32x4 xreg[XV], dreg[DV];
  int xoff[XV], doff[DV], xdst[XV], ddst[DV];
  unsigned xstat = 0, xtop = 0, xbot = 0, dstat = 0, xmask = 0;
#pragma unroll
  for (int i = 0; i < XV; ++i) {
    const int slot = tid + i * 256;
    const int q = slot & 15, pix = slot >> 4;
    const int prow = pix / kPW, pcol = pix - prow * kPW;
    const int ww = pcol - 1, c = c0 + q * 4;
    const bool inb = slot < XPIX * 16;
    const bool ok = inb && ww >= 0 && ww < kW && c < C;
    xoff[i] = ((prow - 1) * kW + ww) * C + c;
    xdst[i] = inb ? (q >> 3) * X_HALF + pix * 64 + (q & 7) * 8 : -1;
    xstat |= (ok ? 1u : 0u) << i;
    xtop |= (prow == 0 ? 1u : 0u) << i;
    xbot |= (prow == WG_ROWS + 1 ? 1u : 0u) << i;
  }
#pragma unroll
  for (int i = 0; i < DV; ++i) {
    const int slot = tid + i * 256;
    const int q = slot & 15, pix = slot >> 4;
    const int n = n0 + q * 4;
    doff[i] = pix * N + n;
    ddst[i] = (q >> 3) * D_HALF + pix * 64 + (q & 7) * 8;
    dstat |= (n < N ? 1u : 0u) << i;
  }
  auto gload = [&](int pr) {
    const int b = pr / pairs_per_img, h0 = (pr - b * pairs_per_img) * WG_ROWS;
    const float* xrow = p.x + ((size_t)b * p.H + h0) * kW * C;
    const float* dyb = p.dy + ((size_t)b * p.H + h0) * kW * N;
    xmask = xstat & ~(h0 == 0 ? xtop : 0u) & ~(h0 + WG_ROWS >= p.H ? xbot : 0u);
#pragma unroll
    for (int i = 0; i < XV; ++i) xreg[i] = *reinterpret_cast<const f32x4*>(((xmask >> i) & 1u) ? xrow + xoff[i] : p.x);
#pragma unroll
    for (int i = 0; i < DV; ++i) dreg[i] = *reinterpret_cast<const f32x4*>(((dstat >> i) & 1u) ? dyb + doff[i] : p.dy);
  };
  auto split_store = [&](unsigned char* base, int plane_stride, int dst, f32x4 v, float s) {
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 h, l;
      split2(v[e] * s, h, l);
      hi[e] = h; lo[e] = l;
    }
    *reinterpret_cast<f16x4*>(base + dst) = hi;
    *reinterpret_cast<f16x4*>(base + plane_stride + dst) = lo;
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      if (xdst[i] < 0) continue;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      split_store(xs, X_PLANE, xdst[i], ((xmask >> i) & 1u) ? xreg[i] : z, sx);
    }
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      split_store(ds, D_PLANE, ddst[i], ((dstat >> i) & 1u) ? dreg[i] : z, sg);
    }
  };

  const int grp_q = (lane & 15) >> 2, grp_p = lane & 3, cb16 = ((lane >> 4) & 1) * 16;
  const int lane_off = (8 * lh + grp_q) * 64 + (cb16 + 4 * grp_p) * 2;
  const unsigned char* xa = xs + wci * X_HALF + lane_off;
  const unsigned char* db = ds + wco * D_HALF + lane_off;

  if (pair_begin < pair_end) {
    gload(pair_begin);
    lstore();
  }
  __syncthreads();
  for (int pr = pair_begin; pr < pair_end; ++pr) {
    const bool has_next = pr + 1 < pair_end;
    if (has_next) gload(pr + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int ks = 0; ks < DPIX / 16; ++ks) {
      const int rr = ks >> 1, w0 = (ks & 1) * 16;
      f16x8 bfr[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) bfr[pl] = tr_read8h(db + pl * D_PLANE + (rr * kW + w0) * 64);
      f16x8 af[2][2];
      const unsigned char* xk = xa + (rr * kPW + w0) * 64;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) af[0][pl] = tr_read8h(xk + pl * X_PLANE);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t + 1 < 9) {
          const int kh = (t + 1) / 3, kw = (t + 1) % 3;
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) af[(t + 1) & 1][pl] = tr_read8h(xk + pl * X_PLANE + (kh * kPW + kw) * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][1], bfr[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][0], bfr[1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][0], bfr[0], acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (has_next) {
      lstore();
      __syncthreads();
    }
  }

  float* slab = p.slab + (size_t)blockIdx.x * 9 * C * N;
  const int n = n0 + wco * 32 + li;
  if (n < N) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wci * 32 + mfma32_row(r, lane);
        if (c < C) slab[((size_t)t * C + c) * N + n] = (acc[t][r] * inv_x) * inv_g;
      }
  }
}

// ---- plane-fed, row-split weight-gradient kernel.
// Inputs are the split planes that the forward / input-gradient convolutions write as a by-product
// ([B][C/16][H*W][plane][16] fp16, scaled per image), so staging is a plain copy: the fp32 -> 2 x fp16 split is done
// once per tensor instead of once per consuming block, and nothing competes with the MFMAs for the VALU.  The
// accumulators live in the units of the image being multiplied (s_x[b] s_dy[b]): when the pixel range crosses into the
// next image they are multiplied by the exact power of two between the two images' scales (in place, on the accumulator
// registers).
// A block owns one kernel row kh (3 taps) of a 128 ci x 128 co tile, a wave 64 ci x 64 co x 3 taps (192 accumulator
// registers, one wave per SIMD): every transposed x fragment feeds two co tiles and every dy fragment six MFMAs,
// ~45 % of the LDS traffic per MFMA of the 9-tap kernel above, and no vertical halo.  The three kh blocks of one pixel
// range run on the same XCD (same blockIdx.x, S % 8 == 0): the planes come from HBM once and from L2 twice.
// Three row pairs are in flight: pair p is multiplied out of LDS buffer (p & 1), pair p + 1 sits in registers and is
// copied into the other buffer, pair p + 2 is being fetched (a whole iteration of latency budget); one barrier / pair.
constexpr int WG3_T = 128;
constexpr int X3PIX = WG_ROWS * kPW;                          // 68 pixels: 2 rows x (32 + 2 halo columns)
constexpr int X3_HALF = X3PIX * 64 + 64, X3_PLANE = 4 * X3_HALF + 32;   // +64: the 4 halves of a pixel start 16 banks
constexpr int D3_HALF = DPIX * 64 + 64, D3_PLANE = 4 * D3_HALF + 32;    // apart; +32: plane 1 sits 8 banks further
constexpr int WG3_BUF = 2 * X3_PLANE + 2 * D3_PLANE;          // 35392 + 33344 = 68736
constexpr int WG3_SMEM = 2 * WG3_BUF;                         // 137472

struct WgradArgsP {
  const unsigned char* xs; const unsigned char* dys;   // split planes of x [B][C/16][HW][2][16], dy [B][N/16][HW][2][16]
  const unsigned* xmax; const unsigned* dymax;         // [B] per-image maxima the planes were scaled with
  float* slab;
  int B, H, C, N, S;
  unsigned long long* stamps;                          // dev-only (mulan_set_debug_buffer)
  // XF32 instantiation (one-tap kernel only): x arrives in fp32 as the virtual concat [x1 | x2] ([B][HW][C1], [B][HW][C - C1])
  // and is split while it is staged; xmax holds the maxima of the concat (elementwise max of the two tensors' maxima)
  const float* x1f; const float* x2f; int C1;
  const unsigned* xmax2;                               // (XF32) maxima of x2: the concat's are the elementwise max of xmax, xmax2
  // (w16 kernel, mulan_conv3x3_wgrad_f16x3_planes_fold) slab reductions of EARLIER launches that this launch's blocks
  // perform in their prologue, while their first operand loads are in flight
  mulan_slab_reduction pend[2];
  int npend;
};

// out[e] (+)= sum over the S slabs, in slab order (the order -- and therefore the bits -- of slab_reduce_h_kernel), four
// outputs per thread; gtid / gthreads: this thread's index in / the size of the whole launch.  Up to 16 loads in flight.
__device__ __forceinline__ void fold_slab_reduction(const mulan_slab_reduction& r, int gtid, int gthreads) {
  const int E4 = r.E >> 2;
  for (int e = gtid; e < E4; e += gthreads) {
    const f32x4* src = reinterpret_cast<const f32x4*>(r.slab) + e;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 16 <= r.S; i += 16) {
      f32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(i + u) * E4];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; i + 4 <= r.S; i += 4) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[(size_t)(i + u) * E4];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u];
    }
    for (; i < r.S; ++i) s += src[(size_t)i * E4];
    f32x4* o = reinterpret_cast<f32x4*>(r.out) + e;
    *o = r.accumulate ? *o + s : s;
  }
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));


__device__ __forceinline__ int clamped_exp(unsigned maxbits) {
  const int e = (int)((maxbits >> 23) & 255u);
  return e < 14 ? 14 : (e > 254 ? 254 : e);
}


// TAPS = 3: one kernel row of a 3x3 convolution per block (blockIdx.y = kh).  TAPS = 1: a 1x1 "convolution", i.e. the
// weight gradient x^T dy of a per-pixel dense layer (nin_shortcut) from the planes its forward kernel handed on: the
// same block with only the centre tap.
// XF32 (TAPS = 1): the x operand is read in fp32 and split in the staging path instead of arriving as planes.  The
// dense weight gradient is memory bound (48 MFMAs per wave and row pair against 68 KB of operands), so the split
// arithmetic is free there, and the layer's forward kernel no longer has to write the planes of its input (134 MB per
// nin_shortcut at E = 128: 27 us of a 83 us launch, profiles/r03_linear_probe.log).  Same values as the planes the
// forward kernel used to hand on (split2 of x * s with the concat's per-image scale): bit-identical dw.
// PROBE16 (dev, timing only, wrong numbers): every 32x32x16 MFMA replaced by two 16x16x32 ones on the same fragments --
// the same operand traffic and matrix-core time on the shape the chip clocks higher under its power limit
// ABL (dev, timing only, wrong numbers; tools/wgrad_ab.py --tunes 7=N): 2 = the x fragments of taps kw = 0, 2 are not read
// (the centre tap's are used: what sharing one fragment per k step between the three taps could save at most), 3 = no
// staging (no global loads, no LDS stores: the main loop multiplies what the prologue staged), 4 = 2 + 3, 5 = no slab
// stores, 6 = no barrier in the loop (+ 3)
template <int TAPS, bool XF32 = false, bool PROBE16 = false, int ABL = 0>
__global__ __launch_bounds__(256) void conv3x3_wgrad_f16x3_planes_kernel(WgradArgsP p) {
  static_assert(!XF32 || TAPS == 1, "fp32 x operand: one-tap kernel only");
  constexpr int NQ = 4 * TAPS;                   // stages per row pair: 4 k steps x TAPS
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wci = wave >> 1, wco = wave & 1;
  const int C = p.C, N = p.N;
  const int kh = TAPS == 3 ? blockIdx.y : 1;
  const int ntn = N / WG3_T;
  const int c0 = (blockIdx.z / ntn) * WG3_T, n0 = (blockIdx.z % ntn) * WG3_T;
  const int nchc = C / 16, nchn = N / 16;
  const int pairs_per_img = p.H / WG_ROWS;
  const int total_pairs = p.B * pairs_per_img;
  auto xmax_of = [&](int b) {
    unsigned m = row_max16(p.xmax, b);
    if constexpr (XF32) { if (p.xmax2) m = max(m, row_max16(p.xmax2, b)); }
    return m;
  };
  const int pair_begin = (int)((long long)blockIdx.x * total_pairs / p.S);
  const int pair_end = (int)((long long)(blockIdx.x + 1) * total_pairs / p.S);

  f32x16 acc[TAPS][2][2];   // [kw][ci tile][co tile]
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

  // staging units of 16 bytes (8 channels of one plane of one pixel): unit e = tid + 256 i,
  //   e & 3 -> (plane, channel half),  e >> 2 = chunk * pixels + pixel   (consecutive lanes: consecutive 64-B records)
  constexpr int XV = (X3PIX * 32 + 255) / 256;  // 9 (2176 units)
  constexpr int DV = (DPIX * 32) / 256;         // 8 (2048 units)
  i32x4 xreg[XV], dreg[DV];
  const __amdgpu_buffer_rsrc_t xs_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.xs), 0, p.B * nchc * 65536, kBufWord3);
  const __amdgpu_buffer_rsrc_t ds_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.dys), 0, p.B * nchn * 65536, kBufWord3);
  const int u = tid & 3, upl = u >> 1, uh = u & 1;
  // fp32 x: float4 unit t = tid + 256 i (i < 8): pixel t >> 5 of the pair's 64, channel quad t & 31 of the 128-ci tile
  const bool x_first = c0 < p.C1;
  const int ldxf = x_first ? p.C1 : C - p.C1, cxf = x_first ? c0 : c0 - p.C1;
  const __amdgpu_buffer_rsrc_t xf_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x_first ? p.x1f : p.x2f), 0, XF32 ? (int)((size_t)p.B * 1024 * ldxf * 4) : 0, kBufWord3);
  int b_stage = -1;
  float sx_stage = 0.f;
  auto gload_x1 = [&](int pr, int i) {
    const int b = pr / pairs_per_img, h0 = (pr - b * pairs_per_img) * WG_ROWS;
    if constexpr (XF32) {
      if (i < 8) {
        const int t = tid + 256 * i;
        const int pix = t >> 5, q = t & 31;
        const unsigned off = (unsigned)((((b * 1024 + h0 * kW + pix) * ldxf) + cxf + q * 4) * 4);
        xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(xf_rsrc, off, 0, 0);
      }
      return;
    }
    const int t = (tid >> 2) + 64 * i;
    const int cc = t / X3PIX, pix = t - cc * X3PIX;
    const int prow = pix / kPW, pcol = pix - prow * kPW;
    const int hh = h0 + kh - 1 + prow;
    const bool ok = t < 8 * X3PIX && pcol >= 1 && pcol <= kW && hh >= 0 && hh < p.H;
    const unsigned off = (unsigned)((((b * nchc + (c0 >> 4) + cc) * 1024 + hh * kW + pcol - 1) * 2 + upl) * 32 + uh * 16);
    xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(xs_rsrc, ok ? off : 0xffffffffu, 0, 0);
  };
  auto gload_d1 = [&](int pr, int i) {
    const int b = pr / pairs_per_img, h0 = (pr - b * pairs_per_img) * WG_ROWS;
    const int t = (tid >> 2) + 64 * i;
    const int cc = t >> 6, pix = t & 63;
    const unsigned off = (unsigned)((((b * nchn + (n0 >> 4) + cc) * 1024 + h0 * kW + pix) * 2 + upl) * 32 + uh * 16);
    dreg[i] = __builtin_amdgcn_raw_buffer_load_b128(ds_rsrc, off, 0, 0);
  };
  // branch free (the loop body must stay one basic block): the 128 units past the x tile land in a dummy area
  // (XF32) the scale of the image the staged pair belongs to; set_stage_scale(pr) before the pair's first store
  auto set_stage_scale = [&](int pr) {
    if constexpr (XF32) {
      const int b = pr / pairs_per_img;
      if (b != b_stage) {
        float inv;
        scale_of(xmax_of(b), sx_stage, inv);
        b_stage = b;
      }
    }
  };
  auto store_x = [&](unsigned char* buf, int i) {
    if constexpr (XF32) {
      if (i < 8) {
        const int t = tid + 256 * i;
        const int pix = t >> 5, q = t & 31;
        const int cc = q >> 2, pixl = (pix >> 5) * kPW + (pix & 31) + 1;       // interior pixel of the haloed row layout
        const f32x4 v = __builtin_bit_cast(f32x4, xreg[i]);
        f16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 h, l;
          split2(v[e] * sx_stage, h, l);
          hi[e] = h; lo[e] = l;
        }
        unsigned char* d = buf + (cc >> 1) * X3_HALF + pixl * 64 + (cc & 1) * 32 + (q & 3) * 8;
        *reinterpret_cast<f16x4*>(d) = hi;
        *reinterpret_cast<f16x4*>(d + X3_PLANE) = lo;
      }
      return;
    }
    const int t = (tid >> 2) + 64 * i;
    const int cc = t / X3PIX, pix = t - cc * X3PIX;
    const int dst = upl * X3_PLANE + (cc >> 1) * X3_HALF + pix * 64 + (cc & 1) * 32 + uh * 16;
    *reinterpret_cast<i32x4*>(t < 8 * X3PIX ? buf + dst : smem + 2 * WG3_BUF) = xreg[i];
  };
  auto store_d = [&](unsigned char* buf, int i) {
    const int t = (tid >> 2) + 64 * i;
    const int cc = t >> 6, pix = t & 63;
    const int dst = 2 * X3_PLANE + upl * D3_PLANE + (cc >> 1) * D3_HALF + pix * 64 + (cc & 1) * 32 + uh * 16;
    *reinterpret_cast<i32x4*>(buf + dst) = dreg[i];
  };

  // lane part of every transposing read (see conv3x3_wgrad_bf16x6_kernel)
  const int grp_q = (lane & 15) >> 2, grp_p = lane & 3, cb16 = ((lane >> 4) & 1) * 16;
  const int lane_off = (8 * lh + grp_q) * 64 + (cb16 + 4 * grp_p) * 2;
  const int xa_off = (wci * 2) * X3_HALF + lane_off;
  const int db_off = 2 * X3_PLANE + (wco * 2) * D3_HALF + lane_off;

  const bool stamp = p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0;
  if (stamp) { p.stamps[0] = __builtin_amdgcn_s_memtime(); p.stamps[30] = __builtin_amdgcn_s_memrealtime(); }
  const int last = pair_end - 1;
  int b_acc = pair_begin < pair_end ? pair_begin / pairs_per_img : 0;   // image whose scales the accumulators carry
  if (pair_begin < pair_end) {
    const int p1 = min(pair_begin + 1, last);
#pragma unroll
    for (int i = 0; i < XV; ++i) gload_x1(pair_begin, i);
#pragma unroll
    for (int i = 0; i < DV; ++i) gload_d1(pair_begin, i);
    set_stage_scale(pair_begin);
#pragma unroll
    for (int i = 0; i < XV; ++i) store_x(smem, i);
#pragma unroll
    for (int i = 0; i < DV; ++i) store_d(smem, i);
#pragma unroll
    for (int i = 0; i < XV; ++i) gload_x1(p1, i);
#pragma unroll
    for (int i = 0; i < DV; ++i) gload_d1(p1, i);
  }
  __syncthreads();
  if (stamp) p.stamps[1] = __builtin_amdgcn_s_memtime();
  for (int pr = pair_begin; pr < pair_end; ++pr) {
    if (stamp && pr - pair_begin < 20) p.stamps[2 + pr - pair_begin] = __builtin_amdgcn_s_memtime();
    const int cur = (pr - pair_begin) & 1;
    const unsigned char* bc = smem + cur * WG3_BUF;
    unsigned char* bn = smem + (cur ^ 1) * WG3_BUF;
    const int pr2 = min(pr + 2, last);                 // past the end: a harmless re-read of the last pair
    set_stage_scale(min(pr + 1, last));                // (XF32) the pair in the staging registers is pair pr + 1
    const int b_now = pr / pairs_per_img;
    if (b_now != b_acc) {                              // crossed into the next image: move the accumulators to its units
      const int d = (clamped_exp(xmax_of(b_acc)) - clamped_exp(xmax_of(b_now))) +
                    (clamped_exp(row_max16(p.dymax, b_acc)) - clamped_exp(row_max16(p.dymax, b_now)));
      // acc *= 2^d in place on the accumulator registers (kept in the "a" class so that the register allocation of
      // the main loop is not disturbed by this rare path); four registers per block so the moves interleave.
      // Neighbouring images mostly share their exponents: nothing to do then.
      if (d != 0) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
              float t0, t1, t2, t3;
              asm volatile(
                  "v_accvgpr_read_b32 %4, %0\n\tv_accvgpr_read_b32 %5, %1\n\t"
                  "v_accvgpr_read_b32 %6, %2\n\tv_accvgpr_read_b32 %7, %3\n\ts_nop 0\n\t"
                  "v_ldexp_f32 %4, %4, %8\n\tv_ldexp_f32 %5, %5, %8\n\tv_ldexp_f32 %6, %6, %8\n\tv_ldexp_f32 %7, %7, %8\n\t"
                  "s_nop 1\n\t"
                  "v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\t"
                  "v_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
                  : "+a"(acc[t][i][j][r]), "+a"(acc[t][i][j][r + 1]), "+a"(acc[t][i][j][r + 2]), "+a"(acc[t][i][j][r + 3]),
                    "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                  : "v"(d));
            }
      }
      b_acc = b_now;
    }
    const unsigned char* xa = bc + xa_off;
    const unsigned char* db = bc + db_off;
    f16x8 af[2][2][2], bfr[2][2][2];                   // [buffer][tile][plane]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) bfr[0][j][pl] = tr_read8h(db + pl * D3_PLANE + j * D3_HALF);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) af[0][i][pl] = tr_read8h(xa + pl * X3_PLANE + i * X3_HALF + (TAPS == 1 ? 64 : 0));
#pragma unroll
    for (int q = 0; q < NQ; ++q) {                     // q = ks * TAPS + tap; k step ks = 16 pixels of row rr = ks >> 1
      const int ks = q / TAPS, kw = TAPS == 3 ? q - ks * 3 : 1, ti = TAPS == 3 ? kw : 0;
      if (q + 1 < NQ) {                                // next stage's x fragments
        const int ks1 = (q + 1) / TAPS, kw1 = TAPS == 3 ? (q + 1) - ks1 * 3 : 1;
        const int rr = ks1 >> 1, w0 = (ks1 & 1) * 16;
        if ((ABL == 2 || ABL == 4) && kw1 != 0) {       // (ablation) one fragment read per k step
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[(q + 1) & 1][i][pl] = af[q & 1][i][pl];
        } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
            af[(q + 1) & 1][i][pl] = tr_read8h(xa + pl * X3_PLANE + i * X3_HALF + (rr * kPW + w0 + kw1) * 64);
        }
      }
      if ((TAPS == 1 || kw == 1) && ks + 1 < 4) {      // next k step's dy fragments
        const int rr = (ks + 1) >> 1, w0 = ((ks + 1) & 1) * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
            bfr[(ks + 1) & 1][j][pl] = tr_read8h(db + pl * D3_PLANE + j * D3_HALF + (rr * kW + w0) * 64);
      }
      // staging, each register refilled at once with the same unit of pair p + 2.  TAPS = 3: x unit q in stages
      // 0..8, dy unit q - 4 in stages 4..11.  TAPS = 1 (4 stages): units q, q + 4, (q + 8).
      if (ABL == 3 || ABL == 4 || ABL == 6) {
      } else if (TAPS == 3) {
        if (q < XV) { store_x(bn, q); gload_x1(pr2, q); }
        if (q >= 4) { store_d(bn, q - 4); gload_d1(pr2, q - 4); }
      } else {
#pragma unroll
        for (int uu = q; uu < XV; uu += 4) { store_x(bn, uu); gload_x1(pr2, uu); }
#pragma unroll
        for (int uu = q; uu < DV; uu += 4) { store_d(bn, uu); gload_d1(pr2, uu); }
      }
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        constexpr int PA[3] = {1, 0, 0};
        constexpr int PB[3] = {0, 1, 0};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if constexpr (PROBE16) {
              f32x16& a16 = acc[ti][i][j];
              f32x4 c0 = {a16[0], a16[1], a16[2], a16[3]}, c1 = {a16[4], a16[5], a16[6], a16[7]};
              c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[q & 1][i][PA[term]], bfr[ks & 1][j][PB[term]], c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[q & 1][i][PA[term]], bfr[ks & 1][j][PB[term]], c1, 0, 0, 0);
              a16[0] = c0[0]; a16[1] = c0[1]; a16[2] = c0[2]; a16[3] = c0[3];
              a16[4] = c1[0]; a16[5] = c1[1]; a16[6] = c1[2]; a16[7] = c1[3];
            } else {
              acc[ti][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[q & 1][i][PA[term]], bfr[ks & 1][j][PB[term]],
                                                                     acc[ti][i][j], 0, 0, 0);
            }
          }
      }
#pragma unroll
      for (int g = 0; g < (PROBE16 ? 24 : 12); ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, PROBE16 ? 1 : 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, PROBE16 ? 2 : 4, 0);
        if (TAPS == 1 || (g & (PROBE16 ? 7 : 3)) == (PROBE16 ? 3 : 1)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (TAPS == 1 || (g & (PROBE16 ? 7 : 3)) == (PROBE16 ? 7 : 3)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ABL != 6) __syncthreads();
  }

  if (stamp) p.stamps[22] = __builtin_amdgcn_s_memtime();
  float sdummy, inv_x, inv_g;
  scale_of(xmax_of(b_acc), sdummy, inv_x);
  scale_of(row_max16(p.dymax, b_acc), sdummy, inv_g);
  float* slab = p.slab + (size_t)blockIdx.x * 9 * C * N;
  const int ntaps = TAPS == 3 ? 9 : 1;
  slab = p.slab + (size_t)blockIdx.x * ntaps * C * N;
#pragma unroll
  for (int kw = 0; kw < TAPS; ++kw)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wco * 64 + j * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = c0 + wci * 64 + i * 32 + mfma32_row(r, lane);
          if (ABL != 5 || acc[kw][i][j][r] == 12345.678f)
          slab[((size_t)(TAPS == 3 ? kh * 3 + kw : 0) * C + c) * N + n] = (acc[kw][i][j][r] * inv_x) * inv_g;
        }
      }
  if (stamp) { p.stamps[23] = __builtin_amdgcn_s_memtime(); p.stamps[31] = __builtin_amdgcn_s_memrealtime(); }
}

int wgrad_splits_p(int B, int H, int C, int N, int share_chip) {   // 3 kh blocks per (tile, pixel range); S % 8 == 0 keeps them on one XCD
  const int tiles = (C / WG3_T) * (N / WG3_T);
  const int pairs = B * (H / WG_ROWS);
  // 240 blocks when the launch has the chip to itself; share_chip = 1 (an ARGUMENT of the entry point: the caller says
  // that the launch shares the chip with another stream's kernels -- the train step's weight-gradient stream): 120 for
  // the one- and two-tile shapes, so that the other stream keeps half of the CUs and both run side by side (a block
  // owns its CU); four tiles and more keep 240 (measured: E = 256 is slower with fewer).  tune[1] > 0: explicit
  // target (dev only).
  int target = share_chip == 1 && tiles <= 2 ? 120 : 240;
  if (g_mulan_tune[1] > 0) target = g_mulan_tune[1];
  int S = target / (3 * tiles);
  if (S >= 8 && g_mulan_tune[6] != 1) S &= ~7;      // tune[6] = 1: dev switch, no XCD alignment
  if (S < 1) S = 1;
  if (S > pairs) S = pairs;
  return S;
}

// ---- the 3-tap plane-fed weight gradient as a block of EIGHT waves (round 6).
// Same block tile as conv3x3_wgrad_f16x3_planes_kernel<3> (one kernel row kh of 128 ci x 128 co over a range of row
// pairs), same LDS image, same staging scheme (pair p multiplied, p + 1 in registers, p + 2 in flight), same order of
// summation per output element -- but a wave owns 32 ci x 64 co x 3 taps (96 accumulators, <= 256 registers), so TWO
// waves share every SIMD: the four-wave block issues 144 MFMAs per row pair in 5.76 k cycles (4.6 k would be back to
// back: what a wave loses after each barrier and on every fragment wait nobody fills, profiles/r03_wgrad_block_timeline.log);
// here the partner wave's MFMAs fill those slots.  Price: 10 fragment reads per 18 MFMAs instead of 16 per 36.
// The halo columns of the x rows (always zero) are written once in the prologue instead of being staged with every
// pair: 2048 + 2048 sixteen-byte units per pair = 4 + 4 per thread, no partial slots.
// Grid: one dimension, decoded so that the blocks of one pixel range (3 kh x tiles) run on one XCD (block L runs on
// XCD L % 8) for ANY split count -- the launcher can use 255 of the 256 CUs (S = 85 at one tile) instead of 240.
constexpr int W8_THREADS = 512;

__device__ __forceinline__ void w8_decode(int L, int S, int tiles, int& s, int& kh, int& tile) {
  const int nslot = 3 * tiles, full = S & ~7;
  int slot;
  if (L < full * nslot) {
    const int g = L / (8 * nslot), rem = L - g * 8 * nslot;
    slot = rem >> 3;
    s = g * 8 + (rem & 7);
  } else {
    const int l2 = L - full * nslot, G = S - full;
    slot = l2 / G;
    s = full + (l2 - slot * G);
  }
  tile = slot / 3;
  kh = slot - tile * 3;
}

template <int ABL = 0>
__global__ __launch_bounds__(W8_THREADS) void conv3x3_wgrad_f16x3_w8_kernel(WgradArgsP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wci = wave >> 1, wco = wave & 1;
  const int C = p.C, N = p.N;
  const int ntn = N / WG3_T;
  int bs, kh, tile;
  w8_decode(blockIdx.x, p.S, (C / WG3_T) * ntn, bs, kh, tile);
  const int c0 = (tile / ntn) * WG3_T, n0 = (tile % ntn) * WG3_T;
  const int nchc = C / 16, nchn = N / 16;
  const int pairs_per_img = p.H / WG_ROWS;
  const int total_pairs = p.B * pairs_per_img;
  const int pair_begin = (int)((long long)bs * total_pairs / p.S);
  const int pair_end = (int)((long long)(bs + 1) * total_pairs / p.S);

  f32x16 acc[3][2];   // [kw][co tile]
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

  // Staging.  Unit i (0..3) of a thread: 16 bytes = 8 channels of one plane of one pixel;
  //   tid & 3 -> (plane, channel half), chunk = (tid >> 8) + 2 i, pixel = (tid >> 2) & 63 of the pair's 64.
  // A wave's 16 pixels lie in ONE image row (wrow), so everything that changes from pair to pair is wave-uniform:
  // the row's byte offset goes into the scalar offset of the buffer load, the lane part is a constant, and a row
  // outside the image is one OR of bit 31 into that constant (>= num_records: the load returns zeros).  A load
  // costs its own issue slot and two scalar adds -- the first form of this kernel spent ~15 instructions per load on
  // per-lane address arithmetic, clustered in one MFMA gap: ~600 of 5650 cycles per row pair (profiles/r06_w8_ablation.log).
  i32x4 xreg[4], dreg[4];
  const __amdgpu_buffer_rsrc_t xs_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.xs), 0, 0x80000000, kBufWord3);
  const __amdgpu_buffer_rsrc_t ds_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.dys), 0, 0x80000000, kBufWord3);
  const int u = tid & 3, upl = u >> 1, uh = u & 1;
  const int spix = (tid >> 2) & 63, scc = (tid >> 8) & 1;
  const int wrow = (wave & 3) >> 1;                                    // image row of this wave's units inside the pair
  const unsigned vlane = (unsigned)(((scc * 1024 + (spix & 31)) * 2 + upl) * 32 + uh * 16);
  const int xdst0 = upl * X3_PLANE + (wrow * kPW + (spix & 31) + 1) * 64 + scc * 32 + uh * 16;
  const int ddst0 = 2 * X3_PLANE + upl * D3_PLANE + spix * 64 + scc * 32 + uh * 16;
  const unsigned xtile = (unsigned)((c0 >> 4) * 65536), dtile = (unsigned)((n0 >> 4) * 65536);
  // fetch pointer (the pair whose loads are being issued): image fb, pair fr inside it; scalar offsets and lane word
  int fb = 0, fr = 0;
  unsigned sx = 0, sd = 0, vx = vlane;
  auto set_fetch = [&](int b, int r) {
    fb = b; fr = r;
    const int hx = r * WG_ROWS + kh - 1 + wrow;                        // image row of this wave's x units
    sx = (unsigned)(b * nchc) * 65536u + xtile + (unsigned)(hx * (kW * 64));
    sd = (unsigned)(b * nchn) * 65536u + dtile + (unsigned)((r * WG_ROWS + wrow) * (kW * 64));
    vx = vlane | ((unsigned)hx >= (unsigned)p.H ? 0x80000000u : 0u);
  };
  auto gload_x1 = [&](int i) {
    if (ABL & 2) return;
    xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(xs_rsrc, vx, sx + (unsigned)i * 131072u, 0);
  };
  auto gload_d1 = [&](int i) {
    if (ABL & 2) return;
    dreg[i] = __builtin_amdgcn_raw_buffer_load_b128(ds_rsrc, vlane, sd + (unsigned)i * 131072u, 0);
  };
  auto store_x = [&](unsigned char* buf, int i) {
    if (!(ABL & 1)) *reinterpret_cast<i32x4*>(buf + xdst0 + i * X3_HALF) = xreg[i];
  };
  auto store_d = [&](unsigned char* buf, int i) {
    if (!(ABL & 1)) *reinterpret_cast<i32x4*>(buf + ddst0 + i * D3_HALF) = dreg[i];
  };

  const int grp_q = (lane & 15) >> 2, grp_p = lane & 3, cb16 = ((lane >> 4) & 1) * 16;
  const int lane_off = (8 * lh + grp_q) * 64 + (cb16 + 4 * grp_p) * 2;
  const int xa_off = wci * X3_HALF + lane_off;
  const int db_off = 2 * X3_PLANE + (wco * 2) * D3_HALF + lane_off;

  const bool stamp = p.stamps && blockIdx.x == 0 && threadIdx.x == 0;
  if (stamp) { p.stamps[0] = __builtin_amdgcn_s_memtime(); p.stamps[30] = __builtin_amdgcn_s_memrealtime(); }
  if (tid < 256) {   // the zero halo columns (pixel columns 0 and 33) of both rows, planes, halves and buffers
    const int rec = tid >> 2;
    const int zb = rec >> 5, zpl = (rec >> 4) & 1, zh = (rec >> 2) & 3, zr = (rec >> 1) & 1, zc = (rec & 1) * (kPW - 1);
    const i32x4 z = {0, 0, 0, 0};
    *reinterpret_cast<i32x4*>(smem + zb * WG3_BUF + zpl * X3_PLANE + zh * X3_HALF + (zr * kPW + zc) * 64 + (tid & 3) * 16) = z;
  }
  const int last = pair_end - 1;
  int b_acc = pair_begin < pair_end ? pair_begin / pairs_per_img : 0;
  auto advance_fetch = [&]() {                         // to the next pair of this block's range; stays on the last one
    if (fb * pairs_per_img + fr < last) {
      int r = fr + 1, b = fb;
      if (r == pairs_per_img) { r = 0; ++b; }
      set_fetch(b, r);
    }
  };
  if (pair_begin < pair_end) {
    set_fetch(b_acc, pair_begin - b_acc * pairs_per_img);
#pragma unroll
    for (int i = 0; i < 4; ++i) gload_x1(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) gload_d1(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_x(smem, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_d(smem, i);
    advance_fetch();
#pragma unroll
    for (int i = 0; i < 4; ++i) gload_x1(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) gload_d1(i);
  }
  __syncthreads();
  if (stamp) p.stamps[1] = __builtin_amdgcn_s_memtime();
  int b_now = b_acc, r_now = pair_begin - b_acc * pairs_per_img;     // the pair being multiplied
  for (int pr = pair_begin; pr < pair_end; ++pr) {
    if (stamp && pr - pair_begin < 20) p.stamps[2 + pr - pair_begin] = __builtin_amdgcn_s_memtime();
    const int cur = (pr - pair_begin) & 1;
    const unsigned char* bc = smem + cur * WG3_BUF;
    unsigned char* bn = smem + (cur ^ 1) * WG3_BUF;
    advance_fetch();                                   // -> pair pr + 2 (the registers hold pr + 1)
    if (b_now != b_acc) {                              // next image: move the accumulators to its units (exact powers of two)
      const int d = (clamped_exp(row_max16(p.xmax, b_acc)) - clamped_exp(row_max16(p.xmax, b_now))) +
                    (clamped_exp(row_max16(p.dymax, b_acc)) - clamped_exp(row_max16(p.dymax, b_now)));
      if (d != 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = ldexpf(acc[t][j][r], d);
      }
      b_acc = b_now;
    }
    if (++r_now == pairs_per_img) { r_now = 0; ++b_now; }
    const unsigned char* xa = bc + xa_off;
    const unsigned char* db = bc + db_off;
    f16x8 af[2][2], bfr[2][2][2];                      // [buffer][plane], [buffer][co tile][plane]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) bfr[0][j][pl] = tr_read8h(db + pl * D3_PLANE + j * D3_HALF);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) af[0][pl] = tr_read8h(xa + pl * X3_PLANE);
#pragma unroll
    for (int q = 0; q < 12; ++q) {                     // q = ks * 3 + kw; k step ks = 16 pixels of row ks >> 1
      const int ks = q / 3, kw = q - ks * 3;
      if (q + 1 < 12) {
        const int ks1 = (q + 1) / 3, kw1 = (q + 1) - ks1 * 3;
        const int rr = ks1 >> 1, w0 = (ks1 & 1) * 16;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          af[(q + 1) & 1][pl] = tr_read8h(xa + pl * X3_PLANE + (rr * kPW + w0 + kw1) * 64);
      }
      if (kw == 1 && ks + 1 < 4) {
        const int rr = (ks + 1) >> 1, w0 = ((ks + 1) & 1) * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
            bfr[(ks + 1) & 1][j][pl] = tr_read8h(db + pl * D3_PLANE + j * D3_HALF + (rr * kW + w0) * 64);
      }
      // staging: x unit ks in stage 3 ks, dy unit ks in stage 3 ks + 1; each register refilled at once with pair p + 2
      // (ABL: timing probes, wrong numbers: bit 0 = no LDS stores, bit 1 = no global loads, 16 = two 16x16x32 MFMAs per 32x32x16)
      if (kw == 0) { store_x(bn, ks); gload_x1(ks); }
      if (kw == 1) { store_d(bn, ks); gload_d1(ks); }
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        constexpr int PA[3] = {1, 0, 0};
        constexpr int PB[3] = {0, 1, 0};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr ((ABL & 16) != 0) {         // (timing probe, wrong numbers) two 16x16x32 MFMAs on the same fragments
            f32x16& a16 = acc[kw][j];
            f32x4 c0 = {a16[0], a16[1], a16[2], a16[3]}, c1 = {a16[4], a16[5], a16[6], a16[7]};
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[q & 1][PA[term]], bfr[ks & 1][j][PB[term]], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[q & 1][PA[term]], bfr[ks & 1][j][PB[term]], c1, 0, 0, 0);
            a16[0] = c0[0]; a16[1] = c0[1]; a16[2] = c0[2]; a16[3] = c0[3];
            a16[4] = c1[0]; a16[5] = c1[1]; a16[6] = c1[2]; a16[7] = c1[3];
          } else {
            acc[kw][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[q & 1][PA[term]], bfr[ks & 1][j][PB[term]], acc[kw][j], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < ((ABL & 16) ? 12 : 6); ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (ABL & 16) ? 1 : 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (ABL & 16) ? 1 : 2, 0);
        if (g == ((ABL & 16) ? 3 : 1)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (g == ((ABL & 16) ? 7 : 3)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  if (stamp) p.stamps[22] = __builtin_amdgcn_s_memtime();
  float sdummy, inv_x, inv_g;
  scale_of(row_max16(p.xmax, b_acc), sdummy, inv_x);
  scale_of(row_max16(p.dymax, b_acc), sdummy, inv_g);
  float* slab = p.slab + (size_t)bs * 9 * C * N;
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wco * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wci * 32 + mfma32_row(r, lane);
        slab[((size_t)(kh * 3 + kw) * C + c) * N + n] = (acc[kw][j][r] * inv_x) * inv_g;
      }
    }
  if (stamp) { p.stamps[23] = __builtin_amdgcn_s_memtime(); p.stamps[31] = __builtin_amdgcn_s_memrealtime(); }
}

// ---- the eight-wave block on v_mfma_f32_16x16x32_f16 (round 6): the shipped form.
// The timing probe of the 32x32x16 block above (every MFMA replaced by two 16x16x32 on the same fragments) ran the train
// step 4.5 ms shorter (76.4 -> 72.9 ms): under the chip's power limit the 16x16 shape holds a 12-15 % higher clock for the
// same matrix work (MI355X_MICROARCH.md, clock note 7), and the energy it does not draw is there for the kernels on the
// other stream.  K = 32 of one MFMA is one whole image row of 32 pixels: a row pair is 2 k steps x 3 taps = 6 stages of
// 24 MFMAs per wave (wave = 32 ci x 64 co x 3 taps = 2 x 4 tiles of 16 x 16, 96 accumulators).
// Transposing reads for this operand shape put the four 16-lane groups of a wave on pixels 8 g .. 8 g + 7 of the SAME 16
// channels: with 64-byte pixel records, groups g and g + 1 of a 32-lane bank group would hit the same banks (8 pixels =
// 512 bytes).  The two 16-channel halves of a record therefore swap places on pixel columns with bit 3 set
// (swizzle by the column inside the padded row, applied by the staging stores and by every fragment address): conflict free
// for every tap shift.
// One barrier per row pair, placed BEFORE the last stage: every LDS read of the pair has completed by then, so the last
// stage prefetches the first fragments of the next pair from the other buffer and no wave starts a pair by waiting for
// LDS (the 32x32 block stalls there, both waves of a SIMD at once).
__device__ __forceinline__ f16x8 tr_read8h2(const unsigned char* lo_addr, const unsigned char* hi_addr) {
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lo_addr));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(hi_addr));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(f16x8, v);
}

template <int ABL = 0>
__global__ __launch_bounds__(W8_THREADS) void conv3x3_wgrad_f16x3_w16_kernel(WgradArgsP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wci = wave >> 1, wco = wave & 1;
  const int C = p.C, N = p.N;
  const int ntn = N / WG3_T;
  int bs, kh, tile;
  w8_decode(blockIdx.x, p.S, (C / WG3_T) * ntn, bs, kh, tile);
  const int c0 = (tile / ntn) * WG3_T, n0 = (tile % ntn) * WG3_T;
  const int nchc = C / 16, nchn = N / 16;
  const int pairs_per_img = p.H / WG_ROWS;
  const int total_pairs = p.B * pairs_per_img;
  const int pair_begin = (int)((long long)bs * total_pairs / p.S);
  const int pair_end = (int)((long long)(bs + 1) * total_pairs / p.S);

  f32x4 acc[3][2][4];   // [kw][ci tile][co tile]
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: as in the 32x32 block (wave-uniform row, scalar offsets); the LDS destination carries the swizzle
  i32x4 xreg[4], dreg[4];
  const __amdgpu_buffer_rsrc_t xs_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.xs), 0, 0x80000000, kBufWord3);
  const __amdgpu_buffer_rsrc_t ds_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.dys), 0, 0x80000000, kBufWord3);
  const int u = tid & 3, upl = u >> 1, uh = u & 1;
  const int spix = (tid >> 2) & 63, scol = spix & 31, scc = (tid >> 8) & 1;
  const int wrow = (wave & 3) >> 1;
  const unsigned vlane = (unsigned)(((scc * 1024 + scol) * 2 + upl) * 32 + uh * 16);
  const int xdst0 = upl * X3_PLANE + (wrow * kPW + scol + 1) * 64 + (scc ^ (((scol + 1) >> 3) & 1)) * 32 + uh * 16;
  const int ddst0 = 2 * X3_PLANE + upl * D3_PLANE + spix * 64 + (scc ^ ((scol >> 3) & 1)) * 32 + uh * 16;
  const unsigned xtile = (unsigned)((c0 >> 4) * 65536), dtile = (unsigned)((n0 >> 4) * 65536);
  int fb = 0, fr = 0;
  unsigned sx = 0, sd = 0, vx = vlane;
  auto set_fetch = [&](int b, int r) {
    fb = b; fr = r;
    const int hx = r * WG_ROWS + kh - 1 + wrow;
    sx = (unsigned)(b * nchc) * 65536u + xtile + (unsigned)(hx * (kW * 64));
    sd = (unsigned)(b * nchn) * 65536u + dtile + (unsigned)((r * WG_ROWS + wrow) * (kW * 64));
    vx = vlane | ((unsigned)hx >= (unsigned)p.H ? 0x80000000u : 0u);
  };
  auto gload_x1 = [&](int i) {
    if (ABL & 2) return;
    xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(xs_rsrc, vx, sx + (unsigned)i * 131072u, 0);
  };
  auto gload_d1 = [&](int i) {
    if (ABL & 2) return;
    dreg[i] = __builtin_amdgcn_raw_buffer_load_b128(ds_rsrc, vlane, sd + (unsigned)i * 131072u, 0);
  };
  auto store_x = [&](unsigned char* buf, int i) {
    if (!(ABL & 1)) *reinterpret_cast<i32x4*>(buf + xdst0 + i * X3_HALF) = xreg[i];
  };
  auto store_d = [&](unsigned char* buf, int i) {
    if (!(ABL & 1)) *reinterpret_cast<i32x4*>(buf + ddst0 + i * D3_HALF) = dreg[i];
  };

  // fragment addressing: lane = (k group g = lane >> 4: pixels 8 g .. 8 g + 7; q, pq: the 4 pixels x 4 channel quads a
  // 16-lane group presents to the transposing read).  Padded column of the pixel a lane addresses in read s (0: pixels
  // .. + 3, 1: .. + 7) of tap kw: pc = kw + 8 g + q + 4 s; its swizzle bit (pc >> 3) & 1 = (g + carry) & 1 with
  // carry = (kw + q + 4 s) >> 3 -- three distinct lane patterns: no carry (every s = 0 read, kw = 0, all of dy),
  // (kw = 1, s = 1), (kw = 2, s = 1).
  const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
  const int lane_px = (8 * g + q) * 64 + pq * 8;
  const int par0 = g & 1, par1 = (g + ((1 + q + 4) >> 3)) & 1, par2 = (g + ((2 + q + 4) >> 3)) & 1;
  // x: [variant: 0 no carry, 1 (kw 1, s 1), 2 (kw 2, s 1)][ci tile]
  int xoff[3][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    xoff[0][ti] = wci * X3_HALF + lane_px + (ti ^ par0) * 32;
    xoff[1][ti] = wci * X3_HALF + lane_px + (ti ^ par1) * 32;
    xoff[2][ti] = wci * X3_HALF + lane_px + (ti ^ par2) * 32;
  }
  int doff[2];   // [co sub-tile inside a 32-channel record]
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) doff[tj] = 2 * X3_PLANE + (wco * 2) * D3_HALF + lane_px + (tj ^ par0) * 32;

  f16x8 af[2][2][2], bfr[2][4][2];                     // [buffer][ci tile][plane], [buffer][co tile][plane]
  auto read_a = [&](const unsigned char* buf, int slot, int rr, int kw) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const int o = pl * X3_PLANE + (rr * kPW + kw) * 64;
        af[slot][ti][pl] = tr_read8h2(buf + xoff[0][ti] + o, buf + xoff[kw][ti] + o + 4 * 64);
      }
  };
  auto read_b = [&](const unsigned char* buf, int slot, int rr) {
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const unsigned char* a = buf + doff[tj & 1] + pl * D3_PLANE + (tj >> 1) * D3_HALF + rr * kW * 64;
        bfr[slot][tj][pl] = tr_read8h2(a, a + 4 * 64);
      }
  };

  const bool stamp = p.stamps && blockIdx.x == 0 && threadIdx.x == 0;
  if (stamp) { p.stamps[0] = __builtin_amdgcn_s_memtime(); p.stamps[30] = __builtin_amdgcn_s_memrealtime(); }
  if (tid < 256) {   // the zero halo columns (pixel columns 0 and 33) of both rows, planes, halves and buffers
    const int rec = tid >> 2;
    const int zb = rec >> 5, zpl = (rec >> 4) & 1, zh = (rec >> 2) & 3, zr = (rec >> 1) & 1, zc = (rec & 1) * (kPW - 1);
    const i32x4 z = {0, 0, 0, 0};
    *reinterpret_cast<i32x4*>(smem + zb * WG3_BUF + zpl * X3_PLANE + zh * X3_HALF + (zr * kPW + zc) * 64 + (tid & 3) * 16) = z;
  }
  const int last = pair_end - 1;
  int b_acc = pair_begin < pair_end ? pair_begin / pairs_per_img : 0;
  auto advance_fetch = [&]() {
    if (fb * pairs_per_img + fr < last) {
      int r = fr + 1, b = fb;
      if (r == pairs_per_img) { r = 0; ++b; }
      set_fetch(b, r);
    }
  };
  if (pair_begin < pair_end) {
    set_fetch(b_acc, pair_begin - b_acc * pairs_per_img);
#pragma unroll
    for (int i = 0; i < 4; ++i) gload_x1(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) gload_d1(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_x(smem, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_d(smem, i);
    advance_fetch();
#pragma unroll
    for (int i = 0; i < 4; ++i) gload_x1(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) gload_d1(i);
  }
  // slab reductions handed over by earlier launches on this stream (their slabs are complete: stream order): a share of
  // the outputs per block, behind the operand loads just issued -- a few microseconds of a prologue that waits for
  // memory anyway, instead of a 576-block launch of its own behind every weight gradient (8.8 us alone, 15-55 us beside
  // the main chain's kernels: profiles/r06_step_timeline.log)
  for (int k = 0; k < p.npend; ++k) fold_slab_reduction(p.pend[k], blockIdx.x * W8_THREADS + tid, gridDim.x * W8_THREADS);
  __syncthreads();
  read_b(smem, 0, 0);
  read_a(smem, 0, 0, 0);
  if (stamp) p.stamps[1] = __builtin_amdgcn_s_memtime();
  int b_now = b_acc, r_now = pair_begin - b_acc * pairs_per_img;
  for (int pr = pair_begin; pr < pair_end; ++pr) {
    if (stamp && pr - pair_begin < 20) p.stamps[2 + pr - pair_begin] = __builtin_amdgcn_s_memtime();
    const int cur = (pr - pair_begin) & 1;
    const unsigned char* bc = smem + cur * WG3_BUF;
    unsigned char* bn = smem + (cur ^ 1) * WG3_BUF;
    advance_fetch();                                   // -> pair pr + 2 (the registers hold pr + 1)
    if (b_now != b_acc) {                              // next image: move the accumulators to its units (exact powers of two)
      const int d = (clamped_exp(row_max16(p.xmax, b_acc)) - clamped_exp(row_max16(p.xmax, b_now))) +
                    (clamped_exp(row_max16(p.dymax, b_acc)) - clamped_exp(row_max16(p.dymax, b_now)));
      if (d != 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[t][i][j][r] = ldexpf(acc[t][i][j][r], d);
      }
      b_acc = b_now;
    }
    if (++r_now == pairs_per_img) { r_now = 0; ++b_now; }
#pragma unroll
    for (int st = 0; st < 6; ++st) {                   // st = rr * 3 + kw: image row rr of the pair (one k step), tap kw
      const int rr = st / 3, kw = st - rr * 3;
      if (st == 5) __syncthreads();                    // every LDS read of this pair is done: the other buffer is complete
      if (st < 5) read_a(bc, (st + 1) & 1, (st + 1) / 3, (st + 1) % 3);
      else read_a(bn, 0, 0, 0);                        // first fragments of the next pair
      if (st == 1) read_b(bc, 1, 1);
      if (st == 5) read_b(bn, 0, 0);
      if (st < 4) { store_x(bn, st); gload_x1(st); store_d(bn, st); gload_d1(st); }
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        constexpr int PA[3] = {1, 0, 0};
        constexpr int PB[3] = {0, 1, 0};
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int tj = 0; tj < 4; ++tj)
            acc[kw][ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[st & 1][ti][PA[term]], bfr[rr][tj][PB[term]],
                                                                     acc[kw][ti][tj], 0, 0, 0);
      }
      // the stage as a pipeline: one transposing read behind each of the first 8 (24: the stages that also fetch the
      // eight dy fragments) MFMAs, the two LDS stores and the two global loads of the stage spread behind later ones
#pragma unroll
      for (int gq = 0; gq < 24; ++gq) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (gq < 8 || st == 1 || st == 5) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (st < 4 && (gq == 9 || gq == 13)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (st < 4 && (gq == 11 || gq == 15)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if (stamp) p.stamps[22] = __builtin_amdgcn_s_memtime();
  float sdummy, inv_x, inv_g;
  scale_of(row_max16(p.xmax, b_acc), sdummy, inv_x);
  scale_of(row_max16(p.dymax, b_acc), sdummy, inv_g);
  float* slab = p.slab + (size_t)bs * 9 * C * N;
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const int n = n0 + wco * 64 + tj * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = c0 + wci * 32 + ti * 16 + 4 * g + r;
          slab[((size_t)(kh * 3 + kw) * C + c) * N + n] = (acc[kw][ti][tj][r] * inv_x) * inv_g;
        }
      }
  if (stamp) { p.stamps[23] = __builtin_amdgcn_s_memtime(); p.stamps[31] = __builtin_amdgcn_s_memrealtime(); }
}

// split count of the eight-wave kernel: any S works (w8_decode keeps a pixel range's blocks on one XCD), so a launch
// that owns the chip takes 255-256 CUs; beside the main chain (share_chip) the balance point stays 120 blocks
int wgrad_splits_w8(int B, int H, int C, int N, int share_chip) {
  const int tiles = (C / WG3_T) * (N / WG3_T);
  const int pairs = B * (H / WG_ROWS);
  int target = share_chip == 1 && tiles <= 2 ? 120 : 256;
  if (g_mulan_tune[1] > 0) target = g_mulan_tune[1];
  int S = target / (3 * tiles);
  if (S < 1) S = 1;
  if (S > pairs) S = pairs;
  return S;
}

__global__ void slab_reduce_h_kernel(const float* __restrict__ slab, float* __restrict__ out, int S, int E,
                                     int accumulate) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  float s = 0.f;
  int i = 0;
  for (; i + 20 <= S; i += 20) {         // (S = 40 / 80 in the train step: 20 loads in flight, summed in index order)
    float v[20];
#pragma unroll
    for (int u = 0; u < 20; ++u) v[u] = slab[(size_t)(i + u) * E + e];
#pragma unroll
    for (int u = 0; u < 20; ++u) s += v[u];
  }
  for (; i + 8 <= S; i += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(i + u) * E + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < S; ++i) s += slab[(size_t)i * E + e];
  out[e] = accumulate ? out[e] + s : s;
}

int wgrad_splits_h(int B, int H, int C, int N) {
  const int tiles = ((C + WG_T - 1) / WG_T) * ((N + WG_T - 1) / WG_T);
  const int pairs = B * (H / WG_ROWS);
  const int target = g_mulan_tune[1] > 0 ? g_mulan_tune[1] : 256;
  int S = target / tiles;
  if (S < 1) S = 1;
  if (S > pairs) S = pairs;
  while (S > 1 && pairs / S < 4) --S;
  return S;
}

// ---- once-per-step weight preparation for every eligible parameter leaf at a time (two launches instead of two
// per layer and direction): maxima of all leaves, then both packed operands (forward and gradient) of all leaves.
// Leaf record (8 x int64): [0] element offset in the flat parameter buffer, [1] kind (0: 3x3 conv [3,3,C,N],
// 1: dense [C,N]), [2] C, [3] N, [4] byte offset of the forward operand in the packed buffer or -1, [5] of the
// gradient operand (tap-flipped / transposed) or -1, [6] number of elements.
__global__ __launch_bounds__(256) void param_maxima_kernel(const float* __restrict__ flat, const long long* __restrict__ leaves,
                                                           unsigned* __restrict__ out) {
  __shared__ unsigned red[4];
  const long long* L = leaves + (size_t)blockIdx.x * 8;
  const f32x4* row = reinterpret_cast<const f32x4*>(flat + L[0]);
  const size_t len4 = (size_t)L[6] / 4;
  unsigned m = 0;
  for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < len4; i += (size_t)kMaxParts * 256) {
    const f32x4 a = row[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(a[e]) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[(size_t)blockIdx.x * kMaxParts + blockIdx.y] = max(max(red[0], red[1]), max(red[2], red[3]));
}

__global__ __launch_bounds__(256) void param_pack_kernel(const float* __restrict__ flat, const long long* __restrict__ leaves,
                                                         const unsigned* __restrict__ maxima, unsigned char* __restrict__ packed) {
  const long long* L = leaves + (size_t)blockIdx.x * 8;
  const int dir = blockIdx.y;                       // 0: forward operand, 1: gradient operand
  const long long dst_off = L[4 + dir];
  if (dst_off < 0) return;
  const float* w = flat + L[0];
  const int kind = (int)L[1], C = (int)L[2], N = (int)L[3];
  _Float16* wp = reinterpret_cast<_Float16*>(packed + dst_off);
  float sw, inv_w;
  scale_of(row_max16(maxima, blockIdx.x), sw, inv_w);
  const int taps = kind == 0 ? 9 : 1;
  const int Kin = dir ? N : C, Nout = dir ? C : N;
  const size_t total = (size_t)taps * Kin * Nout;
  for (size_t i = (size_t)blockIdx.z * 256 + threadIdx.x; i < total; i += (size_t)gridDim.z * 256) {
    const int k = (int)(i % 16);
    size_t r = i / 16;
    const int o = (int)(r % Nout); r /= Nout;
    const int cc = (int)(r % (Kin / 16));
    const int t = (int)(r / (Kin / 16));
    const int kin = cc * 16 + k;
    // forward: Wl[t][kin][o] = w[t][kin][o]; gradient: Wl[t][kin][o] = w[taps-1-t][o][kin]   (w is [taps][C][N])
    const float v = dir ? w[((size_t)(taps - 1 - t) * C + o) * N + kin] : w[((size_t)t * C + kin) * N + o];
    _Float16 h, l;
    split2(v * sw, h, l);
    _Float16* dst = wp + (((size_t)(t * (Kin / 16) + cc) * Nout + o) * 2) * 16 + k;
    dst[0] = h; dst[16] = l;
  }
}

}  // namespace

// out[r][0..15] = fp32 bit patterns of 16 partial maxima of |x[r, 0:row_len]| (row_len % 4 == 0, x 16-byte aligned);
// the maximum of a row's 16 entries is the per-image maximum that feeds the power-of-two operand scales of the f16x3
// kernels below.  (The fused GroupNorm kernel writes the same format for its output as a by-product.)
MULAN_API int mulan_absmax_rows(const float* x, unsigned* out, int rows, size_t row_len, hipStream_t stream) {
  if (rows <= 0 || row_len == 0 || row_len % 4 != 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(absmax_rows_kernel, dim3(rows, kMaxParts), dim3(256), 0, stream, x, out, row_len / 4);
  MULAN_CHECK_LAUNCH();
}

// c = a + b (rows x row_len, c may alias a or b) and out = mulan_absmax_rows(c): gradient accumulation of a tensor with
// two consumers fused with the maxima pass of the layer behind it
MULAN_API int mulan_add_absmax_rows(const float* a, const float* b, float* c, unsigned* out, int rows, size_t row_len,
                                    hipStream_t stream) {
  if (rows <= 0 || row_len == 0 || row_len % 4 != 0 || !a || !b || !c || !out) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(add_absmax_rows_kernel, dim3(rows, kMaxParts), dim3(256), 0, stream, a, b, c, out, row_len / 4,
                     static_cast<float*>(nullptr), 0);
  MULAN_CHECK_LAUNCH();
}

// The same with the column sums of c as a by-product: every row is a [row_len / ncols, ncols] matrix (pixels x channels);
// colpart [rows][16][ncols] receives 16 partial column sums per row (the sum of all rows * 16 vectors is the column sum of
// c: the bias gradient of the layer whose output gradient c is, without a second pass over c).  ncols % 4 == 0,
// 256 % (ncols / 4) == 0, row_len % ncols == 0.
MULAN_API int mulan_add_absmax_rows_colsum(const float* a, const float* b, float* c, unsigned* out, float* colpart,
                                           int rows, size_t row_len, int ncols, hipStream_t stream) {
  if (rows <= 0 || row_len == 0 || row_len % 4 != 0 || !a || !b || !c || !out || !colpart || ncols <= 0 || ncols % 4 != 0 ||
      256 % (ncols / 4) != 0 || row_len % ncols != 0)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(add_absmax_rows_kernel, dim3(rows, kMaxParts), dim3(256), 0, stream, a, b, c, out, row_len / 4,
                     colpart, ncols);
  MULAN_CHECK_LAUNCH();
}

MULAN_API size_t mulan_conv3x3_pack_f16x3_bytes(int C, int N) { return (size_t)9 * C * N * 2 * 2; }

// Packs (and scales, splits) the weights for mulan_conv3x3_fwd_f16x3; wmax[16] = mulan_absmax_rows(w, 1 row).
MULAN_API int mulan_conv3x3_pack_f16x3(const float* w, void* wp, const unsigned* wmax, int C, int N, int flip,
                                       hipStream_t stream) {
  const int Kin = flip ? N : C;
  if (Kin % 16 != 0 || C <= 0 || N <= 0 || !wmax) return (int)hipErrorInvalidValue;
  const size_t total = (size_t)9 * C * N;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(conv3x3_pack_f16x3_kernel, dim3(blocks), dim3(256), 0, stream, w, static_cast<_Float16*>(wp), wmax,
                     C, N, flip);
  MULAN_CHECK_LAUNCH();
}

// Eligibility: W == 32, H % 4 == 0, C % 16 == 0, N % 128 == 0 (the ResBlock convolutions); everything else goes
// through mulan_conv3x3_fwd.  xmax = mulan_absmax_rows(x, B rows); wp / wmax from mulan_conv3x3_pack_f16x3.
MULAN_API size_t mulan_conv3x3_planes_bytes(int B, int H, int W, int C) { return (size_t)B * H * W * C * 4; }

// xs (optional, mulan_conv3x3_planes_bytes): receives the split planes of x, the input format of
// mulan_conv3x3_wgrad_f16x3_planes.
// _alone: the same launch with the caller's word that no other stream's kernels share the chip (alone != 0: a forward
// pass, an evaluator, the ODE likelihood's vector-Jacobian product): small launches may then run as k-split blocks
// (conv3x3_f16x3_v3.hip, KS).  mulan_conv3x3_fwd_f16x3 = alone 0.
static int conv3x3_fwd_f16x3_impl(const float* x, const unsigned* xmax, const void* wp, const unsigned* wmax,
                                  const float* bias, const float* cbias, int cbias_mode, const float* res, float* y,
                                  void* xs, unsigned* ymax, int alone, int B, int H, int W, int C, int N,
                                  hipStream_t stream) {
  if (W != kW || H % TROWS != 0 || B <= 0 || C % CK != 0 || C <= 0 || N % BN != 0 || N <= 0 || !xmax || !wmax)
    return (int)hipErrorInvalidValue;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_f16x3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  ConvArgsH a{x, xmax, static_cast<const unsigned char*>(wp), wmax, bias, cbias, res, y, B, H, C, N,
              cbias ? cbias_mode : 0, g_mulan_debug_buffer, static_cast<unsigned char*>(xs), ymax};
  a.alone = alone;
  if (xs && (H % TR2 != 0 || (size_t)B * H * W * C * 4 >= 0x80000000ull)) return (int)hipErrorInvalidValue;
  if (ymax && (H % TR2 != 0 || (H / TR2) * (N / BN) > kMaxParts)) return (int)hipErrorInvalidValue;
  // default: the 16x16x32 / two-blocks-per-CU kernel (conv3x3_f16x3_v3.hip); tune[3] = 2 / 1: dev A/B switches to the
  // 32x32x16 one-block-per-CU kernel / the 2 x 2-tile kernel below
  if (g_mulan_tune[3] == 0 && mulan_conv3x3_f16x3_v3_eligible(H, C, N) && (size_t)B * H * W * C * 4 < 0x80000000ull)
    return mulan_launch_conv3x3_f16x3_v3(a, stream);
  if (H % TR2 == 0 && (g_mulan_tune[3] != 1 || xs || ymax)) {   // tune[3] = 1: dev A/B switch to the 2 x 2-tile variant
    static bool configured2 = false;
    if (!configured2) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_f16x3_v2_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_B);
      if (e != hipSuccess) return (int)e;
      configured2 = true;
    }
    hipLaunchKernelGGL(conv3x3_f16x3_v2_kernel, dim3(B * (H / TR2), N / BN), dim3(256), SMEM2_B, stream, a);
    MULAN_CHECK_LAUNCH();
  }
  dim3 grid(B * (H / TROWS), N / BN);
  hipLaunchKernelGGL(conv3x3_f16x3_kernel, grid, dim3(256), SMEM_B, stream, a);
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_conv3x3_fwd_f16x3(const float* x, const unsigned* xmax, const void* wp, const unsigned* wmax,
                                      const float* bias, const float* cbias, int cbias_mode, const float* res,
                                      float* y, void* xs, unsigned* ymax, int B, int H, int W, int C, int N,
                                      hipStream_t stream) {
  return conv3x3_fwd_f16x3_impl(x, xmax, wp, wmax, bias, cbias, cbias_mode, res, y, xs, ymax, 0, B, H, W, C, N, stream);
}
MULAN_API int mulan_conv3x3_fwd_f16x3_alone(const float* x, const unsigned* xmax, const void* wp, const unsigned* wmax,
                                            const float* bias, const float* cbias, int cbias_mode, const float* res,
                                            float* y, void* xs, unsigned* ymax, int alone, int B, int H, int W, int C,
                                            int N, hipStream_t stream) {
  return conv3x3_fwd_f16x3_impl(x, xmax, wp, wmax, bias, cbias, cbias_mode, res, y, xs, ymax, alone, B, H, W, C, N,
                                stream);
}

// The forward convolution fed with the split planes of its input (mulan_groupnorm_fwd_planes writes them; xmax is the
// [B][16] array that kernel filled with the bound the planes are scaled with).  Same epilogue and by-products as
// mulan_conv3x3_fwd_f16x3, minus the plane output -- the input already is one, for the weight-gradient kernel too.
// Needs the shapes of the two-blocks-per-CU kernel: H % 8 == 0, C % 32 == 0, N % 128 == 0.
// _stats: ystats (optional) receives the partial sums of y and y^2 per (image, row tile of this launch, channel quad),
// [B][H / mulan_conv3x3_f16x3_tile_rows(B, H, N, ymax != NULL)][N / 4][2] -- what mulan_groupnorm_fwd_stream forms the
// statistics of the NEXT GroupNorm from (norm2 behind conv1, the next block's norm1 behind conv2: model_vdm.py:622-644).
// alone != 0: the caller has no other stream's kernels on the chip during this launch (a forward pass, an evaluator):
// launches of at most 256 blocks then run as k-split blocks of eight waves (conv3x3_f16x3_v3.hip, KS).
MULAN_API int mulan_conv3x3_fwd_f16x3_planes_in_stats(const void* xplanes, const unsigned* xmax, const void* wp,
                                                      const unsigned* wmax, const float* bias, const float* cbias,
                                                      int cbias_mode, const float* res, float* y, unsigned* ymax,
                                                      float* ystats, int alone, int B, int H, int W, int C, int N,
                                                      hipStream_t stream) {
  if (W != kW || B <= 0 || C <= 0 || N <= 0 || !xplanes || !xmax || !wmax || !mulan_conv3x3_f16x3_v3_eligible(H, C, N) ||
      (size_t)B * H * W * C * 4 >= 0x80000000ull || (ymax && (H / TR2) * (N / BN) > kMaxParts))
    return (int)hipErrorInvalidValue;
  ConvArgsH a{nullptr, xmax, static_cast<const unsigned char*>(wp), wmax, bias, cbias, res, y, B, H, C, N,
              cbias ? cbias_mode : 0, g_mulan_debug_buffer, nullptr, ymax, static_cast<const unsigned char*>(xplanes)};
  a.ystats = ystats;
  a.alone = alone;
  return mulan_launch_conv3x3_f16x3_v3(a, stream);
}

MULAN_API int mulan_conv3x3_fwd_f16x3_planes_in(const void* xplanes, const unsigned* xmax, const void* wp,
                                                const unsigned* wmax, const float* bias, const float* cbias,
                                                int cbias_mode, const float* res, float* y, unsigned* ymax, int B, int H,
                                                int W, int C, int N, hipStream_t stream) {
  return mulan_conv3x3_fwd_f16x3_planes_in_stats(xplanes, xmax, wp, wmax, bias, cbias, cbias_mode, res, y, ymax, nullptr, 0,
                                                 B, H, W, C, N, stream);
}

// y = conv3x3(act(GroupNorm([x1 | x2]))) + bias + cbias + res with the normalisation done inside the convolution's patch
// fill (ResnetBlock norm1 + swish -> conv1, norm2 + swish -> conv2 where no dropout is drawn: ldm/model_vdm.py:622-656 in
// the evaluators and the sampler): x1 (, x2: equal widths) are the fp32 inputs of the GroupNorm.
//  * xstats1 == NULL: mean / rstd / bound are what mulan_groupnorm_stats left; the result is, bit for bit, that of
//    mulan_groupnorm_fwd_planes + mulan_conv3x3_fwd_f16x3_planes_in.
//  * xstats1 (, xstats2) given: the partial sums the convolutions that PRODUCED x1 (, x2) left through their `ystats`
//    ([B][xstats_tiles][C1 / 4][2]: sum and sum of squares per image, row tile and channel quad; xstats_tiles = H / the
//    producer's tile rows, mulan_conv3x3_f16x3_tile_rows): every block forms mean / rstd
//    and the bound itself -- no pass over x in front of the convolution at all -- and mean / rstd / bound are OUTPUTS (for
//    a later backward pass).  Same formulas, another summation order: statistics agree to fp32 rounding.
//  * ystats (optional, [B][H / mulan_conv3x3_f16x3_tile_rows(B, H, N, ymax != NULL)][N / 4][2]): this launch's partial
//    sums of y for the next GroupNorm.
// The normalised tensor is not written unless yplanes_out (optional, mulan_conv3x3_planes_bytes(B, 32, 32, C1 + C2)
// bytes) asks for it as the weight-gradient kernel's operand.
MULAN_API int mulan_conv3x3_fwd_f16x3_gn_in(const float* x1, const float* x2, int C1, int C2, const float* gamma,
                                            const float* beta, float* mean, float* rstd, int G, int act, float eps,
                                            unsigned* bound, const float* xstats1, const float* xstats2, int xstats_tiles,
                                            const void* wp, const unsigned* wmax, const float* bias, const float* cbias,
                                            int cbias_mode, const float* res, float* y, unsigned* ymax, float* ystats,
                                            void* yplanes_out, int B, int H, int W, int N, hipStream_t stream) {
  const int C = C1 + (x2 ? C2 : 0);
  if (W != kW || B <= 0 || C <= 0 || N <= 0 || G <= 0 || !x1 || !gamma || !beta || !mean || !rstd || !bound || !wmax ||
      (x2 && C2 != C1) || C % G != 0 || (C / G) % 4 != 0 || C > 512 || !mulan_conv3x3_f16x3_v3_eligible(H, C, N) ||
      (size_t)B * H * W * C * 4 >= 0x80000000ull || (ymax && (H / TR2) * (N / BN) > kMaxParts) ||
      (xstats1 && x2 && !xstats2) || (ystats && yplanes_out) ||
      (xstats1 && xstats_tiles != H / 8 && xstats_tiles != H / 4 && xstats_tiles != H / 2))
    return (int)hipErrorInvalidValue;
  ConvArgsH a{x1, bound, static_cast<const unsigned char*>(wp), wmax, bias, cbias, res, y, B, H, C, N,
              cbias ? cbias_mode : 0, g_mulan_debug_buffer, static_cast<unsigned char*>(yplanes_out), ymax, nullptr,
              x2, mean, rstd, gamma, beta, act, G,
              ystats, xstats1, xstats2, xstats1 ? mean : nullptr, xstats1 ? rstd : nullptr, xstats1 ? bound : nullptr, eps,
              0, xstats_tiles, 1};       // alone: forward-only paths (and the forward pass of a train step)
  return mulan_launch_conv3x3_f16x3_v3(a, stream);
}

// Image rows per block (8, 4 or 2) of the two-blocks-per-CU convolution kernel for a launch of B images with N output
// channels: the tallest tile that still gives the launch two blocks per CU.  `ystats` of mulan_conv3x3_fwd_f16x3_gn_in
// holds H / rows row tiles per image ([B][H / rows][N / 4][2]): the caller sizes it with this function and hands the count
// on as `xstats_tiles`.  with_ymax: the launch also writes the maxima array (16 partials per image limit the tile count).
MULAN_API int mulan_conv3x3_f16x3_tile_rows(int B, int H, int N, int with_ymax) {
  if (B <= 0 || H % 8 != 0 || N <= 0 || N % BN != 0) return 8;
  return mulan_conv3x3_f16x3_v3_tile_rows(B, H, N, with_ymax != 0);
}

MULAN_API size_t mulan_conv3x3_wgrad_f16x3_workspace(int B, int H, int W, int C, int N) {
  if (W != kW || H % WG_ROWS != 0) return 0;
  return (size_t)wgrad_splits_h(B, H, C, N) * 9 * C * N * sizeof(float);
}

// dw[3,3,C,N] (+)= sum x (x) dy with the 3-pass fp16 split; xmax / dymax are the per-image maxima [B] of x and dy.
MULAN_API int mulan_conv3x3_wgrad_f16x3(const float* x, const unsigned* xmax, const float* dy, const unsigned* dymax,
                                        float* dw, float* workspace, int B, int H, int W, int C, int N, int accumulate,
                                        hipStream_t stream) {
  if (W != kW || H % WG_ROWS != 0 || B <= 0 || C % 4 != 0 || N % 4 != 0 || !xmax || !dymax)
    return (int)hipErrorInvalidValue;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16x3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WG_SMEM);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  const int S = wgrad_splits_h(B, H, C, N);
  WgradArgsH a{x, dy, workspace, xmax, dymax, B, H, C, N, S};
  dim3 grid(S, (C + WG_T - 1) / WG_T, (N + WG_T - 1) / WG_T);
  hipLaunchKernelGGL(conv3x3_wgrad_f16x3_kernel, grid, dim3(256), WG_SMEM, stream, a);
  const int E = 9 * C * N;
  hipLaunchKernelGGL(slab_reduce_h_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, workspace, dw, S, E, accumulate);
  MULAN_CHECK_LAUNCH();
}

static int launch_wgrad_w16(const void* xs, const unsigned* xmax, const void* dys, const unsigned* dymax, float* workspace,
                            int B, int H, int C, int N, int S8, const mulan_slab_reduction* pending, int n_pending,
                            int abl, hipStream_t stream) {
  static bool configured16 = false;
  if (!configured16) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16x3_w16_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WG3_SMEM + 64);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16x3_w16_kernel<3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, WG3_SMEM + 64);
    if (e != hipSuccess) return (int)e;
    configured16 = true;
  }
  WgradArgsP a8{static_cast<const unsigned char*>(xs), static_cast<const unsigned char*>(dys), xmax, dymax, workspace,
                B, H, C, N, S8, g_mulan_debug_buffer};
  a8.npend = n_pending;
  for (int k = 0; k < n_pending; ++k) a8.pend[k] = pending[k];
  const dim3 grid8(3 * S8 * (C / WG3_T) * (N / WG3_T));
  if (abl == 3)
    hipLaunchKernelGGL(conv3x3_wgrad_f16x3_w16_kernel<3>, grid8, dim3(W8_THREADS), WG3_SMEM + 64, stream, a8);
  else
    hipLaunchKernelGGL(conv3x3_wgrad_f16x3_w16_kernel<0>, grid8, dim3(W8_THREADS), WG3_SMEM + 64, stream, a8);
  return 0;
}

// ---- slab reductions folded into the NEXT weight-gradient launch (round 6).  In the backward pass of a train step the
// 3x3 weight gradients follow one another on one stream; a launch made with _fold writes its slabs and does NOT sum
// them: the caller hands the record {slab, out, S, E, accumulate} to a later _fold launch (up to two records per launch,
// summed by its blocks' prologues in slab order: the bits of the separate reduction kernel) or to mulan_slab_reduce.
MULAN_API int mulan_conv3x3_wgrad_f16x3_planes_splits(int B, int H, int W, int C, int N, int share_chip) {
  if (W != kW || H % WG_ROWS != 0 || B <= 0 || C % WG3_T != 0 || N % WG3_T != 0) return 0;
  return wgrad_splits_w8(B, H, C, N, share_chip);
}

MULAN_API int mulan_conv3x3_wgrad_f16x3_planes_fold(const void* xs, const unsigned* xmax, const void* dys,
                                                    const unsigned* dymax, float* workspace, int B, int H, int W, int C,
                                                    int N, int share_chip, const mulan_slab_reduction* pending,
                                                    int n_pending, hipStream_t stream) {
  if (W != kW || H % WG_ROWS != 0 || B <= 0 || C % WG3_T != 0 || N % WG3_T != 0 || !xs || !dys || !xmax || !dymax ||
      !workspace || n_pending < 0 || n_pending > 2 || (n_pending > 0 && !pending) ||
      (size_t)B * H * W * (C > N ? C : N) * 4 >= 0x80000000ull)
    return (int)hipErrorInvalidValue;
  for (int k = 0; k < n_pending; ++k)
    if (!pending[k].slab || !pending[k].out || pending[k].S <= 0 || pending[k].E <= 0 || pending[k].E % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(pending[k].slab) | reinterpret_cast<uintptr_t>(pending[k].out)) % 16 != 0)
      return (int)hipErrorInvalidValue;
  const int rc = launch_wgrad_w16(xs, xmax, dys, dymax, workspace, B, H, C, N, wgrad_splits_w8(B, H, C, N, share_chip),
                                  pending, n_pending, 0, stream);
  if (rc != 0) return rc;
  MULAN_CHECK_LAUNCH();
}

MULAN_API int mulan_slab_reduce(const float* slab, float* out, int S, int E, int accumulate, hipStream_t stream) {
  if (!slab || !out || S <= 0 || E <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(slab_reduce_h_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, slab, out, S, E, accumulate);
  MULAN_CHECK_LAUNCH();
}

MULAN_API size_t mulan_conv3x3_wgrad_f16x3_planes_workspace(int B, int H, int W, int C, int N, int share_chip) {
  if (W != kW || H % WG_ROWS != 0 || C % WG3_T != 0 || N % WG3_T != 0) return 0;
  const int Sw = wgrad_splits_w8(B, H, C, N, share_chip), Sp = wgrad_splits_p(B, H, C, N, share_chip);
  return (size_t)(Sw > Sp ? Sw : Sp) * 9 * C * N * sizeof(float);     // (either kernel: tune[29] picks at launch time)
}

// dw[3,3,C,N] (+)= sum x (x) dy from the split planes written by mulan_conv3x3_fwd_f16x3 (xs: of the forward input,
// dys: of the output gradient, written by the input-gradient convolution); needs C % 128 == 0 and N % 128 == 0.
MULAN_API int mulan_conv3x3_wgrad_f16x3_planes(const void* xs, const unsigned* xmax, const void* dys,
                                               const unsigned* dymax, float* dw, float* workspace, int B, int H, int W,
                                               int C, int N, int accumulate, int share_chip, hipStream_t stream) {
  if (W != kW || H % WG_ROWS != 0 || B <= 0 || C % WG3_T != 0 || N % WG3_T != 0 || !xs || !dys || !xmax || !dymax ||
      (size_t)B * H * W * (C > N ? C : N) * 4 >= 0x80000000ull)
    return (int)hipErrorInvalidValue;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16x3_planes_kernel<3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WG3_SMEM + 64);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  const int E = 9 * C * N;
  if (g_mulan_tune[29] != 1) {        // the eight-wave block (default); tune[29] = 1: the four-wave block (dev A/B)
    static bool configured8 = false;
    if (!configured8) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16x3_w8_kernel<0>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, WG3_SMEM + 64);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16x3_w8_kernel<3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, WG3_SMEM + 64);
      if (e != hipSuccess) return (int)e;
      configured8 = true;
    }
    const int S8 = wgrad_splits_w8(B, H, C, N, share_chip);
    WgradArgsP a8{static_cast<const unsigned char*>(xs), static_cast<const unsigned char*>(dys), xmax, dymax, workspace,
                  B, H, C, N, S8, g_mulan_debug_buffer};
    const dim3 grid8(3 * S8 * (C / WG3_T) * (N / WG3_T));
#define MULAN_W8_ABL(V)                                                                                          \
  case V:                                                                                                          \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16x3_w8_kernel<V>),                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, WG3_SMEM + 64);                          \
    hipLaunchKernelGGL(conv3x3_wgrad_f16x3_w8_kernel<V>, grid8, dim3(W8_THREADS), WG3_SMEM + 64, stream, a8);      \
    break;
    if (g_mulan_tune[29] == 0) {      // 16x16x32 block (shipped)
      const int rc = launch_wgrad_w16(xs, xmax, dys, dymax, workspace, B, H, C, N, S8, nullptr, 0, g_mulan_tune[7], stream);
      if (rc != 0) return rc;
    } else
    switch (g_mulan_tune[7]) {   // tune[29] = 2: the 32x32x16 eight-wave block; dev: timing probes (wrong numbers)
      MULAN_W8_ABL(2) MULAN_W8_ABL(3) MULAN_W8_ABL(16) MULAN_W8_ABL(19)
      default:
        hipLaunchKernelGGL(conv3x3_wgrad_f16x3_w8_kernel<0>, grid8, dim3(W8_THREADS), WG3_SMEM + 64, stream, a8);
    }
#undef MULAN_W8_ABL
    hipLaunchKernelGGL(slab_reduce_h_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, workspace, dw, S8, E, accumulate);
    MULAN_CHECK_LAUNCH();
  }
  const int S = wgrad_splits_p(B, H, C, N, share_chip);
  WgradArgsP a{static_cast<const unsigned char*>(xs), static_cast<const unsigned char*>(dys), xmax, dymax, workspace,
               B, H, C, N, S, g_mulan_debug_buffer};
  const dim3 grid(S, 3, (C / WG3_T) * (N / WG3_T));
#define MULAN_WG_ABL(PROBE, ABLV)                                                                                       \
  {                                                                                                                     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16x3_planes_kernel<3, false, PROBE, ABLV>),  \
                        hipFuncAttributeMaxDynamicSharedMemorySize, WG3_SMEM + 64);                                     \
    hipLaunchKernelGGL((conv3x3_wgrad_f16x3_planes_kernel<3, false, PROBE, ABLV>), grid, dim3(256), WG3_SMEM + 64,      \
                       stream, a);                                                                                      \
  }
  switch (g_mulan_tune[7]) {   // dev: timing probes (wrong numbers)
    case 1: MULAN_WG_ABL(true, 0) break;
    case 2: MULAN_WG_ABL(false, 2) break;
    case 3: MULAN_WG_ABL(false, 3) break;
    case 4: MULAN_WG_ABL(false, 4) break;
    case 5: MULAN_WG_ABL(false, 5) break;
    case 6: MULAN_WG_ABL(false, 6) break;
    default:
      hipLaunchKernelGGL(conv3x3_wgrad_f16x3_planes_kernel<3>, grid, dim3(256), WG3_SMEM + 64, stream, a);
  }
#undef MULAN_WG_ABL
  hipLaunchKernelGGL(slab_reduce_h_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, workspace, dw, S, E, accumulate);
  MULAN_CHECK_LAUNCH();
}

// Maxima of n parameter leaves in one launch (leaf records: see param_maxima_kernel); out is [n][16].
MULAN_API int mulan_param_maxima(const float* flat, const long long* leaves, int n, unsigned* out, hipStream_t stream) {
  if (n <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(param_maxima_kernel, dim3(n, kMaxParts), dim3(256), 0, stream, flat, leaves, out);
  MULAN_CHECK_LAUNCH();
}

// Both packed f16x3 operands (forward: mulan_conv3x3_pack_f16x3 flip = 0 / mulan_linear_pack_f16x3 transpose = 0;
// gradient: flip = 1 / transpose = 1) of n parameter leaves in one launch, into `packed` at the records' offsets.
MULAN_API int mulan_param_pack_f16x3(const float* flat, const long long* leaves, int n, const unsigned* maxima,
                                     void* packed, hipStream_t stream) {
  if (n <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(param_pack_kernel, dim3(n, 2, 16), dim3(256), 0, stream, flat, leaves, maxima,
                     static_cast<unsigned char*>(packed));
  MULAN_CHECK_LAUNCH();
}

// ---- weight gradient of a per-pixel dense layer from split planes: dw[C, N] (+)= x^T dy, x planes [B][C/16][HW][2][16]
// (handed on by mulan_linear_f16x3), dy planes [B][N/16][HW][2][16] (handed on by the convolution that consumed the same
// dy); xmax / dymax are the per-image maxima the planes were scaled with.  Needs C % 128 == 0, N % 128 == 0.
static int linear_wgrad_splits(int B, int H, int C, int N, int share_chip) {
  const int tiles = (C / WG3_T) * (N / WG3_T);
  const int pairs = B * (H / WG_ROWS);
  const int target = share_chip == 1 ? 160 : 240;     // (shared chip, see wgrad_splits_p: scan 64 ... 240)
  int S = target / tiles;
  if (S < 1) S = 1;
  if (S > pairs) S = pairs;
  return S;
}

MULAN_API size_t mulan_linear_wgrad_f16x3_planes_workspace(int B, int H, int W, int C, int N, int share_chip) {
  if (W != kW || H % WG_ROWS != 0 || C % WG3_T != 0 || N % WG3_T != 0) return 0;
  return (size_t)linear_wgrad_splits(B, H, C, N, share_chip) * C * N * sizeof(float);
}

MULAN_API int mulan_linear_wgrad_f16x3_planes(const void* xs, const unsigned* xmax, const void* dys,
                                              const unsigned* dymax, float* dw, float* workspace, int B, int H, int W,
                                              int C, int N, int accumulate, int share_chip, hipStream_t stream) {
  if (W != kW || H % WG_ROWS != 0 || B <= 0 || C % WG3_T != 0 || N % WG3_T != 0 || !xs || !dys || !xmax || !dymax ||
      (size_t)B * H * W * (C > N ? C : N) * 4 >= 0x80000000ull)
    return (int)hipErrorInvalidValue;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16x3_planes_kernel<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WG3_SMEM + 64);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  const int S = linear_wgrad_splits(B, H, C, N, share_chip);
  WgradArgsP a{static_cast<const unsigned char*>(xs), static_cast<const unsigned char*>(dys), xmax, dymax, workspace,
               B, H, C, N, S, g_mulan_debug_buffer};
  hipLaunchKernelGGL(conv3x3_wgrad_f16x3_planes_kernel<1>, dim3(S, 1, (C / WG3_T) * (N / WG3_T)), dim3(256),
                     WG3_SMEM + 64, stream, a);
  const int E = C * N;
  hipLaunchKernelGGL(slab_reduce_h_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, workspace, dw, S, E, accumulate);
  MULAN_CHECK_LAUNCH();
}

// The same weight gradient with x in fp32 (round 3): dw[C1 + C2, N] (+)= [x1 | x2]^T dy, x1 [B, H*W, C1] and x2 [B, H*W, C2]
// (x2 / C2 may be NULL / 0) read as they are and split while staged, dy as the planes the convolution that consumed the
// same dy handed on.  xmax, xmax2 [B][16]: the maxima of x1 and of x2 (the kernel takes their elementwise max: the maxima
// of the concat), i.e. the scale the layer's forward pass used: results equal mulan_linear_wgrad_f16x3_planes on the planes that pass would have
// written, bit for bit -- which the forward kernel therefore need not write.  C1 and C2 multiples of 128.
MULAN_API size_t mulan_linear_wgrad_f16x3_x32_workspace(int B, int H, int W, int C, int N, int share_chip) {
  return mulan_linear_wgrad_f16x3_planes_workspace(B, H, W, C, N, share_chip);
}

MULAN_API int mulan_linear_wgrad_f16x3_x32(const float* x1, const float* x2, int C1, int C2, const unsigned* xmax,
                                           const unsigned* xmax2, const void* dys, const unsigned* dymax, float* dw, float* workspace, int B,
                                           int H, int W, int N, int accumulate, int share_chip, hipStream_t stream) {
  const int C = C1 + C2;
  if (W != kW || H != 32 || B <= 0 || C1 <= 0 || C1 % WG3_T != 0 || C2 < 0 || C2 % WG3_T != 0 || N % WG3_T != 0 || !x1 ||
      (C2 > 0 && (!x2 || !xmax2)) || !dys || !xmax || !dymax || (size_t)B * H * W * (C > N ? C : N) * 4 >= 0x80000000ull)
    return (int)hipErrorInvalidValue;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16x3_planes_kernel<1, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WG3_SMEM + 64);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  const int S = linear_wgrad_splits(B, H, C, N, share_chip);
  WgradArgsP a{nullptr, static_cast<const unsigned char*>(dys), xmax, dymax, workspace, B, H, C, N, S, g_mulan_debug_buffer,
               x1, x2, C1, C2 > 0 ? xmax2 : nullptr};
  hipLaunchKernelGGL((conv3x3_wgrad_f16x3_planes_kernel<1, true>), dim3(S, 1, (C / WG3_T) * (N / WG3_T)), dim3(256),
                     WG3_SMEM + 64, stream, a);
  const int E = C * N;
  hipLaunchKernelGGL(slab_reduce_h_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, workspace, dw, S, E, accumulate);
  MULAN_CHECK_LAUNCH();
}

// out[b] = x[b]^T @ dy[b] per image:  [C, N] = sum over the H * 32 rows of image b of x[row, :]^T dy[row, :], both
// operands as split planes (the by-products of mulan_linear_f16x3 / _batched) scaled with xmax[b] / dymax[b].  The
// attention core's dV = P^T dO and dK = dS^T Q (autodiff of ldm/model_vdm.py:775-796) on the dense weight-gradient
// kernel with one slab per image and no slab reduction.
MULAN_API int mulan_bmm_tn_f16x3_planes(const void* xs, const unsigned* xmax, const void* dys, const unsigned* dymax,
                                        float* out, int B, int H, int W, int C, int N, hipStream_t stream) {
  if (W != kW || H % WG_ROWS != 0 || B <= 0 || C % WG3_T != 0 || N % WG3_T != 0 || !xs || !dys || !xmax || !dymax || !out ||
      (size_t)B * H * W * (C > N ? C : N) * 4 >= 0x80000000ull)
    return (int)hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16x3_planes_kernel<1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, WG3_SMEM + 64);
  if (e != hipSuccess) return (int)e;
  WgradArgsP a{static_cast<const unsigned char*>(xs), static_cast<const unsigned char*>(dys), xmax, dymax, out,
               B, H, C, N, B, g_mulan_debug_buffer};
  hipLaunchKernelGGL(conv3x3_wgrad_f16x3_planes_kernel<1>, dim3(B, 1, (C / WG3_T) * (N / WG3_T)), dim3(256),
                     WG3_SMEM + 64, stream, a);
  MULAN_CHECK_LAUNCH();
}
