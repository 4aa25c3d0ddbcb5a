// 3x3 convolution, f16x3 arithmetic (see conv3x3_f16x3.hip), third generation: the forward AND input-gradient kernel of
// the ResnetBlock convolutions (ldm/model_vdm.py:633-650) for C % 32 == 0, N % 128 == 0.
//
// What changed against conv3x3_f16x3_v2_kernel, and why (profiles/r02_shape_probe.log, tools/shape_probe.hip):
//  * TWO co-resident blocks per CU (<= 256 registers, 64 KB of LDS each) with a static priority for every second one.
//    With one block per CU the epilogue of a tile (15-21 k of 85 k cycles: 128 KB read + 128 KB written per CU, all CUs
//    at the same moment) leaves the matrix cores idle.  s_setprio is strict for MFMA issue (a prioritised wave runs at
//    its solo speed, the other one fills every gap), so the prioritised block runs ahead and each block's prologue /
//    epilogue is covered by the other block's main loop; blocks i and i + 256 share a CU (breadth-first dispatch,
//    measured; used for speed only).
//  * v_mfma_f32_16x16x32_f16 instead of 32x32x16: the chip is power limited under this load (a bare 32x32x16 loop holds
//    1.69 GHz) and holds a ~19 % higher clock on the 16x16 shape at the same operand traffic.
//  * Roles swapped: A = weights (rows = couts), B = pixels (columns), so a lane's 4 accumulator registers of a tile are
//    4 consecutive couts of one pixel: the epilogue adds bias / FiLM bias / residual and stores float4s straight from
//    the accumulator layout, no LDS transpose.
//
// K = 32 of one MFMA = two "units" (16-channel chunk, tap): lanes 0-31 carry unit 2s, lanes 32-63 unit 2s + 1 of the
// sequence u = chunk * 9 + tap; nine steps cover a chunk pair.  Block = 8 image rows x 128 couts, wave = all 8 rows x 32
// couts = 16 pixel tiles x 2 cout tiles (128 accumulator registers; only 16 registers of weight fragments per step, and
// no two waves fetch the same weights).  Patches: [plane][row][col][16 ch] fp16, 32 B per
// pixel and plane (conflict-free for the ds_read_b128 lane groups of this operand layout), three 16-channel buffers in
// rotation (chunk c, c + 1 being multiplied, c + 2 / c + 3 being filled), two barriers per chunk pair.  Weight
// fragments come straight from the packed tensor (L1 / L2 resident), one step ahead.
#include <type_traits>
#include "common.h"
#include "f16x3_common.h"

namespace {

using namespace f16x3;

// Tile height TR (image rows per block): 8 (the shape of rounds 2-4: 16 pixel tiles per wave), 4 or 2.  With 8 rows a
// launch has B * 4 * (N / 128) blocks: one dispatch round of two blocks per CU at B = 128, ONE block per CU at 64 images
// per GPU (the 8-GPU operating point of BASELINE configs[2]: no co-resident partner, 0.487 against 0.533) and a quarter of
// the chip at the 16 images per GPU of a sampling batch.  Shorter tiles (round 5) give those launches their two blocks
// per CU back; the price is the halo (TR + 2 patch rows per TR rows of output) and half / a quarter of the MFMAs per
// weight-fragment load.  Everything below is derived from TR.
constexpr int WDS = 3;                              // weight-fragment ring of the short-tile instantiations (3 or 9: see the kernel)
constexpr int GN_ENT = 48;                          // GroupNorm-fed fill: table entry (scale x 4, beta x 4, mean) per channel quad
constexpr int GN_MAXC = 512;
template <int TR> struct V3Geo {
  static constexpr int ROWS = TR + 2;               // patch rows (halo above and below)
  static constexpr int PLANE = ROWS * kPW * 32;     // one plane of one 16-channel chunk (TR = 8: 10880 B)
  static constexpr int BUF = 2 * PLANE;
  static constexpr int DUMMY = 3 * BUF;             // dummy target for the slots past the patch (512 B) / for the DMA of a row
  static constexpr int SMEM = 3 * BUF + 1024;       // outside the image (1 KB).  TR = 8: 66304 -- two blocks per CU fit the 160 KB
  static constexpr int PV = (ROWS * kPW * 4 + 255) / 256;   // float4 patch slots per thread and chunk (TR = 8: 6, 1360 of 1536 used)
  static constexpr int NK = (PV + 1) / 2;           // ... in pairs (one register set each): 3, 2, 2
  static constexpr int PPW = (2 * ROWS) / 4;        // LDS-DMA pieces (plane, row) per wave and chunk: 5, 3, 2
  static constexpr int GN_SLOT = SMEM + GN_MAXC / 4 * GN_ENT;   // 16 floats behind the table: [0] bound, [4..11] block reduction
  static constexpr int SMEM_GN = GN_SLOT + 64;      // TR = 8: 72000
  static_assert(TR == 8 || TR == 4 || TR == 2, "tile height");
  static_assert(3 * BUF >= 16 * (GN_MAXC / 4) * 8, "the statistics prologue stages 16 tiles x C / 4 partial sums in the patch buffers");
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

// acc += A B in place on accumulator registers ("a" class): the 128 accumulator registers stay out of the compiler's
// allocation games (with the builtin it rotates them through the vector registers and spills at the 256-register cap).
// PAD 1: two wait states in front for a VALU-written operand (the weight-fragment copy at the end of a chunk pair).
// PAD 2: the last MFMA of a chunk pair (the loop may end behind it).  The compiler does not know that these statements
// are matrix instructions and handles none of their hazards: at the loop exit it reads accumulators back (register
// shuffles, spills) right away, and a v_accvgpr_read that comes too early returns the value from BEFORE the last MFMA
// -- seen as wrong outputs in the last pixel tile, in blocks that run without other stalls (raised priority, B >= 128).
// The wait therefore sits inside the same statement, where nothing can be scheduled in front of it; it overlaps the
// barrier that follows, and the other block of the CU issues meanwhile.
template <int PAD>
__device__ __forceinline__ void mfma16(f32x4v& acc, const f16x3::f16x8& a, const f16x3::f16x8& b, bool) {
  if (PAD == 1) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else if (PAD == 2)
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                 : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// ABL: dev-only ablation bits (timing only, results wrong): 1 no LDS fragment reads, 2 no weight loads, 4 no patch
// fills, 8 no barriers in the main loop, 32 patch fill without the split arithmetic, 16 without the split arithmetic
// and without the plane stores
// PIN: the input arrives as split planes (written by the GroupNorm in front, mulan_groupnorm_fwd_planes): the fill is a
// 16-byte copy per slot -- no split arithmetic and, above all, no plane stores out of this kernel (they cost 5-13 % of
// a launch: conv_ab ablation 16 / 32 of round 2).
// GNF (fp32 input only): the input is the fp32 tensor IN FRONT of a GroupNorm (+ SiLU); the fill normalises, activates,
// clamps and splits each float4 with the arithmetic of gn_fwd_kernel's planes mode (same expressions, same bound and
// scale: the patches hold bit for bit what mulan_groupnorm_fwd_planes would have handed over), from a per-block table
// (scale = gamma rstd, beta, mean per channel quad of this image) in LDS.  GNF = 1: nothing else (forward-only paths: the
// normalised tensor never reaches HBM); GNF = 2: the planes are also stored (xs) for a weight-gradient kernel.
// DMA (PIN only, round 4): the patch fill is an LDS-DMA (global_load_lds_dwordx4): one wave instruction moves the 32
// interior pixels of one (plane, row) of a chunk's patch -- 64 lanes x 16 B = 1 KB, contiguous in the [plane][row][col][16 ch]
// image -- straight into LDS; no staging registers, no ds_write, no per-slot address arithmetic in the MFMA shadows.  A
// chunk is 2 planes x 10 rows = 20 instructions, 5 per wave, issued in one step (chunk 2 j + 2 into buffer C in step 0,
// chunk 2 j + 3 into buffer A in step 5, right behind the barrier that frees it).  The halo columns and the rows outside
// the image are never written: the buffers are zeroed once at the start (a row outside the image is DMA'd into the dummy
// area instead: no branch in the loop body).  The barriers of the loop wait for the DMAs (vmcnt) like for any LDS write.
// KS (round 5): k-split.  KS = 2: the block is two groups of four waves; group g runs the loop above over its half of the
// chunk pairs in its own LDS region (patches, GroupNorm table), the barriers are shared (both groups execute the same
// schedule), and at the end the groups swap halves of their accumulators through LDS and each finishes half of the pixel
// tiles.  For launches of at most 256 blocks -- one block per CU, one wave per SIMD, nothing to fill the stalls of the
// main loop (3 x the MFMA issue time at 16 images, profiles/r05_gnf_timeline.log): two waves per SIMD without a second
// patch fill, without more LDS reads per MFMA (an 8-wave block of 16-cout waves would read every pixel fragment twice)
// and without traffic through memory (a split-k PAIR of blocks would exchange the 32 MB output tile through HBM).  The
// sum of an output element is (first half of the channels) + (second half): deterministic, not the bits of KS = 1.
template <int ABL, bool PIN = false, int GNF = 0, bool DMA = false, int TR = 8, int KS = 1>
__global__ __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) void conv3x3_f16x3_v3_kernel(ConvArgsH p) {
  static_assert(!(PIN && GNF), "GroupNorm-fed fill reads fp32");
  static_assert(!DMA || PIN, "LDS-DMA fill: plane-fed instantiation only");
  static_assert(KS == 1 || (KS == 2 && ABL == 0), "k-split: one or two groups of four waves");
  using Geo = V3Geo<TR>;
  constexpr int TR3 = TR, NPT = 2 * TR;              // NPT: 16-pixel tiles per wave (two per image row)
  constexpr int P3_ROWS = Geo::ROWS, P3_PLANE = Geo::PLANE, P3_BUF = Geo::BUF, P3_DUMMY = Geo::DUMMY, SMEM3_B = Geo::SMEM;
  constexpr int PV3 = 2 * Geo::NK, NK = Geo::NK, PPW = Geo::PPW, GN_SLOT = Geo::GN_SLOT;
  // places of the patch traffic inside a step of NPT pixel tiles: the two slots of a pair are stored behind tiles ST0 / ST1
  // and fetched behind LD0 / LD1; the four weight fragments of the next step behind tiles WL0 .. WL0 + 3
  constexpr int ST0 = NPT == 16 ? 6 : (NPT == 8 ? 3 : 0), ST1 = NPT == 16 ? 8 : (NPT == 8 ? 4 : 1);
  constexpr int LD0 = NPT == 16 ? 10 : (NPT == 8 ? 5 : 2), LD1 = NPT == 16 ? 12 : (NPT == 8 ? 6 : 3);
  constexpr int WL0 = NPT > 4 ? 1 : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_blk[];
  constexpr int REGION = GNF ? Geo::SMEM_GN : Geo::SMEM;      // LDS of one group
  static_assert(KS == 1 || KS * REGION >= NPT * 8192 + 512, "k-split: the accumulator exchange + the tail reductions fit the two regions");
  // (KS = 2) tid / wave: within the group; everything below that is indexed by them or lives in `smem` is per group
  const int tid_b = threadIdx.x;
  const int kgrp = KS == 2 ? __builtin_amdgcn_readfirstlane(tid_b >> 8) : 0;
  unsigned char* const smem = smem_blk + kgrp * REGION;
  const int tid = KS == 2 ? (tid_b & 255) : tid_b, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, grp = lane >> 4, sel = lane >> 5, khalf = grp & 1;
  const int tiles_per_img = p.H / TR3;
  // Block -> (pixel tile, cout block).  With two or more cout blocks every one of them reads the same patch: dispatched
  // gridDim.x blocks apart (the plain 2-D order) the second read misses the caches once the input outgrows them
  // (B = 1000 copies in the dense evaluator: 1 GB of planes per launch).  Paired (ids 8 apart: the same XCD under the
  // round-robin dispatch, a few hundred ns apart) the re-read hits that XCD's L2.  g_pair: set by the launcher.
  const int lin = blockIdx.y * gridDim.x + blockIdx.x;
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.pair_cols) { const int per = 8 * (int)gridDim.y; bx = (lin / per) * 8 + (lin & 7); by = (lin % per) >> 3; }
  const int b = bx / tiles_per_img;
  const int h0 = (bx % tiles_per_img) * TR3;
  const int n0 = by * BN;
  const int C = p.C, N = p.N;
  const int nchunks = C / CK, npairs = nchunks / 2;
  const int j0 = KS == 2 ? kgrp * (npairs / 2) : 0, j1 = KS == 2 ? j0 + npairs / 2 : npairs;   // this group's chunk pairs
  const int ldx = (GNF && p.x2) ? C / 2 : C;          // channels per pixel of the tensor(s) behind x (, x2)
  const int nch1 = ldx / CK;
  // every second block of a CU runs ahead (blocks i and i + 256 share a CU under breadth-first dispatch)
  if (KS == 1 && (((blockIdx.y * gridDim.x + blockIdx.x) >> 8) & 1)) __builtin_amdgcn_s_setprio(1);
  float sx, inv_x, sw, inv_w;
  scale_of(row_max16(p.xmax, b), sx, inv_x);
  scale_of(row_max16(p.wmax, 0), sw, inv_w);
  // dev-only timeline (mulan_set_debug_buffer; tools/conv_ab.py --timeline): per block start / loop start / loop end / end
  const unsigned tl_blk = blockIdx.y * gridDim.x + blockIdx.x;
  const bool tl = p.stamps && tid_b == 0 && tl_blk < 2048;
  if (tl) p.stamps[64 + 4 * tl_blk] = __builtin_amdgcn_s_memrealtime();

  f32x4v acc[NPT][2];                                // [pixel tile][cout tile]
#pragma unroll
  for (int i = 0; i < NPT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      PIN ? static_cast<void*>(const_cast<unsigned char*>(p.xplanes)) : static_cast<void*>(const_cast<float*>(p.x)), 0,
      (int)((size_t)p.B * p.H * kW * ldx * 4), kBufWord3);
  const __amdgpu_buffer_rsrc_t wp_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(p.wp), 0, 9 * C * N * 4, kBufWord3);
  const __amdgpu_buffer_rsrc_t xs_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.xs + (size_t)b * nchunks * 65536, 0, (p.xs && by == 0) ? nchunks * 65536 : 0, kBufWord3);

  // ---- patch slots: slot = tid + 256 s -> (pixel of the 10 x 34 halo patch, channel quad q = tid & 3).  The slot
  // geometry is recomputed where it is used (from a laundered tid, so that it is not hoisted into 18 live registers).
  struct Slot { unsigned goff; int ldst; unsigned emit; };
  auto slot_of = [&](int t, int s) {
    const int slot = t + s * 256;
    const int q = slot & 3, pix = slot >> 2;
    const int prow = pix / kPW, pcol = pix - prow * kPW;
    const int hh = h0 + prow - 1, ww = pcol - 1;
    // (bitwise, not short-circuit: these must stay selects, a branch would split the loop body)
    const bool inb = slot < P3_ROWS * kPW * 4;
    const bool ok = inb & ((unsigned)hh < (unsigned)p.H) & ((unsigned)ww < (unsigned)kW);
    Slot r;
    // PIN: unit q = (plane, 8-channel half) of the pixel's 64-byte plane record, copied as it is
    const unsigned g = PIN ? (unsigned)(((b * nchunks * p.H + hh) * kW + ww) * 64 + q * 16)
                           : (unsigned)((((b * p.H + hh) * kW + ww) * ldx + q * 4) * 4);
    r.goff = ok ? g : 0x80000000u;
    r.ldst = inb ? (PIN ? (q >> 1) * P3_PLANE + pix * 32 + (q & 1) * 16 : pix * 32 + q * 8) : P3_DUMMY + (t & 63) * 8;
    const bool interior = inb & ((unsigned)(prow - 1) < (unsigned)TR3) & ((unsigned)ww < (unsigned)kW);
    const unsigned e = (unsigned)(((hh * kW + ww) * 2) * 32 + q * 8);
    r.emit = interior ? e : 0xffffffffu;
    return r;
  };
  auto load_slot = [&](const Slot& sl, int cc) {
    if constexpr (GNF != 0) {
      // flat loads from a wave-uniform base (tensor, chunk) + the slot's 32-bit offset: no buffer descriptors (two more of
      // them push uniform values out of the scalar registers: waterfall loops around the weight loads).  Slots outside
      // the image read offset 0 -- any valid address: the fill zeroes them.
      const bool second = cc >= nch1;
      const unsigned long long bb = reinterpret_cast<unsigned long long>(second ? p.x2 : p.x) + (size_t)(second ? cc - nch1 : cc) * (CK * 4);
      typedef const char __attribute__((address_space(1)))* gchar_p;
      typedef const i32x4 __attribute__((address_space(1)))* gvec_p;
      gchar_p base = (gchar_p)(   // (scalar base + 32-bit lane offset: the saddr form of global_load)
          ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bb >> 32)) << 32) |
          (unsigned)__builtin_amdgcn_readfirstlane((int)bb));
      return *(gvec_p)(base + (sl.goff & 0x7fffffffu));
    } else {
      return __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, sl.goff, PIN ? cc * (p.H * kW * 64) : cc * CK * 4, 0));
    }
  };
  float gn_bound = 0.f;
  if constexpr (GNF != 0) {
    const int cpg = C / p.gn_groups;
    float* gslot = reinterpret_cast<float*>(smem + GN_SLOT);
    if (p.xstats) {
      // the bound of gn_fwd_kernel's planes mode, formed here: (sqrt(n) max|gamma| + max|beta|) / keep with keep = 1
      float gm = 0.f, bm = 0.f;
      for (int cc = tid; cc < C; cc += 256) { gm = fmaxf(gm, fabsf(p.gn_gamma[cc])); bm = fmaxf(bm, fabsf(p.gn_beta[cc])); }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { gm = fmaxf(gm, __shfl_xor(gm, o, 64)); bm = fmaxf(bm, __shfl_xor(bm, o, 64)); }
      if (lane == 0) { gslot[4 + wave] = gm; gslot[8 + wave] = bm; }
      __syncthreads();
      gm = fmaxf(fmaxf(gslot[4], gslot[5]), fmaxf(gslot[6], gslot[7]));
      bm = fmaxf(fmaxf(gslot[8], gslot[9]), fmaxf(gslot[10], gslot[11]));
      gn_bound = sqrtf((float)(p.H * kW * cpg)) * gm + bm;
      if (tid == 0) gslot[0] = gn_bound;
      scale_of(__float_as_uint(gn_bound), sx, inv_x);
      if (p.xmax_out && by == 0 && h0 == 0 && tid < kMaxParts) p.xmax_out[b * kMaxParts + tid] = tid ? 0u : __float_as_uint(gn_bound);
    } else {
      gn_bound = __uint_as_float(row_max16(p.xmax, b));
    }
    const int qpg = cpg >> 2, nq1 = ldx >> 2, ntile = p.xstats_tiles > 0 ? p.xstats_tiles : p.H / 8;   // (the PRODUCER's tiles)
    const float inv_n = 1.f / (float)(p.H * kW * cpg);
    typedef float f32x2s __attribute__((ext_vector_type(2)));
    f32x2s* part = reinterpret_cast<f32x2s*>(smem);    // [ntile][C / 4] (sum, sum of squares): the patch buffers are still free
    if (p.xstats) {
      // all partial sums of this image into LDS, every thread's loads in flight together.  (Summed straight from global
      // memory by the C / 4 threads below, the ntile loads of a thread are a dependent chain of L2 round trips, 16 of
      // them behind a launch with 2-row tiles: the sampler's reverse step at 16 images went 7.12 -> 7.01 ms with this;
      // what is left of the prologue is 4-5 us of a 30-33 us block, profiles/r05_gnf_timeline.log.)
      const int nq = C >> 2, total = ntile * nq;
      for (int i0 = tid; i0 < total; i0 += 4 * 256) {
        f32x2s v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * 256;
          if (i < total) {
            const int t = i / nq, qq = i - t * nq;
            const float* src = qq < nq1 ? p.xstats + (((size_t)b * ntile + t) * nq1 + qq) * 2
                                        : p.xstats2 + (((size_t)b * ntile + t) * nq1 + (qq - nq1)) * 2;
            v[u] = *reinterpret_cast<const f32x2s*>(src);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + u * 256 < total) part[i0 + u * 256] = v[u];
      }
      __syncthreads();
    }
    for (int e = tid; e < C / 4; e += 256) {
      const int c = e * 4, g = c / cpg;
      float mean, rstd;
      if (p.xstats) {      // sums of the group's channel quads over the row tiles, in a fixed order
        float s1 = 0.f, s2 = 0.f;
        for (int qq = g * qpg; qq < (g + 1) * qpg; ++qq)
          for (int t = 0; t < ntile; ++t) { const f32x2s v = part[t * (C >> 2) + qq]; s1 += v[0]; s2 += v[1]; }
        mean = s1 * inv_n;
        rstd = rsqrtf(fmaxf(0.f, s2 * inv_n - mean * mean) + p.gn_eps);
        if (p.gn_mean_out && by == 0 && h0 == 0 && (e % qpg) == 0) {
          p.gn_mean_out[b * p.gn_groups + g] = mean;
          p.gn_rstd_out[b * p.gn_groups + g] = rstd;
        }
      } else {
        mean = p.gn_mean[b * p.gn_groups + g];
        rstd = p.gn_rstd[b * p.gn_groups + g];
      }
      const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gn_gamma + c);
      float* te = reinterpret_cast<float*>(smem + SMEM3_B + e * GN_ENT);
      *reinterpret_cast<f32x4*>(te) = f32x4{ga[0] * rstd, ga[1] * rstd, ga[2] * rstd, ga[3] * rstd};
      *reinterpret_cast<f32x4*>(te + 4) = *reinterpret_cast<const f32x4*>(p.gn_beta + c);
      te[8] = mean;
    }
    __syncthreads();
  }
  // (GNF) table entry of channel quad q of chunk cc.  Read where it is used: fetched a few MFMAs ahead (9 more live
  // registers) the main loop spills
  struct GnEntry { f32x4 sc, be; float mean; };
  auto gn_entry = [&](int q, int cc) {
    GnEntry g;
    const float* te = reinterpret_cast<const float*>(smem + SMEM3_B + (cc * 4 + q) * GN_ENT);
    g.sc = *reinterpret_cast<const f32x4*>(te);
    g.be = *reinterpret_cast<const f32x4*>(te + 4);
    g.mean = te[8];
    return g;
  };
  // split one float4 and store it into patch buffer `pbuf` (byte offset in LDS) and into the plane tensor
  auto store_slot = [&](int pbuf, const Slot& sl, i32x4 raw, int cc) {
    if (PIN) {
      *reinterpret_cast<i32x4*>(smem + (sl.ldst >= P3_DUMMY ? sl.ldst & ~15 : pbuf + sl.ldst)) = raw;
      return;
    }
    f32x4 v = __builtin_bit_cast(f32x4, raw);
    f16x4 hi, lo;
    if constexpr (GNF != 0) {
      // two-wide arithmetic as in gn_fwd_kernel (v_pk_add / v_pk_fma / v_pk_mul, v_cvt_pk_f16_f32): half the issue slots
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
      const GnEntry gn = gn_entry((sl.ldst >> 3) & 3, cc);   // ldst = pix * 32 + q * 8 (dummy slots too)
      const f32x4 sc = gn.sc, be = gn.be;
      const float mean = gn.mean;
      const bool ok = sl.goff != 0x80000000u;          // the convolution pads the NORMALISED tensor with zeros
      const float bnd = ok ? gn_bound : 0.f;           // (clamping to +-0 zeroes the slot)
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        f32x2 u = __builtin_elementwise_fma(f32x2{v[e], v[e + 1]} - mean, f32x2{sc[e], sc[e + 1]}, f32x2{be[e], be[e + 1]});
        if (p.gn_act) {
          const f32x2 t = u * -1.4426950408889634f;
          const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.f;
          u = u * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        }
        const f32x2 vs = f32x2{__builtin_amdgcn_fmed3f(u[0], -bnd, bnd), __builtin_amdgcn_fmed3f(u[1], -bnd, bnd)} * sx;
        const f16x2v h = __builtin_convertvector(vs, f16x2v);
        const f16x2v l = __builtin_convertvector(vs - __builtin_convertvector(h, f32x2), f16x2v);
        hi[e] = h[0]; hi[e + 1] = h[1]; lo[e] = l[0]; lo[e + 1] = l[1];
      }
    } else
    if (ABL & 48) {         // timing only (16 / 32): realistic operand values without the split arithmetic
#pragma unroll
      for (int e = 0; e < 4; ++e) { hi[e] = (_Float16)(v[e] * sx); lo[e] = (_Float16)(v[e] * (sx * 0.0004f)); }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 h, l;
        split2(v[e] * sx, h, l);
        hi[e] = h; lo[e] = l;
      }
    }
    unsigned char* d = smem + (sl.ldst >= P3_DUMMY ? sl.ldst : pbuf + sl.ldst);
    *reinterpret_cast<f16x4*>(d) = hi;
    *reinterpret_cast<f16x4*>(d + (sl.ldst >= P3_DUMMY ? 0 : P3_PLANE)) = lo;
    if ((ABL & 16) || GNF == 1) return;   // 16: timing only, no plane stores either (32: the plane stores stay)
    const unsigned eo = sl.emit != 0xffffffffu ? sl.emit + (unsigned)cc * 65536u : 0xffffffffu;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, hi), xs_rsrc, eo, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, lo), xs_rsrc, eo == 0xffffffffu ? eo : eo + 32, 0, 0);
  };

  // (DMA) the five (plane, row) pieces this wave moves per chunk: piece ri = 5 wave + k -> plane ri / 10, patch row ri % 10
  auto dma_piece = [&](int pbuf, int cc, int k) {
    typedef __attribute__((address_space(3))) void* lds_p;
    typedef const __attribute__((address_space(1))) void* gbl_p;
    const int ri = wave * PPW + k;
    const int plane = ri / P3_ROWS, prow = ri - plane * P3_ROWS;
    const int hh = h0 + prow - 1;
    const bool inside = (unsigned)hh < (unsigned)p.H;                 // wave-uniform
    const int hc = inside ? hh : 0;
    const size_t goff = ((((size_t)b * nchunks + cc) * p.H + hc) * kW + (lane >> 1)) * 64 + plane * 32 + (lane & 1) * 16;
    const int ldst = inside ? pbuf + plane * P3_PLANE + (prow * kPW + 1) * 32 : P3_DUMMY;
    __builtin_amdgcn_global_load_lds((gbl_p)(p.xplanes + goff), (lds_p)(smem + ldst), 16, 0, 0);
  };

  // ---- operand addressing
  // pixel fragments (B operand): lane = pixel l15 of the 16-pixel tile, k group grp: channels khalf * 8 .. + 7 of unit sel
  const int xlane = l15 * 32 + khalf * 16;
  // weight fragments (A operand): lane = cout l15 of the 16-cout tile, same k groups
  const unsigned wlane = (unsigned)((n0 + wave * 32 + l15) * 64 + khalf * 16);
  const int unit_stride = N * 64;                    // bytes between (tap, chunk) tiles of the packed weights

  // weight-fragment ring: the fragments of step s + WD - 1 are fetched during step s.  8-row tiles: one step ahead (a step
  // is 1.5 k cycles of MFMA issue).  Short tiles: a step is 380 / 770 cycles, less than an L2 round trip: two steps ahead
  // (a ring of 3 divides the nine steps of a chunk pair: no copy at the pair boundary).  Sampler at 16 images 5.15 ->
  // 4.94 ms per reverse step.  A ring of 9 (8 steps ahead, 144 registers) spills to scratch: 7.7 ms.
  constexpr int WD = TR == 8 ? 2 : WDS;
  f16x8 wf[WD][2][2];                                // [ring slot][cout tile][plane]
  f16x8 xf[3][2];                                    // [ring slot][plane]: fragments are read two pixel tiles ahead
  auto load_w = [&](f16x8 (&w)[2][2], int uA, int uB) {        // uA / uB: tile index tap * nchunks + chunk of lanes < 32 / >= 32
    const int umin = min(uA, uB);                                // (offsets stay non-negative: the range check is unsigned)
    const unsigned vo = wlane + (unsigned)(((sel ? uB : uA) - umin) * unit_stride);
    const int so = umin * unit_stride;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        w[ct][pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wp_rsrc, vo + ct * 1024 + pl * 32, so, 0));
  };
  auto load_w1 = [&](f16x8 (&w)[2][2], int uA, int uB, int i, int sl) {  // fragment i = 2 ct + plane of the same
    const int umin = min(uA, uB);
    const unsigned vo = wlane + (unsigned)(((sl ? uB : uA) - umin) * unit_stride);
    w[i >> 1][i & 1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wp_rsrc, vo + (i >> 1) * 1024 + (i & 1) * 32,
                                                                                      umin * unit_stride, 0));
  };
  auto read_x = [&](f16x8 (&x)[2], int xaddr, int pt) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
      x[pl] = *reinterpret_cast<const f16x8*>(smem + xaddr + ((pt >> 1) * kPW + (pt & 1) * 16) * 32 + pl * P3_PLANE);
  };
  // unit u of a chunk pair (0..17): chunk u / 9 of the pair, tap u % 9
  auto tile_index = [&](int j, int u) { return (u % 9) * nchunks + 2 * j + u / 9; };
  auto x_unit_off = [&](int bufA, int bufB, int u) {
    const int tap = u % 9;
    return (u / 9 ? bufB : bufA) + ((tap / 3) * kPW + tap % 3) * 32;
  };

  // ---- prologue: chunks 0 and 1 into buffers 0 and 1, weights of step 0
  int bufA = 0, bufB = P3_BUF, bufC = 2 * P3_BUF;
  i32x4 stgA[2], stgB[2];
  if constexpr (DMA) {
    // zero the three buffers once (halo columns, rows outside the image), then the first two chunks by DMA
#pragma unroll
    for (int i = 0; i < (3 * P3_BUF + 4095) / 4096; ++i) {
      const int o = (i * 256 + tid) * 16;
      if (o < 3 * P3_BUF) *reinterpret_cast<i32x4*>(smem + o) = i32x4{0, 0, 0, 0};
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPW; ++k) { dma_piece(bufA, 2 * j0, k); dma_piece(bufB, nchunks > 1 ? 2 * j0 + 1 : 0, k); }
  } else {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      i32x4 r[PV3];
#pragma unroll
      for (int s = 0; s < PV3; ++s) r[s] = load_slot(slot_of(tid, s), 2 * j0 + c);
#pragma unroll
      for (int s = 0; s < PV3; ++s) store_slot(c ? bufB : bufA, slot_of(tid, s), r[s], 2 * j0 + c);
    }
  }
#pragma unroll
  for (int d = 0; d < WD - 1; ++d) load_w(wf[d], tile_index(j0, 2 * d), tile_index(j0, 2 * d + 1));    // (WD <= 9: steps of pair j0)
  // patch slots in flight: two register sets (A, B) of two slots each, fetched two steps before they are split and
  // stored (the fetch is an HBM / Infinity-Cache access; one step is ~1.5 k cycles of MFMA issue):
  //   step:   0        1        2        3        4        5        6        7        8
  //   store:  A k0 c0  B k1 c0  A k2 c0  -        -        A k0 c1  B k1 c1  A k2 c1  -
  //   fetch:  A k2 c0  -        -        A k0 c1  B k1 c1  A k2 c1  -        A k0 c0' B k1 c0'
  // (k: slot pair, c0 / c1: the two chunks being filled, c0': the first chunk of the next pair's fill)
  if constexpr (!DMA) {
    const int c2 = min(2 * j0 + 2, nchunks - 1);
    stgA[0] = load_slot(slot_of(tid, 0), c2);
    stgA[1] = load_slot(slot_of(tid, 1), c2);
    stgB[0] = load_slot(slot_of(tid, 2), c2);
    stgB[1] = load_slot(slot_of(tid, 3), c2);
  }
  // (DMA) the pieces of the first two chunks were issued above and other waves read them behind this barrier: nothing but
  // the issuing wave's vmcnt orders a global_load_lds against another wave's ds_read, and the barrier drains lgkmcnt
  // only (the in-loop barriers carry the same wait; ADVICE r05: this one had been left to whatever vmcnt(0) the compiler
  // happened to emit, which the TR = 4 / 2 and KS = 2 instantiations change)
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int xaddr = xlane + (sel ? x_unit_off(bufA, bufB, 1) : x_unit_off(bufA, bufB, 0));   // of the step being multiplied
  read_x(xf[0], xaddr, 0);
  read_x(xf[1], xaddr, 1);

  if (tl) p.stamps[65 + 4 * tl_blk] = __builtin_amdgcn_s_memrealtime();
  const unsigned long long cyc0 = tl ? __builtin_amdgcn_s_memtime() : 0ull;
  for (int j = j0; j < j1; ++j) {
    int t_l = tid;
    asm volatile("" : "+v"(t_l));                    // launder: slot geometry is recomputed inside the loop
    const int cfill0 = min(2 * j + 2, nchunks - 1), cfill1 = min(2 * j + 3, nchunks - 1);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      // next step's operands: step s + 1 of this pair, or step 0 of the next pair (buffers rotated: A' = C, B' = A);
      // past the end: a harmless reload of the last step.  (The lane's unit selector is laundered per step so that the
      // nine selected tap offsets are formed where they are used instead of living in registers across the loop.)
      const bool last = (s == 8);
      const bool over = last && (j + 1 >= j1);
      const int jn = last ? (over ? j : j + 1) : j;
      const int sn = last ? (over ? 8 : 0) : s + 1;
      const int nA = (last && !over) ? bufC : bufA, nB = (last && !over) ? bufA : bufB;
      int sel_s = sel;
      asm volatile("" : "+v"(sel_s));
      const int xaddr_n = xlane + (sel_s ? x_unit_off(nA, nB, 2 * sn + 1) : x_unit_off(nA, nB, 2 * sn));
      // (wave-uniform by construction; said explicitly, because with the GroupNorm-fed fill the compiler otherwise forms
      // them in vector registers and wraps every weight load in a waterfall loop -- branches inside the loop body)
      // the weight fragments fetched in this step: those of step s + WD - 1 (past this group's last step: a harmless
      // reload of the last one)
      const int w_ahead = s + WD - 1;
      const bool w_over = j + w_ahead / 9 >= j1;
      const int wj = w_over ? j1 - 1 : j + w_ahead / 9, ws = w_over ? 8 : w_ahead % 9;
      const int uAn = __builtin_amdgcn_readfirstlane(tile_index(wj, 2 * ws));
      const int uBn = __builtin_amdgcn_readfirstlane(tile_index(wj, 2 * ws + 1));
      // patch traffic of this step (table above): steps 0-2 fill chunk 2 j + 2 into buffer C, steps 5-7 chunk 2 j + 3
      // into buffer A (free after the step-4 barrier)
      const bool fill0 = s < NK, fill1 = s >= 5 && s < 5 + NK;
      const int fk = fill0 ? s : s - 5;              // slot pair stored in this step
      const bool st_useB = (fk == 1);
      int lk = -1, lcc = 0;                          // slot pair fetched in this step and its chunk
      if (s == 0 && NK == 3) { lk = 2; lcc = cfill0; }
      else if (s == 3) { lk = 0; lcc = cfill1; }
      else if (s == 4) { lk = 1; lcc = cfill1; }
      else if (s == 5 && NK == 3) { lk = 2; lcc = cfill1; }
      else if (s == 7) { lk = 0; lcc = min(2 * j + 4, nchunks - 1); }
      else if (s == 8) { lk = 1; lcc = min(2 * j + 4, nchunks - 1); }
      const bool ld_useB = (lk == 1);
      // The MFMAs are volatile asm statements: memory operations keep their place between them (this is the issue
      // order), the address / split arithmetic floats into the shadows.
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        const int cur = (s * NPT + pt) % 3, nxt = (cur + 2) % 3;
        const int xa = pt < NPT - 2 ? xaddr : xaddr_n, pn = pt < NPT - 2 ? pt + 2 : pt - (NPT - 2);
        const int xo = ((pn >> 1) * kPW + (pn & 1) * 16) * 32;
        // small terms first: w_l x_h, w_h x_l, w_h x_h; the two cout tiles alternate so that an MFMA never waits for the
        // accumulator of the one right in front of it
        if (pt == 0) mfma16<1>(acc[pt][0], wf[s % WD][0][1], xf[cur][0], false);
        else mfma16<0>(acc[pt][0], wf[s % WD][0][1], xf[cur][0], false);
        if (!(ABL & 1)) xf[nxt][0] = *reinterpret_cast<const f16x8*>(smem + xa + xo);
        mfma16<0>(acc[pt][1], wf[s % WD][1][1], xf[cur][0], false);
        mfma16<0>(acc[pt][0], wf[s % WD][0][0], xf[cur][1], false);
        if (!(ABL & 2) && pt >= WL0 && pt < WL0 + 4) load_w1(wf[(s + WD - 1) % WD], uAn, uBn, pt - WL0, sel_s);
        if constexpr (DMA) {
          // chunk 2 j + 2 -> buffer C in step 0 (free since the barrier that ended the pair before), chunk 2 j + 3 -> buffer A
          // in step 5 (free since the step-4 barrier): five pieces per wave, one behind every second pixel tile
          // (at the head of the step: the compiler's next wait for the weight fragments -- the in-order vmcnt counter makes it
          // a wait for these pieces too -- comes at the end of the step)
          if (!(ABL & 4) && (s == 0 || s == 5) && pt < PPW)
            dma_piece(s == 0 ? bufC : bufA, s == 0 ? cfill0 : cfill1, pt);
        } else {
        if (!(ABL & 4) && (pt == ST0 || pt == ST1) && (fill0 || fill1)) {
          i32x4& r = st_useB ? stgB[pt == ST1] : stgA[pt == ST1];
          asm volatile("" : "+v"(r));                // pins the split arithmetic here (it would float to the step's top)
          store_slot(fill0 ? bufC : bufA, slot_of(t_l, 2 * fk + (pt == ST1)), r, fill0 ? cfill0 : cfill1);
        }
        if (!(ABL & 4) && (pt == LD0 || pt == LD1) && lk >= 0)
          (ld_useB ? stgB[pt == LD1] : stgA[pt == LD1]) = load_slot(slot_of(t_l, 2 * lk + (pt == LD1)), lcc);
        }
        mfma16<0>(acc[pt][1], wf[s % WD][1][0], xf[cur][1], false);
        if (!(ABL & 1)) xf[nxt][1] = *reinterpret_cast<const f16x8*>(smem + xa + xo + P3_PLANE);
        mfma16<0>(acc[pt][0], wf[s % WD][0][0], xf[cur][0], false);
        if (s == 8 && pt == NPT - 1) mfma16<2>(acc[pt][1], wf[s % WD][1][0], xf[cur][0], false);
        else mfma16<0>(acc[pt][1], wf[s % WD][1][0], xf[cur][0], false);
      }
      xaddr = xaddr_n;
      if (!(ABL & 8) && (s == 4 || s == 8)) {
        // (DMA) the barrier must also cover this wave's LDS-DMA pieces: other waves read them behind it, and nothing but
        // the issuing wave's vmcnt orders a global_load_lds against a later ds_read.  The compiler's barrier waits for
        // lgkmcnt only; the vmcnt(0) it happens to emit for the weight fragments of this step is not a guarantee
        // (ADVICE r04).  Free: the counter is already drained here.
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    const int t = bufA; bufA = bufC; bufC = bufB; bufB = t;      // (A, B, C) <- (C, A, B)
    // nine steps per pair: the weights fetched during step 8 sit in buffer 1, the next pair starts on buffer 0
    // (a ring of 3 or 9 slots divides the nine steps: nothing to move)
    if constexpr (WD == 2) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wf[0][ct][pl] = wf[1][ct][pl];
    }
  }
  if (tl) p.stamps[66 + 4 * tl_blk] = __builtin_amdgcn_s_memrealtime();
  if (tl && tl_blk < 32) p.stamps[tl_blk] = __builtin_amdgcn_s_memtime() - cyc0;   // core cycles spent in the main loop

  // ---- epilogue: straight from the accumulator layout (lane: pixel l15 of the tile, couts 4 grp .. 4 grp + 3)
  if (GNF != 0 && p.xstats) scale_of(__float_as_uint(*reinterpret_cast<const float*>(smem + GN_SLOT)), sx, inv_x);
  else scale_of(row_max16(p.xmax, b), sx, inv_x);    // (re-derived here: nothing of it lives across the main loop)
  scale_of(row_max16(p.wmax, 0), sw, inv_w);
  constexpr int HPT = NPT / 2;
  if constexpr (KS == 2) {
    // the groups swap halves: group 0 keeps pixel tiles [0, HPT) and receives group 1's sums for them, group 1 the rest.
    // Slot (pixel tile, cout tile, thread of the group): the receiver is the same thread of the other group (same wave,
    // same lane -> the same couts and pixels).  The exchange overlays both regions (the bound above is read already).
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, HPT> IH;
    f32x4v* xch = reinterpret_cast<f32x4v*>(smem_blk);
    auto send = [&](auto lo) {
#pragma unroll
      for (int i = 0; i < HPT; ++i)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) xch[((decltype(lo)::value + i) * 2 + ct) * 256 + tid] = acc[decltype(lo)::value + i][ct];
    };
    auto recv = [&](auto lo) {
#pragma unroll
      for (int i = 0; i < HPT; ++i)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4v o = xch[((decltype(lo)::value + i) * 2 + ct) * 256 + tid];
          f32x4v& a = acc[decltype(lo)::value + i][ct];
          a[0] += o[0]; a[1] += o[1]; a[2] += o[2]; a[3] += o[3];
        }
    };
    // (DMA) the exchange overlays the patch buffers: no piece of any wave may still be in flight towards them
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // every wave is through its last LDS read and has its scales
    if (kgrp == 0) send(IH{}); else send(I0{});
    __syncthreads();
    if (kgrp == 0) recv(I0{}); else recv(IH{});
  }
  const float* __restrict__ res = p.res;
  const float* __restrict__ cbp = p.cbias;
  float* __restrict__ yout = p.y;
  const int nb = n0 + wave * 32 + grp * 4;
  f32x4 bias4[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    bias4[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4[ct] = *reinterpret_cast<const f32x4*>(p.bias + nb + ct * 16);
    if (p.cbias_mode == 1) {
      const f32x4 c = *reinterpret_cast<const f32x4*>(cbp + (size_t)b * N + nb + ct * 16);
      bias4[ct][0] += c[0]; bias4[ct][1] += c[1]; bias4[ct][2] += c[2]; bias4[ct][3] += c[3];
    }
  }
  unsigned omax = 0;
  float ys1[2] = {0.f, 0.f}, ys2[2] = {0.f, 0.f};    // (ystats) sums of y and y^2 over this lane's pixels, per cout tile
  // one straight-line body per (residual, per-pixel FiLM bias) combination: the loads of all tiles can be in flight together
  auto finish = [&](auto has_res, auto has_cb2, auto has_ys, auto lo, auto hi) {
#pragma unroll
    for (int pt = decltype(lo)::value; pt < decltype(hi)::value; ++pt) {
      const size_t pixbase = ((((size_t)b * p.H + h0 + (pt >> 1)) * kW) + (pt & 1) * 16 + l15) * N + nb;
      f32x4 add[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) add[ct] = bias4[ct];
      if constexpr (decltype(has_cb2)::value) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 c = *reinterpret_cast<const f32x4*>(cbp + pixbase + ct * 16);
          add[ct][0] += c[0]; add[ct][1] += c[1]; add[ct][2] += c[2]; add[ct][3] += c[3];
        }
      }
      if constexpr (decltype(has_res)::value) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 c = *reinterpret_cast<const f32x4*>(res + pixbase + ct * 16);   // (nt loads here: conv 0.527 vs 0.538)
          add[ct][0] += c[0]; add[ct][1] += c[1]; add[ct][2] += c[2]; add[ct][3] += c[3];
        }
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (acc[pt][ct][e] * inv_x) * inv_w + add[ct][e];
          omax = max(omax, __float_as_uint(o[e]) & 0x7fffffffu);
        }
        if constexpr (decltype(has_ys)::value) {
          ys1[ct] += (o[0] + o[1]) + (o[2] + o[3]);
          ys2[ct] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
        }
        *reinterpret_cast<f32x4*>(yout + pixbase + ct * 16) = o;
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  typedef std::integral_constant<int, 0> P0;
  typedef std::integral_constant<int, HPT> PH;
  typedef std::integral_constant<int, NPT> PN;
  auto finish3 = [&](auto has_ys, auto lo, auto hi) {
    if (res) { if (p.cbias_mode == 2) finish(T{}, T{}, has_ys, lo, hi); else finish(T{}, F{}, has_ys, lo, hi); }
    else { if (p.cbias_mode == 2) finish(F{}, T{}, has_ys, lo, hi); else finish(F{}, F{}, has_ys, lo, hi); }
  };
  auto finish2 = [&](auto has_ys) {
    if constexpr (KS == 2) { if (kgrp == 0) finish3(has_ys, P0{}, PH{}); else finish3(has_ys, PH{}, PN{}); }
    else finish3(has_ys, P0{}, PN{});
  };
  if constexpr (GNF == 1) finish2(T{});
  else { if (p.ystats) finish2(T{}); else finish2(F{}); }
  if (p.ystats) {   // this lane's 4 couts of a tile are one channel quad: sum over the 16 pixel lanes
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { ys1[ct] += __shfl_xor(ys1[ct], o, 64); ys2[ct] += __shfl_xor(ys2[ct], o, 64); }
  }
  if (p.ymax) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o, 64));
  }
  const int part = (h0 / TR3) * gridDim.y + by, nparts = tiles_per_img * gridDim.y;
  if constexpr (KS == 2) {
    // both groups' parts of the two reductions through the tail of the block's LDS (behind the accumulator exchange)
    unsigned* ured = reinterpret_cast<unsigned*>(smem_blk + NPT * 8192);
    float* ysx = reinterpret_cast<float*>(smem_blk + NPT * 8192 + 64);          // [wave][grp][ct][2]
    if (p.ymax && lane == 0) ured[kgrp * 4 + wave] = omax;
    if (p.ystats && kgrp == 1 && l15 == 0) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) { ysx[((wave * 4 + grp) * 2 + ct) * 2] = ys1[ct]; ysx[((wave * 4 + grp) * 2 + ct) * 2 + 1] = ys2[ct]; }
    }
    __syncthreads();
    if (p.ystats && kgrp == 0 && l15 == 0) {          // one writer per quad: (rows of group 0) + (rows of group 1)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int quad = (nb + ct * 16) >> 2;
        float* d = p.ystats + (((size_t)b * tiles_per_img + h0 / TR3) * (N >> 2) + quad) * 2;
        d[0] = ys1[ct] + ysx[((wave * 4 + grp) * 2 + ct) * 2]; d[1] = ys2[ct] + ysx[((wave * 4 + grp) * 2 + ct) * 2 + 1];
      }
    }
    if (p.ymax) {
      if (tid_b == 0) {
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) m = max(m, ured[i]);
        p.ymax[b * 16 + part] = m;
      }
      if (part == 0 && tid_b >= nparts && tid_b < 16) p.ymax[b * 16 + tid_b] = 0u;
    }
  } else {
    if (p.ystats && l15 == 0) {                        // one writer per quad
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int quad = (nb + ct * 16) >> 2;
        float* d = p.ystats + (((size_t)b * tiles_per_img + h0 / TR3) * (N >> 2) + quad) * 2;
        d[0] = ys1[ct]; d[1] = ys2[ct];
      }
    }
    if (p.ymax) {   // this block is partial maximum number (row tile, cout block) of image b; unused entries zeroed
      __syncthreads();
      unsigned* ured = reinterpret_cast<unsigned*>(smem);
      if (lane == 0) ured[wave] = omax;
      __syncthreads();
      if (tid == 0) p.ymax[b * 16 + part] = max(max(ured[0], ured[1]), max(ured[2], ured[3]));
      if (part == 0 && tid >= nparts && tid < 16) p.ymax[b * 16 + tid] = 0u;
    }
  }
  if (tl) p.stamps[67 + 4 * tl_blk] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace

bool mulan_conv3x3_f16x3_v3_eligible(int H, int C, int N) { return H % 8 == 0 && C % 32 == 0 && N % BN == 0; }

// Tile height of a launch: the tallest of 8 / 4 / 2 rows that still gives every CU a block (256 blocks), as long as the
// maxima array (16 partials per image: row tiles x cout blocks) holds.  Measured (profiles/r05_tile_rows_bench.log, the
// train step of BASELINE configs[2] per GPU batch): 64 images -- 8 rows 43.4 ms, 4 rows 44.4 (two half blocks per CU do
// not beat one whole block: the halo and the weight fragments per MFMA cost what the co-residence gains), 32 images --
// 30.3 / 28.6 / 29.3 ms for 8 / 4 / 2 rows, 16 images -- 26.2 / 23.9 / 23.4 ms.  tune[23] = 8 / 4 / 2: dev override.
int mulan_conv3x3_f16x3_v3_tile_rows(int B, int H, int N, bool with_ymax) {
  const int nb = N / BN;
  int tr = 8;
  for (int t : {8, 4, 2}) {
    if (with_ymax && (H / t) * nb > kMaxParts) break;
    tr = t;
    if (B * (H / t) * nb >= 256) break;
  }
  const int forced = g_mulan_tune[23];
  if ((forced == 8 || forced == 4 || forced == 2) && !(with_ymax && (H / forced) * nb > kMaxParts)) tr = forced;
  return tr;
}

// k-split of a launch (see the kernel): 2 for launches of at most 256 blocks whose caller says that no other stream's
// kernels share the chip (`alone`: a k-split block takes 130 KB of a CU's LDS, the weight-gradient blocks of the train
// step's second stream would queue behind it), with an even number of chunk pairs.  tune[27]: 1 = never, 2 = whenever the
// shape allows (dev / tests).
int mulan_conv3x3_f16x3_v3_ksplit(int B, int H, int C, int N, bool with_ymax, int alone) {
  const int tr = mulan_conv3x3_f16x3_v3_tile_rows(B, H, N, with_ymax);
  const bool shape_ok = (C / 32) % 2 == 0 && C >= 64;
  if (!shape_ok || g_mulan_tune[27] == 1) return 1;
  if (g_mulan_tune[27] == 2) return 2;
  // 8-row tiles stay whole: measured at 64 images (profiles/r05_ksplit.log) the exchange of 128 accumulator registers
  // and the two patch prologues cost more (epilogue 12 -> 19 us, prologue 5.6 -> 8.6) than the main loop gains (46 -> 45)
  return (alone && tr < 8 && B * (H / tr) * (N / BN) <= 256) ? 2 : 1;
}

namespace {
template <int ABL, bool PIN, int GNF, bool DMA, int TR, int KS = 1>
int launch_v3(const ConvArgsH& a, hipStream_t stream) {
  constexpr int smem = KS * (GNF ? V3Geo<TR>::SMEM_GN : V3Geo<TR>::SMEM);
  static bool configured = false;                   // (one flag per instantiation)
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_f16x3_v3_kernel<ABL, PIN, GNF, DMA, TR, KS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  const dim3 grid(a.B * (a.H / TR), a.N / BN);
  hipLaunchKernelGGL((conv3x3_f16x3_v3_kernel<ABL, PIN, GNF, DMA, TR, KS>), grid, dim3(256 * KS), smem, stream, a);
  return (int)hipGetLastError();
}
template <bool PIN, int GNF, bool DMA>
int launch_v3_rows(const ConvArgsH& a, int tr, int ks, hipStream_t stream) {
  if (ks == 2) {
    if (tr == 4) return launch_v3<0, PIN, GNF, DMA, 4, 2>(a, stream);
    if (tr == 2) return launch_v3<0, PIN, GNF, DMA, 2, 2>(a, stream);
    return launch_v3<0, PIN, GNF, DMA, 8, 2>(a, stream);
  }
  if (tr == 4) return launch_v3<0, PIN, GNF, DMA, 4>(a, stream);
  if (tr == 2) return launch_v3<0, PIN, GNF, DMA, 2>(a, stream);
  return launch_v3<0, PIN, GNF, DMA, 8>(a, stream);
}
}  // namespace

int mulan_launch_conv3x3_f16x3_v3(const f16x3::ConvArgsH& a_in, hipStream_t stream) {
  f16x3::ConvArgsH a = a_in;
  const int tr = mulan_conv3x3_f16x3_v3_tile_rows(a.B, a.H, a.N, a.ymax != nullptr);
  const unsigned gx = (unsigned)(a.B * (a.H / tr)), gy = (unsigned)(a.N / BN);
  a.pair_cols = (gy > 1 && gx % 8 == 0 && g_mulan_tune[13] != 1) ? 1 : 0;    // tune[13] = 1: dev A/B, plain 2-D order
  const int ks = mulan_conv3x3_f16x3_v3_ksplit(a.B, a.H, a.C, a.N, a.ymax != nullptr, a.alone);
  if (a.gn_mean) {   // GroupNorm-fed forward convolution: with (xs) / without the planes as a by-product
    if (a.C > GN_MAXC) return (int)hipErrorInvalidValue;
    return a.xs ? launch_v3_rows<false, 2, false>(a, tr, ks, stream) : launch_v3_rows<false, 1, false>(a, tr, ks, stream);
  }
  if (a.xplanes) {   // plane-fed forward convolution
    if (g_mulan_tune[18] == 1)     // dev A/B: the register-staged patch fill of rounds 2-3
      return launch_v3_rows<true, 0, false>(a, tr, 1, stream);
    // LDS-DMA patch fill (round 4: 76.72 / 76.79 / 76.71 vs 77.02 / 76.75 / 76.93 ms per step)
    return launch_v3_rows<true, 0, true>(a, tr, ks, stream);
  }
  switch (tr == 8 ? g_mulan_tune[4] : 0) {   // dev-only ablations (tools/conv_ab.py --ablate), 8-row tiles
    case 1: return launch_v3<1, false, 0, false, 8>(a, stream);
    case 2: return launch_v3<2, false, 0, false, 8>(a, stream);
    case 4: return launch_v3<4, false, 0, false, 8>(a, stream);
    case 8: return launch_v3<8, false, 0, false, 8>(a, stream);
    case 15: return launch_v3<15, false, 0, false, 8>(a, stream);
    case 16: return launch_v3<16, false, 0, false, 8>(a, stream);
    case 32: return launch_v3<32, false, 0, false, 8>(a, stream);
    default: break;
  }
  return launch_v3_rows<false, 0, false>(a, tr, ks, stream);
}
