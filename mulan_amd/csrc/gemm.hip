// Batched fp32 GEMM on v_mfma_f32_32x32x2_f32 for every Dense / 1x1 / attention contraction of the
// path: nin_shortcut (ldm/model_vdm.py:652-653), attention q,k,v,proj_out and QK^T / PV
// (model_vdm.py:676-685,775-796), cond MLP dense0/dense1 + cond_proj (model_vdm.py:337-338,639-641;
// ldm/ldm_unet.py:38-45,89-90), the gamma MLP (ldm/model_mulan_epsilon.py:531-538) and the
// encoder head (model_mulan_epsilon.py:153-154), plus their autodiff transposes.
//
//   C[b] = alpha * op(A[b]) @ op(B[b]) + bias[n] + beta * R[b]
//
// op(A) is [M,K]: TA=0 -> A stored [M][lda] (k contiguous), TA=1 -> stored [K][lda] (m contiguous).
// op(B) is [K,N]: TB=0 -> B stored [K][ldb] (n contiguous), TB=1 -> stored [N][ldb] (k contiguous).
// k-contiguous operands are staged [row][k] (stride KC+4) and read with ds_read_b128 under the
// k permutation k = 8c + 4*(lane>>5) + j; k-major operands are staged [k][row] and read with
// ds_read_b32 under the same permutation, so any TA/TB combination shares one MFMA loop.
// Long-K / few-tile shapes (weight gradients, K = B*1024) take a split-K path: raw partial slabs +
// a fixed-order reduce (deterministic, no float atomics).
#include "common.h"

namespace {

constexpr int KC = 16;
constexpr int KS = KC + 4;   // row stride for k-contiguous staging ((KS/4) odd)

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias; const float* R;
  int M, N, K, lda, ldb, ldc, ldr;
  long long sA, sB, sC, sR;   // batch strides (elements)
  float alpha, beta;
  int ksplit, kchunk;         // split-K (batch == 1): blockIdx.z owns k in [z*kchunk, (z+1)*kchunk)
  float* ws;                  // [ksplit][M][N] raw partial products
};

// VEC: both operands have 16-byte aligned rows along their contiguous dimension.  VEC loaders are branch
// free (clamped address, validity mask applied at the LDS store) so the prefetch never forces a vmcnt(0)
// in front of the MFMA cluster.
template <int BM, int BN, int WM, int WN, int TA, int TB, int VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  constexpr int MT = BM / 32 / WM, NT = BN / 32 / WN;
  constexpr int A_F = TA ? KC * BM : BM * KS;
  constexpr int B_F = TB ? BN * KS : KC * BN;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_F + B_F)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, bz = blockIdx.z;
  const bool split = p.ksplit > 1;
  const float* A = p.A + (split ? 0 : (size_t)bz * p.sA);
  const float* B = p.B + (split ? 0 : (size_t)bz * p.sB);
  const int M = p.M, N = p.N;
  const int kbeg = split ? bz * p.kchunk : 0;
  const int K = split ? min(p.K, kbeg + p.kchunk) : p.K;   // exclusive upper bound of this block's k range

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int AV = (BM * KC / 4 + 255) / 256;
  constexpr int BV = (BN * KC / 4 + 255) / 256;
  f32x4 areg[AV], breg[BV];
  unsigned amask = 0, bmask = 0;

  // chunk-invariant part of the prefetch addressing (VEC path): per-slot pointer at k = 0, row validity and the
  // slot's k offset inside a chunk, so each chunk's prefetch is a pointer bump + one compare per slot
  const float* aptr[AV]; const float* bptr[BV];
  int ako[AV], bko[BV];
  unsigned arow = 0, brow = 0;
  auto setup = [&](const float** ptr, int* ko, unsigned& rowok, const float* base, int ld, int row0, int rows_total,
                   bool kcontig, int TILE, int nslots) {
#pragma unroll
    for (int s = 0; s < nslots; ++s) {
      const int slot = tid + s * 256;
      int row, k;
      if (kcontig) {
        const int r = slot / (KC / 4), kq = slot - r * (KC / 4);
        row = row0 + r; k = kq * 4;
      } else {
        const int kk = slot / (TILE / 4), rq = slot - kk * (TILE / 4);
        row = row0 + rq * 4; k = kk;
      }
      const bool ok = slot < TILE * KC / 4 && row < rows_total;
      ptr[s] = ok ? base + (kcontig ? (size_t)row * ld + k : (size_t)k * ld + row) : base;
      ko[s] = k;
      rowok |= (ok ? 1u : 0u) << s;
    }
  };
  if (VEC) {
    setup(aptr, ako, arow, A, p.lda, m0, M, !TA, BM, AV);
    setup(bptr, bko, brow, B, p.ldb, n0, N, TB, BN, BV);
  }
  auto gload_fast = [&](f32x4* reg, unsigned& mask, const float* const* ptr, const int* ko, unsigned rowok,
                        const float* base, int ld, int k0, bool kcontig, int nslots) {
    mask = 0;
    const size_t step = kcontig ? (size_t)k0 : (size_t)k0 * ld;   // wave uniform
#pragma unroll
    for (int s = 0; s < nslots; ++s) {
      const bool ok = ((rowok >> s) & 1u) && (k0 + ko[s] < K);
      reg[s] = *reinterpret_cast<const f32x4*>(ok ? ptr[s] + step : base);
      mask |= (ok ? 1u : 0u) << s;
    }
  };

  // `TILE` rows x KC tile of a matrix stored either [row][k] (kcontig) or [k][row].
  auto gload = [&](f32x4* reg, unsigned& mask, const float* base, int ld, int row0, int rows_total, int k0,
                   bool kcontig, int TILE, int nslots) {
    mask = 0;
#pragma unroll
    for (int s = 0; s < nslots; ++s) {
      const int slot = tid + s * 256;
      int row, k;
      if (kcontig) {
        const int r = slot / (KC / 4), kq = slot - r * (KC / 4);
        row = row0 + r; k = k0 + kq * 4;
      } else {
        const int kk = slot / (TILE / 4), rq = slot - kk * (TILE / 4);
        row = row0 + rq * 4; k = k0 + kk;
      }
      const bool ok = slot < TILE * KC / 4 && row < rows_total && k < K;
      const size_t off = kcontig ? (size_t)row * ld + k : (size_t)k * ld + row;
      if (VEC) {
        reg[s] = *reinterpret_cast<const f32x4*>(ok ? base + off : base);
        mask |= (ok ? 1u : 0u) << s;
      } else {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
          const float* src = base + off;
          const int lim = kcontig ? K - k : rows_total - row;    // valid elements along the contiguous dim
          v[0] = src[0];
          if (lim > 1) v[1] = src[1];
          if (lim > 2) v[2] = src[2];
          if (lim > 3) v[3] = src[3];
        }
        reg[s] = v;
        mask |= 1u << s;
      }
    }
  };
  auto lstore = [&](const f32x4* reg, unsigned mask, float* dst, bool kcontig, int TILE, int nslots) {
#pragma unroll
    for (int s = 0; s < nslots; ++s) {
      const int slot = tid + s * 256;
      if (slot < TILE * KC / 4) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 v = ((mask >> s) & 1u) ? reg[s] : z;
        if (kcontig) {
          const int r = slot / (KC / 4), kq = slot - r * (KC / 4);
          *reinterpret_cast<f32x4*>(dst + r * KS + kq * 4) = v;
        } else {
          *reinterpret_cast<f32x4*>(dst + slot * 4) = v;   // [k][TILE]
        }
      }
    }
  };

  const int nk = (K - kbeg + KC - 1) / KC;
  float* As0 = smem;
  float* Bs0 = smem + 2 * A_F;
  gload(areg, amask, A, p.lda, m0, M, kbeg, !TA, BM, AV);
  gload(breg, bmask, B, p.ldb, n0, N, kbeg, TB, BN, BV);
  lstore(areg, amask, As0, !TA, BM, AV);
  lstore(breg, bmask, Bs0, TB, BN, BV);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const float* As = As0 + (kt & 1) * A_F;
    const float* Bs = Bs0 + (kt & 1) * B_F;
    const bool has_next = kt + 1 < nk;
    if (has_next) {
      if (VEC) {
        gload_fast(areg, amask, aptr, ako, arow, A, p.lda, kbeg + (kt + 1) * KC, !TA, AV);
        gload_fast(breg, bmask, bptr, bko, brow, B, p.ldb, kbeg + (kt + 1) * KC, TB, BV);
      } else {
        gload(areg, amask, A, p.lda, m0, M, kbeg + (kt + 1) * KC, !TA, BM, AV);
        gload(breg, bmask, B, p.ldb, n0, N, kbeg + (kt + 1) * KC, TB, BN, BV);
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // prefetch loads stay above the MFMA cluster
    {
      // b128 fragments of the whole chunk up front; b32 fragments double-buffered one k-step ahead
      f32x4 a4[KC / 8][MT], b4[KC / 8][NT];
#pragma unroll
      for (int k8 = 0; k8 < KC / 8; ++k8) {
        if (!TA) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            a4[k8][mt] = *reinterpret_cast<const f32x4*>(As + ((wm * MT + mt) * 32 + li) * KS + k8 * 8 + 4 * lh);
        }
        if (TB) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            b4[k8][nt] = *reinterpret_cast<const f32x4*>(Bs + ((wn * NT + nt) * 32 + li) * KS + k8 * 8 + 4 * lh);
        }
      }
      float acur[MT], anext[MT], bcur[NT], bnext[NT];
      auto frag = [&](int st, float* av, float* bv) {
        const int k8 = st >> 2, j = st & 3;
        const int kk = k8 * 8 + 4 * lh + j;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = TA ? As[kk * BM + (wm * MT + mt) * 32 + li] : a4[k8][mt][j];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = TB ? b4[k8][nt][j] : Bs[kk * BN + (wn * NT + nt) * 32 + li];
      };
      frag(0, acur, bcur);
#pragma unroll
      for (int st = 0; st < KC / 2; ++st) {
        if (st + 1 < KC / 2) frag(st + 1, anext, bnext);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma32(acur[mt], bcur[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acur[mt] = anext[mt];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bcur[nt] = bnext[nt];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) {
      lstore(areg, amask, As0 + ((kt + 1) & 1) * A_F, !TA, BM, AV);
      lstore(breg, bmask, Bs0 + ((kt + 1) & 1) * B_F, TB, BN, BV);
    }
    __syncthreads();
  }

  if (split) {   // raw partial products; bias / alpha / residual are applied by splitk_reduce_kernel
    float* W = p.ws + (size_t)bz * M * N;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + (wn * NT + nt) * 32 + li;
      if (n >= N) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * MT + mt) * 32 + mfma32_row(r, lane);
          if (m < M) W[(size_t)m * N + n] = acc[mt][nt][r];
        }
    }
    return;
  }
  // residual loads of a whole tile are issued before its stores (see conv3x3_fwd_kernel's epilogue)
  float* __restrict__ Cb = p.C + (size_t)bz * p.sC;
  const float* __restrict__ Rb = p.R ? p.R + (size_t)bz * p.sR : nullptr;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + (wn * NT + nt) * 32 + li;
    if (n >= N) continue;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int mb = m0 + (wm * MT + mt) * 32;
      float add[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) add[r] = bv;
      if (Rb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + mfma32_row(r, lane);
          if (m < M) add[r] += p.beta * Rb[(size_t)m * p.ldr + n];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + mfma32_row(r, lane);
        if (m < M) Cb[(size_t)m * p.ldc + n] = p.alpha * acc[mt][nt][r] + add[r];
      }
    }
  }
}

// C = alpha * sum_s ws[s] + bias + beta * R   (fixed order => bitwise reproducible)
__global__ void splitk_reduce_kernel(GemmArgs p) {
  const size_t E = (size_t)p.M * p.N;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    int i = 0;
    for (; i + 8 <= p.ksplit; i += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p.ws[(size_t)(i + u) * E + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < p.ksplit; ++i) s += p.ws[(size_t)i * E + e];
    const int m = (int)(e / p.N), n = (int)(e - (size_t)m * p.N);
    float v = p.alpha * s + (p.bias ? p.bias[n] : 0.f);
    if (p.R) v += p.beta * p.R[(size_t)m * p.ldr + n];
    p.C[(size_t)m * p.ldc + n] = v;
  }
}

// split-K plan: only when the output has too few tiles to fill the chip and K is long
// long-K products whose output is a whole number of 128 x 128 tiles use those (each operand element is fetched once
// per tile row / column instead of twice with 64 x 64 tiles): the dense weight gradients [128 x 131072] x [131072 x 128]
bool ksplit_big_tile(int M, int N, int K) { return K >= 4096 && M % 128 == 0 && N % 128 == 0; }

int plan_ksplit(int M, int N, int K, int batch, int* kchunk) {
  *kchunk = K;
  if (batch != 1) return 1;
  const bool big = ksplit_big_tile(M, N, K);
  const long long tiles = big ? (long long)(M / 128) * (N / 128) : (long long)((M + 63) / 64) * ((N + 63) / 64);
  int s;
  if (K >= 4096) {
    if (tiles >= 256) return 1;
    s = (int)((big ? 256 : 512) / tiles);
    if (s > K / 512) s = K / 512;
  } else {
    // a handful of tiles with a few hundred k steps each (the [B,512] x [512,128] FiLM projections of every
    // ResnetBlock) would run as one latency-bound wavefront per tile: cut K into 64-deep pieces
    if (K < 256) return 1;
    if (tiles > 8) {
      // the M = batch GEMMs of the gamma MLP ([128, 3072] x [3072, 3072]: 96 tiles of 64 x 64, each streaming a
      // 786 KB weight panel): too few workgroups to pull the 37.7 MB of weights at HBM speed; cut K to fill the CUs
      if (tiles >= 128 || K < 1024) return 1;
      s = (int)((768 + tiles - 1) / tiles);      // measured on [128, 3072] x [3072, 3072]: 98 us unsplit, 53 us at
      if (s > K / 256) s = K / 256;              // 3 splits, 36.6 us at 8, 37.5 us at 16
    } else {
      s = K / 64;
    }
  }
  if (s < 2) return 1;
  int kc = ((K + s - 1) / s + 15) / 16 * 16;
  *kchunk = kc;
  return (K + kc - 1) / kc;
}

template <int BM, int BN, int WM, int WN>
void launch(const GemmArgs& a, int ta, int tb, int vec, int batch, hipStream_t st) {
  dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, batch), blk(256);
#define L(TA, TB, V) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, TA, TB, V>), grid, blk, 0, st, a)
  if (vec) {
    if (!ta && !tb) L(0, 0, 1); else if (!ta && tb) L(0, 1, 1); else if (ta && !tb) L(1, 0, 1); else L(1, 1, 1);
  } else {
    if (!ta && !tb) L(0, 0, 0); else if (!ta && tb) L(0, 1, 0); else if (ta && !tb) L(1, 0, 0); else L(1, 1, 0);
  }
#undef L
}

}  // namespace

MULAN_API size_t mulan_gemm_workspace(int M, int N, int K, int batch) {
  int kc;
  const int s = plan_ksplit(M, N, K, batch, &kc);
  return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

MULAN_API int mulan_gemm(const float* A, const float* B, float* C, const float* bias, const float* R, int M, int N,
                         int K, int lda, int ldb, int ldc, int ldr, int transA, int transB, int batch,
                         long long strideA, long long strideB, long long strideC, long long strideR, float alpha,
                         float beta, float* workspace, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return (int)hipErrorInvalidValue;
  GemmArgs a{A, B, C, bias, R, M, N, K, lda, ldb, ldc, ldr, strideA, strideB, strideC, strideR, alpha, beta,
             1, K, nullptr};
  // float4 global loads need 16-byte aligned rows along the contiguous dimension of both operands.
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int contigA = transA ? M : K, contigB = transB ? K : N;
  const int vec = al(A) && al(B) && (lda % 4 == 0) && (ldb % 4 == 0) && (contigA % 4 == 0) && (contigB % 4 == 0) &&
                  (strideA % 4 == 0) && (strideB % 4 == 0);
  if (workspace) {
    int kc;
    const int s = plan_ksplit(M, N, K, batch, &kc);
    if (s > 1) {
      a.ksplit = s; a.kchunk = kc; a.ws = workspace;
      if (ksplit_big_tile(M, N, K)) launch<128, 128, 2, 2>(a, transA, transB, vec, s, stream);
      else launch<64, 64, 2, 2>(a, transA, transB, vec, s, stream);
      const size_t E = (size_t)M * N;
      const int blocks = (int)((E + 255) / 256 > 2048 ? 2048 : (E + 255) / 256);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
      MULAN_CHECK_LAUNCH();
    }
  }
  const long long tiles128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * batch;
  if (M >= 128 && N >= 128 && tiles128 >= 256) {
    launch<128, 128, 2, 2>(a, transA, transB, vec, batch, stream);
  } else if (N <= 32) {
    launch<128, 32, 4, 1>(a, transA, transB, vec, batch, stream);
  } else {
    launch<64, 64, 2, 2>(a, transA, transB, vec, batch, stream);
  }
  MULAN_CHECK_LAUNCH();
}
