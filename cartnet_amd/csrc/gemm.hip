// fp32 MFMA GEMM for the dense per-edge / per-node Linears of CartNet and their gradients (gfx950).
//
// Tile: 128 x BN (BN = 64/128/256) outputs per 256-thread workgroup, K-step 16; 4 wavefronts in a 2x2 grid,
// each owning (64 x BN/2) = 2 x (BN/64) MFMA tiles of 32x32 (v_mfma_f32_32x32x2_f32, 16 accumulator VGPRs per
// tile).  Operands are staged global -> registers -> LDS (so SiLU can be applied in flight) with the next
// K-step's global loads issued before the current step's MFMAs; 2 workgroups per CU overlap each other's
// barriers.  The k index inside an 8-deep group is permuted (lane half h takes k = 4h..4h+3) so a k-contiguous
// operand is fetched from LDS with one ds_read_b128 per four MFMAs; fp32 MFMA is an exact fmaf chain, the
// permutation only changes the (already arbitrary) summation order.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 16;
constexpr int KPAD = BK + 4;  // LDS row stride (floats) of a k-contiguous tile: 5 x 16 B slots -> conflict-free b128

struct GemmFlags {
  int vecA, vecB, kchunk;
};

template <int ROWS, bool KS>
struct Stager {
  static constexpr int NU = ROWS / 64;  // float4 units per thread per K-step
  f32x4 reg[NU];

  // Global -> registers.  rows_limit bounds the row (m or n) index, kend the k index.
  __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int row0, int rows_limit, int k0,
                                       int kend, bool vec, int tid) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int unit = tid + u * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (!KS) {
        const int row = unit >> 2, kq = unit & 3;
        const int grow = row0 + row, gk = k0 + kq * 4;
        if (grow < rows_limit && gk < kend) {
          const float* ptr = base + (size_t)grow * ld + gk;
          if (vec && gk + 3 < kend) {
            v = *reinterpret_cast<const f32x4*>(ptr);
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (gk + c < kend) v[c] = ptr[c];
          }
        }
      } else {
        const int k = unit / (ROWS / 4), r4 = unit % (ROWS / 4);
        const int gk = k0 + k, grow = row0 + r4 * 4;
        if (gk < kend && grow < rows_limit) {
          const float* ptr = base + (size_t)gk * ld + grow;
          if (vec && grow + 3 < rows_limit) {
            v = *reinterpret_cast<const f32x4*>(ptr);
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (grow + c < rows_limit) v[c] = ptr[c];
          }
        }
      }
      reg[u] = v;
    }
  }

  // Registers -> LDS (optionally through SiLU).
  __device__ __forceinline__ void store(float* __restrict__ lds, bool act, int tid) const {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int unit = tid + u * 256;
      f32x4 v = reg[u];
      if (act) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = cn_silu(v[c]);
      }
      if (!KS) {
        const int row = unit >> 2, kq = unit & 3;
        *reinterpret_cast<f32x4*>(lds + row * KPAD + kq * 4) = v;
      } else {
        const int k = unit / (ROWS / 4), r4 = unit % (ROWS / 4);
        *reinterpret_cast<f32x4*>(lds + k * ROWS + r4 * 4) = v;
      }
    }
  }
};

template <bool A_KS, bool B_KS, int BN>
__global__ __launch_bounds__(256, 2) void cn_gemm_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  constexpr int TM = 2;
  constexpr int TN = BN / 64;
  constexpr int WN = BN / 2;  // columns per wave
  __shared__ __attribute__((aligned(16))) float sA[A_KS ? BK * BM : BM * KPAD];
  __shared__ __attribute__((aligned(16))) float sB[B_KS ? BK * BN : BN * KPAD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;
  const int g = blockIdx.z;
  const int split = blockIdx.y;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int kbeg = split * fl.kchunk;
  const int kend = min(p.K, kbeg + fl.kchunk);
  const bool a_act = p.a_act != 0, b_act = p.b_act != 0;

  Stager<BM, A_KS> stA;
  Stager<BN, B_KS> stB;

  for (int s = 0; s < p.nsegs; ++s) {
    const int idx = (p.ngroups > 1) ? g : s;
    const float* __restrict__ Ab = p.A[idx];
    const float* __restrict__ Bb = p.B[idx];
    if (kbeg >= kend) break;
    stA.load(Ab, p.lda, row0, p.M, kbeg, kend, fl.vecA != 0, tid);
    stB.load(Bb, p.ldb, col0, p.N, kbeg, kend, fl.vecB != 0, tid);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      stA.store(sA, a_act, tid);
      stB.store(sB, b_act, tid);
      __syncthreads();
      if (k0 + BK < kend) {
        stA.load(Ab, p.lda, row0, p.M, k0 + BK, kend, fl.vecA != 0, tid);
        stB.load(Bb, p.ldb, col0, p.N, k0 + BK, kend, fl.vecB != 0, tid);
      }
#pragma unroll
      for (int kg = 0; kg < BK / 8; ++kg) {
        float af[TM][4], bf[TN][4];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          if (!A_KS) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&sA[(wm * 64 + a * 32 + li) * KPAD + kg * 8 + lh * 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) af[a][j] = v[j];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) af[a][j] = sA[(kg * 8 + lh * 4 + j) * BM + wm * 64 + a * 32 + li];
          }
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          if (!B_KS) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&sB[(wn * WN + b * 32 + li) * KPAD + kg * 8 + lh * 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[b][j] = v[j];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[b][j] = sB[(kg * 8 + lh * 4 + j) * BN + wn * WN + b * 32 + li];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][j], bf[b][j], acc[a][b], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  // ---------------------------------------------------------------- epilogue
  float* C = p.C[g];  // may alias resid / dact (in-place use), so no __restrict__ on the epilogue pointers
  if (p.splitk > 1) {
    C += (size_t)split * p.M * p.ldc;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int grow = row0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (grow >= p.M) continue;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int gcol = col0 + wn * WN + b * 32 + li;
          if (gcol < p.N) C[(size_t)grow * p.ldc + gcol] = acc[a][b][r];
        }
      }
    return;
  }

  const float* __restrict__ bias = p.bias[g];
  const float* __restrict__ gi = p.gather_i[g];
  const float* __restrict__ gj = p.gather_j[g];
  const float* resid = p.resid[g];
  const float* dact = p.dact[g];
  float* cpre = p.cpre[g];
  double* __restrict__ colsum = p.colsum[g];
  double* __restrict__ colsq = p.colsq[g];
  const bool out_act = p.out_act != 0;

  float biasv[TN];
  double cs[TN], cq[TN];   // fp64: BatchNorm variance = E[v^2] - mean^2 must not lose digits to cancellation
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int gcol = col0 + wn * WN + b * 32 + li;
    biasv[b] = (bias && gcol < p.N) ? bias[gcol] : 0.f;
    cs[b] = 0.0;
    cq[b] = 0.0;
  }

#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int grow = row0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (grow >= p.M) continue;
      int ti = 0, sj = 0;
      if (gi) {
        ti = p.tgt[grow];
        sj = p.src[grow];
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int gcol = col0 + wn * WN + b * 32 + li;
        if (gcol >= p.N) continue;
        float v = acc[a][b][r] + biasv[b];
        if (gi) v += gi[(size_t)ti * p.ldg + gcol] + gj[(size_t)sj * p.ldg + gcol];
        if (resid) v += resid[(size_t)grow * p.ldr + gcol];
        if (dact) v *= cn_dsilu(dact[(size_t)grow * p.ldd + gcol]);
        if (colsum) {
          cs[b] += (double)v;
          cq[b] += (double)v * (double)v;
        }
        if (cpre) cpre[(size_t)grow * p.ldc + gcol] = v;
        if (out_act) v = cn_silu(v);
        C[(size_t)grow * p.ldc + gcol] = v;
      }
    }

  if (colsum) {
    // rows of this block -> one partial per column: lane halves, then the two waves stacked in M.
    double* red = reinterpret_cast<double*>(sA);  // [2 (sum,sq)][2 (wm)][BN] doubles <= sizeof(sA); K loop ended with a barrier
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      cs[b] += __shfl_xor(cs[b], 32);
      cq[b] += __shfl_xor(cq[b], 32);
      if (lh == 0) {
        red[(0 * 2 + wm) * BN + wn * WN + b * 32 + li] = cs[b];
        red[(1 * 2 + wm) * BN + wn * WN + b * 32 + li] = cq[b];
      }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
      const int gcol = col0 + c;
      if (gcol < p.N) {
        colsum[(size_t)tile_m * p.N + gcol] = red[(0 * 2 + 0) * BN + c] + red[(0 * 2 + 1) * BN + c];
        if (colsq) colsq[(size_t)tile_m * p.N + gcol] = red[(1 * 2 + 0) * BN + c] + red[(1 * 2 + 1) * BN + c];
      }
    }
  }
}

__global__ void cn_splitk_reduce_kernel(const float* __restrict__ slabs, int splitk, int M, int N,
                                        float* __restrict__ out, int ldo) {
  const size_t total = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int s = 0; s < splitk; ++s) acc += slabs[(size_t)s * total + i];
    const size_t m = i / N, n = i % N;
    out[m * ldo + n] = acc;
  }
}

__global__ __launch_bounds__(1024) void cn_colsum_finalize_kernel(const double* __restrict__ parts, int nparts, int N,
                                                                  float* __restrict__ out) {
  __shared__ double red[16 * 64];
  const double tot = cn_block_colsum(parts, nparts, N, blockIdx.x * 64, red);
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x < 64 && c < N) out[c] = (float)tot;
}

__global__ void cn_colsum_finalize_f32_kernel(const float* __restrict__ parts, int nparts, int N,
                                              float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double acc = 0.0;
  for (int q = 0; q < nparts; ++q) acc += (double)parts[(size_t)q * N + n];
  out[n] = (float)acc;
}

template <bool A_KS, bool B_KS>
void launch_gemm(const CartnetGemmArgs& a, const GemmFlags& fl, hipStream_t st) {
  const int tiles_m = cn_ceil_div(a.M, BM);
  if (a.N > 128) {
    dim3 grid(tiles_m * cn_ceil_div(a.N, 256), a.splitk, a.ngroups);
    hipLaunchKernelGGL((cn_gemm_kernel<A_KS, B_KS, 256>), grid, dim3(256), 0, st, a, fl);
  } else if (a.N > 64) {
    dim3 grid(tiles_m * cn_ceil_div(a.N, 128), a.splitk, a.ngroups);
    hipLaunchKernelGGL((cn_gemm_kernel<A_KS, B_KS, 128>), grid, dim3(256), 0, st, a, fl);
  } else {
    dim3 grid(tiles_m * cn_ceil_div(a.N, 64), a.splitk, a.ngroups);
    hipLaunchKernelGGL((cn_gemm_kernel<A_KS, B_KS, 64>), grid, dim3(256), 0, st, a, fl);
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int cartnet_gemm(const CartnetGemmArgs* args, void* stream) {
  CN_CHECK(args != nullptr, "cartnet_gemm: null args");
  CartnetGemmArgs a = *args;
  CN_CHECK(a.M >= 0 && a.N >= 0 && a.K >= 0, "cartnet_gemm: negative shape M=%d N=%d K=%d", a.M, a.N, a.K);
  CN_CHECK(a.ngroups >= 1 && a.ngroups <= CARTNET_MAX_GROUPS, "cartnet_gemm: ngroups=%d out of range", a.ngroups);
  CN_CHECK(a.nsegs >= 1 && a.nsegs <= CARTNET_MAX_GROUPS, "cartnet_gemm: nsegs=%d out of range", a.nsegs);
  CN_CHECK(!(a.ngroups > 1 && a.nsegs > 1), "cartnet_gemm: groups and K-segments are mutually exclusive");
  CN_CHECK(a.splitk >= 1, "cartnet_gemm: splitk=%d", a.splitk);
  if (a.M == 0 || a.N == 0) return 0;
  const int nptr = a.ngroups > 1 ? a.ngroups : a.nsegs;
  bool vecA = (a.lda % 4 == 0), vecB = (a.ldb % 4 == 0);
  for (int i = 0; i < nptr; ++i) {
    CN_CHECK(a.A[i] && a.B[i], "cartnet_gemm: null operand %d", i);
    vecA = vecA && aligned16(a.A[i]);
    vecB = vecB && aligned16(a.B[i]);
  }
  for (int gI = 0; gI < a.ngroups; ++gI) {
    CN_CHECK(a.C[gI] != nullptr, "cartnet_gemm: null output %d", gI);
    CN_CHECK((a.gather_i[gI] == nullptr) == (a.gather_j[gI] == nullptr), "cartnet_gemm: gather_i/gather_j must pair");
    if (a.gather_i[gI]) CN_CHECK(a.tgt && a.src, "cartnet_gemm: gather epilogue needs tgt/src");
    if (a.colsq[gI]) CN_CHECK(a.colsum[gI] != nullptr, "cartnet_gemm: colsq needs colsum");
  }
  CN_CHECK(a.lda >= (a.a_kstrided ? a.M : a.K) || a.K == 0, "cartnet_gemm: lda=%d too small", a.lda);
  CN_CHECK(a.ldb >= (a.b_kstrided ? a.N : a.K) || a.K == 0, "cartnet_gemm: ldb=%d too small", a.ldb);
  CN_CHECK(a.ldc >= a.N, "cartnet_gemm: ldc=%d < N=%d", a.ldc, a.N);
  if (a.splitk > 1) {
    CN_CHECK(a.nsegs == 1, "cartnet_gemm: split-K with K-segments is not supported");
    for (int gI = 0; gI < a.ngroups; ++gI)
      CN_CHECK(!a.bias[gI] && !a.gather_i[gI] && !a.resid[gI] && !a.dact[gI] && !a.cpre[gI] && !a.colsum[gI] &&
                   !a.out_act,
               "cartnet_gemm: split-K writes raw partial slabs, no epilogue allowed");
  }
  GemmFlags fl;
  fl.vecA = vecA ? 1 : 0;
  fl.vecB = vecB ? 1 : 0;
  int kchunk = (a.K + a.splitk - 1) / a.splitk;
  kchunk = ((kchunk + BK - 1) / BK) * BK;
  if (kchunk == 0) kchunk = BK;
  fl.kchunk = kchunk;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!a.a_kstrided && !a.b_kstrided) launch_gemm<false, false>(a, fl, st);
  else if (!a.a_kstrided && a.b_kstrided) launch_gemm<false, true>(a, fl, st);
  else if (a.a_kstrided && a.b_kstrided) launch_gemm<true, true>(a, fl, st);
  else CN_CHECK(false, "cartnet_gemm: a_kstrided=1 with b_kstrided=0 is not a shape this path uses");
  CN_LAUNCH_CHECK("cartnet_gemm");
  return 0;
}

extern "C" int cartnet_splitk_reduce(const float* slabs, int32_t splitk, int32_t M, int32_t N, float* out,
                                     int32_t ldo, void* stream) {
  CN_CHECK(slabs && out, "cartnet_splitk_reduce: null pointer");
  CN_CHECK(splitk >= 1 && M >= 0 && N >= 0 && ldo >= N, "cartnet_splitk_reduce: bad shape");
  if (M == 0 || N == 0) return 0;
  const size_t total = (size_t)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cn_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     slabs, splitk, M, N, out, ldo);
  CN_LAUNCH_CHECK("cartnet_splitk_reduce");
  return 0;
}

extern "C" int cartnet_colsum_finalize(const double* parts, int32_t nparts, int32_t N, float* out, void* stream) {
  CN_CHECK(parts && out, "cartnet_colsum_finalize: null pointer");
  CN_CHECK(nparts >= 0 && N >= 0, "cartnet_colsum_finalize: bad shape");
  if (N == 0) return 0;
  hipLaunchKernelGGL(cn_colsum_finalize_kernel, dim3(cn_ceil_div(N, 64)), dim3(1024), 0,
                     reinterpret_cast<hipStream_t>(stream), parts, nparts, N, out);
  CN_LAUNCH_CHECK("cartnet_colsum_finalize");
  return 0;
}

extern "C" int cartnet_colsum_finalize_f32(const float* parts, int32_t nparts, int32_t N, float* out, void* stream) {
  CN_CHECK(parts && out, "cartnet_colsum_finalize_f32: null pointer");
  CN_CHECK(nparts >= 0 && N >= 0, "cartnet_colsum_finalize_f32: bad shape");
  if (N == 0) return 0;
  hipLaunchKernelGGL(cn_colsum_finalize_f32_kernel, dim3(cn_ceil_div(N, 128)), dim3(128), 0,
                     reinterpret_cast<hipStream_t>(stream), parts, nparts, N, out);
  CN_LAUNCH_CHECK("cartnet_colsum_finalize_f32");
  return 0;
}
