// The DMA-fed fp32-MFMA kernel of gemm_f32.h with a 128 x 128 output tile.  Same memory pipeline
// (weight tile by direct-to-LDS DMA from the cartnet_gemm_pack_b image -- a 128-column tile is one 8 KB half of the
// image's [256][16] block -- activation tile through a two-deep register ring, counted s_waitcnt + raw s_barrier), but
// each wave owns 64 x 32 outputs = 32 accumulator VGPRs, so that the kernel fits ~80 VGPRs and 36 KB of LDS and THREE
// workgroups share a CU (6 waves per SIMD) instead of two: more waves to cover each other's epilogues and barriers,
// at the price of reading every activation row tile once per 128 columns.  Used for the gather-epilogue launches only
// (use_f32nn128 in gemm.hip has the measurements).  Own translation unit (gemm_f32w128.hip) so that it cannot disturb the
// register allocation of the 256-wide kernels.
#pragma once
#include "gemm_f32.h"

namespace cn_gemm {

constexpr int W128_BN = 128;
constexpr int W128_B_BYTES = W128_BN * BK * 4;          // 8 KB per K-step per 128-column tile
constexpr int W128_BUF_BYTES = F32_A_BYTES + W128_B_BYTES;

// GSTK != 0: the dE product with the gate statistics of the layer below in its epilogue (CartnetGemmArgs.gst_*) as kernels
// of their own -- 1: with the edge residual (epilogue kind 2 | 128), 2: without (kind 128; the last layer: the head does
// not read the edge features) -- so that the extra epilogue registers cannot disturb the allocation of the other forms
// (one kernel carrying both kinds spilled 30 registers and lost 27 us per launch; each alone: 80 VGPRs, none).
// (A kernel of its own for the node-term gather form -- kind 1, no spills either -- measured within noise: 393-394 vs 394-399 us.)
template <bool A_ACT, int GSTK = 0>
__global__ __launch_bounds__(NTHREADS, 6) void cn_gemm_f32nn128_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  using S = Shape<W128_BN>;
  static_assert(S::TM == 2 && S::TN == 1 && S::WGM == 2 && S::WGN == 4, "wave tile is 64 x 32");
  static_assert(2 * W128_BUF_BYTES / 4 >= (NTHREADS / 64) * SCR_FLOATS, "epilogue scratch must fit");
  __shared__ __attribute__((aligned(16))) float smem[2 * W128_BUF_BYTES / 4];
  char* lds = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / S::WGN, wn = wid % S::WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_n = p.N / W128_BN;
  int bx, g;
  cn_block_map(bx, g, tiles_n);
  const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * W128_BN;
  const int nsteps = p.K / BK;

  f32x16 acc[2][1];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;

  const int arow = tid >> 2, akq = tid & 3;
  const unsigned a_voff = ((unsigned)min(row0 + arow, p.M - 1) * (unsigned)p.lda + akq * 4) * 4u;
  const int a_lds = (arow * KPAD + akq * 4) * 4;
  const unsigned b_voff = lane * 16;
  const float* a0 = p.A[g];
  // image: per 256-column tile and K-step a [256][16] block of 16 KB; this tile is its half (tile_n & 1)
  const char* b0 = reinterpret_cast<const char*>(p.b_split[g]) + (size_t)(tile_n >> 1) * nsteps * F32_B_BYTES +
                   (size_t)(tile_n & 1) * W128_B_BYTES;
  const unsigned lds_b = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + F32_A_BYTES + wid * 1024;

  auto a_issue = [&](f32x4& dst, int v) {
    const float* base = a0 + v * BK;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(a_voff), "s"(base) : "memory");
  };
  auto a_store = [&](f32x4 v, int buf) {
    if (A_ACT) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = fast_silu(v[c]);
    }
    *reinterpret_cast<f32x4*>(lds + buf * W128_BUF_BYTES + a_lds) = v;
  };
  // B tile of K-step v: 8 pieces of 1 KB, one per wave
  auto b_issue = [&](int v, int buf) {
    const char* src = b0 + (size_t)v * F32_B_BYTES + wid * 1024;
    const unsigned dst = lds_b + buf * W128_BUF_BYTES;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(dst), "v"(b_voff), "s"(src) : "memory", "m0");
  };
  f32x4 af[2][2], bf[2];
  auto frags = [&](int buf, int kg) {
    const float* sA = reinterpret_cast<const float*>(lds + buf * W128_BUF_BYTES);
    const char* sB = lds + buf * W128_BUF_BYTES + F32_A_BYTES;
#pragma unroll
    for (int a = 0; a < 2; ++a)
      af[kg][a] = *reinterpret_cast<const f32x4*>(&sA[(wm * S::WM + a * 32 + li) * KPAD + kg * 8 + lh * 4]);
    bf[kg] = *reinterpret_cast<const f32x4*>(sB + f32_swz(wn * S::WN + li, kg * 2 + lh));
  };
  auto mma2 = [&](int kg, int j) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
      acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg][a][j], bf[kg][j], acc[a][0], 0, 0, 0);
  };
  auto step = [&](auto cur_c, int u, f32x4& r) {
    constexpr int CUR = decltype(cur_c)::value;
    frags(CUR, 0);
    frags(CUR, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < nsteps) {
      a_store(r, CUR ^ 1);
      b_issue(u + 1, CUR ^ 1);
    }
    if (u + 3 < nsteps) a_issue(r, u + 3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int j = 0; j < 4; ++j) mma2(kg, j);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 3 < nsteps) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto step_full = [&](auto cur_c, int u, f32x4& r) {
    constexpr int CUR = decltype(cur_c)::value;
    frags(CUR, 0);
    frags(CUR, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (A_ACT) {
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] = fast_silu(r[c]);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma2(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<f32x4*>(lds + (CUR ^ 1) * W128_BUF_BYTES + a_lds) = r;
    __builtin_amdgcn_sched_barrier(0);
    mma2(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    b_issue(u + 1, CUR ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    mma2(0, 2);
    __builtin_amdgcn_sched_barrier(0);
    a_issue(r, u + 3);
    __builtin_amdgcn_sched_barrier(0);
    mma2(0, 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) mma2(1, j);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (nsteps > 0) {
    // all of the pipeline head's loads in ONE memory round trip (K-step 0 into a third register set, 1 and 2 into the ring)
    f32x4 r0, r1, rt;
    a_issue(rt, 0);
    b_issue(0, 0);
    if (nsteps > 1) a_issue(r1, 1);
    if (nsteps > 2) a_issue(r0, 2);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(rt) :: "memory");
    a_store(rt, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int u = 0;
    for (; u + 4 < nsteps; u += 2) {
      step_full(std::integral_constant<int, 0>{}, u, r1);
      step_full(std::integral_constant<int, 1>{}, u + 1, r0);
    }
    for (; u < nsteps; u += 2) {
      step(std::integral_constant<int, 0>{}, u, r1);
      if (u + 1 < nsteps) step(std::integral_constant<int, 1>{}, u + 1, r0);
    }
  }
#define CN_EPIW(K) epilogue_wide<W128_BN, K>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem, kind)
  if constexpr (GSTK == 1) {
    const int kind = 130;
    CN_EPIW(130);
  } else if constexpr (GSTK == 2) {
    const int kind = 128;
    CN_EPIW(128);
  } else {
    const int kind = (p.gather_i[g] ? 1 : 0) | (p.resid[g] ? 2 : 0) | (p.dact[g] ? 4 : 0) |
                     (p.colsum[g] ? (p.colsq[g] ? 16 : 8) : 0) | (p.cpre[g] ? 32 : 0) | (p.out_act ? 64 : 0) |
                   (p.dact_kind ? 256 : 0);
    switch (kind) {
      case 0: CN_EPIW(0); break;
      case 1: CN_EPIW(1); break;
      case 16: CN_EPIW(16); break;
      case 96: CN_EPIW(96); break;
      case 2: CN_EPIW(2); break;
      case 4: CN_EPIW(4); break;
      case 8: CN_EPIW(8); break;      // column sums only (iComformer: bias gradients of lin_key / lin_value)
      case 12: CN_EPIW(12); break;
      case 14: CN_EPIW(14); break;
      case 268: CN_EPIW(268); break;  // * softplus'(pre), bias gradient (+ resid: 270): iComformer's RBF branches
      case 270: CN_EPIW(270); break;
      case 352: CN_EPIW(352); break;  // keep the pre-activation, softplus: the RBF branches forward
      default: CN_EPIW(-1); break;
    }
  }
#undef CN_EPIW
}

}  // namespace cn_gemm
