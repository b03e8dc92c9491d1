// The A_ACT form of cn_gemm_x3nn_kernel (gemm_x3.h) that also WRITES the activated operand silu(A) in fp32
// (CartnetGemmArgs.a_act_out), exactly as gemm_f32ao.h does for the fp32-MFMA kernel: the value is in registers on its
// way to being split, one 16-byte global store per thread per K-step.  The weight gradient dW = dY^T silu(pre) then
// takes the plain-operand transposing-read kernel (207 us in-step against 476 us with the SiLU recomputed on both
// column halves).  Own translation unit (gemm_x3ao.hip).
#pragma once
#include "gemm_x3.h"

namespace cn_gemm {

template <bool ONE>
__global__ __launch_bounds__(NTHREADS, 4) void cn_gemm_x3nn_actout_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  constexpr bool A_ACT = true;
  using S = Shape<X3_BN>;
  static_assert(S::TM == 2 && S::TN == 2, "wave tile is 64 x 64");
  __shared__ __attribute__((aligned(16))) float smem[2 * X3_BUF_BYTES / 4];
  char* lds = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / S::WGN, wn = wid % S::WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_n = p.N / X3_BN;
  int bx, g;
  cn_block_map(bx, g, tiles_n);
  const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * X3_BN;
  const int nk = p.K / BK;
  const int nsteps = nk;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // this thread's share of the A tile: row tid/4, k-quad tid%4 (rows past M are clamped; the epilogue drops them)
  const int arow = tid >> 2, akq = tid & 3;
  const unsigned a_voff = ((unsigned)min(row0 + arow, p.M - 1) * (unsigned)p.lda + akq * 4) * 4u;   // bytes
  const int a_lds = x3_offset(arow, akq >> 1) + (akq & 1) * 8;
  const unsigned b_voff = lane * 16;
  const size_t b_tile = (size_t)tile_n * nk * X3_B_BYTES;

  // one K range per launch (the host folds K-segments that are adjacent column blocks of one matrix into one K)
  const float* a0 = p.A[g];
  const char* b0 = reinterpret_cast<const char*>(p.b_split[g]) + b_tile;
  auto a_base = [&](int v) -> const float* { return a0 + v * BK; };
  auto b_base = [&](int v) -> const char* { return b0 + (size_t)v * X3_B_BYTES; };
  // register load of this thread's float4 of K-step v, outside the compiler's memory-counter bookkeeping
  auto a_issue = [&](f32x4& dst, int v) {
    const float* base = a_base(v);
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(a_voff), "s"(base) : "memory");
  };
  // the activated fp32 values leave for memory as they are staged (see gemm_f32ao.h: no predicate, duplicates write
  // identical bits)
  // (scalar base + the A tile's own 32-bit lane offset, as in a_issue: no address VGPRs)
  const float* h0 = p.a_act_out[g];
  auto h_store = [&](f32x4 v, int step_v) {
    const float* base = h0 + step_v * BK;
    // s_nop: a store of more than 8 bytes reads its data registers up to two cycles after issue; the hazard
    // recogniser does not look inside inline asm, and the next VALU instruction may overwrite them
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(a_voff), "v"(v), "s"(base) : "memory");
  };
  auto a_store = [&](f32x4 v, int buf, int step_v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = fast_silu(v[c]);
    h_store(v, step_v);
    char* dst = lds + buf * X3_BUF_BYTES + a_lds;
    if constexpr (ONE) {   // plain bf16 operands (precision 2): the high piece only
      *reinterpret_cast<bf16x4*>(dst) = __builtin_convertvector(v, bf16x4);
    } else {
      const Split3 s = split3(v);
      *reinterpret_cast<bf16x4*>(dst) = s.h;
      *reinterpret_cast<bf16x4*>(dst + X3_A_PLANE) = s.m;
      *reinterpret_cast<bf16x4*>(dst + 2 * X3_A_PLANE) = s.l;
    }
  };
  // B tile of K-step v: 24 KB, a lane-linear copy; wave w moves the 1 KB pieces w, w+8, w+16.  Issued as inline asm
  // (scalar base + one lane-offset VGPR; the builtin keeps a 64-bit address pair per piece in VGPRs, which this
  // kernel cannot afford at 128 registers).
  const unsigned lds_b = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + X3_A_BYTES + wid * 1024;
  auto b_issue = [&](int v, int buf) {
    const char* src = b_base(v) + wid * 1024;
    const unsigned dst = lds_b + buf * X3_BUF_BYTES;
#pragma unroll
    for (int j = 0; j < (ONE ? 1 : 3); ++j)   // piece j of every wave belongs to plane j (h, m, l)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(dst + j * 8192), "v"(b_voff), "s"(src + j * 8192) : "memory", "m0");
  };
  bf16x8 ah[2], am[2], al[2], bh, bm, bl;
  auto frag_a = [&](int buf) {
    const char* cA = lds + buf * X3_BUF_BYTES;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const char* q = cA + x3_offset(wm * S::WM + a * 32 + li, lh);
      ah[a] = *reinterpret_cast<const bf16x8*>(q);
      if constexpr (!ONE) {
        am[a] = *reinterpret_cast<const bf16x8*>(q + X3_A_PLANE);
        al[a] = *reinterpret_cast<const bf16x8*>(q + 2 * X3_A_PLANE);
      }
    }
  };
  auto frag_b = [&](int buf, int b) {
    const char* q = lds + buf * X3_BUF_BYTES + X3_A_BYTES + x3_offset(wn * S::WN + b * 32 + li, lh);
    bh = *reinterpret_cast<const bf16x8*>(q);
    if constexpr (!ONE) {
      bm = *reinterpret_cast<const bf16x8*>(q + X3_B_PLANE);
      bl = *reinterpret_cast<const bf16x8*>(q + 2 * X3_B_PLANE);
    }
  };
  auto mma = [&](int b) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {   // small terms first
      if constexpr (ONE) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh, acc[a][b], 0, 0, 0);
        continue;
      }
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh, acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl, acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bm, acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bh, acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bm, acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh, acc[a][b], 0, 0, 0);
    }
  };
  // one K-step; r holds the A tile of step u+1 on entry and receives the load of step u+3
  auto step = [&](auto cur_c, int u, f32x4& r) {
    constexpr int CUR = decltype(cur_c)::value;
    frag_a(CUR);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < nsteps) {
      a_store(r, CUR ^ 1, u + 1);
      __builtin_amdgcn_sched_barrier(0);
      b_issue(u + 1, CUR ^ 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (u + 3 < nsteps) a_issue(r, u + 3);
    __builtin_amdgcn_sched_barrier(0);
    frag_b(CUR, 0);
    mma(0);
    frag_b(CUR, 1);
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 3 < nsteps) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // The same K-step for the steady state (u + 3 < nsteps: nothing conditional), scheduled by hand: the MFMA chain
  // starts as soon as the first fragments are in, and everything else of the step -- the split of the next A tile
  // (three slices), its LDS writes, the DMA issue, the A load, the second half's B fragments -- sits in the shadow of
  // MFMAs (an MFMA holds the issue port 8 of its 32 cycles).  The B registers are reloaded with the second column
  // half as soon as the last MFMA that reads them has issued: products are ordered by B plane (h, m, l) per half.
#define CN_SB() __builtin_amdgcn_sched_barrier(0)
#define CN_MMA(a, b, x, y) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a][b], 0, 0, 0)
  auto step_full = [&](auto cur_c, int u, f32x4& r) {
    constexpr int CUR = decltype(cur_c)::value;
    const char* qb = lds + CUR * X3_BUF_BYTES + X3_A_BYTES + x3_offset(wn * S::WN + li, lh);
    char* wdst = lds + (CUR ^ 1) * X3_BUF_BYTES + a_lds;
    frag_a(CUR);
    bh = *reinterpret_cast<const bf16x8*>(qb);
    bm = *reinterpret_cast<const bf16x8*>(qb + X3_B_PLANE);
    bl = *reinterpret_cast<const bf16x8*>(qb + 2 * X3_B_PLANE);
    CN_SB();
    b_issue(u + 1, CUR ^ 1);     // first thing after the barrier that freed the buffer: the DMA gets the whole K-step to
    CN_SB();                     // land (issued after the third MFMA pair the plain kernel ran 242 us, here 230 us)
    if (A_ACT) {
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] = fast_silu(r[c]);
    }
    h_store(r, u + 1);            // after the DMA issue, before the A load: vmcnt(1) below still names that load
    const bf16x4 sh = __builtin_convertvector(r, bf16x4);                       // slice 1 (hides the LDS latency)
    const f32x4 r1 = r - __builtin_convertvector(sh, f32x4);
    CN_SB();
    CN_MMA(0, 0, al[0], bh);
    CN_MMA(1, 0, al[1], bh);
    CN_SB();
    const bf16x4 sm = __builtin_convertvector(r1, bf16x4);                      // slice 2
    const f32x4 r2 = r1 - __builtin_convertvector(sm, f32x4);
    CN_SB();
    CN_MMA(0, 0, am[0], bh);
    CN_MMA(1, 0, am[1], bh);
    CN_SB();
    const bf16x4 sl = __builtin_convertvector(r2, bf16x4);                      // slice 3 + the three LDS writes
    *reinterpret_cast<bf16x4*>(wdst) = sh;
    *reinterpret_cast<bf16x4*>(wdst + X3_A_PLANE) = sm;
    *reinterpret_cast<bf16x4*>(wdst + 2 * X3_A_PLANE) = sl;
    CN_SB();
    CN_MMA(0, 0, ah[0], bh);
    CN_MMA(1, 0, ah[1], bh);
    CN_SB();
    bh = *reinterpret_cast<const bf16x8*>(qb + 32 * 32);                        // second column half, high plane
    CN_SB();
    CN_MMA(0, 0, am[0], bm);
    CN_MMA(1, 0, am[1], bm);
    CN_SB();
    a_issue(r, u + 3);
    CN_SB();
    CN_MMA(0, 0, ah[0], bm);
    CN_MMA(1, 0, ah[1], bm);
    CN_SB();
    bm = *reinterpret_cast<const bf16x8*>(qb + 32 * 32 + X3_B_PLANE);
    CN_SB();
    CN_MMA(0, 0, ah[0], bl);
    CN_MMA(1, 0, ah[1], bl);
    CN_SB();
    bl = *reinterpret_cast<const bf16x8*>(qb + 32 * 32 + 2 * X3_B_PLANE);
    CN_SB();
    CN_MMA(0, 1, al[0], bh);
    CN_MMA(1, 1, al[1], bh);
    CN_MMA(0, 1, am[0], bh);
    CN_MMA(1, 1, am[1], bh);
    CN_MMA(0, 1, ah[0], bh);
    CN_MMA(1, 1, ah[1], bh);
    CN_MMA(0, 1, am[0], bm);
    CN_MMA(1, 1, am[1], bm);
    CN_MMA(0, 1, ah[0], bm);
    CN_MMA(1, 1, ah[1], bm);
    CN_MMA(0, 1, ah[0], bl);
    CN_MMA(1, 1, ah[1], bl);
    CN_SB();
    asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#undef CN_MMA
#undef CN_SB

  if (nsteps > 0) {
    // all of the pipeline head's loads in ONE memory round trip (K-step 0 into a third register set, 1 and 2 into the ring)
    f32x4 r0, r1, rt;
    a_issue(rt, 0);
    b_issue(0, 0);
    if (nsteps > 1) a_issue(r1, 1);
    if (nsteps > 2) a_issue(r0, 2);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(rt) :: "memory");
    a_store(rt, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int u = 0;
    if constexpr (!ONE) {
      for (; u + 4 < nsteps; u += 2) {        // both steps satisfy u + 3 < nsteps
        step_full(std::integral_constant<int, 0>{}, u, r1);
        step_full(std::integral_constant<int, 1>{}, u + 1, r0);
      }
    }
    for (; u < nsteps; u += 2) {
      step(std::integral_constant<int, 0>{}, u, r1);
      if (u + 1 < nsteps) step(std::integral_constant<int, 1>{}, u + 1, r0);
    }
  }
  x3_epilogue<0>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem);
}

}  // namespace cn_gemm
