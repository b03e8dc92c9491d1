// The bf16x3 activation x weight kernel of gemm_x3.h on the 16x16x32 MFMA shape.
//
// Why: the bf16x3 main loop runs at 0.94 of the matrix pipe's cycles, but the chip holds only 1.53-1.70 GHz under it
// (2.37 GHz on zeros: tools/experiments/exp_clock_x3.py, tools/experiments/micro/x3_shape.hip) -- the loop is bound by the power the matrix
// work draws, and MI355X holds a higher clock on v_mfma_f32_16x16x32_bf16 than on v_mfma_f32_32x32x16_bf16 at equal
// cycles per FLOP (MI355X_MICROARCH.md, 'DVFS give-back' item 7).  Measured on this kernel's loop body in isolation,
// operands from LDS, random data: +7.8 % FLOP/s with the 20 fragment reads per K-step used here, +13.9 % at the 12
// reads of the 32x32 form (every ds_read_b128 per wave and K-step costs ~0.7 %).
//
// How, without touching the data path: the LDS images (three bf16 planes of [rows][16 k], x3_offset), the weight DMA,
// the activation split and the pipeline are those of cn_gemm_x3nn_kernel.  A 16x16x32 instruction sums over 32 k-slots;
// with 16 real k per K-step, slots 0-15 and 16-31 carry TWO of the six piece products:
//     [h|m] x [h|h] = hh + mh      [h|m] x [m|m] = hm + mm      [h|l] x [l|h] = hl + lh
// (lanes 0-31 read the first piece's plane, lanes 32-63 the second's: the plane is a per-lane address term, the
// fragment read is still one ds_read_b128).  Per wave and K-step: 4 row tiles x 2 + 4 column tiles x 3 = 20 reads, 48
// MFMAs of 16 cycles = the 768 matrix cycles of the 24 32x32x16 MFMAs.  Three passes over the 16 accumulator tiles
// (small terms first), so only one set of four A fragments and one B fragment are live at a time.
// The 64 x 64 wave tile is 4 x 4 tiles of f32x4; the wide epilogue takes that layout (gemm_kernel.h, acc_block_to_scr).
#pragma once
#include "gemm_x3.h"

namespace cn_gemm {

// GSTK (as in gemm_f32w128.h): 1 / 2 = the dE product with the gate statistics of the layer below in its epilogue, with /
// without the edge residual, as kernels of their own.
template <bool A_ACT, int GSTK = 0>
__global__ __launch_bounds__(NTHREADS, 4) void cn_gemm_x3nn16_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  using S = Shape<X3_BN>;
  static_assert(S::WM == 64 && S::WN == 64, "wave tile is 64 x 64");
  __shared__ __attribute__((aligned(16))) float smem[2 * X3_BUF_BYTES / 4];
  char* lds = reinterpret_cast<char*>(smem);

  CN_PHASE(0);
  CN_PHASE_ID();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / S::WGN, wn = wid % S::WGN;
  const int l16 = lane & 15, kh = (lane >> 4) & 1, up = lane >> 5;
  const int tiles_n = p.N / X3_BN;
  int bx, g;
  cn_block_map(bx, g, tiles_n);
  const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * X3_BN;
  const int nsteps = p.K / BK;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

  // staging: as cn_gemm_x3nn_kernel (one float4 of the A tile per thread, the B tile as three 1 KB DMA pieces per wave)
  const int arow = tid >> 2, akq = tid & 3;
  const unsigned a_voff = ((unsigned)min(row0 + arow, p.M - 1) * (unsigned)p.lda + akq * 4) * 4u;   // bytes
  const int a_lds = x3_offset(arow, akq >> 1) + (akq & 1) * 8;
  const unsigned b_voff = lane * 16;
  const float* a0 = p.A[g];
  const char* b0 = reinterpret_cast<const char*>(p.b_split[g]) + (size_t)tile_n * nsteps * X3_B_BYTES;
  auto a_issue = [&](f32x4& dst, int v) {
    const float* base = a0 + v * BK;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(a_voff), "s"(base) : "memory");
  };
  auto a_store = [&](f32x4 v, int buf) {
    if (A_ACT) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = fast_silu(v[c]);
    }
    char* dst = lds + buf * X3_BUF_BYTES + a_lds;
    const Split3 s = split3(v);
    *reinterpret_cast<bf16x4*>(dst) = s.h;
    *reinterpret_cast<bf16x4*>(dst + X3_A_PLANE) = s.m;
    *reinterpret_cast<bf16x4*>(dst + 2 * X3_A_PLANE) = s.l;
  };
  const unsigned lds_b = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + X3_A_BYTES + wid * 1024;
  auto b_issue = [&](int v, int buf) {
    const char* src = b0 + (size_t)v * X3_B_BYTES + wid * 1024;
    const unsigned dst = lds_b + buf * X3_BUF_BYTES;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(dst + j * 8192), "v"(b_voff), "s"(src + j * 8192) : "memory", "m0");
  };

  auto b_piece = [&](int v, int buf, int j) {   // one of the three pieces of b_issue
    const char* src = b0 + (size_t)v * X3_B_BYTES + wid * 1024 + j * 8192;
    const unsigned dst = lds_b + buf * X3_BUF_BYTES + j * 8192;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(dst), "v"(b_voff), "s"(src) : "memory", "m0");
  };

  // fragment addresses.  Row tile a of the wave: rows wm*64 + 16 a + l16, so bit 4 of the row is a & 1 and x3_offset's
  // swap is kh ^ (a & 1): one lane term for even tiles, one for odd ones; the plane of the second piece is a lane term too.
  const int a_even = (wm * 64 + l16) * 32 + (kh << 4), a_odd = (wm * 64 + l16) * 32 + ((kh ^ 1) << 4);
  const int b_even = X3_A_BYTES + (wn * 64 + l16) * 32 + (kh << 4), b_odd = X3_A_BYTES + (wn * 64 + l16) * 32 + ((kh ^ 1) << 4);
  const int a_hm = up ? X3_A_PLANE : 0, a_hl = up ? 2 * X3_A_PLANE : 0;           // [h|m], [h|l]
  const int b_lh = up ? 0 : 2 * X3_B_PLANE;                                        // [l|h]; [h|h] = 0, [m|m] = one plane
  auto rd = [&](const char* q) { return *reinterpret_cast<const bf16x8*>(q); };
#define CN_MMA16(a, b, x, y) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[a][b], 0, 0, 0)
  // the 48 MFMAs of one K-step on buffer `buf`
  auto mma_step = [&](const char* base) {
    const char* ae = base + a_even;
    const char* ao = base + a_odd;
    const char* be = base + b_even;
    const char* bo = base + b_odd;
    bf16x8 af[4], bf;
    // pass 1: hl + lh
    af[0] = rd(ae + a_hl);
    af[1] = rd(ao + a_hl + 512);
    af[2] = rd(ae + a_hl + 1024);
    af[3] = rd(ao + a_hl + 1536);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      bf = rd(((b & 1) ? bo : be) + b_lh + b * 512);
#pragma unroll
      for (int a = 0; a < 4; ++a) CN_MMA16(a, b, af[a], bf);
    }
    // pass 2: hm + mm, pass 3: hh + mh
    af[0] = rd(ae + a_hm);
    af[1] = rd(ao + a_hm + 512);
    af[2] = rd(ae + a_hm + 1024);
    af[3] = rd(ao + a_hm + 1536);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      bf = rd(((b & 1) ? bo : be) + X3_B_PLANE + b * 512);
#pragma unroll
      for (int a = 0; a < 4; ++a) CN_MMA16(a, b, af[a], bf);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      bf = rd(((b & 1) ? bo : be) + b * 512);
#pragma unroll
      for (int a = 0; a < 4; ++a) CN_MMA16(a, b, af[a], bf);
    }
  };
#undef CN_MMA16
  // one K-step; r holds the A tile of step u+1 on entry and receives the load of step u+3 (pipeline: gemm_x3.h)
  auto step = [&](auto cur_c, int u, f32x4& r) {
    constexpr int CUR = decltype(cur_c)::value;
    if (u + 1 < nsteps) {
      b_issue(u + 1, CUR ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      a_store(r, CUR ^ 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (u + 3 < nsteps) a_issue(r, u + 3);
    __builtin_amdgcn_sched_barrier(0);
    mma_step(lds + CUR * X3_BUF_BYTES);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 3 < nsteps) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // The same K-step for the steady state (u + 3 < nsteps), placed by hand: after the barrier every wave of the
  // workgroup is at the same point, so whatever is not an MFMA at the top of the step leaves the matrix pipe idle in
  // all of them at once (the compiler-scheduled `step` above: 3,810 cycles per K-step against the pipe's 3,072).  Here
  // the step opens with the fragment reads, and the DMA issue, the three slices of the split, its LDS writes and the A
  // load sit between groups of four MFMAs; B fragments alternate between two registers, two groups ahead; the second
  // set of A fragments arrives in two halves around the last group that uses the first.
  // Where the three DMA pieces of the next weight tile go (measured with clock stamps at K = 1024, same box, cycles per
  // K-step / in-kernel clock / launch time): 0 = all three right after the barrier 3,619 / 1.72-1.74 GHz / 825-830 us;
  // 1 = one after each of the first three MFMA groups 3,507 / 1.65 / 819; 2 = after groups 4-6 3,660 / 1.76 / 827;
  // 3 = after groups 1, 4, 7 3,676 / 1.74 / 834.  Without the DMA at all the step takes 3,387 cycles: the issue of a
  // piece is the loop's largest non-matrix cost.  Cycles saved come back partly as a lower clock (the loop is
  // power-bound): placement 1 is worth ~1 % of the launch.
#ifndef CN_DMA_POS
#define CN_DMA_POS 1
#endif
  // after MFMA group n (0..11) of the step: DMA piece j if the placement says so, then the A load once all three are out
#define CN_AFTER(n)                                                                                      \
  if (CN_DMA_POS == 1 && (n) <= 2) b_piece(u + 1, CUR ^ 1, (n));                                           \
  if (CN_DMA_POS == 2 && (n) >= 4 && (n) <= 6) b_piece(u + 1, CUR ^ 1, (n) - 4);                           \
  if (CN_DMA_POS == 3 && ((n) == 1 || (n) == 4 || (n) == 7)) b_piece(u + 1, CUR ^ 1, ((n) - 1) / 3);       \
  if ((CN_DMA_POS <= 1 && (n) == 2) || (CN_DMA_POS == 2 && (n) == 6) || (CN_DMA_POS == 3 && (n) == 7)) a_issue(r, u + 3)
#define CN_SB() __builtin_amdgcn_sched_barrier(0)
#define CN_G(A0, A1, A2, A3, B, b)                                                          \
  acc[0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, B, acc[0][b], 0, 0, 0);          \
  acc[1][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B, acc[1][b], 0, 0, 0);          \
  acc[2][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B, acc[2][b], 0, 0, 0);          \
  acc[3][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3, B, acc[3][b], 0, 0, 0)
  auto step_full = [&](auto cur_c, int u, f32x4& r) {
    constexpr int CUR = decltype(cur_c)::value;
    const char* base = lds + CUR * X3_BUF_BYTES;
    const char* ae = base + a_even;
    const char* ao = base + a_odd;
    const char* be = base + b_even;
    const char* bo = base + b_odd;
    char* wdst = lds + (CUR ^ 1) * X3_BUF_BYTES + a_lds;
    bf16x8 p0 = rd(ae + a_hl), p1 = rd(ao + a_hl + 512), p2 = rd(ae + a_hl + 1024), p3 = rd(ao + a_hl + 1536);
    bf16x8 b0 = rd(be + b_lh), b1 = rd(bo + b_lh + 512);
    CN_SB();
    if (CN_DMA_POS == 0) b_issue(u + 1, CUR ^ 1);
    CN_SB();
    if (A_ACT) {
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] = fast_silu(r[c]);
    }
    // split slices on PACKED conversions: one v_cvt_pk_bf16_f32 per pair is both the stored piece and, shifted / masked,
    // its two fp32 values (18 vector instructions per K-step instead of 30; removing the split altogether saves 82 of
    // the step's ~3,600 cycles, so this is worth little -- it is here because it is not slower)
    u32x2 sh, sm, sl;
    f32x4 r1, r2;
    x3_slice(r, sh, r1);
    CN_SB();
    CN_G(p0, p1, p2, p3, b0, 0);                 // pass 1: hl + lh
    CN_SB();
    CN_AFTER(0);
    CN_SB();
    b0 = rd(be + b_lh + 1024);
    x3_slice(r1, sm, r2);
    CN_SB();
    CN_G(p0, p1, p2, p3, b1, 1);
    CN_SB();
    CN_AFTER(1);
    CN_SB();
    b1 = rd(bo + b_lh + 1536);
    sl[0] = x3_pk(r2[0], r2[1]);
    sl[1] = x3_pk(r2[2], r2[3]);
    *reinterpret_cast<u32x2*>(wdst) = sh;
    *reinterpret_cast<u32x2*>(wdst + X3_A_PLANE) = sm;
    *reinterpret_cast<u32x2*>(wdst + 2 * X3_A_PLANE) = sl;
    CN_SB();
    CN_G(p0, p1, p2, p3, b0, 2);
    CN_SB();
    CN_AFTER(2);
    CN_SB();
    b0 = rd(be + X3_B_PLANE);
    bf16x8 q0 = rd(ae + a_hm), q1 = rd(ao + a_hm + 512);
    CN_SB();
    CN_G(p0, p1, p2, p3, b1, 3);
    CN_SB();
    CN_AFTER(3);
    CN_SB();
    bf16x8 q2 = rd(ae + a_hm + 1024), q3 = rd(ao + a_hm + 1536);
    b1 = rd(bo + X3_B_PLANE + 512);
    CN_SB();
    CN_G(q0, q1, q2, q3, b0, 0);                 // pass 2: hm + mm
    CN_SB();
    CN_AFTER(4);
    CN_SB();
    b0 = rd(be + X3_B_PLANE + 1024);
    CN_SB();
    CN_G(q0, q1, q2, q3, b1, 1);
    CN_SB();
    CN_AFTER(5);
    CN_SB();
    b1 = rd(bo + X3_B_PLANE + 1536);
    CN_SB();
    CN_G(q0, q1, q2, q3, b0, 2);
    CN_SB();
    CN_AFTER(6);
    CN_SB();
    b0 = rd(be);
    CN_SB();
    CN_G(q0, q1, q2, q3, b1, 3);
    CN_SB();
    CN_AFTER(7);
    CN_SB();
    b1 = rd(bo + 512);
    CN_SB();
    CN_G(q0, q1, q2, q3, b0, 0);                 // pass 3: hh + mh
    CN_SB();
    CN_AFTER(8);
    CN_SB();
    b0 = rd(be + 1024);
    CN_SB();
    CN_G(q0, q1, q2, q3, b1, 1);
    CN_SB();
    CN_AFTER(9);
    CN_SB();
    b1 = rd(bo + 1536);
    CN_SB();
    CN_G(q0, q1, q2, q3, b0, 2);
    CN_SB();
    CN_AFTER(10);
    CN_G(q0, q1, q2, q3, b1, 3);
    CN_SB();
    CN_AFTER(11);
    CN_SB();
    asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#undef CN_G
#undef CN_AFTER
#undef CN_SB

  if (nsteps > 0) {
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
    CN_STAMP_BEGIN();
    CN_PHASE(1);
    int u = 0;
    for (; u + 4 < nsteps; u += 2) {        // both steps satisfy u + 3 < nsteps
      step_full(std::integral_constant<int, 0>{}, u, r1);
      step_full(std::integral_constant<int, 1>{}, u + 1, r0);
    }
    for (; u < nsteps; u += 2) {
      step(std::integral_constant<int, 0>{}, u, r1);
      if (u + 1 < nsteps) step(std::integral_constant<int, 1>{}, u + 1, r0);
    }
    CN_STAMP_END();
    CN_PHASE(2);
  }
  if constexpr (GSTK == 1) epilogue_wide<X3_BN, 130>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem, 130);
  else if constexpr (GSTK == 2) epilogue_wide<X3_BN, 128>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem, 128);
  else x3_epilogue<0>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem);
#ifdef CN_PHASE_STAMP
  CN_PHASE(3);                                             // last store issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CN_PHASE(4);                                             // ... and acknowledged
#endif
}

}  // namespace cn_gemm
