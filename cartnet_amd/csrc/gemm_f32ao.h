// The A_ACT form of cn_gemm_f32nn_kernel (gemm_f32.h) that also WRITES the activated operand: silu(A) is computed in
// registers on its way into LDS anyway (once per element per group when the output is one 256-column tile wide), so
// storing it costs one 16-byte global store per thread per K-step in the shadow of the MFMAs.  The second Linear's
// weight gradient dW = dY^T silu(pre) then reads silu(pre) as a plain operand and takes the all-DMA kernel
// (380 us against 471 us for the register-staged kernel that recomputes the SiLU).  Own translation unit
// (gemm_f32ao.hip): the hand-placed K-step of the shared kernels is not disturbed.
//
// vmcnt accounting: the steady-state K-step ends with s_waitcnt vmcnt(1) = "all but the youngest memory operation have
// retired", the youngest being the A load of step u+3.  The extra store is issued BEFORE that load (it retires in
// order), so the count still names the same load.
#pragma once
#include "gemm_f32.h"

// Cache policy of the silu(A) store.  Its next reader is the weight gradient of this Linear, milliseconds later: written
// through and not kept (sc0 sc1 nt), the 363 MB per launch stop evicting the output `gs` that the gate kernel reads next
// -- same-box A B A B A B, 60 steps each: 14.01-14.06 vs 14.19-14.20 ms per step (nt alone / sc1 alone: half of that;
// the same bits on the weight-gradient kernel's operand loads, non-temporal A loads here: nothing / +0.1 ms).
#ifdef CN_NO_STREAM_STORES
#define CN_AO_STORE_POLICY ""
#else
#define CN_AO_STORE_POLICY "sc0 sc1 nt"
#endif

namespace cn_gemm {

__global__ __launch_bounds__(NTHREADS, 4) void cn_gemm_f32nn_actout_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  constexpr bool A_ACT = true;
  using S = Shape<F32_BN>;
  static_assert(S::TM == 2 && S::TN == 2, "wave tile is 64 x 64");
  __shared__ __attribute__((aligned(16))) float smem[2 * F32_BUF_BYTES / 4];
  char* lds = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / S::WGN, wn = wid % S::WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_n = p.N / F32_BN;
  int bx, g;
  cn_block_map(bx, g, tiles_n);
  const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * F32_BN;
  const int nsteps = p.K / BK;

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
  const int a_lds = (arow * KPAD + akq * 4) * 4;
  const unsigned b_voff = lane * 16;
  const float* a0 = p.A[g];
  const char* b0 = reinterpret_cast<const char*>(p.b_split[g]) + (size_t)tile_n * nsteps * F32_B_BYTES;
  const unsigned lds_b = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + F32_A_BYTES + wid * 1024;

  auto a_issue = [&](f32x4& dst, int v) {
    const float* base = a0 + v * BK;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(a_voff), "s"(base) : "memory");
  };
  // the activated A values leave for memory as they are staged: K-step v of this thread's row -> h[row, v*16 + akq*4 ..].
  // No predicate: rows past M are clamped to row M-1 (a_voff) and a second column tile stages the same values, so every
  // duplicate store writes identical bits to the same address.
  // (scalar base + the A tile's own 32-bit lane offset, as in a_issue: no address VGPRs)
  const float* h0 = p.a_act_out[g];
  auto h_store = [&](f32x4 v, int step_v) {
    const float* base = h0 + step_v * BK;
    // s_nop: a store of more than 8 bytes reads its data registers up to two cycles after issue; the hazard
    // recogniser does not look inside inline asm, and the next VALU instruction may overwrite them
    asm volatile("global_store_dwordx4 %0, %1, %2 " CN_AO_STORE_POLICY "\n\ts_nop 1" :: "v"(a_voff), "v"(v), "s"(base) : "memory");
  };
  auto a_store = [&](f32x4 v, int buf, int step_v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = fast_silu(v[c]);
    *reinterpret_cast<f32x4*>(lds + buf * F32_BUF_BYTES + a_lds) = v;
    h_store(v, step_v);
  };
  // B tile of K-step v: 16 pieces of 1 KB; wave w moves pieces w and w+8
  auto b_issue = [&](int v, int buf) {
    const char* src = b0 + (size_t)v * F32_B_BYTES + wid * 1024;
    const unsigned dst = lds_b + buf * F32_BUF_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(dst + j * 8192), "v"(b_voff), "s"(src + j * 8192) : "memory", "m0");
  };
  f32x4 af[2][2], bf[2][2];     // [k-group of 8][tile]: k = kg*8 + lh*4 + j for element j
  auto frags = [&](int buf, int kg) {
    const float* sA = reinterpret_cast<const float*>(lds + buf * F32_BUF_BYTES);
    const float* sB = reinterpret_cast<const float*>(lds + buf * F32_BUF_BYTES + F32_A_BYTES);
#pragma unroll
    for (int a = 0; a < 2; ++a)
      af[kg][a] = *reinterpret_cast<const f32x4*>(&sA[(wm * S::WM + a * 32 + li) * KPAD + kg * 8 + lh * 4]);
#pragma unroll
    for (int b = 0; b < 2; ++b)
      bf[kg][b] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(sB) +
                                                  f32_swz(wn * S::WN + b * 32 + li, kg * 2 + lh));
  };
  auto mma4 = [&](int kg, int j) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg][a][j], bf[kg][b][j], acc[a][b], 0, 0, 0);
  };
  // generic K-step (pipeline head and tail); r holds the A tile of step u+1 on entry, receives the load of step u+3
  auto step = [&](auto cur_c, int u, f32x4& r) {
    constexpr int CUR = decltype(cur_c)::value;
    frags(CUR, 0);
    frags(CUR, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < nsteps) {
      a_store(r, CUR ^ 1, u + 1);
      b_issue(u + 1, CUR ^ 1);
    }
    if (u + 3 < nsteps) a_issue(r, u + 3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int j = 0; j < 4; ++j) mma4(kg, j);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 3 < nsteps) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // steady state (u + 3 < nsteps): fragments of both k-groups first, then the MFMA chain with the staging of the next
  // tiles in its shadow
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
    mma4(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<f32x4*>(lds + (CUR ^ 1) * F32_BUF_BYTES + a_lds) = r;
    __builtin_amdgcn_sched_barrier(0);
    mma4(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    h_store(r, u + 1);
    __builtin_amdgcn_sched_barrier(0);
    b_issue(u + 1, CUR ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    mma4(0, 2);
    __builtin_amdgcn_sched_barrier(0);
    a_issue(r, u + 3);
    __builtin_amdgcn_sched_barrier(0);
    mma4(0, 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) mma4(1, j);
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
    a_store(rt, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int u = 0;
    for (; u + 4 < nsteps; u += 2) {        // both steps satisfy u + 3 < nsteps
      step_full(std::integral_constant<int, 0>{}, u, r1);
      step_full(std::integral_constant<int, 1>{}, u + 1, r0);
    }
    for (; u < nsteps; u += 2) {
      step(std::integral_constant<int, 0>{}, u, r1);
      if (u + 1 < nsteps) step(std::integral_constant<int, 1>{}, u + 1, r0);
    }
  }
  // epilogue (shared with gemm_kernel.h)
  const int kind = (p.gather_i[g] ? 1 : 0) | (p.resid[g] ? 2 : 0) | (p.dact[g] ? 4 : 0) |
                   (p.colsum[g] ? (p.colsq[g] ? 16 : 8) : 0) | (p.cpre[g] ? 32 : 0) | (p.out_act ? 64 : 0) |
                   (p.dact_kind ? 256 : 0);
#define CN_EPIW(K) epilogue_wide<F32_BN, K>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem, kind)
  switch (kind) {
    case 0: CN_EPIW(0); break;
    case 1: CN_EPIW(1); break;
    case 16: CN_EPIW(16); break;
    case 96: CN_EPIW(96); break;
    case 2: CN_EPIW(2); break;
    case 4: CN_EPIW(4); break;
    case 12: CN_EPIW(12); break;
    case 14: CN_EPIW(14); break;
    default: CN_EPIW(-1); break;
  }
#undef CN_EPIW
}

}  // namespace cn_gemm
