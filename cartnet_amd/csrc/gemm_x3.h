// Second-generation bf16x3 split-operand GEMM kernels (CartnetGemmArgs.precision == 1) for gfx950.
//
// The arithmetic is the one described in gemm_kernel.h ("bf16x3"): every fp32 operand element is split exactly into
// three bf16 pieces and each fp32 product is rebuilt from six v_mfma_f32_32x32x16_bf16 products accumulated in fp32.
// What changes is where the splitting work goes, because in the first kernel the vector ALU (splitting BOTH operand
// tiles every K-step in every workgroup) cost as many issue cycles as the matrix pipe had work:
//
//  * NN kernel (activations x weights, cn_gemm_x3nn_kernel): the weight operand arrives PRE-SPLIT
//    (cartnet_gemm_split_b, once per step per matrix -- weights are a few hundred KB) as an exact image of the LDS
//    tile, so its staging is a lane-linear direct-to-LDS copy (global_load_lds_dwordx4: no VGPRs, no VALU, no
//    ds_write); only the activation tile (one float4 per thread per K-step) is split in flight.
//  * TN kernel (weight gradients dW = dY^T X, both operands k-strided activations, cn_gemm_x3tn_kernel): the tiles
//    are staged as [k][m] bf16 planes -- the layout the fp32 rows convert to without a transpose -- and the MFMA
//    operands (8 consecutive k per lane) are fetched with the transposing LDS read ds_read_b64_tr_b16.
//
// Tile shape, wave grid, accumulator layout and therefore the epilogues are those of gemm_kernel.h.
#pragma once
#include "gemm_kernel.h"

namespace cn_gemm {

constexpr int X3_BN = 256;
constexpr int X3_A_PLANE = BM * 32;                       // [128 rows][16 k] bf16
constexpr int X3_A_BYTES = 3 * X3_A_PLANE;                // 12 KB
constexpr int X3_B_PLANE = X3_BN * 32;                    // [256 rows][16 k] bf16
constexpr int X3_B_BYTES = 3 * X3_B_PLANE;                // 24 KB per K-step per 256-column tile
constexpr int X3_BUF_BYTES = X3_A_BYTES + X3_B_BYTES;     // 36 KB; two buffers per workgroup, two workgroups per CU

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

template <int KIND_RT, class ACC>
__device__ __forceinline__ void x3_epilogue(const CartnetGemmArgs& p, ACC& acc, int g, int row0, int col0,
                                            int tile_m, int wm, int wn, int lane, int tid, float* smem) {
  const int kind = (p.gather_i[g] ? 1 : 0) | (p.resid[g] ? 2 : 0) | (p.dact[g] ? 4 : 0) |
                   (p.colsum[g] ? (p.colsq[g] ? 16 : 8) : 0) | (p.cpre[g] ? 32 : 0) | (p.out_act ? 64 : 0) |
                   (p.dact_kind ? 256 : 0);
#define CN_EPIW(K) epilogue_wide<X3_BN, K>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem, kind)
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

// K-loop of the precision-2 kernels ("ONE": plain bf16 operands, only the first of the three planes of every LDS
// buffer is in use).  The loop of the three-plane kernel gives a B tile's DMA one K-step to land and reads a step's
// fragments after that step's barrier; at precision 2 a K-step is four MFMAs per wave, so the step was a chain of
// latencies (barrier -> LDS reads -> MFMAs -> DMA wait: 240 ns hot, more from HBM) and a 156-workgroup, K = 256 product
// of configs[2] spent more time in sixteen of them than computing.  Here
//   * the two unused B planes of both buffers are ring slots (slot s = plane s/2 of buffer s%2): the B tile of K-step v
//     lives in slot v%4, is issued FOUR steps ahead and has landed two steps before it is multiplied;
//   * A tiles wait in four register sets, loaded four steps before the step that writes them to LDS (buffer v%2);
//   * the fragments of step u+1 are read during step u, into a second fragment register set, under step u's MFMAs --
//     after a barrier a wave starts its MFMA chain at once.
// Same LDS footprint, same products in the same order per accumulator (bit-identical results).  Memory operations retire
// in issue order; per step the order is B(u+4), A(u+6), so at the end of step u -- when B(u+2) and A(u+3) must be in --
// the ones that may still fly are A(u+4), B(u+3), A(u+5), B(u+4), A(u+6), each counted only if it exists.
#ifndef CN_ONE_DEEP
#define CN_ONE_DEEP 1
#endif
__device__ __forceinline__ void x3_wait_all_but(int n) {
  switch (n) {
    case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
  }
}
__device__ __forceinline__ unsigned x3_one_slot(int s) { return (s & 1) * X3_BUF_BYTES + (s >> 1) * X3_B_PLANE; }

// a_frag / b_frag: this lane's byte offset of its first A fragment inside an A buffer / first B fragment inside a B slot
// (x3_offset of the wave tile's first row; the second 32-row fragment is 1 KB further).
template <class RA, class AIssue, class AStore, class BIssue>
__device__ __forceinline__ void x3_one_deep_loop(const int n, const char* lds, const int a_frag, const int b_frag,
                                                 f32x16 (&acc)[2][2], AIssue&& a_issue, AStore&& a_store,
                                                 BIssue&& b_issue_slot) {
  using std::integral_constant;
  using std::true_type;
  using std::false_type;
  RA r0, r1, r2, r3;
  bf16x8 fa[2][2], fb[2][2];
  auto frags = [&](auto set_c, auto slot_c) {      // the fragments of the K-step whose B tile lives in ring slot SLOT
    constexpr int P = decltype(set_c)::value, SLOT = decltype(slot_c)::value;
    const char* qa = lds + (SLOT & 1) * X3_BUF_BYTES + a_frag;
    const char* qb = lds + x3_one_slot(SLOT) + X3_A_BYTES + b_frag;
    fa[P][0] = *reinterpret_cast<const bf16x8*>(qa);
    fa[P][1] = *reinterpret_cast<const bf16x8*>(qa + 1024);
    fb[P][0] = *reinterpret_cast<const bf16x8*>(qb);
    fb[P][1] = *reinterpret_cast<const bf16x8*>(qb + 1024);
  };
  auto mma = [&](auto set_c) {
    constexpr int P = decltype(set_c)::value;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int a = 0; a < 2; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[P][a], fb[P][b], acc[a][b], 0, 0, 0);
  };
  auto in_flight = [&](int u) {
    return (u + 4 < n ? 2 : 0) + (u + 3 < n ? 1 : 0) + (u + 5 < n ? 1 : 0) + (u + 6 < n ? 1 : 0);
  };
  // r holds the A tile of step u+2 on entry and receives the load of step u+6.  FULL: steady state (u + 6 < n), nothing
  // conditional and a constant wait (the scalar bookkeeping of the general step costs as much as the step's MFMAs).
  auto step = [&](auto u4_c, auto full_c, int u, RA& r) {
    constexpr int U4 = decltype(u4_c)::value;
    constexpr bool FULL = decltype(full_c)::value;
    constexpr int P = U4 & 1;
    mma(integral_constant<int, P>{});
    __builtin_amdgcn_sched_barrier(0);
    if (FULL || u + 2 < n) a_store(r, P);
    __builtin_amdgcn_sched_barrier(0);
    if (FULL || u + 4 < n) b_issue_slot(u + 4, U4);
    if (FULL || u + 6 < n) a_issue(r, u + 6);
    __builtin_amdgcn_sched_barrier(0);
    if (FULL || u + 1 < n) frags(integral_constant<int, P ^ 1>{}, integral_constant<int, (U4 + 1) & 3>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FULL) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    else x3_wait_all_but(in_flight(u));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  if (n <= 0) return;
  // head.  A product shorter than six K-steps re-reads its last tile into the registers / slots nobody will use, so that
  // ONE wait with a fixed count covers every length (a wait per length makes the compiler merge the register sets
  // behind it -- with copies of registers whose loads are still in flight).
  const int last = n - 1;
  a_issue(r0, 0);
  a_issue(r1, min(1, last));
  a_issue(r2, min(2, last));
  b_issue_slot(0, 0);
  b_issue_slot(min(1, last), 1);
  a_issue(r3, min(3, last));
  b_issue_slot(min(2, last), 2);
  b_issue_slot(min(3, last), 3);
  asm volatile("s_waitcnt vmcnt(3)" : "+v"(r0), "+v"(r1) :: "memory");     // A(0), A(1), A(2), B(0), B(1) have landed
  a_store(r0, 0);
  a_store(r1, 1);
  __builtin_amdgcn_sched_barrier(0);
  a_issue(r0, min(4, last));
  a_issue(r1, min(5, last));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  frags(integral_constant<int, 0>{}, integral_constant<int, 0>{});
  // step 0 overwrites A buffer 0 and ring slot 0: every wave must have these fragments first (in the loop a step's
  // fragments are read BEFORE the barrier that precedes the overwrite; here they can only be read after one)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  int u = 0;
  for (; u + 10 <= n; u += 4) {      // u + 3 + 6 < n: four steady-state steps
    step(integral_constant<int, 0>{}, true_type{}, u, r2);
    step(integral_constant<int, 1>{}, true_type{}, u + 1, r3);
    step(integral_constant<int, 2>{}, true_type{}, u + 2, r0);
    step(integral_constant<int, 3>{}, true_type{}, u + 3, r1);
  }
  for (; u < n; u += 4) {
    step(integral_constant<int, 0>{}, false_type{}, u, r2);
    if (u + 1 < n) step(integral_constant<int, 1>{}, false_type{}, u + 1, r3);
    if (u + 2 < n) step(integral_constant<int, 2>{}, false_type{}, u + 2, r0);
    if (u + 3 < n) step(integral_constant<int, 3>{}, false_type{}, u + 3, r1);
  }
}

// C[g] = epilogue(sum_s (silu?)(A[s]) @ B[s]), A fp32 [M, K] row-major (k-contiguous), B given pre-split
// (p.b_split[s]: image written by cartnet_gemm_split_b for the [K, N] operand).  Requirements (checked on the host):
// nsegs == 1, K % 16 == 0, N % 256 == 0, A rows 16-byte aligned and M * lda * 4 < 2^32, every epilogue operand
// 16-byte aligned.
//
// Pipeline (one barrier per 16-deep K-step, two LDS buffers, two workgroups per CU):
//   K-step u multiplies buffer u&1.  Each wave issues its A-fragment reads, and while they fly splits the A tile of
//   step u+1 (in registers since step u-2) into the other buffer, starts the DMA of the B tile of step u+1 into it and
//   the register load of the A tile of step u+3, then runs the 24 MFMAs.  Memory operations retire in issue order, so
//   "all but the youngest one" (s_waitcnt vmcnt(1)) before the barrier means: the DMA has landed, the A load issued
//   after it stays in flight across the barrier -- every A load (HBM) gets two K-steps to arrive, every DMA (L2) one.
//   The A loads are inline asm so that the compiler's own counter bookkeeping (which drains everything whenever a DMA
//   is outstanding) stays out of the loop; all waits in the loop are written by hand.
template <bool A_ACT, bool ONE>
__global__ __launch_bounds__(NTHREADS, 4) void cn_gemm_x3nn_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
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
  auto a_store = [&](f32x4 v, int buf) {
    if (A_ACT) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = fast_silu(v[c]);
    }
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
  auto b_issue_slot = [&](int v, int slot) {     // precision 2: ring slot of x3_one_deep_loop
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_b + x3_one_slot(slot)), "v"(b_voff), "s"(b_base(v) + wid * 1024) : "memory", "m0");
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
      a_store(r, CUR ^ 1);
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

  if constexpr (ONE && CN_ONE_DEEP) {
    CN_STAMP_BEGIN();
    x3_one_deep_loop<f32x4>(nsteps, lds, x3_offset(wm * S::WM + li, lh), x3_offset(wn * S::WN + li, lh), acc, a_issue, a_store,
                            b_issue_slot);
    CN_STAMP_END();
  } else if (nsteps > 0) {
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
    CN_STAMP_BEGIN();
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
    CN_STAMP_END();
  }
  x3_epilogue<0>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients: C[g] (+ split-K slabs) = A[g]^T @ (silu?)(B[g]), A fp32 [K, M] and B fp32 [K, N] row-major
// (k-strided), reduction over the K rows (edges / atoms).  128-row tiles of M (M % 4 == 0; a ragged last tile is
// computed whole and stored up to M), 256-column tiles of N.  LDS tile: 3 planes of [16 k][ROWS] bf16 (k-major, exactly how the fp32 rows convert), the 64-byte chunks of
// row k XOR-ed with k & 3 so that the four k-rows a transposing read touches fall on different banks.
template <int ROWS>
__device__ __forceinline__ int x3t_offset(int k, int col) {   // byte offset of element (k, col) in a plane
  const int byte = col * 2;
  return k * (ROWS * 2) + ((((byte >> 6) ^ (k & 3)) << 6) | (byte & 63));
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// 8 consecutive k (k0 .. k0+7) of column `col` as one MFMA operand register: two transposing reads of 4 k each.
// Lane 4q+p of a 16-lane group addresses row k0+q (k0+4+q), columns cbase + 4p .. 4p+3 of the group's 16 columns.
template <int ROWS>
__device__ __forceinline__ bf16x8 x3t_read(const char* plane, int k0, int cbase, int lane) {
  const int q = (lane >> 2) & 3, pp = lane & 3;
  const int col = cbase + pp * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(plane + x3t_offset<ROWS>(k0 + q, col)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(plane + x3t_offset<ROWS>(k0 + 4 + q, col)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

constexpr int X3T_A_PLANE = BK * BM * 2;        // [16 k][128 m] bf16 = 4 KB
constexpr int X3T_B_PLANE = BK * X3_BN * 2;     // [16 k][256 n] bf16 = 8 KB
constexpr int X3T_A_BYTES = 3 * X3T_A_PLANE;
constexpr int X3T_BUF_BYTES = 3 * (X3T_A_PLANE + X3T_B_PLANE);   // 36 KB

template <bool B_ACT, bool ONE>
__global__ __launch_bounds__(NTHREADS, 4) void cn_gemm_x3tn_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  using S = Shape<X3_BN>;
  __shared__ __attribute__((aligned(16))) float smem[2 * X3T_BUF_BYTES / 4];
  char* lds = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / S::WGN, wn = wid % S::WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_n = p.N / X3_BN;
  int bx, by;
  cn_splitk_block_map(bx, by);     // tiles of one K-chunk on one XCD (gemm_kernel.h)
  const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * X3_BN;
  const int g = blockIdx.z;
  const int split = fl.split0 + by;
  const int kbeg = fl.k_lo + by * fl.kchunk;
  const int kend = min(fl.k_hi, kbeg + fl.kchunk);
  const int nsteps = (kend - kbeg) / BK;          // whole K-steps only (the host gives the tail to the fp32 kernel)

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // staging units: (k, 4 consecutive columns).  A: 16 x 32 = 512 units (one per thread); B: 16 x 64 = 1024 (two).
  const float* __restrict__ Ab = p.A[g];
  const float* __restrict__ Bb = p.B[g];
  const int ak = tid >> 5, ac = (tid & 31) * 4;
  const int bk0 = tid >> 6, bc = (tid & 63) * 4;          // second unit: k + 8
  // ragged last row tile (M % 128 != 0, M % 4 == 0): columns past M re-read the last four valid ones; the output rows
  // they feed are never stored
  const float* a_ptr = Ab + (size_t)(kbeg + ak) * p.lda + min(row0 + ac, p.M - 4);
  const float* b_ptr = Bb + (size_t)(kbeg + bk0) * p.ldb + col0 + bc;
  const size_t a_step = (size_t)BK * p.lda, b_step = (size_t)BK * p.ldb, b_half = (size_t)8 * p.ldb;
  const int a_lds = x3t_offset<BM>(ak, ac);
  const int b_lds0 = X3T_A_BYTES + x3t_offset<X3_BN>(bk0, bc);
  const int b_lds1 = X3T_A_BYTES + x3t_offset<X3_BN>(bk0 + 8, bc);

  f32x4 ra, rb0, rb1;
  auto load = [&](int u) {
    ra = *reinterpret_cast<const f32x4*>(a_ptr + u * a_step);
    rb0 = *reinterpret_cast<const f32x4*>(b_ptr + u * b_step);
    rb1 = *reinterpret_cast<const f32x4*>(b_ptr + u * b_step + b_half);
  };
  auto put = [&](f32x4 v, char* dst, int plane_bytes, bool act) {
    if (act) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = fast_silu(v[c]);
    }
    if constexpr (ONE) {
      *reinterpret_cast<bf16x4*>(dst) = __builtin_convertvector(v, bf16x4);
    } else {
      const Split3 s = split3(v);
      *reinterpret_cast<bf16x4*>(dst) = s.h;
      *reinterpret_cast<bf16x4*>(dst + plane_bytes) = s.m;
      *reinterpret_cast<bf16x4*>(dst + 2 * plane_bytes) = s.l;
    }
  };
  auto store = [&](int buf) {
    char* base = lds + buf * X3T_BUF_BYTES;
    put(ra, base + a_lds, X3T_A_PLANE, false);
    put(rb0, base + b_lds0, X3T_B_PLANE, B_ACT);
    put(rb1, base + b_lds1, X3T_B_PLANE, B_ACT);
  };
  auto compute = [&](int buf) {
    const char* cA = lds + buf * X3T_BUF_BYTES;
    const char* cB = cA + X3T_A_BYTES;
    const int grp16 = ((lane >> 4) & 1) * 16;     // which 16 columns of the 32-wide MFMA tile this lane group holds
    bf16x8 ah[2], am[2], al[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int cb = wm * S::WM + a * 32 + grp16;
      ah[a] = x3t_read<BM>(cA, 8 * lh, cb, lane);
      if constexpr (!ONE) {
        am[a] = x3t_read<BM>(cA + X3T_A_PLANE, 8 * lh, cb, lane);
        al[a] = x3t_read<BM>(cA + 2 * X3T_A_PLANE, 8 * lh, cb, lane);
      }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int cb = wn * S::WN + b * 32 + grp16;
      const bf16x8 bh = x3t_read<X3_BN>(cB, 8 * lh, cb, lane);
      if constexpr (ONE) {
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh, acc[a][b], 0, 0, 0);
        continue;
      }
      const bf16x8 bm = x3t_read<X3_BN>(cB + X3T_B_PLANE, 8 * lh, cb, lane);
      const bf16x8 bl = x3t_read<X3_BN>(cB + 2 * X3T_B_PLANE, 8 * lh, cb, lane);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bm, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bh, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bm, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh, acc[a][b], 0, 0, 0);
      }
    }
  };

  if (nsteps > 0) {
    load(0);
    store(0);
    if (nsteps > 1) load(1);
    __syncthreads();
    int u = 0;
    for (; u < nsteps; ++u) {
      const int cur = u & 1;
      if (u + 1 < nsteps) store(cur ^ 1);
      if (u + 2 < nsteps) load(u + 2);
      compute(cur);
      __syncthreads();
    }
  }

  if (p.splitk > 1) {   // raw partial slab
    float* __restrict__ C = p.C[g] + (size_t)split * p.M * p.ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int grow = row0 + wm * S::WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (grow >= p.M) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int gcol = col0 + wn * S::WN + b * 32 + li;
          C[(size_t)grow * p.ldc + gcol] = acc[a][b][r];
        }
      }
    return;
  }
  x3_epilogue<0>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem);
}

}  // namespace cn_gemm
