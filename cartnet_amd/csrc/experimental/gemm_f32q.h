// EXPERIMENT (round 4), NOT part of libcartnet_hip.so: measured slower than gemm_f32.h on every form, alone and inside the
// training step (profiles/r04_exp_phases.md has the numbers and the in-kernel stamps).  Built only with
// CARTNET_BUILD_EXPERIMENTAL=1 (cartnet_amd/build.py adds this directory's sources and -DCN_EXPERIMENTAL_Q; the switch
// is then cartnet_gemm_experimental_q(1), tools/experiments/ab_q.sh).  Kept as the record of what "more resident workgroups" buys.
//
// fp32-MFMA kernel for activations x weights with FOUR small workgroups per CU instead of two large ones.
//
// What the in-kernel stamps of round 3 showed about gemm_f32.h (profiles/r03_exp_phases.md): its two 512-thread
// workgroups per CU alternate -- one multiplies while the other stores, retires, is replaced and prefetches -- and
// everything outside the main loop (~31 us at K = 256) only just fits under the partner's 30.7 us loop, so the matrix
// pipe idles at every hand-over (9 % of the launch) and runs at 0.91 next to the partner's epilogue.  Demand on the pipe
// = 2 x 30.7 / (30.7 + 31) = 0.995: no slack.  This kernel cuts the same work into 128 x 128 tiles owned by 256-thread
// workgroups (4 waves in a 2 x 2 grid, one per SIMD, the same 64 x 64 wave tile and accumulator layout), four of which
// share a CU: a workgroup's lone main loop needs the pipe for ~14 us of a ~45 us life, so demand is ~1.25 and two or more
// loops overlap most of the time.  Registers per wave are unchanged (<= 128), LDS is 40 KB per workgroup (a three-deep
// ring for the activation tile, whose rows come from HBM, two stages for the L2-resident weight tile).
//
// Both operand tiles arrive by direct-to-LDS DMA (global_load_lds_dwordx4): the weight tile from the
// cartnet_gemm_pack_b image (a 128-column tile is one 8 KB half of its [256][16] block), the activation tile [128][16]
// straight from its rows -- every lane fetches one 16-byte k-quad of one row, and the XOR swizzle of the LDS image
// (f32_swz) is applied on the SOURCE side (the lane that fills slot s of a row fetches k-quad s ^ ((row >> 2) & 3)), so
// the image is conflict-free for the ds_read_b128 fragment reads without a register round trip.  No staging VGPRs, no
// VALU in the main loop.  SiLU on the A operand (layer GEMM 2) is applied in LDS by the lane whose DMA delivered the
// bytes, between the wait for the landing and the barrier that publishes the stage.
#pragma once
#include "../gemm_f32.h"

namespace cn_gemm {

constexpr int Q_NT = 256;
constexpr int Q_BN = 128;
constexpr int Q_A_BYTES = BM * BK * 4;                  // [128 rows][16 k] fp32, 64-byte rows, swizzled slots: 8 KB
constexpr int Q_B_BYTES = Q_BN * BK * 4;                // 8 KB per K-step per 128-column tile
constexpr int Q_A_STAGES = 3;                            // activation rows come from HBM: two K-steps to land
constexpr int Q_B_STAGES = 2;                            // the weight image is L2-resident: one K-step
constexpr int Q_LDS_BYTES = Q_A_STAGES * Q_A_BYTES + Q_B_STAGES * Q_B_BYTES;      // 40 KB: four workgroups = 160 KB

#ifdef CN_PHASE_STAMP
static __device__ unsigned long long cn_phase_dbg2[8192 * 4];     // wave 0 of every workgroup: shader cycles by loop phase
#endif

struct ShapeQ {                                         // 4 waves, 2 x 2, each 64 x 64
  static constexpr int WGM = 2, WGN = 2, WM = 64, WN = 64, TM = 2, TN = 2;
};

template <bool A_ACT>
__global__ __launch_bounds__(Q_NT, 4) void cn_gemm_f32nnq_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  using S = ShapeQ;
  static_assert(Q_LDS_BYTES / 4 >= (Q_NT / 64) * SCR_FLOATS, "epilogue scratch must fit");
  static_assert(Q_LDS_BYTES >= 2 * S::WGM * Q_BN * 8, "statistics scratch must fit");
  __shared__ __attribute__((aligned(16))) float smem[Q_LDS_BYTES / 4];
  char* lds = reinterpret_cast<char*>(smem);

  CN_PHASE(0);
  CN_PHASE_ID();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_n = p.N / Q_BN;
  int bx, g;
  cn_block_map(bx, g, tiles_n);
  const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * Q_BN;
  const int nsteps = p.K / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // A tile: 8 pieces of 1 KB = 16 rows each; wave w moves pieces w and w + 4 (rows 16 w + lane / 4 and 64 more).  The
  // lane that fills slot (lane & 3) of its row fetches the k-quad the swizzle keeps there (rows past M are clamped; the
  // epilogue drops them).
  const int rloc = lane >> 2;
  const int kq = (lane & 3) ^ ((rloc >> 2) & 3);
  const int r1 = wid * 16 + rloc;
  const unsigned a_voff1 = ((unsigned)min(row0 + r1, p.M - 1) * (unsigned)p.lda + kq * 4) * 4u;        // bytes
  const unsigned a_voff2 = ((unsigned)min(row0 + r1 + 64, p.M - 1) * (unsigned)p.lda + kq * 4) * 4u;
  const unsigned b_voff = lane * 16;
  const float* a0 = p.A[g];
  // image: per 256-column tile and K-step a [256][16] block of 16 KB; this tile is its half (tile_n & 1)
  const char* b0 = reinterpret_cast<const char*>(p.b_split[g]) + (size_t)(tile_n >> 1) * nsteps * F32_B_BYTES +
                   (size_t)(tile_n & 1) * Q_B_BYTES + wid * 1024;
  const unsigned lds_w = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + wid * 1024;

  constexpr int LDS_B0 = Q_A_STAGES * Q_A_BYTES;
  auto issue_a = [&](int v) {
    const float* asrc = a0 + v * BK;
    const unsigned dst = lds_w + (v % Q_A_STAGES) * Q_A_BYTES;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(dst), "v"(a_voff1), "s"(asrc) : "memory", "m0");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(dst + 4096), "v"(a_voff2), "s"(asrc) : "memory", "m0");
  };
  auto issue_b = [&](int v) {
    const char* bsrc = b0 + (size_t)v * F32_B_BYTES;
    const unsigned dst = lds_w + LDS_B0 + (v & 1) * Q_B_BYTES;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(dst), "v"(b_voff), "s"(bsrc) : "memory", "m0");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(dst + 4096), "v"(b_voff), "s"(bsrc + 4096) : "memory", "m0");
  };
  // SiLU on the A operand: every lane activates, in place, the two 16-byte slots its own DMA filled (wait for them first)
  auto act_own = [&](int v) {
    if (A_ACT) {
      char* q = lds + (v % Q_A_STAGES) * Q_A_BYTES + wid * 1024 + lane * 16;
      f32x4 v0 = *reinterpret_cast<const f32x4*>(q), v1 = *reinterpret_cast<const f32x4*>(q + 4096);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v0[c] = fast_silu(v0[c]);
        v1[c] = fast_silu(v1[c]);
      }
      *reinterpret_cast<f32x4*>(q) = v0;
      *reinterpret_cast<f32x4*>(q + 4096) = v1;
    }
  };
  f32x4 af[2][2], bf[2][2];     // [k-group of 8][tile]: k = kg*8 + lh*4 + j for element j
  // per-lane fragment addresses inside a stage (the stage base is wave-uniform and changes per K-step)
  int a_fo[2][2], b_fo[2][2];
#pragma unroll
  for (int kg = 0; kg < 2; ++kg)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a_fo[kg][t] = f32_swz(wm * S::WM + t * 32 + li, kg * 2 + lh);
      b_fo[kg][t] = LDS_B0 + f32_swz(wn * S::WN + t * 32 + li, kg * 2 + lh);
    }
  auto frags = [&](int u, int kg) {
    const char* sA = lds + (u % Q_A_STAGES) * Q_A_BYTES;
    const char* sB = lds + (u & 1) * Q_B_BYTES;
#pragma unroll
    for (int a = 0; a < 2; ++a) af[kg][a] = *reinterpret_cast<const f32x4*>(sA + a_fo[kg][a]);
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[kg][b] = *reinterpret_cast<const f32x4*>(sB + b_fo[kg][b]);
  };
  auto mma16 = [&](int kg) {
#ifdef CQ_NO_MMA       /* diagnostic: the memory pipeline alone (one MFMA per k-group keeps the fragments alive) */
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg][0][0] + af[kg][1][1], bf[kg][0][0] + bf[kg][1][1], acc[0][0], 0, 0, 0);
    return;
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg][a][j], bf[kg][b][j], acc[a][b], 0, 0, 0);
  };

  if (nsteps > 0) {
    // pipeline head: tiles 0 and 1 of both operands and A(2); tile 0 is needed first
    issue_a(0);
    issue_b(0);
    if (nsteps > 1) {
      issue_a(1);
      issue_b(1);
    }
    if (nsteps > 2) issue_a(2);
    if (nsteps > 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (nsteps > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (A_ACT) {
      act_own(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    frags(0, 0);
    frags(0, 1);
    CN_STAMP_BEGIN();
    CN_PHASE(1);
#ifdef CN_PHASE_STAMP
    unsigned long long cq_mma = 0, cq_wait = 0, cq_bar = 0;      // wave 0's cycles: issue of a step, DMA wait, barrier
#define CQ_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define CQ_T(x)
#endif
    // Software-pipelined around ONE barrier per K-step, placed between the two k-groups of a tile.  Loop iteration u
    // (there is a tile u + 1) starts with k-group 0 of tile u already issued to the matrix pipe and k-group 1 of tile u
    // in registers:  (1) this wave's share of A(u + 1) / B(u + 1) has landed (vmcnt: all but the youngest two,
    // A(u + 2)); barrier: everybody's has, and everybody's fragment reads of tile u are done.  (2) k-group 0 of tile
    // u + 1 -> the registers the MFMAs before the barrier have consumed; B(u + 2) and A(u + 3) are DMA'd into the stages
    // tile u has just freed.  (3) 16 MFMAs of k-group 1 of tile u, covering the latency of (2).  (4) k-group 1 of tile
    // u + 1, covered by (5) the 16 MFMAs of k-group 0 of tile u + 1.  Neither LDS latency nor DMA issue time is exposed;
    // the loop's back edge sits where no LDS read is young (the compiler drains lgkmcnt there).
    CQ_T(cqs);
    mma16(0);
    for (int u = 0; u + 1 < nsteps; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      CQ_T(cq1);
#ifndef CQ_NO_DMA
      if (u + 2 < nsteps) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      if (A_ACT) {                          // this wave's share of A(u + 1) has landed: activate it in place
        act_own(u + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      CQ_T(cq2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      CQ_T(cq3);
      frags(u + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
#ifndef CQ_NO_DMA      /* diagnostic: the matrix pipeline alone (stale tiles) */
      if (u + 2 < nsteps) issue_b(u + 2);
      if (u + 3 < nsteps) issue_a(u + 3);
#elif CQ_NO_DMA == 1   /* ... with the weight DMA only */
      if (u + 2 < nsteps) issue_b(u + 2);
#elif CQ_NO_DMA == 2   /* ... with the activation DMA only */
      if (u + 3 < nsteps) issue_a(u + 3);
#endif
      __builtin_amdgcn_sched_barrier(0);
      mma16(1);
      __builtin_amdgcn_sched_barrier(0);
      frags(u + 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma16(0);
#ifdef CN_PHASE_STAMP
      cq_wait += cq2 - cq1;
      cq_bar += cq3 - cq2;
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    mma16(1);
#ifdef CN_PHASE_STAMP
    cq_mma = __builtin_amdgcn_s_memtime() - cqs;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // the epilogue reuses the LDS
    asm volatile("" ::: "memory");
    CN_STAMP_END();
    CN_PHASE(2);
#ifdef CN_PHASE_STAMP
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) {
      const unsigned cn_lin = blockIdx.x + gridDim.x * blockIdx.z;
      if (cn_lin < 8192) {
        cn_phase_dbg2[cn_lin * 4 + 0] = cq_mma;
        cn_phase_dbg2[cn_lin * 4 + 1] = cq_wait;
        cn_phase_dbg2[cn_lin * 4 + 2] = cq_bar;
      }
    }
#endif
  }
  const int kind = (p.gather_i[g] ? 1 : 0) | (p.resid[g] ? 2 : 0) | (p.dact[g] ? 4 : 0) |
                   (p.colsum[g] ? (p.colsq[g] ? 16 : 8) : 0) | (p.cpre[g] ? 32 : 0) | (p.out_act ? 64 : 0);
#define CN_EPIW(K) epilogue_wide_s<ShapeQ, Q_BN, Q_NT, K>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem, kind)
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
#ifdef CN_PHASE_STAMP
  CN_PHASE(3);                                             // last store issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CN_PHASE(4);                                             // ... and acknowledged
#endif
}

}  // namespace cn_gemm
