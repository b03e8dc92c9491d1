// Second-generation fp32-MFMA kernel for activations x weights (CartnetGemmArgs.precision == 0 with a pre-packed
// weight operand): the memory pipeline of gemm_x3.h -- weight tile by direct-to-LDS DMA from an image packed once per
// step (cartnet_gemm_pack_b), activation tile through a two-deep register ring of inline-asm loads, hand-counted
// s_waitcnt + raw s_barrier, steady-state K-step placed by hand -- around the exact-fp32 v_mfma_f32_32x32x2_f32 chain
// of gemm_kernel.h.  Same tile (128 x 256), wave grid, accumulator layout and epilogues.
#pragma once
#include "gemm_kernel.h"

namespace cn_gemm {

constexpr int F32_BN = 256;
constexpr int F32_A_BYTES = BM * KPAD * 4;              // [128 rows][20 floats]: 10 KB (register-staged, padded rows)
// weight image (cartnet_gemm_pack_b): per 256-column tile and K-step [256 cols][16 k] fp32, rows of 64 bytes whose
// four 16-byte slots are XOR-ed with (col >> 2) & 3 -- unpadded, so that it can be DMA'd lane-linearly, and
// conflict-free for the ds_read_b128 fragment reads (16 consecutive rows hit 16 distinct (bank quadrant, slot) pairs)
constexpr int F32_B_BYTES = F32_BN * BK * 4;            // 16 KB per K-step per 256-column tile
constexpr int F32_BUF_BYTES = F32_A_BYTES + F32_B_BYTES;
__device__ __host__ __forceinline__ int f32_swz(int row, int quad) { return row * 64 + ((quad ^ ((row >> 2) & 3)) << 4); }

template <bool A_ACT>
__global__ __launch_bounds__(NTHREADS, 4) void cn_gemm_f32nn_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  using S = Shape<F32_BN>;
  static_assert(S::TM == 2 && S::TN == 2, "wave tile is 64 x 64");
  __shared__ __attribute__((aligned(16))) float smem[2 * F32_BUF_BYTES / 4];
  char* lds = reinterpret_cast<char*>(smem);

  CN_PHASE(0);
  CN_PHASE_ID();
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
  auto a_store = [&](f32x4 v, int buf) {
    if (A_ACT) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = fast_silu(v[c]);
    }
    *reinterpret_cast<f32x4*>(lds + buf * F32_BUF_BYTES + a_lds) = v;
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
      a_store(r, CUR ^ 1);
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
  // tiles in its shadow.  (Round 3, measured with in-kernel stamps -- tools/experiments/exp_phases.py, profiles/r03_exp_phases.md --
  // and NOT adopted: the barrier in the middle of the chain with the next step's first fragments read behind it, and
  // the two waves of a SIMD staging at different places of the chain.  Neither moves a launch.)
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
#ifdef CN_PHASE_STAMP
  CN_PHASE(3);                                             // last store issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CN_PHASE(4);                                             // ... and acknowledged
#endif
}

}  // namespace cn_gemm

namespace cn_gemm {

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients on the fp32 MFMA: C[g] (+ split-K slabs) = A[g]^T @ (silu?)(B[g]), A fp32 [K, M], B fp32 [K, N]
// row-major, reduction over the K rows.  Both operand tiles ([16 k][128 m] and [16 k][256 n] fp32) are exactly what
// the rows look like in memory, so BOTH are staged by direct-to-LDS DMA: no VGPR staging, no VALU, three LDS stages
// (24 KB each) so that every tile has two K-steps to land; one barrier per K-step:
//     s_waitcnt vmcnt(3)   this wave's three DMA pieces of step u have landed (step u+1's three stay in flight)
//     s_barrier            everybody's have, and everybody is done reading stage (u+2) % 3 (used by step u-1)
//     issue the DMA of step u+2 into that stage; read the fragments of step u; 32 MFMAs.
// SiLU on the B operand: in place in LDS, one K-step before the tile is published (B_ACT below).
// an LDS dword through a 32-bit address in ONE register + a constant the compiler folds into the instruction's offset field
__device__ __forceinline__ float cn_lds_ld1(unsigned addr) {
  return *reinterpret_cast<__attribute__((address_space(3))) float*>((unsigned long)addr);
}
constexpr int F32T_A_BYTES = BK * BM * 4;          // 8 KB
constexpr int F32T_B_BYTES = BK * F32_BN * 4;      // 16 KB
constexpr int F32T_STAGE = F32T_A_BYTES + F32T_B_BYTES;
// Four stages = 96 KB of LDS: at most ONE of these workgroups per CU, on purpose (model.hip, split_k: the launches share
// every CU with the kernels of the main stream for the whole of backward; 52 KB stay free for one of theirs).
#ifndef CN_TN_STAGES
#define CN_TN_STAGES 4
#endif
constexpr int F32T_NSTAGE = CN_TN_STAGES;

#ifdef CN_TN_STAMP
// Diagnostic build (tools/exp_tn_stamps.py): per workgroup [main-loop shader cycles, 100 MHz ticks, K-steps, 0] and per wave
// [cycles in the counted wait, cycles in the barrier]; nothing else reads the buffers.
static __device__ unsigned long long cn_tn_dbg[1024 * 4];
static __device__ unsigned long long cn_tn_dbg_wave[1024 * 8 * 2];
#endif

// B_ACT (NST = 5): SiLU on the B operand IN PLACE in LDS -- every thread activates the 32 bytes of a B tile that its own two
// DMA pieces brought (no other wave's data: its own vmcnt is the only dependency), two K-steps after their issue and one
// barrier before the tile is published; each element once (the fragment-side form of round 2 activated it in both waves
// that read it: 15 % slower).  The fifth stage keeps two K-steps between a tile's issue and its activation.
template <bool B_ACT, int NST = F32T_NSTAGE>
__global__ __launch_bounds__(NTHREADS, 4) void cn_gemm_f32tn_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  using S = Shape<F32_BN>;
  __shared__ __attribute__((aligned(16))) float smem[NST * F32T_STAGE / 4];
  char* lds = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / S::WGN, wn = wid % S::WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_n = p.N / F32_BN;
  int bx, by;
  cn_splitk_block_map(bx, by);     // tiles of one K-chunk on one XCD (gemm_kernel.h)
  const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * F32_BN;
  const int g = blockIdx.z;
  const int split = fl.split0 + by;
  const int kbeg = fl.k_lo + by * fl.kchunk;
  const int kend = min(fl.k_hi, kbeg + fl.kchunk);
  const int nsteps = (kend - kbeg) / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // DMA pieces of 1 KB: wave w moves A rows 2w, 2w+1 (one piece) and B rows w, w+8 (two pieces)
  const float* a_base = p.A[g] + (size_t)(kbeg + 2 * wid) * p.lda + row0;
  const float* b_base = p.B[g] + (size_t)(kbeg + wid) * p.ldb + col0;
  // ragged last row tile (M % 128 != 0, M % 4 == 0): lanes past M re-read the last four valid columns of their k-row; the
  // output rows they feed are never stored
  const int a_col = min(row0 + (lane & 31) * 4, p.M - 4) - row0;
  const unsigned a_voff = ((unsigned)(lane >> 5) * (unsigned)p.lda + (unsigned)a_col) * 4u;
  const unsigned b_voff = lane * 16;
  const size_t a_step = (size_t)BK * p.lda, b_step = (size_t)BK * p.ldb, b_half = (size_t)8 * p.ldb;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  auto issue = [&](int v) {
    const unsigned st = lds0 + (v % NST) * F32T_STAGE;
    const float* sa = a_base + v * a_step;
    const float* sb = b_base + v * b_step;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(st + wid * 1024), "v"(a_voff), "s"(sa) : "memory", "m0");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(st + F32T_A_BYTES + wid * 1024), "v"(b_voff), "s"(sb) : "memory", "m0");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(st + F32T_A_BYTES + (wid + 8) * 1024), "v"(b_voff), "s"(sb + b_half) : "memory", "m0");
  };

  // fragments of one 8-deep k-group: element j of the MFMA chain takes k = kg*8 + 2j + lh; one ds_read2_b32 fetches the
  // two 32-row (A) / 32-column (B) halves of a wave tile for one k
  float af[2][2][4], bf[2][2][4];      // [register set = k-group][tile][j]
  auto frags = [&](int u, int kg) {
    const float* sA = reinterpret_cast<const float*>(lds + (u % NST) * F32T_STAGE);
    const float* sB = sA + F32T_A_BYTES / 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kg * 8 + 2 * j + lh;
#pragma unroll
      for (int a = 0; a < 2; ++a) af[kg][a][j] = sA[k * BM + wm * S::WM + a * 32 + li];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        bf[kg][b][j] = sB[k * F32_BN + wn * S::WN + b * 32 + li];
      }
    }
  };
  auto mma16 = [&](int kg) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg][a][j], bf[kg][b][j], acc[a][b], 0, 0, 0);
  };

  // B_ACT: this thread's own 2 x 16 bytes of B tile v (pieces wid and wid + 8 of its wave's DMAs), activated in place
  auto silu_stage = [&](int stage) {
    char* q = lds + stage * F32T_STAGE + F32T_A_BYTES + wid * 1024 + lane * 16;
    f32x4 x0 = *reinterpret_cast<f32x4*>(q), x1 = *reinterpret_cast<f32x4*>(q + 8192);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      x0[c] = fast_silu(x0[c]);
      x1[c] = fast_silu(x1[c]);
    }
    *reinterpret_cast<f32x4*>(q) = x0;
    *reinterpret_cast<f32x4*>(q + 8192) = x1;
  };

  // Software-pipelined around ONE barrier per K-step, placed between the two k-groups of a tile (round 4; the loop the
  // compiler made of the plain form kept four fragment registers and waited for LDS -- s_waitcnt lgkmcnt(0) -- in front
  // of every fourth MFMA: 0.78-0.80 of the matrix pipe isolated).  Iteration u (there is a tile u + 1) starts with
  // k-group 0 of tile u issued to the matrix pipe and k-group 1 in registers:  (1) this wave's three DMA pieces of tile
  // u + 1 have landed (vmcnt(3): tile u + 2's stay in flight) and its own LDS reads are done; barrier: everybody's have,
  // so tile u + 1 is published AND the stage of tile u is free.  (2) k-group 0 of tile u + 1 -> the registers the MFMAs
  // before the barrier consumed; tile u + 3 is DMA'd into tile u's stage (three stages, two and a half K-steps to
  // land).  (3) 16 MFMAs of k-group 1 of tile u cover the latency of (2).  (4) k-group 1 of tile u + 1, covered by
  // (5) the 16 MFMAs of k-group 0 of tile u + 1.
  if (nsteps > 0) {
    constexpr int PF = NST - 1;          // tiles in flight ahead of the one being multiplied
    static_assert(NST >= 4 && NST <= 6, "four to six stages");
#pragma unroll
    for (int v = 0; v < PF; ++v)
      if (v < nsteps) issue(v);
    if constexpr (B_ACT) {      // tiles 0 and 1 of this wave have landed: activate them (tile t >= 2: in iteration t - 2)
      static_assert(!B_ACT || NST >= 5, "two K-steps between a tile's issue and its activation");
      if (nsteps >= PF) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (PF - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      silu_stage(0);
      if (nsteps > 1) silu_stage(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      if (nsteps >= PF) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (PF - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    frags(0, 0);
    frags(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma16(0);
#ifdef CN_TN_STAMP
    unsigned long long st_w = 0, st_b = 0;
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Steady state unrolled by four (round 6): with the stage index a compile-time constant every fragment read is ONE
    // ds_read2st64_b32 off one of eight base registers (offsets in units of 256 B: k and k + 2 of one 32-row block) and every
    // DMA destination an immediate -- the rolled loop below recomputed 24 LDS addresses in vector instructions per K-step (on
    // an fp32 MFMA loop they are not free: 4,695 cycles per K-step measured, tools/exp_tn_stamps.py) and took five scalar
    // branches.  It keeps the head, the tail and K ranges too short for this one.
    constexpr int NHI = (NST + 1) / 2;
    unsigned fbA[2][NHI], fbB[2][NHI];   // [block][pair of stages]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int hi = 0; hi < NHI; ++hi) {
        fbA[x][hi] = lds0 + hi * 2 * F32T_STAGE + lh * (BM * 4) + (wm * S::WM + x * 32 + li) * 4;
        fbB[x][hi] = lds0 + hi * 2 * F32T_STAGE + F32T_A_BYTES + lh * (F32_BN * 4) + (wn * S::WN + x * 32 + li) * 4;
        asm volatile("" : "+v"(fbA[x][hi]), "+v"(fbB[x][hi]));
      }
    auto frags_c = [&](auto st_c, auto kg_c) {
      constexpr int ST = decltype(st_c)::value, KG = decltype(kg_c)::value;
      constexpr int OST = (ST & 1) * F32T_STAGE;
#pragma unroll
      for (int j = 0; j < 4; ++j)               // k = KG*8 + 2j + lh (lh is in the base register)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          af[KG][x][j] = cn_lds_ld1(fbA[x][ST >> 1] + (OST + (KG * 8 + 2 * j) * (BM * 4)));
          bf[KG][x][j] = cn_lds_ld1(fbB[x][ST >> 1] + (OST + (KG * 8 + 2 * j) * (F32_BN * 4)));
        }
    };
    // DMA of tile v into stage ST: the three pieces of issue() with immediate LDS destinations
    auto issue_c = [&](int v, auto st_c) {
      constexpr int ST = decltype(st_c)::value;
      const float* sa = a_base + v * a_step;
      const float* sb = b_base + v * b_step;
      const float* sb2 = sb + b_half;
      const unsigned lw = lds0 + wid * 1024, va = a_voff, vb = b_voff;    // (asm operands alone do not capture in a generic lambda)
      asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(lw), "v"(va), "s"(sa), "n"(ST * F32T_STAGE) : "memory", "m0", "scc");
      asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(lw), "v"(vb), "s"(sb), "n"(ST * F32T_STAGE + F32T_A_BYTES) : "memory", "m0", "scc");
      asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(lw), "v"(vb), "s"(sb2), "n"(ST * F32T_STAGE + F32T_A_BYTES + 8192) : "memory", "m0", "scc");
    };
    // one steady-state iteration (tile u -> u + 1; there are tiles u + 2 and u + 3): the rolled loop's body with constants
    auto iter_c = [&](int u, auto c_c) {
      constexpr int Cc = decltype(c_c)::value;           // u % NSTAGE
      __builtin_amdgcn_sched_barrier(0);
#ifdef CN_TN_STAMP
      const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#endif
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(3 * (PF - 2)) : "memory");    // tile u + 1 landed; u + 2 .. stay in flight
#ifdef CN_TN_STAMP
      const unsigned long long st_t1 = __builtin_amdgcn_s_memtime();
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef CN_TN_STAMP
      const unsigned long long st_t2 = __builtin_amdgcn_s_memtime();
      st_w += st_t1 - st_t0;
      st_b += st_t2 - st_t1;
#endif
      frags_c(std::integral_constant<int, (Cc + 1) % NST>{}, std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      issue_c(u + PF, std::integral_constant<int, (Cc + PF) % NST>{});
      __builtin_amdgcn_sched_barrier(0);
      mma16(1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (B_ACT) {      // tiles u + 3 .. u + PF stay in flight: this wave's pieces of tile u + 2 have landed
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (PF - 2)) : "memory");
        silu_stage((Cc + 2) % NST);
        __builtin_amdgcn_sched_barrier(0);
      }
      frags_c(std::integral_constant<int, (Cc + 1) % NST>{}, std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      mma16(0);
    };
    int u = 0;
#ifndef CN_TN_ROLLED
    for (; u + 2 * NST - 1 <= nsteps; u += NST) {          // every (u + c) + PF < nsteps
      iter_c(u, std::integral_constant<int, 0>{});
      iter_c(u + 1, std::integral_constant<int, 1>{});
      iter_c(u + 2, std::integral_constant<int, 2>{});
      iter_c(u + 3, std::integral_constant<int, 3>{});
      if constexpr (NST > 4) iter_c(u + 4, std::integral_constant<int, 4>{});
      if constexpr (NST > 5) iter_c(u + 5, std::integral_constant<int, 5>{});
    }
#endif
    for (; u + 1 < nsteps; ++u) {
      __builtin_amdgcn_sched_barrier(0);
#ifdef CN_TN_STAMP
      const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#endif
      if (u + PF <= nsteps) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(3 * (PF - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef CN_TN_STAMP
      const unsigned long long st_t1 = __builtin_amdgcn_s_memtime();
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef CN_TN_STAMP
      const unsigned long long st_t2 = __builtin_amdgcn_s_memtime();
      st_w += st_t1 - st_t0;
      st_b += st_t2 - st_t1;
#endif
      frags(u + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + PF < nsteps) issue(u + PF);
      __builtin_amdgcn_sched_barrier(0);
      mma16(1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (B_ACT) {
        if (u + 2 < nsteps) {
          if (u + PF < nsteps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (PF - 2)) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          silu_stage((u + 2) % NST);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      frags(u + 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma16(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma16(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // the epilogue reuses the LDS
    asm volatile("" ::: "memory");
#ifdef CN_TN_STAMP
    {
      const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      if (lin < 1024 && lane == 0) {
        cn_tn_dbg_wave[(lin * 8 + wid) * 2] = st_w;
        cn_tn_dbg_wave[(lin * 8 + wid) * 2 + 1] = st_b;
        if (wid == 0) {
          cn_tn_dbg[lin * 4] = __builtin_amdgcn_s_memtime() - st_c0;
          cn_tn_dbg[lin * 4 + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
          cn_tn_dbg[lin * 4 + 2] = (unsigned long long)nsteps;
        }
      }
    }
#endif
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
  const int kind = (p.gather_i[g] ? 1 : 0) | (p.resid[g] ? 2 : 0) | (p.dact[g] ? 4 : 0) |
                   (p.colsum[g] ? (p.colsq[g] ? 16 : 8) : 0) | (p.cpre[g] ? 32 : 0) | (p.out_act ? 64 : 0) |
                   (p.dact_kind ? 256 : 0);
  epilogue_wide<F32_BN, -1>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem, kind);
}

}  // namespace cn_gemm
