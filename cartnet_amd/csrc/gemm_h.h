// Half-storage forms of the precision-2 kernels (CartnetGemmArgs.a_half / b_half / c_half / dact_half): the plain-bf16
// kernels of gemm_x3.h (one MFMA product per block, fp32 accumulate) with operands that ALREADY live in memory as bf16
// and / or an output written as bf16 -- SURVEY.md 8d "config 3: bf16 storage / fp32 accumulate".  A bf16 operand is
// loaded as 8 bytes per thread per K-step instead of 16 and, unless a SiLU sits on it, goes to LDS unconverted.
// Own translation unit (gemm_h.hip): the kernels of the other precisions are not touched.
#pragma once
#include <type_traits>
#include "gemm_x3.h"

namespace cn_gemm {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// epilogue_wide (gemm_kernel.h) with the output stored as bf16 (C_H) and / or the silu' operand read as bf16 (D_H); the
// column sums, residual, node-term gathers, bias and cpre stay fp32.
template <int BN, int KIND, bool C_H, bool D_H>
__device__ __forceinline__ void epilogue_wide_h(const CartnetGemmArgs& p, f32x16 (&acc)[Shape<BN>::TM][Shape<BN>::TN],
                                              int g, int row0, int col0, int tile_m, int wm, int wn, int lane, int tid,
                                              float* smem, int rt_kind) {
  using S = Shape<BN>;
  const int kind = (KIND >= 0) ? KIND : rt_kind;
  const bool GATHER = kind & 1, RESID = kind & 2, DACT = kind & 4, SUM1 = kind & 8, SUM2 = kind & 16,
             CPRE = kind & 32, OUTACT = kind & 64;
  const int li = lane & 31, lh = lane >> 5;
  const int c4 = lane & 7, rsub = lane >> 3;
  float* C = p.C[g];
  const float* __restrict__ bias = p.bias[g];
  const float* __restrict__ gi = p.gather_i[g];
  const float* __restrict__ gj = p.gather_j[g];
  const float* resid = p.resid[g];
  const float* dact = p.dact[g];
  float* cpre = p.cpre[g];
  float* scr = smem + (tid >> 6) * SCR_FLOATS;

  f32x4 bias4[S::TN], sum4[S::TN];
  double cs[S::TN], cq[S::TN];
#pragma unroll
  for (int b = 0; b < S::TN; ++b) {
    const int gcol = col0 + wn * S::WN + b * 32 + c4 * 4;
    bias4[b] = bias ? ldv4(bias + gcol) : f32x4{0.f, 0.f, 0.f, 0.f};
    sum4[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    cs[b] = 0.0;
    cq[b] = 0.0;
  }
  if (SUM2) {   // BatchNorm statistics of v = acc + bias, taken in the accumulator layout (column = lane)
#pragma unroll
    for (int b = 0; b < S::TN; ++b) {
      const float bv = bias ? bias[col0 + wn * S::WN + b * 32 + li] : 0.f;
#pragma unroll
      for (int a = 0; a < S::TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int grow = row0 + wm * S::WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (grow < p.M) {
            const double v = (double)(acc[a][b][r] + bv);
            cs[b] += v;
            cq[b] += v * v;
          }
        }
    }
  }
#pragma unroll
  for (int a = 0; a < S::TM; ++a) {
    int grow[4], ti[4], sj[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      grow[i] = row0 + wm * S::WM + a * 32 + rsub + 8 * i;
      ti[i] = 0;
      sj[i] = 0;
      if (GATHER && grow[i] < p.M) {
        ti[i] = p.tgt[grow[i]];
        sj[i] = p.src[grow[i]];
      }
    }
#pragma unroll
    for (int b = 0; b < S::TN; ++b) {
      const int gcol = col0 + wn * S::WN + b * 32 + c4 * 4;
#pragma unroll
      for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * SCR_LD + li] = acc[a][b][r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = ldv4(scr + (rsub + 8 * i) * SCR_LD + c4 * 4);
        if (grow[i] >= p.M) continue;
        v += bias4[b];
        if (GATHER) v += ldv4(gi + (size_t)ti[i] * p.ldg + gcol) + ldv4(gj + (size_t)sj[i] * p.ldg + gcol);
        if (RESID) v += ldv4(resid + (size_t)grow[i] * p.ldr + gcol);
        if (DACT) {
          f32x4 d;
          if constexpr (D_H)
            d = __builtin_convertvector(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(dact) +
                                                                         (size_t)grow[i] * p.ldd + gcol), f32x4);
          else
            d = ldv4(dact + (size_t)grow[i] * p.ldd + gcol);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] *= fast_dsilu(d[q]);
        }
        if (SUM1) sum4[b] += v;
        if (CPRE) stv4(cpre + (size_t)grow[i] * p.ldc + gcol, v);
        if (OUTACT) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fast_silu(v[q]);
        }
        if constexpr (C_H)
          *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(C) + (size_t)grow[i] * p.ldc + gcol) =
              __builtin_convertvector(v, bf16x4);
        else
          stv4(C + (size_t)grow[i] * p.ldc + gcol, v);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (SUM1 || SUM2) {
    double* red = reinterpret_cast<double*>(smem);   // [2][WGM][BN]
    double* __restrict__ colsum = p.colsum[g];
    double* __restrict__ colsq = p.colsq[g];
    __syncthreads();   // every wave is done with its transpose scratch
#pragma unroll
    for (int b = 0; b < S::TN; ++b) {
      if (SUM2) {
        double s = cs[b], q = cq[b];
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        if (lh == 0) {
          red[(0 * S::WGM + wm) * BN + wn * S::WN + b * 32 + li] = s;
          red[(1 * S::WGM + wm) * BN + wn * S::WN + b * 32 + li] = q;
        }
      } else {
        f32x4 s = sum4[b];     // rows of this wave's tile are spread over the 8 lanes that share c4
#pragma unroll
        for (int o = 8; o <= 32; o <<= 1)
#pragma unroll
          for (int q = 0; q < 4; ++q) s[q] += __shfl_xor(s[q], o);
        if (rsub == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) red[(0 * S::WGM + wm) * BN + wn * S::WN + b * 32 + c4 * 4 + q] = (double)s[q];
        }
      }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += NTHREADS) {
      const int gcol = col0 + c;
      if (gcol < p.N) {
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int w = 0; w < S::WGM; ++w) {
          s += red[(0 * S::WGM + w) * BN + c];
          if (SUM2) q += red[(1 * S::WGM + w) * BN + c];
        }
        colsum[(size_t)tile_m * p.N + gcol] = s;
        if (SUM2) colsq[(size_t)tile_m * p.N + gcol] = q;
      }
    }
  }
}


template <bool C_H, bool D_H>
__device__ __forceinline__ void h_epilogue(const CartnetGemmArgs& p, f32x16 (&acc)[2][2], int g, int row0, int col0,
                                           int tile_m, int wm, int wn, int lane, int tid, float* smem) {
  const int kind = (p.gather_i[g] ? 1 : 0) | (p.resid[g] ? 2 : 0) | (p.dact[g] ? 4 : 0) |
                   (p.colsum[g] ? (p.colsq[g] ? 16 : 8) : 0) | (p.cpre[g] ? 32 : 0) | (p.out_act ? 64 : 0);
#define CN_EPIH(K) epilogue_wide_h<X3_BN, K, C_H, D_H>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem, kind)
  switch (kind) {
    case 0: CN_EPIH(0); break;
    case 1: CN_EPIH(1); break;
    case 16: CN_EPIH(16); break;
    case 4: CN_EPIH(4); break;
    default: CN_EPIH(-1); break;
  }
#undef CN_EPIH
}

template <bool A_ACT, bool A_H, bool C_H, bool D_H>
__global__ __launch_bounds__(NTHREADS, 4) void cn_gemm_hnn_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  constexpr bool ONE = true;
  using RA = typename std::conditional<A_H, u32x2, f32x4>::type;     // one thread's share of an A tile row: 4 elements
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
  const unsigned a_voff = ((unsigned)min(row0 + arow, p.M - 1) * (unsigned)p.lda + akq * 4) * (A_H ? 2u : 4u);   // bytes
  const int a_lds = x3_offset(arow, akq >> 1) + (akq & 1) * 8;
  const unsigned b_voff = lane * 16;
  const size_t b_tile = (size_t)tile_n * nk * X3_B_BYTES;

  // one K range per launch (the host folds K-segments that are adjacent column blocks of one matrix into one K)
  const char* a0 = reinterpret_cast<const char*>(p.A[g]);
  const char* b0 = reinterpret_cast<const char*>(p.b_split[g]) + b_tile;
  auto a_base = [&](int v) -> const char* { return a0 + v * BK * (A_H ? 2 : 4); };
  auto b_base = [&](int v) -> const char* { return b0 + (size_t)v * X3_B_BYTES; };
  // register load of this thread's float4 of K-step v, outside the compiler's memory-counter bookkeeping
  auto a_issue = [&](RA& dst, int v) {
    const char* base = a_base(v);
    if constexpr (A_H) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(a_voff), "s"(base) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(a_voff), "s"(base) : "memory");
  };
  auto a_store = [&](RA raw, int buf) {
    if constexpr (A_H && !A_ACT) {      // bf16 in memory, bf16 in LDS: the eight bytes go through unchanged
      *reinterpret_cast<u32x2*>(lds + buf * X3_BUF_BYTES + a_lds) = raw;
      return;
    }
    f32x4 v;
    if constexpr (A_H) v = __builtin_convertvector(__builtin_bit_cast(bf16x4, raw), f32x4);
    else v = raw;
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
  auto b_issue_slot = [&](int v, int slot) {     // ring slot of x3_one_deep_loop
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
  auto step = [&](auto cur_c, int u, RA& r) {
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

  if constexpr (CN_ONE_DEEP) {
    x3_one_deep_loop<RA>(nsteps, lds, x3_offset(wm * S::WM + li, lh), x3_offset(wn * S::WN + li, lh), acc, a_issue, a_store,
                         b_issue_slot);
  } else if (nsteps > 0) {
    RA r0, r1;
    a_issue(r0, 0);
    b_issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0) :: "memory");
    a_store(r0, 0);
    if (nsteps > 1) a_issue(r1, 1);
    if (nsteps > 2) a_issue(r0, 2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(r0), "+v"(r1) :: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int u = 0;
    for (; u < nsteps; u += 2) {
      step(std::integral_constant<int, 0>{}, u, r1);
      if (u + 1 < nsteps) step(std::integral_constant<int, 1>{}, u + 1, r0);
    }
  }
  h_epilogue<C_H, D_H>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem);
}


template <bool B_ACT, bool A_H, bool B_H>
__global__ __launch_bounds__(NTHREADS, 4) void cn_gemm_htn_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  constexpr bool ONE = true;
  constexpr int EA = A_H ? 2 : 4, EB = B_H ? 2 : 4;     // bytes per stored element
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
  // every K-step of the slab, also a ragged last one: rows past K are read from row K-1 (in bounds) and the A rows
  // among them are zeroed on their way into LDS, so they add nothing (the fp32 kernel that takes the < 16-row tail of
  // the other precisions cannot read bf16 operands)
  const int nsteps = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // staging units: (k, 4 consecutive columns).  A: 16 x 32 = 512 units (one per thread); B: 16 x 64 = 1024 (two).
  const char* __restrict__ Ab = reinterpret_cast<const char*>(p.A[g]);
  const char* __restrict__ Bb = reinterpret_cast<const char*>(p.B[g]);
  const int ak = tid >> 5, ac = (tid & 31) * 4;
  const int bk0 = tid >> 6, bc = (tid & 63) * 4;          // second unit: k + 8
  // ragged last row tile (M % 128 != 0, M % 4 == 0): columns past M re-read the last four valid ones; the output rows
  // they feed are never stored
  const size_t a_col = (size_t)min(row0 + ac, p.M - 4), b_col = (size_t)(col0 + bc);
  const int klast = p.K - 1;
  const int a_lds = x3t_offset<BM>(ak, ac);
  const int b_lds0 = X3T_A_BYTES + x3t_offset<X3_BN>(bk0, bc);
  const int b_lds1 = X3T_A_BYTES + x3t_offset<X3_BN>(bk0 + 8, bc);

  using TA = typename std::conditional<A_H, u32x2, f32x4>::type;
  using TB = typename std::conditional<B_H, u32x2, f32x4>::type;
  TA ra;
  TB rb0, rb1;
  bool a_live = true;        // this thread's A row of the step being stored lies inside K
  bool a_live_next = true;
  auto load = [&](int u) {
    const int ka = kbeg + u * BK + ak, kb = kbeg + u * BK + bk0;
    a_live_next = ka <= klast;
    ra = *reinterpret_cast<const TA*>(Ab + ((size_t)min(ka, klast) * p.lda + a_col) * EA);
    rb0 = *reinterpret_cast<const TB*>(Bb + ((size_t)min(kb, klast) * p.ldb + b_col) * EB);
    rb1 = *reinterpret_cast<const TB*>(Bb + ((size_t)min(kb + 8, klast) * p.ldb + b_col) * EB);
  };
  auto put = [&](auto raw, char* dst, int plane_bytes, bool act) {
    constexpr bool H = sizeof(raw) == 8;
    if constexpr (H) {
      if (!act) {     // bf16 in memory, bf16 in LDS
        *reinterpret_cast<u32x2*>(dst) = raw;
        return;
      }
    }
    f32x4 v;
    if constexpr (H) v = __builtin_convertvector(__builtin_bit_cast(bf16x4, raw), f32x4);
    else v = raw;
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
    a_live = a_live_next;
    if (!a_live) {
      TA z;
#pragma unroll
      for (int c = 0; c < (int)(sizeof(TA) / 4); ++c) z[c] = 0;
      ra = z;
    }
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
