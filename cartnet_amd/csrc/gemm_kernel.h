// fp32 MFMA GEMM for the dense per-edge / per-node Linears of CartNet and their gradients (gfx950).
//
// Workgroup: 512 threads = 8 wavefronts (2 per SIMD) computing a 128 x BN output tile (BN = 256 / 128 / 64) with
// v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain, 16 accumulator VGPRs per 32x32 tile).  Waves form a 2x4 grid
// (4x2 for BN = 64); each owns (128/WGM) x (BN/WGN) outputs = at most 2x2 MFMA tiles = 64 accumulator VGPRs, so the
// kernel stays under 128 VGPRs and two workgroups (4 waves per SIMD) share a CU: while some waves stage operands or
// run their epilogue the others keep the matrix pipe busy.
//
// K loop: K-step 16, operands staged global -> registers -> LDS (SiLU can be applied in flight), LDS double
// buffered so there is ONE barrier per K-step: iteration t writes tile t+1 (loaded during t-1) into the other
// buffer, issues the global loads of tile t+2, then runs the MFMAs of tile t.
// A k-contiguous operand is read from LDS with one ds_read_b128 per four MFMAs: inside an 8-deep k group lane half
// h takes k = 4h..4h+3 (both operands use the same permutation; it only reorders an fp32 sum).
// Full tiles take a predicate-free load path; edge tiles / odd K / unaligned operands a checked one.
#pragma once
#include "common.h"
#include <type_traits>

namespace cn_gemm {

constexpr int BM = 128;
constexpr int BK = 16;
constexpr int KPAD = BK + 4;  // LDS row stride (floats) of a k-contiguous tile: 5 x 16 B slots -> conflict-free b128
constexpr int NTHREADS = 512;

struct GemmFlags {
  int vecA, vecB, kchunk;
  int tile_m0;   // first row tile this launch covers (full tiles and the ragged last row tile are separate launches)
  int split0;    // slab index of this launch's first K chunk (split-K: whole K-steps and the < 16-row tail are separate launches)
  int k_lo, k_hi;  // K range this launch reduces over; chunk y covers [k_lo + y*kchunk, min(k_hi, ...))
  int wide;        // every epilogue operand has 16-byte aligned rows -> vector epilogue
  int x3;          // bf16x3 split-operand MFMA requested (CartnetGemmArgs.precision == 1)
};

// SiLU inside the GEMM uses the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each); forward and backward use the
// same functions, so the recomputed activation in the weight-gradient GEMM equals the forward one bit for bit.
// (__builtin_amdgcn_rcpf is the bare v_rcp_f32; __frcp_rn / 1.0f / x compile to the correctly rounded division sequence --
//  v_div_scale, v_rcp, four FMAs, v_div_fmas, v_div_fixup -- ten VALU instructions per element in front of the MFMAs)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_silu(float x) { return x * fast_sigmoid(x); }
__device__ __forceinline__ float fast_dsilu(float x) {
  const float s = fast_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}
// softplus(x) = max(x, 0) + log1p(exp(-|x|)) on the hardware exp / log (comformer_ops.hip's softplus_f: the same expression,
// so a value is the same whether an epilogue or the element-wise kernel produced it); threshold 20 as torch's.
__device__ __forceinline__ float fast_softplus(float x) {
  if (x > 20.f) return x;
  const float t = __expf(-fabsf(x));
  const float l = t < 4.8828125e-4f ? t - 0.5f * t * t : __logf(1.0f + t);
  return fmaxf(x, 0.f) + l;
}

// One operand tile (ROWS x BK) per K-step: global -> registers -> LDS.
template <int ROWS, bool KS, bool ACT>
struct Stager {
  static constexpr int UNITS = ROWS * BK / 4;                       // float4 units per K-step
  static constexpr int NU = (UNITS + NTHREADS - 1) / NTHREADS;      // per thread
  static constexpr bool PARTIAL = (UNITS % NTHREADS) != 0;
  static constexpr int LDS_FLOATS = KS ? BK * ROWS : ROWS * KPAD;
  f32x4 reg[NU];

  template <bool FAST>
  __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int row0, int rows_limit, int k0,
                                       int kend, bool vec, int tid) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int unit = tid + u * NTHREADS;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (!PARTIAL || unit < UNITS) {
        if (!KS) {
          const int row = unit >> 2, kq = unit & 3;
          // FAST: rows past the end of a ragged last tile are clamped to the last valid row (their products are
          // computed and dropped by the epilogue's row check) so the load needs no predicate.
          const int grow = FAST ? min(row0 + row, rows_limit - 1) : row0 + row, gk = k0 + kq * 4;
          const float* ptr = base + (size_t)grow * ld + gk;
          if (FAST) {
            v = *reinterpret_cast<const f32x4*>(ptr);
          } else if (grow < rows_limit && gk < kend) {
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
          const float* ptr = base + (size_t)gk * ld + grow;
          if (FAST) {
            v = *reinterpret_cast<const f32x4*>(ptr);
          } else if (gk < kend && grow < rows_limit) {
            if (vec && grow + 3 < rows_limit) {
              v = *reinterpret_cast<const f32x4*>(ptr);
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if (grow + c < rows_limit) v[c] = ptr[c];
            }
          }
        }
      }
      reg[u] = v;
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int unit = tid + u * NTHREADS;
      if (PARTIAL && unit >= UNITS) continue;
      f32x4 v = reg[u];
      if (ACT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fast_silu(v[c]);
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


// ------------------------------------------------------------------------------------------------------------------
// bf16x3 split-operand path ("precision 1").  Every fp32 operand element x is split exactly into three bf16 pieces
// x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); 3 x 8 significand bits cover fp32's 24) while it
// is staged into LDS, and each fp32 product is rebuilt from SIX bf16 MFMA products accumulated in fp32:
//     x*y ~= h*h' + h*m' + m*h' + h*l' + l*h' + m*m'      (dropped: m*l', l*m', l*l' <= 2^-23 |x*y|, below fp32 rounding)
// v_mfma_f32_32x32x16_bf16 retires a 32x32x16 block in 32 cycles against 8 x 64 cycles for the fp32 MFMA, so six of
// them are 2.67x faster than the native fp32 matrix path at fp32-level accuracy (measured on the golden fixtures:
// same 1e-5 parity budget).  The accumulator layout is the same as the fp32 MFMA's, so the epilogues are shared.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Diagnostic build only (-DCN_CLOCK_STAMP, tools/experiments/exp_clock_x3.py): one (shader clock ticks, 100 MHz ticks) pair per
// workgroup around the main loop, into a buffer nothing else reads; the product build has no stamp.
#ifdef CN_CLOCK_STAMP
static __device__ unsigned long long cn_clock_dbg[2 * 4096];
#define CN_STAMP_BEGIN() const unsigned long long cn_t0 = __builtin_amdgcn_s_memtime(), cn_w0 = __builtin_amdgcn_s_memrealtime()
#define CN_STAMP_END()                                                                   \
  if (threadIdx.x == 0) {                                                                \
    cn_clock_dbg[2 * (blockIdx.x & 4095)] = __builtin_amdgcn_s_memtime() - cn_t0;        \
    cn_clock_dbg[2 * (blockIdx.x & 4095) + 1] = __builtin_amdgcn_s_memrealtime() - cn_w0; \
  }
#else
#define CN_STAMP_BEGIN()
#define CN_STAMP_END()
#endif
// Second diagnostic (-DCN_PHASE_STAMP, tools/experiments/exp_phases.py): per workgroup of the fp32 activation x weight kernel the
// 100 MHz time at entry, main-loop start, main-loop end, last store issued and all stores acknowledged, with the
// hardware id of the CU it ran on -- the life of every workgroup on every CU of one launch.
#ifdef CN_PHASE_STAMP
static __device__ unsigned long long cn_phase_dbg[8192 * 8];
#define CN_PHASE(slot)                                                                                         \
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) {   /* wave 0, a scalar branch */                 \
    const unsigned cn_lin = blockIdx.x + gridDim.x * blockIdx.z;                                               \
    if (cn_lin < 8192) cn_phase_dbg[cn_lin * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();                   \
  }
#define CN_PHASE_ID()                                                                                          \
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) {                                                 \
    const unsigned cn_hw = __builtin_amdgcn_s_getreg((4 | (0 << 6) | (31 << 11)));   /* HW_REG_HW_ID, 32 bits */  \
    const unsigned cn_xcc = __builtin_amdgcn_s_getreg((20 | (0 << 6) | (31 << 11))); /* HW_REG_XCC_ID */         \
    const unsigned cn_lin = blockIdx.x + gridDim.x * blockIdx.z;                                               \
    if (cn_lin < 8192) cn_phase_dbg[cn_lin * 8 + 5] = ((unsigned long long)cn_xcc << 32) | cn_hw;              \
  }
#else
#define CN_PHASE(slot)
#define CN_PHASE_ID()
#endif

struct Split3 {
  bf16x4 h, m, l;
};

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two fp32 -> one dword of two bf16 (round to nearest even: v_cvt_pk_bf16_f32), first value in the low half
__device__ __forceinline__ unsigned x3_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// One slice of the exact split on PACKED conversions: piece = bf16(v) as two dwords, rest = v - piece.  The packed
// dword is both the stored piece and, shifted / masked, its two fp32 values: 9 vector instructions per slice of four
// (2 v_cvt_pk, 2 shifts, 2 ands, 2 v_pk_add_f32... ) against 14 when every element is converted and widened alone.
__device__ __forceinline__ void x3_slice(const f32x4& v, u32x2& piece, f32x4& rest) {
  piece[0] = x3_pk(v[0], v[1]);
  piece[1] = x3_pk(v[2], v[3]);
  rest[0] = v[0] - __builtin_bit_cast(float, piece[0] << 16);
  rest[1] = v[1] - __builtin_bit_cast(float, piece[0] & 0xffff0000u);
  rest[2] = v[2] - __builtin_bit_cast(float, piece[1] << 16);
  rest[3] = v[3] - __builtin_bit_cast(float, piece[1] & 0xffff0000u);
}

// (The packed slices were measured in split3 too: the weight-gradient kernels, which split three float4 per thread and
// K-step, got 2-3 % SLOWER with 12 fewer vector instructions per float4 -- same-box A/B, round 3 -- so the general
// form stays as it was and only gemm_x3s.h uses x3_slice.)
__device__ __forceinline__ Split3 split3(f32x4 v) {
  Split3 s;
  s.h = __builtin_convertvector(v, bf16x4);
  const f32x4 r1 = v - __builtin_convertvector(s.h, f32x4);
  s.m = __builtin_convertvector(r1, bf16x4);
  const f32x4 r2 = r1 - __builtin_convertvector(s.m, f32x4);
  s.l = __builtin_convertvector(r2, bf16x4);
  return s;
}

// LDS image of one operand tile: 3 planes (h, m, l) of [ROWS][16 k] bf16 = 32 B per row, the two 16-byte k-halves of
// rows 16..31 (mod 32) swapped.  A ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,
// 28-31} and the same + 32: MI355X_MICROARCH.md, LDS); with this swap every group touches 16 distinct 16-byte slots of
// the 256-byte row of banks for BOTH fragment shapes in use: 32 rows x one k-half per half-wave (32x32x16 MFMA) and
// 16 rows x two k-halves per half-wave (16x16x32 MFMA, gemm_x3s.h).
__device__ __forceinline__ int x3_offset(int row, int khalf) { return row * 32 + ((khalf ^ ((row >> 4) & 1)) << 4); }

template <int ROWS, bool KS, bool ACT>
struct Stager3 {
  static constexpr int UNITS = ROWS * 4;                            // (row, k-quad) units of 4 consecutive k
  static constexpr int NU = (UNITS + NTHREADS - 1) / NTHREADS;
  static constexpr int PLANE_BYTES = ROWS * 32;
  static constexpr int LDS_FLOATS = 3 * PLANE_BYTES / 4;
  f32x4 reg[NU];

  template <bool FAST>
  __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int row0, int rows_limit, int k0,
                                       int kend, bool vec, int tid) {
    static_assert(FAST, "the bf16x3 path only has the predicate-free kernel");
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int unit = tid + u * NTHREADS;
      if (!KS) {
        const int row = unit >> 2, kq = unit & 3;
        const int grow = min(row0 + row, rows_limit - 1);
        reg[u] = *reinterpret_cast<const f32x4*>(base + (size_t)grow * ld + k0 + kq * 4);
      } else {   // lanes run along the contiguous row index: four coalesced dword loads, one per k
        const int row = unit % ROWS, kq = unit / ROWS;
        const float* ptr = base + (size_t)(k0 + kq * 4) * ld + row0 + row;
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = ptr[(size_t)c * ld];
        reg[u] = v;
      }
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
    char* base = reinterpret_cast<char*>(lds);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int unit = tid + u * NTHREADS;
      const int row = KS ? unit % ROWS : unit >> 2;
      const int kq = KS ? unit / ROWS : unit & 3;
      f32x4 v = reg[u];
      if (ACT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fast_silu(v[c]);
      }
      const Split3 s = split3(v);
      char* dst = base + x3_offset(row, kq >> 1) + (kq & 1) * 8;
      *reinterpret_cast<bf16x4*>(dst) = s.h;
      *reinterpret_cast<bf16x4*>(dst + PLANE_BYTES) = s.m;
      *reinterpret_cast<bf16x4*>(dst + 2 * PLANE_BYTES) = s.l;
    }
  }
};

template <int BN>
struct Shape {
  static constexpr int WGM = (BN >= 128) ? 2 : 4;   // wave grid
  static constexpr int WGN = (BN >= 128) ? 4 : 2;
  static constexpr int WM = BM / WGM;               // rows per wave
  static constexpr int WN = BN / WGN;               // columns per wave
  static constexpr int TM = WM / 32;
  static constexpr int TN = WN / 32;
};

// Epilogue flags: bit 0 gather, 1 resid, 2 dact, 3 column sums (fp32 per lane), 4 column sums + squares (fp64),
// 5 cpre, 6 out_act, (7 gate statistics: wide form only,) 8 the dact factor is sigmoid (softplus') instead of silu'.  KIND >= 0: compile-time flags (branch-free per-element code); KIND < 0: read at run time.
template <int BN, int KIND>
__device__ __forceinline__ void epilogue(const CartnetGemmArgs& p, f32x16 (&acc)[Shape<BN>::TM][Shape<BN>::TN], int g,
                                         int row0, int col0, int tile_m, int wm, int wn, int li, int lh, int tid,
                                         double* red, int rt_kind) {
  using S = Shape<BN>;
  const int kind = (KIND >= 0) ? KIND : rt_kind;
  const bool GATHER = kind & 1, RESID = kind & 2, DACT = kind & 4, SUM1 = kind & 8, SUM2 = kind & 16,
             CPRE = kind & 32, OUTACT = kind & 64;
  float* C = p.C[g];                       // may alias resid / dact (in-place use): no __restrict__
  const float* __restrict__ bias = p.bias[g];
  const float* __restrict__ gi = p.gather_i[g];
  const float* __restrict__ gj = p.gather_j[g];
  const float* resid = p.resid[g];
  const float* dact = p.dact[g];
  float* cpre = p.cpre[g];

  float biasv[S::TN];
  double cs[S::TN], cq[S::TN];
  float csf[S::TN];
#pragma unroll
  for (int b = 0; b < S::TN; ++b) {
    const int gcol = col0 + wn * S::WN + b * 32 + li;
    biasv[b] = (bias && gcol < p.N) ? bias[gcol] : 0.f;
    cs[b] = 0.0;
    cq[b] = 0.0;
    csf[b] = 0.f;
  }
#pragma unroll
  for (int a = 0; a < S::TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int grow = row0 + wm * S::WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (grow >= p.M) continue;
      int ti = 0, sj = 0;
      if (GATHER) {
        ti = p.tgt[grow];
        sj = p.src[grow];
      }
#pragma unroll
      for (int b = 0; b < S::TN; ++b) {
        const int gcol = col0 + wn * S::WN + b * 32 + li;
        if (gcol >= p.N) continue;
        float v = acc[a][b][r] + biasv[b];
        if (GATHER) v += gi[(size_t)ti * p.ldg + gcol] + gj[(size_t)sj * p.ldg + gcol];
        if (RESID) v += resid[(size_t)grow * p.ldr + gcol];
        if (DACT) {
          const float d = dact[(size_t)grow * p.ldd + gcol];
          v *= (kind & 256) ? fast_sigmoid(d) : fast_dsilu(d);      // bit 8: dact_kind 1, softplus'
        }
        if (SUM1) csf[b] += v;
        if (SUM2) {
          cs[b] += (double)v;
          cq[b] += (double)v * (double)v;
        }
        if (CPRE) cpre[(size_t)grow * p.ldc + gcol] = v;
        if (OUTACT) v = (kind & 256) ? fast_softplus(v) : fast_silu(v);
        C[(size_t)grow * p.ldc + gcol] = v;
      }
    }
  if (SUM1 || SUM2) {
    // rows of this block -> one fp64 partial per column: lane halves, then the WGM waves stacked in M.
    double* __restrict__ colsum = p.colsum[g];
    double* __restrict__ colsq = p.colsq[g];
    __syncthreads();   // the LDS staging buffers are dead after the K loop's last barrier: reuse as red[2][WGM][BN]
#pragma unroll
    for (int b = 0; b < S::TN; ++b) {
      double s = SUM2 ? cs[b] : (double)csf[b];
      double q = cq[b];
      s += __shfl_xor(s, 32);
      q += __shfl_xor(q, 32);
      if (lh == 0) {
        red[(0 * S::WGM + wm) * BN + wn * S::WN + b * 32 + li] = s;
        red[(1 * S::WGM + wm) * BN + wn * S::WN + b * 32 + li] = q;
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
          q += red[(1 * S::WGM + w) * BN + c];
        }
        colsum[(size_t)tile_m * p.N + gcol] = s;
        if (SUM2) colsq[(size_t)tile_m * p.N + gcol] = q;
      }
    }
  }
}


// Wide epilogue: each 32x32 accumulator tile is transposed through a per-wave LDS scratch (32 rows x 36 floats) so
// that a lane owns 4 consecutive columns of a row: node-term gathers, residual / pre-activation reads and the output
// go to memory as 16-byte vectors (8 rows x 128 B per wave instruction) instead of 4-byte scalars -- a quarter of the
// memory instructions of the scalar epilogue.  Needs 16-byte aligned rows everywhere (checked on the host: fl.wide).
constexpr int SCR_LD = 36;
constexpr int SCR_FLOATS = 32 * SCR_LD;

__device__ __forceinline__ f32x4 ldv4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void stv4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// A store whose next reader is milliseconds away (what forward keeps for backward): written through, not kept in the
// caches (round 5: the same policy on the kept silu(pre) of the layers' second products was worth 0.17 ms per step --
// gemm_f32ao.h).  (s_nop: a store of more than 8 bytes reads its data registers up to two cycles after issue and the
// hazard recogniser does not look inside inline asm.)
__device__ __forceinline__ void stv4_stream(float* p, f32x4 v) {
#ifdef CN_NO_STREAM_STORES
  stv4(p, v);
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#endif
}

// Accumulator layouts the wide epilogue takes: 32x32 MFMA tiles (f32x16 per tile: column = lane & 31, rows
// (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) or 16x16 tiles (f32x4 per tile: column = lane & 15, rows 4 (lane >> 4) + r;
// a 32x32 block of the wave tile is 2 x 2 of them).  Block (a, b) of the wave tile -> the wave's scratch, row-major.
template <int TM, int TN>
__device__ __forceinline__ void acc_block_to_scr(const f32x16 (&acc)[TM][TN], int a, int b, float* scr, int lane,
                                                 int ld) {
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * ld + li] = acc[a][b][r];
}
template <int TM, int TN>
__device__ __forceinline__ void acc_block_to_scr(const f32x4 (&acc)[TM][TN], int a, int b, float* scr, int lane,
                                                 int ld) {
  const int li = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int ta = 0; ta < 2; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) scr[(16 * ta + 4 * lq + r) * ld + 16 * tb + li] = acc[2 * a + ta][2 * b + tb][r];
}

// S: the wave grid (Shape<BN> for the 512-thread kernels, ShapeQ for the 256-thread one of gemm_f32q.h); NT: threads per
// workgroup.
template <class S, int BN, int NT, int KIND, class ACC>
__device__ __forceinline__ void epilogue_wide_s(const CartnetGemmArgs& p, ACC& acc,
                                                int g, int row0, int col0, int tile_m, int wm, int wn, int lane, int tid,
                                                float* smem, int rt_kind) {
  constexpr bool MF16 = sizeof(acc[0][0]) == sizeof(f32x4);     // 16x16 tiles
  const int kind = (KIND >= 0) ? KIND : rt_kind;
  const bool GATHER = kind & 1, RESID = kind & 2, DACT = kind & 4, SUM1 = kind & 8, SUM2 = kind & 16,
             CPRE = kind & 32, OUTACT = kind & 64;
  // bit 7: gate statistics (CartnetGemmArgs.gst_*), compile-time kinds only -- the run-time form (KIND < 0) does not carry
  // the code, and cartnet_gemm refuses a launch with gst_g set that would not reach a kernel with the case
  constexpr bool GST = KIND >= 0 && (KIND & 128) != 0;
  constexpr int KBASE = KIND >= 0 ? (KIND & 127) : KIND;      // the kind without the extension bits
  // bit 8: CartnetGemmArgs.dact_kind = 1, the activation of this launch is softplus: v *= sigmoid(dact) (softplus') instead
  // of silu'(dact), out_act = softplus instead of SiLU
  const bool DSP = (kind & 256) != 0;
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
  double cs[S::TN][2], cq[S::TN][2];      // [.][1]: the second 16-column half of a block (16x16 tiles only)
#pragma unroll
  for (int b = 0; b < S::TN; ++b) {
    const int gcol = col0 + wn * S::WN + b * 32 + c4 * 4;
    bias4[b] = bias ? ldv4(bias + gcol) : f32x4{0.f, 0.f, 0.f, 0.f};
    sum4[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    cs[b][0] = cs[b][1] = 0.0;
    cq[b][0] = cq[b][1] = 0.0;
  }
  f32x4 gmean[S::TN], grstd[S::TN], ggam[S::TN], gbet[S::TN], gs1[S::TN], gs2[S::TN];
  if constexpr (GST) {
#pragma unroll
    for (int b = 0; b < S::TN; ++b) {
      const int gcol = col0 + wn * S::WN + b * 32 + c4 * 4;
      // per column: ghat = g * rstd - mean * rstd;  -log2(e) * bn = ghat * (-log2(e) gamma) - log2(e) beta
      grstd[b] = ldv4(p.gst_mean_rstd + p.N + gcol);
      gmean[b] = -ldv4(p.gst_mean_rstd + gcol) * grstd[b];
      ggam[b] = ldv4(p.gst_gamma + gcol) * -1.44269504088896f;
      gbet[b] = ldv4(p.gst_beta + gcol) * -1.44269504088896f;
      gs1[b] = f32x4{0.f, 0.f, 0.f, 0.f};
      gs2[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (SUM2) {   // BatchNorm statistics of v = acc + bias, taken in the accumulator layout (column = lane)
    if constexpr (MF16) {
      const int l16 = lane & 15, lq = lane >> 4;
#pragma unroll
      for (int b = 0; b < S::TN; ++b)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          const float bv = bias ? bias[col0 + wn * S::WN + b * 32 + tb * 16 + l16] : 0.f;
#pragma unroll
          for (int a = 0; a < 2 * S::TM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int grow = row0 + wm * S::WM + a * 16 + 4 * lq + r;
              if (grow < p.M) {
                const double v = (double)(acc[a][2 * b + tb][r] + bv);
                cs[b][tb] += v;
                cq[b][tb] += v * v;
              }
            }
        }
    } else {
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
              cs[b][0] += v;
              cq[b][0] += v * v;
            }
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
#ifdef CN_EPI_RAW      /* diagnostic: same stores, same addresses, no transpose and no operands (WRONG values) */
      if constexpr (!MF16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (grow[i] < p.M)
            stv4(C + (size_t)grow[i] * p.ldc + gcol,
                 f32x4{acc[a][b][4 * i], acc[a][b][4 * i + 1], acc[a][b][4 * i + 2], acc[a][b][4 * i + 3]});
#ifdef CN_PHASE_STAMP
        if (a == 0 && b == 0) { CN_PHASE(6); }
        if (a == 0 && b == S::TN - 1) { CN_PHASE(7); }
#endif
        continue;
      }
#endif
      acc_block_to_scr(acc, a, b, scr, lane, SCR_LD);
      __builtin_amdgcn_wave_barrier();
      // The two epilogues with ONE operand row per output row and nothing else (residual: dE; `silu'`: dpre) load the
      // block's four operand rows together before its first store: loads and stores retire through one in-order
      // counter, so a row that loads its operand after the previous row's store waits for that store's acknowledgement
      // too -- four load + store round trips in a row per block (profiles/r03_exp_phases.md; -1.3 % / -1 % per launch).
      // The forms with more operands or statistics spill when they do the same (measured), so they keep the row loop.
      // The store-only forms (plain, and the one with BatchNorm statistics, which are taken before this loop) read the
      // block's four rows into four register sets and issue its four stores back to back: the compiler puts
      // s_waitcnt vmcnt(0) in front of any write to a register a store in flight has read from, and with ONE set
      // recycled row by row that was a wait for the previous row's store to be ACKNOWLEDGED on every row (1-2 us next
      // to a busy partner workgroup: profiles/r03_exp_phases.md); now it is one wait per block.  Same-box A B A B:
      // K = 80 / 128 launches -4 %, K = 256 -0.5 % (fp32) and -1...-2 % (bf16x3); the training step within noise.
      constexpr bool SETS = (KBASE == 0 || KBASE == 16) && !GST;
      if constexpr (SETS) {
        f32x4 vv[4];
        float* ptr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          vv[i] = ldv4(scr + (rsub + 8 * i) * SCR_LD + c4 * 4) + bias4[b];
          ptr[i] = C + (size_t)min(grow[i], p.M - 1) * p.ldc + gcol;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (grow[i] < p.M) stv4(ptr[i], vv[i]);
        // values and addresses stay in their own registers until the last store of the block has been issued
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(vv[i]), "v"(ptr[i]));
        __builtin_amdgcn_wave_barrier();
#ifdef CN_PHASE_STAMP
        if (a == 0 && b == 0) { CN_PHASE(6); }
        if (a == 0 && b == S::TN - 1) { CN_PHASE(7); }
#endif
        continue;
      }
      // (bf16x3 kernel only, where the epilogue is longer than the partner's main loop and therefore on the critical
      //  path: the gather form loads its two node-term rows for two output rows at a time -- 279.6 -> 273.7 us per layer
      //  GEMM 1; the 128-wide fp32 kernel that takes this form at precision 0 has 80 registers and spills with it)
      constexpr bool BATCH_G = KBASE == 1 && MF16;
      f32x4 og[4];
      constexpr bool BATCH = (KBASE == 2 || KBASE == 4) && !GST;
      f32x4 op[4];
      if constexpr (BATCH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int gr = min(grow[i], p.M - 1);            // rows past M load row M - 1 and are dropped below
          op[i] = (KBASE == 2) ? ldv4(resid + (size_t)gr * p.ldr + gcol) : ldv4(dact + (size_t)gr * p.ldd + gcol);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (BATCH_G) {
          if ((i & 1) == 0) {
#pragma unroll
            for (int k = i; k < i + 2; ++k) {
              const bool in = grow[k] < p.M;
              og[2 * (k - i)] = ldv4(gi + (size_t)(in ? ti[k] : 0) * p.ldg + gcol);
              og[2 * (k - i) + 1] = ldv4(gj + (size_t)(in ? sj[k] : 0) * p.ldg + gcol);
            }
          }
        }
        f32x4 v = ldv4(scr + (rsub + 8 * i) * SCR_LD + c4 * 4);
        if (grow[i] >= p.M) continue;
        v += bias4[b];
        if (GATHER) v += BATCH_G ? og[2 * (i & 1)] + og[2 * (i & 1) + 1]
                                 : ldv4(gi + (size_t)ti[i] * p.ldg + gcol) + ldv4(gj + (size_t)sj[i] * p.ldg + gcol);
        if (RESID) v += BATCH ? op[i] : ldv4(resid + (size_t)grow[i] * p.ldr + gcol);
        if (DACT) {
          const f32x4 d = BATCH ? op[i] : ldv4(dact + (size_t)grow[i] * p.ldd + gcol);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] *= DSP ? fast_sigmoid(d[q]) : fast_dsilu(d[q]);
        }
        if (SUM1) sum4[b] += v;
        if constexpr (GST) {
          // v = de_out of the layer below: its share of that layer's BatchNorm-backward sums, sum(v w) and sum(v w ghat),
          // w = env sigma' (the gate is recomputed from the kept pre-activation g and the kept statistics)
          const f32x4 gv = ldv4(p.gst_g + (size_t)grow[i] * p.gst_ld + gcol);
          const float ev = p.gst_env ? p.gst_env[grow[i]] : 1.0f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // z (1 - z) = t / (1 + t)^2 with t = exp(-bn): one v_exp_f32, one v_rcp_f32.  (the clamp keeps t finite: beyond
            // it z (1 - z) < 1e-37 either way, and inf * 0 would be a NaN)
            const float ghat = gv[q] * grstd[b][q] + gmean[b][q];
            const float t = __builtin_amdgcn_exp2f(fminf(ghat * ggam[b][q] + gbet[b][q], 126.0f));
            const float r = __builtin_amdgcn_rcpf(1.0f + t);
            const float tv = (v[q] * ev) * ((t * r) * r);
            gs1[b][q] += tv;
            gs2[b][q] += tv * ghat;
          }
        }
        if (CPRE) stv4_stream(cpre + (size_t)grow[i] * p.ldc + gcol, v);
        if (OUTACT) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = DSP ? fast_softplus(v[q]) : fast_silu(v[q]);
        }
        stv4(C + (size_t)grow[i] * p.ldc + gcol, v);
      }
      __builtin_amdgcn_wave_barrier();
#ifdef CN_PHASE_STAMP
      if (a == 0 && b == 0) { CN_PHASE(6); }
      if (a == 0 && b == S::TN - 1) { CN_PHASE(7); }
#endif
    }
  }
  if (SUM1 || SUM2 || GST) {
    double* red = reinterpret_cast<double*>(smem);   // [2][WGM][BN]
    double* __restrict__ colsum = p.colsum[g];
    double* __restrict__ colsq = p.colsq[g];
    __syncthreads();   // every wave is done with its transpose scratch
#pragma unroll
    for (int b = 0; b < S::TN; ++b) {
      if constexpr (GST) {
        f32x4 s = gs1[b], t = gs2[b];     // fp32 over a lane's 4 * TM rows, fp64 from here on
#pragma unroll
        for (int o = 8; o <= 32; o <<= 1)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            s[q] += __shfl_xor(s[q], o);
            t[q] += __shfl_xor(t[q], o);
          }
        if (rsub == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            red[(0 * S::WGM + wm) * BN + wn * S::WN + b * 32 + c4 * 4 + q] = (double)s[q];
            red[(1 * S::WGM + wm) * BN + wn * S::WN + b * 32 + c4 * 4 + q] = (double)t[q];
          }
        }
      } else if (SUM2) {
        if constexpr (MF16) {
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) {
            double s = cs[b][tb], q = cq[b][tb];
            s += __shfl_xor(s, 16);
            q += __shfl_xor(q, 16);
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            if (lane < 16) {
              red[(0 * S::WGM + wm) * BN + wn * S::WN + b * 32 + tb * 16 + lane] = s;
              red[(1 * S::WGM + wm) * BN + wn * S::WN + b * 32 + tb * 16 + lane] = q;
            }
          }
        } else {
          double s = cs[b][0], q = cq[b][0];
          s += __shfl_xor(s, 32);
          q += __shfl_xor(q, 32);
          if (lh == 0) {
            red[(0 * S::WGM + wm) * BN + wn * S::WN + b * 32 + li] = s;
            red[(1 * S::WGM + wm) * BN + wn * S::WN + b * 32 + li] = q;
          }
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
    for (int c = tid; c < BN; c += NT) {
      const int gcol = col0 + c;
      if (gcol < p.N) {
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int w = 0; w < S::WGM; ++w) {
          s += red[(0 * S::WGM + w) * BN + c];
          if (SUM2 || GST) q += red[(1 * S::WGM + w) * BN + c];
        }
        colsum[(size_t)tile_m * p.N + gcol] = s;
        if (SUM2 || GST) colsq[(size_t)tile_m * p.N + gcol] = q;
      }
    }
  }
}

template <int BN, int KIND, class ACC>
__device__ __forceinline__ void epilogue_wide(const CartnetGemmArgs& p, ACC& acc,
                                              int g, int row0, int col0, int tile_m, int wm, int wn, int lane, int tid,
                                              float* smem, int rt_kind) {
  epilogue_wide_s<Shape<BN>, BN, NTHREADS, KIND>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem, rt_kind);
}

// The same with several column tiles per row tile (N = 512: two 256-wide tiles; the 128-wide kernel at N = 256): the
// column tiles of a row tile read the same A rows as its groups do, so all `tiles_n x groups` workgroups of a row tile
// go 8 dispatch slots apart (PMC, round 3: layer GEMM 1 on the 128-wide kernel fetched e twice, 548 MB per launch
// against 181 MB of e + the node-term gathers).  bx = tile_m * tiles_n + tile_n as in the plain order.
__device__ __forceinline__ void cn_block_map(int& bx, int& g, int tiles_n);

// Block -> (tile, K-chunk) map of the split-K weight-gradient kernels (grid: x = tiles, y = K-chunks, z = groups).  The
// tiles of one K-chunk read the same rows of one operand (M = 256: two row tiles share the whole B chunk).  Workgroups
// go to the 8 XCDs round-robin in dispatch order (x fastest, then y), so neighbours in x never share an L2 and the shared
// chunk used to come from beyond the L2 once per tile (PMC, round 3: 1155 MB fetched per edge-sized fp32 launch against
// 726 MB of operands; 793 MB with this map).  The tiles of a K-chunk sit 8 dispatch slots apart: same XCD, microseconds
// apart.  Chunks past the last multiple of 8 keep the plain order.
__device__ __forceinline__ void cn_splitk_block_map(int& bx, int& by) {
  bx = blockIdx.x;
  by = blockIdx.y;
  const int T = gridDim.x, Y = gridDim.y, L = bx + T * by, full = (Y >> 3) * 8 * T;
  if (T > 1 && L < full) {
    const int chunk = L / (8 * T), r = L - chunk * 8 * T;
    by = chunk * 8 + (r & 7);
    bx = r >> 3;
  }
}

// Block -> (tile, group) map for grouped launches without split-K.  Workgroups go to the 8 XCDs round-robin in
// dispatch order (x fastest, then z), and groups of one row tile read the same A rows (layer GEMM1: both MLPs read e;
// node projections: four products of x): the groups of a tile are placed 8 dispatch slots apart -- same XCD, same
// L2, a few microseconds apart -- instead of a whole grid apart.
__device__ __forceinline__ void cn_block_map(int& bx, int& g) {
  const int X = gridDim.x, G = gridDim.z;
  if (G == 1 || gridDim.y != 1) {
    bx = blockIdx.x;
    g = blockIdx.z;
    return;
  }
  const int L = blockIdx.x + X * blockIdx.z;
  const int X8 = X & ~7;
  if (L < X8 * G) {
    const int chunk = L / (8 * G), w = L - chunk * 8 * G;
    g = w >> 3;
    bx = chunk * 8 + (w & 7);
  } else {
    const int r = L - X8 * G, rem = X - X8;
    g = r / rem;
    bx = X8 + r % rem;
  }
}

__device__ __forceinline__ void cn_block_map(int& bx, int& g, int tiles_n) {
  if (tiles_n <= 1 || gridDim.y != 1) {
    cn_block_map(bx, g);
    return;
  }
  const int X = gridDim.x, G = gridDim.z;
  const int TM = X / tiles_n, S = tiles_n * G;      // row tiles; workgroups sharing one row tile's A rows
  const int L = blockIdx.x + X * blockIdx.z;
  const int TM8 = TM & ~7;
  int tile_m, sub;
  if (L < TM8 * S) {
    const int chunk = L / (8 * S), w = L - chunk * 8 * S;
    tile_m = chunk * 8 + (w & 7);
    sub = w >> 3;
  } else {
    const int r = L - TM8 * S, rem = TM - TM8;
    tile_m = TM8 + r % rem;
    sub = r / rem;
  }
  g = sub / tiles_n;
  bx = tile_m * tiles_n + sub % tiles_n;
}

// FAST: every tile of the launch is full, K is a whole number of K-steps and rows are 16-byte aligned -> the operand
// loads carry no predicates.  The checked variant is a separate kernel so its register needs do not leak into this one.
template <bool A_KS, bool B_KS, int BN, bool A_ACT, bool B_ACT, bool FAST, int PREC>
__global__ __launch_bounds__(NTHREADS, 4) void cn_gemm_kernel(const CartnetGemmArgs p, const GemmFlags fl) {
  using S = Shape<BN>;
  using StA = typename std::conditional<PREC == 1, Stager3<BM, A_KS, A_ACT>, Stager<BM, A_KS, A_ACT>>::type;
  using StB = typename std::conditional<PREC == 1, Stager3<BN, B_KS, B_ACT>, Stager<BN, B_KS, B_ACT>>::type;
  constexpr int BUF = StA::LDS_FLOATS + StB::LDS_FLOATS;
  static_assert(2 * BUF * sizeof(float) >= 2 * S::WGM * BN * sizeof(double), "statistics scratch must fit in LDS");
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / S::WGN, wn = wid % S::WGN;
  const int li = lane & 31, lh = lane >> 5;

  const int tiles_n = (p.N + BN - 1) / BN;
  int bx, g;
  cn_block_map(bx, g);
  const int tile_m = fl.tile_m0 + bx / tiles_n, tile_n = bx % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;
  const int split = fl.split0 + blockIdx.y;

  f32x16 acc[S::TM][S::TN];
#pragma unroll
  for (int a = 0; a < S::TM; ++a)
#pragma unroll
    for (int b = 0; b < S::TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int kbeg = fl.k_lo + blockIdx.y * fl.kchunk;
  const int kend = min(fl.k_hi, kbeg + fl.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  StA stA;
  StB stB;

  auto compute = [&](const float* __restrict__ sA, const float* __restrict__ sB) {
    if constexpr (PREC == 1) {
      // one K-step = one 32x32x16 bf16 MFMA per product term; operands: 8 consecutive k per lane (k-half = lane >> 5)
      const char* cA = reinterpret_cast<const char*>(sA);
      const char* cB = reinterpret_cast<const char*>(sB);
      bf16x8 ah[S::TM], am[S::TM], al[S::TM];
#pragma unroll
      for (int a = 0; a < S::TM; ++a) {
        const char* q = cA + x3_offset(wm * S::WM + a * 32 + li, lh);
        ah[a] = *reinterpret_cast<const bf16x8*>(q);
        am[a] = *reinterpret_cast<const bf16x8*>(q + StA::PLANE_BYTES);
        al[a] = *reinterpret_cast<const bf16x8*>(q + 2 * StA::PLANE_BYTES);
      }
#pragma unroll
      for (int b = 0; b < S::TN; ++b) {
        const char* q = cB + x3_offset(wn * S::WN + b * 32 + li, lh);
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(q);
        const bf16x8 bm = *reinterpret_cast<const bf16x8*>(q + StB::PLANE_BYTES);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(q + 2 * StB::PLANE_BYTES);
#pragma unroll
        for (int a = 0; a < S::TM; ++a) {   // small terms first
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bm, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bh, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bm, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh, acc[a][b], 0, 0, 0);
        }
      }
      return;
    } else {
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float af[S::TM][4], bf[S::TN][4];
#pragma unroll
      for (int a = 0; a < S::TM; ++a) {
        if (!A_KS) {
          const f32x4 v =
              *reinterpret_cast<const f32x4*>(&sA[(wm * S::WM + a * 32 + li) * KPAD + kg * 8 + lh * 4]);
#pragma unroll
          for (int j = 0; j < 4; ++j) af[a][j] = v[j];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) af[a][j] = sA[(kg * 8 + lh * 4 + j) * BM + wm * S::WM + a * 32 + li];
        }
      }
#pragma unroll
      for (int b = 0; b < S::TN; ++b) {
        if (!B_KS) {
          const f32x4 v =
              *reinterpret_cast<const f32x4*>(&sB[(wn * S::WN + b * 32 + li) * KPAD + kg * 8 + lh * 4]);
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[b][j] = v[j];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[b][j] = sB[(kg * 8 + lh * 4 + j) * BN + wn * S::WN + b * 32 + li];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < S::TM; ++a)
#pragma unroll
          for (int b = 0; b < S::TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][j], bf[b][j], acc[a][b], 0, 0, 0);
    }
    }
  };

  auto kloop = [&](const float* __restrict__ Ab, const float* __restrict__ Bb) {
    const bool va = fl.vecA != 0, vb = fl.vecB != 0;
    stA.template load<FAST>(Ab, p.lda, row0, p.M, kbeg, kend, va, tid);
    stB.template load<FAST>(Bb, p.ldb, col0, p.N, kbeg, kend, vb, tid);
    stA.store(smem, tid);
    stB.store(smem + StA::LDS_FLOATS, tid);
    if (nk > 1) {
      stA.template load<FAST>(Ab, p.lda, row0, p.M, kbeg + BK, kend, va, tid);
      stB.template load<FAST>(Bb, p.ldb, col0, p.N, kbeg + BK, kend, vb, tid);
    }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      float* cur = smem + (t & 1) * BUF;
      float* nxt = smem + ((t & 1) ^ 1) * BUF;
      if (t + 1 < nk) {               // tile t+1 (loaded during iteration t-1) -> the other LDS buffer
        stA.store(nxt, tid);
        stB.store(nxt + StA::LDS_FLOATS, tid);
      }
      if (t + 2 < nk) {               // tile t+2 -> registers, lands while tile t is multiplied
        stA.template load<FAST>(Ab, p.lda, row0, p.M, kbeg + (t + 2) * BK, kend, va, tid);
        stB.template load<FAST>(Bb, p.ldb, col0, p.N, kbeg + (t + 2) * BK, kend, vb, tid);
      }
      compute(cur, cur + StA::LDS_FLOATS);
      __syncthreads();
    }
  };

  if (nk > 0) {
    for (int s = 0; s < p.nsegs; ++s) {
      const int idx = (p.ngroups > 1) ? g : s;
      kloop(p.A[idx], p.B[idx]);
    }
  }

  // ---------------------------------------------------------------- epilogue
  if (p.splitk > 1) {
    float* __restrict__ C = p.C[g] + (size_t)split * p.M * p.ldc;
#pragma unroll
    for (int a = 0; a < S::TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int grow = row0 + wm * S::WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (grow >= p.M) continue;
#pragma unroll
        for (int b = 0; b < S::TN; ++b) {
          const int gcol = col0 + wn * S::WN + b * 32 + li;
          if (gcol < p.N) C[(size_t)grow * p.ldc + gcol] = acc[a][b][r];
        }
      }
    return;
  }
  constexpr bool WIDE_OK = FAST && (2 * BUF >= (NTHREADS / 64) * SCR_FLOATS);
  double* red = reinterpret_cast<double*>(smem);
  const int kind = (p.gather_i[g] ? 1 : 0) | (p.resid[g] ? 2 : 0) | (p.dact[g] ? 4 : 0) |
                   (p.colsum[g] ? (p.colsq[g] ? 16 : 8) : 0) | (p.cpre[g] ? 32 : 0) | (p.out_act ? 64 : 0) |
                   (p.dact_kind ? 256 : 0);
  if (WIDE_OK && fl.wide) {
#define CN_EPIW(K) epilogue_wide<BN, K>(p, acc, g, row0, col0, tile_m, wm, wn, lane, tid, smem, kind)
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
    return;
  }
#define CN_EPI(K) epilogue<BN, K>(p, acc, g, row0, col0, tile_m, wm, wn, li, lh, tid, red, kind)
  switch (kind) {               // the combinations the CartNet path uses get branch-free code
    case 0: CN_EPI(0); break;    // (+bias) store                          node projections, heads, 2nd-Linear sender
    case 1: CN_EPI(1); break;    // gather node terms                      layer GEMM1
    case 16: CN_EPI(16); break;  // BatchNorm statistics                   layer GEMM2 (gate)
    case 96: CN_EPI(96); break;  // keep pre-activation, SiLU              encoders
    case 2: CN_EPI(2); break;    // + resid                                dE, dX
    case 4: CN_EPI(4); break;    // * silu'(pre)
    case 12: CN_EPI(12); break;  // * silu'(pre), bias gradient            dpre
    case 14: CN_EPI(14); break;  // + resid, * silu'(pre), bias gradient   layer 0 into the encoders
    default: CN_EPI(-1); break;  // anything else: flags read at run time
  }
#undef CN_EPI
}

// gemm_x3.hip
void launch_x3nn(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st);
void launch_x3tn(bool b_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st);
// gemm_f32.hip
void launch_f32nn(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st);
void launch_f32tn(bool b_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st);
// gemm_f32w128.hip: 128-wide DMA-fed fp32 kernel, chosen by use_f32nn128 (gemm.hip)
void launch_f32nn128(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st);
bool use_f32nn128(const CartnetGemmArgs& a);
#ifdef CN_EXPERIMENTAL_Q
// experimental/gemm_f32q.hip (not in the product build): 128 x 128 tiles on 256-thread workgroups, four per CU
void launch_f32nnq(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st);
bool use_f32nnq(const CartnetGemmArgs& a);
#endif
// gemm_f32p.hip: the persistent kernel (one workgroup per CU, two accumulator sets; K = 256 / 512)
bool use_f32p(const CartnetGemmArgs& a);
void launch_f32p(const CartnetGemmArgs& a, hipStream_t st);
// gemm_f32ao.hip: the a_act form of the 256-wide kernel that also writes silu(A) (CartnetGemmArgs.a_act_out)
void launch_f32nn_actout(const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st);
// gemm_x3s.hip: the bf16x3 activation x weight kernel on the 16x16x32 MFMA shape (precision 1; the 32x32x16 kernel of
// gemm_x3.h serves precision 2, one MFMA product per block)
void launch_x3nn16(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st);
// gemm_x3ao.hip: the same for the pre-split bf16x3 / bf16 kernel
void launch_x3nn_actout(const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st);
// gemm_h.hip: precision 2 with operands / output stored as bf16 (CartnetGemmArgs.*_half); false = combination not compiled
bool launch_hnn(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st);
bool launch_htn(bool b_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st);
inline bool any_half(const CartnetGemmArgs& a) { return a.a_half || a.b_half || a.c_half || a.dact_half; }
extern thread_local bool g_half_launched;     // set by launch_variant when a half-storage kernel took the launch

// K tail of a split-K weight gradient on the DMA-fed kernels (the < 16 rows behind the last whole K-step) as the LAST slab:
// out[m, n] = sum_k A[k, m] (silu?)(B[k, n]), a few thousand FMAs per output row.  As a launch of the checked MFMA kernel
// -- two workgroups with the register and LDS footprint of a GEMM -- it could not share a CU with the resident
// weight-gradient workgroups (one per CU, 144 VGPRs) and waited for one of them to finish: up to 0.34 ms at the end of the
// step's main stream.  This one fits anywhere.
template <bool B_ACT>
__global__ __launch_bounds__(256) void cn_gemm_tn_tail_kernel(const CartnetGemmArgs p, int k_lo, int split) {
  const int g = blockIdx.y;
  const float* __restrict__ A = p.A[g];
  const float* __restrict__ B = p.B[g];
  float* __restrict__ C = p.C[g] + (size_t)split * p.M * p.ldc;
  const unsigned n4 = (unsigned)p.N / 4u, total = (unsigned)p.M * n4;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const int m = (int)(i / n4), n = (int)(i % n4) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = k_lo; k < p.K; ++k) {
      const float a = A[(size_t)k * p.lda + m];
      f32x4 b = ldv4(B + (size_t)k * p.ldb + n);
      if (B_ACT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = fast_silu(b[q]);
      }
      acc += a * b;
    }
    stv4(C + (size_t)m * p.ldc + n, acc);
  }
}

template <bool A_KS, bool B_KS, int BN, bool A_ACT, bool B_ACT>
void launch_variant(const CartnetGemmArgs& a, GemmFlags fl, hipStream_t st) {
  const int tiles_n = cn_ceil_div(a.N, BN);
  const int full_m = a.M / BM, rag_m = (a.M % BM) ? 1 : 0;
  auto launch = [&](auto fast_tag, int m0, int nm, int s0, int ns, int k_lo, int k_hi, int kchunk) {
    constexpr bool FAST = decltype(fast_tag)::value;
    if (nm <= 0 || ns <= 0) return;
    fl.tile_m0 = m0;
    fl.split0 = s0;
    fl.k_lo = k_lo;
    fl.k_hi = k_hi;
    fl.kchunk = kchunk;
    // bf16x3 kernels exist for the 256-wide tile and the operand layouts of the model's big GEMMs
    // (weight-gradient GEMMs, A_KS, stay on the fp32 MFMA: their k-strided operands need a transposing LDS fill that
    //  measured 2.4x slower than the fp32 kernel)
    constexpr bool X3_OK = FAST && BN == 256 && B_KS && !A_KS;
    if constexpr (X3_OK) {
      if (fl.x3) {
        // second-generation kernel (gemm_x3.h) when the weight operand comes pre-split and the epilogue is vectorisable
        bool presplit = fl.wide && a.splitk == 1 && m0 == 0 && a.nsegs == 1 && (double)a.M * a.lda * 4.0 < 4294967296.0;
        const int nptr = a.ngroups > 1 ? a.ngroups : a.nsegs;
        for (int i = 0; i < nptr; ++i) presplit = presplit && a.b_split[i] != nullptr;
        if (presplit && fl.x3 == 2 && any_half(a)) {
          g_half_launched = launch_hnn(A_ACT, a, fl, dim3(nm * tiles_n, ns, a.ngroups), st);
          return;
        }
        if (presplit) {
          if (A_ACT && a.a_act_out[0]) launch_x3nn_actout(a, fl, dim3(nm * tiles_n, ns, a.ngroups), st);
          else launch_x3nn(A_ACT, a, fl, dim3(nm * tiles_n, ns, a.ngroups), st);
          return;
        }
        if (fl.x3 == 1 && !any_half(a)) {   // first-generation kernel (both operands split in flight); precision 2 has no such form
          hipLaunchKernelGGL((cn_gemm_kernel<A_KS, B_KS, BN, A_ACT, B_ACT, true, 1>), dim3(nm * tiles_n, ns, a.ngroups),
                             dim3(NTHREADS), 0, st, a, fl);
          return;
        }
      }
    }
    // fp32 MFMA with a pre-packed weight operand: second-generation kernel (gemm_f32.h)
    if constexpr (FAST && BN == 256 && B_KS && !A_KS) {
      if (!fl.x3) {
        bool prepacked = fl.wide && a.splitk == 1 && m0 == 0 && a.nsegs == 1 && (double)a.M * a.lda * 4.0 < 4294967296.0;
        const int nptr = a.ngroups > 1 ? a.ngroups : a.nsegs;
        for (int i = 0; i < nptr; ++i) prepacked = prepacked && a.b_split[i] != nullptr;
#ifdef CN_P_WHY
        if (!prepacked && a.M >= 100000)
          fprintf(stderr, "f32p NOIMG: M=%d N=%d K=%d groups=%d segs=%d wide=%d splitk=%d m0=%d img0=%d\n", a.M, a.N, a.K, a.ngroups,
                  a.nsegs, (int)fl.wide, a.splitk, m0, a.b_split[0] != nullptr);
#endif
        if (prepacked) {
          if (use_f32p(a)) launch_f32p(a, st);
          else if (A_ACT && a.a_act_out[0]) launch_f32nn_actout(a, fl, dim3(nm * tiles_n, ns, a.ngroups), st);
#ifdef CN_EXPERIMENTAL_Q
          else if (use_f32nnq(a)) launch_f32nnq(A_ACT, a, fl, dim3(nm * tiles_n * 2, ns, a.ngroups), st);
#endif
          else if (use_f32nn128(a)) launch_f32nn128(A_ACT, a, fl, dim3(nm * tiles_n * 2, ns, a.ngroups), st);
          else launch_f32nn(A_ACT, a, fl, dim3(nm * tiles_n, ns, a.ngroups), st);
          return;
        }
      }
    }
    if (any_half(a)) return;      // bf16 operands: only the half-storage kernels may read them (cartnet_gemm reports it)
    hipLaunchKernelGGL((cn_gemm_kernel<A_KS, B_KS, BN, A_ACT, B_ACT, FAST, 0>), dim3(nm * tiles_n, ns, a.ngroups),
                       dim3(NTHREADS), 0, st, a, fl);
  };
  auto round_up = [](int v) { return ((v + BK - 1) / BK) * BK; };
  // weight gradients at precision 1 / 2: the transposing-read kernel takes every row tile (also a ragged last one)
  // over the whole K-steps; a K tail (< 16 rows) is one more slab from the checked fp32 kernel
  if constexpr (A_KS && B_KS && BN == 256 && !A_ACT) {
    // precision 0: the all-DMA fp32 kernel (gemm_f32.h); a ragged last row tile needs M % 4 == 0 (its lanes clamp)
    // (SiLU on the B operand: the five-stage instance that activates its DMA'd tiles in place in LDS, gemm_f32.h -- 10 %
    //  slower than a plain operand; on the fragments it was 15 % and the register-staged kernel 2-3x)
    if (fl.x3 == 2 && any_half(a)) {
      // half-storage weight gradient: one kernel over all of K (it masks a ragged last K-step itself), S slabs
      if (a.M % 4 == 0 && fl.vecA && fl.vecB && a.nsegs == 1 && a.N % BN == 0 && a.M > 0 && a.K >= 1 &&
          (a.splitk > 1 || fl.wide)) {
        fl.tile_m0 = 0; fl.split0 = 0; fl.k_lo = 0; fl.k_hi = a.K;
        fl.kchunk = round_up(cn_ceil_div(a.K, a.splitk));
        g_half_launched = launch_htn(B_ACT, a, fl, dim3(cn_ceil_div(a.M, BM) * tiles_n, a.splitk, a.ngroups), st);
      }
      return;
    }
#ifdef CN_TN_BACT_STAGED    /* A/B builds: SiLU-on-B weight gradients at precision 0 on the register-staged kernel */
    const bool tn_ok = a.M % 4 == 0 && (fl.x3 || !B_ACT);
#else
    const bool tn_ok = a.M % 4 == 0;
#endif
    auto launch_tn = [&](dim3 grid) {
      if (fl.x3) launch_x3tn(B_ACT, a, fl, grid, st);
      else launch_f32tn(B_ACT, a, fl, grid, st);
    };
    if (tn_ok && fl.vecA && fl.vecB && a.nsegs == 1 && a.N % BN == 0 && a.M > 0 && a.K >= BK) {
      const int K16x = (a.K / BK) * BK, tailx = a.K - K16x;
      const int tiles_mx = cn_ceil_div(a.M, BM);
      if (a.splitk == 1 && tailx == 0 && fl.wide) {
        fl.tile_m0 = 0; fl.split0 = 0; fl.k_lo = 0; fl.k_hi = a.K; fl.kchunk = K16x;
        launch_tn(dim3(tiles_mx * tiles_n, 1, a.ngroups));
        return;
      }
      if (a.splitk > 1) {
        const int nfastx = tailx ? a.splitk - 1 : a.splitk;
        fl.tile_m0 = 0; fl.split0 = 0; fl.k_lo = 0; fl.k_hi = K16x;
        fl.kchunk = round_up(cn_ceil_div(K16x, nfastx));
        launch_tn(dim3(tiles_mx * tiles_n, nfastx, a.ngroups));
        if (tailx) {
          const long long items = (long long)a.M * (a.N / 4);
          const int blocks = (int)(items / 256 + 1 > 2048 ? 2048 : items / 256 + 1);
          hipLaunchKernelGGL((cn_gemm_tn_tail_kernel<B_ACT>), dim3(blocks, a.ngroups), dim3(256), 0, st, a, K16x, nfastx);
        }
        return;
      }
    }
  }
  // the predicate-free kernel covers ragged row tiles too when A is k-contiguous (row clamp in Stager::load)
  const bool fast_ok = fl.vecA && fl.vecB && (a.N % BN == 0) && (full_m > 0 || !A_KS) && a.M > 0;
  const int fast_m = A_KS ? full_m : full_m + rag_m;      // row tiles the predicate-free kernel takes
  const int slow_m = (full_m + rag_m) - fast_m;
  const int K16 = (a.K / BK) * BK, tail = a.K - K16;
  if (a.splitk == 1) {
    const int kc = round_up(a.K > 0 ? a.K : 1);
    if (fast_ok && tail == 0 && a.K > 0) {
      launch(std::true_type{}, 0, fast_m, 0, 1, 0, a.K, kc);
      launch(std::false_type{}, fast_m, slow_m, 0, 1, 0, a.K, kc);
    } else {
      launch(std::false_type{}, 0, full_m + rag_m, 0, 1, 0, a.K, kc);
    }
    return;
  }
  // split-K: slabs 0..nfast-1 cover whole K-steps of [0, K16) (predicate-free kernel); if K has a < 16-row tail it
  // becomes the last slab, computed by the checked kernel over just those rows.
  const int nfast = tail ? a.splitk - 1 : a.splitk;
  const int kc = round_up(cn_ceil_div(K16 > 0 ? K16 : 1, nfast));
  if (fast_ok) {
    launch(std::true_type{}, 0, fast_m, 0, nfast, 0, K16, kc);
    launch(std::false_type{}, fast_m, slow_m, 0, nfast, 0, K16, kc);
  } else {
    launch(std::false_type{}, 0, full_m + rag_m, 0, nfast, 0, K16, kc);
  }
  if (tail) launch(std::false_type{}, 0, full_m + rag_m, nfast, 1, K16, a.K, BK);
}

// Operand layouts / fused activations the CartNet path uses.  Returns false for an unsupported combination.
template <int BN>
bool launch_bn(const CartnetGemmArgs& a, const GemmFlags& fl, hipStream_t st) {
  const int combo = (a.a_kstrided ? 1 : 0) | (a.b_kstrided ? 2 : 0) | (a.a_act ? 4 : 0) | (a.b_act ? 8 : 0);
  switch (combo) {
    case 0: launch_variant<false, false, BN, false, false>(a, fl, st); return true;   // Y = X W^T
    case 4: launch_variant<false, false, BN, true, false>(a, fl, st); return true;    // Y = silu(X) W^T
    case 2: launch_variant<false, true, BN, false, false>(a, fl, st); return true;    // dX = dY W ; Y = X Wt
    case 6: launch_variant<false, true, BN, true, false>(a, fl, st); return true;     // Y = silu(X) Wt
    case 3: launch_variant<true, true, BN, false, false>(a, fl, st); return true;     // dW = dY^T X
    case 11: launch_variant<true, true, BN, false, true>(a, fl, st); return true;     // dW = dY^T silu(X)
    default: return false;
  }
}

}  // namespace cn_gemm
