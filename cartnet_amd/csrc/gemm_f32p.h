// Third-generation fp32-MFMA kernel for activations x weights (CartnetGemmArgs.precision == 0 with a pre-packed weight
// operand, K = 256, 512 or 768): PERSISTENT, one 512-thread workgroup per CU, TWO accumulator sets.
//
// What the second generation (gemm_f32.h) loses at K = 256 is outside its main loop (profiles/r03_exp_phases.md: a lone
// loop runs at 0.97 of the matrix pipe, the launch at 0.79-0.80): prologue, epilogue and hand-over of 2,768 short-lived
// workgroups, two per CU, whose epilogues cost the partner's loop 6 %.  Here a workgroup owns its CU (140 KB of LDS, up to
// 256 registers per lane) and walks its 10-11 tiles in a loop:
//   * the operand rings simply continue across the tile boundary -- weight and activation tiles of K-step s are DMA'd three
//     (four) steps ahead -- so tile i+1's first K-steps are in flight while tile i finishes: no prologue per tile;
//   * tile i's epilogue runs INSIDE tile i+1's MFMA chain, from the other accumulator set, one slice of 4 accumulator
//     registers per K-step (16 slices = 16 K-steps), in the ACCUMULATOR layout: a register of a 32x32 block is two
//     128-byte row segments, so every store is a full-line dword access; an epilogue operand (residual, pre-activation,
//     gathered node terms) of a slice is 8 consecutive rows x 128 B = ONE 16-byte-per-lane DMA into a wave-private LDS ring,
//     read back in the accumulator layout; rows past M are dropped by the buffer descriptor's range check (no predicates:
//     the number of memory operations per step is a compile-time constant, which the counted waits below rely on);
//   * fragments of k-group 0 of step u+1 are read under step u's MFMAs, so a wave leaves the barrier with its first 16
//     MFMAs ready to issue.
// Every vector-memory operation of the loop is inline asm and counted by hand: loads, stores and DMAs retire through ONE
// in-order counter, so "the tiles of step u+2 have landed" is s_waitcnt vmcnt(N(u)) with N(u) = the operations issued after
// those DMAs, a constexpr of the step's position in the tile (PCount).  What must have been ACKNOWLEDGED at that wait are
// the stores of step u-2 and older, 1.5+ K-steps (2.5 us) after their issue (measured: ~43 cycles per step in the wait).
// NO load of the loop has a register destination: tiles AND epilogue operands travel by LDS-DMA (the operands into a
// wave-private ring), so nothing the register allocator does -- a spill, a copy, a reuse after the last use -- can meet a
// load in flight.  (The first version kept the activation tile and the operands in register rings: correct while it did
// not spill, garbage from the forms that did.)
//
// Step u of a tile (MODE FULL), program order; vector-memory operations marked *:
//   F1 <- fragments (u, k-group 1)
//   MFMA group (kg 0, j 0)        * B(u+3) DMA x2, * A(u+3 | u+4) DMA    (their stages' last readers passed barrier(u-1))
//                                 [* statistics of the tile before the previous one (step 0)]
//   MFMA (0,1)   [A_ACT: own 16 bytes of A(u+2) back from LDS]   * epilogue operand DMAs for slice u+3
//   MFMA (0,2)   [A_ACT: SiLU, ds_write in place, * silu(A(u+2)) store]
//   MFMA (0,3)   F0 <- fragments (u+1, k-group 0); operands of slice u from the wave's ring
//   MFMA (1,0)   slice u of the PREVIOUS tile: arithmetic, * 2 stores
//   MFMA (1,1)   * 2 stores      MFMA (1,2)  [column statistics to LDS]     MFMA (1,3)
//   s_waitcnt vmcnt(N(u)) lgkmcnt(0); s_barrier
// The gather form (KIND & 1: the layer's first product, v += node terms of the edge's two atoms) has two operands per value:
// 8 row-gathering DMAs per slice, issued TWO steps ahead and BEFORE the step's tile DMAs, into a three-slot ring (48 KB; a
// fourth slot would not fit), whose slot index therefore rotates at run time; the tile's atom indices arrive by two DMAs at
// its step 0 in a wave-private, double-buffered 1 KB of LDS.
// Measured (profiles/r06_exp_f32p_*.txt): 4,380-4,395 cycles per K-step against 4,096 of matrix work (0.935) -- DMA issue
// ~35, slices ~80, fragment reads ~22, one barrier per step ~150 -- at a clock the chip lowers as the pipe fills (2.28-2.39
// GHz; 2.19 with the barriers compiled out): the plain two-group layer product 335-352 us against 359-380 (same boxes).
#pragma once
#include "gemm_f32.h"

namespace cn_gemm {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int P_STEPS = 16;                                     // K-steps that carry an epilogue slice (= slices per tile)
constexpr int P_A_BYTES = BM * BK * 4;                          // [128 rows][64 B], XOR-swizzled like the weight image
constexpr int P_BIAS_FLOATS = 1024;
constexpr int P_LDS_B = 0;                                      // 4 x 16 KB weight images (all within a 16-bit ds offset)
constexpr int P_LDS_A = 4 * F32_B_BYTES;                        // 4 (8 with SiLU on A) x 8 KB activation tiles
enum { P_FULL = 0, P_PRIME = 1, P_DRAIN = 2 };

__device__ __forceinline__ i32x4 p_make_srd(const void* ptr, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)(ptr ? bytes : 0u), 0x00020000};
}

// LDS accesses through a 32-bit base address held in ONE register + a compile-time offset (the 16-bit offset field of the
// ds instructions): addressed from the array's symbol, every (stage, k-group, block) combination beyond the first 64 KB got
// a register of its own -- 44 of them live across a whole tile.
typedef __attribute__((address_space(3))) f32x4 p_lds_f32x4;
typedef __attribute__((address_space(3))) float p_lds_f32;
__device__ __forceinline__ f32x4 p_lds_ld4(unsigned base, int off) { return *reinterpret_cast<p_lds_f32x4*>((unsigned long)(base + off)); }
__device__ __forceinline__ void p_lds_st4(unsigned base, int off, f32x4 v) { *reinterpret_cast<p_lds_f32x4*>((unsigned long)(base + off)) = v; }
__device__ __forceinline__ float p_lds_ld1(unsigned base, int off) { return *reinterpret_cast<p_lds_f32*>((unsigned long)(base + off)); }

template <int I, int N, class F>
__device__ __forceinline__ void p_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    p_static_for<I + 1, N>(f);
  }
}

// Vector-memory operations of step u of a tile, in issue order: [B DMA x2][A DMA][statistics store / IX index DMAs (step 0)]
// [EL operand DMAs of slice (u+LA) mod NS][silu(A) store][ES stores of slice u]; with ELF the operand DMAs come FIRST (the
// gather form: LA = 2, so they must be among "everything up to the three DMAs" of their step).  Slices ride on the first 16
// steps of a tile.
template <int EL, int ES, bool AO, bool STATS, int NS, int LA = 3, bool ELF = false, int IX = 0>
struct PCount {
  static constexpr int el_at(int u) { return (((u + LA) % NS) < P_STEPS) ? EL : 0; }
  static constexpr int es(int u) { return ((u % NS) < P_STEPS) ? ES : 0; }
  static constexpr int st(int u) { return ((u % NS) == 0) ? (STATS ? 1 : 0) + IX : 0; }
  static constexpr int tot(int u) { return 3 + st(u) + el_at(u) + (AO ? 1 : 0) + es(u); }
  // s_waitcnt at the end of step u: everything up to the three DMAs of step u-1 has completed
  static constexpr int wait_count(int u) {
    return tot((u + NS - 1) % NS) - 3 - (ELF ? el_at((u + NS - 1) % NS) : 0) + tot(u % NS);
  }
  // drain (slices only, [EL operand DMAs of slice u+LA][ES stores of slice u] per step): the operands of slice u, issued in
  // step u-LA
  static constexpr int del(int w) { return (w + LA < P_STEPS) ? EL : 0; }
  static constexpr int drain_count(int u) {
    if (u < LA) return 0;
    int n = LA * ES;
    for (int w = u - LA + 1; w <= u; ++w) n += del(w);
    return n;
  }
};

#ifdef CN_P_STAMP
// Diagnostic build (tools/exp_f32p_stamps.py): per workgroup and tile [shader clock at the tile's end, cycles wave 0 spent in
// the counted waits, 100 MHz time, cycles in the barriers]; per wave the two totals; nothing else reads the buffers.
static __device__ unsigned long long cn_p_dbg[256 * 16 * 4];
static __device__ unsigned long long cn_p_dbg_wave[256 * 8 * 2];
#endif

// KIND: epilogue bits as in gemm_kernel.h (2 resid, 4 dact, 8 column sums, 16 column sums + squares (fp64), 32 cpre,
// 64 out_act, 256 softplus family) + 1: node-term gather (v += gather_i[tgt[m]] + gather_j[src[m]]) + 512: the gate
// statistics of CartNet's dE product (CartnetGemmArgs.gst_*: with 2 | 16 -- the residual may be missing, its descriptor then
// reads zeros).  NS: K-steps per tile (16: K = 256; 32: K = 512 -- the
// second 16 steps carry no slice).
template <bool A_ACT, bool ACT_OUT, int KIND, int NS>
__global__ __launch_bounds__(NTHREADS, 2) void cn_gemm_f32p_kernel(const CartnetGemmArgs p, const int tiles_m) {
  using S = Shape<F32_BN>;
  constexpr bool RESID = (KIND & 2) != 0, DACT = (KIND & 4) != 0, SUM1 = (KIND & 8) != 0, SUM2 = (KIND & 16) != 0,
                 CPRE = (KIND & 32) != 0, OUTACT = (KIND & 64) != 0;
  constexpr bool GATHER = (KIND & 1) != 0, GST = (KIND & 512) != 0;
  constexpr bool STATS = SUM1 || SUM2;
  // operand DMAs per slice: one per operand -- a slice is 8 consecutive rows x 128 B of the tile in every block of the
  // accumulator layout, i.e. ONE 16-byte-per-lane DMA (lane l: row l / 8, 16-byte piece l % 8) per operand
  // Two operands (the gather form's node terms; resid AND dact): a four-slot ring would not fit, so LA = 2, three slots whose
  // index rotates at run time, and the operand DMAs of a step in front of its tile DMAs (PCount, ELF)
  constexpr bool TWOOP = (RESID && DACT) || GST;
  constexpr bool ROT = GATHER || TWOOP;
  constexpr int EL = ROT ? 2 : ((RESID || DACT) ? 1 : 0);
  constexpr int LA = ROT ? 2 : 3;                               // ... issued this many steps before the slice
  constexpr int NSLOT = LA + 1;                                 // slots of the wave's operand ring (EL x 1 KB each)
  constexpr int SLOT_BYTES = EL * 1024;
  constexpr int ES = 4 + (CPRE ? 4 : 0);                        // stores per slice
  constexpr int NA = A_ACT ? 8 : 4;                             // stages of the activation ring
  constexpr int AD = A_ACT ? 4 : 3;                             // ... and how far ahead its DMA runs (the SiLU pass needs a step)
  static_assert(NS == 16 || NS == 32 || NS == 48, "K = 256, 512 or 768");
  static_assert(!(A_ACT && EL > 0), "the eight activation stages and the operand ring do not fit together");
  static_assert(!ACT_OUT || A_ACT, "silu(A) is written where it is computed");
  static_assert(!GATHER || (!RESID && !DACT && !STATS && !CPRE && NS == 16), "the gather form: bias + node terms (+ out_act)");
  static_assert(!TWOOP || !CPRE, "resid + dact: no kept pre-activation");
  static_assert(!GST || (RESID && SUM2 && !DACT && !CPRE && !OUTACT && !A_ACT && NS == 32), "the gate-statistics form: K = 512, resid, two sums");
  using Cnt = PCount<EL, ES, ACT_OUT, STATS, NS, LA, ROT, GATHER ? 2 : (GST ? 1 : 0)>;
  static_assert(Cnt::wait_count(0) <= 63 && Cnt::wait_count(1) <= 63, "vmcnt is a 6-bit counter");
  constexpr int LDS_BIAS = P_LDS_A + NA * P_A_BYTES;            // bias[g][n] of every group, staged once
  constexpr int LDS_RED = LDS_BIAS + P_BIAS_FLOATS * 4;         // double red[2][2][256]: column statistics of one tile
  constexpr int LDS_OPR = LDS_RED + (STATS ? 2 * 2 * 256 * 8 : 0);   // per wave NSLOT slots x EL x 1 KB: epilogue operands in flight
  constexpr int LDS_IDX = LDS_OPR + 8 * NSLOT * SLOT_BYTES;    // gather: per wave [tile parity][tgt | src][64 rows]
  // (the gate-statistics form keeps the envelope of the tile's rows there: per wave [tile parity][64 rows])
  constexpr int LDS_BYTES = LDS_IDX + (GATHER ? 8 * 1024 : (GST ? 8 * 512 : 0));
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");

  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / S::WGN, wn = wid % S::WGN;
  const int li = lane & 31, lh = lane >> 5;

  // ---- this workgroup's tiles: blocks b and b + 8 share an XCD (speed only); XCD x owns one eighth of the row tiles and
  // its workgroups walk them in order, the `subs` = column tiles x groups of a row tile side by side (same A rows, same L2).
  // nslot % subs == 0 (host): a workgroup keeps ONE (group, column tile) for all its tiles -- descriptors, weight image, bias
  // and column offsets are workgroup constants, a tile is its first row
  const int tiles_n = p.N / F32_BN;
  const int subs = tiles_n * p.ngroups;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int own = tiles_m > xcd ? (tiles_m - xcd + 7) / 8 : 0;
  // ... a CONTIGUOUS range of row tiles per XCD (own = tiles_m / 8, one more on the first tiles_m % 8): the rows a crystal's
  // edges gather from the node-term tables then pass through ONE L2 instead of eight (gather form: 651 -> 504 MB fetched per
  // launch against the interleaved ranges x, x + 8, ...; launch times equal)
  const int xcd_base = xcd * (tiles_m / 8) + min(xcd, tiles_m % 8);
  const int per_xcd_slots = nslot / subs;                    // workgroups of this XCD that share `sub`
  const int wslot = slot / subs;
  const int n_my = own > wslot ? (own - wslot + per_xcd_slots - 1) / per_xcd_slots : 0;
  if (n_my == 0) return;
  int sub = slot % subs, g = sub / tiles_n, tile_n = sub - g * tiles_n, col0 = tile_n * F32_BN;
  const float* a_base = p.A[g];
  const char* b_base = reinterpret_cast<const char*>(p.b_split[g]) + (size_t)tile_n * NS * F32_B_BYTES;
  const unsigned cbytes = ((unsigned)(p.M - 1) * (unsigned)p.ldc + (unsigned)p.N) * 4u;
#ifdef CN_P_NOSTORE       /* timing only: every epilogue store is dropped by its descriptor (issued, counted) */
  i32x4 c_srd = p_make_srd(nullptr, 0);
#else
  i32x4 c_srd = p_make_srd(p.C[g], cbytes);
#endif
  i32x4 p_srd = p_make_srd(CPRE ? p.cpre[g] : nullptr, cbytes);
  const int ldo = RESID ? p.ldr : p.ldd;                        // the epilogue operand (resid or dact; with both: resid)
  i32x4 o_srd = p_make_srd(RESID ? p.resid[g] : (DACT ? p.dact[g] : nullptr), ((unsigned)(p.M - 1) * (unsigned)ldo + (unsigned)p.N) * 4u);
  const int ld2 = GST ? p.gst_ld : p.ldd;                       // the second operand: dact, or the gate pre-activation
  i32x4 d_srd = p_make_srd(TWOOP ? (GST ? p.gst_g : p.dact[g]) : nullptr, ((unsigned)(p.M - 1) * (unsigned)ld2 + (unsigned)p.N) * 4u);
  i32x4 e_srd = p_make_srd(GST ? p.gst_env : nullptr, (unsigned)p.M * 4u);     // envelope: rows past M read 0 and drop out of the sums
  i32x4 h_srd = p_make_srd(ACT_OUT ? p.a_act_out[g] : nullptr, ((unsigned)(p.M - 1) * (unsigned)p.lda + (unsigned)p.K) * 4u);
  // gather: the two node-term tables (gather_rows bounds both: use_f32p) and the edge -> atom index arrays; an index of a row
  // past M reads 0 and the index slots start as zeros, so every gathered row is one the caller vouches for (or row 0)
  const unsigned gbytes = GATHER ? ((unsigned)(p.gather_rows - 1) * (unsigned)p.ldg + (unsigned)p.N) * 4u : 0u;
  i32x4 gi_srd = p_make_srd(GATHER ? p.gather_i[g] : nullptr, gbytes);
  i32x4 gj_srd = p_make_srd(GATHER ? p.gather_j[g] : nullptr, gbytes);
  i32x4 t_srd = p_make_srd(GATHER ? p.tgt : nullptr, (unsigned)p.M * 4u);
  i32x4 s_srd = p_make_srd(GATHER ? p.src : nullptr, (unsigned)p.M * 4u);
  // waves 0-3 write the column sums, waves 4-7 the squares (a missing statistic: null descriptor, the store is dropped)
  i32x4 st_srd = p_make_srd(STATS ? ((wid >> 2) ? (SUM2 ? p.colsq[g] : nullptr) : p.colsum[g]) : nullptr,
                            (unsigned)tiles_m * (unsigned)p.N * 8u);
  int bias_idx = g * p.N + col0;

  struct Ctx {
    int row0;                        // first row of the tile
    unsigned a_voff;                 // per lane: this thread's 16 bytes of the tile's K-step 0
  };
  // this thread's share of an activation tile: wave w moves rows 16w..16w+15 as one lane-linear KB -- lane l lands in row
  // 16w + l/4, 16-byte slot l%4 -- so it FETCHES the k-quad that slot holds under the image's XOR swizzle (f32_swz), and
  // the fragment reads are the weight image's.  Rows past M are clamped (their products are computed and dropped).
  const int arow = tid >> 2, akq = (tid & 3) ^ ((arow >> 2) & 3);
  auto make_ctx = [&](int i) {
    Ctx c;
    const int k = wslot + per_xcd_slots * min(i, n_my - 1);   // past the end: the last tile again (prefetches stay in bounds)
#ifdef CN_P_XCD_INTERLEAVED      /* A/B builds: XCD x owns the row tiles x, x + 8, ... */
    c.row0 = (k * 8 + xcd) * BM;
#else
    c.row0 = (xcd_base + k) * BM;
#endif
    c.a_voff = ((unsigned)min(c.row0 + arow, p.M - 1) * (unsigned)p.lda + akq * 4) * 4u;
    return c;
  };
  // A tile past the last row: every epilogue access of it fails its descriptor's range check -- loads return 0, stores are
  // dropped, all of them are issued and counted (the epilogue of "the tile before the first")
  auto null_epi = [&](Ctx c) {
    c.row0 = tiles_m * BM;
    c.a_voff = (unsigned)p.M * (unsigned)p.lda * 4u;
    return c;
  };

  // ---- per-lane constants
  const unsigned lv_c = (unsigned)((wm * S::WM + 4 * lh) * p.ldc + col0 + wn * S::WN + li) * 4u;     // this lane's element of a block
  const unsigned lv_o = (unsigned)((wm * S::WM + (lane >> 3)) * ldo + col0 + wn * S::WN + (lane & 7) * 4) * 4u;   // operand DMAs: row l / 8, piece l % 8
  const int lane_row = wm * S::WM + 4 * lh;
  const unsigned ldc4 = (unsigned)p.ldc * 4u, ldo4 = (unsigned)ldo * 4u;
  const unsigned lv_d = (unsigned)((wm * S::WM + (lane >> 3)) * ld2 + col0 + wn * S::WN + (lane & 7) * 4) * 4u, ldd4 = (unsigned)ld2 * 4u;
  const unsigned lv_g = (unsigned)(col0 + wn * S::WN + (lane & 7) * 4) * 4u, ldg4 = (unsigned)p.ldg * 4u;   // gather: this lane's piece of a row
  const unsigned lv_ix = (unsigned)(wm * S::WM + lane) * 4u;    // ... and its row of the wave's 64 in the index arrays
  const unsigned b_voff = lane * 16, b_voff2 = lane * 16 + 8192;
  unsigned lds_bw = lds0 + P_LDS_B + wid * 1024;
  unsigned lds_aw = lds0 + P_LDS_A + wid * 1024;
  unsigned lds_ow = lds0 + LDS_OPR + wid * (NSLOT * SLOT_BYTES);
  unsigned lds_iw = lds0 + LDS_IDX + wid * (GST ? 512 : 1024);
  unsigned env_base = lds0 + LDS_IDX + wid * 512 + lh * 16;      // gate statistics: the envelope of this lane's four rows of a slice
  asm volatile("" : "+v"(env_base));
  const char* b_base_w = b_base + wid * 1024;
  // one base register per (region, k-group): the swizzle's XOR makes the two k-groups of a fragment differ by more than a constant
  unsigned fa0 = lds0 + P_LDS_A + f32_swz(wm * S::WM + li, 0 + lh), fa1 = lds0 + P_LDS_A + f32_swz(wm * S::WM + li, 2 + lh);
  unsigned fb0 = lds0 + P_LDS_B + f32_swz(wn * S::WN + li, 0 + lh), fb1 = lds0 + P_LDS_B + f32_swz(wn * S::WN + li, 2 + lh);
  unsigned own_base = lds0 + P_LDS_A + tid * 16;                 // this thread's own 16 bytes of an activation stage
  unsigned opr_base = lds0 + LDS_OPR + wid * (NSLOT * SLOT_BYTES) + lh * 512 + li * 4;   // operand slot [8 rows][32 columns]: rows 4lh.., column li
  unsigned idx_base = lds0 + LDS_IDX + wid * 1024 + (lane >> 3) * 4;     // the row this lane gathers, within an 8-row group of the index arrays
  asm volatile("" : "+v"(fa0), "+v"(fa1), "+v"(fb0), "+v"(fb1), "+v"(own_base), "+v"(opr_base), "+v"(idx_base));
  // gather: the operand ring has three slots and a tile sixteen slices -- the slot being filled rotates at run time (scalar)
  unsigned slot_w = 0;
  float* sbias = reinterpret_cast<float*>(lds + LDS_BIAS);
  double* red = reinterpret_cast<double*>(lds + LDS_RED);

  for (int i = tid; i < p.ngroups * p.N; i += NTHREADS) {
    const int gg = i / p.N, n = i - gg * p.N;
    sbias[i] = p.bias[gg] ? p.bias[gg][n] : 0.f;
  }
  if constexpr (GATHER) {
    for (int i = tid; i < 8 * 256; i += NTHREADS) reinterpret_cast<int*>(lds + LDS_IDX)[i] = 0;
    __syncthreads();
  }
  if constexpr (GST) {
    // no bias in this form (use_f32p): its region holds, per column of the tile, rstd | -mean rstd | gamma | beta, the last
    // two times -log2(e) (the gate's sigmoid through exp2) -- the second-generation kernel's constants (gemm_kernel.h)
    __syncthreads();
    for (int n = tid; n < F32_BN; n += NTHREADS) {
      const float rs = p.gst_mean_rstd[p.N + col0 + n];
      sbias[n] = rs;
      sbias[256 + n] = -p.gst_mean_rstd[col0 + n] * rs;
      sbias[512 + n] = p.gst_gamma[col0 + n] * -1.44269504088896f;
      sbias[768 + n] = p.gst_beta[col0 + n] * -1.44269504088896f;
    }
    for (int i = tid; i < 8 * 128; i += NTHREADS) reinterpret_cast<float*>(lds + LDS_IDX)[i] = 0.f;
  }

  f32x16 acc[2][2][2];            // [set][block row][block column]; a tile's first MFMAs start from 0: never initialised
  f32x4 af[2][2], bf[2][2];       // fragments [k-group][block]
  float bv = 0.f;
  float g_rstd = 0.f, g_mean = 0.f, g_gam = 0.f, g_bet = 0.f, gs1 = 0.f, gs2 = 0.f;     // gate statistics: this lane's column
  int rows_in = 0;                // rows of the tile in its epilogue that exist, seen from this lane's first row
  double cs = 0.0, cq = 0.0;
  float csf = 0.f;
#ifdef CN_P_STAMP
  unsigned long long stall_w = 0, stall_b = 0;
#endif

  // ---- vector-memory operations (all inline asm: counted by hand; no instruction offset on an LDS-DMA: it is added to the
  // LDS address as well as to the global one).  Step offsets go through an empty asm: sixteen loop-invariant source addresses
  // would otherwise be hoisted into 32 SGPRs (the first build spilled them); the LDS destination is an immediate added to
  // this wave's base in the same statement.
  auto a_dma = [&](const Ctx& c, auto st_c, auto stage_c) {
    unsigned off = decltype(st_c)::value * BK * 4;
    asm volatile("" : "+s"(off));
    const char* base = reinterpret_cast<const char*>(a_base) + off;
    constexpr int DST = decltype(stage_c)::value * P_A_BYTES;
    const unsigned la = lds_aw;
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(la), "v"(c.a_voff), "s"(base), "n"(DST) : "memory", "m0", "scc");
  };
  // B tile of K-step `step` of this workgroup's weight image: 16 pieces of 1 KB, wave w moves pieces w and w + 8
  auto b_issue = [&](auto step_c, auto stage_c) {
    unsigned off = decltype(step_c)::value * F32_B_BYTES;
    asm volatile("" : "+s"(off));
    const char* src = b_base_w + off;
    constexpr int DST = decltype(stage_c)::value * F32_B_BYTES;
    const unsigned lb = lds_bw, v1 = b_voff, v2 = b_voff2;     // (asm operands alone do not capture in a generic lambda)
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lb), "v"(v1), "s"(src), "n"(DST) : "memory", "m0", "scc");
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lb), "v"(v2), "s"(src), "n"(DST + 8192) : "memory", "m0", "scc");
  };
  auto h_store = [&](f32x4 v, const Ctx& c, auto st_c) {
    constexpr int OFF = decltype(st_c)::value * BK * 4;
    const i32x4 srd = h_srd;
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:%3 sc0 sc1 nt\n\ts_nop 1"
                 :: "v"(v), "v"(c.a_voff), "s"(srd), "n"(OFF) : "memory");
  };
  // 16 bytes per lane of an epilogue operand into this wave's ring: the 8 rows x 128 B of a slice at slot * 1024 (the
  // 128-byte column offset of block column 1 travels in the scalar offset: global address only)
  auto o_dma = [&](unsigned voff, auto dst_c, auto b_c) {
    constexpr int DST = decltype(dst_c)::value;
    const unsigned lo = lds_ow;
    const i32x4 srd = o_srd;
    const int soff = decltype(b_c)::value * 128;
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds"
                 :: "s"(lo), "v"(voff), "s"(srd), "n"(DST), "s"(soff) : "memory", "m0", "scc");
  };
  // the same from a node-term table (descriptor `srd`), into the slot at byte `lo` of LDS (run-time, scalar)
  auto g_dma = [&](unsigned voff, const i32x4& srd_in, unsigned lo, auto dst_c, auto b_c) {
    constexpr int DST = decltype(dst_c)::value;
    const i32x4 srd = srd_in;
    const int soff = decltype(b_c)::value * 128;
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds"
                 :: "s"(lo), "v"(voff), "s"(srd), "n"(DST), "s"(soff) : "memory", "m0", "scc");
  };
  // the tile's target / source atoms of this wave's 64 rows into the index slot of the tile's parity
  auto idx_dma = [&](const Ctx& c, auto par_c) {
    constexpr int DST = decltype(par_c)::value * 512;
    const unsigned li_ = lds_iw, voff = (unsigned)c.row0 * 4u + lv_ix;
    const i32x4 ts = t_srd, ss = s_srd;
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 :: "s"(li_), "v"(voff), "s"(ts), "n"(DST) : "memory", "m0", "scc");
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 :: "s"(li_), "v"(voff), "s"(ss), "n"(DST + 256) : "memory", "m0", "scc");
  };
  // gate statistics: the envelope of the tile's rows (this wave's 64) into the slot of the tile's parity
  auto env_dma = [&](const Ctx& c, auto par_c) {
    constexpr int DST = decltype(par_c)::value * 256;
    const unsigned li_ = lds_iw, voff = (unsigned)c.row0 * 4u + lv_ix;
    const i32x4 es = e_srd;
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 :: "s"(li_), "v"(voff), "s"(es), "n"(DST) : "memory", "m0", "scc");
  };
  auto st1 = [&](float v, unsigned voff, const i32x4& srd, auto off_c) {
    constexpr int OFF = decltype(off_c)::value;
    asm volatile("buffer_store_dword %0, %1, %2, 0 offen offset:%3" :: "v"(v), "v"(voff), "s"(srd), "n"(OFF) : "memory");
  };
  auto st1_stream = [&](float v, unsigned voff, const i32x4& srd, auto off_c) {
    constexpr int OFF = decltype(off_c)::value;
    asm volatile("buffer_store_dword %0, %1, %2, 0 offen offset:%3 sc0 sc1 nt" :: "v"(v), "v"(voff), "s"(srd), "n"(OFF) : "memory");
  };

  // ---- LDS
  // (a block row / column further is +32 rows of 64 bytes: the swizzle term only depends on (row >> 2) & 3)
  auto frags = [&](int sa, int sb, int kg) {
#pragma unroll
    for (int a = 0; a < 2; ++a) af[kg][a] = p_lds_ld4(kg ? fa1 : fa0, sa * P_A_BYTES + a * 2048);
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[kg][b] = p_lds_ld4(kg ? fb1 : fb0, sb * F32_B_BYTES + b * 2048);
  };

  // ---- epilogue pieces of slice SL = b*8 + a*4 + q: registers 4q..4q+3 of block (a, b) = rows a*32 + 8q + 4lh + j, column
  // b*32 + li (column half b first: one bias value and one pair of column sums live at a time)
  auto slice_loads = [&](auto sl_c, const Ctx& c) {
    constexpr int SL = decltype(sl_c)::value, b = SL >> 3, a = (SL >> 2) & 1, q = SL & 3;
    o_dma(lv_o + ((unsigned)c.row0 + (unsigned)(a * 32 + 8 * q)) * ldo4, std::integral_constant<int, (SL & 3) * 1024>{},
          std::integral_constant<int, b>{});
  };
  // gather, operand K (0: by target, 1: by source) of slice SL of the tile whose indices sit in parity PAR: rows
  // a*32 + 8q .. + 7 of the wave's 64 -> each lane reads the index of ITS row, one row-gathering DMA
  auto gather_loads = [&](auto sl_c, auto par_c, auto k_c) {
    constexpr int SL = decltype(sl_c)::value, PAR = decltype(par_c)::value, K = decltype(k_c)::value;
    constexpr int b = SL >> 3, a = (SL >> 2) & 1, q = SL & 3;
    const int ix = *reinterpret_cast<__attribute__((address_space(3))) int*>(
        (unsigned long)(idx_base + (PAR * 512 + K * 256 + (a * 32 + 8 * q) * 4)));
    const unsigned lo = lds_ow + slot_w;
    const unsigned voff = __umul24((unsigned)ix, ldg4) + lv_g;
    g_dma(voff, K ? gj_srd : gi_srd, lo, std::integral_constant<int, K * 1024>{}, std::integral_constant<int, b>{});
  };
  // resid + dact: operand K (0: resid, 1: dact) of slice SL of tile c into the slot being filled
  auto two_loads = [&](auto sl_c, const Ctx& c, auto k_c) {
    constexpr int SL = decltype(sl_c)::value, K = decltype(k_c)::value, b = SL >> 3, a = (SL >> 2) & 1, q = SL & 3;
    const unsigned rr = (unsigned)c.row0 + (unsigned)(a * 32 + 8 * q);
    g_dma(K ? lv_d + rr * ldd4 : lv_o + rr * ldo4, K ? d_srd : o_srd, lds_ow + slot_w, std::integral_constant<int, K * 1024>{},
          std::integral_constant<int, b>{});
  };
  auto slice_operands = [&](auto sl_c, float (&o)[ROT ? 8 : 4]) {
    constexpr int SL = decltype(sl_c)::value;
    if constexpr (ROT) {
      // the slot filled two steps ago = the one after the slot being filled now
      const unsigned rb = opr_base + (slot_w == 2 * SLOT_BYTES ? 0u : slot_w + SLOT_BYTES);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = p_lds_ld1(rb, (j >> 2) * 1024 + (j & 3) * 128);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = p_lds_ld1(opr_base, (SL & 3) * 1024 + j * 128);
    }
  };
  auto slice_math = [&](auto sl_c, auto set_c, const Ctx& c, const float (&o)[ROT ? 8 : 4], float (&v)[4]) {
    constexpr int SL = decltype(sl_c)::value, SETP = decltype(set_c)::value, b = SL >> 3, a = (SL >> 2) & 1, q = SL & 3;
    if constexpr ((SL & 7) == 0 && GST) {
      const int n = wn * S::WN + b * 32 + li;
      g_rstd = sbias[n];
      g_mean = sbias[256 + n];
      g_gam = sbias[512 + n];
      g_bet = sbias[768 + n];
      gs1 = gs2 = 0.f;
    }
    if constexpr ((SL & 7) == 0 && !GST) {
      bv = sbias[bias_idx + wn * S::WN + b * 32 + li];
      if constexpr (STATS) {
        cs = cq = 0.0;
        csf = 0.f;
        rows_in = p.M - c.row0 - lane_row;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = acc[SETP][a][b][4 * q + j] + bv;
      if constexpr (GATHER) t += o[j] + o[4 + j];
      if constexpr (RESID) t += o[j];
      if constexpr (DACT) t *= (KIND & 256) ? fast_sigmoid(o[TWOOP ? 4 + j : j]) : fast_dsilu(o[TWOOP ? 4 + j : j]);
      v[j] = t;
    }
    if constexpr (GST) {
      // v = de_out of the layer below: its share of that layer's BatchNorm-backward sums, sum(v w) and sum(v w ghat), w = env
      // sigma' with the gate recomputed from the kept pre-activation (the second-generation kernel's arithmetic, fp32 over
      // the lane's 32 rows of a column half).  A row past M has envelope 0 (descriptor) and finite factors: it adds 0.
      const f32x4 ev = p_lds_ld4(env_base, SETP * 256 + (a * 32 + 8 * q) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ghat = o[4 + j] * g_rstd + g_mean;
        const float t = __builtin_amdgcn_exp2f(fminf(ghat * g_gam + g_bet, 126.0f));
        const float r = __builtin_amdgcn_rcpf(1.0f + t);
        const float tv = (v[j] * ev[j]) * ((t * r) * r);
        gs1 += tv;
        gs2 += tv * ghat;
      }
    }
    if constexpr (STATS && !GST) {
      // only the matrix's last row tile (and the tile "past the last row") has rows that do not exist: a scalar branch keeps
      // the 64 row predicates of a tile (64 SGPR pairs the compiler computed up front) out of the common path
      if (c.row0 + BM <= p.M) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (SUM2) {
            const double d = (double)v[j];
            cs += d;
            cq += d * d;
          } else {
            csf += v[j];
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float tm = (a * 32 + 8 * q + j) < rows_in ? v[j] : 0.f;
          if constexpr (SUM2) {
            const double d = (double)tm;
            cs += d;
            cq += d * d;
          } else {
            csf += tm;
          }
        }
      }
    }
  };
  auto slice_stores = [&](auto sl_c, const Ctx& c, float (&v)[4], auto j0_c, auto j1_c) {
    constexpr int SL = decltype(sl_c)::value, b = SL >> 3, a = (SL >> 2) & 1, q = SL & 3;
    constexpr int J0 = decltype(j0_c)::value, J1 = decltype(j1_c)::value;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
      const unsigned voff = lv_c + ((unsigned)c.row0 + (unsigned)(a * 32 + 8 * q + j)) * ldc4;
      if constexpr (CPRE) st1_stream(v[j], voff, p_srd, std::integral_constant<int, b * 128>{});
      float o = v[j];
      if constexpr (OUTACT) o = (KIND & 256) ? fast_softplus(o) : fast_silu(o);
      st1(o, voff, c_srd, std::integral_constant<int, b * 128>{});
    }
  };
  // column statistics of one 32-column half of the finished tile: lane halves, then into red[which][wm][column] (read at the
  // next tile's step 0, behind a barrier)
  auto stats_to_lds = [&](int b) {
    double s = GST ? (double)gs1 : (SUM2 ? cs : (double)csf), q2 = GST ? (double)gs2 : cq;
    s += __shfl_xor(s, 32);
    if constexpr (SUM2) q2 += __shfl_xor(q2, 32);
    if (lh == 0) {
      red[(0 * 2 + wm) * F32_BN + wn * S::WN + b * 32 + li] = s;
      if constexpr (SUM2) red[(1 * 2 + wm) * F32_BN + wn * S::WN + b * 32 + li] = q2;
    }
  };
  // ... and out: thread t -> statistic t / 256 (sums | squares), column t % 256 of the tile; one 8-byte store per thread
  auto stats_flush = [&](int row0) {
    const int which = wid >> 2, c = tid & 255;
    const double v = red[(which * 2 + 0) * F32_BN + c] + red[(which * 2 + 1) * F32_BN + c];
    const unsigned voff = (unsigned)((row0 / BM) * p.N + col0 + c) * 8u;
    const i32x4 srd = st_srd;
    asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" :: "v"(v), "v"(voff), "s"(srd) : "memory");
  };

  // FIRST: the first four MFMAs of a tile take the constant 0 as their C operand -- the epilogue only READS the other
  // accumulator set (a partial write of a 16-register tuple per slice cost copies and spills)
  auto mma4 = [&](auto set_c, auto first_c, int kg, int j) {
    constexpr int SET = decltype(set_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[SET][a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg][a][j], bf[kg][b][j], FIRST ? zero : acc[SET][a][b], 0, 0, 0);
  };

#ifdef CN_P_PRIO          /* experiment: static priority for the younger wave of each SIMD */
  if (wid >= 4) __builtin_amdgcn_s_setprio(CN_P_PRIO);
#endif
  Ctx prev, cur, next;
  int pend_row0 = tiles_m * BM;          // statistics waiting in `red` for their store (tile before `prev`; none yet)

  // One K-step.  U: step within the tile (compile time); SET: accumulator set the MFMAs write (the epilogue reads the other).
  auto step = [&](auto u_c, auto set_c, auto mode_c) {
    constexpr int U = decltype(u_c)::value, SET = decltype(set_c)::value, MODE = decltype(mode_c)::value;
    constexpr bool MAIN = MODE != P_DRAIN;            // main-loop memory operations are issued
    constexpr bool COMPUTE = MODE == P_FULL;          // fragments, MFMAs, waits, barrier
#if defined(CN_P_X) && (CN_P_X & 2)   /* timing only: no epilogue at all (the accumulators leave once, at the end) */
    constexpr bool SLICE = false, LOADS = false;
#else
    constexpr bool SLICE = U < P_STEPS;               // this step carries slice U of the previous tile
    // ... and the operand DMAs of slice (U + LA) % NS (the drain has no next tile)
    constexpr bool LOADS = EL > 0 && ((U + LA) % NS) < P_STEPS && !(MODE == P_DRAIN && U + LA >= P_STEPS);
#endif
    constexpr int SL_LD = (U + LA) % NS;
    using ParL = std::integral_constant<int, (U + LA < NS) ? (SET ^ 1) : SET>;     // gather: parity of that slice's tile
    const Ctx& cA = (U + AD < NS) ? cur : next;       // tile of K-step U + AD
    // tile of K-step U + 2 (the activation tile that gets its SiLU now; nothing real in the priming step)
    const Ctx& cA2 = MODE == P_PRIME ? prev : ((U + 2 < NS) ? cur : next);
    const Ctx& cL = (U + LA < NS) ? prev : cur;       // tile whose slice (U + LA) % NS gets its operands now
    using SetC = std::integral_constant<int, SET>;
    using PrevC = std::integral_constant<int, SET ^ 1>;
    using SlC = std::integral_constant<int, U % P_STEPS>;
    float v[4], o[ROT ? 8 : 4] = {};
    f32x4 own;

    if constexpr (COMPUTE) {
#ifdef CN_P_FLIP
      __builtin_amdgcn_s_setprio(CN_P_FLIP);
#endif
#if !(defined(CN_P_X) && (CN_P_X & 4))
      frags(U & (NA - 1), U & 3, 1);
#endif
      __builtin_amdgcn_sched_barrier(0);
      mma4(SetC{}, std::integral_constant<bool, U == 0>{}, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    auto main_dmas = [&]() {
      b_issue(std::integral_constant<int, (U + 3) % NS>{}, std::integral_constant<int, (U + 3) & 3>{});
      a_dma(cA, std::integral_constant<int, (U + AD) % NS>{}, std::integral_constant<int, (U + AD) & (NA - 1)>{});
      if constexpr (GATHER && (U % NS) == 0) idx_dma(cur, SetC{});
      if constexpr (GST && (U % NS) == 0) env_dma(cur, SetC{});
    };
    // two operands: the operand DMAs of the step come BEFORE its tile DMAs (PCount, ELF), spread over the first MFMA groups
    if constexpr (GATHER && LOADS) gather_loads(std::integral_constant<int, SL_LD % P_STEPS>{}, ParL{}, std::integral_constant<int, 0>{});
    if constexpr (TWOOP && LOADS) two_loads(std::integral_constant<int, SL_LD % P_STEPS>{}, cL, std::integral_constant<int, 0>{});
#if defined(CN_P_X) && (CN_P_X & 1)
    if constexpr (MAIN && !COMPUTE && !ROT) main_dmas();
#else
    if constexpr (MAIN && !ROT) main_dmas();
#endif
    if constexpr (STATS && (U % NS) == 0) {
      if constexpr (MODE != P_PRIME) stats_flush(pend_row0);
      else stats_flush(tiles_m * BM);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (COMPUTE) {
      mma4(SetC{}, std::false_type{}, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (A_ACT) {
        own = p_lds_ld4(own_base, ((U + 2) & (NA - 1)) * P_A_BYTES);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (GATHER && LOADS) gather_loads(std::integral_constant<int, SL_LD % P_STEPS>{}, ParL{}, std::integral_constant<int, 1>{});
    if constexpr (TWOOP && LOADS) two_loads(std::integral_constant<int, SL_LD % P_STEPS>{}, cL, std::integral_constant<int, 1>{});
    if constexpr (!ROT && LOADS) slice_loads(std::integral_constant<int, SL_LD % P_STEPS>{}, cL);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (COMPUTE) {
      mma4(SetC{}, std::false_type{}, 0, 2);
      __builtin_amdgcn_sched_barrier(0);

      // (the SiLU of the activation tile sits behind the step's first MFMA groups: vector work in the first few hundred
      //  cycles after a barrier costs the younger wave of a SIMD its start -- MI355X_MICROARCH.md, two waves per SIMD, item 6)
      if constexpr (A_ACT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) own[c] = fast_silu(own[c]);
        p_lds_st4(own_base, ((U + 2) & (NA - 1)) * P_A_BYTES, own);
      }
    }
    if constexpr (ROT && MAIN) main_dmas();
    if constexpr (COMPUTE && ACT_OUT) h_store(own, cA2, std::integral_constant<int, (U + 2) % NS>{});
    if constexpr (MODE == P_PRIME && ACT_OUT) h_store(f32x4{0.f, 0.f, 0.f, 0.f}, cA2, std::integral_constant<int, (U + 2) % NS>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (COMPUTE) {
      mma4(SetC{}, std::false_type{}, 0, 3);
      __builtin_amdgcn_sched_barrier(0);
#ifdef CN_P_FLIP
      __builtin_amdgcn_s_setprio(0);
#endif
#if !(defined(CN_P_X) && (CN_P_X & 4))
      frags((U + 1) & (NA - 1), (U + 1) & 3, 0);
#endif
      if constexpr (SLICE && EL > 0) slice_operands(SlC{}, o);
      __builtin_amdgcn_sched_barrier(0);
      mma4(SetC{}, std::false_type{}, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MODE == P_DRAIN && STATS && U == 1) {     // every wave has read `red` (step 0) before anyone refills it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if constexpr (SLICE) {
      if constexpr (MODE == P_DRAIN && EL > 0) {
        // no main-loop operations between the slices here: the operands of this slice, three slices' DMAs stay ahead
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cnt::drain_count(U)) : "memory");
        slice_operands(SlC{}, o);
      }
      if constexpr (MODE != P_PRIME) slice_math(SlC{}, PrevC{}, prev, o, v);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = 0.f;
      }
      slice_stores(SlC{}, prev, v, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (COMPUTE) {
      mma4(SetC{}, std::false_type{}, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SLICE) slice_stores(SlC{}, prev, v, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (COMPUTE) {
      mma4(SetC{}, std::false_type{}, 1, 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SLICE && STATS && (U & 7) == 7 && MODE != P_PRIME) stats_to_lds(U >> 3);
    if constexpr (COMPUTE) {
      __builtin_amdgcn_sched_barrier(0);
      mma4(SetC{}, std::false_type{}, 1, 3);
      __builtin_amdgcn_sched_barrier(0);
#ifdef CN_P_STAMP
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef CN_P_X
      asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(Cnt::wait_count(U)) : "memory");
#endif
#ifdef CN_P_STAMP
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
#if !(defined(CN_P_X) && (CN_P_X & 8))
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
#ifdef CN_P_STAMP
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      stall_w += t1 - t0;
      stall_b += t2 - t1;
#endif
    }
    if constexpr (ROT) slot_w = slot_w == 2 * SLOT_BYTES ? 0u : slot_w + SLOT_BYTES;
  };

  // ---- prologue: the first K-steps of the first tile in ONE memory round trip, then the operations of "step -1" in order
  cur = make_ctx(0);
  next = make_ctx(1);
  prev = null_epi(cur);
  b_issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  b_issue(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  p_static_for<0, AD - 1>([&](auto s_c) { a_dma(cur, s_c, s_c); });          // K-steps 0 .. AD-2; step -1 brings K-step AD-1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (A_ACT) {               // K-steps 0 and 1 get their SiLU here, K-step 2 in step 0
    p_static_for<0, 2>([&](auto s_c) {
      f32x4 t = p_lds_ld4(own_base, decltype(s_c)::value * P_A_BYTES);
#pragma unroll
      for (int c = 0; c < 4; ++c) t[c] = fast_silu(t[c]);
      p_lds_st4(own_base, decltype(s_c)::value * P_A_BYTES, t);
      if constexpr (ACT_OUT) h_store(t, cur, s_c);
    });
  }
  {
    // "step -1" = step NS-1 of a tile before the first: its memory operations only (B(2) into stage 2, A(AD-1), epilogue
    // traffic of a tile past the last row), so that the counted waits of the first real steps see the steady-state queue
    Ctx keep_cur = cur, keep_next = next;
    next = cur;                          // K-steps NS+2 .. of the virtual tile = the first tile's
    cur = null_epi(cur);
    step(std::integral_constant<int, NS - 1>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, P_PRIME>{});
    cur = keep_cur;
    next = keep_next;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  frags(0, 0, 0);

  auto tile_body = [&](auto set_c) {
    p_static_for<0, NS>([&](auto u_c) { step(u_c, set_c, std::integral_constant<int, P_FULL>{}); });
  };
  auto advance = [&](int it) {           // after tile `it`: its epilogue is next; statistics of the one before are in `red`
#ifdef CN_P_STAMP
    if (tid == 0 && it < 16) {
      unsigned long long* d = cn_p_dbg + ((size_t)blockIdx.x * 16 + it) * 4;
      d[0] = __builtin_amdgcn_s_memtime();
      d[1] = stall_w;
      d[2] = __builtin_amdgcn_s_memrealtime();
      d[3] = stall_b;
    }
#endif
    if constexpr (STATS) pend_row0 = prev.row0;
    prev = cur;
    cur = next;
    next = make_ctx(it + 2);
  };

  // ---- the tile loop; the last tile's epilogue on its own (the operand DMAs of its slices 0..2 left in the last three
  // steps of the loop).  One drain per accumulator set, each behind its own loop exit: no merge of the two sets' registers.
  auto drain = [&](auto set_c) {
    using SetNext = std::integral_constant<int, decltype(set_c)::value ^ 1>;     // "the set the MFMAs would write"
    p_static_for<0, P_STEPS>([&](auto u_c) { step(u_c, SetNext{}, std::integral_constant<int, P_DRAIN>{}); });
  };
  for (int it = 0;; it += 2) {
    tile_body(std::integral_constant<int, 0>{});
    advance(it);
    if (it + 1 >= n_my) {
      drain(std::integral_constant<int, 0>{});
      break;
    }
    tile_body(std::integral_constant<int, 1>{});
    advance(it + 1);
    if (it + 2 >= n_my) {
      drain(std::integral_constant<int, 1>{});
      break;
    }
  }
  if constexpr (STATS) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stats_flush(prev.row0);
  }
#if defined(CN_P_X) && (CN_P_X & 2)
  {
    float t = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) t += acc[s2][a][b][r];
    if (t == 12345.678f) p.C[0][tid] = t;
  }
#endif
#ifdef CN_P_STAMP
  if (lane == 0) {
    cn_p_dbg_wave[((size_t)blockIdx.x * 8 + wid) * 2] = stall_w;
    cn_p_dbg_wave[((size_t)blockIdx.x * 8 + wid) * 2 + 1] = stall_b;
  }
  if (tid == 0 && n_my < 16) {
    unsigned long long* d = cn_p_dbg + ((size_t)blockIdx.x * 16 + n_my) * 4;
    d[0] = __builtin_amdgcn_s_memtime();
    d[1] = (unsigned long long)n_my;
    d[2] = __builtin_amdgcn_s_memrealtime();
    d[3] = 0;
  }
#endif
}

}  // namespace cn_gemm
