// Persistent fp32-MFMA activation x weight kernel (gemm_f32p.h): instantiations, eligibility, launcher.
// Own translation unit: 32 hand-placed K-steps per instantiation, up to 256 registers per lane.
#include "gemm_f32p.h"

namespace cn_gemm {

// One explicit instantiation per compiled form, split over two translation units (each form is ~25 s of hipcc)
template <bool A_ACT, bool ACT_OUT, int KIND, int NS>
void p_launch(const CartnetGemmArgs& a, int grid, hipStream_t st) {
  hipLaunchKernelGGL((cn_gemm_f32p_kernel<A_ACT, ACT_OUT, KIND, NS>), dim3(grid), dim3(NTHREADS), 0, st, a,
                     (a.M + BM - 1) / BM);
}
#define CN_P_FORMS_A(X)                                                                             \
  X(false, false, 0, 16)   /* plain (+ bias) */                                                     \
  X(true, false, 0, 16)    /* silu(A), bias */                                                      \
  X(true, false, 16, 16)   /* layer GEMM 2: silu(A), bias, BatchNorm statistics */                  \
  X(true, true, 16, 16)    /* ... + silu(A) written */                                              \
  X(false, false, 4, 16)   /* dpre: * silu'(pre) */                                                 \
  X(false, false, 12, 16)  /* ... + bias gradient (column sums): the encoder's dhe, iComformer's dpr */
#define CN_P_FORMS_B(X)                                                                             \
  X(true, true, 0, 16)     /* iComformer's second Linears: silu(A), bias, silu(A) written */        \
  X(false, false, 352, 16) /* iComformer's RBF branches: pre kept, softplus(pre) out */             \
  X(true, true, 96, 32)    /* the edge encoder's second Linear: K = 512, pre kept, silu out */      \
  X(false, false, 2, 32)   /* K = 512 (two folded segments) + residual */                          \
  X(false, false, 1, 16)   /* the layer's first product: bias + node terms gathered by target / source atom */
#define CN_P_FORMS_C(X)                                                                             \
  X(false, false, 0, 32)   /* K = 512 (two folded segments), plain */                               \
  X(false, false, 8, 16)   /* + column sums (a bias gradient) */                                    \
  X(false, false, 16, 16)  /* + BatchNorm statistics, no activation on A */                         \
  X(false, false, 268, 32) /* K = 512: * sigmoid(pre) (softplus') + bias gradient */                \
  X(false, false, 2, 48)   /* K = 768 (three folded segments) + residual */                        \
  X(false, false, 14, 32)  /* K = 512 + residual, * silu'(pre), bias gradient: the encoder end of CartNet's backward */ \
  X(false, false, 270, 32) /* ... the softplus family's (iComformer) */                              \
  X(false, false, 530, 32) /* CartNet's dE: K = 512 (+ residual), the gate statistics of the layer below (gst_*) */
#define CN_P_EXTERN(AA, AO, KD, NSV) extern template void p_launch<AA, AO, KD, NSV>(const CartnetGemmArgs&, int, hipStream_t);
#define CN_P_DEFINE(AA, AO, KD, NSV) template void p_launch<AA, AO, KD, NSV>(const CartnetGemmArgs&, int, hipStream_t);
#if defined(CN_P_UNIT_B)
CN_P_FORMS_B(CN_P_DEFINE)
#elif defined(CN_P_UNIT_C)
CN_P_FORMS_C(CN_P_DEFINE)
#else
CN_P_FORMS_B(CN_P_EXTERN)
CN_P_FORMS_C(CN_P_EXTERN)
CN_P_FORMS_A(CN_P_DEFINE)
#endif

#if !defined(CN_P_UNIT_B) && !defined(CN_P_UNIT_C)
namespace {
int p_kind(const CartnetGemmArgs& a) {
  if (a.gst_g) return 2 | 16 | 512;     // (one form, with or without the residual: a missing one reads zeros)
  return (a.gather_i[0] ? 1 : 0) | (a.resid[0] ? 2 : 0) | (a.dact[0] ? 4 : 0) | (a.colsum[0] ? (a.colsq[0] ? 16 : 8) : 0) | (a.cpre[0] ? 32 : 0) |
         (a.out_act ? 64 : 0) | (a.dact_kind ? 256 : 0);
}

// the compiled forms: (a_act, a_act_out, kind, K / 16)
typedef void (*PLaunch)(const CartnetGemmArgs&, int, hipStream_t);
PLaunch p_find(bool a_act, bool act_out, int kind, int ns) {
#define CN_P(AA, AO, KD, NSV) \
  if (a_act == AA && act_out == AO && kind == KD && ns == NSV) return &p_launch<AA, AO, KD, NSV>;
  CN_P_FORMS_A(CN_P)
  CN_P_FORMS_B(CN_P)
#ifndef CN_P_NO_C         /* A/B builds: the third unit's forms stay on the second-generation kernels */
  CN_P_FORMS_C(CN_P)
#endif
#undef CN_P
  return nullptr;
}
}  // namespace

// A launch the persistent kernel takes.  The caller (launch_variant) has already established the DMA-fed path: precision 0,
// weight images for every group, 16-byte aligned rows, one K-segment, no split-K, full 256-column tiles.
bool use_f32p(const CartnetGemmArgs& a) {
  // tile_policy: 3 forces this kernel (any size it has the form for), 128 / 256 exclude it; 0 (and 1: the iComformer path's
  // grouped products) take it from 4 tiles per CU up.  CN_F32P_DEFAULT (A/B builds): 0 never, 1 policy 0 only, 2 both.
  // Same-box A B A B (profiles/r06_exp_f32p_step_ab.txt): CartNet's step 13.68-13.73 -> 13.46-13.47 ms, iComformer's
  // 27.31-27.34 -> 26.75-27.04 ms.
#ifndef CN_F32P_DEFAULT
#define CN_F32P_DEFAULT 2
#endif
#ifdef CN_P_WHY    /* diagnostic build: which edge-sized launches stay on the second-generation kernels, and why */
  if (a.M >= 100000 && a.tile_policy != 128 && a.tile_policy != 256) {
    const int subs = (a.N / F32_BN) * a.ngroups;
    const bool form = p_find(a.a_act != 0, a.a_act_out[0] != nullptr, p_kind(a), a.K / BK) != nullptr;
    const bool taken = (a.K == 256 || a.K == 512 || a.K == 768) && a.N % F32_BN == 0 && 32 % (subs ? subs : 1) == 0 && form;
    fprintf(stderr, "f32p %s: M=%d N=%d K=%d groups=%d policy=%d a_act=%d act_out=%d kind=%d gather=%d gst=%d resid=%d dact=%d form=%d\n",
            taken ? "TAKEN" : "LEFT", a.M, a.N, a.K, a.ngroups, a.tile_policy, a.a_act, a.a_act_out[0] != nullptr, p_kind(a),
            a.gather_i[0] != nullptr, a.gst_g != nullptr, a.resid[0] != nullptr, a.dact[0] != nullptr, (int)form);
  }
#endif
  if (a.tile_policy == 128 || a.tile_policy == 256) return false;
  if (a.tile_policy == 0 && CN_F32P_DEFAULT < 1) return false;
  if (a.tile_policy == 1 && CN_F32P_DEFAULT < 2) return false;
  if (a.K != 256 && a.K != 512 && a.K != 768) return false;
  if (a.N % F32_BN != 0 || a.ngroups * a.N > P_BIAS_FLOATS || a.M < 2) return false;
  if (32 % ((a.N / F32_BN) * a.ngroups) != 0) return false;   // a workgroup keeps one (group, column tile): 32 slots per XCD
  const long long tiles = (long long)((a.M + BM - 1) / BM) * (a.N / F32_BN) * a.ngroups;
  if (tiles < 1024 && a.tile_policy != 3) return false;      // fewer than 4 tiles per CU: the 2,768-workgroup kernels fill the chip as well
  // the gate-statistics form: what cartnet_gemm admits for gst_* (one group, no bias / dact / cpre / out_act), with an envelope
  if (a.gst_g && (!a.gst_env || a.ngroups != 1 || a.bias[0] || a.dact[0] || a.cpre[0] || a.out_act || a.a_act || !a.colsum[0] ||
                  !a.colsq[0] || (double)((double)a.M + 384.0) * a.gst_ld * 4.0 >= 4294967296.0))
    return false;
#ifdef CN_P_NO_GATHER     /* A/B builds: the gather launches stay on the second-generation kernels */
  if (a.gather_i[0]) return false;
#endif
  if (a.gather_i[0] && (a.gather_rows <= 0 || !a.tgt || !a.src || (double)a.gather_rows * a.ldg * 4.0 >= 4294967296.0))
    return false;                                              // (the gather form's 32-bit offsets into the node-term tables)
  // every byte offset is 32 bits, every buffer descriptor's record count too
  const double lim = 4294967296.0;
  // (+ 384 rows: "the tile past the last row" and the rows of a slice must not wrap around)
  const double rows = (double)a.M + 384.0;
  if (rows * a.ldc * 4.0 >= lim || rows * a.lda * 4.0 >= lim) return false;
  if (a.resid[0] && rows * a.ldr * 4.0 >= lim) return false;
  if (a.dact[0] && rows * a.ldd * 4.0 >= lim) return false;
  for (int g = 1; g < a.ngroups; ++g)       // one epilogue form per launch (statistics may be missing for a group: dropped)
    if ((a.resid[g] != nullptr) != (a.resid[0] != nullptr) || (a.dact[g] != nullptr) != (a.dact[0] != nullptr) ||
        (a.cpre[g] != nullptr) != (a.cpre[0] != nullptr) || (a.colsum[g] && !a.colsum[0]) || (a.colsq[g] && !a.colsq[0]) ||
        (a.a_act_out[g] != nullptr) != (a.a_act_out[0] != nullptr) || (a.gather_i[g] != nullptr) != (a.gather_i[0] != nullptr))
      return false;
  for (int g = 0; g < a.ngroups; ++g)
    if ((a.gather_i[g] != nullptr) != (a.gather_j[g] != nullptr)) return false;
  return p_find(a.a_act != 0, a.a_act_out[0] != nullptr, p_kind(a), a.K / BK) != nullptr;
}

void launch_f32p(const CartnetGemmArgs& a, hipStream_t st) {
  PLaunch f = p_find(a.a_act != 0, a.a_act_out[0] != nullptr, p_kind(a), a.K / BK);
  // one workgroup per CU (116 KB of LDS each: two cannot share one), 32 per XCD
  f(a, 256, st);
}

#endif  // first unit

}  // namespace cn_gemm

#if defined(CN_P_STAMP) && !defined(CN_P_UNIT_B) && !defined(CN_P_UNIT_C)
// diagnostic build: the per-workgroup, per-tile stamps of the last launch (256 x 16 x 4 64-bit words)
extern "C" int cartnet_debug_p_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_gemm::cn_p_dbg), sizeof(unsigned long long) * 256 * 16 * 4);
}
extern "C" int cartnet_debug_p_waves(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_gemm::cn_p_dbg_wave), sizeof(unsigned long long) * 256 * 8 * 2);
}
#endif
