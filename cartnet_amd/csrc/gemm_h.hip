// Instantiations and launchers of the half-storage precision-2 kernels (gemm_h.h).
#include "gemm_h.h"

namespace cn_gemm {

// which operands live in memory as bf16 is a launch-time choice; the combinations the model uses are compiled
bool launch_hnn(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st) {
  const int key = (a_act ? 1 : 0) | (a.a_half ? 2 : 0) | (a.c_half ? 4 : 0) | (a.dact_half ? 8 : 0);
#define CN_H(KEY, ...) case KEY: hipLaunchKernelGGL((cn_gemm_hnn_kernel<__VA_ARGS__>), grid, dim3(NTHREADS), 0, st, a, fl); return true
  switch (key) {
    CN_H(4, false, false, true, false);        // fp32 A -> bf16 C                      (layer GEMM 1: pre)
    CN_H(3, true, true, false, false);         // silu(bf16 A) -> fp32 C                (layer GEMM 2 reading pre)
    CN_H(7, true, true, true, false);          // silu(bf16 A) -> bf16 C                (layer GEMM 2: pre -> gs)
    CN_H(8, false, false, false, true);        // fp32 A, silu'(bf16) -> fp32 C         (dpre reading pre)
    CN_H(10, false, true, false, true);        // bf16 A, silu'(bf16) -> fp32 C         (dpre reading dgs and pre)
    CN_H(14, false, true, true, true);         // bf16 A, silu'(bf16) -> bf16 C         (dpre: dgs, pre -> dpre)
    CN_H(2, false, true, false, false);        // bf16 A -> fp32 C                      (dE reading dpre)
    default: return false;
  }
#undef CN_H
}

bool launch_htn(bool b_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st) {
  const int key = (b_act ? 1 : 0) | (a.a_half ? 2 : 0) | (a.b_half ? 4 : 0);
#define CN_H(KEY, ...) case KEY: hipLaunchKernelGGL((cn_gemm_htn_kernel<__VA_ARGS__>), grid, dim3(NTHREADS), 0, st, a, fl); return true
  switch (key) {
    CN_H(5, true, false, true);                // fp32 dY, silu(bf16 X)                 (dW2 reading pre)
    CN_H(7, true, true, true);                 // bf16 dY, silu(bf16 X)                 (dW2 reading dgs and pre)
    CN_H(2, false, true, false);               // bf16 dY, fp32 X                       (dW1e reading dpre)
    CN_H(6, false, true, true);                // bf16 dY, bf16 X
    CN_H(4, false, false, true);               // fp32 dY, bf16 X
    default: return false;
  }
#undef CN_H
}

}  // namespace cn_gemm
