// Instantiations of the activated-operand-writing bf16x3 / bf16 kernel (gemm_x3ao.h), own translation unit.
#include "gemm_x3ao.h"

namespace cn_gemm {

void launch_x3nn_actout(const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st) {
  if (fl.x3 == 2) hipLaunchKernelGGL((cn_gemm_x3nn_actout_kernel<true>), grid, dim3(NTHREADS), 0, st, a, fl);
  else hipLaunchKernelGGL((cn_gemm_x3nn_actout_kernel<false>), grid, dim3(NTHREADS), 0, st, a, fl);
}

}  // namespace cn_gemm
