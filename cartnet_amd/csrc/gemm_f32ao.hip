// Instantiation of the activated-operand-writing fp32 kernel (gemm_f32ao.h), own translation unit.
#include "gemm_f32ao.h"

namespace cn_gemm {

void launch_f32nn_actout(const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL(cn_gemm_f32nn_actout_kernel, grid, dim3(NTHREADS), 0, st, a, fl);
}

}  // namespace cn_gemm
