// Instantiations of the 128-wide DMA-fed fp32 kernel (gemm_f32w128.h), own translation unit.
#include "gemm_f32w128.h"

namespace cn_gemm {

void launch_f32nn128(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st) {
  // (cartnet_gemm checked a launch with gst_g: one group, no a_act)
  if (a.gst_g && a.resid[0]) hipLaunchKernelGGL((cn_gemm_f32nn128_kernel<false, 1>), grid, dim3(NTHREADS), 0, st, a, fl);
  else if (a.gst_g) hipLaunchKernelGGL((cn_gemm_f32nn128_kernel<false, 2>), grid, dim3(NTHREADS), 0, st, a, fl);
  else if (a_act) hipLaunchKernelGGL((cn_gemm_f32nn128_kernel<true>), grid, dim3(NTHREADS), 0, st, a, fl);
  else hipLaunchKernelGGL((cn_gemm_f32nn128_kernel<false>), grid, dim3(NTHREADS), 0, st, a, fl);
}

}  // namespace cn_gemm
