// Instantiations of the 16x16x32-shape bf16x3 activation x weight kernel (gemm_x3s.h), own translation unit.
#include "gemm_x3s.h"

namespace cn_gemm {

void launch_x3nn16(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st) {
  // (cartnet_gemm checked a launch with gst_g: one group, no a_act)
  if (a.gst_g && a.resid[0]) hipLaunchKernelGGL((cn_gemm_x3nn16_kernel<false, 1>), grid, dim3(NTHREADS), 0, st, a, fl);
  else if (a.gst_g) hipLaunchKernelGGL((cn_gemm_x3nn16_kernel<false, 2>), grid, dim3(NTHREADS), 0, st, a, fl);
  else if (a_act) hipLaunchKernelGGL((cn_gemm_x3nn16_kernel<true>), grid, dim3(NTHREADS), 0, st, a, fl);
  else hipLaunchKernelGGL((cn_gemm_x3nn16_kernel<false>), grid, dim3(NTHREADS), 0, st, a, fl);
}

}  // namespace cn_gemm

#ifdef CN_CLOCK_STAMP
// diagnostic build: copies this translation unit's stamp buffer out (4096 pairs of 64-bit counters)
extern "C" int cartnet_debug_clock_x3s(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_gemm::cn_clock_dbg), sizeof(unsigned long long) * 2 * 4096);
}
#endif

#ifdef CN_PHASE_STAMP
// diagnostic build: the per-workgroup phase stamps of the last launches (8192 x 8 64-bit words)
extern "C" int cartnet_debug_phase_x3s(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_gemm::cn_phase_dbg), sizeof(unsigned long long) * 8 * 8192);
}
#endif
