// Instantiations of the four-workgroups-per-CU fp32 kernel (gemm_f32q.h), own translation unit.
#include "gemm_f32q.h"

namespace cn_gemm {

static int g_use_q = 0;
bool use_f32nnq(const CartnetGemmArgs& a) {
  if (g_use_q == 0) return false;
  if (g_use_q == 2) return a.gather_i[0] == nullptr;     // gather launches stay on the 128-wide kernel
  return true;
}

void launch_f32nnq(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st) {
  if (a_act) hipLaunchKernelGGL((cn_gemm_f32nnq_kernel<true>), grid, dim3(Q_NT), 0, st, a, fl);
  else hipLaunchKernelGGL((cn_gemm_f32nnq_kernel<false>), grid, dim3(Q_NT), 0, st, a, fl);
}

}  // namespace cn_gemm

// 0: off (default), 1: every prepacked activation x weight product, 2: all but the node-term gather launches
extern "C" int cartnet_gemm_experimental_q(int mode) {
  cn_gemm::g_use_q = mode;
  return 0;
}

#ifdef CN_PHASE_STAMP
// diagnostic build: the per-workgroup phase stamps of the last launches (8192 x 8 64-bit words)
extern "C" int cartnet_debug_phase_f32q(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_gemm::cn_phase_dbg), sizeof(unsigned long long) * 8 * 8192);
}
extern "C" int cartnet_debug_phase2_f32q(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_gemm::cn_phase_dbg2), sizeof(unsigned long long) * 4 * 8192);
}
#endif

#ifdef CN_CLOCK_STAMP
extern "C" int cartnet_debug_clock_f32q(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_gemm::cn_clock_dbg), sizeof(unsigned long long) * 2 * 4096);
}
#endif
