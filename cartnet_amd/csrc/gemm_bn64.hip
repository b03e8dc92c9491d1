// Instantiations of the fp32 MFMA GEMM for 64-column output tiles (split per tile width to compile in parallel).
#include "gemm_kernel.h"

namespace cn_gemm {
template bool launch_bn<64>(const CartnetGemmArgs&, const GemmFlags&, hipStream_t);
}
