// bf16x3 second-generation kernels: instantiations, launchers, and the weight pre-split (cartnet_gemm_split_b).
#include "gemm_x3.h"

namespace cn_gemm {

void launch_x3nn(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st) {
  const bool one = fl.x3 == 2;   // precision 2: plain bf16 operands, one MFMA product
  if (!one) {      // precision 1: the six piece products on the 16x16x32 MFMA shape (gemm_x3s.h; round 3: -1...-4 % per launch)
    launch_x3nn16(a_act, a, fl, grid, st);
    return;
  }
  if (a_act) hipLaunchKernelGGL((cn_gemm_x3nn_kernel<true, true>), grid, dim3(NTHREADS), 0, st, a, fl);
  else hipLaunchKernelGGL((cn_gemm_x3nn_kernel<false, true>), grid, dim3(NTHREADS), 0, st, a, fl);
}

// (Round 4: padding these to one workgroup per CU, as cn_gemm_f32tn_kernel is by its four stages, measured no gain at
//  bf16x3 -- 10.48-10.51 vs 10.48 ms -- and a loss with bf16 storage, 6.41 vs 6.34 ms: not done.)
void launch_x3tn(bool b_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st) {
  const bool one = fl.x3 == 2;
  if (b_act) {
    if (one) hipLaunchKernelGGL((cn_gemm_x3tn_kernel<true, true>), grid, dim3(NTHREADS), 0, st, a, fl);
    else hipLaunchKernelGGL((cn_gemm_x3tn_kernel<true, false>), grid, dim3(NTHREADS), 0, st, a, fl);
  } else {
    if (one) hipLaunchKernelGGL((cn_gemm_x3tn_kernel<false, true>), grid, dim3(NTHREADS), 0, st, a, fl);
    else hipLaunchKernelGGL((cn_gemm_x3tn_kernel<false, false>), grid, dim3(NTHREADS), 0, st, a, fl);
  }
}

}  // namespace cn_gemm

namespace {

constexpr int SPLIT_MAX_JOBS = 80;     // 2.5 KB of kernel arguments: the model's 66 images go in one launch

struct SplitJobs {
  const float* src[SPLIT_MAX_JOBS];
  char* dst[SPLIT_MAX_JOBS];
  int K[SPLIT_MAX_JOBS], N[SPLIT_MAX_JOBS], sk[SPLIT_MAX_JOBS], sn[SPLIT_MAX_JOBS];
};

// One thread per (column n, 8 consecutive k): three 16-byte pieces of the LDS image the NN kernel copies verbatim.
__global__ __launch_bounds__(256) void cn_split_b_kernel(const SplitJobs jobs) {
  const int j = blockIdx.y;
  const int K = jobs.K[j], N = jobs.N[j];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * (K / 8)) return;
  const int n = idx % N, k0 = (idx / N) * 8;
  const float* __restrict__ src = jobs.src[j] + (size_t)n * jobs.sn[j] + (size_t)k0 * jobs.sk[j];
  const int sk = jobs.sk[j];
  f32x4 v0, v1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v0[i] = src[(size_t)i * sk];
    v1[i] = src[(size_t)(i + 4) * sk];
  }
  const cn_gemm::Split3 s0 = cn_gemm::split3(v0), s1 = cn_gemm::split3(v1);
  const int t = k0 / cn_gemm::BK, khalf = (k0 >> 3) & 1;
  const int tile_n = n / cn_gemm::X3_BN, nl = n % cn_gemm::X3_BN;
  char* d = jobs.dst[j] + ((size_t)tile_n * (K / cn_gemm::BK) + t) * cn_gemm::X3_B_BYTES + cn_gemm::x3_offset(nl, khalf);
  *reinterpret_cast<cn_gemm::bf16x4*>(d) = s0.h;
  *reinterpret_cast<cn_gemm::bf16x4*>(d + 8) = s1.h;
  *reinterpret_cast<cn_gemm::bf16x4*>(d + cn_gemm::X3_B_PLANE) = s0.m;
  *reinterpret_cast<cn_gemm::bf16x4*>(d + cn_gemm::X3_B_PLANE + 8) = s1.m;
  *reinterpret_cast<cn_gemm::bf16x4*>(d + 2 * cn_gemm::X3_B_PLANE) = s0.l;
  *reinterpret_cast<cn_gemm::bf16x4*>(d + 2 * cn_gemm::X3_B_PLANE + 8) = s1.l;
}

}  // namespace

extern "C" size_t cartnet_gemm_split_b_bytes(int32_t K, int32_t N) {
  if (K <= 0 || N <= 0 || K % cn_gemm::BK != 0 || N % cn_gemm::X3_BN != 0) return 0;
  return (size_t)K * N * 6;
}

extern "C" int cartnet_gemm_split_b(const float* const* src, void* const* dst, const int32_t* K, const int32_t* N,
                                    const int32_t* stride_k, const int32_t* stride_n, int32_t njobs, void* stream) {
  CN_CHECK(src && dst && K && N && stride_k && stride_n && njobs >= 0, "cartnet_gemm_split_b: bad arguments");
  for (int j0 = 0; j0 < njobs; j0 += SPLIT_MAX_JOBS) {
    SplitJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    const int n = njobs - j0 < SPLIT_MAX_JOBS ? njobs - j0 : SPLIT_MAX_JOBS;
    int max_units = 0;
    for (int j = 0; j < n; ++j) {
      const int i = j0 + j;
      CN_CHECK(src[i] && dst[i], "cartnet_gemm_split_b: null pointer in job %d", i);
      CN_CHECK(K[i] > 0 && N[i] > 0 && K[i] % cn_gemm::BK == 0 && N[i] % cn_gemm::X3_BN == 0,
               "cartnet_gemm_split_b: job %d: K=%d must be a multiple of %d and N=%d of %d", i, K[i], cn_gemm::BK, N[i],
               cn_gemm::X3_BN);
      CN_CHECK((reinterpret_cast<uintptr_t>(dst[i]) & 15u) == 0, "cartnet_gemm_split_b: job %d: dst must be 16-byte aligned", i);
      jobs.src[j] = src[i];
      jobs.dst[j] = static_cast<char*>(dst[i]);
      jobs.K[j] = K[i];
      jobs.N[j] = N[i];
      jobs.sk[j] = stride_k[i];
      jobs.sn[j] = stride_n[i];
      const int units = N[i] * (K[i] / 8);
      if (units > max_units) max_units = units;
    }
    hipLaunchKernelGGL(cn_split_b_kernel, dim3(cn_ceil_div(max_units, 256), n), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), jobs);
    CN_LAUNCH_CHECK("cartnet_gemm_split_b");
  }
  return 0;
}

#ifdef CN_CLOCK_STAMP
// diagnostic build: copies this translation unit's stamp buffer out (4096 pairs of 64-bit counters)
extern "C" int cartnet_debug_clock_x3(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_gemm::cn_clock_dbg), sizeof(unsigned long long) * 2 * 4096);
}
#endif
