// fp32-MFMA second-generation kernel: instantiations, launcher, and the weight packing (cartnet_gemm_pack_b).
#include "gemm_f32.h"

namespace cn_gemm {

void launch_f32nn(bool a_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st) {
  if (a_act) hipLaunchKernelGGL((cn_gemm_f32nn_kernel<true>), grid, dim3(NTHREADS), 0, st, a, fl);
  else hipLaunchKernelGGL((cn_gemm_f32nn_kernel<false>), grid, dim3(NTHREADS), 0, st, a, fl);
}

void launch_f32tn(bool b_act, const CartnetGemmArgs& a, const GemmFlags& fl, dim3 grid, hipStream_t st) {
  // SiLU on the B operand (dW = dY^T silu(X) without a kept silu(X)): the five-stage instance that activates its tiles in
  // place in LDS (gemm_f32.h); the fragment-side form of round 2 measured 15 % slower and is gone
  if (b_act) hipLaunchKernelGGL((cn_gemm_f32tn_kernel<true, 5>), grid, dim3(NTHREADS), 0, st, a, fl);
  else hipLaunchKernelGGL((cn_gemm_f32tn_kernel<false>), grid, dim3(NTHREADS), 0, st, a, fl);
}

}  // namespace cn_gemm

namespace {

constexpr int PACK_MAX_JOBS = 80;     // 2.5 KB of kernel arguments: the model's 66 images go in one launch

struct PackJobs {
  const float* src[PACK_MAX_JOBS];
  char* dst[PACK_MAX_JOBS];
  int K[PACK_MAX_JOBS], N[PACK_MAX_JOBS], sk[PACK_MAX_JOBS], sn[PACK_MAX_JOBS];
};

// One thread per (column n, 4 consecutive k): the LDS image of the fp32 kernels (gemm_f32.h: 64-byte rows, swizzled
// 16-byte slots).
__global__ __launch_bounds__(256) void cn_pack_b_kernel(const PackJobs jobs) {
  const int j = blockIdx.y;
  const int K = jobs.K[j], N = jobs.N[j];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * (K / 4)) return;
  const int n = idx % N, k0 = (idx / N) * 4;
  const float* __restrict__ src = jobs.src[j] + (size_t)n * jobs.sn[j] + (size_t)k0 * jobs.sk[j];
  const int sk = jobs.sk[j];
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = src[(size_t)i * sk];
  const int t = k0 / cn_gemm::BK, kk = k0 % cn_gemm::BK;
  const int tile_n = n / cn_gemm::F32_BN, nl = n % cn_gemm::F32_BN;
  char* d = jobs.dst[j] + ((size_t)tile_n * (K / cn_gemm::BK) + t) * cn_gemm::F32_B_BYTES +
            cn_gemm::f32_swz(nl, kk >> 2);
  *reinterpret_cast<f32x4*>(d) = v;
}

}  // namespace

extern "C" size_t cartnet_gemm_pack_b_bytes(int32_t K, int32_t N) {
  if (K <= 0 || N <= 0 || K % cn_gemm::BK != 0 || N % cn_gemm::F32_BN != 0) return 0;
  return (size_t)K * N * 4;
}

extern "C" int cartnet_gemm_pack_b(const float* const* src, void* const* dst, const int32_t* K, const int32_t* N,
                                   const int32_t* stride_k, const int32_t* stride_n, int32_t njobs, void* stream) {
  CN_CHECK(src && dst && K && N && stride_k && stride_n && njobs >= 0, "cartnet_gemm_pack_b: bad arguments");
  for (int j0 = 0; j0 < njobs; j0 += PACK_MAX_JOBS) {
    PackJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    const int n = njobs - j0 < PACK_MAX_JOBS ? njobs - j0 : PACK_MAX_JOBS;
    int max_units = 0;
    for (int j = 0; j < n; ++j) {
      const int i = j0 + j;
      CN_CHECK(src[i] && dst[i], "cartnet_gemm_pack_b: null pointer in job %d", i);
      CN_CHECK(K[i] > 0 && N[i] > 0 && K[i] % cn_gemm::BK == 0 && N[i] % cn_gemm::F32_BN == 0,
               "cartnet_gemm_pack_b: job %d: K=%d must be a multiple of %d and N=%d of %d", i, K[i], cn_gemm::BK, N[i],
               cn_gemm::F32_BN);
      CN_CHECK((reinterpret_cast<uintptr_t>(dst[i]) & 15u) == 0, "cartnet_gemm_pack_b: job %d: dst must be 16-byte aligned", i);
      jobs.src[j] = src[i];
      jobs.dst[j] = static_cast<char*>(dst[i]);
      jobs.K[j] = K[i];
      jobs.N[j] = N[i];
      jobs.sk[j] = stride_k[i];
      jobs.sn[j] = stride_n[i];
      const int units = N[i] * (K[i] / 4);
      if (units > max_units) max_units = units;
    }
    hipLaunchKernelGGL(cn_pack_b_kernel, dim3(cn_ceil_div(max_units, 256), n), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), jobs);
    CN_LAUNCH_CHECK("cartnet_gemm_pack_b");
  }
  return 0;
}

#ifdef CN_CLOCK_STAMP
// diagnostic build: copies this translation unit's stamp buffer out (4096 pairs of 64-bit counters)
extern "C" int cartnet_debug_clock_f32(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_gemm::cn_clock_dbg), sizeof(unsigned long long) * 2 * 4096);
}
#endif

#ifdef CN_PHASE_STAMP
// diagnostic build: the per-workgroup phase stamps of the last launches (8192 x 8 64-bit words)
extern "C" int cartnet_debug_phase_f32(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cn_gemm::cn_phase_dbg), sizeof(unsigned long long) * 8 * 8192);
}
#endif

#ifdef CN_TN_STAMP
// diagnostic build: the weight-gradient kernel's stamps of the last launch
extern "C" int cartnet_debug_tn_stamps(unsigned long long* wg, unsigned long long* waves) {
  if (hipMemcpyFromSymbol(wg, HIP_SYMBOL(cn_gemm::cn_tn_dbg), sizeof(unsigned long long) * 1024 * 4) != hipSuccess) return 1;
  return (int)hipMemcpyFromSymbol(waves, HIP_SYMBOL(cn_gemm::cn_tn_dbg_wave), sizeof(unsigned long long) * 1024 * 8 * 2);
}
#endif
