// Fused Adam over one flat fp32 buffer (all 2.5 M CartNet parameters in a single launch).
// Semantics of torch.optim.Adam with weight_decay = 0, amsgrad = False (reference: main.py:208).
#include "common.h"
#include <math.h>

namespace {
__global__ void cn_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                               float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                               float bc1, float bc2_sqrt, float gscale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = m[i] + (1.0f - b1) * (gi - m[i]);          // lerp form used by torch
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}
}  // namespace

extern "C" int cartnet_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 float lr, float beta1, float beta2, float eps, int32_t step, float grad_scale,
                                 void* stream) {
  CN_CHECK(n >= 0 && step >= 1, "cartnet_adam_step: bad n/step");
  if (n == 0) return 0;
  CN_CHECK(param && grad && exp_avg && exp_avg_sq, "cartnet_adam_step: null pointer");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cn_adam_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param,
                     grad, exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2),
                     grad_scale);
  CN_LAUNCH_CHECK("cartnet_adam_step");
  return 0;
}
