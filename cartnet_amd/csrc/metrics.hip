// ADP evaluation metrics on the GPU (reference: train/metrics.py:30-180): ellipsoid volume error,
// similarity index S12 and the voxelised 3-D IoU of the predicted / true thermal ellipsoids, the step right after the
// hot path at test time.  One workgroup per atom.  The reference materialises two [M, P^3] Mahalanobis maps; here
// every voxel is classified in registers and only two counters per atom leave the CU, so the kernel is pure fp32 ALU
// work (2 * 14 flops per voxel) with 72 B of input and 12 B of output per atom.
//
// Arithmetic: the per-atom 3x3 algebra (determinants, inverses, S12, volumes) is evaluated in fp64 and rounded once --
// S12 = 100 (1 - num/den) cancels catastrophically in fp32 when pred ~ true.  The voxel test follows the reference's
// fp32 evaluation (normalise by the larger Frobenius norm, x^T Sigma^-1 x, sqrt(.) < 1) so that voxel decisions agree
// with it except for voxels within fp32 rounding of the surface.
#include "common.h"
#include <math.h>

namespace {

constexpr double kSmooth = 1e-8;      // train/metrics.py:11

__device__ __forceinline__ double det3(const double* a) {
  return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

__device__ __forceinline__ void inv3(const double* a, double* o) {
  const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const double d = a[0] * c00 + a[1] * c01 + a[2] * c02;
  const double r = 1.0 / d;
  o[0] = c00 * r;
  o[1] = (a[2] * a[7] - a[1] * a[8]) * r;
  o[2] = (a[1] * a[5] - a[2] * a[4]) * r;
  o[3] = c01 * r;
  o[4] = (a[0] * a[8] - a[2] * a[6]) * r;
  o[5] = (a[2] * a[3] - a[0] * a[5]) * r;
  o[6] = c02 * r;
  o[7] = (a[1] * a[6] - a[0] * a[7]) * r;
  o[8] = (a[0] * a[4] - a[1] * a[3]) * r;
}

// x^T S x < 1 as the reference evaluates it: mult = p @ S (row vector times matrix), sum(mult * p), sqrt, < 1
// (train/metrics.py:139-146).  mxy* hold the x and y terms of mult, shared by a whole z column.
__device__ __forceinline__ bool inside(float x, float y, float z, float m0xy, float m1xy, float m2xy,
                                       const float* s) {
  const float m0 = fmaf(z, s[6], m0xy), m1 = fmaf(z, s[7], m1xy), m2 = fmaf(z, s[8], m2xy);
  const float q = __fadd_rn(__fadd_rn(__fmul_rn(m0, x), __fmul_rn(m1, y)), __fmul_rn(m2, z));
  // sqrtf(q) < 1 without the square root: correctly rounded sqrt maps the largest float below 1 (1 - 2^-24) to
  // itself (sqrt(1 - e) = 1 - e/2 - e^2/8 - ... lies just under the midpoint 1 - 2^-25), so for q >= 0 the two tests
  // agree on every float; q < 0 (not positive definite) gives NaN < 1 = false in the reference.
  return q >= 0.0f && q < 1.0f;
}

__global__ __launch_bounds__(256) void cn_adp_metrics_kernel(const float* __restrict__ pred,
                                                             const float* __restrict__ tru, int M,
                                                             const float* __restrict__ grid, int P,
                                                             float* __restrict__ vol_err, float* __restrict__ sim,
                                                             float* __restrict__ iou) {
  const int a = blockIdx.x, tid = threadIdx.x;
  float pf[9], tf[9];
  double p[9], t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    pf[i] = pred[(size_t)a * 9 + i];
    tf[i] = tru[(size_t)a * 9 + i];
    p[i] = pf[i];
    t[i] = tf[i];
  }
  if (tid == 0) {
    if (vol_err) {   // train/metrics.py:42-58 (the reference's "true" volume is the prediction's: kept)
      const double v1 = (4.0 / 3.0) * M_PI * sqrt(det3(p)), v2 = (4.0 / 3.0) * M_PI * sqrt(det3(t));
      vol_err[a] = (float)(fabs(v1 - v2) / (v1 + kSmooth));
    }
    if (sim) {       // train/metrics.py:76-94
      double it[9], ip[9], prod[9], sum[9];
      inv3(t, it);
      inv3(p, ip);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          prod[r * 3 + c] = it[r * 3] * ip[c] + it[r * 3 + 1] * ip[3 + c] + it[r * 3 + 2] * ip[6 + c];
          sum[r * 3 + c] = it[r * 3 + c] + ip[r * 3 + c];
        }
      const double num = 2.8284271247461903 * pow(det3(prod), 0.25);
      const double den = sqrt(det3(sum));
      sim[a] = (float)(100.0 * (1.0 - num / den));
    }
  }
  if (!iou) return;

  // train/metrics.py:158-166: scale both matrices by the larger Frobenius norm (fp32 division, as the reference)
  double sp = 0.0, st = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    sp += p[i] * p[i];
    st += t[i] * t[i];
  }
  const float np_ = (float)sqrt(sp), nt_ = (float)sqrt(st);
  const float nrm = np_ > nt_ ? np_ : nt_;
  double pn[9], tn[9], ipd[9], itd[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    pn[i] = (double)(pf[i] / nrm);
    tn[i] = (double)(tf[i] / nrm);
  }
  inv3(pn, ipd);
  inv3(tn, itd);
  float ip[9], it[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    ip[i] = (float)ipd[i];
    it[i] = (float)itd[i];
  }

  int inter = 0, uni = 0;
  for (int pair = tid; pair < P * P; pair += 256) {
    const float x = grid[pair / P], y = grid[pair % P];
    const float p0 = fmaf(y, ip[3], x * ip[0]), p1 = fmaf(y, ip[4], x * ip[1]), p2 = fmaf(y, ip[5], x * ip[2]);
    const float t0 = fmaf(y, it[3], x * it[0]), t1 = fmaf(y, it[4], x * it[1]), t2 = fmaf(y, it[5], x * it[2]);
    for (int k = 0; k < P; ++k) {
      const float z = grid[k];
      const bool in_p = inside(x, y, z, p0, p1, p2, ip);
      const bool in_t = inside(x, y, z, t0, t1, t2, it);
      inter += (in_p && in_t);
      uni += (in_p || in_t);
    }
  }
  __shared__ int red[8];
#pragma unroll
  for (int off = 32; off; off >>= 1) {
    inter += __shfl_down(inter, off, 64);
    uni += __shfl_down(uni, off, 64);
  }
  if ((tid & 63) == 0) {
    red[(tid >> 6) * 2] = inter;
    red[(tid >> 6) * 2 + 1] = uni;
  }
  __syncthreads();
  if (tid == 0) {   // train/metrics.py:96-112: (|A & B| + SMOOTH) / (|A | B| + SMOOTH) on float counts
    const float fi = (float)(red[0] + red[2] + red[4] + red[6]), fu = (float)(red[1] + red[3] + red[5] + red[7]);
    iou[a] = (fi + 1e-8f) / (fu + 1e-8f);
  }
}

}  // namespace

extern "C" int cartnet_adp_metrics(const float* pred, const float* truth, int32_t M, const float* grid,
                                   int32_t num_points, float* volume_error, float* similarity_index, float* iou,
                                   void* stream) {
  CN_CHECK(M >= 0, "cartnet_adp_metrics: M must be >= 0 (got %d)", M);
  if (M == 0) return 0;
  CN_CHECK(pred && truth, "cartnet_adp_metrics: null pointer");
  CN_CHECK(volume_error || similarity_index || iou, "cartnet_adp_metrics: no output requested");
  CN_CHECK(!iou || (grid && num_points >= 1 && num_points <= 1024),
           "cartnet_adp_metrics: the IoU needs grid[num_points], 1 <= num_points <= 1024 (got %d)", num_points);
  hipLaunchKernelGGL(cn_adp_metrics_kernel, dim3(M), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pred, truth,
                     M, grid, num_points, volume_error, similarity_index, iou);
  CN_LAUNCH_CHECK("cartnet_adp_metrics");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Training loss (reference: train/metrics.py:15-28 -- L1Loss and MSELoss, mean over all elements): both means from one
// launch, their gradient from another.  The eager form is eight dependent launches per step (sub, abs, mean, and in
// backward fill, div, sign, mul, mul); at the small-crystal configurations a launch is 5-6 us of a 1.3 ms step.
// Two small launches forward (up to 64 slices, then one wave adds them -- one workgroup alone took 30 us for the 111k
// elements of the benchmark batch), one backward; fp64 sums in a fixed order (bitwise reproducible).
namespace {

constexpr int LOSS_MAX_PARTS = 64;

// stage 1: up to 64 workgroups, each a contiguous slice, fp64 sums in a fixed order -> parts[2][nparts]
__global__ __launch_bounds__(256) void cn_loss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                              long long n, int nparts, double* __restrict__ parts) {
  __shared__ double red[2][4];
  const long long per = (n + nparts - 1) / nparts, lo = (long long)blockIdx.x * per, hi = min(n, lo + per);
  double sa = 0.0, sq = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const double d = (double)pred[i] - (double)truth[i];
    sa += fabs(d);
    sq += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sa += __shfl_xor(sa, o);
    sq += __shfl_xor(sq, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sa;
    red[1][threadIdx.x >> 6] = sq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    parts[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    parts[nparts + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// both stages in one launch when one slice covers the input (n <= 4096: the scalar targets of configs[2], 64 per step):
// the same sums in the same order as stage 1 + stage 2 with nparts = 1, one launch less on the step's critical chain
__global__ __launch_bounds__(256) void cn_loss_small_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                            long long n, float* __restrict__ out,
                                                            float* __restrict__ unit) {
  __shared__ double red[2][4];
  double sa = 0.0, sq = 0.0;
  const float inv = 1.0f / (float)n;
  for (long long i = threadIdx.x; i < n; i += 256) {
    const double d = (double)pred[i] - (double)truth[i];
    sa += fabs(d);
    sq += d * d;
    if (unit) {       // the expressions of cn_loss_bwd_kernel with g_mae[0] = 1, g_mse = NULL and the other way round
      const float df = pred[i] - truth[i];
      const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
      unit[i] = (1.0f * inv) * sgn + 0.f * df;
      unit[n + i] = 0.f * sgn + (1.0f * (2.0f * inv)) * df;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sa += __shfl_xor(sa, o);
    sq += __shfl_xor(sq, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sa;
    red[1][threadIdx.x >> 6] = sq;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const double s = 0.0 + (red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
    out[threadIdx.x] = (float)(s / (double)n);
  }
}

// stage 2: one wave adds the partial sums in order
__global__ __launch_bounds__(64) void cn_loss_finalize_kernel(const double* __restrict__ parts, int nparts, long long n,
                                                              float* __restrict__ out) {
  if (threadIdx.x < 2) {
    double s = 0.0;
    for (int k = 0; k < nparts; ++k) s += parts[threadIdx.x * nparts + k];
    out[threadIdx.x] = (float)(s / (double)n);
  }
}

// dpred = g_mae * sign(d) / n + g_mse * 2 d / n; the two upstream gradients are device scalars (either may be null = 0)
__global__ __launch_bounds__(256) void cn_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                          long long n, const float* __restrict__ g_mae,
                                                          const float* __restrict__ g_mse, float* __restrict__ dpred) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float inv = 1.0f / (float)n;
  const float ga = g_mae ? g_mae[0] * inv : 0.f, gs = g_mse ? g_mse[0] * (2.0f * inv) : 0.f;
  const float d = pred[i] - truth[i];
  const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  dpred[i] = ga * sgn + gs * d;
}

}  // namespace

extern "C" int32_t cartnet_loss_nparts(int64_t n) {
  const long long want = (n + 4095) / 4096;
  return (int32_t)(want < 1 ? 1 : (want > LOSS_MAX_PARTS ? LOSS_MAX_PARTS : want));
}

extern "C" int cartnet_loss_fwd(const float* pred, const float* truth, int64_t n, double* parts, float* out2,
                                void* stream) {
  CN_CHECK(pred && truth && parts && out2 && n > 0, "cartnet_loss_fwd: null pointer or n = %lld", (long long)n);
  const int nparts = cartnet_loss_nparts(n);
  if (nparts == 1) {
    hipLaunchKernelGGL(cn_loss_small_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pred, truth,
                       (long long)n, out2, static_cast<float*>(nullptr));
    CN_LAUNCH_CHECK("cartnet_loss_fwd");
    return 0;
  }
  hipLaunchKernelGGL(cn_loss_partial_kernel, dim3(nparts), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pred,
                     truth, (long long)n, nparts, parts);
  hipLaunchKernelGGL(cn_loss_finalize_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), parts, nparts,
                     (long long)n, out2);
  CN_LAUNCH_CHECK("cartnet_loss_fwd");
  return 0;
}

extern "C" int cartnet_loss_fwd_unit(const float* pred, const float* truth, int64_t n, float* out2, float* unit,
                                     void* stream) {
  CN_CHECK(pred && truth && out2 && unit && n > 0, "cartnet_loss_fwd_unit: null pointer or n = %lld", (long long)n);
  CN_CHECK(cartnet_loss_nparts(n) == 1, "cartnet_loss_fwd_unit: n = %lld is more than one slice (4096 elements)", (long long)n);
  hipLaunchKernelGGL(cn_loss_small_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pred, truth,
                     (long long)n, out2, unit);
  CN_LAUNCH_CHECK("cartnet_loss_fwd_unit");
  return 0;
}

extern "C" int cartnet_loss_bwd(const float* pred, const float* truth, int64_t n, const float* g_mae, const float* g_mse,
                                float* dpred, void* stream) {
  CN_CHECK(pred && truth && dpred && n > 0, "cartnet_loss_bwd: null pointer or n = %lld", (long long)n);
  hipLaunchKernelGGL(cn_loss_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), pred, truth, (long long)n, g_mae, g_mse, dpred);
  CN_LAUNCH_CHECK("cartnet_loss_bwd");
  return 0;
}
