// eComformer's equivariant update (reference: models/comformer_conv.py:197-280, ComformerConvEqui on two
// TensorProductConvLayer).  The reference builds it on e3nn (o3.spherical_harmonics + FullyConnectedTensorProduct with
// per-edge weights, not in the build image); what that composition computes for the irreps it is given is restated here
// and in oracle/ecomformer_ref.py (derivation there; parity with e3nn itself is UNPINNED):
//
//   sh  = [1 | sqrt(3) r | sqrt(5) Y2(r)]  (r = unit edge vector, "component" normalisation: |Y_l|^2 = 2l+1)
//   layer 1 (64x0e -> 64x0e + 8x1o + 8x2e), per edge (source j <- target i), weights w [5120] from the edge MLP:
//        t = (1/8) x_i W,  W = [64x64 | 64x8 | 64x8] (u-major);   out = [t_0 | t_1 (x) sh_1 | t_2 (x) sh_2]
//        h1_j = mean over the edges leaving j of out, + pad(x_j)
//   layer 2 (64x0e + 8x1o + 8x2e -> 64x0e): in = [s_i | <v1_i, sh_1>/sqrt(3) | <v2_i, sh_2>/sqrt(5)] (80 scalars),
//        out = (1/sqrt(80)) in W,  W = [80 x 64];   o2_j = mean over the edges leaving j of out
// Any orthonormal real basis of l = 2 gives the same o2 (layer 2 only contracts Y2 with Y2), so the basis below need
// not be e3nn's.  One workgroup per SOURCE atom walks its outgoing edges in CSC order (cartnet_csr_build's perm):
// fixed summation order, no atomics.  The per-edge weight rows (20 KB each) are read exactly once per pass: HBM-bound.
#include "common.h"

namespace {

constexpr int NS = 64;              // scalar channels
constexpr int NVEC = 8;             // vector / tensor channels
constexpr int H1 = NS + 3 * NVEC + 5 * NVEC;      // 128
constexpr int NW = NS * NS + 2 * NS * NVEC;       // 5120 weights per edge, both layers
constexpr int NIN2 = NS + 2 * NVEC;               // 80

__device__ __forceinline__ void sh12(const float* __restrict__ dir, float (&s1)[3], float (&s2)[5]) {
  float x = dir[0], y = dir[1], z = dir[2];
  const float n = rsqrtf(fmaxf(x * x + y * y + z * z, 1e-30f));
  x *= n; y *= n; z *= n;
  const float r3 = 1.7320508075688772f, r5 = 2.23606797749979f;
  s1[0] = r3 * x; s1[1] = r3 * y; s1[2] = r3 * z;
  s2[0] = r5 * r3 * x * y;
  s2[1] = r5 * r3 * y * z;
  s2[2] = r5 * 0.5f * (3.f * z * z - 1.f);
  s2[3] = r5 * r3 * x * z;
  s2[4] = r5 * 0.5f * r3 * (x * x - y * y);
}

// ------------------------------------------------------------------------------------------------ layer 1 forward
__global__ __launch_bounds__(256) void cn_equi_tp1_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ w,
                                                              const float* __restrict__ dir,
                                                              const int* __restrict__ colptr, const int* __restrict__ perm,
                                                              const int* __restrict__ tgt, int N, float* __restrict__ h1) {
  __shared__ float xi[NS];
  __shared__ float part[4][NIN2];
  __shared__ float tt[NIN2];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = colptr[j], k1 = colptr[j + 1];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int k = k0; k < k1; ++k) {
    const int e = perm[k], i = tgt[e];
    if (tid < NS) xi[tid] = x0[(size_t)i * NS + tid];
    __syncthreads();
    const float* __restrict__ we = w + (size_t)e * NW;
    float p0 = 0.f, pv = 0.f;
#pragma unroll 4
    for (int u = wave * 16; u < wave * 16 + 16; ++u) {
      const float xu = xi[u];
      p0 += we[u * NS + lane] * xu;
      if (lane < 16) pv += we[NS * NS + (lane >> 3) * NS * NVEC + u * NVEC + (lane & 7)] * xu;
    }
    part[wave][lane] = p0;
    if (lane < 16) part[wave][NS + lane] = pv;
    __syncthreads();
    if (tid < NIN2) tt[tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    __syncthreads();
    float s1[3], s2[5];
    sh12(dir + (size_t)e * 3, s1, s2);
    if (tid < NS) a0 += tt[tid];
    if (tid < 3 * NVEC) a1 += tt[NS + tid / 3] * s1[tid % 3];
    if (tid < 5 * NVEC) a2 += tt[NS + NVEC + tid / 5] * s2[tid % 5];
  }
  const float scale = (k1 > k0) ? 0.125f / (float)(k1 - k0) : 0.f;
  float* out = h1 + (size_t)j * H1;
  if (tid < NS) out[tid] = a0 * scale + x0[(size_t)j * NS + tid];
  if (tid < 3 * NVEC) out[NS + tid] = a1 * scale;
  if (tid < 5 * NVEC) out[NS + 3 * NVEC + tid] = a2 * scale;
}

// layer 1 backward: dw [E, 5120] (gradient of the per-edge weights) and dxe [E, 64] (gradient w.r.t. the gathered
// target features, reduced over targets by the caller with cartnet_segment_sum); the residual's dx0 += dh1[:, :64]
// is the caller's too.
__global__ __launch_bounds__(256) void cn_equi_tp1_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ w,
                                                              const float* __restrict__ dir,
                                                              const int* __restrict__ colptr, const int* __restrict__ perm,
                                                              const int* __restrict__ tgt, const float* __restrict__ dh1,
                                                              int N, float* __restrict__ dw, float* __restrict__ dxe) {
  __shared__ float xi[NS];
  __shared__ float dt[NIN2];
  __shared__ float g[H1];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = colptr[j], k1 = colptr[j + 1];
  if (k1 == k0) return;
  const float scale = 0.125f / (float)(k1 - k0);
  if (tid < H1) g[tid] = dh1[(size_t)j * H1 + tid] * scale;
  __syncthreads();
  for (int k = k0; k < k1; ++k) {
    const int e = perm[k], i = tgt[e];
    float s1[3], s2[5];
    sh12(dir + (size_t)e * 3, s1, s2);
    if (tid < NS) {
      xi[tid] = x0[(size_t)i * NS + tid];
      dt[tid] = g[tid];
    } else if (tid < NS + NVEC) {
      const int q = tid - NS;
      dt[tid] = g[NS + q * 3] * s1[0] + g[NS + q * 3 + 1] * s1[1] + g[NS + q * 3 + 2] * s1[2];
    } else if (tid < NIN2) {
      const int q = tid - NS - NVEC;
      float t = 0.f;
#pragma unroll
      for (int m = 0; m < 5; ++m) t += g[NS + 3 * NVEC + q * 5 + m] * s2[m];
      dt[tid] = t;
    }
    __syncthreads();
    const float* __restrict__ we = w + (size_t)e * NW;
    float* __restrict__ dwe = dw + (size_t)e * NW;
    // weight gradient: outer product x_i (x) dt, 5120 entries as 1280 float4 (four consecutive outputs of one u)
    for (int q4 = tid; q4 < NW / 4; q4 += 256) {
      const int q = q4 * 4;
      int u, c;
      if (q < NS * NS) { u = q >> 6; c = q & 63; }
      else { const int r = q - NS * NS; u = (r & (NS * NVEC - 1)) >> 3; c = NS + (r >> 9) * NVEC + (r & 7); }
      const float xu = xi[u];
      *reinterpret_cast<f32x4*>(dwe + q) = f32x4{xu * dt[c], xu * dt[c + 1], xu * dt[c + 2], xu * dt[c + 3]};
    }
    // dx_i[u] = sum_c W[u, c] dt[c]: one row per wave pass, reduced across the wave
    for (int u = wave * 16; u < wave * 16 + 16; ++u) {
      float v = we[u * NS + lane] * dt[lane];
      if (lane < 16) v += we[NS * NS + (lane >> 3) * NS * NVEC + u * NVEC + (lane & 7)] * dt[NS + lane];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) dxe[(size_t)e * NS + u] = v;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ layer 2 forward
__global__ __launch_bounds__(256) void cn_equi_tp2_fwd_kernel(const float* __restrict__ h1, const float* __restrict__ w,
                                                              const float* __restrict__ dir,
                                                              const int* __restrict__ colptr, const int* __restrict__ perm,
                                                              const int* __restrict__ tgt, int N, float* __restrict__ o2) {
  __shared__ float hi[H1];
  __shared__ float in[NIN2];
  __shared__ float part[4][NS];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = colptr[j], k1 = colptr[j + 1];
  float acc = 0.f;
  for (int k = k0; k < k1; ++k) {
    const int e = perm[k], i = tgt[e];
    if (tid < H1) hi[tid] = h1[(size_t)i * H1 + tid];
    __syncthreads();
    float s1[3], s2[5];
    sh12(dir + (size_t)e * 3, s1, s2);
    if (tid < NS) in[tid] = hi[tid];
    else if (tid < NS + NVEC) {
      const int q = tid - NS;
      in[tid] = (hi[NS + q * 3] * s1[0] + hi[NS + q * 3 + 1] * s1[1] + hi[NS + q * 3 + 2] * s1[2]) * 0.5773502691896258f;
    } else if (tid < NIN2) {
      const int q = tid - NS - NVEC;
      float t = 0.f;
#pragma unroll
      for (int m = 0; m < 5; ++m) t += hi[NS + 3 * NVEC + q * 5 + m] * s2[m];
      in[tid] = t * 0.4472135954999579f;
    }
    __syncthreads();
    const float* __restrict__ we = w + (size_t)e * NW;
    float p = 0.f;
#pragma unroll 4
    for (int u = wave * 20; u < wave * 20 + 20; ++u) p += we[u * NS + lane] * in[u];
    part[wave][lane] = p;
    __syncthreads();
    if (tid < NS) acc += (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
  }
  const float scale = (k1 > k0) ? 0.11180339887498948f / (float)(k1 - k0) : 0.f;     // 1 / sqrt(80)
  if (tid < NS) o2[(size_t)j * NS + tid] = acc * scale;
}

// layer 2 backward: dw [E, 5120] and dhe [E, 128] (gradient w.r.t. the gathered h1 rows; reduced over targets by the
// caller)
__global__ __launch_bounds__(256) void cn_equi_tp2_bwd_kernel(const float* __restrict__ h1, const float* __restrict__ w,
                                                              const float* __restrict__ dir,
                                                              const int* __restrict__ colptr, const int* __restrict__ perm,
                                                              const int* __restrict__ tgt, const float* __restrict__ do2,
                                                              int N, float* __restrict__ dw, float* __restrict__ dhe) {
  __shared__ float hi[H1];
  __shared__ float in[NIN2];
  __shared__ float din[NIN2];
  __shared__ float g[NS];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = colptr[j], k1 = colptr[j + 1];
  if (k1 == k0) return;
  const float scale = 0.11180339887498948f / (float)(k1 - k0);
  if (tid < NS) g[tid] = do2[(size_t)j * NS + tid] * scale;
  __syncthreads();
  for (int k = k0; k < k1; ++k) {
    const int e = perm[k], i = tgt[e];
    if (tid < H1) hi[tid] = h1[(size_t)i * H1 + tid];
    __syncthreads();
    float s1[3], s2[5];
    sh12(dir + (size_t)e * 3, s1, s2);
    if (tid < NS) in[tid] = hi[tid];
    else if (tid < NS + NVEC) {
      const int q = tid - NS;
      in[tid] = (hi[NS + q * 3] * s1[0] + hi[NS + q * 3 + 1] * s1[1] + hi[NS + q * 3 + 2] * s1[2]) * 0.5773502691896258f;
    } else if (tid < NIN2) {
      const int q = tid - NS - NVEC;
      float t = 0.f;
#pragma unroll
      for (int m = 0; m < 5; ++m) t += hi[NS + 3 * NVEC + q * 5 + m] * s2[m];
      in[tid] = t * 0.4472135954999579f;
    }
    __syncthreads();
    const float* __restrict__ we = w + (size_t)e * NW;
    float* __restrict__ dwe = dw + (size_t)e * NW;
    for (int q4 = tid; q4 < NW / 4; q4 += 256) {
      const int q = q4 * 4, c = q & 63;
      const float iu = in[q >> 6];
      *reinterpret_cast<f32x4*>(dwe + q) = f32x4{iu * g[c], iu * g[c + 1], iu * g[c + 2], iu * g[c + 3]};
    }
    for (int u = wave * 20; u < wave * 20 + 20; ++u) {
      float v = we[u * NS + lane] * g[lane];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) din[u] = v;
    }
    __syncthreads();
    float* __restrict__ d = dhe + (size_t)e * H1;
    if (tid < NS) d[tid] = din[tid];
    else if (tid < NS + 3 * NVEC) {
      const int q = tid - NS;
      d[tid] = din[NS + q / 3] * s1[q % 3] * 0.5773502691896258f;
    } else if (tid < H1) {
      const int q = tid - NS - 3 * NVEC;
      d[tid] = din[NS + NVEC + q / 5] * s2[q % 5] * 0.4472135954999579f;
    }
    __syncthreads();
  }
}

// per-block fp64 partial column sums and sums of squares of x [R, C] (BatchNorm statistics of a tensor that no GEMM
// epilogue produced); finalise with cartnet_bn_finalize
constexpr int STAT_ROWS_PER_BLOCK = 4;
__global__ __launch_bounds__(256) void cn_colstats_partial_kernel(const float* __restrict__ x, int ld, int R, int C,
                                                                  double* __restrict__ ps, double* __restrict__ pq) {
  __shared__ double red[STAT_ROWS_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool active = c < C;
    f64x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    for (int r = blockIdx.x * STAT_ROWS_PER_BLOCK + wid; r < R; r += gridDim.x * STAT_ROWS_PER_BLOCK)
      if (active) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * ld + c);
        cn_acc4(s, v);
        cn_acc4(q, v * v);
      }
    cn_block_store_parts(s, red, ps, C, c, active, wid, lane);
    cn_block_store_parts(q, red, pq, C, c, active, wid, lane);
  }
}

inline int stat_parts(int R) {
  int b = cn_ceil_div(R, STAT_ROWS_PER_BLOCK);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return b;
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int cartnet_equi_tp1_fwd(const float* x0, const float* w, const float* cart_dir, const int32_t* colptr,
                                    const int32_t* perm, const int32_t* tgt, int32_t N, float* h1, void* stream) {
  CN_CHECK(N >= 0, "cartnet_equi_tp1_fwd: N=%d", N);
  if (N == 0) return 0;
  CN_CHECK(x0 && w && cart_dir && colptr && perm && tgt && h1, "cartnet_equi_tp1_fwd: null pointer");
  hipLaunchKernelGGL(cn_equi_tp1_fwd_kernel, dim3(N), dim3(256), 0, ST(stream), x0, w, cart_dir, colptr, perm, tgt, N, h1);
  CN_LAUNCH_CHECK("cartnet_equi_tp1_fwd");
  return 0;
}

extern "C" int cartnet_equi_tp1_bwd(const float* x0, const float* w, const float* cart_dir, const int32_t* colptr,
                                    const int32_t* perm, const int32_t* tgt, const float* dh1, int32_t N, float* dw,
                                    float* dxe, void* stream) {
  CN_CHECK(N >= 0, "cartnet_equi_tp1_bwd: N=%d", N);
  if (N == 0) return 0;
  CN_CHECK(x0 && w && cart_dir && colptr && perm && tgt && dh1 && dw && dxe, "cartnet_equi_tp1_bwd: null pointer");
  hipLaunchKernelGGL(cn_equi_tp1_bwd_kernel, dim3(N), dim3(256), 0, ST(stream), x0, w, cart_dir, colptr, perm, tgt, dh1, N,
                     dw, dxe);
  CN_LAUNCH_CHECK("cartnet_equi_tp1_bwd");
  return 0;
}

extern "C" int cartnet_equi_tp2_fwd(const float* h1, const float* w, const float* cart_dir, const int32_t* colptr,
                                    const int32_t* perm, const int32_t* tgt, int32_t N, float* o2, void* stream) {
  CN_CHECK(N >= 0, "cartnet_equi_tp2_fwd: N=%d", N);
  if (N == 0) return 0;
  CN_CHECK(h1 && w && cart_dir && colptr && perm && tgt && o2, "cartnet_equi_tp2_fwd: null pointer");
  hipLaunchKernelGGL(cn_equi_tp2_fwd_kernel, dim3(N), dim3(256), 0, ST(stream), h1, w, cart_dir, colptr, perm, tgt, N, o2);
  CN_LAUNCH_CHECK("cartnet_equi_tp2_fwd");
  return 0;
}

extern "C" int cartnet_equi_tp2_bwd(const float* h1, const float* w, const float* cart_dir, const int32_t* colptr,
                                    const int32_t* perm, const int32_t* tgt, const float* do2, int32_t N, float* dw,
                                    float* dhe, void* stream) {
  CN_CHECK(N >= 0, "cartnet_equi_tp2_bwd: N=%d", N);
  if (N == 0) return 0;
  CN_CHECK(h1 && w && cart_dir && colptr && perm && tgt && do2 && dw && dhe, "cartnet_equi_tp2_bwd: null pointer");
  hipLaunchKernelGGL(cn_equi_tp2_bwd_kernel, dim3(N), dim3(256), 0, ST(stream), h1, w, cart_dir, colptr, perm, tgt, do2, N,
                     dw, dhe);
  CN_LAUNCH_CHECK("cartnet_equi_tp2_bwd");
  return 0;
}

extern "C" int cartnet_colstats_nparts(int32_t R) { return stat_parts(R); }

extern "C" int cartnet_colstats_partial(const float* x, int32_t ld, int32_t R, int32_t C, double* parts_sum,
                                        double* parts_sq, void* stream) {
  CN_CHECK(R >= 0 && C >= 4 && C % 4 == 0 && ld % 4 == 0 && ld >= C, "cartnet_colstats_partial: C/ld must be multiples of 4");
  CN_CHECK((x || R == 0) && parts_sum && parts_sq, "cartnet_colstats_partial: null pointer");
  hipLaunchKernelGGL(cn_colstats_partial_kernel, dim3(stat_parts(R)), dim3(256), 0, ST(stream), x, ld, R, C, parts_sum,
                     parts_sq);
  CN_LAUNCH_CHECK("cartnet_colstats_partial");
  return 0;
}
