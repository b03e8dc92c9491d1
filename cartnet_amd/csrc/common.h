// Shared helpers for the gfx950 kernels of libcartnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/cartnet_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WAVE 64

void cartnet_set_error(const char* fmt, ...);

#define CN_CHECK(cond, ...)                \
  do {                                     \
    if (!(cond)) {                         \
      cartnet_set_error(__VA_ARGS__);      \
      return 1;                            \
    }                                      \
  } while (0)

#define CN_LAUNCH_CHECK(name)                                                    \
  do {                                                                           \
    hipError_t _e = hipGetLastError();                                           \
    if (_e != hipSuccess) {                                                      \
      cartnet_set_error("%s: launch failed: %s", name, hipGetErrorString(_e));   \
      return 2;                                                                  \
    }                                                                            \
  } while (0)

// Hardware exp / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each): expf + a true division cost ~25 VALU instructions per
// element, which the gate kernels (E x D sigmoids per layer and direction) pay while they share the chip with a GEMM.
__device__ __forceinline__ float cn_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float cn_silu(float x) { return x * cn_sigmoid(x); }
// d/dx [x * sigmoid(x)] = s * (1 + x * (1 - s))
__device__ __forceinline__ float cn_dsilu(float x) {
  float s = cn_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}

typedef double f64x4 __attribute__((ext_vector_type(4)));

// Per-block partial column sums, kept in fp64 (see cartnet_hip.h "partial sums").  Four waves each hold the sums
// of channels c..c+3 (lane l -> c = c0 + 4l); wave 0 adds them in wave order and writes parts[blockIdx.x][c..c+3].
// lds: 4 * 256 doubles.
__device__ __forceinline__ void cn_block_store_parts_row(f64x4 v, double* lds, double* parts, int D, int c, bool active,
                                                         int wid, int lane, int row) {
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) lds[wid * 256 + lane * 4 + q] = v[q];
  __syncthreads();
  if (wid == 0 && active) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double t = lds[lane * 4 + q];
#pragma unroll
      for (int w = 1; w < 4; ++w) t += lds[w * 256 + lane * 4 + q];
      parts[(size_t)row * D + c + q] = t;
    }
  }
}
__device__ __forceinline__ void cn_block_store_parts(f64x4 v, double* lds, double* parts, int D, int c, bool active,
                                                     int wid, int lane) {
  cn_block_store_parts_row(v, lds, parts, D, c, active, wid, lane, blockIdx.x);
}

// BatchNorm groups (CartnetGroups, cartnet_hip.h): group `g` of a per-node / per-edge kernel owns the nodes [n0, n1)
// and the statistics row g.  node_gptr == nullptr: one group, the whole batch.  With `reverse` (kernels that re-read
// what their predecessor has just streamed in ascending order, see "Sweep direction" in edge_ops.hip) the groups and the
// workgroups inside a group are dealt in descending order, so that the first workgroups start on the highest rows.
__device__ __forceinline__ void cn_group_range(const int* __restrict__ node_gptr, int N, bool reverse, int& g, int& bx,
                                               int& n0, int& n1) {
  const bool flip = reverse && node_gptr != nullptr;
  g = flip ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
  bx = flip ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
  n0 = node_gptr ? node_gptr[g] : 0;
  n1 = node_gptr ? node_gptr[g + 1] : N;
}

__device__ __forceinline__ void cn_acc4(f64x4& a, f32x4 v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] += (double)v[q];
}

// Column sum of a [nparts][N] fp64 partial-sum matrix for CN_SUM_COLS consecutive columns, by a 1024-thread block:
// 64 row groups sum rows g, g+64, ... each (eight independent loads in flight per thread, added in row order), then
// thread c (< CN_SUM_COLS) adds the 64 group sums in group order.  Returns the total in threads 0..CN_SUM_COLS-1 (column
// col0 + tid); fixed summation order -> bitwise reproducible.  Sixteen columns per block put N/16 blocks on a matrix of
// a few thousand rows (the 1,384 row tiles of an edge-sized GEMM), which made the separate folding pass that used to
// precede every finaliser (20 us per call, 22 calls per step) unnecessary.
constexpr int CN_SUM_COLS = 16;
__device__ __forceinline__ double cn_block_colsum(const double* __restrict__ parts, int nparts, int N, int col0,
                                                  double* lds /* [64][CN_SUM_COLS] */) {
  const int tid = threadIdx.x, g = tid >> 4, cl = tid & 15;
  const int c = col0 + cl;
  double acc = 0.0;
  if (c < N) {
    const double* __restrict__ col = parts + c;
    int p = g;
    for (; p + 7 * 64 < nparts; p += 8 * 64) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(p + u * 64) * N];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; p < nparts; p += 64) acc += col[(size_t)p * N];
  }
  __syncthreads();
  lds[g * CN_SUM_COLS + cl] = acc;
  __syncthreads();
  double tot = 0.0;
  if (tid < CN_SUM_COLS)
    for (int q = 0; q < 64; ++q) tot += lds[q * CN_SUM_COLS + tid];
  return tot;
}

static inline int cn_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// CartnetGroups on the host: NULL = one group with the kernel's own default number of workgroups.
static inline bool cn_groups_ok(const CartnetGroups* g) {
  return !g || (g->G >= 1 && g->G <= 65535 && g->node_gptr && g->edge_gptr && g->edge_parts >= 1 && g->node_parts >= 1);
}
static inline dim3 cn_group_grid(const CartnetGroups* g, int default_parts, bool per_edge) {
  return g ? dim3(per_edge ? g->edge_parts : g->node_parts, g->G) : dim3(default_parts, 1);
}
