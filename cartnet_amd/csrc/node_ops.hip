// Per-node kernels: atom embedding, BatchNorm statistics, node update, Cholesky / scalar heads.
// All are tiny next to the per-edge work (N ~ E/14); they favour determinism (fixed-order sums, fp64 finalise)
// over the last few percent of bandwidth.
#include "common.h"
#include <math.h>

namespace {

constexpr int NODES_PER_BLOCK = 4;
constexpr int MAX_NODE_PARTS = 256;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

inline int node_parts(int N) {
  int b = cn_ceil_div(N, NODES_PER_BLOCK);
  if (b > MAX_NODE_PARTS) b = MAX_NODE_PARTS;
  if (b < 1) b = 1;
  return b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------------------------------------ embedding
// z / batch ids outside their tables are clamped (never an out-of-bounds read) and reported in the status word of the
// forward pass: bit 16 = atomic number outside [0, n_types), bit 32 = batch id outside [0, Bg).  The reference's
// nn.Embedding raises for the former; here the error surfaces through ops.raise_on_graph_status.
// z == nullptr with a table: every atom takes row 0 (the "no atom types, no temperature" ablation, cartnet.py:150-151).
__global__ void cn_node_embed_kernel(const int64_t* __restrict__ z, const int64_t* __restrict__ batch,
                                     const float* __restrict__ temperature, const float* __restrict__ emb,
                                     const float* __restrict__ wt, const float* __restrict__ bt,
                                     const float* __restrict__ bias, int N, int C, int n_types, int Bg,
                                     int* __restrict__ status, float* __restrict__ x0) {
  const long long total = (long long)N * C;
  int bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / C), c = (int)(i % C);
    float v = 0.f;
    if (emb) {
      long long a = z ? z[n] : 0;
      if (a < 0 || a >= n_types) { bad |= 16; a = a < 0 ? 0 : n_types - 1; }
      v = emb[(size_t)a * C + c];
    }
    if (wt) {
      // Linear(1 -> C) on the crystal's temperature, then broadcast to its atoms (cartnet.py:145)
      long long g = batch[n];
      if (g < 0 || g >= Bg) { bad |= 32; g = g < 0 ? 0 : Bg - 1; }
      const float t = temperature[g] * wt[c] + bt[c];
      v = emb ? v + t : t;
    }
    if (bias) v += bias[c];
    x0[i] = v;
  }
  if (bad && status) atomicOr(status, bad);
}

// Column sums of dx0 and of T[batch[n]] * dx0 (temperature projection gradients), per-block partials.
__global__ __launch_bounds__(256) void cn_embed_bwd_cols_kernel(const int64_t* __restrict__ batch,
                                                                const float* __restrict__ temperature,
                                                                const float* __restrict__ dx0, int N, int C, int Bg,
                                                                double* __restrict__ parts_w,
                                                                double* __restrict__ parts_b) {
  __shared__ double red[NODES_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool active = c < C;
    f64x4 pw = {0, 0, 0, 0}, pb = {0, 0, 0, 0};
    const int stride = gridDim.x * NODES_PER_BLOCK;
    for (int n0 = blockIdx.x * NODES_PER_BLOCK + wid; n0 < N; n0 += 4 * stride) {
      if (!active) continue;
      f32x4 d[4];
      float t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {      // four independent rows in flight; accumulated in row order below
        const int n = n0 + u * stride;
        d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        t[u] = 0.f;
        if (n < N) {
          d[u] = ld4(dx0 + (size_t)n * C + c);
          if (temperature) {
            long long g = batch[n];                 // clamped like the forward (which reported it)
            g = g < 0 ? 0 : (g >= Bg ? Bg - 1 : g);
            t[u] = temperature[g];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        cn_acc4(pw, d[u] * t[u]);
        cn_acc4(pb, d[u]);
      }
    }
    cn_block_store_parts(pw, red, parts_w, C, c, active, wid, lane);
    cn_block_store_parts(pb, red, parts_b, C, c, active, wid, lane);
  }
}

// ------------------------------------------------------------------------------------------------ transpose
// dst[c, r] = src[r, c] for up to TRANSPOSE_MAX_JOBS matrices per launch (weights -> [in, out] once per forward, so that every forward
// GEMM streams its weight tile with fully coalesced rows).
constexpr int TRANSPOSE_MAX_JOBS = 40;     // 1280 B of kernel arguments: the model's 34 weights go in one launch
struct TransposeJobs {
  const float* src[TRANSPOSE_MAX_JOBS];
  float* dst[TRANSPOSE_MAX_JOBS];
  int rows[TRANSPOSE_MAX_JOBS], cols[TRANSPOSE_MAX_JOBS], lds[TRANSPOSE_MAX_JOBS], ldd[TRANSPOSE_MAX_JOBS];
};

__global__ __launch_bounds__(256) void cn_transpose_kernel(const TransposeJobs jobs) {
  __shared__ float tile[32][33];
  const int j = blockIdx.z;
  const int rows = jobs.rows[j], cols = jobs.cols[j];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  if (r0 >= rows || c0 >= cols) return;
  const float* __restrict__ src = jobs.src[j];
  float* __restrict__ dst = jobs.dst[j];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[(size_t)(r0 + i) * jobs.lds[j] + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) dst[(size_t)(c0 + i) * jobs.ldd[j] + r0 + tx] = tile[tx][i];
}

// ------------------------------------------------------------------------------------------------ BatchNorm stats
// Groups (G > 1: parts are [G][nparts][C], counts come from cnt_ptr[g+1] - cnt_ptr[g]): every group gets its own row of
// mean_rstd, and the running statistics receive the G momentum updates one after the other, in group order -- what G
// consecutive forward calls on the micro-batches would have done (num_batches_tracked += G).
__global__ __launch_bounds__(1024) void cn_bn_finalize_kernel(
    const double* __restrict__ parts_sum, const double* __restrict__ parts_sq, int nparts, long long count, int C,
    float eps, float momentum, int training, float* __restrict__ running_mean, float* __restrict__ running_var,
    int64_t* __restrict__ nbt, float* __restrict__ mean_rstd, int G, const int* __restrict__ cnt_ptr,
    const double* __restrict__ count_dev /* optional: the row count as a device double (sync-BatchNorm) */) {
  __shared__ double red[64 * CN_SUM_COLS];
  const int c = blockIdx.x * CN_SUM_COLS + threadIdx.x;
  const bool owner = threadIdx.x < CN_SUM_COLS && c < C;
  if (blockIdx.x == 0 && threadIdx.x == 0 && training && nbt) nbt[0] += G;
  if (training) {
    double rm = 0.0, rv = 0.0;
    if (owner && running_mean) {
      rm = (double)running_mean[c];
      rv = (double)running_var[c];
    }
    for (int g = 0; g < G; ++g) {
      const size_t off = (size_t)g * nparts * C;
      const double s = cn_block_colsum(parts_sum + off, nparts, C, blockIdx.x * CN_SUM_COLS, red);
      const double q = cn_block_colsum(parts_sq + off, nparts, C, blockIdx.x * CN_SUM_COLS, red);
      if (!owner) continue;
      const long long cnt = count_dev ? (long long)(count_dev[0] + 0.5)
                                      : (cnt_ptr ? (long long)(cnt_ptr[g + 1] - cnt_ptr[g]) : count);
      const double n = (double)cnt;
      const double mean = n > 0 ? s / n : 0.0;
      double var = n > 0 ? q / n - mean * mean : 0.0;
      if (var < 0.0) var = 0.0;
      mean_rstd[(size_t)g * 2 * C + c] = (float)mean;
      mean_rstd[(size_t)g * 2 * C + C + c] = (float)(1.0 / sqrt(var + (double)eps));
      if (running_mean) {
        const double unb = cnt > 1 ? var * (n / (n - 1.0)) : var;
        // rounded to fp32 after every update, like the reference's running buffers between two forward calls
        rm = (double)(float)((1.0 - (double)momentum) * rm + (double)momentum * mean);
        rv = (double)(float)((1.0 - (double)momentum) * rv + (double)momentum * unb);
      }
    }
    if (owner && running_mean) {
      running_mean[c] = (float)rm;
      running_var[c] = (float)rv;
    }
  } else {
    if (!owner) return;
    const float m = running_mean[c], r = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
    for (int g = 0; g < G; ++g) {
      mean_rstd[(size_t)g * 2 * C + c] = m;
      mean_rstd[(size_t)g * 2 * C + C + c] = r;
    }
  }
}

// Sync-BatchNorm (cartnet_hip.h): row[which*C + c] = column sum of parts_{a,b} (which = blockIdx.y), row[2C] = the local
// row count; out_{a,b} (optional) = the same local sums as fp32.
__global__ __launch_bounds__(1024) void cn_bn_sync_gather_kernel(const double* __restrict__ parts_a,
                                                                 const double* __restrict__ parts_b, int nparts, int C,
                                                                 double local_count, double* __restrict__ row,
                                                                 float* __restrict__ out_a, float* __restrict__ out_b) {
  __shared__ double red[64 * CN_SUM_COLS];
  const int which = blockIdx.y;
  const double t = cn_block_colsum(which ? parts_b : parts_a, nparts, C, blockIdx.x * CN_SUM_COLS, red);
  const int c = blockIdx.x * CN_SUM_COLS + threadIdx.x;
  if (blockIdx.x == 0 && which == 0 && threadIdx.x == 0) row[2 * (size_t)C] = local_count;
  if (threadIdx.x >= CN_SUM_COLS || c >= C) return;
  row[(size_t)which * C + c] = t;
  float* __restrict__ out = which ? out_b : out_a;
  if (out) out[c] = (float)t;
}

__global__ void cn_bn_sync_scale_kernel(const double* __restrict__ row, int C, double local_count,
                                        float* __restrict__ sums) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * C) return;
  const double total = row[2 * (size_t)C];
  sums[i] = total > 0.0 ? (float)(row[i] * (local_count / total)) : 0.f;
}

// node_gptr[g] = graph_ptr[min(g * group_size, Bg)], edge_gptr[g] = rowptr[node_gptr[g]]   (g <= G)
__global__ void cn_group_ptrs_kernel(const int64_t* __restrict__ graph_ptr, int Bg, int group_size,
                                     const int* __restrict__ rowptr, int G, int* __restrict__ node_gptr,
                                     int* __restrict__ edge_gptr) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > G) return;
  const long long cr = (long long)g * group_size;
  const int n = (int)graph_ptr[cr < Bg ? cr : Bg];
  node_gptr[g] = n;
  edge_gptr[g] = rowptr[n];
}

// Per-group column sums / sums of squares of x [rows, C] (row stride ld): group g = blockIdx.y owns rows
// [row_gptr[g], row_gptr[g+1]); partial row (g * gridDim.x + blockIdx.x) of parts_sum / parts_sq.  Four independent
// row loads in flight per wave, fp64 accumulation (BatchNorm variance as E[v^2] - mean^2 needs the digits).
__global__ __launch_bounds__(256) void cn_colstats_grouped_kernel(const float* __restrict__ x, int ld, int C,
                                                                  const int* __restrict__ row_gptr,
                                                                  double* __restrict__ parts_sum,
                                                                  double* __restrict__ parts_sq) {
  __shared__ double red[NODES_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r0 = row_gptr[blockIdx.y], r1 = row_gptr[blockIdx.y + 1];
  const int prow = blockIdx.y * gridDim.x + blockIdx.x;
  const int stride = gridDim.x * NODES_PER_BLOCK;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool active = c < C;
    f64x4 ps = {0, 0, 0, 0}, pq = {0, 0, 0, 0};
    if (active) {
      for (int r = r0 + blockIdx.x * NODES_PER_BLOCK + wid; r < r1; r += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rr = r + u * stride;
          v[u] = rr < r1 ? ld4(x + (size_t)rr * ld + c) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ps[q] += (double)v[u][q];
            pq[q] += (double)v[u][q] * (double)v[u][q];
          }
      }
    }
    cn_block_store_parts_row(ps, red, parts_sum, C, c, active, wid, lane, prow);
    cn_block_store_parts_row(pq, red, parts_sq, C, c, active, wid, lane, prow);
  }
}

// Grouped statistics in two steps so that the groups reduce in parallel (one workgroup per 64 columns x group; a single
// workgroup walking 16 groups x ~200 partial rows took 160 us per BatchNorm).
// Step 1: group g's mean / rstd -> mean_rstd[g]; its mean and unbiased variance are parked, as doubles, in row 0 of the
// group's (now consumed) partial rows for step 2.
__global__ __launch_bounds__(1024) void cn_bn_group_stats_kernel(double* __restrict__ parts_sum,
                                                                 double* __restrict__ parts_sq, int nparts, int C,
                                                                 float eps, const int* __restrict__ cnt_ptr,
                                                                 float* __restrict__ mean_rstd) {
  __shared__ double red[64 * CN_SUM_COLS];
  const int g = blockIdx.y;
  const int c = blockIdx.x * CN_SUM_COLS + threadIdx.x;
  const size_t off = (size_t)g * nparts * C;
  const double s = cn_block_colsum(parts_sum + off, nparts, C, blockIdx.x * CN_SUM_COLS, red);
  const double q = cn_block_colsum(parts_sq + off, nparts, C, blockIdx.x * CN_SUM_COLS, red);
  if (threadIdx.x >= CN_SUM_COLS || c >= C) return;
  const long long cnt = (long long)(cnt_ptr[g + 1] - cnt_ptr[g]);
  const double n = (double)cnt;
  const double mean = n > 0 ? s / n : 0.0;
  double var = n > 0 ? q / n - mean * mean : 0.0;
  if (var < 0.0) var = 0.0;
  mean_rstd[(size_t)g * 2 * C + c] = (float)mean;
  mean_rstd[(size_t)g * 2 * C + C + c] = (float)(1.0 / sqrt(var + (double)eps));
  parts_sum[off + c] = mean;
  parts_sq[off + c] = cnt > 1 ? var * (n / (n - 1.0)) : var;
}

// Step 2: the G momentum updates of the running statistics, in group order (rounded to fp32 after every update, like
// the reference's buffers between two forward calls); num_batches_tracked += G.
__global__ void cn_bn_running_kernel(const double* __restrict__ parts_sum, const double* __restrict__ parts_sq,
                                     int nparts, int C, int G, float momentum, float* __restrict__ running_mean,
                                     float* __restrict__ running_var, int64_t* __restrict__ nbt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) nbt[0] += G;
  if (c >= C || !running_mean) return;
  double rm = (double)running_mean[c], rv = (double)running_var[c];
  for (int g = 0; g < G; ++g) {
    const size_t off = (size_t)g * nparts * C + c;
    rm = (double)(float)((1.0 - (double)momentum) * rm + (double)momentum * parts_sum[off]);
    rv = (double)(float)((1.0 - (double)momentum) * rv + (double)momentum * parts_sq[off]);
  }
  running_mean[c] = (float)rm;
  running_var[c] = (float)rv;
}

// Backward statistics per group: sums[g][which*D + c] = column sum of group g's partial rows (which = blockIdx.y: 0
// parts_a, 1 parts_b); the double total is parked in row 0 of the group's partial rows for cn_group_grads_kernel, which
// adds the groups in order (the BatchNorm affine gradients are shared by the groups).
__global__ __launch_bounds__(1024) void cn_group_sums_kernel(double* __restrict__ parts_a, double* __restrict__ parts_b,
                                                             int nparts, int D, float* __restrict__ sums) {
  __shared__ double red[64 * CN_SUM_COLS];
  const int which = blockIdx.y, g = blockIdx.z;
  double* __restrict__ parts = (which ? parts_b : parts_a) + (size_t)g * nparts * D;
  const int c = blockIdx.x * CN_SUM_COLS + threadIdx.x;
  const double t = cn_block_colsum(parts, nparts, D, blockIdx.x * CN_SUM_COLS, red);
  if (threadIdx.x >= CN_SUM_COLS || c >= D) return;
  sums[(size_t)g * 2 * D + which * D + c] = (float)t;
  parts[c] = t;
}

__global__ void cn_group_grads_kernel(const double* __restrict__ parts_a, const double* __restrict__ parts_b, int nparts,
                                      int D, int G, float* __restrict__ grad_a, float* __restrict__ grad_b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const double* __restrict__ parts = blockIdx.y ? parts_b : parts_a;
  float* __restrict__ grad = blockIdx.y ? grad_b : grad_a;
  if (c >= D || !grad) return;
  double total = 0.0;
  for (int g = 0; g < G; ++g) total += parts[(size_t)g * nparts * D + c];
  grad[c] = (float)total;
}

// ------------------------------------------------------------------------------------------------ node update
__global__ void cn_node_update_fwd_kernel(const float* __restrict__ aggr, const float* __restrict__ x_in,
                                          const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, int N, int D, float* __restrict__ x_out,
                                          const int* __restrict__ node_gptr) {
  int gi, bx, n0, n1;
  cn_group_range(node_gptr, N, false, gi, bx, n0, n1);
  mean_rstd += (size_t)gi * 2 * D;
  const long long base4 = (long long)n0 * D / 4, total4 = (long long)(n1 - n0) * D / 4;
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < total4;
       j += (long long)gridDim.x * blockDim.x) {
    const long long i = base4 + j;
    const int c = (int)((i * 4) % D);
    const f32x4 a = ld4(aggr + i * 4), xi = ld4(x_in + i * 4);
    const f32x4 mean = ld4(mean_rstd + c), rstd = ld4(mean_rstd + D + c), gam = ld4(gamma + c), bet = ld4(beta + c);
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = cn_silu((a[q] - mean[q]) * rstd[q] * gam[q] + bet[q]) + xi[q];
    st4(x_out + i * 4, o);
  }
}

// MODE 0: partial sums of dxn and dxn*ahat.  MODE 1: daggr.
template <int MODE>
__global__ __launch_bounds__(256) void cn_node_update_bwd_kernel(
    const float* __restrict__ aggr, const float* __restrict__ dx_out, const float* __restrict__ mean_rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ sums, float inv_count,
    int N, int D, double* __restrict__ parts_a, double* __restrict__ parts_b, float* __restrict__ daggr,
    const int* __restrict__ node_gptr, const float* __restrict__ bc) {
  __shared__ double red[NODES_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int gi, bx, n0, n1;
  cn_group_range(node_gptr, N, false, gi, bx, n0, n1);
  mean_rstd += (size_t)gi * 2 * D;
  if (MODE == 1) sums += (size_t)gi * 2 * D;
  if (MODE == 1 && node_gptr && inv_count != 0.f) inv_count = n1 > n0 ? 1.0f / (float)(n1 - n0) : 0.f;
  const int prow = gi * gridDim.x + bx;
  for (int c0 = 0; c0 < D; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool active = c < D;
    f32x4 mean = {0, 0, 0, 0}, rstd = {0, 0, 0, 0}, gam = {0, 0, 0, 0}, bet = {0, 0, 0, 0};
    f32x4 m_a = {0, 0, 0, 0}, m_b = {0, 0, 0, 0};
    if (active) {
      mean = ld4(mean_rstd + c);
      rstd = ld4(mean_rstd + D + c);
      gam = ld4(gamma + c);
      bet = ld4(beta + c);
      if (MODE == 1) {
        m_a = ld4(sums + c) * inv_count;
        m_b = ld4(sums + D + c) * inv_count;
      }
    }
    f64x4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0};
    for (int n = n0 + blockIdx.x * NODES_PER_BLOCK + wid; n < n1; n += gridDim.x * NODES_PER_BLOCK) {
      if (!active) continue;
      const f32x4 a = ld4(aggr + (size_t)n * D + c), dy = ld4(dx_out + (size_t)n * D + c);
      f32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float ahat = (a[q] - mean[q]) * rstd[q];
        const float dxn = dy[q] * cn_dsilu(ahat * gam[q] + bet[q]);
        if (MODE == 0) {
          pa[q] += (double)dxn;
          pb[q] += (double)dxn * (double)ahat;
        } else {
          o[q] = gam[q] * rstd[q] * (dxn - m_a[q] - ahat * m_b[q]);
        }
      }
      if (MODE == 1) {
        st4(daggr + (size_t)n * D + c, o);
        if (bc) {   // daggr's share of the gate BatchNorm-backward sums: sum_t daggr[t] B[t], sum_t daggr[t] C[t]
          const f32x4 bb = ld4(bc + (size_t)n * 2 * D + c), cc = ld4(bc + (size_t)n * 2 * D + D + c);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            pa[q] += (double)(o[q] * bb[q]);
            pb[q] += (double)(o[q] * cc[q]);
          }
        }
      }
    }
    if (MODE == 0 || bc) {
      cn_block_store_parts_row(pa, red, parts_a, D, c, active, wid, lane, prow);
      cn_block_store_parts_row(pb, red, parts_b, D, c, active, wid, lane, prow);
    }
  }
}

// ------------------------------------------------------------------------------------------------ heads
// Single workgroup: out_index[n] = rank of atom n among masked atoms (or -1), count[0] = number of masked atoms.
__global__ __launch_bounds__(1024) void cn_mask_index_kernel(const uint8_t* __restrict__ mask, int N,
                                                             int* __restrict__ out_index, int* __restrict__ count) {
  __shared__ int wave_cnt[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int n0 = 0; n0 < N; n0 += 1024) {
    const int n = n0 + tid;
    const bool f = (n < N) && mask[n] != 0;
    const unsigned long long b = __ballot(f);
    const int rank_in_wave = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wid] = __popcll(b);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wid; ++w) off += wave_cnt[w];
    if (n < N) out_index[n] = f ? off + rank_in_wave : -1;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wave_cnt[w];
      base_s += t;
    }
    __syncthreads();
  }
  if (tid == 0 && count) count[0] = base_s;
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float dsoftplus_f(float x) { return x > 20.f ? 1.f : cn_sigmoid(x); }

constexpr int HEAD_MAX_H = 512;   // hidden width handled by one wave: up to 8 values per lane

__global__ __launch_bounds__(256) void cn_cholesky_head_fwd_kernel(const float* __restrict__ hid,
                                                                   const int* __restrict__ out_index,
                                                                   const float* __restrict__ W2,
                                                                   const float* __restrict__ b2, int N, int H,
                                                                   float* __restrict__ p6, float* __restrict__ pred) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int n = blockIdx.x * NODES_PER_BLOCK + wid; n < N; n += gridDim.x * NODES_PER_BLOCK) {
    const int m = out_index[n];
    if (m < 0) continue;
    float p[6] = {0, 0, 0, 0, 0, 0};
    for (int c = lane; c < H; c += 64) {
      const float hv = cn_silu(hid[(size_t)n * H + c]);
#pragma unroll
      for (int r = 0; r < 6; ++r) p[r] += hv * W2[r * H + c];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) p[r] = wave_sum(p[r]) + b2[r];
    if (lane == 0) {
      const float d0 = softplus_f(p[0]), d1 = softplus_f(p[1]), d2 = softplus_f(p[2]);
      const float a = p[3], b = p[4], c = p[5];
      // L = [[d0,a,b],[0,d1,c],[0,0,d2]] ; U = L^T L
      float* u = pred + (size_t)m * 9;
      u[0] = d0 * d0;  u[1] = d0 * a;           u[2] = d0 * b;
      u[3] = d0 * a;   u[4] = a * a + d1 * d1;  u[5] = a * b + d1 * c;
      u[6] = d0 * b;   u[7] = a * b + d1 * c;   u[8] = b * b + c * c + d2 * d2;
#pragma unroll
      for (int r = 0; r < 6; ++r) p6[(size_t)m * 6 + r] = p[r];
    }
  }
}

// dhid + per-block partials (parts row layout: 6*H dW2, 6 db2, 2 pad, H column sums of dhid = grad of head.MLP.0.bias).
__global__ __launch_bounds__(256) void cn_cholesky_head_bwd_kernel(const float* __restrict__ hid,
                                                                   const int* __restrict__ out_index,
                                                                   const float* __restrict__ W2,
                                                                   const float* __restrict__ p6,
                                                                   const float* __restrict__ dpred, int N, int H,
                                                                   float* __restrict__ dhid,
                                                                   float* __restrict__ parts) {
  __shared__ float red[NODES_PER_BLOCK][7 * HEAD_MAX_H + 8];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int per_lane = (H + 63) / 64;  // <= 8
  float wacc[6][8];
  float hacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float bacc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) wacc[r][j] = 0.f;
  for (int n = blockIdx.x * NODES_PER_BLOCK + wid; n < N; n += gridDim.x * NODES_PER_BLOCK) {
    const int m = out_index[n];
    if (m < 0) {
      for (int c = lane; c < H; c += 64) dhid[(size_t)n * H + c] = 0.f;
      continue;
    }
    const float* G = dpred + (size_t)m * 9;
    const float* p = p6 + (size_t)m * 6;
    const float d0 = softplus_f(p[0]), d1 = softplus_f(p[1]), d2 = softplus_f(p[2]);
    const float a = p[3], b = p[4], c5 = p[5];
    // S = G + G^T ; dL = L S ; keep the upper-triangular entries
    const float S00 = 2.f * G[0], S01 = G[1] + G[3], S02 = G[2] + G[6];
    const float S11 = 2.f * G[4], S12 = G[5] + G[7], S22 = 2.f * G[8];
    float dp[6];
    dp[0] = (d0 * S00 + a * S01 + b * S02) * dsoftplus_f(p[0]);   // dL00
    dp[3] = d0 * S01 + a * S11 + b * S12;                          // dL01
    dp[4] = d0 * S02 + a * S12 + b * S22;                          // dL02
    dp[1] = (d1 * S11 + c5 * S12) * dsoftplus_f(p[1]);            // dL11
    dp[5] = d1 * S12 + c5 * S22;                                   // dL12
    dp[2] = (d2 * S22) * dsoftplus_f(p[2]);                        // dL22
#pragma unroll
    for (int r = 0; r < 6; ++r) bacc[r] += dp[r];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = lane + j * 64;
      if (j < per_lane && c < H) {
        const float hp = hid[(size_t)n * H + c];
        const float hv = cn_silu(hp);
        float dh = 0.f;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          dh += dp[r] * W2[r * H + c];
          wacc[r][j] += dp[r] * hv;
        }
        dh *= cn_dsilu(hp);
        hacc[j] += dh;
        dhid[(size_t)n * H + c] = dh;
      }
    }
  }
  // block partials
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = lane + j * 64;
      if (c < H) red[wid][r * H + c] = wacc[r][j];
    }
    if (lane == 0) red[wid][6 * H + r] = bacc[r];
  }
  if (lane < 2) red[wid][6 * H + 6 + lane] = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = lane + j * 64;
    if (c < H) red[wid][6 * H + 8 + c] = hacc[j];
  }
  __syncthreads();
  const int row = 7 * H + 8;
  for (int i = threadIdx.x; i < row; i += 256) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NODES_PER_BLOCK; ++w) t += red[w][i];
    parts[(size_t)blockIdx.x * row + i] = t;
  }
}

// Scalar head: one wave per crystal.  The two halves of the wave take alternate atoms, a lane four columns (16-byte
// loads; H % 4 == 0 as D % 8 == 0): every load of the crystal is independent of every other and ONE wave reduction closes
// it -- the atom-at-a-time form (a wave sum per atom) was a chain of one memory round trip per atom, 17 us for 2-20 atoms.
__global__ __launch_bounds__(64) void cn_scalar_head_fwd_kernel(const float* __restrict__ hid,
                                                                const float* __restrict__ w2,
                                                                const float* __restrict__ b2,
                                                                const int64_t* __restrict__ graph_ptr, int Bg, int H,
                                                                float* __restrict__ out) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int n0 = (int)graph_ptr[g], n1 = (int)graph_ptr[g + 1];
  const float bias = b2[0];
  const int half = lane >> 5, c4 = (lane & 31) * 4;
  float acc = 0.f;
  for (int c = c4; c < H; c += 128) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(w2 + c);
#pragma unroll 4
    for (int n = n0 + half; n < n1; n += 2) {
      const f32x4 h = *reinterpret_cast<const f32x4*>(hid + (size_t)n * H + c);
      acc += cn_silu(h[0]) * w[0] + cn_silu(h[1]) * w[1] + cn_silu(h[2]) * w[2] + cn_silu(h[3]) * w[3];
    }
  }
  const int cnt = n1 - n0;
  const float tot = wave_sum(acc) + (float)cnt * bias;
  if (lane == 0) out[g] = tot / (float)(cnt > 0 ? cnt : 1);
}

__global__ __launch_bounds__(256) void cn_scalar_head_bwd_kernel(const float* __restrict__ hid,
                                                                 const float* __restrict__ w2,
                                                                 const int64_t* __restrict__ graph_ptr,
                                                                 const int64_t* __restrict__ batch,
                                                                 const float* __restrict__ dout, int N, int Bg, int H,
                                                                 float* __restrict__ dhid,
                                                                 float* __restrict__ parts) {
  __shared__ float red[NODES_PER_BLOCK][2 * HEAD_MAX_H + 8];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float wacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float hacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float bacc = 0.f;
  for (int n = blockIdx.x * NODES_PER_BLOCK + wid; n < N; n += gridDim.x * NODES_PER_BLOCK) {
    long long gl = batch[n];
    const int g = (int)(gl < 0 ? 0 : (gl >= Bg ? Bg - 1 : gl));
    const int cnt = (int)(graph_ptr[g + 1] - graph_ptr[g]);
    const float dv = dout[g] / (float)(cnt > 0 ? cnt : 1);
    bacc += dv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = lane + j * 64;
      if (c < H) {
        const float hp = hid[(size_t)n * H + c];
        wacc[j] += dv * cn_silu(hp);
        const float dh = dv * w2[c] * cn_dsilu(hp);
        hacc[j] += dh;
        dhid[(size_t)n * H + c] = dh;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = lane + j * 64;
    if (c < H) {
      red[wid][c] = wacc[j];
      red[wid][H + 8 + c] = hacc[j];
    }
  }
  if (lane < 8) red[wid][H + lane] = (lane == 0) ? bacc : 0.f;
  __syncthreads();
  const int row = 2 * H + 8;
  for (int i = threadIdx.x; i < row; i += 256) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NODES_PER_BLOCK; ++w) t += red[w][i];
    parts[(size_t)blockIdx.x * row + i] = t;
  }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int cartnet_node_embed(const int64_t* z, const int64_t* batch, const float* temperature, const float* emb,
                                  const float* wt, const float* bt, const float* bias, int32_t N, int32_t C,
                                  int32_t n_types, int32_t Bg, int32_t* status, float* x0, void* stream) {
  CN_CHECK(N >= 0 && C >= 1, "cartnet_node_embed: bad sizes");
  if (N == 0) return 0;
  CN_CHECK(x0 && (emb || wt), "cartnet_node_embed: need an embedding table or a temperature projection");
  CN_CHECK(!emb || n_types >= 1, "cartnet_node_embed: the embedding table needs n_types >= 1 rows");
  CN_CHECK(!wt || (bt && temperature && batch && Bg >= 1),
           "cartnet_node_embed: temperature projection needs wt, bt, T, batch and Bg >= 1");
  long long blocks = ((long long)N * C + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cn_node_embed_kernel, dim3((int)blocks), dim3(256), 0, ST(stream), z, batch, temperature, emb,
                     wt, bt, bias, N, C, n_types, Bg, status, x0);
  CN_LAUNCH_CHECK("cartnet_node_embed");
  return 0;
}

extern "C" int cartnet_node_nparts(int32_t N) { return node_parts(N); }

extern "C" int cartnet_node_embed_bwd(const int64_t* batch, const float* temperature, const float* dx0, int32_t N,
                                      int32_t C, int32_t Bg, double* parts_w, double* parts_b, void* stream) {
  CN_CHECK(N >= 0 && C >= 4 && C % 4 == 0, "cartnet_node_embed_bwd: C=%d must be a multiple of 4", C);
  CN_CHECK((dx0 || N == 0) && parts_w && parts_b, "cartnet_node_embed_bwd: null pointer");
  CN_CHECK(!temperature || (batch && Bg >= 1), "cartnet_node_embed_bwd: temperature needs batch and Bg >= 1");
  hipLaunchKernelGGL(cn_embed_bwd_cols_kernel, dim3(node_parts(N)), dim3(256), 0, ST(stream), batch, temperature, dx0,
                     N, C, Bg, parts_w, parts_b);
  CN_LAUNCH_CHECK("cartnet_node_embed_bwd");
  return 0;
}

extern "C" int cartnet_bn_finalize(double* parts_sum, double* parts_sq, int32_t nparts, int64_t count,
                                   int32_t C, float eps, float momentum, int32_t training, float* running_mean,
                                   float* running_var, int64_t* num_batches_tracked, float* mean_rstd,
                                   const CartnetGroups* groups, int32_t parts_over_edges, int32_t count_over_edges,
                                   void* stream) {
  CN_CHECK(C >= 1 && mean_rstd, "cartnet_bn_finalize: bad arguments");
  if (training) CN_CHECK(parts_sum && parts_sq && nparts >= 0 && count >= 0, "cartnet_bn_finalize: missing partial sums");
  else CN_CHECK(running_mean && running_var, "cartnet_bn_finalize: eval mode needs running statistics");
  CN_CHECK((running_mean == nullptr) == (running_var == nullptr), "cartnet_bn_finalize: running stats must pair");
  CN_CHECK(cn_groups_ok(groups), "cartnet_bn_finalize: bad groups");
  int G = 1;
  if (groups) {   // [G][parts][C] partial rows, one statistics row per group
    G = groups->G;
    nparts = parts_over_edges ? groups->edge_parts : groups->node_parts;
    if (training) {
      hipLaunchKernelGGL(cn_bn_group_stats_kernel, dim3(cn_ceil_div(C, CN_SUM_COLS), G), dim3(1024), 0, ST(stream), parts_sum,
                         parts_sq, nparts, C, eps, count_over_edges ? groups->edge_gptr : groups->node_gptr, mean_rstd);
      CN_LAUNCH_CHECK("cartnet_bn_finalize/groups");
      hipLaunchKernelGGL(cn_bn_running_kernel, dim3(cn_ceil_div(C, 256)), dim3(256), 0, ST(stream), parts_sum, parts_sq,
                         nparts, C, G, momentum, running_mean, running_var, num_batches_tracked);
      CN_LAUNCH_CHECK("cartnet_bn_finalize/running");
      return 0;
    }
  }
  hipLaunchKernelGGL(cn_bn_finalize_kernel, dim3(cn_ceil_div(C, CN_SUM_COLS)), dim3(1024), 0, ST(stream), parts_sum, parts_sq,
                     nparts, (long long)count, C, eps, momentum, training, running_mean, running_var,
                     num_batches_tracked, mean_rstd, G, (const int*)nullptr, (const double*)nullptr);
  CN_LAUNCH_CHECK("cartnet_bn_finalize");
  return 0;
}

extern "C" int cartnet_bn_sync_gather(const double* parts_a, const double* parts_b, int32_t nparts, int32_t C,
                                      int64_t local_count, double* row, float* out_a, float* out_b, void* stream) {
  CN_CHECK(parts_a && parts_b && row && nparts >= 0 && C >= 1 && local_count >= 0, "cartnet_bn_sync_gather: bad arguments");
  hipLaunchKernelGGL(cn_bn_sync_gather_kernel, dim3(cn_ceil_div(C, CN_SUM_COLS), 2), dim3(1024), 0, ST(stream), parts_a,
                     parts_b, nparts, C, (double)local_count, row, out_a, out_b);
  CN_LAUNCH_CHECK("cartnet_bn_sync_gather");
  return 0;
}

extern "C" int cartnet_bn_finalize_row(const double* row, int32_t C, float eps, float momentum, float* running_mean,
                                       float* running_var, int64_t* num_batches_tracked, float* mean_rstd, void* stream) {
  CN_CHECK(row && mean_rstd && C >= 1, "cartnet_bn_finalize_row: bad arguments");
  CN_CHECK((running_mean == nullptr) == (running_var == nullptr), "cartnet_bn_finalize_row: running stats must pair");
  hipLaunchKernelGGL(cn_bn_finalize_kernel, dim3(cn_ceil_div(C, CN_SUM_COLS)), dim3(1024), 0, ST(stream), row, row + C, 1,
                     (long long)0, C, eps, momentum, 1, running_mean, running_var, num_batches_tracked, mean_rstd, 1,
                     (const int*)nullptr, row + 2 * (size_t)C);
  CN_LAUNCH_CHECK("cartnet_bn_finalize_row");
  return 0;
}

extern "C" int cartnet_bn_sync_scale(const double* row, int32_t C, int64_t local_count, float* sums, void* stream) {
  CN_CHECK(row && sums && C >= 1 && local_count >= 0, "cartnet_bn_sync_scale: bad arguments");
  hipLaunchKernelGGL(cn_bn_sync_scale_kernel, dim3(cn_ceil_div(2 * C, 256)), dim3(256), 0, ST(stream), row, C,
                     (double)local_count, sums);
  CN_LAUNCH_CHECK("cartnet_bn_sync_scale");
  return 0;
}

extern "C" int cartnet_group_ptrs(const int64_t* graph_ptr, int32_t Bg, int32_t group_size, const int32_t* rowptr,
                                  int32_t G, int32_t* node_gptr, int32_t* edge_gptr, void* stream) {
  CN_CHECK(graph_ptr && rowptr && node_gptr && edge_gptr, "cartnet_group_ptrs: null pointer");
  CN_CHECK(Bg >= 1 && group_size >= 1 && G == (Bg + group_size - 1) / group_size,
           "cartnet_group_ptrs: G=%d does not match ceil(Bg=%d / group_size=%d)", G, Bg, group_size);
  hipLaunchKernelGGL(cn_group_ptrs_kernel, dim3(cn_ceil_div(G + 1, 256)), dim3(256), 0, ST(stream), graph_ptr, Bg,
                     group_size, rowptr, G, node_gptr, edge_gptr);
  CN_LAUNCH_CHECK("cartnet_group_ptrs");
  return 0;
}

extern "C" int cartnet_colstats_grouped(const float* x, int32_t ld, int32_t C, const CartnetGroups* groups,
                                        double* parts_sum, double* parts_sq, void* stream) {
  CN_CHECK(groups && cn_groups_ok(groups), "cartnet_colstats_grouped: groups required");
  CN_CHECK(C >= 4 && C % 4 == 0 && ld % 4 == 0 && ld >= C, "cartnet_colstats_grouped: C=%d ld=%d must be multiples of 4", C, ld);
  CN_CHECK(x && parts_sum && parts_sq, "cartnet_colstats_grouped: null pointer");
  hipLaunchKernelGGL(cn_colstats_grouped_kernel, dim3(groups->edge_parts, groups->G), dim3(256), 0, ST(stream), x, ld, C,
                     groups->edge_gptr, parts_sum, parts_sq);
  CN_LAUNCH_CHECK("cartnet_colstats_grouped");
  return 0;
}

extern "C" int cartnet_group_sums_finalize(double* parts_a, double* parts_b, int32_t D, const CartnetGroups* groups,
                                           int32_t over_edges, float* sums, float* grad_a, float* grad_b,
                                           void* stream) {
  CN_CHECK(groups && cn_groups_ok(groups), "cartnet_group_sums_finalize: groups required");
  CN_CHECK(parts_a && parts_b && sums && D >= 1, "cartnet_group_sums_finalize: null pointer");
  const int nparts = over_edges ? groups->edge_parts : groups->node_parts;
  hipLaunchKernelGGL(cn_group_sums_kernel, dim3(cn_ceil_div(D, CN_SUM_COLS), 2, groups->G), dim3(1024), 0, ST(stream), parts_a,
                     parts_b, nparts, D, sums);
  CN_LAUNCH_CHECK("cartnet_group_sums_finalize");
  if (grad_a || grad_b) {
    hipLaunchKernelGGL(cn_group_grads_kernel, dim3(cn_ceil_div(D, 256), 2), dim3(256), 0, ST(stream), parts_a, parts_b,
                       nparts, D, groups->G, grad_a, grad_b);
    CN_LAUNCH_CHECK("cartnet_group_sums_finalize/grads");
  }
  return 0;
}

extern "C" int cartnet_node_update_fwd(const float* aggr, const float* x_in, const float* mean_rstd,
                                       const float* gamma, const float* beta, int32_t N, int32_t D, float* x_out,
                                       const CartnetGroups* groups, void* stream) {
  CN_CHECK(N >= 0 && D >= 4 && D % 4 == 0, "cartnet_node_update_fwd: D=%d must be a multiple of 4", D);
  if (N == 0) return 0;
  CN_CHECK(aggr && x_in && mean_rstd && gamma && beta && x_out, "cartnet_node_update_fwd: null pointer");
  CN_CHECK(cn_groups_ok(groups), "cartnet_node_update_fwd: bad groups");
  const int G = groups ? groups->G : 1;
  long long blocks = ((long long)N * D / 4 / G + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cn_node_update_fwd_kernel, dim3((int)blocks, G), dim3(256), 0, ST(stream), aggr, x_in, mean_rstd,
                     gamma, beta, N, D, x_out, groups ? groups->node_gptr : nullptr);
  CN_LAUNCH_CHECK("cartnet_node_update_fwd");
  return 0;
}

extern "C" int cartnet_node_update_bwd_stats(const float* aggr, const float* dx_out, const float* mean_rstd,
                                             const float* gamma, const float* beta, int32_t N, int32_t D,
                                             double* parts_a, double* parts_b, const CartnetGroups* groups,
                                             void* stream) {
  CN_CHECK(N >= 0 && D >= 4 && D % 4 == 0, "cartnet_node_update_bwd_stats: D=%d must be a multiple of 4", D);
  CN_CHECK(aggr && dx_out && mean_rstd && gamma && beta && parts_a && parts_b,
           "cartnet_node_update_bwd_stats: null pointer");
  CN_CHECK(cn_groups_ok(groups), "cartnet_node_update_bwd_stats: bad groups");
  hipLaunchKernelGGL(cn_node_update_bwd_kernel<0>, cn_group_grid(groups, node_parts(N), false), dim3(256), 0, ST(stream),
                     aggr, dx_out, mean_rstd, gamma, beta, (const float*)nullptr, 0.f, N, D, parts_a, parts_b,
                     (float*)nullptr, groups ? groups->node_gptr : nullptr, (const float*)nullptr);
  CN_LAUNCH_CHECK("cartnet_node_update_bwd_stats");
  return 0;
}

extern "C" int cartnet_node_update_bwd_apply(const float* aggr, const float* dx_out, const float* mean_rstd,
                                             const float* gamma, const float* beta, const float* sums,
                                             int32_t training, int32_t N, int32_t D, float* daggr,
                                             const CartnetGroups* groups, void* stream) {
  CN_CHECK(N >= 0 && D >= 4 && D % 4 == 0, "cartnet_node_update_bwd_apply: D=%d must be a multiple of 4", D);
  if (N == 0) return 0;
  CN_CHECK(aggr && dx_out && mean_rstd && gamma && beta && sums && daggr, "cartnet_node_update_bwd_apply: null pointer");
  CN_CHECK(cn_groups_ok(groups), "cartnet_node_update_bwd_apply: bad groups");
  const float inv = (training && N > 0) ? (float)(1.0 / (double)N) : 0.f;
  hipLaunchKernelGGL(cn_node_update_bwd_kernel<1>, cn_group_grid(groups, node_parts(N), false), dim3(256), 0, ST(stream),
                     aggr, dx_out, mean_rstd, gamma, beta, sums, inv, N, D, (double*)nullptr, (double*)nullptr, daggr,
                     groups ? groups->node_gptr : nullptr, (const float*)nullptr);
  CN_LAUNCH_CHECK("cartnet_node_update_bwd_apply");
  return 0;
}

extern "C" int cartnet_node_update_bwd_apply_bc(const float* aggr, const float* dx_out, const float* mean_rstd,
                                                const float* gamma, const float* beta, const float* sums,
                                                int32_t training, int32_t N, int32_t D, float* daggr, const float* bc,
                                                double* parts_a, double* parts_b, void* stream) {
  CN_CHECK(N >= 0 && D >= 4 && D % 4 == 0, "cartnet_node_update_bwd_apply_bc: D=%d must be a multiple of 4", D);
  CN_CHECK(aggr && dx_out && mean_rstd && gamma && beta && sums && daggr && bc && parts_a && parts_b,
           "cartnet_node_update_bwd_apply_bc: null pointer");
  const float inv = (training && N > 0) ? (float)(1.0 / (double)N) : 0.f;
  // (N == 0 still launches: the workgroups write their zero partial rows)
  hipLaunchKernelGGL(cn_node_update_bwd_kernel<1>, dim3(node_parts(N), 1), dim3(256), 0, ST(stream), aggr, dx_out, mean_rstd,
                     gamma, beta, sums, inv, N, D, parts_a, parts_b, daggr, (const int*)nullptr, bc);
  CN_LAUNCH_CHECK("cartnet_node_update_bwd_apply_bc");
  return 0;
}

extern "C" int cartnet_mask_index(const uint8_t* mask, int32_t N, int32_t* out_index, int32_t* count, void* stream) {
  CN_CHECK(N >= 0 && (N == 0 || (mask && out_index)), "cartnet_mask_index: bad arguments");
  hipLaunchKernelGGL(cn_mask_index_kernel, dim3(1), dim3(1024), 0, ST(stream), mask, N, out_index, count);
  CN_LAUNCH_CHECK("cartnet_mask_index");
  return 0;
}

extern "C" int cartnet_cholesky_head_fwd(const float* hid, const int32_t* out_index, const float* W2, const float* b2,
                                         int32_t N, int32_t H, float* p6, float* pred, void* stream) {
  CN_CHECK(N >= 0 && H >= 1 && H <= HEAD_MAX_H, "cartnet_cholesky_head_fwd: H=%d out of range (max %d)", H, HEAD_MAX_H);
  if (N == 0) return 0;
  CN_CHECK(hid && out_index && W2 && b2 && p6 && pred, "cartnet_cholesky_head_fwd: null pointer");
  hipLaunchKernelGGL(cn_cholesky_head_fwd_kernel, dim3(cn_ceil_div(N, 4) > 2048 ? 2048 : cn_ceil_div(N, 4)), dim3(256),
                     0, ST(stream), hid, out_index, W2, b2, N, H, p6, pred);
  CN_LAUNCH_CHECK("cartnet_cholesky_head_fwd");
  return 0;
}

extern "C" int cartnet_cholesky_head_bwd(const float* hid, const int32_t* out_index, const float* W2, const float* p6,
                                         const float* dpred, int32_t N, int32_t H, float* dhid, float* parts,
                                         void* stream) {
  CN_CHECK(N >= 0 && H >= 1 && H <= HEAD_MAX_H, "cartnet_cholesky_head_bwd: H=%d out of range (max %d)", H, HEAD_MAX_H);
  CN_CHECK(hid && out_index && W2 && p6 && dpred && dhid && parts, "cartnet_cholesky_head_bwd: null pointer");
  hipLaunchKernelGGL(cn_cholesky_head_bwd_kernel, dim3(node_parts(N)), dim3(256), 0, ST(stream), hid, out_index, W2,
                     p6, dpred, N, H, dhid, parts);
  CN_LAUNCH_CHECK("cartnet_cholesky_head_bwd");
  return 0;
}

extern "C" int cartnet_scalar_head_fwd(const float* hid, const float* w2, const float* b2, const int64_t* graph_ptr,
                                       int32_t Bg, int32_t H, float* out, void* stream) {
  CN_CHECK(Bg >= 0 && H >= 1, "cartnet_scalar_head_fwd: bad sizes");
  if (Bg == 0) return 0;
  CN_CHECK(hid && w2 && b2 && graph_ptr && out, "cartnet_scalar_head_fwd: null pointer");
  hipLaunchKernelGGL(cn_scalar_head_fwd_kernel, dim3(Bg), dim3(64), 0, ST(stream), hid, w2, b2, graph_ptr, Bg, H, out);
  CN_LAUNCH_CHECK("cartnet_scalar_head_fwd");
  return 0;
}

extern "C" int cartnet_scalar_head_bwd(const float* hid, const float* w2, const int64_t* graph_ptr,
                                       const int64_t* batch, const float* dout, int32_t N, int32_t Bg, int32_t H,
                                       float* dhid, float* parts, void* stream) {
  CN_CHECK(N >= 0 && Bg >= 0 && H >= 1 && H <= HEAD_MAX_H, "cartnet_scalar_head_bwd: H=%d out of range", H);
  CN_CHECK(hid && w2 && graph_ptr && batch && dout && dhid && parts, "cartnet_scalar_head_bwd: null pointer");
  hipLaunchKernelGGL(cn_scalar_head_bwd_kernel, dim3(node_parts(N)), dim3(256), 0, ST(stream), hid, w2, graph_ptr,
                     batch, dout, N, Bg, H, dhid, parts);
  CN_LAUNCH_CHECK("cartnet_scalar_head_bwd");
  return 0;
}

extern "C" int cartnet_transpose(const float* const* src, float* const* dst, const int32_t* rows, const int32_t* cols,
                                 const int32_t* lds, const int32_t* ldd, int32_t njobs, void* stream) {
  CN_CHECK(src && dst && rows && cols && lds && ldd && njobs >= 1 && njobs <= TRANSPOSE_MAX_JOBS,
           "cartnet_transpose: njobs=%d out of range (1..%d)", njobs, TRANSPOSE_MAX_JOBS);
  TransposeJobs jobs;
  int max_r = 0, max_c = 0;
  for (int j = 0; j < TRANSPOSE_MAX_JOBS; ++j) {
    const bool on = j < njobs;
    jobs.src[j] = on ? src[j] : nullptr;
    jobs.dst[j] = on ? dst[j] : nullptr;
    jobs.rows[j] = on ? rows[j] : 0;
    jobs.cols[j] = on ? cols[j] : 0;
    jobs.lds[j] = on ? lds[j] : 0;
    jobs.ldd[j] = on ? ldd[j] : 0;
    if (on) {
      CN_CHECK(src[j] && dst[j] && rows[j] >= 0 && cols[j] >= 0 && lds[j] >= cols[j] && ldd[j] >= rows[j],
               "cartnet_transpose: bad job %d", j);
      if (rows[j] > max_r) max_r = rows[j];
      if (cols[j] > max_c) max_c = cols[j];
    }
  }
  if (max_r == 0 || max_c == 0) return 0;
  hipLaunchKernelGGL(cn_transpose_kernel, dim3(cn_ceil_div(max_c, 32), cn_ceil_div(max_r, 32), njobs), dim3(256), 0,
                     ST(stream), jobs);
  CN_LAUNCH_CHECK("cartnet_transpose");
  return 0;
}
