// Kernels that only the iComformer path needs (reference: models/comformer.py:75-132, models/comformer_conv.py:21-193):
// Gaussian RBF expansion, lattice length / angle features, element-wise softplus pieces, the query x key product with
// its BatchNorm statistics, and the softplus(x + bn(o)) residual update.  The dense work reuses cartnet_gemm, the
// gated aggregation reuses cartnet_gate_scatter_*, reductions reuse the fp64 partial-sum machinery.
#include "common.h"
#include <math.h>

namespace {

constexpr int NODES_PER_BLOCK = 4;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// softplus(x) = max(x, 0) + log1p(exp(-|x|)) on the hardware exp / log (the libm log1pf(expf(x)) made the element-wise
// softplus passes compute-bound: 2.1 TB/s on the 3E x C angle features against 5.2 TB/s for the other element-wise ops).
// t = exp(-|x|) is in (0, 1]; below 2^-11 the series t - t^2/2 is exact to fp32 where log(1 + t) would lose t's low bits.
__device__ __forceinline__ float softplus_f(float x) {
  if (x > 20.f) return x;        // (as torch.nn.functional.softplus: threshold 20)
  const float t = __expf(-fabsf(x));
  const float l = t < 4.8828125e-4f ? t - 0.5f * t * t : __logf(1.0f + t);
  return fmaxf(x, 0.f) + l;
}

inline int seg_parts(int S) {
  int b = cn_ceil_div(S, NODES_PER_BLOCK);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return b;
}

// out[r, k] = exp(-gamma (v[r] - centers[k])^2)      (models/utils.py:125-129)
__global__ void cn_rbf_expand_kernel(const float* __restrict__ v, long long n, const float* __restrict__ centers,
                                     int bins, float gamma, float* __restrict__ out, int ldo) {
  const long long total = n * bins;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / bins;
    const int k = (int)(i - r * bins);
    const float d = v[r] - centers[k];
    out[r * ldo + k] = expf(-gamma * (d * d));
  }
}

// The same for bins % 4 == 0 with 256 % (bins / 4) == 0 (the models' bins = C): a thread owns four fixed columns and
// walks rows -- no 64-bit division per element, one 16-byte store per thread and row, v_exp_f32 instead of expf (round 3
// note: the element-indexed form above took 0.5 ms per iComformer step)
__global__ __launch_bounds__(256) void cn_rbf_expand4_kernel(const float* __restrict__ v, long long n,
                                                             const float* __restrict__ centers, int q /* bins / 4 */,
                                                             float gamma, float* __restrict__ out, int ldo) {
  const int c4 = threadIdx.x % q, rl = threadIdx.x / q, rpb = 256 / q;
  const f32x4 cen = *reinterpret_cast<const f32x4*>(centers + 4 * c4);
  for (long long r = (long long)blockIdx.x * rpb + rl; r < n; r += (long long)gridDim.x * rpb) {
    const float x = v[r];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = x - cen[j];
      o[j] = __expf(-gamma * (d * d));
    }
    *reinterpret_cast<f32x4*>(out + r * ldo + 4 * c4) = o;
  }
}

// edge_feat[e] = -0.75 / dist[e];  nei_len[g, i] = -0.75 / |cell[g, i]|;
// nei_cos[e, i] = clamp(<cell[g(e), i], dir[e]> / (|cell[g(e), i]| |dir[e]|), -1, 1), g(e) = batch[src[e]]
// (models/comformer.py:117-120, bond_cosine :18-23)
__global__ void cn_lattice_features_kernel(const float* __restrict__ cell, const int64_t* __restrict__ batch,
                                           const int* __restrict__ src, const float* __restrict__ dist,
                                           const float* __restrict__ dir, long long E, int Bg,
                                           float* __restrict__ edge_feat, float* __restrict__ nei_len,
                                           float* __restrict__ nei_cos) {
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = i0; i < (long long)Bg * 3; i += stride) {
    const float* c = cell + i * 3;
    nei_len[i] = -0.75f / sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  }
  for (long long e = i0; e < E; e += stride) {
    edge_feat[e] = -0.75f / dist[e];
    const int g = (int)batch[src[e]];
    const float dx = dir[e * 3], dy = dir[e * 3 + 1], dz = dir[e * 3 + 2];
    const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float* c = cell + ((size_t)g * 3 + a) * 3;
      const float cn = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
      float v = (c[0] * dx + c[1] * dy + c[2] * dz) / (cn * dn);
      v = fminf(fmaxf(v, -1.f), 1.f);
      nei_cos[e * 3 + a] = v;
    }
  }
}

// Element-wise helpers on [rows, cols] views (16-byte aligned rows).
// op 0: out = softplus(a); 1: out = a * sigmoid(b)  (softplus backward); 2: out = a + b; 3: out = a * scale
__global__ void cn_eltwise_kernel(int op, const float* a, const float* b, float* out, long long rows, int cols4,
                                  int lda, int ldb, int ldo, float scale) {
  const long long total = rows * cols4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols4;
    const int c = (int)(i - r * cols4) * 4;
    const f32x4 x = ld4(a + r * lda + c);
    f32x4 y = {0, 0, 0, 0};
    if (op == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = softplus_f(x[q]);
    } else if (op == 1) {
      const f32x4 p = ld4(b + r * ldb + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = x[q] * (p[q] > 20.f ? 1.f : cn_sigmoid(p[q]));
    } else if (op == 2) {
      y = x + ld4(b + r * ldb + c);
    } else {
      y = x * scale;
    }
    st4(out + r * ldo + c, y);
  }
}

// alpha[r] = key[r] * q[s] * scale for the rows r of segment s (+ fp64 column statistics of alpha).
__global__ __launch_bounds__(256) void cn_rowmul_fwd_kernel(const float* __restrict__ key, int ldk,
                                                            const float* __restrict__ q, int ldq,
                                                            const int* __restrict__ ptr, int S, int C, float scale,
                                                            float* __restrict__ alpha, int lda,
                                                            double* __restrict__ parts_sum,
                                                            double* __restrict__ parts_sq) {
  __shared__ double red[NODES_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool active = c < C;
    f64x4 ps = {0, 0, 0, 0}, pq = {0, 0, 0, 0};
    for (int s = blockIdx.x * NODES_PER_BLOCK + wid; s < S; s += gridDim.x * NODES_PER_BLOCK) {
      if (!active) continue;
      const f32x4 qv = ld4(q + (size_t)s * ldq + c) * scale;
      for (int r = ptr[s]; r < ptr[s + 1]; ++r) {
        const f32x4 a = ld4(key + (size_t)r * ldk + c) * qv;
        if (alpha) st4(alpha + (size_t)r * lda + c, a);      // NULL: statistics only (alpha is never materialised)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ps[k] += (double)a[k];
          pq[k] += (double)a[k] * (double)a[k];
        }
      }
    }
    cn_block_store_parts(ps, red, parts_sum, C, c, active, wid, lane);
    cn_block_store_parts(pq, red, parts_sq, C, c, active, wid, lane);
  }
}

// dalpha (in place) <- dkey = dalpha * q[s] * scale;  dq[s] = sum_r dalpha[r] * key[r] * scale   (fixed row order)
// SUMS (C <= 256, one column chunk): also the fp64 column partials of dkey and of dq, one row per workgroup -- the bias
// gradients of key_update.2 and lin_query, which were a pass over dkey (181 MB at the benchmark batch, 544 MB in the edge
// layer) and one over dq each.
template <bool SUMS>
__global__ __launch_bounds__(256) void cn_rowmul_bwd_kernel(float* dalpha, int lda, const float* __restrict__ key,
                                                            int ldk, const float* __restrict__ q, int ldq,
                                                            const int* __restrict__ ptr, int S, int C, float scale,
                                                            float* __restrict__ dq, int lddq,
                                                            double* __restrict__ parts_dkey,
                                                            double* __restrict__ parts_dq) {
  __shared__ double red[SUMS ? NODES_PER_BLOCK * 256 : 1];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int chunks = (C + 255) / 256;
  f64x4 pk = {0, 0, 0, 0}, pq = {0, 0, 0, 0};
  for (long long it = (long long)blockIdx.x * NODES_PER_BLOCK + wid; it < (long long)S * chunks;
       it += (long long)gridDim.x * NODES_PER_BLOCK) {
    const int s = (int)(it / chunks);
    const int c = (int)(it % chunks) * 256 + lane * 4;
    if (c >= C) continue;
    const f32x4 qv = ld4(q + (size_t)s * ldq + c) * scale;
    f32x4 acc = {0, 0, 0, 0};
    for (int r = ptr[s]; r < ptr[s + 1]; ++r) {
      const f32x4 da = ld4(dalpha + (size_t)r * lda + c);
      acc += da * ld4(key + (size_t)r * ldk + c);
      const f32x4 dk = da * qv;
      st4(dalpha + (size_t)r * lda + c, dk);
      if (SUMS) cn_acc4(pk, dk);
    }
    acc = acc * scale;
    st4(dq + (size_t)s * lddq + c, acc);
    if (SUMS) cn_acc4(pq, acc);
  }
  if (SUMS) {      // chunks == 1: a lane's columns are the same in every iteration
    const int c = lane * 4;
    cn_block_store_parts(pk, red, parts_dkey, C, c, c < C, wid, lane);
    cn_block_store_parts(pq, red, parts_dq, C, c, c < C, wid, lane);
  }
}

// The attention gate forward with the query x key product recomputed in place of a stored alpha (round 5):
//   gs = [key | msg] [R, 2D];  alpha = key * q[s] * scale;  z = sigmoid(bn_att(alpha));  aggr[s] = sum_r z * msg
//   BC: also B[s] = sum_r msg w, C[s] = sum_r msg w ahat with w = z (1 - z) (bc [S, 2D]: the backward pass's BatchNorm sums
//   are then sums over the segments, cartnet_coldot_bc_partial).
// cartnet_rowmul_fwd wrote alpha (R x D) and cartnet_gate_scatter_fwd read it back; here the statistics pass
// (cartnet_rowmul_fwd with alpha = NULL) only reads, and this kernel reads the key rows where that one read alpha.
constexpr int ATT_BATCH = 8;
template <bool BC>
__global__ __launch_bounds__(256) void cn_att_gate_fwd_kernel(const float* __restrict__ gs, const float* __restrict__ q, int ldq,
                                                              const int* __restrict__ ptr,
                                                              const float* __restrict__ mean_rstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float qscale, int S, int D,
                                                              float* __restrict__ aggr, float* __restrict__ bc) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int ld = 2 * D;
  const int chunks = (D + 255) / 256;
  const long long items = (long long)S * chunks;
  for (long long it0 = (long long)blockIdx.x * NODES_PER_BLOCK + wid; it0 < items;
       it0 += (long long)gridDim.x * NODES_PER_BLOCK) {
    const long long it = items - 1 - it0;      // descending: the statistics pass in front ran ascending
    const int t = (int)(it / chunks);
    const int c = (int)(it % chunks) * 256 + lane * 4;
    if (c >= D) continue;
    const f32x4 mean = ld4(mean_rstd + c), rstd = ld4(mean_rstd + D + c), gam = ld4(gamma + c), shift = ld4(beta + c);
    const f32x4 qv = ld4(q + (size_t)t * ldq + c) * qscale;
    const int k0 = ptr[t], k1 = ptr[t + 1];
    f32x4 acc = {0, 0, 0, 0}, accb = {0, 0, 0, 0}, accc = {0, 0, 0, 0};
    for (int k = k0; k < k1; k += ATT_BATCH) {
      f32x4 kv[ATT_BATCH], mv[ATT_BATCH];
#pragma unroll
      for (int u = 0; u < ATT_BATCH; ++u) {
        const int kk = min(k + u, k1 - 1);
#ifndef CN_NO_GATE_LOAD_NT      /* last use of the key / msg rows before backward (as in cn_gate_scatter_fwd_kernel) */
        kv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gs + (size_t)kk * ld + c));
        mv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gs + (size_t)kk * ld + D + c));
#else
        kv[u] = ld4(gs + (size_t)kk * ld + c);
        mv[u] = ld4(gs + (size_t)kk * ld + D + c);
#endif
      }
#pragma unroll
      for (int u = 0; u < ATT_BATCH; ++u) {
        if (k + u >= k1) break;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float ahat = (kv[u][i] * qv[i] - mean[i]) * rstd[i];
          const float z = cn_sigmoid(ahat * gam[i] + shift[i]);
          acc[i] += z * mv[u][i];
          if (BC) {
            const float sw = mv[u][i] * z * (1.0f - z);
            accb[i] += sw;
            accc[i] += sw * ahat;
          }
        }
      }
    }
    st4(aggr + (size_t)t * D + c, acc);
    if (BC) {
      st4(bc + (size_t)t * ld + c, accb);
      st4(bc + (size_t)t * ld + D + c, accc);
    }
  }
}

// The attention gate's backward and the query x key product's backward in ONE pass over the rows (round 5).
//   forward (comformer_conv.py:90-99): alpha = key * q[s] * scale, z = sigmoid(bn_att(alpha)), aggr[s] = sum_r z * msg
//   here, per row r of segment s, with gs = [alpha | msg] in place -> [dkey | dmsg]:
//     dbn = daggr[s] * msg * z (1 - z);  dalpha = gamma rstd (dbn - sum_a / n - ahat sum_b / n);  dmsg = daggr[s] * z
//     dkey = dalpha * q[s] * scale;       dq[s] = scale * sum_r dalpha * key
//   + fp64 column partials of dkey, dmsg and dq (the bias gradients of key_update.2, lin_msg_update.2 and lin_query).
// As cartnet_gate_scatter_bwd_apply followed by cartnet_rowmul_bwd_sums it wrote dalpha, read it back and wrote dkey over
// it: one write and one read of [R, C] per attention block (181 MB each at the benchmark batch, 544 MB in the edge layer).
// Descending sweep: the statistics pass in front (or the products that wrote daggr) ran ascending.
// KEY_IN_GS: gs[:, :D] holds the key rows themselves (alpha = key * q[s] * scale is recomputed here as the forward gate
// kernel cn_att_gate_fwd_kernel recomputed it: alpha is never written), `key` is unused.
template <bool KEY_IN_GS>
__global__ __launch_bounds__(256) void cn_att_gate_bwd_kernel(
    float* gs, const float* __restrict__ key, int ldk, const float* __restrict__ q, int ldq,
    const float* __restrict__ daggr, const int* __restrict__ ptr, const float* __restrict__ mean_rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ sums, float inv_count,
    float scale, int S, int D, float* __restrict__ dq, int lddq, double* __restrict__ parts_dkey,
    double* __restrict__ parts_dmsg, double* __restrict__ parts_dq) {
  __shared__ double red[NODES_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int ld = 2 * D;
  const int stride = gridDim.x * NODES_PER_BLOCK, nsweeps = (S + stride - 1) / stride;
  for (int c0 = 0; c0 < D; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool active = c < D;
    f32x4 mean = {0, 0, 0, 0}, rstd = {0, 0, 0, 0}, gam = {0, 0, 0, 0}, shift = {0, 0, 0, 0};
    f32x4 m_a = {0, 0, 0, 0}, m_b = {0, 0, 0, 0};
    if (active) {
      mean = ld4(mean_rstd + c);
      rstd = ld4(mean_rstd + D + c);
      gam = ld4(gamma + c);
      shift = ld4(beta + c);
      m_a = ld4(sums + c) * inv_count;
      m_b = ld4(sums + D + c) * inv_count;
    }
    f64x4 tk = {0, 0, 0, 0}, tm = {0, 0, 0, 0}, tq = {0, 0, 0, 0};
    for (int j = nsweeps - 1; j >= 0; --j) {
      const int t = blockIdx.x * NODES_PER_BLOCK + wid + j * stride;
      if (t >= S || !active) continue;
      const int k0 = ptr[t], k1 = ptr[t + 1];
      const f32x4 dm = ld4(daggr + (size_t)t * D + c);
      const f32x4 qv = ld4(q + (size_t)t * ldq + c) * scale;
      f32x4 pk = {0, 0, 0, 0}, pm = {0, 0, 0, 0}, accq = {0, 0, 0, 0};   // fp32 over one segment's rows, fp64 across segments
#pragma unroll 2
      for (int k = k0; k < k1; ++k) {
        f32x4 a = ld4(gs + (size_t)k * ld + c);
        const f32x4 m = ld4(gs + (size_t)k * ld + D + c);
        f32x4 kv;
        if (KEY_IN_GS) {
          kv = a;
          a = kv * qv;
        } else {
          kv = ld4(key + (size_t)k * ldk + c);
        }
        f32x4 dkv, dmv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float ahat = (a[i] - mean[i]) * rstd[i];
          const float z = cn_sigmoid(ahat * gam[i] + shift[i]);
          const float dbn = dm[i] * m[i] * z * (1.0f - z);
          const float da = gam[i] * rstd[i] * (dbn - m_a[i] - ahat * m_b[i]);
          dkv[i] = da * qv[i];
          dmv[i] = dm[i] * z;
          accq[i] += da * kv[i];
        }
        st4(gs + (size_t)k * ld + c, dkv);
        st4(gs + (size_t)k * ld + D + c, dmv);
        pk += dkv;
        pm += dmv;
      }
      accq = accq * scale;
      st4(dq + (size_t)t * lddq + c, accq);
      cn_acc4(tk, pk);
      cn_acc4(tm, pm);
      cn_acc4(tq, accq);
    }
    cn_block_store_parts(tk, red, parts_dkey, D, c, active, wid, lane);
    cn_block_store_parts(tm, red, parts_dmsg, D, c, active, wid, lane);
    cn_block_store_parts(tq, red, parts_dq, D, c, active, wid, lane);
  }
}

// y = softplus(x + bn(o))   (ComformerConv.forward, comformer_conv.py:88)
__global__ void cn_softplus_update_fwd_kernel(const float* __restrict__ o, const float* __restrict__ x,
                                              const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, long long N, int D,
                                              float* __restrict__ y) {
  const long long total4 = N * D / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % D);
    const f32x4 a = ld4(o + i * 4), xi = ld4(x + i * 4);
    const f32x4 mean = ld4(mean_rstd + c), rstd = ld4(mean_rstd + D + c), gam = ld4(gamma + c), bet = ld4(beta + c);
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = softplus_f(xi[q] + (a[q] - mean[q]) * rstd[q] * gam[q] + bet[q]);
    st4(y + i * 4, r);
  }
}

// MODE 0: du = dy * sigmoid(u), u = x + bn(o); partial sums of du and du * ohat.
// MODE 1: do = gamma * rstd * (du - sum_a/N - ohat * sum_b/N); dx = du (+ dx_add); with parts_a: column partials of do.
template <int MODE>
__global__ __launch_bounds__(256) void cn_softplus_update_bwd_kernel(
    const float* __restrict__ o, const float* __restrict__ x, const float* __restrict__ dy,
    const float* __restrict__ mean_rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ sums, float inv_count, int N, int D, double* __restrict__ parts_a,
    double* __restrict__ parts_b, float* __restrict__ d_o, const float* dx_add, float* dx) {
  __shared__ double red[NODES_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
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
    for (int n = blockIdx.x * NODES_PER_BLOCK + wid; n < N; n += gridDim.x * NODES_PER_BLOCK) {
      if (!active) continue;
      const f32x4 a = ld4(o + (size_t)n * D + c), xi = ld4(x + (size_t)n * D + c), g = ld4(dy + (size_t)n * D + c);
      f32x4 vo, vx;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float ohat = (a[q] - mean[q]) * rstd[q];
        const float u = xi[q] + ohat * gam[q] + bet[q];
        const float du = g[q] * (u > 20.f ? 1.f : cn_sigmoid(u));
        if (MODE == 0) {
          pa[q] += (double)du;
          pb[q] += (double)du * (double)ohat;
        } else {
          vo[q] = gam[q] * rstd[q] * (du - m_a[q] - ohat * m_b[q]);
          vx[q] = du;
        }
      }
      if (MODE == 1) {
        st4(d_o + (size_t)n * D + c, vo);
        if (parts_a) cn_acc4(pa, vo);
        if (dx_add) vx += ld4(dx_add + (size_t)n * D + c);
        st4(dx + (size_t)n * D + c, vx);
      }
    }
    if (MODE == 0) {
      cn_block_store_parts(pa, red, parts_a, D, c, active, wid, lane);
      cn_block_store_parts(pb, red, parts_b, D, c, active, wid, lane);
    } else if (parts_a) {     // MODE 1: the column partials of d_o (the bias gradient of the Linear that produced o)
      cn_block_store_parts(pa, red, parts_a, D, c, active, wid, lane);
    }
  }
}

// Column partial sums of a [R, C] view (bias gradients): parts[block][c], fp64.
__global__ __launch_bounds__(256) void cn_colsum_partial_kernel(const float* __restrict__ x, int ld, int R, int C,
                                                                double* __restrict__ parts) {
  __shared__ double red[NODES_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool active = c < C;
    f64x4 ps = {0, 0, 0, 0};
    for (int r = blockIdx.x * NODES_PER_BLOCK + wid; r < R; r += gridDim.x * NODES_PER_BLOCK)
      if (active) cn_acc4(ps, ld4(x + (size_t)r * ld + c));
    cn_block_store_parts(ps, red, parts, C, c, active, wid, lane);
  }
}

// out = a * sigmoid(b) (softplus backward, cartnet_eltwise op 1) with the fp64 column partials of out in the same pass:
// the bias gradient of the Linear in front of the softplus, which was a second pass over `out`.
__global__ __launch_bounds__(256) void cn_softplus_bwd_sums_kernel(const float* __restrict__ a, int lda,
                                                                   const float* __restrict__ b, int ldb,
                                                                   float* __restrict__ out, int ldo, int R, int C,
                                                                   double* __restrict__ parts) {
  __shared__ double red[NODES_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool active = c < C;
    f64x4 ps = {0, 0, 0, 0};
    for (int r = blockIdx.x * NODES_PER_BLOCK + wid; r < R; r += gridDim.x * NODES_PER_BLOCK)
      if (active) {
        const f32x4 x = ld4(a + (size_t)r * lda + c), p = ld4(b + (size_t)r * ldb + c);
        f32x4 y;
#pragma unroll
        for (int q = 0; q < 4; ++q) y[q] = x[q] * (p[q] > 20.f ? 1.f : cn_sigmoid(p[q]));
        st4(out + (size_t)r * ldo + c, y);
        cn_acc4(ps, y);
      }
    cn_block_store_parts(ps, red, parts, C, c, active, wid, lane);
  }
}

// Column partial sums of d[r, c] * bc[r, c] and d[r, c] * bc[r, C + c] (bc [R, 2C]): the targets' share of the gate's
// BatchNorm-backward sums from the per-target sums the forward gate kernel left (cartnet_gate_scatter_fwd_bc).
__global__ __launch_bounds__(256) void cn_coldot_bc_partial_kernel(const float* __restrict__ d, int ld,
                                                                   const float* __restrict__ bc, int R, int C,
                                                                   double* __restrict__ parts_a, double* __restrict__ parts_b) {
  __shared__ double red[NODES_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool active = c < C;
    f64x4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0};
    for (int r = blockIdx.x * NODES_PER_BLOCK + wid; r < R; r += gridDim.x * NODES_PER_BLOCK)
      if (active) {
        const f32x4 v = ld4(d + (size_t)r * ld + c);
        cn_acc4(pa, v * ld4(bc + (size_t)r * 2 * C + c));
        cn_acc4(pb, v * ld4(bc + (size_t)r * 2 * C + C + c));
      }
    cn_block_store_parts(pa, red, parts_a, C, c, active, wid, lane);
    cn_block_store_parts(pb, red, parts_b, C, c, active, wid, lane);
  }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int cartnet_rbf_expand(const float* v, int64_t n, const float* centers, int32_t bins, float gamma,
                                  float* out, int32_t ldo, void* stream) {
  CN_CHECK(n >= 0 && bins >= 1 && ldo >= bins, "cartnet_rbf_expand: bad sizes");
  if (n == 0) return 0;
  CN_CHECK(v && centers && out, "cartnet_rbf_expand: null pointer");
  const int q = bins / 4;
  if (bins % 4 == 0 && q >= 1 && q <= 256 && 256 % q == 0 && ldo % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15u) == 0 && (reinterpret_cast<uintptr_t>(centers) & 15u) == 0) {
    const int rpb = 256 / q;
    long long blocks4 = (n + rpb - 1) / rpb;
    if (blocks4 > 16384) blocks4 = 16384;
    hipLaunchKernelGGL(cn_rbf_expand4_kernel, dim3((int)blocks4), dim3(256), 0, ST(stream), v, (long long)n, centers, q,
                       gamma, out, ldo);
    CN_LAUNCH_CHECK("cartnet_rbf_expand");
    return 0;
  }
  long long blocks = (n * bins + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cn_rbf_expand_kernel, dim3((int)blocks), dim3(256), 0, ST(stream), v, (long long)n, centers, bins,
                     gamma, out, ldo);
  CN_LAUNCH_CHECK("cartnet_rbf_expand");
  return 0;
}

extern "C" int cartnet_lattice_features(const float* cell, const int64_t* batch, const int32_t* src32,
                                        const float* cart_dist, const float* cart_dir, int64_t E, int32_t Bg,
                                        float* edge_feat, float* nei_len, float* nei_cos, void* stream) {
  CN_CHECK(E >= 0 && Bg >= 1, "cartnet_lattice_features: bad sizes");
  CN_CHECK(cell && nei_len && (E == 0 || (batch && src32 && cart_dist && cart_dir && edge_feat && nei_cos)),
           "cartnet_lattice_features: null pointer");
  long long blocks = ((E > Bg * 3 ? E : Bg * 3) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cn_lattice_features_kernel, dim3((int)blocks), dim3(256), 0, ST(stream), cell, batch, src32,
                     cart_dist, cart_dir, (long long)E, Bg, edge_feat, nei_len, nei_cos);
  CN_LAUNCH_CHECK("cartnet_lattice_features");
  return 0;
}

extern "C" int cartnet_eltwise(int32_t op, const float* a, const float* b, float* out, int64_t rows, int32_t cols,
                               int32_t lda, int32_t ldb, int32_t ldo, float scale, void* stream) {
  CN_CHECK(op >= 0 && op <= 3, "cartnet_eltwise: op=%d", op);
  CN_CHECK(rows >= 0 && cols >= 4 && cols % 4 == 0 && lda % 4 == 0 && ldo % 4 == 0 && lda >= cols && ldo >= cols,
           "cartnet_eltwise: cols / leading dimensions must be multiples of 4");
  if (rows == 0) return 0;
  CN_CHECK(a && out, "cartnet_eltwise: null pointer");
  if (op == 1 || op == 2) CN_CHECK(b && ldb % 4 == 0 && ldb >= cols, "cartnet_eltwise: second operand missing");
  long long blocks = (rows * (cols / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cn_eltwise_kernel, dim3((int)blocks), dim3(256), 0, ST(stream), op, a, b, out, (long long)rows,
                     cols / 4, lda, ldb, ldo, scale);
  CN_LAUNCH_CHECK("cartnet_eltwise");
  return 0;
}

extern "C" int cartnet_rowmul_fwd(const float* key, int32_t ldk, const float* q, int32_t ldq, const int32_t* ptr,
                                  int32_t S, int32_t C, float scale, float* alpha, int32_t lda, double* parts_sum,
                                  double* parts_sq, void* stream) {
  CN_CHECK(S >= 0 && C >= 4 && C % 4 == 0 && ldk % 4 == 0 && ldq % 4 == 0 && lda % 4 == 0,
           "cartnet_rowmul_fwd: C and leading dimensions must be multiples of 4");
  CN_CHECK(key && q && ptr && parts_sum && parts_sq, "cartnet_rowmul_fwd: null pointer");
  hipLaunchKernelGGL(cn_rowmul_fwd_kernel, dim3(seg_parts(S)), dim3(256), 0, ST(stream), key, ldk, q, ldq, ptr, S, C,
                     scale, alpha, lda, parts_sum, parts_sq);
  CN_LAUNCH_CHECK("cartnet_rowmul_fwd");
  return 0;
}

extern "C" int cartnet_rowmul_bwd(float* dalpha, int32_t lda, const float* key, int32_t ldk, const float* q,
                                  int32_t ldq, const int32_t* ptr, int32_t S, int32_t C, float scale, float* dq,
                                  int32_t lddq, void* stream) {
  CN_CHECK(S >= 0 && C >= 4 && C % 4 == 0 && ldk % 4 == 0 && ldq % 4 == 0 && lda % 4 == 0 && lddq % 4 == 0,
           "cartnet_rowmul_bwd: C and leading dimensions must be multiples of 4");
  if (S == 0) return 0;
  CN_CHECK(dalpha && key && q && ptr && dq, "cartnet_rowmul_bwd: null pointer");
  long long blocks = ((long long)S * ((C + 255) / 256) + NODES_PER_BLOCK - 1) / NODES_PER_BLOCK;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cn_rowmul_bwd_kernel<false>, dim3((int)blocks), dim3(256), 0, ST(stream), dalpha, lda, key, ldk, q,
                     ldq, ptr, S, C, scale, dq, lddq, (double*)nullptr, (double*)nullptr);
  CN_LAUNCH_CHECK("cartnet_rowmul_bwd");
  return 0;
}

// The same pass, leaving the fp64 column partials of dkey and dq as well: cartnet_segment_nparts(S) rows of C each
// (-> cartnet_colsum_finalize).  C <= 256.
extern "C" int cartnet_rowmul_bwd_sums(float* dalpha, int32_t lda, const float* key, int32_t ldk, const float* q,
                                       int32_t ldq, const int32_t* ptr, int32_t S, int32_t C, float scale, float* dq,
                                       int32_t lddq, double* parts_dkey, double* parts_dq, void* stream) {
  CN_CHECK(S >= 0 && C >= 4 && C % 4 == 0 && C <= 256 && ldk % 4 == 0 && ldq % 4 == 0 && lda % 4 == 0 && lddq % 4 == 0,
           "cartnet_rowmul_bwd_sums: C <= 256, C and leading dimensions multiples of 4");
  CN_CHECK(parts_dkey && parts_dq && (S == 0 || (dalpha && key && q && ptr && dq)), "cartnet_rowmul_bwd_sums: null pointer");
  hipLaunchKernelGGL(cn_rowmul_bwd_kernel<true>, dim3(seg_parts(S)), dim3(256), 0, ST(stream), dalpha, lda, key, ldk, q,
                     ldq, ptr, S, C, scale, dq, lddq, parts_dkey, parts_dq);
  CN_LAUNCH_CHECK("cartnet_rowmul_bwd_sums");
  return 0;
}

extern "C" int cartnet_att_gate_fwd(const float* gs, const float* q, int32_t ldq, const int32_t* ptr, const float* mean_rstd,
                                    const float* gamma, const float* beta, float scale, int32_t S, int32_t D, float* aggr,
                                    float* bc, void* stream) {
  CN_CHECK(S >= 0 && D >= 4 && D % 4 == 0 && ldq % 4 == 0 && ldq >= D, "cartnet_att_gate_fwd: D / ldq must be multiples of 4");
  if (S == 0) return 0;
  CN_CHECK(gs && q && ptr && mean_rstd && gamma && beta && aggr, "cartnet_att_gate_fwd: null pointer");
  CN_CHECK(!bc || (reinterpret_cast<uintptr_t>(bc) & 15u) == 0, "cartnet_att_gate_fwd: bc must be 16-byte aligned");
  long long blocks = ((long long)S * ((D + 255) / 256) + NODES_PER_BLOCK - 1) / NODES_PER_BLOCK;
  if (blocks > 4096) blocks = 4096;
  if (bc)
    hipLaunchKernelGGL(cn_att_gate_fwd_kernel<true>, dim3((int)blocks), dim3(256), 0, ST(stream), gs, q, ldq, ptr, mean_rstd,
                       gamma, beta, scale, S, D, aggr, bc);
  else
    hipLaunchKernelGGL(cn_att_gate_fwd_kernel<false>, dim3((int)blocks), dim3(256), 0, ST(stream), gs, q, ldq, ptr, mean_rstd,
                       gamma, beta, scale, S, D, aggr, bc);
  CN_LAUNCH_CHECK("cartnet_att_gate_fwd");
  return 0;
}

extern "C" int cartnet_att_gate_bwd_apply(float* gs, const float* key, int32_t ldk, const float* q, int32_t ldq,
                                          const float* daggr, const int32_t* ptr, const float* mean_rstd,
                                          const float* gamma, const float* beta, const float* sums, int64_t count,
                                          int32_t training, float scale, int32_t S, int32_t D, float* dq, int32_t lddq,
                                          double* parts_dkey, double* parts_dmsg, double* parts_dq, void* stream) {
  CN_CHECK(S >= 0 && D >= 4 && D % 4 == 0 && ldk % 4 == 0 && ldq % 4 == 0 && lddq % 4 == 0 && ldk >= D && ldq >= D && lddq >= D,
           "cartnet_att_gate_bwd_apply: D and leading dimensions must be multiples of 4");
  CN_CHECK(mean_rstd && gamma && beta && sums && parts_dkey && parts_dmsg && parts_dq &&
               (S == 0 || (gs && q && daggr && ptr && dq)),
           "cartnet_att_gate_bwd_apply: null pointer");
  const float inv = (training && count > 0) ? (float)(1.0 / (double)count) : 0.f;
  if (key)
    hipLaunchKernelGGL(cn_att_gate_bwd_kernel<false>, dim3(seg_parts(S)), dim3(256), 0, ST(stream), gs, key, ldk, q, ldq, daggr,
                       ptr, mean_rstd, gamma, beta, sums, inv, scale, S, D, dq, lddq, parts_dkey, parts_dmsg, parts_dq);
  else
    hipLaunchKernelGGL(cn_att_gate_bwd_kernel<true>, dim3(seg_parts(S)), dim3(256), 0, ST(stream), gs, key, ldk, q, ldq, daggr,
                       ptr, mean_rstd, gamma, beta, sums, inv, scale, S, D, dq, lddq, parts_dkey, parts_dmsg, parts_dq);
  CN_LAUNCH_CHECK("cartnet_att_gate_bwd_apply");
  return 0;
}

extern "C" int cartnet_softplus_update_fwd(const float* o, const float* x, const float* mean_rstd, const float* gamma,
                                           const float* beta, int64_t N, int32_t D, float* y, void* stream) {
  CN_CHECK(N >= 0 && D >= 4 && D % 4 == 0, "cartnet_softplus_update_fwd: D=%d must be a multiple of 4", D);
  if (N == 0) return 0;
  CN_CHECK(o && x && mean_rstd && gamma && beta && y, "cartnet_softplus_update_fwd: null pointer");
  long long blocks = (N * D / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cn_softplus_update_fwd_kernel, dim3((int)blocks), dim3(256), 0, ST(stream), o, x, mean_rstd, gamma,
                     beta, (long long)N, D, y);
  CN_LAUNCH_CHECK("cartnet_softplus_update_fwd");
  return 0;
}

extern "C" int cartnet_softplus_update_bwd_stats(const float* o, const float* x, const float* dy,
                                                 const float* mean_rstd, const float* gamma, const float* beta,
                                                 int32_t N, int32_t D, double* parts_a, double* parts_b,
                                                 void* stream) {
  CN_CHECK(N >= 0 && D >= 4 && D % 4 == 0, "cartnet_softplus_update_bwd_stats: D=%d must be a multiple of 4", D);
  CN_CHECK(o && x && dy && mean_rstd && gamma && beta && parts_a && parts_b,
           "cartnet_softplus_update_bwd_stats: null pointer");
  hipLaunchKernelGGL(cn_softplus_update_bwd_kernel<0>, dim3(seg_parts(N)), dim3(256), 0, ST(stream), o, x, dy,
                     mean_rstd, gamma, beta, (const float*)nullptr, 0.f, N, D, parts_a, parts_b, (float*)nullptr,
                     (const float*)nullptr, (float*)nullptr);
  CN_LAUNCH_CHECK("cartnet_softplus_update_bwd_stats");
  return 0;
}

static int softplus_update_bwd_apply(const char* who, const float* o, const float* x, const float* dy,
                                     const float* mean_rstd, const float* gamma, const float* beta, const float* sums,
                                     int32_t training, int32_t N, int32_t D, float* d_o, const float* dx_add, float* dx,
                                     double* parts_do, void* stream) {
  CN_CHECK(N >= 0 && D >= 4 && D % 4 == 0, "%s: D=%d must be a multiple of 4", who, D);
  if (N == 0 && !parts_do) return 0;
  CN_CHECK(mean_rstd && gamma && beta && sums && (N == 0 || (o && x && dy && d_o && dx)), "%s: null pointer", who);
  const float inv = (training && N > 0) ? (float)(1.0 / (double)N) : 0.f;
  hipLaunchKernelGGL(cn_softplus_update_bwd_kernel<1>, dim3(seg_parts(N)), dim3(256), 0, ST(stream), o, x, dy,
                     mean_rstd, gamma, beta, sums, inv, N, D, parts_do, (double*)nullptr, d_o, dx_add, dx);
  CN_LAUNCH_CHECK(who);
  return 0;
}

extern "C" int cartnet_softplus_update_bwd_apply(const float* o, const float* x, const float* dy,
                                                 const float* mean_rstd, const float* gamma, const float* beta,
                                                 const float* sums, int32_t training, int32_t N, int32_t D, float* d_o,
                                                 const float* dx_add, float* dx, void* stream) {
  return softplus_update_bwd_apply("cartnet_softplus_update_bwd_apply", o, x, dy, mean_rstd, gamma, beta, sums, training, N, D,
                                   d_o, dx_add, dx, nullptr, stream);
}

// ... leaving the fp64 column partials of d_o too (cartnet_segment_nparts(N) rows of D -> cartnet_colsum_finalize)
extern "C" int cartnet_softplus_update_bwd_apply_sums(const float* o, const float* x, const float* dy,
                                                      const float* mean_rstd, const float* gamma, const float* beta,
                                                      const float* sums, int32_t training, int32_t N, int32_t D,
                                                      float* d_o, const float* dx_add, float* dx, double* parts_do,
                                                      void* stream) {
  CN_CHECK(parts_do, "cartnet_softplus_update_bwd_apply_sums: null pointer");
  return softplus_update_bwd_apply("cartnet_softplus_update_bwd_apply_sums", o, x, dy, mean_rstd, gamma, beta, sums, training,
                                   N, D, d_o, dx_add, dx, parts_do, stream);
}

// out = a * sigmoid(b) over [R, C] views + the fp64 column partials of out (cartnet_segment_nparts(R) rows of C)
extern "C" int cartnet_softplus_bwd_sums(const float* a, int32_t lda, const float* b, int32_t ldb, float* out, int32_t ldo,
                                         int32_t R, int32_t C, double* parts, void* stream) {
  CN_CHECK(R >= 0 && C >= 4 && C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0 && lda >= C && ldb >= C && ldo >= C,
           "cartnet_softplus_bwd_sums: C and leading dimensions must be multiples of 4");
  CN_CHECK(parts && (R == 0 || (a && b && out)), "cartnet_softplus_bwd_sums: null pointer");
  hipLaunchKernelGGL(cn_softplus_bwd_sums_kernel, dim3(seg_parts(R)), dim3(256), 0, ST(stream), a, lda, b, ldb, out, ldo, R, C,
                     parts);
  CN_LAUNCH_CHECK("cartnet_softplus_bwd_sums");
  return 0;
}

extern "C" int cartnet_segment_nparts(int32_t S) { return seg_parts(S); }

extern "C" int cartnet_coldot_bc_partial(const float* d, int32_t ld, const float* bc, int32_t R, int32_t C, double* parts_a,
                                         double* parts_b, void* stream) {
  CN_CHECK(R >= 0 && C >= 4 && C % 4 == 0 && ld % 4 == 0 && ld >= C, "cartnet_coldot_bc_partial: C/ld must be multiples of 4");
  CN_CHECK(((d && bc) || R == 0) && parts_a && parts_b, "cartnet_coldot_bc_partial: null pointer");
  hipLaunchKernelGGL(cn_coldot_bc_partial_kernel, dim3(seg_parts(R)), dim3(256), 0, ST(stream), d, ld, bc, R, C, parts_a,
                     parts_b);
  CN_LAUNCH_CHECK("cartnet_coldot_bc_partial");
  return 0;
}

extern "C" int cartnet_colsum_partial(const float* x, int32_t ld, int32_t R, int32_t C, double* parts, void* stream) {
  CN_CHECK(R >= 0 && C >= 4 && C % 4 == 0 && ld % 4 == 0 && ld >= C, "cartnet_colsum_partial: C/ld must be multiples of 4");
  CN_CHECK((x || R == 0) && parts, "cartnet_colsum_partial: null pointer");
  hipLaunchKernelGGL(cn_colsum_partial_kernel, dim3(seg_parts(R)), dim3(256), 0, ST(stream), x, ld, R, C, parts);
  CN_LAUNCH_CHECK("cartnet_colsum_partial");
  return 0;
}
