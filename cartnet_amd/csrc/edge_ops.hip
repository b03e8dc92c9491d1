// Per-edge kernels of the CartNet layer (HBM-bound): Cartesian edge features, the neighbour-equalised gate with
// its CSR segmented reduction (forward and backward) and row-segment sums.
//
// Layout: edge rows of width D (fp32) are walked one row per wavefront-instruction: lane l owns channels
// 4l..4l+3 of a 256-channel chunk, so a D=256 row is one coalesced 1 KiB dwordx4 access.  One wavefront owns one
// target node and visits its edges in edge order (edges are sorted by target), so sums reproduce the sequential
// CPU scatter_add_ order and need no atomics.
#include "common.h"
#include <math.h>

namespace {

constexpr int NODES_PER_BLOCK = 4;   // one wave per node
constexpr int MAX_PARTS = 1024;
#ifndef CN_GATE_BATCH
#define CN_GATE_BATCH 8
#endif
constexpr int GATE_BATCH = CN_GATE_BATCH;     // edges whose row loads are in flight together per wave (8: 5.28 TB/s forward, 4: 4.98)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Rows kept in memory as bf16 ("half storage", the *_h entry points): `base` then points at bf16
// elements, `idx` counts elements either way; the arithmetic stays fp32.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <bool H>
__device__ __forceinline__ f32x4 ldrow(const float* base, size_t idx) {
  if constexpr (H) return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(base) + idx), f32x4);
  else return ld4(base + idx);
}
template <bool H>
__device__ __forceinline__ void strow(float* base, size_t idx, f32x4 v) {
  if constexpr (H) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + idx) = __builtin_convertvector(v, bf16x4);
  else st4(base + idx, v);
}

// One workgroup per FEAT_EDGES edges: the per-edge scalars (cutoff, exp(-alpha d), envelope) are computed once by
// the first FEAT_EDGES threads and shared through LDS, so an output element costs one exp instead of two and a cos.
constexpr int FEAT_EDGES = 64;

__global__ __launch_bounds__(256) void cn_edge_features_kernel(
    const float* __restrict__ dist, const float* __restrict__ dir, const float* __restrict__ means,
    const float* __restrict__ betas, long long E, int R, int invariant, float radius, float env_radius,
    float* __restrict__ feat, int ldf, float* __restrict__ env) {
  __shared__ float s_cut[FEAT_EDGES], s_ex[FEAT_EDGES];
  const float alpha = 5.0f / radius;
  const float kPi = 3.14159265358979323846f;
  for (long long e0 = (long long)blockIdx.x * FEAT_EDGES; e0 < E; e0 += (long long)gridDim.x * FEAT_EDGES) {
    const int ne = (int)min((long long)FEAT_EDGES, E - e0);
    if ((int)threadIdx.x < ne) {
      const float d = dist[e0 + threadIdx.x];
      s_cut[threadIdx.x] = (d < radius) ? 0.5f * (cosf(d * kPi / radius) + 1.0f) : 0.f;
      s_ex[threadIdx.x] = expf(alpha * (-d));
      if (env) env[e0 + threadIdx.x] = (d < env_radius) ? 0.5f * (cosf(d * kPi / env_radius) + 1.0f) : 0.f;
    }
    __syncthreads();
    const int total = ne * ldf;
    for (int i = threadIdx.x; i < total; i += 256) {
      const int el = i / ldf, c = i - el * ldf;
      float v = 0.f;
      if (c < R) {
        const float t = s_ex[el] - means[c];
        v = s_cut[el] * expf(-betas[c] * t * t);
      } else if (!invariant && c < R + 3) {
        v = dir[(e0 + el) * 3 + (c - R)];
      }
      feat[e0 * ldf + i] = v;
    }
    __syncthreads();
  }
}

// Sweep direction.  These kernels stream several hundred MB that the previous kernel has just written or read in
// ascending row order; the 256 MB memory-side cache (MALL) then still holds the highest rows.  Walking the nodes in
// descending sweeps ("reverse") finds them there instead of in HBM: the forward gate runs against the GEMM that wrote
// gs, the backward apply pass against the statistics pass, the by-source segment sum against the by-target one
// (measured at the benchmark shape: backward pair 331 -> 288 us, segment-sum pair 150 -> 129 us).  Results do not
// depend on the direction except for the order in which a workgroup adds its nodes into the fp64 partial sums.
// ------------------------------------------------------------------------------------------------ gate forward
template <bool GH>
__global__ __launch_bounds__(256) void cn_gate_scatter_fwd_kernel(
    const float* __restrict__ gs, const float* __restrict__ e_in, const float* __restrict__ env,
    const int* __restrict__ rowptr, const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, int N, int D, float* __restrict__ e_out, float* __restrict__ aggr,
    double* __restrict__ parts_sum, double* __restrict__ parts_sq, int reverse, const int* __restrict__ node_gptr,
    float* __restrict__ bc) {
  __shared__ double red[NODES_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int ld = 2 * D;
  int gi, bx, n0, n1;
  cn_group_range(node_gptr, N, reverse != 0, gi, bx, n0, n1);
  mean_rstd += (size_t)gi * 2 * D;
  const int prow = gi * gridDim.x + bx;
  const int stride = gridDim.x * NODES_PER_BLOCK, nsweeps = (n1 - n0 + stride - 1) / stride;
  for (int c0 = 0; c0 < D; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool active = c < D;
    f32x4 mean = {0, 0, 0, 0}, scale = {0, 0, 0, 0}, shift = {0, 0, 0, 0}, rstd = {0, 0, 0, 0};
    if (active) {
      mean = ld4(mean_rstd + c);
      rstd = ld4(mean_rstd + D + c);
      scale = rstd * ld4(gamma + c);
      shift = ld4(beta + c);
    }
    f64x4 ps = {0, 0, 0, 0}, pq = {0, 0, 0, 0};
    for (int j = 0; j < nsweeps; ++j) {
      const int t = n0 + bx * NODES_PER_BLOCK + wid + (reverse ? nsweeps - 1 - j : j) * stride;
      if (t >= n1) continue;
      const int k0 = rowptr[t], k1 = rowptr[t + 1];
      f32x4 acc = {0, 0, 0, 0}, accb = {0, 0, 0, 0}, accc = {0, 0, 0, 0};
      if (active) {
        // GATE_BATCH edges per round: all row loads of a round are issued before the first use (a round past the end
        // of the segment repeats the last edge and drops it)
        for (int k = k0; k < k1; k += GATE_BATCH) {
          f32x4 g[GATE_BATCH], sv[GATE_BATCH], ei[GATE_BATCH];
          float ev[GATE_BATCH];
#pragma unroll
          for (int u = 0; u < GATE_BATCH; ++u) {
            const int kk = min(k + u, k1 - 1);
#ifndef CN_NO_GATE_LOAD_NT      /* last use of gs before backward: non-temporal (round 5: -0.03...-0.08 ms per step, same-box A B C x 3) */
            if constexpr (!GH) {
              g[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gs + (size_t)kk * ld + c));
              sv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gs + (size_t)kk * ld + D + c));
            } else
#endif
            {
              g[u] = ldrow<GH>(gs, (size_t)kk * ld + c);
              sv[u] = ldrow<GH>(gs, (size_t)kk * ld + D + c);
            }
            ei[u] = e_out ? ld4(e_in + (size_t)kk * D + c) : f32x4{0, 0, 0, 0};     // (non-temporal here too: +-0)
            ev[u] = env ? env[kk] : 1.0f;
          }
#pragma unroll
          for (int u = 0; u < GATE_BATCH; ++u) {
            if (k + u >= k1) break;
            f32x4 sig;
            if (bc) {
              // for the backward pass (cartnet_node_update_bwd_apply_bc): per target, B = sum_e s w and C = sum_e s w ghat with
              // w = d sigma / d bn = env z (1 - z) -- the daggr share of the BatchNorm-backward sums is then a sum over ATOMS
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float z = cn_sigmoid((g[u][q] - mean[q]) * scale[q] + shift[q]);
                sig[q] = ev[u] * z;
                const float sw = sv[u][q] * (sig[q] * (1.0f - z));
                accb[q] += sw;
                accc[q] += sw * ((g[u][q] - mean[q]) * rstd[q]);
              }
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) sig[q] = ev[u] * cn_sigmoid((g[u][q] - mean[q]) * scale[q] + shift[q]);
            }
            if (e_out) st4(e_out + (size_t)(k + u) * D + c, ei[u] + sig);
            acc += sig * sv[u];
          }
        }
        st4(aggr + (size_t)t * D + c, acc);
        if (bc) {
          st4(bc + (size_t)t * ld + c, accb);
          st4(bc + (size_t)t * ld + D + c, accc);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ps[q] += (double)acc[q];
        pq[q] += (double)acc[q] * (double)acc[q];
      }
    }
    cn_block_store_parts_row(ps, red, parts_sum, D, c, active, wid, lane, prow);
    cn_block_store_parts_row(pq, red, parts_sq, D, c, active, wid, lane, prow);
  }
}

// ------------------------------------------------------------------------------------------------ gate backward
// MODE 0: statistics (sum dbn, sum dbn*ghat).  MODE 1: apply in place (g <- dg, s <- ds) + sums of dg, ds.
template <int MODE, bool GH>
__global__ __launch_bounds__(256) void cn_gate_scatter_bwd_kernel(
    float* gs, const float* __restrict__ de_out, const float* __restrict__ daggr, const float* __restrict__ env,
    const int* __restrict__ rowptr, const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ sums, float inv_count, int N, int D,
    double* __restrict__ parts_a, double* __restrict__ parts_b, int reverse, const int* __restrict__ node_gptr) {
  __shared__ double red[NODES_PER_BLOCK * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int ld = 2 * D;
  int gi, bx, n0, n1;
  cn_group_range(node_gptr, N, reverse != 0, gi, bx, n0, n1);
  mean_rstd += (size_t)gi * 2 * D;
  if (MODE == 1) sums += (size_t)gi * 2 * D;
  if (MODE == 1 && node_gptr && inv_count != 0.f) {   // training-mode BatchNorm backward: means over THIS group's edges
    const int eg = rowptr[n1] - rowptr[n0];
    inv_count = eg > 0 ? 1.0f / (float)eg : 0.f;
  }
  const int prow = gi * gridDim.x + bx;
  const int stride = gridDim.x * NODES_PER_BLOCK, nsweeps = (n1 - n0 + stride - 1) / stride;
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
      if (MODE == 1) {
        m_a = ld4(sums + c) * inv_count;
        m_b = ld4(sums + D + c) * inv_count;
      }
    }
    f64x4 ta = {0, 0, 0, 0}, tb = {0, 0, 0, 0};
    for (int j = 0; j < nsweeps; ++j) {
      const int t = n0 + bx * NODES_PER_BLOCK + wid + (reverse ? nsweeps - 1 - j : j) * stride;
      if (t >= n1 || !active) continue;
      const int k0 = rowptr[t], k1 = rowptr[t + 1];
      const f32x4 dm = ld4(daggr + (size_t)t * D + c);
      f32x4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0};   // fp32 over one node's edges, fp64 across nodes
#pragma unroll 2
      for (int k = k0; k < k1; ++k) {
        const f32x4 g = ldrow<GH>(gs, (size_t)k * ld + c);
        const f32x4 s = ldrow<GH>(gs, (size_t)k * ld + D + c);
        f32x4 de = {0, 0, 0, 0};
        if (de_out) de = ld4(de_out + (size_t)k * D + c);
        const float ev = env ? env[k] : 1.0f;
        f32x4 dgv, dsv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float ghat = (g[q] - mean[q]) * rstd[q];
          const float z = cn_sigmoid(ghat * gam[q] + shift[q]);
          const float dsig = dm[q] * s[q] + de[q];
          const float dbn = dsig * ev * z * (1.0f - z);
          if (MODE == 0) {
            pa[q] += dbn;
            pb[q] += dbn * ghat;
          } else {
            dgv[q] = gam[q] * rstd[q] * (dbn - m_a[q] - ghat * m_b[q]);
            dsv[q] = dm[q] * ev * z;
            pa[q] += dgv[q];
            pb[q] += dsv[q];
          }
        }
        if (MODE == 1) {
          strow<GH>(gs, (size_t)k * ld + c, dgv);
          strow<GH>(gs, (size_t)k * ld + D + c, dsv);
        }
      }
      cn_acc4(ta, pa);
      cn_acc4(tb, pb);
    }
    cn_block_store_parts_row(ta, red, parts_a, D, c, active, wid, lane, prow);
    cn_block_store_parts_row(tb, red, parts_b, D, c, active, wid, lane, prow);
  }
}

// ------------------------------------------------------------------------------------------------ segment sums
// One wave per (segment, 256-column slab).  The row loads of a batch are independent (SEG_BATCH x 1 KiB in flight per
// wave; a dependent load-add chain left the kernel latency-bound at 2.5 TB/s); the adds keep position order.  Segment
// bounds and permutation entries are wave-uniform, so they travel through the scalar unit.
#ifndef CN_SEG_BATCH
#define CN_SEG_BATCH 8
#endif
constexpr int SEG_BATCH = CN_SEG_BATCH;

template <bool RH>
__global__ __launch_bounds__(256) void cn_segment_sum_kernel(const float* __restrict__ rows, int ld,
                                                             const int* __restrict__ ptr,
                                                             const int* __restrict__ perm, int N, int W,
                                                             float* __restrict__ out, int ldo, int reverse) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chunks = (W + 255) / 256;
  const long long items = (long long)N * chunks;
  for (long long it0 = (long long)blockIdx.x * NODES_PER_BLOCK + wid; it0 < items;
       it0 += (long long)gridDim.x * NODES_PER_BLOCK) {
    const long long it = reverse ? items - 1 - it0 : it0;
    const int t = (int)(it / chunks);
    const int c = (int)(it % chunks) * 256 + lane * 4;
    if (c >= W) continue;
    const int k0 = ptr[t], k1 = ptr[t + 1];
    f32x4 acc = {0, 0, 0, 0};
    int k = k0;
    for (; k + SEG_BATCH <= k1; k += SEG_BATCH) {
      f32x4 v[SEG_BATCH];
#pragma unroll
      for (int u = 0; u < SEG_BATCH; ++u) v[u] = ldrow<RH>(rows, (size_t)(perm ? perm[k + u] : k + u) * ld + c);
#pragma unroll
      for (int u = 0; u < SEG_BATCH; ++u) acc += v[u];
    }
    if (k < k1) {   // remainder: clamped (repeated) loads, the repeats are dropped before the adds
      f32x4 v[SEG_BATCH];
#pragma unroll
      for (int u = 0; u < SEG_BATCH - 1; ++u) {
        const int kk = min(k + u, k1 - 1);
        v[u] = ldrow<RH>(rows, (size_t)(perm ? perm[kk] : kk) * ld + c);
      }
#pragma unroll
      for (int u = 0; u < SEG_BATCH - 1; ++u)
        if (k + u < k1) acc += v[u];
    }
    st4(out + (size_t)t * ldo + c, acc);
  }
}

// Both sums of a layer's backward in ONE launch: out_t[n] = sum of the rows of target n (contiguous, CSR), out_s[n] = sum
// of the rows of source n (through the CSC permutation).  The two walk the same rows -- an edge row is read once as its
// target's and once as its source's, and both atoms belong to the same crystal -- so the items are dealt out interleaved
// (node n by target, node n by source, node n + 1 by target, ...): the second reader of a row runs microseconds after the
// first and finds it in the L2 / the memory-side cache instead of HBM (two separate launches stream the 363 MB of a layer's
// dpre twice; the descending sweep of the second only saved what the 256 MB cache still held).
template <bool RH>
__global__ __launch_bounds__(256) void cn_segment_sum_pair_kernel(const float* __restrict__ rows, int ld,
                                                                  const int* __restrict__ rowptr,
                                                                  const int* __restrict__ colptr,
                                                                  const int* __restrict__ perm, int N, int W,
                                                                  float* __restrict__ out_t, float* __restrict__ out_s,
                                                                  int ldo, int ochunk, int xcd_blocks) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chunks = (W + 255) / 256;
  const long long items = 2LL * N * chunks;
  // xcd_blocks > 0 (the grid is then a multiple of 8 * xcd_blocks and covers every item once): blocks b, b + 8, ... share an
  // XCD, so XCD x takes the runs x, x + 8, ... of xcd_blocks consecutive blocks' worth of atoms (the host: one run of N / 8
  // atoms per XCD) in atom order: the second reader of an edge row (once by its target, once by its source: both atoms of one
  // crystal) then runs on the SAME XCD within ~140 atoms of the first and finds the row in that XCD's L2 instead of fetching
  // it again
  long long bfirst = blockIdx.x;
  if (xcd_blocks > 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    bfirst = ((long long)(j / xcd_blocks) * 8 + xcd) * xcd_blocks + j % xcd_blocks;
  }
  for (long long it = bfirst * NODES_PER_BLOCK + wid; it < items; it += (long long)gridDim.x * NODES_PER_BLOCK) {
    const int per_node = 2 * chunks;
    const int t = (int)(it / per_node);
    const int sub = (int)(it % per_node);
    const bool by_src = sub >= chunks;
    const int chunk = by_src ? sub - chunks : sub;
    const int c = chunk * 256 + lane * 4;
    if (c >= W) continue;
    const int* __restrict__ ptr = by_src ? colptr : rowptr;
    const int k0 = ptr[t], k1 = ptr[t + 1];
    f32x4 acc = {0, 0, 0, 0};
    int k = k0;
    for (; k + SEG_BATCH <= k1; k += SEG_BATCH) {
      f32x4 v[SEG_BATCH];
#pragma unroll
      for (int u = 0; u < SEG_BATCH; ++u) v[u] = ldrow<RH>(rows, (size_t)(by_src ? perm[k + u] : k + u) * ld + c);
#pragma unroll
      for (int u = 0; u < SEG_BATCH; ++u) acc += v[u];
    }
    if (k < k1) {
      f32x4 v[SEG_BATCH];
#pragma unroll
      for (int u = 0; u < SEG_BATCH - 1; ++u) {
        const int kk = min(k + u, k1 - 1);
        v[u] = ldrow<RH>(rows, (size_t)(by_src ? perm[kk] : kk) * ld + c);
      }
#pragma unroll
      for (int u = 0; u < SEG_BATCH - 1; ++u)
        if (k + u < k1) acc += v[u];
    }
    st4((by_src ? out_s : out_t) + (size_t)t * ldo + (size_t)chunk * ochunk + lane * 4, acc);
  }
}

// Long-segment variant (few, very uneven segments, e.g. atoms grouped by element): pass 1 cuts the sorted positions
// into chunks of LONG_CHUNK rows, one wave per (chunk, 256-column slab), and writes one partial row per run of equal
// segment id at tmp[first position of the run]; pass 2 adds each segment's partial rows in position order.
constexpr int LONG_CHUNK = 32;
constexpr int LONG_BATCH = 16;   // independent row loads in flight per wave

// COMPACT: partial rows are numbered instead of sitting at their position -- a run that starts a chunk is row
// position / LONG_CHUNK, any other run starts its segment and is row nchunks + segment: tmp needs nchunks + nseg rows
// instead of `total` (cartnet_segment_sum_chunked: per-crystal sums over all edges, where `total` rows would be 1 GB).
template <bool COMPACT>
__global__ __launch_bounds__(256) void cn_segment_long_pass1_kernel(const float* __restrict__ rows, int ld,
                                                                    const int* __restrict__ ptr,
                                                                    const int* __restrict__ perm, int nseg, int total,
                                                                    int W, float* __restrict__ tmp) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int slabs = (W + 255) / 256;
  const int nchunks = (total + LONG_CHUNK - 1) / LONG_CHUNK;
  for (long long it = (long long)blockIdx.x * NODES_PER_BLOCK + wid; it < (long long)nchunks * slabs;
       it += (long long)gridDim.x * NODES_PER_BLOCK) {
    const int chunk = (int)(it / slabs);
    const int c = (int)(it % slabs) * 256 + lane * 4;
    const int p0 = chunk * LONG_CHUNK, p1 = min(total, p0 + LONG_CHUNK);
    // Every memory round trip of this wave is a latency nobody hides when the batch is small (736 atoms: 46 waves on the
    // whole chip), so: the chunk's row numbers in ONE load (lane i holds position p0 + i), and the segment of position p0
    // = (number of s with ptr[s] <= p0) - 1, counted by the lanes in one or two loads instead of a binary search's seven.
    const int my_row = (lane < LONG_CHUNK && p0 + lane < p1) ? (perm ? perm[p0 + lane] : p0 + lane) : 0;
    int below = 0;
    for (int s0 = 0; s0 < nseg; s0 += 64) below += (s0 + lane < nseg && ptr[s0 + lane] <= p0) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
    int seg = below > 0 ? below - 1 : 0;
    int next = ptr[seg + 1];         // first position of the next segment (ptr has nseg + 1 entries)
    int run_start = p0;
    auto slot = [&](int start, int seg_of_run) -> size_t {
      if (!COMPACT) return (size_t)start;
      return (size_t)(start % LONG_CHUNK == 0 ? start / LONG_CHUNK : nchunks + seg_of_run);
    };
    f32x4 acc = {0, 0, 0, 0};
    for (int pb = p0; pb < p1; pb += LONG_BATCH) {
      f32x4 v[LONG_BATCH];
#pragma unroll
      for (int u = 0; u < LONG_BATCH; ++u) {   // the loads of a batch are independent; the adds below keep position order
        const int p = pb + u;
        const int r = __shfl(my_row, (p - p0) & 63);
        v[u] = f32x4{0, 0, 0, 0};
        if (p < p1 && c < W) v[u] = ld4(rows + (size_t)r * ld + c);
      }
#pragma unroll
      for (int u = 0; u < LONG_BATCH; ++u) {
        const int p = pb + u;
        if (p < p1) {
          while (seg + 1 <= nseg && p >= next) {   // p starts a new segment: flush the finished run
            if (p > run_start && c < W) st4(tmp + slot(run_start, seg) * W + c, acc);
            acc = f32x4{0, 0, 0, 0};
            run_start = p;
            ++seg;
            next = seg + 1 <= nseg ? ptr[seg + 1] : 0x7fffffff;
          }
          acc += v[u];
        }
      }
    }
    if (p1 > run_start && c < W) st4(tmp + slot(run_start, seg) * W + c, acc);
  }
}

// The COMPACT pass 1 for rows that are FOLD pieces of W / FOLD columns side by side (iComformer's edge layer: [E, 3, 2C]
// seen as [E, 6C]), with the pieces' sum per row as a by-product: fold_out[p, c] = sum_i rows[p, i * W/FOLD + c].  A wave owns
// a 32-row chunk and one 256-column slab of the FOLDED width and reads its FOLD pieces of every row together, so each row
// is read once for both results (the per-crystal sums over all edges and the per-edge sum over the lattice vectors were
// two passes over 1.09 GB at the benchmark batch).
template <int FOLD>
__global__ __launch_bounds__(256) void cn_segment_long_fold_pass1_kernel(const float* __restrict__ rows, int ld,
                                                                         const int* __restrict__ ptr, int nseg, int total,
                                                                         int W, float* __restrict__ tmp,
                                                                         float* __restrict__ fold_out) {
  constexpr int FB = 8;                        // rows whose FOLD x 1 KiB loads are in flight together
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int Wf = W / FOLD;
  const int slabs = (Wf + 255) / 256;
  const int nchunks = (total + LONG_CHUNK - 1) / LONG_CHUNK;
  for (long long it = (long long)blockIdx.x * NODES_PER_BLOCK + wid; it < (long long)nchunks * slabs;
       it += (long long)gridDim.x * NODES_PER_BLOCK) {
    const int chunk = (int)(it / slabs);
    const int c = (int)(it % slabs) * 256 + lane * 4;
    const bool on = c < Wf;
    const int p0 = chunk * LONG_CHUNK, p1 = min(total, p0 + LONG_CHUNK);
    int below = 0;
    for (int s0 = 0; s0 < nseg; s0 += 64) below += (s0 + lane < nseg && ptr[s0 + lane] <= p0) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
    int seg = below > 0 ? below - 1 : 0;
    int next = ptr[seg + 1];
    int run_start = p0;
    auto slot = [&](int start, int seg_of_run) -> size_t {
      return (size_t)(start % LONG_CHUNK == 0 ? start / LONG_CHUNK : nchunks + seg_of_run);
    };
    f32x4 acc[FOLD];
#pragma unroll
    for (int i = 0; i < FOLD; ++i) acc[i] = f32x4{0, 0, 0, 0};
    auto flush = [&](int start, int seg_of_run) {
      if (!on) return;
#pragma unroll
      for (int i = 0; i < FOLD; ++i) st4(tmp + slot(start, seg_of_run) * W + i * Wf + c, acc[i]);
    };
    for (int pb = p0; pb < p1; pb += FB) {
      f32x4 v[FB][FOLD];
#pragma unroll
      for (int u = 0; u < FB; ++u)
#pragma unroll
        for (int i = 0; i < FOLD; ++i) {
          v[u][i] = f32x4{0, 0, 0, 0};
          if (pb + u < p1 && on) v[u][i] = ld4(rows + (size_t)(pb + u) * ld + i * Wf + c);
        }
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int p = pb + u;
        if (p < p1) {
          while (seg + 1 <= nseg && p >= next) {
            if (p > run_start) flush(run_start, seg);
#pragma unroll
            for (int i = 0; i < FOLD; ++i) acc[i] = f32x4{0, 0, 0, 0};
            run_start = p;
            ++seg;
            next = seg + 1 <= nseg ? ptr[seg + 1] : 0x7fffffff;
          }
          f32x4 s = v[u][0];
#pragma unroll
          for (int i = 1; i < FOLD; ++i) s += v[u][i];          // ((piece 0 + piece 1) + piece 2): the order of cn_icf_sum3
#pragma unroll
          for (int i = 0; i < FOLD; ++i) acc[i] += v[u][i];
          if (on) st4(fold_out + (size_t)p * Wf + c, s);
        }
      }
    }
    if (p1 > run_start) flush(run_start, seg);
  }
}

template <bool COMPACT>
__global__ __launch_bounds__(256) void cn_segment_long_pass2_kernel(const float* __restrict__ tmp,
                                                                    const int* __restrict__ ptr, int nseg, int W,
                                                                    float* __restrict__ out, int ldo, int nchunks) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int slabs = (W + 255) / 256;
  for (long long it = (long long)blockIdx.x * NODES_PER_BLOCK + wid; it < (long long)nseg * slabs;
       it += (long long)gridDim.x * NODES_PER_BLOCK) {
    const int s = (int)(it / slabs);
    const int c = (int)(it % slabs) * 256 + lane * 4;
    if (c >= W) continue;
    const int b = ptr[s], e = ptr[s + 1];
    f32x4 acc = {0, 0, 0, 0};
    int p = b;
    while (p < e) {    // partial rows sit at b and at every chunk boundary inside (b, e); batches of independent loads
      int pos[LONG_BATCH];
      f32x4 v[LONG_BATCH];
#pragma unroll
      for (int u = 0; u < LONG_BATCH; ++u) {
        pos[u] = p;
        if (p < e) {
          const size_t row = !COMPACT ? (size_t)p : (size_t)(p % LONG_CHUNK == 0 ? p / LONG_CHUNK : nchunks + s);
          v[u] = ld4(tmp + row * W + c);
          p = (p / LONG_CHUNK + 1) * LONG_CHUNK;
        } else {
          v[u] = f32x4{0, 0, 0, 0};
        }
      }
#pragma unroll
      for (int u = 0; u < LONG_BATCH; ++u)
        if (pos[u] < e) acc += v[u];
    }
    st4(out + (size_t)s * ldo + c, acc);
  }
}

inline int gate_parts(int N) {
  int b = cn_ceil_div(N, NODES_PER_BLOCK);
  if (b > MAX_PARTS) b = MAX_PARTS;
  if (b < 1) b = 1;
  return b;
}

}  // namespace

extern "C" int cartnet_edge_features(const float* cart_dist, const float* cart_dir, const float* means,
                                     const float* betas, int64_t E, int32_t R, int32_t invariant, float radius,
                                     float env_radius, float* feat, int32_t ldf, float* env, void* stream) {
  CN_CHECK(E >= 0 && R >= 1, "cartnet_edge_features: bad sizes");
  CN_CHECK(ldf >= R + (invariant ? 0 : 3), "cartnet_edge_features: ldf=%d too small", ldf);
  CN_CHECK(radius > 0.f && env_radius > 0.f, "cartnet_edge_features: radius must be positive");
  if (E == 0) return 0;
  CN_CHECK(cart_dist && means && betas && feat && (invariant || cart_dir), "cartnet_edge_features: null pointer");
  long long blocks = (E + FEAT_EDGES - 1) / FEAT_EDGES;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cn_edge_features_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     cart_dist, cart_dir, means, betas, (long long)E, R, invariant, radius, env_radius, feat, ldf,
                     env);
  CN_LAUNCH_CHECK("cartnet_edge_features");
  return 0;
}

extern "C" int cartnet_gate_scatter_nparts(int32_t N) { return gate_parts(N); }

static int gate_scatter_fwd_impl(bool half, const float* gs, const float* e_in, const float* env, const int32_t* rowptr,
                                 const float* mean_rstd, const float* gamma, const float* beta, int32_t N,
                                 int32_t D, float* e_out, float* aggr, double* parts_sum, double* parts_sq,
                                 const CartnetGroups* groups, void* stream, float* bc = nullptr) {
  CN_CHECK(N >= 0 && D >= 4 && D % 4 == 0, "cartnet_gate_scatter_fwd: D=%d must be a positive multiple of 4", D);
  CN_CHECK(gs && rowptr && mean_rstd && gamma && beta && aggr && parts_sum && parts_sq,
           "cartnet_gate_scatter_fwd: null pointer");
  CN_CHECK((e_in == nullptr) == (e_out == nullptr), "cartnet_gate_scatter_fwd: e_in and e_out must pair");
  CN_CHECK(cn_groups_ok(groups), "cartnet_gate_scatter_fwd: bad groups");
  if (half)
    hipLaunchKernelGGL(cn_gate_scatter_fwd_kernel<true>, cn_group_grid(groups, gate_parts(N), true), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), gs, e_in, env, rowptr, mean_rstd, gamma, beta, N, D,
                       e_out, aggr, parts_sum, parts_sq, /*reverse=*/1, groups ? groups->node_gptr : nullptr, bc);
  else
    hipLaunchKernelGGL(cn_gate_scatter_fwd_kernel<false>, cn_group_grid(groups, gate_parts(N), true), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), gs, e_in, env, rowptr, mean_rstd, gamma, beta, N, D,
                       e_out, aggr, parts_sum, parts_sq, /*reverse=*/1, groups ? groups->node_gptr : nullptr, bc);
  CN_LAUNCH_CHECK("cartnet_gate_scatter_fwd");
  return 0;
}

extern "C" int cartnet_gate_scatter_fwd(const float* gs, const float* e_in, const float* env, const int32_t* rowptr,
                                        const float* mean_rstd, const float* gamma, const float* beta, int32_t N,
                                        int32_t D, float* e_out, float* aggr, double* parts_sum, double* parts_sq,
                                        const CartnetGroups* groups, void* stream) {
  return gate_scatter_fwd_impl(false, gs, e_in, env, rowptr, mean_rstd, gamma, beta, N, D, e_out, aggr, parts_sum, parts_sq,
                               groups, stream);
}
extern "C" int cartnet_gate_scatter_fwd_bc(const float* gs, const float* e_in, const float* env, const int32_t* rowptr,
                                           const float* mean_rstd, const float* gamma, const float* beta, int32_t N,
                                           int32_t D, float* e_out, float* aggr, double* parts_sum, double* parts_sq,
                                           float* bc, void* stream) {
  CN_CHECK(bc != nullptr && (reinterpret_cast<uintptr_t>(bc) & 15u) == 0, "cartnet_gate_scatter_fwd_bc: bc must be a 16-byte aligned [N, 2D] buffer");
  return gate_scatter_fwd_impl(false, gs, e_in, env, rowptr, mean_rstd, gamma, beta, N, D, e_out, aggr, parts_sum, parts_sq,
                               nullptr, stream, bc);
}
extern "C" int cartnet_gate_scatter_fwd_h(const void* gs_bf16, const float* e_in, const float* env, const int32_t* rowptr,
                                          const float* mean_rstd, const float* gamma, const float* beta, int32_t N,
                                          int32_t D, float* e_out, float* aggr, double* parts_sum, double* parts_sq,
                                          const CartnetGroups* groups, void* stream) {
  return gate_scatter_fwd_impl(true, static_cast<const float*>(gs_bf16), e_in, env, rowptr, mean_rstd, gamma, beta, N, D,
                               e_out, aggr, parts_sum, parts_sq, groups, stream);
}

static int gate_scatter_bwd_stats_impl(bool half, const float* gs, const float* de_out, const float* daggr,
                                       const float* env, const int32_t* rowptr, const float* mean_rstd,
                                       const float* gamma, const float* beta, int32_t N, int32_t D,
                                       double* parts_a, double* parts_b, const CartnetGroups* groups,
                                       void* stream) {
  CN_CHECK(N >= 0 && D >= 4 && D % 4 == 0, "cartnet_gate_scatter_bwd_stats: D=%d must be a multiple of 4", D);
  CN_CHECK(gs && daggr && rowptr && mean_rstd && gamma && beta && parts_a && parts_b,
           "cartnet_gate_scatter_bwd_stats: null pointer");
  CN_CHECK(cn_groups_ok(groups), "cartnet_gate_scatter_bwd_stats: bad groups");
  if (half)
    hipLaunchKernelGGL((cn_gate_scatter_bwd_kernel<0, true>), cn_group_grid(groups, gate_parts(N), true), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), const_cast<float*>(gs), de_out, daggr, env, rowptr,
                       mean_rstd, gamma, beta, (const float*)nullptr, 0.f, N, D, parts_a, parts_b, 0,
                       groups ? groups->node_gptr : nullptr);
  else
    hipLaunchKernelGGL((cn_gate_scatter_bwd_kernel<0, false>), cn_group_grid(groups, gate_parts(N), true), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), const_cast<float*>(gs), de_out, daggr, env, rowptr,
                       mean_rstd, gamma, beta, (const float*)nullptr, 0.f, N, D, parts_a, parts_b, 0,
                       groups ? groups->node_gptr : nullptr);
  CN_LAUNCH_CHECK("cartnet_gate_scatter_bwd_stats");
  return 0;
}

extern "C" int cartnet_gate_scatter_bwd_stats(const float* gs, const float* de_out, const float* daggr,
                                              const float* env, const int32_t* rowptr, const float* mean_rstd,
                                              const float* gamma, const float* beta, int32_t N, int32_t D,
                                              double* parts_a, double* parts_b, const CartnetGroups* groups,
                                              void* stream) {
  return gate_scatter_bwd_stats_impl(false, gs, de_out, daggr, env, rowptr, mean_rstd, gamma, beta, N, D, parts_a, parts_b,
                                     groups, stream);
}
extern "C" int cartnet_gate_scatter_bwd_stats_h(const void* gs_bf16, const float* de_out, const float* daggr,
                                                const float* env, const int32_t* rowptr, const float* mean_rstd,
                                                const float* gamma, const float* beta, int32_t N, int32_t D,
                                                double* parts_a, double* parts_b, const CartnetGroups* groups,
                                                void* stream) {
  return gate_scatter_bwd_stats_impl(true, static_cast<const float*>(gs_bf16), de_out, daggr, env, rowptr, mean_rstd, gamma,
                                     beta, N, D, parts_a, parts_b, groups, stream);
}

static int gate_scatter_bwd_apply_impl(bool half, float* gs, const float* de_out, const float* daggr, const float* env,
                                       const int32_t* rowptr, const float* mean_rstd, const float* gamma,
                                       const float* beta, const float* sums, int64_t E, int32_t training,
                                       int32_t N, int32_t D, double* parts_dg, double* parts_ds,
                                       const CartnetGroups* groups, void* stream) {
  CN_CHECK(N >= 0 && D >= 4 && D % 4 == 0, "cartnet_gate_scatter_bwd_apply: D=%d must be a multiple of 4", D);
  CN_CHECK(gs && daggr && rowptr && mean_rstd && gamma && beta && sums && parts_dg && parts_ds,
           "cartnet_gate_scatter_bwd_apply: null pointer");
  const float inv = (training && E > 0) ? (float)(1.0 / (double)E) : 0.f;
  CN_CHECK(cn_groups_ok(groups), "cartnet_gate_scatter_bwd_apply: bad groups");
  if (half)
    hipLaunchKernelGGL((cn_gate_scatter_bwd_kernel<1, true>), cn_group_grid(groups, gate_parts(N), true), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), gs, de_out, daggr, env, rowptr, mean_rstd, gamma, beta,
                       sums, inv, N, D, parts_dg, parts_ds, /*reverse=*/1, groups ? groups->node_gptr : nullptr);
  else
    hipLaunchKernelGGL((cn_gate_scatter_bwd_kernel<1, false>), cn_group_grid(groups, gate_parts(N), true), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), gs, de_out, daggr, env, rowptr, mean_rstd, gamma, beta,
                       sums, inv, N, D, parts_dg, parts_ds, /*reverse=*/1, groups ? groups->node_gptr : nullptr);
  CN_LAUNCH_CHECK("cartnet_gate_scatter_bwd_apply");
  return 0;
}

extern "C" int cartnet_gate_scatter_bwd_apply(float* gs, const float* de_out, const float* daggr, const float* env,
                                              const int32_t* rowptr, const float* mean_rstd, const float* gamma,
                                              const float* beta, const float* sums, int64_t E, int32_t training,
                                              int32_t N, int32_t D, double* parts_dg, double* parts_ds,
                                              const CartnetGroups* groups, void* stream) {
  return gate_scatter_bwd_apply_impl(false, gs, de_out, daggr, env, rowptr, mean_rstd, gamma, beta, sums, E, training, N, D,
                                     parts_dg, parts_ds, groups, stream);
}
extern "C" int cartnet_gate_scatter_bwd_apply_h(void* gs_bf16, const float* de_out, const float* daggr, const float* env,
                                                const int32_t* rowptr, const float* mean_rstd, const float* gamma,
                                                const float* beta, const float* sums, int64_t E, int32_t training,
                                                int32_t N, int32_t D, double* parts_dg, double* parts_ds,
                                                const CartnetGroups* groups, void* stream) {
  return gate_scatter_bwd_apply_impl(true, static_cast<float*>(gs_bf16), de_out, daggr, env, rowptr, mean_rstd, gamma, beta,
                                     sums, E, training, N, D, parts_dg, parts_ds, groups, stream);
}

static int segment_sum_impl(bool half, const float* rows, int32_t ld, const int32_t* ptr, const int32_t* perm, int32_t N,
                            int32_t W, float* out, int32_t ldo, void* stream) {
  CN_CHECK(N >= 0 && W >= 4 && W % 4 == 0 && ld % 4 == 0 && ldo % 4 == 0 && ld >= W && ldo >= W,
           "cartnet_segment_sum: W=%d ld=%d ldo=%d must be multiples of 4", W, ld, ldo);
  if (N == 0) return 0;
  CN_CHECK(rows && ptr && out, "cartnet_segment_sum: null pointer");
  long long items = (long long)N * ((W + 255) / 256);
  long long blocks = (items + NODES_PER_BLOCK - 1) / NODES_PER_BLOCK;   // one item per wave up to 64k blocks
  if (blocks > 65536) blocks = 65536;
  if (half)
    hipLaunchKernelGGL(cn_segment_sum_kernel<true>, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       rows, ld, ptr, perm, N, W, out, ldo, /*reverse=*/perm ? 1 : 0);
  else
    hipLaunchKernelGGL(cn_segment_sum_kernel<false>, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       rows, ld, ptr, perm, N, W, out, ldo, /*reverse=*/perm ? 1 : 0);
  CN_LAUNCH_CHECK("cartnet_segment_sum");
  return 0;
}

static int segment_sum_pair_impl(bool half, const float* rows, int32_t ld, const int32_t* rowptr, const int32_t* colptr,
                                 const int32_t* perm, int32_t N, int32_t W, float* out_t, float* out_s, int32_t ldo,
                                 int32_t ochunk, void* stream) {
  CN_CHECK(N >= 0 && W >= 4 && W % 4 == 0 && ld % 4 == 0 && ldo % 4 == 0 && ld >= W && ochunk >= 256 && ochunk % 4 == 0 &&
               ldo >= ((W + 255) / 256 - 1) * ochunk + (W - (W - 1) / 256 * 256),
           "cartnet_segment_sum_pair: W=%d ld=%d ldo=%d ochunk=%d (multiples of 4; the last chunk must end inside a row)", W, ld,
           ldo, ochunk);
  if (N == 0) return 0;
  CN_CHECK(rows && rowptr && colptr && perm && out_t && out_s, "cartnet_segment_sum_pair: null pointer");
  long long items = 2LL * N * ((W + 255) / 256);
  long long blocks = (items + NODES_PER_BLOCK - 1) / NODES_PER_BLOCK;
  // One run of N / 8 consecutive atoms per XCD (kernel comment) when every block then has exactly one trip; else the plain
  // dealing.  Alone on the chip, 64 crystals x 194 atoms (tools/experiments/exp_segpair_xcd.py): 682 -> 465 MB fetched,
  // 102.6 -> 79.4 us per launch; runs of 128 / 256 / 512 atoms: 577 / 524 / 498 MB, 86.9 / 84.0 / 81.0 us.
  // (CN_SEG_XCD_NODES: 0 = plain dealing, > 0 = runs of that many atoms; A/B builds)
  int xcd_blocks = 0;
#ifndef CN_SEG_XCD_NODES
#define CN_SEG_XCD_NODES -1
#endif
  if (CN_SEG_XCD_NODES != 0) {
    const long long per_node = 2LL * ((W + 255) / 256);
    const long long nodes = CN_SEG_XCD_NODES > 0 ? (long long)CN_SEG_XCD_NODES : ((long long)N + 7) / 8;
    const long long run = (nodes * per_node + NODES_PER_BLOCK - 1) / NODES_PER_BLOCK;      // blocks per run
    const long long padded = (blocks + 8 * run - 1) / (8 * run) * (8 * run);
    if (padded <= 65536 && blocks >= 8 * run) { xcd_blocks = (int)run; blocks = padded; }
  }
  if (blocks > 65536) blocks = 65536;
  if (half)
    hipLaunchKernelGGL(cn_segment_sum_pair_kernel<true>, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       rows, ld, rowptr, colptr, perm, N, W, out_t, out_s, ldo, ochunk, xcd_blocks);
  else
    hipLaunchKernelGGL(cn_segment_sum_pair_kernel<false>, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       rows, ld, rowptr, colptr, perm, N, W, out_t, out_s, ldo, ochunk, xcd_blocks);
  CN_LAUNCH_CHECK("cartnet_segment_sum_pair");
  return 0;
}

extern "C" int cartnet_segment_sum_pair(const float* rows, int32_t ld, const int32_t* rowptr, const int32_t* colptr,
                                        const int32_t* perm, int32_t N, int32_t W, float* out_t, float* out_s, int32_t ldo,
                                        int32_t ochunk, void* stream) {
  return segment_sum_pair_impl(false, rows, ld, rowptr, colptr, perm, N, W, out_t, out_s, ldo, ochunk, stream);
}
extern "C" int cartnet_segment_sum_pair_h(const void* rows_bf16, int32_t ld, const int32_t* rowptr, const int32_t* colptr,
                                          const int32_t* perm, int32_t N, int32_t W, float* out_t, float* out_s,
                                          int32_t ldo, int32_t ochunk, void* stream) {
  return segment_sum_pair_impl(true, static_cast<const float*>(rows_bf16), ld, rowptr, colptr, perm, N, W, out_t, out_s, ldo,
                               ochunk, stream);
}

extern "C" int cartnet_segment_sum(const float* rows, int32_t ld, const int32_t* ptr, const int32_t* perm, int32_t N,
                                   int32_t W, float* out, int32_t ldo, void* stream) {
  return segment_sum_impl(false, rows, ld, ptr, perm, N, W, out, ldo, stream);
}
extern "C" int cartnet_segment_sum_h(const void* rows_bf16, int32_t ld, const int32_t* ptr, const int32_t* perm, int32_t N,
                                     int32_t W, float* out, int32_t ldo, void* stream) {
  return segment_sum_impl(true, static_cast<const float*>(rows_bf16), ld, ptr, perm, N, W, out, ldo, stream);
}

template <bool COMPACT>
static int segment_sum_long_impl(const char* who, const float* rows, int32_t ld, const int32_t* ptr, const int32_t* perm,
                                 int32_t nseg, int32_t total, int32_t W, float* tmp, float* out, int32_t ldo, void* stream) {
  CN_CHECK(nseg >= 1 && total >= 0 && W >= 4 && W % 4 == 0 && ld % 4 == 0 && ldo % 4 == 0 && ld >= W && ldo >= W,
           "%s: W=%d ld=%d ldo=%d must be multiples of 4", who, W, ld, ldo);
  CN_CHECK((rows || total == 0) && ptr && tmp && out, "%s: null pointer", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int slabs = (W + 255) / 256;
  const int nchunks = (total + LONG_CHUNK - 1) / LONG_CHUNK;
  if (total > 0) {
    long long items = (long long)nchunks * slabs;
    long long blocks = (items + NODES_PER_BLOCK - 1) / NODES_PER_BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cn_segment_long_pass1_kernel<COMPACT>, dim3((int)blocks), dim3(256), 0, st, rows, ld, ptr, perm,
                       nseg, total, W, tmp);
    CN_LAUNCH_CHECK(who);
  }
  long long blocks2 = ((long long)nseg * slabs + NODES_PER_BLOCK - 1) / NODES_PER_BLOCK;
  if (blocks2 > 4096) blocks2 = 4096;
  hipLaunchKernelGGL(cn_segment_long_pass2_kernel<COMPACT>, dim3((int)blocks2), dim3(256), 0, st, tmp, ptr, nseg, W, out,
                     ldo, nchunks);
  CN_LAUNCH_CHECK(who);
  return 0;
}

extern "C" int cartnet_segment_sum_long(const float* rows, int32_t ld, const int32_t* ptr, const int32_t* perm,
                                        int32_t nseg, int32_t total, int32_t W, float* tmp, float* out, int32_t ldo,
                                        void* stream) {
  return segment_sum_long_impl<false>("cartnet_segment_sum_long", rows, ld, ptr, perm, nseg, total, W, tmp, out, ldo, stream);
}

extern "C" int32_t cartnet_segment_chunked_rows(int32_t nseg, int32_t total) {
  return (total + LONG_CHUNK - 1) / LONG_CHUNK + (nseg > 0 ? nseg : 0);
}

extern "C" int cartnet_segment_sum_chunked_fold3(const float* rows, int32_t ld, const int32_t* ptr, int32_t nseg, int32_t total,
                                                 int32_t W, float* tmp, float* out, int32_t ldo, float* fold_out,
                                                 void* stream) {
  CN_CHECK(nseg >= 1 && total >= 0 && W >= 12 && W % 12 == 0 && ld % 4 == 0 && ldo % 4 == 0 && ld >= W && ldo >= W,
           "cartnet_segment_sum_chunked_fold3: W=%d must be a multiple of 12 (three pieces of 16-byte columns), ld=%d ldo=%d of 4",
           W, ld, ldo);
  CN_CHECK((rows || total == 0) && ptr && tmp && out && (fold_out || total == 0), "cartnet_segment_sum_chunked_fold3: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int slabs_f = (W / 3 + 255) / 256, slabs = (W + 255) / 256;
  const int nchunks = (total + LONG_CHUNK - 1) / LONG_CHUNK;
  if (total > 0) {
    long long blocks = ((long long)nchunks * slabs_f + NODES_PER_BLOCK - 1) / NODES_PER_BLOCK;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cn_segment_long_fold_pass1_kernel<3>, dim3((int)blocks), dim3(256), 0, st, rows, ld, ptr, nseg, total, W,
                       tmp, fold_out);
    CN_LAUNCH_CHECK("cartnet_segment_sum_chunked_fold3");
  }
  long long blocks2 = ((long long)nseg * slabs + NODES_PER_BLOCK - 1) / NODES_PER_BLOCK;
  if (blocks2 > 4096) blocks2 = 4096;
  hipLaunchKernelGGL(cn_segment_long_pass2_kernel<true>, dim3((int)blocks2), dim3(256), 0, st, tmp, ptr, nseg, W, out, ldo,
                     nchunks);
  CN_LAUNCH_CHECK("cartnet_segment_sum_chunked_fold3");
  return 0;
}

extern "C" int cartnet_segment_sum_chunked(const float* rows, int32_t ld, const int32_t* ptr, const int32_t* perm,
                                           int32_t nseg, int32_t total, int32_t W, float* tmp, float* out, int32_t ldo,
                                           void* stream) {
  return segment_sum_long_impl<true>("cartnet_segment_sum_chunked", rows, ld, ptr, perm, nseg, total, W, tmp, out, ldo, stream);
}
