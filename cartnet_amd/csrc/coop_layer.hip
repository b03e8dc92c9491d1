// PROTOTYPE (not on the model's path): ONE launch for the forward of one CartNet layer
// (/root/reference/models/cartnet.py:204-274) at the small batches of BASELINE configs[2] (N ~ 736 atoms, E ~ 10k edges),
// where today's chain of seven dependent launches -- node terms, GEMM 1 + gather, GEMM 2 + statistics, BatchNorm finalise,
// gate + segmented sum, BatchNorm finalise, node update -- lasts one tile's latency each plus a boundary.  256 workgroups of
// 256 threads, one per CU, walk five phases separated by four grid barriers:
//   P1  Pn[N, 4D] = x W_{gate_i | aggr_i | gate_j | aggr_j}                                     (48 tile jobs of 64 x 256)
//   P2  per 64-edge tile: pre = e W1e + Pn_i[tgt] + Pn_j[src] + b1 (kept), h = silu(pre) in LDS, g | s = h W2 + b2 (kept);
//       per-tile partial BatchNorm sums of g; g and s stay in REGISTERS across the barrier
//   P3  every workgroup reduces the partials (mean, rstd), sigma = env sigmoid(bn(g)), e_out = e + sigma, sigma s through LDS,
//       one thread per column sums the tile's rows by target run -> part[tile][run]
//   P4  32 workgroups: aggr[n] = sum of the runs of node n, partial sums for the node BatchNorm
//   P5  the same 32: x_out = silu(bn(aggr)) + x
// Arithmetic is precision 2 (operands rounded to bf16, v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 storage); whole-K
// operands: the A tile sits in LDS for all of K, a wave's 64 weight columns come straight from L2 as MFMA fragments (the
// layer's weights are 1 MB of bf16).  Running statistics are not updated (forward values only).  Training-mode statistics,
// envelope on.  Requires: D = 256, every atom has at least one incoming edge (run index = target - first target of the tile).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int CL_D = 256, CL_ROWS = 64, CL_THREADS = 256, CL_GRID = 256;
constexpr int CL_LDA = CL_D + 8;          // bf16 elements per LDS row of a K = 256 operand tile (16-byte shift per row: conflict-free b128)
constexpr int CL_LDH = 2 * CL_D + 8;      // ... of the K = 512 hidden tile
constexpr int CL_NODE_WGS = 32;
constexpr int CL_MAXRUN = CL_ROWS + 1;

struct CoopArgs {
  const float *x, *e, *env;
  const int32_t *tgt, *src, *rowptr;
  const __bf16 *wn, *w1e, *w2;            // [4D][D], [2D][D], [2D][D] (row n = output column, k contiguous)
  const float *b1, *b2, *bn1_w, *bn1_b, *bn2_w, *bn2_b;
  float *Pn, *pre, *gs, *e_out, *aggr, *x_out;
  float *stat1, *part, *stat2;            // [tiles][2D], [tiles][CL_MAXRUN][D], [CL_NODE_WGS][2D]
  unsigned *bar;                          // 3 x 8 x 32 words
  unsigned epoch;
  int N, E, tiles_e, tiles_n;
  float eps;
  unsigned *status;
  unsigned long long *stamps;             // [256][16] 100 MHz times at the phase boundaries (diagnostic; may be NULL)
};

__device__ __forceinline__ float cl_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// Hierarchical counter barrier over 256 workgroups: 32 per group (blockIdx % 8: the blocks that share an XCD), the last of a
// group reports to the top counter, the last group releases all eight generation words.  Monotonic counts: `bar` is zeroed
// once, `epoch` = the number of launches on it so far (0 for the first), barrier k of a launch is generation 4 epoch + k.  Bounded spin: a workgroup that gives up sets the status word (results are then wrong, nothing hangs).
__device__ __forceinline__ void cl_grid_barrier(const CoopArgs& p, unsigned g) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* cnt = p.bar;
    unsigned* top = p.bar + 8 * 32;
    unsigned* gen = p.bar + 9 * 32;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned xg = blockIdx.x & 7;
    const unsigned old = __hip_atomic_fetch_add(&cnt[xg * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == g * (CL_GRID / 8)) {
      const unsigned o2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o2 + 1 == g * 8)
        for (int i = 0; i < 8; ++i) __hip_atomic_store(&gen[i * 32], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int spins = 0;
    while ((int)(__hip_atomic_load(&gen[xg * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 21)) {
        *p.status = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// 64 rows x 256 k of an fp32 matrix -> bf16 LDS tile (rows past `rows` repeat the last row)
__device__ __forceinline__ void cl_stage(const float* __restrict__ src, int ld, int row0, int rows, __bf16* s, int lds_ld) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int u = threadIdx.x + i * CL_THREADS, r = u >> 6, c4 = u & 63;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)min(row0 + r, rows - 1) * ld + c4 * 4);
    *reinterpret_cast<bf16x4*>(s + r * lds_ld + c4 * 4) = __builtin_convertvector(v, bf16x4);
  }
}

// This wave's 64 rows x 64 columns (columns n0 .. n0 + 63 of W [.][256]) of sA[64][K = 256 at column koff] W^T
__device__ __forceinline__ void cl_gemm(const __bf16* sA, int lds_ld, int koff, const __bf16* __restrict__ W, int n0,
                                        f32x16 (&acc)[2][2]) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  bf16x8 bfr[16][2];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int b = 0; b < 2; ++b)
      bfr[ks][b] = *reinterpret_cast<const bf16x8*>(W + (size_t)(n0 + b * 32 + li) * CL_D + ks * 16 + lh * 8);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    bf16x8 afr[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
      afr[a] = *reinterpret_cast<const bf16x8*>(sA + (a * 32 + li) * lds_ld + koff + ks * 16 + lh * 8);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[a], bfr[ks][b], acc[a][b], 0, 0, 0);
  }
}

__global__ __launch_bounds__(CL_THREADS, 1) void cn_coop_layer_fwd_kernel(const CoopArgs p) {
  // LDS: [A tile bf16 64 x 264 | hidden tile bf16 64 x 520] (P1, P2), then [sigma s fp32 64 x 257 | targets | statistics] (P3)
  __shared__ __attribute__((aligned(16))) char lds[64 * CL_LDA * 2 + 64 * CL_LDH * 2 + 512];
  __bf16* sA = reinterpret_cast<__bf16*>(lds);
  __bf16* sH = reinterpret_cast<__bf16*>(lds + 64 * CL_LDA * 2);
  float* sM = reinterpret_cast<float*>(lds);                       // 64 x 257 fp32 = 65,792 B
  int* sT = reinterpret_cast<int*>(lds + 64 * 257 * 4);            // 64 targets
  float* sS = reinterpret_cast<float*>(lds + 64 * 257 * 4 + 256);  // mean | rstd (2 x 256)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int w = blockIdx.x;
  const unsigned g0 = p.epoch * 4;
  f32x16 acc[2][2], accG[2][2], accS[2][2];
  int stamp_i = 0;
  auto stamp = [&]() {
    if (p.stamps && tid == 0) p.stamps[w * 16 + stamp_i] = __builtin_amdgcn_s_memrealtime();
    ++stamp_i;
  };
  stamp();

  // ---- P1: node terms
  if (w < p.tiles_n * 4) {
    const int rt = w >> 2, cb = w & 3;
    cl_stage(p.x, CL_D, rt * CL_ROWS, p.N, sA, CL_LDA);
    __syncthreads();
    cl_gemm(sA, CL_LDA, 0, p.wn, cb * CL_D + wv * 64, acc);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * CL_ROWS + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.N)
#pragma unroll
          for (int b = 0; b < 2; ++b) p.Pn[(size_t)row * (4 * CL_D) + cb * CL_D + wv * 64 + b * 32 + li] = acc[a][b][r];
      }
  }
  stamp();
  cl_grid_barrier(p, g0 + 1);
  stamp();

  // ---- P2: the edge MLPs of one 64-edge tile
  const bool has_tile = w < p.tiles_e;
  const int e0 = w * CL_ROWS;
  int* sIdx = reinterpret_cast<int*>(lds + 64 * CL_LDA * 2 + 64 * CL_LDH * 2);       // [tgt 64 | src 64]
  if (has_tile) {
    cl_stage(p.e, CL_D, e0, p.E, sA, CL_LDA);
    if (tid < 2 * CL_ROWS) sIdx[tid] = (tid < CL_ROWS ? p.tgt : p.src)[min(e0 + (tid & 63), p.E - 1)];
    __syncthreads();
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {            // columns [half*256, +256) of pre: gate MLP | sender MLP
      cl_gemm(sA, CL_LDA, 0, p.w1e, half * CL_D + wv * 64, acc);
      const float bias0 = p.b1[half * CL_D + wv * 64 + li], bias1 = p.b1[half * CL_D + wv * 64 + 32 + li];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float gi[16][2], gj[16][2];                   // a block row's 64 node-term values: all loads before the first use
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int ti = sIdx[rl], sj = sIdx[CL_ROWS + rl];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int col = half * CL_D + wv * 64 + b * 32 + li;
            gi[r][b] = p.Pn[(size_t)ti * (4 * CL_D) + col];
            gj[r][b] = p.Pn[(size_t)sj * (4 * CL_D) + 2 * CL_D + col];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, row = min(e0 + rl, p.E - 1);
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int col = half * CL_D + wv * 64 + b * 32 + li;
            const float v = acc[a][b][r] + (b ? bias1 : bias0) + gi[r][b] + gj[r][b];
            if (e0 + rl < p.E) p.pre[(size_t)row * (2 * CL_D) + col] = v;
            sH[rl * CL_LDH + col] = (__bf16)(v * cl_sigmoid(v));
          }
        }
      }
    }
    __syncthreads();
    cl_gemm(sH, CL_LDH, 0, p.w2, wv * 64, accG);                   // g = h[:, :D] W2g^T
    cl_gemm(sH, CL_LDH, CL_D, p.w2, CL_D + wv * 64, accS);         // s = h[:, D:] W2a^T
    float cs[2] = {0.f, 0.f}, cq[2] = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool in = e0 + rl < p.E;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int col = wv * 64 + b * 32 + li;
          accG[a][b][r] += p.b2[col];
          accS[a][b][r] += p.b2[CL_D + col];
          if (in) {
            p.gs[(size_t)(e0 + rl) * (2 * CL_D) + col] = accG[a][b][r];
            p.gs[(size_t)(e0 + rl) * (2 * CL_D) + CL_D + col] = accS[a][b][r];
            cs[b] += accG[a][b][r];
            cq[b] += accG[a][b][r] * accG[a][b][r];
          }
        }
      }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      cs[b] += __shfl_xor(cs[b], 32);
      cq[b] += __shfl_xor(cq[b], 32);
      if (lh == 0) {
        p.stat1[(size_t)w * (2 * CL_D) + wv * 64 + b * 32 + li] = cs[b];
        p.stat1[(size_t)w * (2 * CL_D) + CL_D + wv * 64 + b * 32 + li] = cq[b];
      }
    }
  }
  stamp();
  cl_grid_barrier(p, g0 + 2);
  stamp();

  // ---- P3: statistics, gate, per-run sums
  if (has_tile) {
    {
      double s = 0.0, q = 0.0;
      int t = 0;
      for (; t + 8 <= p.tiles_e; t += 8) {            // sixteen independent loads in flight, added in tile order
        float vs[8], vq[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          vs[k] = p.stat1[(size_t)(t + k) * (2 * CL_D) + tid];
          vq[k] = p.stat1[(size_t)(t + k) * (2 * CL_D) + CL_D + tid];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s += (double)vs[k];
          q += (double)vq[k];
        }
      }
      for (; t < p.tiles_e; ++t) {
        s += (double)p.stat1[(size_t)t * (2 * CL_D) + tid];
        q += (double)p.stat1[(size_t)t * (2 * CL_D) + CL_D + tid];
      }
      const double mean = s / p.E, var = q / p.E - mean * mean;
      sS[tid] = (float)mean;
      sS[CL_D + tid] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)p.eps));
      if (tid < CL_ROWS) sT[tid] = p.tgt[min(e0 + tid, p.E - 1)];
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, row = min(e0 + rl, p.E - 1);
        const float ev = p.env[row];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int col = wv * 64 + b * 32 + li;
          const float ghat = (accG[a][b][r] - sS[col]) * sS[CL_D + col];
          const float sg = ev * cl_sigmoid(ghat * p.bn1_w[col] + p.bn1_b[col]);
          if (e0 + rl < p.E) p.e_out[(size_t)row * CL_D + col] = p.e[(size_t)row * CL_D + col] + sg;
          sM[rl * 257 + col] = sg * accS[a][b][r];
        }
      }
    __syncthreads();
    {   // thread = column: the tile's rows in order, one partial sum per run of equal targets
      const int nrows = min(CL_ROWS, p.E - e0), t_first = sT[0];
      float accum = 0.f;
      int tcur = t_first;
      for (int r = 0; r < nrows; ++r) {
        const int t = sT[r];
        if (t != tcur) {
          p.part[((size_t)w * CL_MAXRUN + (tcur - t_first)) * CL_D + tid] = accum;
          accum = 0.f;
          tcur = t;
        }
        accum += sM[r * 257 + tid];
      }
      p.part[((size_t)w * CL_MAXRUN + (tcur - t_first)) * CL_D + tid] = accum;
    }
  }
  stamp();
  cl_grid_barrier(p, g0 + 3);
  stamp();

  // ---- P4: per-atom sums + node BatchNorm partials (32 workgroups)
  const int per = (p.N + CL_NODE_WGS - 1) / CL_NODE_WGS, n_lo = w * per, n_hi = min(p.N, n_lo + per);
  if (w < CL_NODE_WGS) {
    float s = 0.f, q = 0.f;
    for (int n = n_lo; n < n_hi; n += 4) {            // four atoms' run partials in flight (an atom's rows touch 1-2 tiles)
      float a[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
      int t1s[4], t0s[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int nn = min(n + k, n_hi - 1);
        const int r0 = p.rowptr[nn], r1 = p.rowptr[nn + 1];
        t0s[k] = r0 / CL_ROWS;
        t1s[k] = (r1 - 1) / CL_ROWS;
        a[k] = p.part[((size_t)t0s[k] * CL_MAXRUN + (nn - p.tgt[t0s[k] * CL_ROWS])) * CL_D + tid];
        if (t1s[k] > t0s[k]) a2[k] = p.part[((size_t)(t0s[k] + 1) * CL_MAXRUN + (nn - p.tgt[(t0s[k] + 1) * CL_ROWS])) * CL_D + tid];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (n + k >= n_hi) break;
        float v = a[k] + a2[k];
        for (int t = t0s[k] + 2; t <= t1s[k]; ++t)       // (more than 128 incoming edges: rare)
          v += p.part[((size_t)t * CL_MAXRUN + (n + k - p.tgt[t * CL_ROWS])) * CL_D + tid];
        p.aggr[(size_t)(n + k) * CL_D + tid] = v;
        s += v;
        q += v * v;
      }
    }
    p.stat2[(size_t)w * (2 * CL_D) + tid] = s;
    p.stat2[(size_t)w * (2 * CL_D) + CL_D + tid] = q;
  }
  stamp();
  cl_grid_barrier(p, g0 + 4);
  stamp();

  // ---- P5: node update
  if (w < CL_NODE_WGS) {
    double s = 0.0, q = 0.0;
    {
      float vs[CL_NODE_WGS], vq[CL_NODE_WGS];
#pragma unroll
      for (int t = 0; t < CL_NODE_WGS; ++t) {
        vs[t] = p.stat2[(size_t)t * (2 * CL_D) + tid];
        vq[t] = p.stat2[(size_t)t * (2 * CL_D) + CL_D + tid];
      }
#pragma unroll
      for (int t = 0; t < CL_NODE_WGS; ++t) {
        s += (double)vs[t];
        q += (double)vq[t];
      }
    }
    const double mean = s / p.N, var = q / p.N - mean * mean;
    const float mu = (float)mean, rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)p.eps));
    const float gw = p.bn2_w[tid], gb = p.bn2_b[tid];
    for (int n = n_lo; n < n_hi; ++n) {
      const float v = (p.aggr[(size_t)n * CL_D + tid] - mu) * rstd * gw + gb;
      p.x_out[(size_t)n * CL_D + tid] = v * cl_sigmoid(v) + p.x[(size_t)n * CL_D + tid];
    }
  }
  stamp();
}

}  // namespace

extern "C" size_t cartnet_coop_layer_workspace_floats(int32_t N, int32_t E) {
  const size_t tiles_e = (size_t)(E + CL_ROWS - 1) / CL_ROWS;
  return tiles_e * 2 * CL_D + tiles_e * CL_MAXRUN * CL_D + (size_t)CL_NODE_WGS * 2 * CL_D + CL_GRID * 16 * 2;   // + stamps
}

extern "C" int cartnet_coop_layer_fwd(const float* x, const float* e, const int32_t* tgt, const int32_t* src,
                                      const int32_t* rowptr, const float* env, const void* wn_bf16, const void* w1e_bf16,
                                      const void* w2_bf16, const float* b1, const float* b2, const float* bn1_w,
                                      const float* bn1_b, const float* bn2_w, const float* bn2_b, int32_t N, int32_t E,
                                      float eps, float* Pn, float* pre, float* gs, float* e_out, float* aggr, float* x_out,
                                      float* work, uint32_t* bar, uint32_t epoch, uint32_t* status, void* stream) {
  CN_CHECK(x && e && tgt && src && rowptr && env && wn_bf16 && w1e_bf16 && w2_bf16 && b1 && b2 && bn1_w && bn1_b && bn2_w &&
               bn2_b && Pn && pre && gs && e_out && aggr && x_out && work && bar && status,
           "cartnet_coop_layer_fwd: null pointer");
  CN_CHECK(N >= 1 && E >= 1 && E <= CL_GRID * CL_ROWS && epoch < (1u << 24),
           "cartnet_coop_layer_fwd: N=%d E=%d epoch=%u out of the prototype's range (E <= %d: one 64-edge tile per workgroup)",
           N, E, epoch, CL_GRID * CL_ROWS);
  CoopArgs a;
  a.x = x; a.e = e; a.env = env; a.tgt = tgt; a.src = src; a.rowptr = rowptr;
  a.wn = static_cast<const __bf16*>(wn_bf16); a.w1e = static_cast<const __bf16*>(w1e_bf16); a.w2 = static_cast<const __bf16*>(w2_bf16);
  a.b1 = b1; a.b2 = b2; a.bn1_w = bn1_w; a.bn1_b = bn1_b; a.bn2_w = bn2_w; a.bn2_b = bn2_b;
  a.Pn = Pn; a.pre = pre; a.gs = gs; a.e_out = e_out; a.aggr = aggr; a.x_out = x_out;
  a.tiles_e = (E + CL_ROWS - 1) / CL_ROWS;
  a.tiles_n = (N + CL_ROWS - 1) / CL_ROWS;
  CN_CHECK(a.tiles_n * 4 <= CL_GRID, "cartnet_coop_layer_fwd: N=%d needs more than %d node-term jobs", N, CL_GRID);
  a.stat1 = work;
  a.part = work + (size_t)a.tiles_e * 2 * CL_D;
  a.stat2 = a.part + (size_t)a.tiles_e * CL_MAXRUN * CL_D;
  a.stamps = reinterpret_cast<unsigned long long*>(a.stat2 + (size_t)CL_NODE_WGS * 2 * CL_D);
  a.bar = bar; a.epoch = epoch; a.N = N; a.E = E; a.eps = eps; a.status = status;
  void* params[] = {&a};
  // cooperative launch: the runtime checks that all 256 workgroups are resident together (the barrier needs them to be)
  hipError_t err = hipLaunchCooperativeKernel(reinterpret_cast<void*>(cn_coop_layer_fwd_kernel), dim3(CL_GRID), dim3(CL_THREADS),
                                              params, 0, reinterpret_cast<hipStream_t>(stream));
  CN_CHECK(err == hipSuccess, "cartnet_coop_layer_fwd: cooperative launch failed: %s", hipGetErrorString(err));
  return 0;
}
