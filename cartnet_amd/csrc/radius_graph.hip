// Periodic radius graph on the GPU (reference: dataset/utils.py:57-237 radius_graph_pbc, used by
// dataset/figshare_dataset.py:65-68).  Emits edges in the reference's order -- target atom, then source atom, then
// periodic image in cartesian_prod(a1, a2, a3) order -- so edge_index[1] comes out sorted and feeds cartnet_csr_build
// directly.  Two passes (count, fill), one wavefront per target atom, one lane per source atom (64 sources per round);
// a lane walks only the periodic images that CAN lie within the radius of its pair (see "image box" below) in the
// reference's image order; the position of a lane's edges inside the row comes from a wave prefix sum of the lanes'
// counts, so there are no atomics and the output is deterministic.
// Arithmetic of the distance test mirrors the reference's fp32 operation order with explicitly rounded (non-fused)
// operations; the image box only decides which candidates are tested, with a margin far above fp32 rounding.
//
// Image box.  The reference tests every image u in [-R1,R1] x [-R2,R2] x [-R3,R3] (R_d = ceil(radius |b_d|), b_d the
// reciprocal lattice vectors) against every pair: 27-125 distance tests per pair of which a handful pass.  With
// f_d = (p_i - p_j) . b_d the fractional offset of the pair, |f_d - u_d| = |(p_i - p_j - cell^T u) . b_d| <= d |b_d|, so an
// image within the radius has u_d in [f_d - radius |b_d|, f_d + radius |b_d|]: at the benchmark crystals (a ~ 13 A,
// radius 5) that is 1-2 values per axis instead of 3, ~6 tests per pair instead of 27.  Same edges, same order.
#include "common.h"
#include <math.h>

// This file is compiled with -ffp-contract=off (cartnet_amd/build.py: EXTRA_FLAGS).  HIP's __fmul_rn / __fadd_rn are
// plain * and +, and under hipcc's default -ffp-contract=fast-honor-pragmas the compiler fused them into FMAs --
// differently in the count and the fill instantiation of the kernel below, which then disagreed about a pair whose d^2
// lies within an ulp of radius^2 (2 of 49k atoms in one 256-crystal launch: two slots of the fill pass stayed unwritten
// and every later crystal of the chunk was shifted).  With contraction off every product and sum is rounded on its own,
// like the reference's torch ops, and both passes evaluate the same expression.
namespace {

constexpr int CN_RG_MAX_REPS = 16;

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }

// reps[g, d] = ceil(radius * |cross(a_{d+1}, a_{d+2}) / V|)   (dataset/utils.py:133-157)
__global__ void cn_rg_reps_kernel(const float* __restrict__ cell, int Bg, float radius, int* __restrict__ reps,
                                  float* __restrict__ recip /* [Bg][12]: b_1, b_2, b_3, radius |b_d| */) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= Bg) return;
  const float* a = cell + (size_t)g * 9;
  auto cross = [&](const float* u, const float* v, float* o) {
    o[0] = sub(mul(u[1], v[2]), mul(u[2], v[1]));
    o[1] = sub(mul(u[2], v[0]), mul(u[0], v[2]));
    o[2] = sub(mul(u[0], v[1]), mul(u[1], v[0]));
  };
  float c23[3], c31[3], c12[3];
  cross(a + 3, a + 6, c23);
  cross(a + 6, a + 0, c31);
  cross(a + 0, a + 3, c12);
  const float vol = add(add(mul(a[0], c23[0]), mul(a[1], c23[1])), mul(a[2], c23[2]));
  const float* cs[3] = {c23, c31, c12};
  for (int d = 0; d < 3; ++d) {
    const float x = cs[d][0] / vol, y = cs[d][1] / vol, z = cs[d][2] / vol;
    const float nrm = sqrtf(add(add(mul(x, x), mul(y, y)), mul(z, z)));
    // a degenerate cell (zero volume: inf / NaN here) must not become an endless image loop: repetitions are capped at
    // CN_RG_MAX_REPS (a lattice vector shorter than radius / 16 is not a crystal), NaN gives 0
    const float want = ceilf(mul(radius, nrm));
    reps[g * 3 + d] = (want >= 0.f && want <= (float)CN_RG_MAX_REPS) ? (int)want : (want > (float)CN_RG_MAX_REPS ? CN_RG_MAX_REPS : 0);
    if (recip) {
      recip[g * 12 + d * 3] = x;
      recip[g * 12 + d * 3 + 1] = y;
      recip[g * 12 + d * 3 + 2] = z;
      recip[g * 12 + 9 + d] = radius * nrm;
    }
  }
}

template <bool FILL>
__global__ __launch_bounds__(256) void cn_rg_kernel(const float* __restrict__ pos, const float* __restrict__ cell,
                                                    const int64_t* __restrict__ graph_ptr,
                                                    const int64_t* __restrict__ batch, const int* __restrict__ reps,
                                                    const float* __restrict__ recip, int N, float r2, float eps2,
                                                    int* __restrict__ deg, const int64_t* __restrict__ rowptr,
                                                    int64_t* __restrict__ ei, long long E, float* __restrict__ dist,
                                                    float* __restrict__ dir, float* __restrict__ dist_sq) {
  const int lane = threadIdx.x & 63;
  const int i1 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i1 >= N) return;
  const int g = (int)batch[i1];
  const int n0 = (int)graph_ptr[g], n = (int)graph_ptr[g + 1] - n0;
  const int R1 = reps[g * 3], R2 = reps[g * 3 + 1], R3 = reps[g * 3 + 2];
  const float* a = cell + (size_t)g * 9;
  const float* rb = recip + (size_t)g * 12;
  const float px = pos[(size_t)i1 * 3], py = pos[(size_t)i1 * 3 + 1], pz = pos[(size_t)i1 * 3 + 2];
  long long out = FILL ? rowptr[i1] : 0;
  int count = 0;
  for (int base = 0; base < n; base += 64) {
    const int i2 = base + lane;
    const bool have = i2 < n;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    int lo1 = 0, hi1 = -1, lo2 = 0, hi2 = -1, lo3 = 0, hi3 = -1;     // empty box for lanes past the last source
    if (have) {
      const float* q = pos + (size_t)(n0 + i2) * 3;
      qx = q[0]; qy = q[1]; qz = q[2];
      const float ex = px - qx, ey = py - qy, ez = pz - qz;
      // fractional offset of the pair and the image interval per axis; the margin (1e-3 of a lattice step, plus 1e-5
      // relative) is four orders above the rounding of these dot products -- it only ever ADDS candidates
      const float f1 = ex * rb[0] + ey * rb[1] + ez * rb[2];
      const float f2 = ex * rb[3] + ey * rb[4] + ez * rb[5];
      const float f3 = ex * rb[6] + ey * rb[7] + ez * rb[8];
      const float m1 = rb[9] + 1e-3f + 1e-5f * fabsf(f1), m2 = rb[10] + 1e-3f + 1e-5f * fabsf(f2),
                  m3 = rb[11] + 1e-3f + 1e-5f * fabsf(f3);
      // clamped in floating point before the conversion (an int conversion of inf / NaN is undefined; fmaxf / fminf
      // drop a NaN operand, which leaves the reference's full range)
      auto lo_of = [](float v, int R) { return (int)fminf((float)(R + 1), fmaxf((float)-R, ceilf(v))); };
      auto hi_of = [](float v, int R) { return (int)fmaxf((float)(-R - 1), fminf((float)R, floorf(v))); };
      lo1 = lo_of(f1 - m1, R1); hi1 = hi_of(f1 + m1, R1);
      lo2 = lo_of(f2 - m2, R2); hi2 = hi_of(f2 + m2, R2);
      lo3 = lo_of(f3 - m3, R3); hi3 = hi_of(f3 + m3, R3);
    }
    // every image of the box in cartesian_prod order (u1 slowest, u3 fastest); `emit` decides what happens to a hit
    auto walk = [&](auto emit) {
      for (int u1i = lo1; u1i <= hi1; ++u1i)
        for (int u2i = lo2; u2i <= hi2; ++u2i)
          for (int u3i = lo3; u3i <= hi3; ++u3i) {
            const float u1 = (float)u1i, u2 = (float)u2i, u3 = (float)u3i;
            // image offset = cell^T u, accumulated as (a1 u1 + a3 u3) + a2 u2 -- the order torch.bmm uses for this sum
            const float ox = add(add(mul(a[0], u1), mul(a[6], u3)), mul(a[3], u2));
            const float oy = add(add(mul(a[1], u1), mul(a[7], u3)), mul(a[4], u2));
            const float oz = add(add(mul(a[2], u1), mul(a[8], u3)), mul(a[5], u2));
            const float dx = sub(px, add(qx, ox)), dy = sub(py, add(qy, oy)), dz = sub(pz, add(qz, oz));
            const float d2 = add(add(mul(dx, dx), mul(dy, dy)), mul(dz, dz));
            if ((d2 <= r2) && (d2 > eps2)) emit(dx, dy, dz, d2);
          }
    };
    int mine = 0;
    walk([&](float, float, float, float) { ++mine; });
    // inclusive prefix sum of the lanes' counts: lane l's edges follow those of lanes 0..l-1 (source order)
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    const int round_total = __shfl(incl, 63);
    if (FILL) {
      long long p = out + (incl - mine);
      walk([&](float dx, float dy, float dz, float d2) {
        if (p < E) {
          ei[p] = (int64_t)(n0 + i2);
          ei[E + p] = (int64_t)i1;
          const float d = sqrtf(d2);
          const float dn = fmaxf(d, 1e-12f);      // F.normalize(vec, p=2, dim=-1, eps=1e-12)
          dist[p] = d;
          if (dist_sq) dist_sq[p] = d2;
          dir[p * 3] = dx / dn;
          dir[p * 3 + 1] = dy / dn;
          dir[p * 3 + 2] = dz / dn;
        }
        ++p;
      });
      out += round_total;
    } else {
      count += round_total;
    }
  }
  if (!FILL && lane == 0) deg[i1] = count;
}

// Neighbour cap (dataset/utils.py:240-360 get_max_neighbors_mask, enforce_max_strictly = False): a target with more
// than k candidate edges keeps those with d^2 <= (k+1)-th smallest d^2 of its row + tolerance, in their original
// order; shorter rows are kept whole.  One wavefront per target.  The (k+1)-th smallest value is found by rank
// counting (every lane ranks its own elements against the whole row), which needs no sort and no scratch.
__global__ __launch_bounds__(256) void cn_cap_count_kernel(const int64_t* __restrict__ rowptr,
                                                           const float* __restrict__ d2, int N, int k, float tol,
                                                           float* __restrict__ cutoff, int* __restrict__ deg) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= N) return;
  const long long b = rowptr[i];
  const int n = (int)(rowptr[i + 1] - b);
  if (n <= k) {
    if (lane == 0) { cutoff[i] = INFINITY; deg[i] = n; }
    return;
  }
  const float* row = d2 + b;
  float kth = 0.f;
  bool found = false;
  for (int e = lane; e < n && !found; e += 64) {
    const float v = row[e];
    int lo = 0, eq = 0;
    for (int j = 0; j < n; ++j) {
      const float w = row[j];
      lo += (w < v);
      eq += (w == v);
    }
    if (lo <= k && k < lo + eq) { kth = v; found = true; }
  }
  // exactly the lanes holding the k-th value found it, and they all hold the same number
  const unsigned long long m = __ballot(found);
  const int owner = m ? __ffsll((long long)m) - 1 : 0;
  kth = m ? __shfl(kth, owner, 64) : INFINITY;          // m == 0 only if the row holds NaNs: keep it whole
  const float c = __fadd_rn(kth, tol);
  int cnt = 0;
  for (int e = lane; e < n; e += 64) cnt += (row[e] <= c);
  for (int off = 32; off; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if (lane == 0) { cutoff[i] = c; deg[i] = cnt; }
}

__global__ __launch_bounds__(256) void cn_cap_fill_kernel(const int64_t* __restrict__ rowptr,
                                                          const int64_t* __restrict__ rowptr_out,
                                                          const float* __restrict__ cutoff,
                                                          const float* __restrict__ d2, const int64_t* __restrict__ ei,
                                                          const float* __restrict__ dist, const float* __restrict__ dir,
                                                          int N, long long E, long long E_out,
                                                          int64_t* __restrict__ ei_out, float* __restrict__ dist_out,
                                                          float* __restrict__ dir_out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= N) return;
  const long long b = rowptr[i];
  const int n = (int)(rowptr[i + 1] - b);
  const float c = cutoff[i];
  long long out = rowptr_out[i];
  for (int base = 0; base < n; base += 64) {
    const int e = base + lane;
    const bool keep = e < n && d2[b + e] <= c;
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const long long p = out + __popcll(m & ((1ull << lane) - 1ull));
      if (p < E_out) {
        const long long q = b + e;
        ei_out[p] = ei[q];
        ei_out[E_out + p] = ei[E + q];
        dist_out[p] = dist[q];
        dir_out[p * 3] = dir[q * 3];
        dir_out[p * 3 + 1] = dir[q * 3 + 1];
        dir_out[p * 3 + 2] = dir[q * 3 + 2];
      }
    }
    out += __popcll(m);
  }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int cartnet_radius_graph_count(const float* pos, const float* cell, const int64_t* graph_ptr,
                                          const int64_t* batch, int32_t N, int32_t Bg, float radius, int32_t* reps,
                                          int32_t* deg, void* stream) {
  CN_CHECK(N >= 0 && Bg >= 1 && radius > 0.f, "cartnet_radius_graph_count: bad sizes");
  CN_CHECK(cell && graph_ptr && reps && (N == 0 || (pos && batch && deg)), "cartnet_radius_graph_count: null pointer");
  float* recip = reinterpret_cast<float*>(reps + 3 * (size_t)Bg);      // second part of the caller's [15 * Bg] buffer
  hipLaunchKernelGGL(cn_rg_reps_kernel, dim3(cn_ceil_div(Bg, 64)), dim3(64), 0, ST(stream), cell, Bg, radius, reps, recip);
  CN_LAUNCH_CHECK("cartnet_radius_graph_count/reps");
  if (N == 0) return 0;
  hipLaunchKernelGGL(cn_rg_kernel<false>, dim3(cn_ceil_div(N, 4)), dim3(256), 0, ST(stream), pos, cell, graph_ptr, batch,
                     reps, recip, N, radius * radius, 0.0001f, deg, (const int64_t*)nullptr, (int64_t*)nullptr, 0LL,
                     (float*)nullptr, (float*)nullptr, (float*)nullptr);
  CN_LAUNCH_CHECK("cartnet_radius_graph_count");
  return 0;
}

extern "C" int cartnet_radius_graph_fill(const float* pos, const float* cell, const int64_t* graph_ptr,
                                         const int64_t* batch, const int32_t* reps, const int64_t* rowptr, int32_t N,
                                         int32_t Bg, float radius, int64_t E, int64_t* edge_index, float* cart_dist,
                                         float* cart_dir, float* cart_dist_sq, void* stream) {
  CN_CHECK(N >= 0 && Bg >= 1 && radius > 0.f && E >= 0, "cartnet_radius_graph_fill: bad sizes");
  if (N == 0 || E == 0) return 0;
  CN_CHECK(pos && cell && graph_ptr && batch && reps && rowptr && edge_index && cart_dist && cart_dir,
           "cartnet_radius_graph_fill: null pointer");
  const float* recip = reinterpret_cast<const float*>(reps + 3 * (size_t)Bg);
  hipLaunchKernelGGL(cn_rg_kernel<true>, dim3(cn_ceil_div(N, 4)), dim3(256), 0, ST(stream), pos, cell, graph_ptr, batch,
                     reps, recip, N, radius * radius, 0.0001f, (int*)nullptr, rowptr, edge_index, (long long)E, cart_dist,
                     cart_dir, cart_dist_sq);
  CN_LAUNCH_CHECK("cartnet_radius_graph_fill");
  return 0;
}

extern "C" int cartnet_neighbor_cap_count(const int64_t* rowptr, const float* dist_sq, int32_t N, int32_t max_neighbors,
                                          float tolerance, float* cutoff, int32_t* deg, void* stream) {
  CN_CHECK(N >= 0 && max_neighbors >= 1 && tolerance >= 0.f, "cartnet_neighbor_cap_count: bad sizes");
  if (N == 0) return 0;
  CN_CHECK(rowptr && dist_sq && cutoff && deg, "cartnet_neighbor_cap_count: null pointer");
  hipLaunchKernelGGL(cn_cap_count_kernel, dim3(cn_ceil_div(N, 4)), dim3(256), 0, ST(stream), rowptr, dist_sq, N,
                     max_neighbors, tolerance, cutoff, deg);
  CN_LAUNCH_CHECK("cartnet_neighbor_cap_count");
  return 0;
}

extern "C" int cartnet_neighbor_cap_fill(const int64_t* rowptr, const int64_t* rowptr_out, const float* cutoff,
                                         const float* dist_sq, const int64_t* edge_index, const float* cart_dist,
                                         const float* cart_dir, int32_t N, int64_t E, int64_t E_out,
                                         int64_t* edge_index_out, float* cart_dist_out, float* cart_dir_out,
                                         void* stream) {
  CN_CHECK(N >= 0 && E >= 0 && E_out >= 0 && E_out <= E, "cartnet_neighbor_cap_fill: bad sizes");
  if (N == 0 || E_out == 0) return 0;
  CN_CHECK(rowptr && rowptr_out && cutoff && dist_sq && edge_index && cart_dist && cart_dir && edge_index_out &&
               cart_dist_out && cart_dir_out,
           "cartnet_neighbor_cap_fill: null pointer");
  hipLaunchKernelGGL(cn_cap_fill_kernel, dim3(cn_ceil_div(N, 4)), dim3(256), 0, ST(stream), rowptr, rowptr_out, cutoff,
                     dist_sq, edge_index, cart_dist, cart_dir, N, (long long)E, (long long)E_out, edge_index_out,
                     cart_dist_out, cart_dir_out);
  CN_LAUNCH_CHECK("cartnet_neighbor_cap_fill");
  return 0;
}
