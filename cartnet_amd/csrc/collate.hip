// Device-side batching from a packed shard resident in HBM (reference: one pickled PyG Data per
// structure loaded by dataset/datasetADP.py:41-42, collated by PyG's DataLoader in 5 worker processes,
// loader/loader.py:114-124, and augmented per sample on the CPU, dataset/datasetADP.py:33-39,76-77).
//
// A shard stores all crystals as flat CSR arrays (cartnet_amd/shard.py).  cartnet_collate gathers the B selected
// crystals into the tensors CartNet.forward reads -- atomic numbers widened to int64, edge indices rebased to the
// batch's atom numbering, `batch` / `ptr` built in place -- with one launch: every block copies 256 consecutive
// output rows of one section (atoms, edges, targets, crystals) and finds the crystal a row belongs to by binary
// search in the [B+1] output offsets.  The optional SO(3) augmentation (y <- R^T y R, cart_dir <- cart_dir R,
// cell <- cell R) and the temperature standardisation are applied on the way through, so an augmented batch costs the
// same single pass: HBM-bound, ~56 B read + ~60 B written per edge.
#include "common.h"

namespace {

__device__ __forceinline__ int find_segment(const int64_t* __restrict__ ptr, int B, int64_t i) {
  int lo = 0, hi = B;            // ptr[lo] <= i < ptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// v' = v R  (row vector times 3x3)
__device__ __forceinline__ void rot_row(const float* v, const float* R, float* o) {
#pragma unroll
  for (int j = 0; j < 3; ++j) o[j] = v[0] * R[j] + v[1] * R[3 + j] + v[2] * R[6 + j];
}

__global__ __launch_bounds__(256) void cn_collate_kernel(CartnetShard s, const int64_t* __restrict__ sel,
                                                         const int64_t* __restrict__ atom_ptr,
                                                         const int64_t* __restrict__ edge_ptr,
                                                         const int64_t* __restrict__ y_ptr, int B, int64_t N,
                                                         int64_t E, int64_t M, const float* __restrict__ rot,
                                                         float t_mean, float t_std, CartnetCollated o, int nbA,
                                                         int nbE, int nbY) {
  int b = blockIdx.x;
  const int tid = threadIdx.x;
  if (b < nbA) {                                                   // ---- atoms
    const int64_t i = (int64_t)b * 256 + tid;
    if (i >= N) return;
    const int g = find_segment(atom_ptr, B, i);
    const int64_t q = s.atom_ptr[sel[g]] + (i - atom_ptr[g]);
    o.x[i] = (int64_t)s.z[q];
    o.batch[i] = g;
    if (o.pos) {
      o.pos[i * 3] = s.pos[q * 3];
      o.pos[i * 3 + 1] = s.pos[q * 3 + 1];
      o.pos[i * 3 + 2] = s.pos[q * 3 + 2];
    }
    if (o.non_h_mask) o.non_h_mask[i] = s.non_h_mask[q];
    return;
  }
  b -= nbA;
  if (b < nbE) {                                                   // ---- edges
    const int64_t e = (int64_t)b * 256 + tid;
    if (e >= E) return;
    const int g = find_segment(edge_ptr, B, e);
    const int64_t q = s.edge_ptr[sel[g]] + (e - edge_ptr[g]);
    const int64_t base = atom_ptr[g];
    o.edge_index[e] = base + s.edge_src[q];
    o.edge_index[E + e] = base + s.edge_tgt[q];
    o.cart_dist[e] = s.cart_dist[q];
    float d[3] = {s.cart_dir[q * 3], s.cart_dir[q * 3 + 1], s.cart_dir[q * 3 + 2]};
    if (rot) {
      float r[3];
      rot_row(d, rot + (size_t)g * 9, r);
      d[0] = r[0]; d[1] = r[1]; d[2] = r[2];
    }
    o.cart_dir[e * 3] = d[0];
    o.cart_dir[e * 3 + 1] = d[1];
    o.cart_dir[e * 3 + 2] = d[2];
    return;
  }
  b -= nbE;
  if (b < nbY) {                                                   // ---- targets
    const int64_t i = (int64_t)b * 256 + tid;
    if (i >= M) return;
    const int g = find_segment(y_ptr, B, i);
    const int64_t q = s.y_ptr[sel[g]] + (i - y_ptr[g]);
    if (s.y_width == 9) {
      float y[9], t[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) y[k] = s.y[q * 9 + k];
      if (rot) {                                                   // R^T (y R)
        const float* R = rot + (size_t)g * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r) rot_row(y + 3 * r, R, t + 3 * r);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) y[r * 3 + c] = R[r] * t[c] + R[3 + r] * t[3 + c] + R[6 + r] * t[6 + c];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) o.y[i * 9 + k] = y[k];
    } else {
      for (int k = 0; k < s.y_width; ++k) o.y[i * s.y_width + k] = s.y[q * s.y_width + k];
    }
    return;
  }
  b -= nbY;
  {                                                                // ---- crystals
    const int g = b * 256 + tid;
    if (g > B) return;
    if (g == B) { o.ptr[B] = N; return; }
    o.ptr[g] = atom_ptr[g];
    const int64_t sg = sel[g];
    if (o.cell) {
      float c[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) c[k] = s.cell[sg * 9 + k];
      if (rot) {
        float t[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) rot_row(c + 3 * r, rot + (size_t)g * 9, t + 3 * r);
#pragma unroll
        for (int k = 0; k < 9; ++k) c[k] = t[k];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) o.cell[(size_t)g * 9 + k] = c[k];
    }
    if (o.temperature) o.temperature[g] = __fdiv_rn(__fsub_rn(s.temperature[sg], t_mean), t_std);
  }
}

}  // namespace

extern "C" int cartnet_collate(const CartnetShard* shard, const int64_t* sel, const int64_t* out_atom_ptr,
                               const int64_t* out_edge_ptr, const int64_t* out_y_ptr, int32_t B, int64_t N, int64_t E,
                               int64_t M, const float* rot, float temp_mean, float temp_std,
                               const CartnetCollated* out, void* stream) {
  CN_CHECK(shard && out, "cartnet_collate: null descriptor");
  CN_CHECK(B >= 1 && N >= 0 && E >= 0 && M >= 0, "cartnet_collate: bad sizes (B=%d)", B);
  CN_CHECK(N < (1LL << 31) * 256 && E < (1LL << 31) * 128, "cartnet_collate: batch too large for one launch");
  CN_CHECK(sel && out_atom_ptr && out_edge_ptr && out_y_ptr, "cartnet_collate: null selection / offsets");
  CN_CHECK(shard->atom_ptr && shard->edge_ptr && shard->y_ptr && shard->z, "cartnet_collate: incomplete shard");
  CN_CHECK(out->x && out->batch && out->ptr, "cartnet_collate: x, batch and ptr outputs are required");
  CN_CHECK(E == 0 || (shard->edge_src && shard->edge_tgt && shard->cart_dist && shard->cart_dir && out->edge_index &&
                      out->cart_dist && out->cart_dir),
           "cartnet_collate: edge arrays missing");
  CN_CHECK(M == 0 || (shard->y && out->y && shard->y_width >= 1), "cartnet_collate: target arrays missing");
  CN_CHECK(!out->pos || shard->pos, "cartnet_collate: pos requested but not in the shard");
  CN_CHECK(!out->non_h_mask || shard->non_h_mask, "cartnet_collate: non_H_mask requested but not in the shard");
  CN_CHECK(!out->cell || shard->cell, "cartnet_collate: cell requested but not in the shard");
  CN_CHECK(!out->temperature || shard->temperature, "cartnet_collate: temperature requested but not in the shard");
  CN_CHECK(temp_std != 0.f, "cartnet_collate: temp_std must be non-zero (1 = no standardisation)");
  const int nbA = (int)((N + 255) / 256), nbE = (int)((E + 255) / 256), nbY = (int)((M + 255) / 256);
  const int nbG = (B + 1 + 255) / 256;
  hipLaunchKernelGGL(cn_collate_kernel, dim3(nbA + nbE + nbY + nbG), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), *shard, sel, out_atom_ptr, out_edge_ptr, out_y_ptr, B, N, E,
                     M, rot, temp_mean, temp_std, *out, nbA, nbE, nbY);
  CN_LAUNCH_CHECK("cartnet_collate");
  return 0;
}
