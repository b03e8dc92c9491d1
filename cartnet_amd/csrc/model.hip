// Whole-network orchestration: CartNet.forward / backward as one host call each (include/cartnet_hip.h,
// "Whole-network entry points").  No kernels here -- this file only sequences the launches of the other
// translation units in the order the reference executes its modules (models/cartnet.py:65-73,142-161,204-274,293-327)
// and carves every intermediate out of one caller-owned workspace.
#include "model_common.h"

namespace {
using namespace cn_model;


struct Work {
  // graph layout
  int *src32, *tgt32, *rowptr, *colptr, *perm, *zperm, *zptr, *zstatus, *idx;
  // encoder
  float *feat, *env, *he_pre, *e0_pre, *e0, *x0, *xa_pre, *xenc;
  int kf, ldf;
  // layers
  float *pre[CARTNET_MAX_LAYERS], *gs[CARTNET_MAX_LAYERS], *mr1[CARTNET_MAX_LAYERS], *aggr[CARTNET_MAX_LAYERS],
      *mr2[CARTNET_MAX_LAYERS], *xl[CARTNET_MAX_LAYERS], *el[CARTNET_MAX_LAYERS];
  // head
  float *hid, *p6;
  // transposed weights [in, out]
  float *edge0T, *edge2T, *atomT, *head0T;
  float *gate0T[CARTNET_MAX_LAYERS], *aggr0T[CARTNET_MAX_LAYERS], *gate2T[CARTNET_MAX_LAYERS],
      *aggr2T[CARTNET_MAX_LAYERS];
  // bf16x3 pre-split weight images (gemm_precision == 1 and D % 256 == 0; else all null)
  char *i_edge0, *i_edge2, *i_atom, *i_edge2_b, *i_atom_b, *i_head0_b;
  char *i_pn[CARTNET_MAX_LAYERS], *i_pre[CARTNET_MAX_LAYERS], *i_gs[CARTNET_MAX_LAYERS], *i_dpre[CARTNET_MAX_LAYERS],
      *i_de[CARTNET_MAX_LAYERS], *i_dx[CARTNET_MAX_LAYERS];
  // forward transients
  float* Pn;
  double *cs, *cq, *ps, *pq, *bnrow;   // bnrow [2D + 2]: the summed row of a sync-BatchNorm exchange
  // Gate BatchNorm-backward sums without a pass over the edges (fp32, one BatchNorm group, no sync-BatchNorm, D = 256):
  // bc[l] [N, 2D] = per-target sums written by the forward gate kernel; gpa / gpb = partial rows of the two sums -- rows
  // [0, tiles_e) from the epilogue of the dE product of the layer above, rows [tiles_e, tiles_e + nparts_n) from the node
  // update's apply pass (include/cartnet_hip.h: cartnet_gate_scatter_fwd_bc).  nullptr = the statistics pass runs.
  float* bc[CARTNET_MAX_LAYERS];
  double *gpa, *gpb;
  // silu(pre) / silu(he_pre), written by the forward GEMMs that activate them (CartnetGemmArgs.a_act_out) for the weight
  // gradients of the second Linears; nullptr = recompute the SiLU in the weight-gradient kernel
  float *act[CARTNET_MAX_LAYERS], *he_act;
  // backward transients
  float *dhid, *head_parts, *head_tot, *dx[2], *de[2], *daggr, *sums1, *sums2, *dPn[2], *dpre[2], *dhe, *dx0, *seg_tmp,
      *slabs, *slabs2, *e0wT, *gate_bnd;
  double *pa, *pb, *pc[2], *pd[2], *cs_misc[4];
  size_t slab_floats;
  int gparts, nparts_n, tiles_e, tiles_n;
  // BatchNorm groups (CartnetModel.bn_group_size > 0 and more than one group in this batch): gparts / nrows_n are then
  // the TOTAL partial-sum rows of the per-edge / per-node statistics kernels (G x parts per group)
  int G, nrows_n;
  int *node_gptr, *edge_gptr;
  CartnetGroups grp;
  const CartnetGroups* groups;   // &grp, or nullptr for one group (set by the entry points, not by carve)
};

// The gate's BatchNorm-backward sums come from the dE epilogue + per-atom sums instead of a statistics pass over gs and
// de_out (544 MB per layer at the benchmark batch): needs a kernel that carries the epilogue (cartnet_gemm_gate_stats_ok:
// fp32 MFMA or bf16x3, D = 256 -- the two K-segments of dE fold into one product --, at least 64 / 96 row tiles), one
// BatchNorm group and per-rank statistics.  Training-mode passes only (the callers check `training`).
inline bool gate_sums_fused(const CartnetModel& m, int G, int tiles_e, long long E) {
#ifdef CN_NO_GATE_FUSE      /* A/B builds only (tools/build_variant.sh): the statistics pass of rounds 1-4 */
  return false;
#endif
  if (!(m.gemm_precision <= 1 && G == 1 && m.half_storage == 0 && m.bn_allreduce == nullptr && E > 0)) return false;
  // ONE predicate for "the dE product takes the kernel with the epilogue" (ADVICE r5): the launch cartnet_model_backward
  // will build (same shapes, strides and flags; placeholder pointers with the alignment the workspace guarantees) is put
  // to gemm.hip's own cartnet_gemm_gate_stats_ok -- forward commits to the per-target sums here, backward checks again.
  (void)tiles_e;
  const int D = m.D;
  float* const ph = reinterpret_cast<float*>(uintptr_t(1) << 20);
  CartnetGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.M = (int)E; a.N = D; a.K = D; a.lda = 2 * D; a.ldb = 3 * D; a.ldc = D; a.ldr = D;
  a.ngroups = 1; a.nsegs = 2; a.splitk = 1; a.b_kstrided = 1; a.precision = m.gemm_precision;
  a.A[0] = ph; a.A[1] = ph + D; a.B[0] = ph; a.B[1] = ph; a.C[0] = ph; a.resid[0] = ph;
  a.b_split_folded = ph;
  a.gst_g = ph; a.gst_ld = 2 * D; a.gst_env = ph; a.gst_mean_rstd = ph; a.gst_gamma = ph; a.gst_beta = ph;
  a.colsum[0] = reinterpret_cast<double*>(ph); a.colsq[0] = reinterpret_cast<double*>(ph);
  return E < (1ll << 31) && cartnet_gemm_gate_stats_ok(&a) == 1;
}

Work carve(const CartnetModel& m, int N, long long E, int Bg, int M, bool need_bwd, char* base, size_t* total) {
  Work w;
  memset(&w, 0, sizeof(w));
  Carver c{base};
  const int D = m.D, L = m.L, H = m.D / 2;
  const size_t En = (size_t)(E > 0 ? E : 1), Nn = (size_t)(N > 0 ? N : 1);
  w.kf = m.invariant ? m.R : m.R + 3;
  w.ldf = (w.kf + 15) / 16 * 16;   // K of the first edge Linear padded to whole K-steps (pad columns are zero)
  w.gparts = cartnet_gate_scatter_nparts(N);
  w.nparts_n = cartnet_node_nparts(N);
  w.nrows_n = w.nparts_n;
  w.G = (m.bn_group_size > 0 && Bg > m.bn_group_size) ? (Bg + m.bn_group_size - 1) / m.bn_group_size : 1;
  if (w.G > 1) {
    const int per = (int)((Nn + (size_t)4 * w.G - 1) / ((size_t)4 * w.G));    // ceil(N / G / nodes-per-workgroup)
    w.grp.G = w.G;
    w.grp.edge_parts = per < 1 ? 1 : (per > 1024 ? 1024 : per);
    w.grp.node_parts = per < 1 ? 1 : (per > 256 ? 256 : per);
    w.gparts = w.G * w.grp.edge_parts;
    w.nrows_n = w.G * w.grp.node_parts;
  }
  const size_t Gn = (size_t)w.G;
  w.node_gptr = c.take<int>(Gn + 1);
  w.edge_gptr = c.take<int>(Gn + 1);
  w.grp.node_gptr = w.node_gptr;
  w.grp.edge_gptr = w.edge_gptr;
  w.tiles_e = tiles_m(E);
  w.tiles_n = tiles_m(N);
  w.src32 = c.take<int>(En);
  w.tgt32 = c.take<int>(En);
  w.rowptr = c.take<int>(Nn + 1);
  w.colptr = c.take<int>(Nn + 1);
  w.perm = c.take<int>(En);
  w.zperm = c.take<int>(Nn);
  w.zptr = c.take<int>((size_t)m.n_types + 2);
  w.zstatus = c.take<int>(4);
  w.idx = c.take<int>(Nn);
  w.feat = c.take<float>(En * w.ldf);
  w.env = c.take<float>(En);
  w.he_pre = (m.half_storage != 0 && m.gemm_precision == 2) ? reinterpret_cast<float*>(c.take<uint16_t>(En * 2 * D))
                                                              : c.take<float>(En * 2 * D);   // bf16 under half storage
  w.e0_pre = c.take<float>(En * D);
  w.e0 = c.take<float>(En * D);
  w.x0 = c.take<float>(Nn * 2 * D);
  w.xa_pre = c.take<float>(Nn * D);
  w.xenc = c.take<float>(Nn * D);
  float* xping[2] = {nullptr, nullptr};
  float* eping[2] = {nullptr, nullptr};
  if (!need_bwd) w.gate_bnd = c.take<float>(cartnet_gate_gemm_eval_workspace((int64_t)En, D) / sizeof(float) + 4);   // inference fusion
  if (!need_bwd) {
    xping[0] = c.take<float>(Nn * D);
    xping[1] = c.take<float>(Nn * D);
    eping[0] = c.take<float>(En * D);
    eping[1] = c.take<float>(En * D);
  }
  // half storage (CartnetModel.half_storage at precision 2): pre / gs / dpre hold bf16 elements -- half the bytes, same
  // element counts and leading dimensions; the float* members then carry bf16 pointers (see hcol below)
  const bool half = m.half_storage != 0 && m.gemm_precision == 2;
  auto take_edge2d = [&]() -> float* {
    return half ? reinterpret_cast<float*>(c.take<uint16_t>(En * 2 * D)) : c.take<float>(En * 2 * D);
  };
  for (int l = 0; l < L; ++l) {
    if (need_bwd || l == 0) {
      w.pre[l] = take_edge2d();
      w.gs[l] = take_edge2d();
      w.aggr[l] = c.take<float>(Nn * D);
    } else {
      w.pre[l] = w.pre[0];
      w.gs[l] = w.gs[0];
      w.aggr[l] = w.aggr[0];
    }
    w.mr1[l] = c.take<float>(Gn * 2 * D);
    w.mr2[l] = c.take<float>(Gn * 2 * D);
    if (l < L - 1) {   // the last layer writes straight into the caller's x_out / e_out
      w.xl[l] = need_bwd ? c.take<float>(Nn * D) : xping[l & 1];
      w.el[l] = need_bwd ? c.take<float>(En * D) : eping[l & 1];
    }
  }
  // fp32 training: the activated operands are kept (1 extra [E, 2D] matrix per layer + one for the encoder) so that the
  // weight gradients dW2 = dY^T silu(pre) read a plain operand (step 15.26 vs 15.43 ms, same box, interleaved, round 2).  Not at precision 1 / 2: there the step got SLOWER (11.74 vs 11.56 ms) -- the bf16 step is
  // power- and HBM-limited and 1.8 GB of extra writes cost more than the cheaper weight-gradient kernel returns.
  // (Re-measured at bf16x3 in round 3: 11.04-11.09 vs 10.87-10.96 ms, slower again.)
  if (need_bwd && m.gemm_precision == 0 && D % 256 == 0) {
    // (Round 6: with SiLU applied in place to the weight-gradient kernel's DMA'd tile -- cn_gemm_f32tn_kernel<true, 5> --
    //  dropping the layers' kept silu(pre) saves 1.45 GB of writes per step and costs 0.09 ms: the product 346 -> 380 us,
    //  the second Linears 405 -> 378; 13.23-13.25 kept vs 13.32-13.35 ms dropped, same box.  Kept; -DCN_DROP_LAYER_ACT drops.)
#ifndef CN_DROP_LAYER_ACT
    for (int l = 0; l < L; ++l) w.act[l] = c.take<float>(En * 2 * D);
#endif
    w.he_act = c.take<float>(En * 2 * D);
  }
  if (need_bwd && gate_sums_fused(m, w.G, w.tiles_e, E)) {
    for (int l = 0; l < L; ++l) w.bc[l] = c.take<float>(Nn * 2 * D);
    w.gpa = c.take<double>((size_t)(w.tiles_e + w.nparts_n) * D);
    w.gpb = c.take<double>((size_t)(w.tiles_e + w.nparts_n) * D);
  }
  w.hid = c.take<float>(Nn * H);
  w.p6 = c.take<float>((size_t)(M > 0 ? M : 1) * 6);
  w.edge0T = c.take<float>((size_t)w.ldf * 2 * D);   // [ldf, 2D]: rows kf.. are zero
  w.edge2T = c.take<float>((size_t)2 * D * D);
  w.atomT = c.take<float>((size_t)2 * D * D);
  w.head0T = c.take<float>((size_t)D * H);
  for (int l = 0; l < L; ++l) {
    w.gate0T[l] = c.take<float>((size_t)3 * D * D);
    w.aggr0T[l] = c.take<float>((size_t)3 * D * D);
    w.gate2T[l] = c.take<float>((size_t)D * D);
    w.aggr2T[l] = c.take<float>((size_t)D * D);
  }
  if (D % 256 == 0) {   // pre-arranged weight images (bf16 planes at precision 1 / 2, fp32 rows at precision 0; sized for the larger)
    const size_t blk = cartnet_gemm_split_b_bytes(D, D);          // one D x D block
    w.i_edge0 = c.take<char>(cartnet_gemm_split_b_bytes(w.ldf, 2 * D));
    w.i_edge2 = c.take<char>(cartnet_gemm_split_b_bytes(2 * D, D));
    w.i_atom = c.take<char>(cartnet_gemm_split_b_bytes(2 * D, D));
    for (int l = 0; l < L; ++l) {
      w.i_pn[l] = c.take<char>(4 * blk);     // gate_i, aggr_i, gate_j, aggr_j
      w.i_pre[l] = c.take<char>(2 * blk);    // gate_e, aggr_e
      w.i_gs[l] = c.take<char>(2 * blk);     // gate2, aggr2
    }
    if (need_bwd) {
      w.i_edge2_b = c.take<char>(cartnet_gemm_split_b_bytes(D, 2 * D));
      w.i_atom_b = c.take<char>(cartnet_gemm_split_b_bytes(D, 2 * D));
      if (H % 16 == 0) w.i_head0_b = c.take<char>(cartnet_gemm_split_b_bytes(H, D));
      for (int l = 0; l < L; ++l) {
        w.i_dpre[l] = c.take<char>(2 * blk);   // gate2, aggr2 as [out, in]
        w.i_de[l] = c.take<char>(2 * blk);     // folded: gate0[:, 2D:], aggr0[:, 2D:]
        w.i_dx[l] = c.take<char>(4 * blk);     // folded: gate0[:, :D], aggr0[:, :D], gate0[:, D:2D], aggr0[:, D:2D]
      }
    }
  }
  w.Pn = c.take<float>(Nn * 4 * D);
  const size_t big = (size_t)(w.tiles_e > w.gparts ? w.tiles_e : w.gparts);
  w.cs = c.take<double>(big * D);
  w.cq = c.take<double>(big * D);
  w.ps = c.take<double>((size_t)w.gparts * D);
  w.pq = c.take<double>((size_t)w.gparts * D);
  w.bnrow = c.take<double>((size_t)2 * D + 2);
  if (need_bwd) {
    w.dhid = c.take<float>(Nn * H);
    w.head_parts = c.take<float>((size_t)w.nparts_n * (7 * H + 8));
    w.head_tot = c.take<float>((size_t)7 * H + 8);
    w.dx[0] = c.take<float>(Nn * D);
    w.dx[1] = c.take<float>(Nn * D);
    w.de[0] = c.take<float>(En * D);
    w.de[1] = c.take<float>(En * D);
    w.daggr = c.take<float>(Nn * D);
    w.sums1 = c.take<float>(Gn * 2 * D);
    w.sums2 = c.take<float>(Gn * 2 * D);
    for (int i = 0; i < 2; ++i) {   // double-buffered: the weight-gradient stream reads them while the next layer runs
      w.dPn[i] = c.take<float>(Nn * 4 * D);
      w.dpre[i] = take_edge2d();
    }
    w.dhe = c.take<float>(En * 2 * D);
    w.dx0 = c.take<float>(Nn * 2 * D);
    w.seg_tmp = c.take<float>(Nn * 2 * D);
    const size_t pmax = (size_t)(w.gparts > w.nrows_n ? w.gparts : w.nrows_n) * 2 * D;
    w.pa = c.take<double>(pmax);
    w.pb = c.take<double>(pmax);
    const size_t tmax = (size_t)(w.tiles_e > w.tiles_n ? w.tiles_e : w.tiles_n);
    for (int i = 0; i < 2; ++i) {
      w.pc[i] = c.take<double>(pmax);
      w.pd[i] = c.take<double>(pmax);
    }
    for (int i = 0; i < 4; ++i) w.cs_misc[i] = c.take<double>(tmax * 2 * D);
    size_t sl = 0;
    auto mx = [&](size_t v) { if (v > sl) sl = v; };
    mx(wgrad_slab_floats(1, N, H, D));
    mx(wgrad_slab_floats(2, E, D, D));
    mx(wgrad_slab_floats(4, N, D, D));
    mx(wgrad_slab_floats(1, E, D, 2 * D));
    mx(wgrad_slab_floats(1, E, 2 * D, w.kf));
    mx(wgrad_slab_floats(1, E, w.ldf, 2 * D));
    w.e0wT = c.take<float>((size_t)w.ldf * 2 * D);
    mx(wgrad_slab_floats(1, N, D, 2 * D));
    w.slab_floats = sl;
    w.slabs = c.take<float>(sl > 0 ? sl : 1);
    // the encoder's first Linear reduces its weight gradient on the main stream while the side stream uses `slabs`
    size_t sl2 = wgrad_slab_floats(1, E, 2 * D, w.kf);
    const size_t sl2b = wgrad_slab_floats(1, E, w.ldf, 2 * D);
    if (sl2b > sl2) sl2 = sl2b;
    w.slabs2 = c.take<float>(sl2 > 0 ? sl2 : 1);
  }
  *total = align_up(c.off);
  return w;
}

// bytes of one D x D weight image at the model's GEMM precision (images of a folded operand sit back to back)
inline size_t img_blk(const CartnetModel& m) {
  return m.gemm_precision == 0 ? cartnet_gemm_pack_b_bytes(m.D, m.D) : cartnet_gemm_split_b_bytes(m.D, m.D);
}

thread_local int g_precision = 0;   // set per call from CartnetModel.gemm_precision

inline CartnetGemmArgs gemm_args(int M, int N, int K, int lda, int ldb, int ldc) {
  CartnetGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.precision = g_precision;
  a.M = M; a.N = N; a.K = K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.ngroups = 1; a.nsegs = 1; a.splitk = 1;
  return a;
}

// the fp32 forms under the _h forms' signatures (void* for the tensor that changes its element type)
inline int cartnet_gate_scatter_fwd_f(const void* gs, const float* e_in, const float* env, const int32_t* rowptr,
                                      const float* mr, const float* ga, const float* be, int32_t N, int32_t D, float* e_out,
                                      float* aggr, double* ps, double* pq, const CartnetGroups* g, void* st) {
  return cartnet_gate_scatter_fwd(static_cast<const float*>(gs), e_in, env, rowptr, mr, ga, be, N, D, e_out, aggr, ps, pq, g, st);
}
inline int cartnet_gate_scatter_bwd_stats_f(const void* gs, const float* de, const float* da, const float* env,
                                            const int32_t* rowptr, const float* mr, const float* ga, const float* be,
                                            int32_t N, int32_t D, double* pa, double* pb, const CartnetGroups* g, void* st) {
  return cartnet_gate_scatter_bwd_stats(static_cast<const float*>(gs), de, da, env, rowptr, mr, ga, be, N, D, pa, pb, g, st);
}
inline int cartnet_gate_scatter_bwd_apply_f(void* gs, const float* de, const float* da, const float* env, const int32_t* rowptr,
                                            const float* mr, const float* ga, const float* be, const float* sums, int64_t E,
                                            int32_t training, int32_t N, int32_t D, double* pdg, double* pds,
                                            const CartnetGroups* g, void* st) {
  return cartnet_gate_scatter_bwd_apply(static_cast<float*>(gs), de, da, env, rowptr, mr, ga, be, sums, E, training, N, D, pdg,
                                        pds, g, st);
}
inline int cartnet_segment_sum_f(const void* rows, int32_t ld, const int32_t* ptr, const int32_t* perm, int32_t N, int32_t W,
                                 float* out, int32_t ldo, void* st) {
  return cartnet_segment_sum(static_cast<const float*>(rows), ld, ptr, perm, N, W, out, ldo, st);
}

// column `elems` of a row-major matrix whose elements are fp32 or (half storage) bf16
inline float* hcol(float* p, size_t elems, bool half) {
  return half ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(p) + elems) : p + elems;
}
inline const float* hcol(const float* p, size_t elems, bool half) { return hcol(const_cast<float*>(p), elems, half); }

// outs[g] = dY[g]^T @ (silu?)(X[g]): reduction over `K` rows, split over workgroups, slabs summed in fixed order.
int wgrad(const float* const* dY, int ldy, const float* const* X, int ldx, float* const* outs, int ldo, long long K,
          int M, int N, int groups, bool b_act, const Work& w, void* st, bool dy_half = false, bool x_half = false) {
  if (K <= 0) {   // no rows: the gradient is zero
    for (int g = 0; g < groups; ++g)
      for (int r = 0; r < M; ++r)
        if (hipMemsetAsync(outs[g] + (size_t)r * ldo, 0, sizeof(float) * N, (hipStream_t)st) != hipSuccess) return 2;
    return 0;
  }
  const int S = split_k(K, wgrad_tiles(groups, M, N));
  CartnetGemmArgs a = gemm_args(M, N, (int)K, ldy, ldx, S > 1 ? N : ldo);
  a.ngroups = groups;
  a.a_kstrided = 1;
  a.b_kstrided = 1;
  a.b_act = b_act ? 1 : 0;
  a.a_half = dy_half ? 1 : 0;
  a.b_half = x_half ? 1 : 0;
  a.splitk = S;
  const float* slabp[CARTNET_MAX_GROUPS];
  for (int g = 0; g < groups; ++g) {
    a.A[g] = dY[g];
    a.B[g] = X[g];
    a.C[g] = S > 1 ? w.slabs + (size_t)g * S * M * N : outs[g];
    slabp[g] = a.C[g];
  }
  RUN(cartnet_gemm(&a, st));
  if (S > 1) RUN(cartnet_splitk_reduce(slabp, outs, groups, S, M, N, ldo, st));
  return 0;
}

int check_model(const CartnetModel* m, const CartnetBatch* b, const char* who) {
  CN_CHECK(m && b, "%s: null model/batch", who);
  CN_CHECK(m->D >= 8 && m->D % 8 == 0 && m->D / 2 <= 512, "%s: dim_in=%d must be a multiple of 8, <= 1024", who, m->D);
  CN_CHECK(m->L >= 1 && m->L <= CARTNET_MAX_LAYERS, "%s: num_layers=%d out of range (1..%d)", who, m->L,
           CARTNET_MAX_LAYERS);
  CN_CHECK(m->R >= 1, "%s: dim_rbf=%d", who, m->R);
  CN_CHECK(m->gemm_precision >= 0 && m->gemm_precision <= 2, "%s: gemm_precision=%d", who, m->gemm_precision);
  CN_CHECK(b->N >= 0 && b->E >= 0 && b->Bg >= 1 && b->M >= 0, "%s: bad batch sizes", who);
  CN_CHECK(b->E < 2147483647LL, "%s: E does not fit int32", who);
  CN_CHECK((long long)b->E * 2 * m->D < 2147483647LL * 4, "%s: batch too large for 32-bit tile indexing", who);
  CN_CHECK(b->z && b->batch && b->graph_ptr && (b->edge_index || b->E == 0) && (b->cart_dist || b->E == 0),
           "%s: null batch pointer", who);
  CN_CHECK(m->invariant || b->cart_dir || b->E == 0, "%s: cart_dir missing", who);
  CN_CHECK(!m->use_temperature || b->temperature, "%s: temperature missing", who);
  CN_CHECK(!m->cholesky || b->non_h_mask, "%s: non_H_mask missing", who);
  const CartnetParams& p = m->p;
  const bool plain = !m->use_temperature && !m->atom_types;   // one learned row for every atom, no atom MLP (cartnet.py:150-151)
  CN_CHECK(m->rbf_means && m->rbf_betas && (plain || (p.atom_w && p.atom_b)) && p.edge0_w && p.edge0_b && p.edge2_w &&
               p.edge2_b && p.head0_w && p.head0_b && p.head2_w && p.head2_b,
           "%s: null parameter", who);
  CN_CHECK(!(m->atom_types || plain) || p.embedding, "%s: embedding missing", who);
  CN_CHECK(!m->use_temperature || (p.temp_w && p.temp_b), "%s: temperature projection missing", who);
  CN_CHECK(m->use_temperature || !m->atom_types || p.enc_bias, "%s: encoder.bias missing", who);
  for (int l = 0; l < m->L; ++l) {
    const CartnetLayerParams& q = p.layer[l];
    CN_CHECK(q.gate0_w && q.gate0_b && q.gate2_w && q.gate2_b && q.aggr0_w && q.aggr0_b && q.aggr2_w && q.aggr2_b &&
                 q.norm_w && q.norm_b && q.norm2_w && q.norm2_b,
             "%s: null parameter in layer %d", who, l);
    CN_CHECK(m->buf[l].norm_mean && m->buf[l].norm_var && m->buf[l].norm2_mean && m->buf[l].norm2_var,
             "%s: null BatchNorm buffer in layer %d", who, l);
  }
  return 0;
}

}  // namespace

namespace {
thread_local EventPool g_events;
}  // namespace

extern "C" size_t cartnet_workspace_bytes(const CartnetModel* model, int32_t N, int64_t E, int32_t Bg, int32_t M,
                                          int32_t need_backward) {
  if (!model || model->L < 1 || model->L > CARTNET_MAX_LAYERS) return 0;
  size_t total = 0;
  carve(*model, N, E, Bg, M, need_backward != 0, nullptr, &total);
  return total;
}

extern "C" int cartnet_model_forward(const CartnetModel* model, const CartnetBatch* batch, void* workspace,
                                     size_t workspace_bytes, int32_t training, int32_t need_backward, float* pred,
                                     float* x_out, float* e_out, int32_t* status, void* st, void* aux_stream) {
  RUN(check_model(model, batch, "cartnet_model_forward"));
  g_precision = model->gemm_precision;
  const CartnetModel& m = *model;
  const CartnetBatch& b = *batch;
  const CartnetParams& P = m.p;
  CN_CHECK(workspace && x_out && status && (e_out || b.E == 0) && (pred || (m.cholesky ? b.M == 0 : b.Bg == 0)),
           "cartnet_model_forward: null output/workspace");
  CN_CHECK((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "cartnet_model_forward: workspace must be 256-byte aligned");
  size_t need = 0;
  Work w = carve(m, b.N, b.E, b.Bg, b.M, need_backward != 0, static_cast<char*>(workspace), &need);
  CN_CHECK(workspace_bytes >= need, "cartnet_model_forward: workspace %zu < required %zu bytes", workspace_bytes, need);
  w.groups = w.G > 1 ? &w.grp : nullptr;
  const int D = m.D, L = m.L, H = D / 2, N = b.N;
  const int E = (int)b.E;
  const bool plain = !m.use_temperature && !m.atom_types;

  // Second stream (optional): what does not depend on the graph -- the weight transposes and images -- and the atom
  // branch of the encoder with the first layer's node terms (few row tiles: they leave most of the chip idle) run on it
  // next to the layout build and the edge encoder; the main stream waits for them where it needs them and the two
  // streams are joined before the call returns.
  g_events.next = 0;
  Streams S{(hipStream_t)st, aux_stream ? (hipStream_t)aux_stream : (hipStream_t)st, aux_stream != nullptr && aux_stream != st,
            &g_events};
  void* sw = (void*)S.side;
#define FORK() do { if (S.fork() != 0) { cartnet_set_error("cartnet_model_forward: stream fork failed"); return 2; } } while (0)
#define MAIN_WAITS(ev) do { if (S.main_waits(ev) != 0) { cartnet_set_error("cartnet_model_forward: stream wait failed"); return 2; } } while (0)
  FORK();      // the side stream starts after whatever the caller queued (the optimiser step that wrote the weights)
  // (the CSC permutation -- one workgroup per crystal, only backward reads it -- goes to the second stream when there is one)
  // (an inference pass -- need_backward = 0 -- does not build it at all)
  const bool want_csc = need_backward != 0, csc_aside = S.dual && want_csc;
  RUN(cartnet_csr_build(b.edge_index, b.E, N, b.graph_ptr, b.Bg, w.src32, w.tgt32, w.rowptr,
                        (want_csc && !csc_aside) ? w.colptr : nullptr, (want_csc && !csc_aside) ? w.perm : nullptr, status, st));
  if (w.groups) RUN(cartnet_group_ptrs(b.graph_ptr, b.Bg, m.bn_group_size, w.rowptr, w.G, w.node_gptr, w.edge_gptr, st));
  // weights -> [in, out]   (side stream)
  {
    constexpr int TB = 40;   // cartnet_transpose takes up to 40 matrices per launch
    const float* src[TB]; float* dst[TB]; int32_t rows[TB], cols[TB], lds[TB], ldd[TB];
    int n = 0;
    auto flush = [&]() -> int {
      if (n == 0) return 0;
      int rc = cartnet_transpose(src, dst, rows, cols, lds, ldd, n, sw);
      n = 0;
      return rc;
    };
    auto add = [&](const float* s, float* d, int r, int c) -> int {
      src[n] = s; dst[n] = d; rows[n] = r; cols[n] = c; lds[n] = c; ldd[n] = r;
      if (++n == TB) return flush();
      return 0;
    };
    if (w.ldf > w.kf &&
        hipMemsetAsync(w.edge0T + (size_t)w.kf * 2 * D, 0, sizeof(float) * (size_t)(w.ldf - w.kf) * 2 * D,
                       S.side) != hipSuccess) {
      cartnet_set_error("cartnet_model_forward: memset failed");
      return 2;
    }
    RUN(add(P.edge0_w, w.edge0T, 2 * D, w.kf));
    RUN(add(P.edge2_w, w.edge2T, D, 2 * D));
    if (!plain) RUN(add(P.atom_w, w.atomT, D, 2 * D));
    RUN(add(P.head0_w, w.head0T, H, D));
    for (int l = 0; l < L; ++l) {
      RUN(add(P.layer[l].gate0_w, w.gate0T[l], D, 3 * D));
      RUN(add(P.layer[l].aggr0_w, w.aggr0T[l], D, 3 * D));
      RUN(add(P.layer[l].gate2_w, w.gate2T[l], D, D));
      RUN(add(P.layer[l].aggr2_w, w.aggr2T[l], D, D));
    }
    RUN(flush());
  }

  // bf16x3: split every weight operand once (forward operands B = W^T: stride_k 1, stride_n ld; backward operands
  // B = W: stride_k ld, stride_n 1)
  if (w.i_edge2) {
    std::vector<const float*> src; std::vector<void*> dst; std::vector<int32_t> Ks, Ns, sk, sn;
    const size_t blk = img_blk(m);
    auto fwd = [&](const float* W, int ld, int K_, int N_, char* img) {
      src.push_back(W); dst.push_back(img); Ks.push_back(K_); Ns.push_back(N_); sk.push_back(1); sn.push_back(ld);
    };
    auto bwd = [&](const float* W, int ld, int K_, int N_, char* img) {
      src.push_back(W); dst.push_back(img); Ks.push_back(K_); Ns.push_back(N_); sk.push_back(ld); sn.push_back(1);
    };
    bwd(w.edge0T, 2 * D, w.ldf, 2 * D, w.i_edge0);     // the zero-padded transposed copy is already [K, N]
    fwd(P.edge2_w, 2 * D, 2 * D, D, w.i_edge2);
    if (!plain) fwd(P.atom_w, 2 * D, 2 * D, D, w.i_atom);
    for (int l = 0; l < L; ++l) {
      const CartnetLayerParams& q = P.layer[l];
      fwd(q.gate0_w, 3 * D, D, D, w.i_pn[l]);
      fwd(q.aggr0_w, 3 * D, D, D, w.i_pn[l] + blk);
      fwd(q.gate0_w + D, 3 * D, D, D, w.i_pn[l] + 2 * blk);
      fwd(q.aggr0_w + D, 3 * D, D, D, w.i_pn[l] + 3 * blk);
      fwd(q.gate0_w + 2 * D, 3 * D, D, D, w.i_pre[l]);
      fwd(q.aggr0_w + 2 * D, 3 * D, D, D, w.i_pre[l] + blk);
      fwd(q.gate2_w, D, D, D, w.i_gs[l]);
      fwd(q.aggr2_w, D, D, D, w.i_gs[l] + blk);
    }
    if (need_backward) {
      bwd(P.edge2_w, 2 * D, D, 2 * D, w.i_edge2_b);
      if (!plain) bwd(P.atom_w, 2 * D, D, 2 * D, w.i_atom_b);
      if (w.i_head0_b) bwd(P.head0_w, D, H, D, w.i_head0_b);
      for (int l = 0; l < L; ++l) {
        const CartnetLayerParams& q = P.layer[l];
        bwd(q.gate2_w, D, D, D, w.i_dpre[l]);
        bwd(q.aggr2_w, D, D, D, w.i_dpre[l] + blk);
        bwd(q.gate0_w + 2 * D, 3 * D, D, D, w.i_de[l]);
        bwd(q.aggr0_w + 2 * D, 3 * D, D, D, w.i_de[l] + blk);
        bwd(q.gate0_w, 3 * D, D, D, w.i_dx[l]);
        bwd(q.aggr0_w, 3 * D, D, D, w.i_dx[l] + blk);
        bwd(q.gate0_w + D, 3 * D, D, D, w.i_dx[l] + 2 * blk);
        bwd(q.aggr0_w + D, 3 * D, D, D, w.i_dx[l] + 3 * blk);
      }
    }
    if (m.gemm_precision == 0)
      RUN(cartnet_gemm_pack_b(src.data(), dst.data(), Ks.data(), Ns.data(), sk.data(), sn.data(), (int32_t)src.size(), sw));
    else
      RUN(cartnet_gemm_split_b(src.data(), dst.data(), Ks.data(), Ns.data(), sk.data(), sn.data(), (int32_t)src.size(), sw));
  }
  hipEvent_t weights_ready = S.mark_side();
  FORK();      // the atom branch and the CSC build below write status bits: after the layout build has reset the word
  if (m.cholesky) RUN(cartnet_mask_index(b.non_h_mask, N, w.idx, nullptr, sw));   // graph-only: off the head's critical path
  if (csc_aside)
    RUN(cartnet_csc_build(w.src32, w.rowptr, b.graph_ptr, b.Bg, N, b.E, w.colptr, w.perm, status, sw));

  const bool half = m.half_storage != 0 && m.gemm_precision == 2;
  CN_CHECK(!m.half_storage || (m.gemm_precision == 2 && !w.groups && D % 256 == 0),
           "cartnet_model_forward: half_storage needs gemm_precision 2, D %% 256 == 0 and no BatchNorm groups");
  // ---- encoder, edges (cartnet.py:159)
  RUN(cartnet_edge_features(b.cart_dist, b.cart_dir, m.rbf_means, m.rbf_betas, b.E, m.R, m.invariant, m.radius,
                            m.env_radius, w.feat, w.ldf, w.env, st));
  MAIN_WAITS(weights_ready);
  {
    CartnetGemmArgs a = gemm_args(E, 2 * D, w.ldf, w.ldf, 2 * D, 2 * D);
    a.A[0] = w.feat; a.B[0] = w.edge0T; a.C[0] = w.he_pre; a.bias[0] = P.edge0_b; a.b_kstrided = 1;
    a.b_split[0] = w.i_edge0; a.c_half = half;
    RUN(cartnet_gemm(&a, st));
  }
  {
    CartnetGemmArgs a = gemm_args(E, D, 2 * D, 2 * D, D, D);
    a.A[0] = w.he_pre; a.B[0] = w.edge2T; a.C[0] = w.e0; a.cpre[0] = w.e0_pre; a.bias[0] = P.edge2_b;
    a.b_kstrided = 1; a.a_act = 1; a.out_act = 1; a.b_split[0] = w.i_edge2; a.a_half = half;
    if (w.he_act) a.a_act_out[0] = w.he_act;
    RUN(cartnet_gemm(&a, st));
  }
  // ---- encoder, atoms (cartnet.py:145-154); out-of-table atomic numbers / batch ids are clamped and reported in `status`
  // (side stream; in stream order behind the weight images it reads)
  if (plain) {   // cartnet.py:150-151: the single row of Embedding(1, D) for every atom, no atom MLP
    RUN(cartnet_node_embed(nullptr, nullptr, nullptr, P.embedding, nullptr, nullptr, nullptr, N, D, 1, b.Bg, status,
                           w.xenc, sw));
  } else {
    RUN(cartnet_node_embed(m.atom_types ? b.z : nullptr, m.use_temperature ? b.batch : nullptr,
                           m.use_temperature ? b.temperature : nullptr, m.atom_types ? P.embedding : nullptr,
                           m.use_temperature ? P.temp_w : nullptr, m.use_temperature ? P.temp_b : nullptr,
                           m.use_temperature ? nullptr : P.enc_bias, N, 2 * D, m.n_types, b.Bg, status, w.x0, sw));
    CartnetGemmArgs a = gemm_args(N, D, 2 * D, 2 * D, D, D);
    a.A[0] = w.x0; a.B[0] = w.atomT; a.C[0] = w.xenc; a.cpre[0] = w.xa_pre; a.bias[0] = P.atom_b;
    a.b_kstrided = 1; a.a_act = 1; a.out_act = 1; a.b_split[0] = w.i_atom;
    RUN(cartnet_gemm(&a, sw));
  }
  // node-side halves of a layer's first Linears: Pn = [gate_i | aggr_i | gate_j | aggr_j]
  auto node_terms = [&](int l, const float* x, void* s_) -> int {
    const CartnetLayerParams& q = P.layer[l];
    CartnetGemmArgs a = gemm_args(N, D, D, D, D, 4 * D);
    a.ngroups = 4; a.b_kstrided = 1;
    const float* Bt[4] = {w.gate0T[l], w.aggr0T[l], w.gate0T[l] + (size_t)D * D, w.aggr0T[l] + (size_t)D * D};
    const size_t blk = img_blk(m);
    for (int g = 0; g < 4; ++g) {
      a.A[g] = x; a.B[g] = Bt[g]; a.C[g] = w.Pn + (size_t)g * D;
      if (w.i_pn[l]) a.b_split[g] = w.i_pn[l] + g * blk;
    }
    a.bias[0] = q.gate0_b; a.bias[1] = q.aggr0_b;
    return cartnet_gemm(&a, s_);
  };
  RUN(node_terms(0, w.xenc, sw));
  hipEvent_t atoms_ready = S.mark_side();

  // BatchNorm statistics from partial column sums; with CartnetModel.bn_allreduce (sync-BatchNorm) the sums of all ranks
  const bool sync_bn = training && m.bn_allreduce != nullptr;
  CN_CHECK(!(sync_bn && w.groups), "cartnet_model_forward: sync-BatchNorm and BatchNorm groups are mutually exclusive");
  auto bn_stats = [&](double* psum, double* psq, int nparts, long long count, float* rmean, float* rvar, int64_t* nbt,
                      float* mean_rstd, int over_edges) -> int {
    if (!sync_bn)
      return cartnet_bn_finalize(psum, psq, nparts, count, D, m.bn_eps, m.bn_momentum, training, rmean, rvar, nbt, mean_rstd,
                                 w.groups, 1, over_edges, st);
    RUN(cartnet_bn_sync_gather(psum, psq, nparts, D, count, w.bnrow, nullptr, nullptr, st));
    CN_CHECK(m.bn_allreduce(m.bn_allreduce_user, w.bnrow, 2 * (int64_t)D + 1, st) == 0,
             "cartnet_model_forward: the sync-BatchNorm all-reduce callback failed");
    return cartnet_bn_finalize_row(w.bnrow, D, m.bn_eps, m.bn_momentum, rmean, rvar, nbt, mean_rstd, st);
  };

  // ---- message-passing layers (cartnet.py:204-274)
  const float* x = w.xenc;
  const float* e = w.e0;
  for (int l = 0; l < L; ++l) {
    const CartnetLayerParams& q = P.layer[l];
    float* x_next = (l == L - 1) ? x_out : w.xl[l];
    float* e_next = (l == L - 1) ? e_out : w.el[l];
    if (l == 0) MAIN_WAITS(atoms_ready);      // x, Pn of layer 0 (and the status bits of the atom branch)
    else RUN(node_terms(l, x, st));
    {  // pre = e W1e^T + Pn_i[tgt] + Pn_j[src]
      CartnetGemmArgs a = gemm_args(E, D, D, D, D, 2 * D);
      a.ngroups = 2; a.b_kstrided = 1;
      a.A[0] = e; a.A[1] = e;
      a.B[0] = w.gate0T[l] + (size_t)2 * D * D; a.B[1] = w.aggr0T[l] + (size_t)2 * D * D;
      a.C[0] = w.pre[l]; a.C[1] = hcol(w.pre[l], D, half); a.c_half = half;
      a.gather_i[0] = w.Pn; a.gather_i[1] = w.Pn + D; a.gather_j[0] = w.Pn + 2 * D; a.gather_j[1] = w.Pn + 3 * D;
      a.ldg = 4 * D; a.tgt = w.tgt32; a.src = w.src32; a.gather_rows = N;
      if (w.i_pre[l]) { a.b_split[0] = w.i_pre[l]; a.b_split[1] = w.i_pre[l] + img_blk(m); }
      RUN(cartnet_gemm(&a, st));
    }
    // Inference (eval-mode BatchNorm, nothing kept for backward, fp32 MFMA): the second Linears, the gate, the per-target
    // sums and the edge residual are ONE kernel -- gs never reaches memory (csrc/gemm_f32gate.hip)
    // (BatchNorm groups change nothing in eval mode -- every group's row of mean_rstd holds the running statistics)
    if (!training && !need_backward && m.gemm_precision == 0 && !half && D % 256 == 0 && w.i_gs[l] &&
        w.gate_bnd && E > 0) {
      RUN(bn_stats(w.cs, w.cq, w.tiles_e, b.E, m.buf[l].norm_mean, m.buf[l].norm_var, m.buf[l].norm_nbt, w.mr1[l], 1));
      CartnetGateGemmArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.pre = w.pre[l]; ga.ldp = 2 * D;
      ga.img_gate = w.i_gs[l]; ga.img_aggr = w.i_gs[l] + img_blk(m);
      ga.bias_gate = q.gate2_b; ga.bias_aggr = q.aggr2_b;
      ga.mean_rstd = w.mr1[l]; ga.gamma = q.norm_w; ga.beta = q.norm_b;
      ga.env = m.use_envelope[l] ? w.env : nullptr;
      ga.e_in = e; ga.e_out = e_next; ga.tgt = w.tgt32; ga.rowptr = w.rowptr; ga.aggr = w.aggr[l]; ga.bnd = w.gate_bnd;
      ga.E = b.E; ga.N = N; ga.D = D;
      RUN(cartnet_gate_gemm_eval(&ga, st));
      RUN(bn_stats(w.ps, w.pq, w.gparts, N, m.buf[l].norm2_mean, m.buf[l].norm2_var, m.buf[l].norm2_nbt, w.mr2[l], 0));
      RUN(cartnet_node_update_fwd(w.aggr[l], x, w.mr2[l], q.norm2_w, q.norm2_b, N, D, x_next, w.groups, st));
      x = x_next;
      e = e_next;
      continue;
    }
    {  // gs = silu(pre) W2^T + b2, BatchNorm statistics of the gate half
      CartnetGemmArgs a = gemm_args(E, D, D, 2 * D, D, 2 * D);
      a.ngroups = 2; a.b_kstrided = 1; a.a_act = 1;
      a.A[0] = w.pre[l]; a.A[1] = hcol(w.pre[l], D, half); a.B[0] = w.gate2T[l]; a.B[1] = w.aggr2T[l];
      a.C[0] = w.gs[l]; a.C[1] = hcol(w.gs[l], D, half); a.bias[0] = q.gate2_b; a.bias[1] = q.aggr2_b;
      a.a_half = half; a.c_half = half;
      if (!w.groups && training) { a.colsum[0] = w.cs; a.colsq[0] = w.cq; }   // eval mode normalises with the running statistics
      if (w.i_gs[l]) { a.b_split[0] = w.i_gs[l]; a.b_split[1] = w.i_gs[l] + img_blk(m); }
      if (w.act[l]) { a.a_act_out[0] = w.act[l]; a.a_act_out[1] = w.act[l] + D; }
      RUN(cartnet_gemm(&a, st));
    }
    // BatchNorm groups: a 128-row GEMM tile may straddle two groups, so the gate statistics are taken per group by a
    // pass of their own over the gate half of gs (181 MB at the benchmark batch, ~40 us) instead of in the epilogue
    if (w.groups && training) RUN(cartnet_colstats_grouped(w.gs[l], 2 * D, D, w.groups, w.cs, w.cq, st));
    RUN(bn_stats(w.cs, w.cq, w.tiles_e, b.E, m.buf[l].norm_mean, m.buf[l].norm_var, m.buf[l].norm_nbt, w.mr1[l], 1));
    if (training && w.bc[l])      // also leaves the per-target sums the backward pass builds its BatchNorm sums from
      RUN(cartnet_gate_scatter_fwd_bc(w.gs[l], e, m.use_envelope[l] ? w.env : nullptr, w.rowptr, w.mr1[l], q.norm_w, q.norm_b,
                                      N, D, e_next, w.aggr[l], w.ps, w.pq, w.bc[l], st));
    else
      RUN((half ? cartnet_gate_scatter_fwd_h : cartnet_gate_scatter_fwd_f)(
          w.gs[l], e, m.use_envelope[l] ? w.env : nullptr, w.rowptr, w.mr1[l], q.norm_w, q.norm_b, N, D, e_next, w.aggr[l],
          w.ps, w.pq, w.groups, st));
    RUN(bn_stats(w.ps, w.pq, w.gparts, N, m.buf[l].norm2_mean, m.buf[l].norm2_var, m.buf[l].norm2_nbt, w.mr2[l], 0));
    RUN(cartnet_node_update_fwd(w.aggr[l], x, w.mr2[l], q.norm2_w, q.norm2_b, N, D, x_next, w.groups, st));
    x = x_next;
    e = e_next;
  }

  // ---- head
  {
    CartnetGemmArgs a = gemm_args(N, H, D, D, H, H);
    a.A[0] = x; a.B[0] = w.head0T; a.C[0] = w.hid; a.bias[0] = P.head0_b; a.b_kstrided = 1;
    RUN(cartnet_gemm(&a, st));
  }
  if (m.cholesky) {
    RUN(cartnet_cholesky_head_fwd(w.hid, w.idx, P.head2_w, P.head2_b, N, H, w.p6, pred, st));
  } else {
    RUN(cartnet_scalar_head_fwd(w.hid, P.head2_w, P.head2_b, b.graph_ptr, b.Bg, H, pred, st));
  }
#undef FORK
#undef MAIN_WAITS
  return 0;
}

// Two streams: the MAIN stream carries the chain of activation gradients (dx, de from layer to layer); everything that
// only produces parameter gradients -- the weight-gradient GEMMs with their split-K reductions and the bias-gradient
// finalisers, a third of the FLOPs -- goes to the SIDE stream, so it fills the tails of the main-stream GEMM grids and
// overlaps the HBM-bound gate / segment kernels.  Buffers the side stream reads are double-buffered per layer parity.
extern "C" int cartnet_model_backward(const CartnetModel* model, const CartnetBatch* batch, void* workspace,
                                      size_t workspace_bytes, int32_t training, const float* dpred, const float* x_out,
                                      const CartnetParams* grads, void* stream, void* aux_stream) {
  RUN(check_model(model, batch, "cartnet_model_backward"));
  g_precision = model->gemm_precision;
  const CartnetModel& m = *model;
  const CartnetBatch& b = *batch;
  const CartnetParams& P = m.p;
  CN_CHECK(workspace && dpred && x_out && grads, "cartnet_model_backward: null argument");
  const CartnetParams& G = *grads;
  const bool plain = !m.use_temperature && !m.atom_types;
  CN_CHECK((plain || (G.atom_w && G.atom_b)) && G.edge0_w && G.edge0_b && G.edge2_w && G.edge2_b && G.head0_w &&
               G.head0_b && G.head2_w && G.head2_b && (!(m.atom_types || plain) || G.embedding) &&
               (!m.use_temperature || (G.temp_w && G.temp_b)) && (m.use_temperature || !m.atom_types || G.enc_bias),
           "cartnet_model_backward: a gradient destination is missing");
  size_t need = 0;
  Work w = carve(m, b.N, b.E, b.Bg, b.M, true, static_cast<char*>(workspace), &need);
  CN_CHECK(workspace_bytes >= need, "cartnet_model_backward: workspace %zu < required %zu bytes", workspace_bytes, need);
  w.groups = w.G > 1 ? &w.grp : nullptr;      // node_gptr / edge_gptr were filled by the forward call
  const int D = m.D, L = m.L, H = D / 2, N = b.N;
  const int E = (int)b.E;
  g_events.next = 0;
  Streams S{(hipStream_t)stream, aux_stream ? (hipStream_t)aux_stream : (hipStream_t)stream,
            aux_stream != nullptr && aux_stream != stream, &g_events};
  void* st = stream;            // main: activation-gradient chain
  void* sw = (void*)S.side;     // side: parameter gradients
#define FORK() do { if (S.fork() != 0) { cartnet_set_error("cartnet_model_backward: stream fork failed"); return 2; } } while (0)

  hipEvent_t sort_done = nullptr;
  // Two slow single-workgroup jobs nobody waits for until the very end (atoms grouped by element for the embedding
  // gradient; the head's second-Linear gradient sums): queued on the side stream behind the first layer's weight
  // gradients, where that stream idles while the main stream runs the next layer's gate kernels -- in front of them
  // they delayed every weight-gradient product of the step by ~0.4 ms.
  const int head_row = m.cholesky ? 7 * H + 8 : 2 * H + 8;
  auto deferred_side_jobs = [&]() -> int {
    if (m.atom_types) {
      RUN(cartnet_sort_by_key(b.z, N, m.n_types, w.zperm, w.zptr, w.zstatus, sw));
      sort_done = S.mark_side();
    }
    RUN(cartnet_colsum_finalize_f32(w.head_parts, w.nparts_n, head_row, w.head_tot, sw));
    const int nw2 = m.cholesky ? 6 * H : H, nb2 = m.cholesky ? 6 : 1;
    if (hipMemcpyAsync(G.head2_w, w.head_tot, sizeof(float) * nw2, hipMemcpyDeviceToDevice, S.side) != hipSuccess ||
        hipMemcpyAsync(G.head2_b, w.head_tot + nw2, sizeof(float) * nb2, hipMemcpyDeviceToDevice, S.side) != hipSuccess ||
        hipMemcpyAsync(G.head0_b, w.head_tot + nw2 + 8, sizeof(float) * H, hipMemcpyDeviceToDevice, S.side) != hipSuccess) {
      cartnet_set_error("cartnet_model_backward: head gradient copy failed");
      return 2;
    }
    return 0;
  };
  // ---- head
  {
    if (m.cholesky)
      RUN(cartnet_cholesky_head_bwd(w.hid, w.idx, P.head2_w, w.p6, dpred, N, H, w.dhid, w.head_parts, st));
    else
      RUN(cartnet_scalar_head_bwd(w.hid, P.head2_w, b.graph_ptr, b.batch, dpred, N, b.Bg, H, w.dhid, w.head_parts, st));
    hipEvent_t dhid_done = S.mark_main();
    CartnetGemmArgs a = gemm_args(N, D, H, H, D, D);
    a.A[0] = w.dhid; a.B[0] = P.head0_w; a.C[0] = w.dx[0]; a.b_kstrided = 1; a.b_split[0] = w.i_head0_b;
    RUN(cartnet_gemm(&a, st));       // (main stream first: see the host-order note in the layer loop)
    if (S.side_waits(dhid_done) != 0) { cartnet_set_error("cartnet_model_backward: stream fork failed"); return 2; }
    const float* dY[1] = {w.dhid};
    const float* X[1] = {x_out};
    float* o[1] = {G.head0_w};
    RUN(wgrad(dY, H, X, D, o, D, N, H, D, 1, false, w, sw));
  }
  float* dx = w.dx[0];
  float* dx_other = w.dx[1];
  float* de = nullptr;   // the head does not read the edge features
  int de_slot = 0;
  const bool pairs = m.gemm_precision >= 1;
  hipEvent_t side_done[CARTNET_MAX_LAYERS + 2];
  for (int i = 0; i < CARTNET_MAX_LAYERS + 2; ++i) side_done[i] = nullptr;

  const bool half = m.half_storage != 0 && m.gemm_precision == 2;
  CN_CHECK(!m.half_storage || (m.gemm_precision == 2 && !w.groups && D % 256 == 0),
           "cartnet_model_backward: half_storage needs gemm_precision 2, D %% 256 == 0 and no BatchNorm groups");
  // sync-BatchNorm: the sums of the BatchNorm backward over all ranks for the apply pass (pre-scaled so that the kernels'
  // division by the LOCAL row count yields sum_global / count_global); the affine gradients keep the local sums
  const bool sync_bn = training && m.bn_allreduce != nullptr;
  CN_CHECK(!(sync_bn && w.groups), "cartnet_model_backward: sync-BatchNorm and BatchNorm groups are mutually exclusive");
  auto bn_sums_sync = [&](double* pa, double* pb, int nparts, long long count, float* sums, float* grad_b, float* grad_w) -> int {
    RUN(cartnet_bn_sync_gather(pa, pb, nparts, D, count, w.bnrow, grad_b, grad_w, st));
    CN_CHECK(m.bn_allreduce(m.bn_allreduce_user, w.bnrow, 2 * (int64_t)D + 1, st) == 0,
             "cartnet_model_backward: the sync-BatchNorm all-reduce callback failed");
    return cartnet_bn_sync_scale(w.bnrow, D, count, sums, st);
  };

  // Gradient buckets (CartnetModel.grad_ready): every gradient of layer l has been enqueued on one of the two streams by
  // now (the BatchNorm affine gradients on the main stream before this layer's last fork, everything else on the side
  // stream) -- a last fork orders the side stream behind the main stream's share, then the caller may queue the bucket's
  // all-reduce there, under the next layer's kernels.  The head's bucket completes with the deferred side jobs.
  auto bucket_ready = [&](int l) -> int {
    if (!m.grad_ready) return 0;
    FORK();
    if (l == L - 1)
      CN_CHECK(m.grad_ready(m.grad_ready_user, L, sw) == 0, "cartnet_model_backward: the grad_ready callback failed (head)");
    CN_CHECK(m.grad_ready(m.grad_ready_user, l, sw) == 0, "cartnet_model_backward: the grad_ready callback failed (layer %d)", l);
    return 0;
  };

  // ---- layers, last to first
  for (int l = L - 1; l >= 0; --l) {
    const CartnetLayerParams& q = P.layer[l];
    const CartnetLayerParams& gq = G.layer[l];
    CN_CHECK(gq.gate0_w && gq.gate0_b && gq.gate2_w && gq.gate2_b && gq.aggr0_w && gq.aggr0_b && gq.aggr2_w &&
                 gq.aggr2_b && gq.norm_w && gq.norm_b && gq.norm2_w && gq.norm2_b,
             "cartnet_model_backward: gradient destination missing in layer %d", l);
    const int par = l & 1;
    const float* x_in = (l == 0) ? w.xenc : w.xl[l - 1];
    const float* e_in = (l == 0) ? w.e0 : w.el[l - 1];
    const float* env = m.use_envelope[l] ? w.env : nullptr;
    const float* pre = w.pre[l];
    float* gs = w.gs[l];
    float* dpre = w.dpre[par];
    float* dPn = w.dPn[par];
    // the side stream must be done with this parity's buffers (used two layers ago)
    if (l + 2 < L && S.main_waits(side_done[l + 2]) != 0) { cartnet_set_error("cartnet_model_backward: wait failed"); return 2; }
    // node update: x_out = silu(bn2(aggr)) + x_in
    RUN(cartnet_node_update_bwd_stats(w.aggr[l], dx, w.mr2[l], q.norm2_w, q.norm2_b, N, D, w.pa, w.pb, w.groups, st));
    if (sync_bn) {
      RUN(bn_sums_sync(w.pa, w.pb, w.nparts_n, N, w.sums2, gq.norm2_b, gq.norm2_w));
    } else if (w.groups) {   // per-group sums for the apply pass, their total = the BatchNorm affine gradients
      RUN(cartnet_group_sums_finalize(w.pa, w.pb, D, w.groups, 0, w.sums2, gq.norm2_b, gq.norm2_w, st));
    } else {
      double* parts[2] = {w.pa, w.pb};
      float* outs[2] = {w.sums2, w.sums2 + D};
      float* grads2[2] = {gq.norm2_b, gq.norm2_w};      // the same sums are the BatchNorm affine gradients
      RUN(cartnet_colsum_finalize2(parts, outs, grads2, 2, w.nparts_n, D, st));
    }
    const bool fused_sums = training && w.bc[l] != nullptr;
    if (fused_sums) {
      // sum(dbn), sum(dbn ghat) = the rows the dE epilogue of layer l + 1 left (de_out's share) + the atoms' share, taken
      // here while daggr is in registers: no statistics pass over gs / de_out
      double* node_a = w.gpa + (size_t)w.tiles_e * D;
      double* node_b = w.gpb + (size_t)w.tiles_e * D;
      RUN(cartnet_node_update_bwd_apply_bc(w.aggr[l], dx, w.mr2[l], q.norm2_w, q.norm2_b, w.sums2, training, N, D, w.daggr,
                                           w.bc[l], node_a, node_b, st));
      double* parts[2] = {de ? w.gpa : node_a, de ? w.gpb : node_b};
      float* outs[2] = {w.sums1, w.sums1 + D};
      float* grads1[2] = {gq.norm_b, gq.norm_w};
      RUN(cartnet_colsum_finalize2(parts, outs, grads1, 2, (de ? w.tiles_e : 0) + w.nparts_n, D, st));
    } else {
    RUN(cartnet_node_update_bwd_apply(w.aggr[l], dx, w.mr2[l], q.norm2_w, q.norm2_b, w.sums2, training, N, D, w.daggr,
                                      w.groups, st));
    // gate * sender aggregation and the edge BatchNorm
    RUN((half ? cartnet_gate_scatter_bwd_stats_h : cartnet_gate_scatter_bwd_stats_f)(
        gs, de, w.daggr, env, w.rowptr, w.mr1[l], q.norm_w, q.norm_b, N, D, w.pa, w.pb, w.groups, st));
    }
    if (fused_sums) {
    } else if (sync_bn) {
      RUN(bn_sums_sync(w.pa, w.pb, w.gparts, b.E, w.sums1, gq.norm_b, gq.norm_w));
    } else if (w.groups) {
      RUN(cartnet_group_sums_finalize(w.pa, w.pb, D, w.groups, 1, w.sums1, gq.norm_b, gq.norm_w, st));
    } else {
      double* parts[2] = {w.pa, w.pb};
      float* outs[2] = {w.sums1, w.sums1 + D};
      float* grads1[2] = {gq.norm_b, gq.norm_w};
      RUN(cartnet_colsum_finalize2(parts, outs, grads1, 2, w.gparts, D, st));
    }
    RUN((half ? cartnet_gate_scatter_bwd_apply_h : cartnet_gate_scatter_bwd_apply_f)(
        gs, de, w.daggr, env, w.rowptr, w.mr1[l], q.norm_w, q.norm_b, w.sums1, b.E, training, N, D, w.pc[par], w.pd[par],
        w.groups, st));   // gs = [dg | ds]
    float* de_in = w.de[de_slot];
    auto side_w2 = [&]() -> int {   // bias gradients of the second Linears, then their weight gradients
      double* parts[2] = {w.pc[par], w.pd[par]};
      float* outs[2] = {gq.gate2_b, gq.aggr2_b};
      RUN(cartnet_colsum_finalize(parts, outs, 2, w.gparts, D, sw));
      const float* dY[2] = {gs, hcol(gs, D, half)};
      const bool kept = w.act[l] != nullptr;            // silu(pre) kept by the forward pass: plain operand (precision 0)
      const float* X[2] = {kept ? w.act[l] : pre, hcol(kept ? w.act[l] : pre, D, half && !kept)};
      float* o[2] = {gq.gate2_w, gq.aggr2_w};
      return wgrad(dY, 2 * D, X, 2 * D, o, D, b.E, D, D, 2, !kept, w, sw, half, half && !kept);
    };
    auto main_dpre = [&]() -> int {   // dpre = (dgs @ W2) * silu'(pre), into this parity's buffer
      CartnetGemmArgs a = gemm_args(E, D, D, 2 * D, D, 2 * D);
      a.ngroups = 2; a.b_kstrided = 1;
      a.A[0] = gs; a.A[1] = hcol(gs, D, half); a.B[0] = q.gate2_w; a.B[1] = q.aggr2_w;
      a.C[0] = dpre; a.C[1] = hcol(dpre, D, half); a.dact[0] = pre; a.dact[1] = hcol(pre, D, half); a.ldd = 2 * D;
      a.a_half = half; a.c_half = half; a.dact_half = half;
      // (no column sums here: the bias gradients of the first Linears are the column sums of dpre over all edges =
      //  the column sums over atoms of its by-target segment sums, 14x fewer rows -- taken from dPn below)
      if (w.i_dpre[l]) { a.b_split[0] = w.i_dpre[l]; a.b_split[1] = w.i_dpre[l] + img_blk(m); }
      return cartnet_gemm(&a, st);
    };
    auto side_w1e = [&]() -> int {   // edge-block weight gradients of the first Linears
      const float* dY[2] = {dpre, hcol(dpre, D, half)};
      const float* X[2] = {e_in, e_in};
      float* o[2] = {gq.gate0_w + 2 * D, gq.aggr0_w + 2 * D};
      return wgrad(dY, 2 * D, X, D, o, 3 * D, b.E, D, D, 2, false, w, sw, half, false);
    };
    auto main_de_in = [&]() -> int {   // de_in = de_out + dpre @ W1[:, 2D:]  (layer 0: continue through the encoder's last SiLU)
      CartnetGemmArgs a = gemm_args(E, D, D, 2 * D, 3 * D, D);
      a.nsegs = 2; a.b_kstrided = 1;
      a.A[0] = dpre; a.A[1] = hcol(dpre, D, half); a.B[0] = q.gate0_w + 2 * D; a.B[1] = q.aggr0_w + 2 * D;
      a.a_half = half;
      a.C[0] = de_in; a.resid[0] = de; a.ldr = D;
      a.b_split_folded = w.i_de[l];
      if (l == 0) { a.dact[0] = w.e0_pre; a.ldd = D; a.colsum[0] = w.cs_misc[0]; }
      if (l > 0 && training && w.bc[l - 1]) {
        // de_in of this layer is de_out of layer l - 1: its share of that layer's gate BatchNorm-backward sums, from the
        // epilogue (gs[l - 1] still holds the forward g: its own backward overwrites it later)
        a.gst_g = w.gs[l - 1]; a.gst_ld = 2 * D;
        a.gst_env = m.use_envelope[l - 1] ? w.env : nullptr;
        a.gst_mean_rstd = w.mr1[l - 1]; a.gst_gamma = P.layer[l - 1].norm_w; a.gst_beta = P.layer[l - 1].norm_b;
        a.colsum[0] = w.gpa; a.colsq[0] = w.gpb;
        CN_CHECK(cartnet_gemm_gate_stats_ok(&a) == 1,
                 "cartnet_model_backward: the dE product of layer %d does not take the gate-statistics kernel (model.hip: "
                 "gate_sums_fused is out of step with gemm.hip: gate_stats_launch_ok)", l);
      }
      if (half && D != 256) {
        // the two K-segments only fold into one product at N = 256 (gemm.hip: segments_fold), and no kernel reads bf16
        // K-segments: two single-segment products instead, the second adding onto the first in place (every output
        // element is read and written by the same thread); the layer-0 epilogue goes on the second
        CartnetGemmArgs a1 = gemm_args(E, D, D, 2 * D, 3 * D, D);
        a1.b_kstrided = 1; a1.a_half = 1;
        a1.A[0] = dpre; a1.B[0] = q.gate0_w + 2 * D; a1.C[0] = de_in; a1.resid[0] = de; a1.ldr = D;
        a1.b_split[0] = w.i_de[l];
        RUN(cartnet_gemm(&a1, st));
        CartnetGemmArgs a2 = a1;
        a2.A[0] = hcol(dpre, D, true); a2.B[0] = q.aggr0_w + 2 * D; a2.resid[0] = de_in;
        a2.b_split[0] = w.i_de[l] + img_blk(m);
        if (l == 0) { a2.dact[0] = w.e0_pre; a2.ldd = D; a2.colsum[0] = w.cs_misc[0]; }
        RUN(cartnet_gemm(&a2, st));
      } else {
        RUN(cartnet_gemm(&a, st));
      }
      if (l == 0) {
        double* parts[1] = {w.cs_misc[0]};
        float* outs[1] = {G.edge2_b};
        RUN(cartnet_colsum_finalize(parts, outs, 1, w.tiles_e, D, st));
      }
      return 0;
    };
    // node-side halves: reduce dpre over each atom's incoming (target) and outgoing (source) edges
    auto segsums = [&](void* s_) -> int {
#ifndef CN_NO_SEG_PAIR
      if (!half) return cartnet_segment_sum_pair(dpre, 2 * D, w.rowptr, w.colptr, w.perm, N, 2 * D, dPn, dPn + 2 * D, 4 * D, 256, s_);
      return cartnet_segment_sum_pair_h(dpre, 2 * D, w.rowptr, w.colptr, w.perm, N, 2 * D, dPn, dPn + 2 * D, 4 * D, 256, s_);
#else                       /* (A/B builds: the two launches of rounds 1-4) */
      RUN((half ? cartnet_segment_sum_h : cartnet_segment_sum_f)(dpre, 2 * D, w.rowptr, nullptr, N, 2 * D, dPn, 4 * D, s_));
      return (half ? cartnet_segment_sum_h : cartnet_segment_sum_f)(dpre, 2 * D, w.colptr, w.perm, N, 2 * D, dPn + 2 * D,
                                                                    4 * D, s_);
#endif
    };
    auto side_wn = [&]() -> int {   // bias gradients of the first Linears = column sums of the by-target half of dPn; node blocks
      double* parts[2] = {w.pc[par], w.pd[par]};      // the second Linears' bias sums were finalised above (same stream)
      float* outs[2] = {gq.gate0_b, gq.aggr0_b};
      RUN(cartnet_colsum_partial(dPn, 4 * D, N, D, parts[0], sw));
      RUN(cartnet_colsum_partial(dPn + D, 4 * D, N, D, parts[1], sw));
      RUN(cartnet_colsum_finalize(parts, outs, 2, cartnet_segment_nparts(N), D, sw));
      const float* dY[4] = {dPn, dPn + D, dPn + 2 * D, dPn + 3 * D};
      const float* X[4] = {x_in, x_in, x_in, x_in};
      float* o[4] = {gq.gate0_w, gq.aggr0_w, gq.gate0_w + D, gq.aggr0_w + D};
      return wgrad(dY, 4 * D, X, D, o, 3 * D, N, D, D, 4, false, w, sw);
    };
    auto main_dx = [&]() -> int {
      CartnetGemmArgs a = gemm_args(N, D, D, 4 * D, 3 * D, D);
      a.nsegs = 4; a.b_kstrided = 1;
      a.A[0] = dPn; a.A[1] = dPn + D; a.A[2] = dPn + 2 * D; a.A[3] = dPn + 3 * D;
      a.B[0] = q.gate0_w; a.B[1] = q.aggr0_w; a.B[2] = q.gate0_w + D; a.B[3] = q.aggr0_w + D;
      a.C[0] = dx_other; a.resid[0] = dx; a.ldr = D;
      a.b_split_folded = w.i_dx[l];
      if (l == 0 && !plain) { a.dact[0] = w.xa_pre; a.ldd = D; a.colsum[0] = w.cs_misc[1]; }
      RUN(cartnet_gemm(&a, st));
      if (l == 0 && !plain) {
        double* parts[1] = {w.cs_misc[1]};
        float* outs[1] = {G.atom_b};
        RUN(cartnet_colsum_finalize(parts, outs, 1, w.tiles_n, D, st));
      }
      return 0;
    };
    // Two orders.  fp32: the weight-gradient stream only carries parameter-gradient work; the main stream the whole chain
    // dpre -> dE -> segment sums -> dX.  bf16x3 / bf16 (pairs): the matrix pipe is far from busy (the kernels wait on memory
    // and barriers), the weight-gradient stream idles 45 % of the step, and dE does not depend on the segment sums -- so
    // segment sums + the node-term weight gradients go to the side stream UNDER dE and only dX waits for them.  At fp32 the
    // same order measured no better (15.27 vs 15.18 ms: both streams already saturate the chip).
    if (pairs && S.dual) {
      // host order: the main stream's dpre and dE are in its queue before the ten launches of the side stream are made
      // (configs[2]: they are ~60 us of host time, and the main stream sat idle for them between dpre and dE)
      // Side-stream order by batch size (same-box A/B, tools/experiments/ab_pairs_order.sh).  Full-size batch: the second Linears'
      // weight gradients first, under dpre -- they need only gs, and the stream would idle until dpre is done (bf16x3 10.37
      // vs 10.52 ms).  Small batch (configs[2]): every kernel is one tile's latency and dX waits for the segment sums, which
      // came ~15 us after dE was done when queued behind those weight gradients (1.20 vs 1.17 ms): segment sums first.
      const bool seg_first = b.E < 32768;
      hipEvent_t apply_done = seg_first ? nullptr : S.mark_main();      // gs = [dg | ds] is final
      RUN(main_dpre());
      hipEvent_t dpre_done = S.mark_main();
      RUN(main_de_in());
      if (!seg_first) {
        if (S.side_waits(apply_done) != 0) { cartnet_set_error("cartnet_model_backward: stream fork failed"); return 2; }
        RUN(side_w2());
      }
      if (S.side_waits(dpre_done) != 0) { cartnet_set_error("cartnet_model_backward: stream fork failed"); return 2; }
      RUN(segsums(sw));
      hipEvent_t seg_done = S.mark_side();
      if (seg_first) RUN(side_w2());
      RUN(side_wn());
      RUN(side_w1e());
      side_done[l] = S.mark_side();
      if (l == L - 1) RUN(deferred_side_jobs());
      RUN(bucket_ready(l));
      if (S.main_waits(seg_done) != 0) { cartnet_set_error("cartnet_model_backward: wait failed"); return 2; }
      RUN(main_dx());
    } else {
      // (Round 6, with dpre and dE as persistent launches that own every CU: queueing the layer's weight gradients BEHIND dE
      //  -- so that the two products run back to back and the weight gradients share the chip with the segment sums, dX, the
      //  node update and the next gate kernels -- measured 13.29-13.32 against 13.14-13.21 ms for this order: dX, a 88 us
      //  product on the critical chain, then waits 400 us for a CU slot beside a weight-gradient launch and the node kernels
      //  run 3x slower beside one; profiles/HISTORY.md, "backward order")
      FORK();
      RUN(side_w2());
      RUN(main_dpre());
      FORK();
      RUN(side_w1e());
      RUN(main_de_in());
      RUN(segsums(st));
      FORK();
      RUN(side_wn());
      side_done[l] = S.mark_side();
      if (l == L - 1) RUN(deferred_side_jobs());
      RUN(bucket_ready(l));
      RUN(main_dx());
    }
    float* t = dx; dx = dx_other; dx_other = t;
    de = de_in;
    de_slot ^= 1;
  }

  // ---- encoder: de = d(e0_pre), dx = d(xa_pre)
  {
    hipEvent_t de_done = S.mark_main();
    CartnetGemmArgs a = gemm_args(E, 2 * D, D, D, 2 * D, 2 * D);
    a.A[0] = de; a.B[0] = P.edge2_w; a.C[0] = w.dhe; a.dact[0] = w.he_pre; a.ldd = 2 * D; a.b_kstrided = 1;
    a.dact_half = half;
    a.b_split[0] = w.i_edge2_b;
    a.colsum[0] = w.cs_misc[2];
    RUN(cartnet_gemm(&a, st));       // dhe = d(he_pre)
    if (S.side_waits(de_done) != 0) { cartnet_set_error("cartnet_model_backward: stream fork failed"); return 2; }
    const float* dY[1] = {de};
    const bool kept = w.he_act != nullptr;
    const float* X[1] = {kept ? w.he_act : w.he_pre};
    float* o[1] = {G.edge2_w};
    RUN(wgrad(dY, D, X, 2 * D, o, 2 * D, b.E, D, 2 * D, 1, !kept, w, sw, false, half && !kept));
    // The first edge Linear's gradients stay on the MAIN stream (they need dhe, and the main stream has nothing else of
    // weight left): the weight-gradient stream still holds layer 0's products and dW2 of the encoder at this point, and
    // queueing these behind them left the main stream idle for the last ~0.8 ms of the step (r03 timeline).
    double* parts[1] = {w.cs_misc[2]};
    float* outs[1] = {G.edge0_b};
    RUN(cartnet_colsum_finalize(parts, outs, 1, w.tiles_e, 2 * D, st));
    const float* dY2[1] = {w.dhe};
    const float* X2[1] = {w.feat};
    float* o2[1] = {G.edge0_w};
    // (the split-K slabs of this product must not be the ones the weight-gradient stream is using: w.slabs2)
    Work w2 = w;
    w2.slabs = w.slabs2;
    if (w.i_edge0) {
      // the DMA-fed weight-gradient kernels want 256-wide column tiles: compute the transpose, featT . dhe = dW0^T [ldf, 2D]
      // (rows kf.. are the zero pad columns of feat; one ragged row tile), and write its first kf rows transposed into the
      // gradient.  (Round 4: at fp32 too -- as dhe^T . feat with N = kf it ran on the general kernel, 0.7 ms at the END of
      // the main stream.)
      const float* fT[1] = {w.feat};
      const float* dh[1] = {w.dhe};
      float* oT[1] = {w.e0wT};
      RUN(wgrad(fT, w.ldf, dh, 2 * D, oT, 2 * D, b.E, w.ldf, 2 * D, 1, false, w2, st));
      const float* tsrc[1] = {w.e0wT};
      float* tdst[1] = {G.edge0_w};
      const int32_t trows[1] = {w.kf}, tcols[1] = {2 * D}, tlds[1] = {2 * D}, tldd[1] = {w.kf};
      RUN(cartnet_transpose(tsrc, tdst, trows, tcols, tlds, tldd, 1, st));
    } else {
      RUN(wgrad(dY2, 2 * D, X2, w.ldf, o2, w.kf, b.E, 2 * D, w.kf, 1, false, w2, st));
    }
  }
  if (plain) {   // every atom read the same learned row: its gradient is the column sum of dx over the atoms
    double* parts[1] = {w.pa};
    float* outs[1] = {G.embedding};
    RUN(cartnet_colsum_partial(dx, D, N, D, parts[0], st));
    RUN(cartnet_colsum_finalize(parts, outs, 1, cartnet_segment_nparts(N), D, st));
  } else {
    const float* dY[1] = {dx};
    const float* X[1] = {w.x0};
    float* o[1] = {G.atom_w};
    RUN(wgrad(dY, D, X, 2 * D, o, 2 * D, N, D, 2 * D, 1, true, w, sw));
    CartnetGemmArgs a = gemm_args(N, 2 * D, D, D, 2 * D, 2 * D);
    a.A[0] = dx; a.B[0] = P.atom_w; a.C[0] = w.dx0; a.dact[0] = w.x0; a.ldd = 2 * D; a.b_kstrided = 1;
    a.b_split[0] = w.i_atom_b;
    RUN(cartnet_gemm(&a, st));
    RUN(cartnet_node_embed_bwd(m.use_temperature ? b.batch : nullptr, m.use_temperature ? b.temperature : nullptr, w.dx0,
                               N, 2 * D, b.Bg, w.pa, w.pb, st));
    if (m.use_temperature) {
      double* parts[2] = {w.pa, w.pb};
      float* outs[2] = {G.temp_w, G.temp_b};
      RUN(cartnet_colsum_finalize(parts, outs, 2, w.nparts_n, 2 * D, st));
    } else {
      double* parts[1] = {w.pb};
      float* outs[1] = {G.enc_bias};
      RUN(cartnet_colsum_finalize(parts, outs, 1, w.nparts_n, 2 * D, st));
    }
    if (m.atom_types && S.main_waits(sort_done) != 0) { cartnet_set_error("cartnet_model_backward: wait failed"); return 2; }
    if (m.atom_types)
      RUN(cartnet_segment_sum_long(w.dx0, 2 * D, w.zptr, w.zperm, m.n_types, N, 2 * D, w.seg_tmp, G.embedding, 2 * D,
                                   st));
  }
  // join: everything queued on the side stream precedes whatever the caller enqueues next on the main stream
  if (S.main_waits(S.mark_side()) != 0) { cartnet_set_error("cartnet_model_backward: stream join failed"); return 2; }
  if (m.grad_ready)
    CN_CHECK(m.grad_ready(m.grad_ready_user, L + 1, st) == 0, "cartnet_model_backward: the grad_ready callback failed (encoder)");
#undef FORK
  return 0;
}

#ifdef CN_HOST_PROFILE
#include <cstdio>
// Prints and resets the per-line host times of this translation unit's RUN(...) statements (diagnostic build only).
extern "C" int cartnet_debug_host_profile(int32_t calls) {
  auto& p = cn_model::host_prof();
  double tot = 0;
  for (int i = 0; i < 4096; ++i) tot += p.us[i];
  std::printf("host time inside RUN(...) statements: %.1f us per call over %d calls\n", tot / calls, calls);
  for (int i = 0; i < 4096; ++i)
    if (p.n[i]) {
      std::printf("  line %4d  %7.1f us/call  x%-4.1f  %.90s\n", i, p.us[i] / calls, (double)p.n[i] / calls, p.what[i]);
      p.us[i] = 0; p.n[i] = 0;
    }
  std::fflush(stdout);
  return 0;
}
#endif

