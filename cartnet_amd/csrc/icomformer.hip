// iComformer (BASELINE configs[4]) sequenced in C++: cartnet_icomformer_forward / _backward, one host call per direction
// (include/cartnet_hip.h).  No arithmetic kernels of its own -- like model.hip this file orders the launches of the
// other translation units the way the reference executes its modules (models/comformer.py:115-132,
// models/comformer_conv.py:71-99,156-193) and carves every intermediate out of one caller-owned workspace.  Round 3 drove
// the same sequence from Python (cartnet_amd/comformer.py, ~250 ctypes calls and ~330 tensor allocations per step:
// 18-25 ms of host time); that file still holds the parameter containers, the eComformer path and the kernel mapping.
//
// Mapping (C = dim_in, heads = 1):
//   * every Linear = cartnet_gemm; key_update / lin_msg_update on cat[k_i, k_j, e] use the CartNet layer's algebraic
//     split: node halves once per atom (edge layer: once per edge / per (crystal, lattice vector)), gathered in the
//     row GEMM's epilogue (comformer_conv.py:90-99);
//   * alpha = q_i * key / sqrt(C) with its bn_att statistics = cartnet_rowmul_fwd over the segments;
//   * msg * sigmoid(bn_att(alpha)) + scatter-add = cartnet_gate_scatter_fwd (no envelope, no edge residual);
//   * softplus(x + bn(lin_concate(out))) = cartnet_softplus_update_fwd;
//   * the edge layer (comformer_conv.py:156-193) is the same block on 3E rows (edge x lattice vector) in segments of
//     three; lin_concate is applied after the sum over the three lattice vectors ((sum_i m_i) W^T + 3 b);
//   * lin_edge is FOLDED into the row block of key_update.0 / lin_msg_update.0 (round 4): no nonlinearity sits between
//     ea = e We^T + be and pr = ea W1e^T + ..., so pr = e (W1e We)^T + W1e be + ... with F = W1e We [C, C] and c = W1e be
//     formed once per step by C x C products.  The row-sized lin_edge product disappears from forward and backward
//     (d(e) = dpr F directly; dF = dpr^T e is the one row-sized weight gradient; dW1e = dF We^T + db be^T,
//     dWe = sum W1e^T dF, dbe = sum W1e^T db are C x C): -0.49 of 3.2 TFLOP per step.  Exact in real arithmetic.
#include "model_common.h"

namespace {
using namespace cn_model;

thread_local EventPool g_icf_events;

// ---- small index kernel: rows r = 3 e + i of the edge layer -------------------------------------------------------------
// idx_edge[r] = e, idx_gl[r] = 3 * crystal(source of e) + i, ptr3[e] = 3 e, gedge_ptr[g] = rowptr[graph_ptr[g]]
__global__ void cn_icf_index_kernel(const int* __restrict__ src32, const int64_t* __restrict__ batch, long long E,
                                    const int64_t* __restrict__ graph_ptr, const int* __restrict__ rowptr, int Bg,
                                    int* __restrict__ idx_edge, int* __restrict__ idx_gl, int* __restrict__ ptr3,
                                    int* __restrict__ gedge_ptr) {
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
  for (long long e = i0; e <= E; e += stride) {
    ptr3[e] = (int)(3 * e);
    if (e < E) {
      const int g = (int)batch[src32[e]];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        idx_edge[3 * e + i] = (int)e;
        idx_gl[3 * e + i] = 3 * g + i;
      }
    }
  }
  for (long long g = i0; g <= Bg; g += stride) gedge_ptr[g] = rowptr[(int)graph_ptr[g]];
}

// out[i, t] += u[i] * v[t]  (C x C block with row stride ldo): the bias part of dW1e = dF We^T + db be^T; blockIdx.y picks
// the (out, u) pair: key / msg in one launch
__global__ void cn_icf_rank1_kernel(float* __restrict__ out0, float* __restrict__ out1, int ldo, const float* __restrict__ u0,
                                    const float* __restrict__ u1, const float* __restrict__ v, int C) {
  float* __restrict__ out = blockIdx.y ? out1 : out0;
  const float* __restrict__ u = blockIdx.y ? u1 : u0;
  const int i = blockIdx.x;
  for (int t = threadIdx.x; t < C; t += blockDim.x) out[(size_t)i * ldo + t] += u[i] * v[t];
}

// out[n] = sum_k a0[k] B0[k * ldb + n] + sum_k a1[k] B1[k * ldb + n]  (a row times two C x C blocks: dbe = W1k[:, 2C:]^T db_k +
// W1m[:, 2C:]^T db_m).  As a one-row cartnet_gemm launch it was a 128-row tile of the general kernel, 30-70 us on the
// weight-gradient stream next to the big products; 16 columns x 16 k-slices per 256-thread block, eight loads in flight per
// thread, slices added in a fixed order.  (First forms: one load per iteration = a chain of 128 round trips, 156 us; the
// same with 1,024-thread blocks: 365 us -- sixteen waves of one block wait for a CU that the resident GEMM workgroups
// have left room on, four fit in the gaps.)
__global__ __launch_bounds__(256) void cn_icf_rowvec2_kernel(float* __restrict__ out, const float* __restrict__ a0,
                                                              const float* __restrict__ B0, const float* __restrict__ a1,
                                                              const float* __restrict__ B1, int ldb, int C) {
  __shared__ float red[16][16];
  const int cl = threadIdx.x & 15, col = blockIdx.x * 16 + cl, sl = threadIdx.x >> 4;
  float acc = 0.f;
  if (col < C) {
    for (int m = 0; m < 2; ++m) {
      const float* __restrict__ a = m ? a1 : a0;
      const float* __restrict__ B = m ? B1 : B0;
      for (int k0 = sl; k0 < C; k0 += 16 * 8) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 16 * u;
          av[u] = k < C ? a[k] : 0.f;
          bv[u] = k < C ? B[(size_t)k * ldb + col] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += av[u] * bv[u];
      }
    }
  }
  red[sl][cl] = acc;
  __syncthreads();
  if (sl == 0 && col < C) {
    float t = red[0][cl];
#pragma unroll
    for (int s = 1; s < 16; ++s) t += red[s][cl];
    out[col] = t;
  }
}

// ---- weight forms of one step -----------------------------------------------------------------------------------------
// forward operand forms (transposed copy [in, out] + LDS image of W^T) and backward images (W as [K = out, N = in])
enum { F_Q, F_K, F_V, F_EDGE, F_K1I, F_M1I, F_K1J, F_M1J, F_K1E, F_M1E, F_K2, F_M2, F_CAT, F_COUNT };
enum { B_QKV /* fold: q, k, v */, B_EDGE, B_E1 /* fold: key0[:, 2C:], msg0[:, 2C:] */, B_K1I, B_M1I, B_K2, B_M2, B_CAT,
       B_COUNT };

struct ConvW {
  const float* W[F_COUNT];   // the views themselves ([C rows] x [C cols], leading dimension ldw)
  int ldw[F_COUNT];
  float* T[F_COUNT];         // [in, out] contiguous copies (only when images are in use)
  char* F[F_COUNT];          // forward images
  char* B[B_COUNT];          // backward images
};

struct Jobs {                // batched image / transpose requests of a step
  std::vector<const float*> psrc; std::vector<void*> pdst; std::vector<int32_t> pK, pN, psk, psn;
  std::vector<const float*> tsrc; std::vector<float*> tdst; std::vector<int32_t> trows, tcols, tlds, tldd;
  void clear() { psrc.clear(); pdst.clear(); pK.clear(); pN.clear(); psk.clear(); psn.clear();
                 tsrc.clear(); tdst.clear(); trows.clear(); tcols.clear(); tlds.clear(); tldd.clear(); }
};
thread_local Jobs g_jobs;

// Round 5: alpha = q_i * key / sqrt(C) is never written -- the second Linear leaves the key rows in gs[:, :C], the
// statistics pass only reads them, and the gate kernels of both directions recompute the product (cartnet_att_gate_fwd,
// cartnet_att_gate_bwd_apply with key = NULL).  Needs the one-chunk kernels (C <= 256).
inline bool icf_alpha_free(int C) {
#ifdef CN_ICF_KEEP_ALPHA
  (void)C;
  return false;
#else
  return C <= 256;
#endif
}

struct IWork {
  // graph
  int *src32, *tgt32, *rowptr, *colptr, *perm, *idx, *idx_edge, *idx_gl, *ptr3, *gedge_ptr, *zperm, *zptr, *zstatus;
  // embeddings / features
  float *x0, *edge_feat, *nl, *nc;
  float *r_e, *pre_e, *e0, *r_nl, *pre_nl, *NLt, *r_na, *pre_na, *NA;
  // attention layers 0..3 and the edge layer (index 4)
  float *QKV[5], *KPi[4], *KPj[4], *pr[5], *keyb[5], *gs[5], *mr1[5], *aggr[5], *o[5], *mr2[5], *y[5];
  float* bc[5];      // conv layers (training): per-target sums of the forward gate kernel for the backward's BatchNorm sums
  float* act[5];     // silu(pr), kept by the forward GEMM that activates it (fp32 with images): dW2 then reads a plain operand
  float *KY, *VY, *Ka, *KYb, *bias3;
  float *Fk[5], *Fm[5], *ck[5], *cm[5], *dFk[5], *dFm[5];    // folded lin_edge (x) W1e per layer: F [C, C], c [C]; gradient temps
  // head
  float *hid, *p6;
  // weights of the step
  ConvW cw[5];
  float *rbfT, *rbfaT;
  char *rbfF, *rbfaF, *head0B;
  bool use_img;
  // statistics scratch (main stream) and (side stream)
  double *pa, *pb, *pc, *pd, *cs, *cq, *cs_side;
  // backward transients
  float *dhid, *head_parts, *head_tot, *dx_head;
  float *d_o[5], *dres[5], *daggr[5], *dQKV[5], *dpr[5], *sums1[5], *sums2[5];
  float *de[4], *dKP[4], *dx[4];       // dKP [N, 4C] = [d(term_i) key | d(term_j) key | d(term_i) msg | d(term_j) msg]
  float *dNA, *dKa, *dKYb, *dKY, *dVY, *dNL3, *de_old, *tmpb;
  float *dpre_e, *dpre_nl, *dpre_na, *gw1, *gw2, *gb1, *gb2, *seg_tmp, *seg_tmp_e, *dx_emb;
  float *slabs;
  size_t slab_floats;
  int tiles_3e, tiles_e, tiles_n, nparts_n, gp_n, gp_e, sp_n, sp_e, sp_3e;
};

size_t img_bytes(int prec, int K, int N) {
  return prec == 0 ? cartnet_gemm_pack_b_bytes(K, N) : cartnet_gemm_split_b_bytes(K, N);
}

IWork icarve(const CartnetIcfModel& m, int N, long long E, int Bg, int M, char* base, size_t* total) {
  IWork w;
  memset(&w, 0, sizeof(w));
  Carver c{base};
  const int C = m.C, H = C / 2;
  const size_t En = (size_t)(E > 0 ? E : 1), Nn = (size_t)(N > 0 ? N : 1), E3 = 3 * En, B3 = (size_t)Bg * 3;
  w.use_img = C % 256 == 0;
  w.tiles_e = tiles_m(E); w.tiles_n = tiles_m(N); w.tiles_3e = tiles_m(3 * E);
  w.nparts_n = cartnet_node_nparts(N);
  w.gp_n = cartnet_gate_scatter_nparts(N); w.gp_e = cartnet_gate_scatter_nparts((int)E);
  w.sp_n = cartnet_segment_nparts(N); w.sp_e = cartnet_segment_nparts((int)E); w.sp_3e = cartnet_segment_nparts((int)(3 * E));
  w.src32 = c.take<int>(En); w.tgt32 = c.take<int>(En); w.rowptr = c.take<int>(Nn + 1); w.colptr = c.take<int>(Nn + 1);
  w.perm = c.take<int>(En); w.idx = c.take<int>(Nn); w.idx_edge = c.take<int>(E3); w.idx_gl = c.take<int>(E3);
  w.ptr3 = c.take<int>(En + 1); w.gedge_ptr = c.take<int>((size_t)Bg + 1);
  w.zperm = c.take<int>(Nn); w.zptr = c.take<int>((size_t)m.n_types + 2); w.zstatus = c.take<int>(4);
  w.x0 = c.take<float>(Nn * C); w.edge_feat = c.take<float>(En); w.nl = c.take<float>(B3); w.nc = c.take<float>(E3);
  w.r_e = c.take<float>(En * C); w.pre_e = c.take<float>(En * C); w.e0 = c.take<float>(En * C);
  w.r_nl = c.take<float>(B3 * C); w.pre_nl = c.take<float>(B3 * C); w.NLt = c.take<float>(B3 * C);
  w.r_na = c.take<float>(E3 * C); w.pre_na = c.take<float>(E3 * C); w.NA = c.take<float>(E3 * C);
  for (int l = 0; l < 5; ++l) {
    const bool edge = l == 4;
    const size_t S = edge ? En : Nn;          // segments (= rows of QKV / aggr / o / y)
    const size_t R = edge ? E3 : En;          // rows of the attention block
    w.QKV[l] = c.take<float>(S * 3 * C);
    w.Fk[l] = c.take<float>((size_t)C * C); w.Fm[l] = c.take<float>((size_t)C * C);
    w.ck[l] = c.take<float>(C); w.cm[l] = c.take<float>(C);
    w.dFk[l] = c.take<float>((size_t)C * C); w.dFm[l] = c.take<float>((size_t)C * C);
    if (!edge) { w.KPi[l] = c.take<float>(Nn * 2 * C); w.KPj[l] = c.take<float>(Nn * 2 * C); }
    w.pr[l] = c.take<float>(R * 2 * C); w.keyb[l] = icf_alpha_free(C) ? nullptr : c.take<float>(R * 2 * C); w.gs[l] = c.take<float>(R * 2 * C);
    // Kept activation.  Rounds 2-3 measured no gain from it on the Python-sequenced path (the cheaper weight gradient sat
    // on a stream with slack); round 4's trace of the C++ sequence shows the register-staged dY^T silu(X) kernel at 1.36 ms
    // per launch, 6.8 ms per step, on a side stream the main stream now WAITS for at its joins
    w.act[l] = (w.use_img && m.gemm_precision == 0) ? c.take<float>(R * 2 * C) : nullptr;
    // (the edge layer's segments are the edges themselves, three rows each: bc would be an [E, 2C] matrix, as many bytes
    //  as a third of the statistics pass it saves -- conv layers only)
    // per-segment sums for the backward BatchNorm sums (conv layers; with the alpha-free forward the edge layer too: its
    // statistics pass would need alpha, and 363 MB written + 544 MB read there replace a 1.09 GB pass)
    w.bc[l] = (edge && !icf_alpha_free(C)) ? nullptr : c.take<float>(S * 2 * C);
    w.mr1[l] = c.take<float>(2 * C); w.aggr[l] = c.take<float>(S * C); w.o[l] = c.take<float>(S * C);
    w.mr2[l] = c.take<float>(2 * C);
    w.y[l] = (l == 3) ? nullptr : c.take<float>(S * C);     // layer 3 writes the caller's x_out
  }
  w.KY = c.take<float>(B3 * C); w.VY = c.take<float>(B3 * C); w.Ka = c.take<float>(En * 2 * C);
  w.KYb = c.take<float>(B3 * 2 * C); w.bias3 = c.take<float>(C);
  w.hid = c.take<float>(Nn * H); w.p6 = c.take<float>((size_t)(M > 0 ? M : 1) * 6);
  // weight forms
  const size_t ib = w.use_img ? img_bytes(m.gemm_precision, C, C) : 0;
  for (int l = 0; l < 5; ++l) {
    if (!w.use_img) continue;
    for (int f = 0; f < F_COUNT; ++f) {
      w.cw[l].T[f] = c.take<float>((size_t)C * C);
      w.cw[l].F[f] = c.take<char>(ib);
    }
    w.cw[l].B[B_QKV] = c.take<char>(3 * ib);
    w.cw[l].B[B_E1] = c.take<char>(2 * ib);
    w.cw[l].B[B_K1I] = c.take<char>(2 * ib);      // fold: key0[:, :C], key0[:, C:2C]  (the i and j node blocks)
    w.cw[l].B[B_M1I] = c.take<char>(2 * ib);      // fold: msg0[:, :C], msg0[:, C:2C]
    for (int k : {B_EDGE, B_K2, B_M2, B_CAT}) w.cw[l].B[k] = c.take<char>(ib);
  }
  if (w.use_img) {
    w.rbfT = c.take<float>((size_t)C * C); w.rbfaT = c.take<float>((size_t)C * C);
    w.rbfF = c.take<char>(ib); w.rbfaF = c.take<char>(ib);
    w.head0B = c.take<char>(img_bytes(m.gemm_precision, H, C));
  }
  // statistics scratch: fp64 partial rows.  The largest producers: GEMM epilogues over 3E rows (tiles_3e rows), the
  // gate / segment kernels (<= 1024 rows)
  const size_t prow = (size_t)std::max(std::max(w.tiles_3e, 1024), w.nparts_n) * C;
  w.pa = c.take<double>(prow); w.pb = c.take<double>(prow); w.pc = c.take<double>(prow); w.pd = c.take<double>(prow);
  w.cs = c.take<double>(prow); w.cq = c.take<double>(prow); w.cs_side = c.take<double>(prow);
  // backward
  const int head_row = 7 * H + 8;
  w.dhid = c.take<float>(Nn * H); w.head_parts = c.take<float>((size_t)w.nparts_n * head_row);
  w.head_tot = c.take<float>(head_row); w.dx_head = c.take<float>(Nn * C);
  for (int l = 0; l < 5; ++l) {
    const bool edge = l == 4;
    const size_t S = edge ? En : Nn, R = edge ? E3 : En;
    w.d_o[l] = c.take<float>(S * C); w.dres[l] = c.take<float>(S * C); w.daggr[l] = c.take<float>(S * C);
    w.dQKV[l] = c.take<float>(S * 3 * C); w.dpr[l] = c.take<float>(R * 2 * C);
    w.sums1[l] = c.take<float>(2 * C); w.sums2[l] = c.take<float>(2 * C);
    if (!edge) {
      w.de[l] = c.take<float>(En * C);
      w.dKP[l] = c.take<float>(Nn * 4 * C); w.dx[l] = c.take<float>(Nn * C);
    }
  }
  w.dNA = c.take<float>(E3 * C); w.dKa = c.take<float>(En * 2 * C);
  w.dKYb = c.take<float>(B3 * 2 * C); w.dKY = c.take<float>(B3 * C); w.dVY = c.take<float>(B3 * C);
  w.dNL3 = c.take<float>(B3 * C); w.de_old = c.take<float>(En * C); w.tmpb = c.take<float>(C);
  w.dpre_e = c.take<float>(En * C); w.dpre_nl = c.take<float>(B3 * C); w.dpre_na = c.take<float>(E3 * C);
  w.gw1 = c.take<float>((size_t)C * C); w.gw2 = c.take<float>((size_t)C * C); w.gb1 = c.take<float>(C); w.gb2 = c.take<float>(C);
  w.seg_tmp = c.take<float>(Nn * C); w.dx_emb = c.take<float>(Nn * C);
  w.seg_tmp_e = c.take<float>((size_t)cartnet_segment_chunked_rows(Bg, (int)En) * 6 * C);     // per-crystal sums over [E, 6C]
  // split-K slabs of the weight-gradient products (one set per stream order: all of them run on the side stream)
  size_t sl = 0;
  auto slab = [&](int groups, long long K, int Mm, int Nn_) { sl = std::max(sl, wgrad_slab_floats(groups, K, Mm, Nn_)); };
  slab(2, 3 * E, C, C); slab(2, E, C, C); slab(1, E, C, C); slab(1, 3 * E, C, C); slab(3, E, C, C); slab(3, N, C, C);
  slab(4, N, C, C); slab(1, N, C, C); slab(1, N, H, C); slab(2, Bg * 3, C, C); slab(1, Bg, C, C); slab(1, Bg * 3, C, C);
  w.slab_floats = sl;
  w.slabs = c.take<float>(sl > 0 ? sl : 1);
  if (total) *total = align_up(c.off);
  return w;
}

inline CartnetGemmArgs gargs(int prec, int M, int N, int K, int lda, int ldb, int ldc) {
  CartnetGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.precision = prec;
  a.tile_policy = 1;      // grouped C x C products on the 128-wide fp32 kernel (CartnetGemmArgs.tile_policy; round 3: -1.8 %)
  a.M = M; a.N = N; a.K = K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.ngroups = 1; a.nsegs = 1; a.splitk = 1;
  return a;
}

// outs[g] = dY[g]^T @ (silu?)(X[g]) over K rows (split-K slabs summed in fixed order); K == 0 -> zeros
int iwgrad(int prec, const float* const* dY, int ldy, const float* const* X, int ldx, float* const* outs, int ldo,
           long long K, int M, int N, int groups, bool b_act, const IWork& w, void* st) {
  if (K <= 0) {
    for (int g = 0; g < groups; ++g)
      for (int r = 0; r < M; ++r)
        if (hipMemsetAsync(outs[g] + (size_t)r * ldo, 0, sizeof(float) * N, (hipStream_t)st) != hipSuccess) return 2;
    return 0;
  }
  const int S = split_k(K, wgrad_tiles(groups, M, N));
  CartnetGemmArgs a = gargs(prec, M, N, (int)K, ldy, ldx, S > 1 ? N : ldo);
  a.ngroups = groups; a.a_kstrided = 1; a.b_kstrided = 1; a.b_act = b_act ? 1 : 0; a.splitk = S;
  const float* slabp[CARTNET_MAX_GROUPS];
  for (int g = 0; g < groups; ++g) {
    a.A[g] = dY[g]; a.B[g] = X[g];
    a.C[g] = S > 1 ? w.slabs + (size_t)g * S * M * N : outs[g];
    slabp[g] = a.C[g];
  }
  RUN(cartnet_gemm(&a, st));
  if (S > 1) RUN(cartnet_splitk_reduce(slabp, outs, groups, S, M, N, ldo, st));
  return 0;
}

int check_icf(const CartnetIcfModel* m, const CartnetBatch* b, const char* who) {
  CN_CHECK(m && b, "%s: null model/batch", who);
  CN_CHECK(m->C >= 8 && m->C % 8 == 0 && m->C / 2 <= 512, "%s: dim_in=%d must be a multiple of 8, <= 1024", who, m->C);
  CN_CHECK(m->gemm_precision >= 0 && m->gemm_precision <= 2, "%s: gemm_precision=%d", who, m->gemm_precision);
  CN_CHECK(b->N >= 0 && b->E >= 0 && b->Bg >= 1 && b->M >= 0, "%s: bad batch sizes", who);
  CN_CHECK(3 * b->E < 2147483647LL && (long long)3 * b->E * 2 * m->C < 2147483647LL * 4, "%s: batch too large for 32-bit indexing", who);
  CN_CHECK(b->z && b->batch && b->graph_ptr && b->temperature && b->non_h_mask &&
               (b->E == 0 || (b->edge_index && b->cart_dist && b->cart_dir)), "%s: null batch pointer", who);
  const CartnetIcfParams& p = m->p;
  CN_CHECK(p.embedding && p.temp_w && p.temp_b && p.rbf_w && p.rbf_b && p.rbf_angle_w && p.rbf_angle_b && p.head0_w &&
               p.head0_b && p.head2_w && p.head2_b && m->rbf_centers && m->rbf_angle_centers, "%s: null parameter", who);
  for (int l = 0; l < 5; ++l) {
    const CartnetIcfConv& q = l < 4 ? p.att[l] : p.edge;
    CN_CHECK(q.query_w && q.query_b && q.key_w && q.key_b && q.value_w && q.value_b && q.edge_w && (l == 4 || q.edge_b) &&
                 q.concate_w && q.concate_b && q.key0_w && q.key0_b && q.key2_w && q.key2_b && q.msg0_w && q.msg0_b &&
                 q.msg2_w && q.msg2_b && q.bn_w && q.bn_b && q.bn_att_w && q.bn_att_b,
             "%s: null parameter in %s", who, l < 4 ? "an attention layer" : "the edge layer");
    if (l == 4)
      for (int i = 0; i < 3; ++i)
        CN_CHECK(q.key_e_w[i] && q.key_e_b[i] && q.value_e_w[i] && q.value_e_b[i], "%s: null lin_key_e / lin_value_e", who);
  }
  return 0;
}

// the views of one conv's weights and, with images in use, the requests that build their forms
void plan_conv(const CartnetIcfConv& q, const float* Fk, const float* Fm, int C, int prec, bool use_img, ConvW& cw, Jobs& J) {
  // (F_EDGE keeps its slot -- lin_edge itself is no longer applied to rows -- so that the indices stay put; F_K1E / F_M1E
  //  are the folded matrices F = W1e We)
  const float* W[F_COUNT] = {q.query_w, q.key_w, q.value_w, q.edge_w, q.key0_w, q.msg0_w, q.key0_w + C, q.msg0_w + C,
                             Fk, Fm, q.key2_w, q.msg2_w, q.concate_w};
  const int ld[F_COUNT] = {C, C, C, C, 3 * C, 3 * C, 3 * C, 3 * C, C, C, C, C, C};
  for (int f = 0; f < F_COUNT; ++f) { cw.W[f] = W[f]; cw.ldw[f] = ld[f]; }
  if (!use_img) return;
  const size_t ib = img_bytes(prec, C, C);
  auto fwd_img = [&](const float* w_, int ldw, char* dst) {       // B[k, n] = W[n, k]
    J.psrc.push_back(w_); J.pdst.push_back(dst); J.pK.push_back(C); J.pN.push_back(C); J.psk.push_back(1); J.psn.push_back(ldw);
  };
  auto k_img = [&](const float* w_, int ldw, char* dst) {         // B[k, n] = W[k, n]
    J.psrc.push_back(w_); J.pdst.push_back(dst); J.pK.push_back(C); J.pN.push_back(C); J.psk.push_back(ldw); J.psn.push_back(1);
  };
  for (int f = 0; f < F_COUNT; ++f) {
    if (f == F_EDGE) continue;
    fwd_img(W[f], ld[f], cw.F[f]);
    J.tsrc.push_back(W[f]); J.tdst.push_back(cw.T[f]); J.trows.push_back(C); J.tcols.push_back(C);
    J.tlds.push_back(ld[f]); J.tldd.push_back(C);
  }
  k_img(q.query_w, C, cw.B[B_QKV]); k_img(q.key_w, C, cw.B[B_QKV] + ib); k_img(q.value_w, C, cw.B[B_QKV] + 2 * ib);
  k_img(Fk, C, cw.B[B_E1]); k_img(Fm, C, cw.B[B_E1] + ib);                  // d(rows) = [dpr_k | dpr_m] [Fk; Fm]
  k_img(q.key0_w, 3 * C, cw.B[B_K1I]); k_img(q.msg0_w, 3 * C, cw.B[B_M1I]);
  k_img(q.key0_w + C, 3 * C, cw.B[B_K1I] + ib); k_img(q.msg0_w + C, 3 * C, cw.B[B_M1I] + ib);
  k_img(q.key2_w, C, cw.B[B_K2]); k_img(q.msg2_w, C, cw.B[B_M2]); k_img(q.concate_w, C, cw.B[B_CAT]);
}

int flush_jobs(Jobs& J, int prec, void* st) {
  if (!J.psrc.empty()) {
    const int n = (int)J.psrc.size();
    if (prec == 0) RUN(cartnet_gemm_pack_b(J.psrc.data(), J.pdst.data(), J.pK.data(), J.pN.data(), J.psk.data(), J.psn.data(), n, st));
    else RUN(cartnet_gemm_split_b(J.psrc.data(), J.pdst.data(), J.pK.data(), J.pN.data(), J.psk.data(), J.psn.data(), n, st));
  }
  for (size_t j0 = 0; j0 < J.tsrc.size(); j0 += 40) {
    const int n = (int)std::min<size_t>(40, J.tsrc.size() - j0);
    RUN(cartnet_transpose(J.tsrc.data() + j0, J.tdst.data() + j0, J.trows.data() + j0, J.tcols.data() + j0,
                          J.tlds.data() + j0, J.tldd.data() + j0, n, st));
  }
  return 0;
}

// group g of a forward product Y = X W^T: the DMA form (transposed copy + image) when images are in use, else W itself
inline void fwd_operand(CartnetGemmArgs& a, int g, const ConvW& cw, int f, bool use_img) {
  if (use_img) { a.B[g] = cw.T[f]; a.b_split[g] = cw.F[f]; }
  else a.B[g] = cw.W[f];
}
inline void fwd_form(CartnetGemmArgs& a, const ConvW& cw, int f, int C, bool use_img) {
  a.b_kstrided = use_img ? 1 : 0;
  a.ldb = use_img ? C : cw.ldw[f];
}

// column sums of x [R, C] (row stride ld) -> out[C] (fp64 partial rows, fixed order)
int colsum(const float* x, int ld, long long R, int C, double* parts, float* out, void* st) {
  RUN(cartnet_colsum_partial(x, ld, (int)R, C, parts, st));
  double* ps[1] = {parts};
  float* os[1] = {out};
  return cartnet_colsum_finalize(ps, os, 1, cartnet_segment_nparts((int)R), C, st);
}

}  // namespace

extern "C" size_t cartnet_icomformer_workspace_bytes(const CartnetIcfModel* model, int32_t N, int64_t E, int32_t Bg,
                                                     int32_t M) {
  if (!model || model->C < 8) return 0;
  size_t total = 0;
  icarve(*model, N, E, Bg, M, nullptr, &total);
  return total;
}

// ---- the attention block shared by ComformerConv and ComformerConv_edge (comformer.py: _Attention) ----------------------
namespace {
struct Att {
  int R, S;                       // rows, segments
  const int* segptr;              // [S + 1]
  const int *idx_i, *idx_j;       // gather indices of the two "node" terms per row
  const float *term_i, *term_j;   // [*, 2C]  (key | msg)
  const float* q;                 // [S, *] with leading dimension ldq
  int ldq;
  long long count;                // rows the bn_att statistics divide by
  int gparts, sparts;             // partial rows of the gate / segment kernels over S segments
  int grows;                      // upper bound of the rows of term_i / term_j
};

int att_forward(const CartnetIcfModel& m, const CartnetIcfConv& P, const CartnetIcfBn& bn_att, const ConvW& cw, const Att& t,
                int l, const float* rows_in, IWork& w, int training, void* st) {
  const int C = m.C, prec = m.gemm_precision;
  const bool afree = icf_alpha_free(C);
  {  // pr = rows_in F^T + c + term_i[idx_i] + term_j[idx_j]      (key | msg; F = W1[:, 2C:] We: lin_edge folded in)
    CartnetGemmArgs a = gargs(prec, t.R, C, C, C, C, 2 * C);
    a.ngroups = 2; fwd_form(a, cw, F_K1E, C, w.use_img);
    a.A[0] = rows_in; a.A[1] = rows_in;
    if (P.edge_b) { a.bias[0] = w.ck[l]; a.bias[1] = w.cm[l]; }
    fwd_operand(a, 0, cw, F_K1E, w.use_img); fwd_operand(a, 1, cw, F_M1E, w.use_img);
    a.C[0] = w.pr[l]; a.C[1] = w.pr[l] + C;
    a.gather_i[0] = t.term_i; a.gather_i[1] = t.term_i + C; a.gather_j[0] = t.term_j; a.gather_j[1] = t.term_j + C;
    a.ldg = 2 * C; a.tgt = t.idx_i; a.src = t.idx_j; a.gather_rows = t.grows;
    RUN(cartnet_gemm(&a, st));
  }
  {  // key' = silu(pr_k) W2k^T + b  -> keyb[:, :C];  msg = silu(pr_m) W2m^T + b -> gs[:, C:]
    CartnetGemmArgs a = gargs(prec, t.R, C, C, 2 * C, C, 2 * C);
    a.ngroups = 2; a.a_act = 1; fwd_form(a, cw, F_K2, C, w.use_img);
    a.A[0] = w.pr[l]; a.A[1] = w.pr[l] + C;
    fwd_operand(a, 0, cw, F_K2, w.use_img); fwd_operand(a, 1, cw, F_M2, w.use_img);
    a.C[0] = afree ? w.gs[l] : w.keyb[l]; a.C[1] = w.gs[l] + C; a.bias[0] = P.key2_b; a.bias[1] = P.msg2_b;
    if (w.act[l]) { a.a_act_out[0] = w.act[l]; a.a_act_out[1] = w.act[l] + C; }
    RUN(cartnet_gemm(&a, st));
  }
  const float scale = 1.0f / sqrtf((float)C);
  RUN(cartnet_rowmul_fwd(afree ? w.gs[l] : w.keyb[l], 2 * C, t.q, t.ldq, t.segptr, t.S, C, scale, afree ? nullptr : w.gs[l],
                         2 * C, w.pa, w.pb, st));
  RUN(cartnet_bn_finalize(w.pa, w.pb, t.sparts, t.count, C, m.bn_eps, m.bn_momentum, training, bn_att.mean, bn_att.var,
                          bn_att.nbt, w.mr1[l], nullptr, 1, 1, st));
  if (afree)
    RUN(cartnet_att_gate_fwd(w.gs[l], t.q, t.ldq, t.segptr, w.mr1[l], P.bn_att_w, P.bn_att_b, scale, t.S, C, w.aggr[l],
                             w.bc[l], st));     // (in eval mode too: the sums identity does not depend on the mode)
  else if (training && w.bc[l])
    RUN(cartnet_gate_scatter_fwd_bc(w.gs[l], nullptr, nullptr, t.segptr, w.mr1[l], P.bn_att_w, P.bn_att_b, t.S, C, nullptr,
                                    w.aggr[l], w.pc, w.pd, w.bc[l], st));
  else
    RUN(cartnet_gate_scatter_fwd(w.gs[l], nullptr, nullptr, t.segptr, w.mr1[l], P.bn_att_w, P.bn_att_b, t.S, C, nullptr,
                                 w.aggr[l], w.pc, w.pd, nullptr, st));
  return 0;
}
}  // namespace

extern "C" int cartnet_icomformer_forward(const CartnetIcfModel* model, const CartnetBatch* batch, const float* cell,
                                          void* workspace, size_t workspace_bytes, int32_t training, float* pred,
                                          float* x_out, int32_t* status, void* st, void* aux_stream) {
  RUN(check_icf(model, batch, "cartnet_icomformer_forward"));
  const CartnetIcfModel& m = *model;
  const CartnetBatch& b = *batch;
  const CartnetIcfParams& P = m.p;
  CN_CHECK(workspace && pred && x_out && status && cell, "cartnet_icomformer_forward: null argument");
  size_t need = 0;
  IWork w = icarve(m, b.N, b.E, b.Bg, b.M, static_cast<char*>(workspace), &need);
  CN_CHECK(workspace_bytes >= need, "cartnet_icomformer_forward: workspace %zu < required %zu bytes", workspace_bytes, need);
  CN_CHECK((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "cartnet_icomformer_forward: workspace must be 256-byte aligned");
  const int C = m.C, H = C / 2, N = b.N, Bg = b.Bg, prec = m.gemm_precision;
  const int E = (int)b.E;
  g_icf_events.next = 0;
  Streams S{(hipStream_t)st, aux_stream ? (hipStream_t)aux_stream : (hipStream_t)st, aux_stream != nullptr && aux_stream != st,
            &g_icf_events};
  void* sw = (void*)S.side;
  if (S.fork() != 0) { cartnet_set_error("cartnet_icomformer_forward: stream fork failed"); return 2; }

  // ---- weight forms of the step: every transposed copy and every image, two batched launches (side stream)
  Jobs& J = g_jobs;
  J.clear();
  for (int l = 0; l < 5; ++l) {
    const CartnetIcfConv& q = l < 4 ? P.att[l] : P.edge;
    {  // F{k,m} = W1{k,m}[:, 2C:] We   (C x C x C each, two groups)
      CartnetGemmArgs a = gargs(0, C, C, C, 3 * C, C, C);
      a.ngroups = 2; a.b_kstrided = 1; a.tile_policy = 0;
      a.A[0] = q.key0_w + 2 * C; a.A[1] = q.msg0_w + 2 * C; a.B[0] = q.edge_w; a.B[1] = q.edge_w;
      a.C[0] = w.Fk[l]; a.C[1] = w.Fm[l];
      RUN(cartnet_gemm(&a, sw));
    }
    if (q.edge_b) {   // c{k,m} = W1{k,m}[:, 2C:] be   (one row times W^T)
      CartnetGemmArgs a = gargs(0, 1, C, C, C, 3 * C, C);
      a.ngroups = 2; a.tile_policy = 0;
      a.A[0] = q.edge_b; a.A[1] = q.edge_b; a.B[0] = q.key0_w + 2 * C; a.B[1] = q.msg0_w + 2 * C;
      a.C[0] = w.ck[l]; a.C[1] = w.cm[l];
      RUN(cartnet_gemm(&a, sw));
    }
    plan_conv(q, w.Fk[l], w.Fm[l], C, prec, w.use_img, w.cw[l], J);
  }
  if (w.use_img) {
    for (int i = 0; i < 2; ++i) {
      const float* W_ = i ? P.rbf_angle_w : P.rbf_w;
      J.psrc.push_back(W_); J.pdst.push_back(i ? w.rbfaF : w.rbfF); J.pK.push_back(C); J.pN.push_back(C); J.psk.push_back(1); J.psn.push_back(C);
      J.tsrc.push_back(W_); J.tdst.push_back(i ? w.rbfaT : w.rbfT); J.trows.push_back(C); J.tcols.push_back(C); J.tlds.push_back(C); J.tldd.push_back(C);
    }
    if (H % 16 == 0) {      // head0 [H, C] as the backward operand [K = H, N = C]
      J.psrc.push_back(P.head0_w); J.pdst.push_back(w.head0B); J.pK.push_back(H); J.pN.push_back(C); J.psk.push_back(C); J.psn.push_back(1);
    }
  }
  RUN(flush_jobs(J, prec, sw));
  hipEvent_t forms_ready = S.mark_side();

  // ---- graph layout, embeddings, lattice features (comformer.py:116-124)
  RUN(cartnet_csr_build(b.edge_index, b.E, N, b.graph_ptr, Bg, w.src32, w.tgt32, w.rowptr, w.colptr, w.perm, status, st));
  RUN(cartnet_node_embed(b.z, b.batch, b.temperature, P.embedding, P.temp_w, P.temp_b, nullptr, N, C, m.n_types, Bg,
                         status, w.x0, st));
  RUN(cartnet_lattice_features(cell, b.batch, w.src32, b.cart_dist, b.cart_dir, b.E, Bg, w.edge_feat, w.nl, w.nc, st));
  {
    long long blocks = (std::max<long long>(b.E + 1, Bg + 1) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cn_icf_index_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)st, w.src32, b.batch,
                       (long long)b.E, b.graph_ptr, w.rowptr, Bg, w.idx_edge, w.idx_gl, w.ptr3, w.gedge_ptr);
    CN_LAUNCH_CHECK("cartnet_icomformer_forward (index kernel)");
  }
  if (S.main_waits(forms_ready) != 0) { cartnet_set_error("cartnet_icomformer_forward: stream wait failed"); return 2; }

  // out = softplus(rbf(vals) W^T + b): keeps (r, pre) for backward
  auto rbf_branch = [&](const float* vals, long long n, const float* centers, float gamma, const float* W_, const float* T_,
                        const char* F_, const float* bias, float* r, float* pre, float* out) -> int {
    RUN(cartnet_rbf_expand(vals, n, centers, C, gamma, r, C, st));
    CartnetGemmArgs a = gargs(prec, (int)n, C, C, C, C, C);
    const bool img = w.use_img && n >= 2048;      // a handful of rows (lattice lengths): the plain NT form
    a.A[0] = r; a.C[0] = pre; a.bias[0] = bias;
    if (img) { a.b_kstrided = 1; a.B[0] = T_; a.b_split[0] = F_; } else { a.B[0] = W_; }
#ifndef CN_ICF_NO_RBF_FUSE
    // pre and softplus(pre) from the product's epilogue (CartnetGemmArgs.dact_kind = 1 with cpre + out_act): the element-wise
    // pass read pre and wrote out, 0.25 ms per step over the E + 3E rows
    a.C[0] = out; a.cpre[0] = pre; a.out_act = 1; a.dact_kind = 1;
    return cartnet_gemm(&a, st);
#else
    RUN(cartnet_gemm(&a, st));
    return cartnet_eltwise(0, pre, nullptr, out, n, C, C, 0, C, 1.0f, st);
#endif
  };
  RUN(rbf_branch(w.edge_feat, b.E, m.rbf_centers, m.gamma_rbf, P.rbf_w, w.rbfT, w.rbfF, P.rbf_b, w.r_e, w.pre_e, w.e0));
  RUN(rbf_branch(w.nl, (long long)Bg * 3, m.rbf_centers, m.gamma_rbf, P.rbf_w, w.rbfT, w.rbfF, P.rbf_b, w.r_nl, w.pre_nl, w.NLt));
  RUN(rbf_branch(w.nc, 3LL * b.E, m.rbf_angle_centers, m.gamma_angle, P.rbf_angle_w, w.rbfaT, w.rbfaF, P.rbf_angle_b,
                 w.r_na, w.pre_na, w.NA));

  // ---- ComformerConv (comformer_conv.py:71-99)
  auto conv = [&](int l, const float* x, const float* e, float* y) -> int {
    const CartnetIcfConv& q = P.att[l];
    const ConvW& cw = w.cw[l];
    {  // q | k | v
      CartnetGemmArgs a = gargs(prec, N, C, C, C, C, 3 * C);
      a.ngroups = 3; fwd_form(a, cw, F_Q, C, w.use_img);
      for (int g = 0; g < 3; ++g) { a.A[g] = x; a.C[g] = w.QKV[l] + g * C; fwd_operand(a, g, cw, F_Q + g, w.use_img); }
      a.bias[0] = q.query_b; a.bias[1] = q.key_b; a.bias[2] = q.value_b;
      RUN(cartnet_gemm(&a, st));
    }
    {  // node terms of key_update.0 / lin_msg_update.0: KPi = [k W1k_i^T + b | v W1m_i^T + b], KPj = [k W1k_j^T | v W1m_j^T]
      CartnetGemmArgs a = gargs(prec, N, C, C, 3 * C, C, 2 * C);
      a.ngroups = 4; fwd_form(a, cw, F_K1I, C, w.use_img);
      const float* k = w.QKV[l] + C; const float* v = w.QKV[l] + 2 * C;
      a.A[0] = k; a.A[1] = v; a.A[2] = k; a.A[3] = v;
      a.C[0] = w.KPi[l]; a.C[1] = w.KPi[l] + C; a.C[2] = w.KPj[l]; a.C[3] = w.KPj[l] + C;
      fwd_operand(a, 0, cw, F_K1I, w.use_img); fwd_operand(a, 1, cw, F_M1I, w.use_img);
      fwd_operand(a, 2, cw, F_K1J, w.use_img); fwd_operand(a, 3, cw, F_M1J, w.use_img);
      a.bias[0] = q.key0_b; a.bias[1] = q.msg0_b;
      RUN(cartnet_gemm(&a, st));
    }
    Att t{E, N, w.rowptr, w.tgt32, w.src32, w.KPi[l], w.KPj[l], w.QKV[l], 3 * C, (long long)b.E, w.gp_n, w.sp_n, N};
    RUN(att_forward(m, q, m.att_bn_att[l], cw, t, l, e, w, training, st));
    {  // o = lin_concate(aggr) with the BatchNorm statistics over atoms
      CartnetGemmArgs a = gargs(prec, N, C, C, C, C, C);
      fwd_form(a, cw, F_CAT, C, w.use_img);
      a.A[0] = w.aggr[l]; a.C[0] = w.o[l]; a.bias[0] = q.concate_b; fwd_operand(a, 0, cw, F_CAT, w.use_img);
      a.colsum[0] = w.cs; a.colsq[0] = w.cq;
      RUN(cartnet_gemm(&a, st));
    }
    RUN(cartnet_bn_finalize(w.cs, w.cq, w.tiles_n, N, C, m.bn_eps, m.bn_momentum, training, m.att_bn[l].mean,
                            m.att_bn[l].var, m.att_bn[l].nbt, w.mr2[l], nullptr, 1, 0, st));
    return cartnet_softplus_update_fwd(w.o[l], x, w.mr2[l], q.bn_w, q.bn_b, N, C, y, st);
  };

  // ---- ComformerConv_edge (comformer_conv.py:156-193) on rows r = 3 e + lattice vector
  auto conv_edge = [&](const float* e, float* y) -> int {
    const int l = 4;
    const CartnetIcfConv& q = P.edge;
    const ConvW& cw = w.cw[l];
    {
      CartnetGemmArgs a = gargs(prec, E, C, C, C, C, 3 * C);
      a.ngroups = 3; fwd_form(a, cw, F_Q, C, w.use_img);
      for (int g = 0; g < 3; ++g) { a.A[g] = e; a.C[g] = w.QKV[l] + g * C; fwd_operand(a, g, cw, F_Q + g, w.use_img); }
      a.bias[0] = q.query_b; a.bias[1] = q.key_b; a.bias[2] = q.value_b;
      RUN(cartnet_gemm(&a, st));
    }
    for (int kv = 0; kv < 2; ++kv) {   // lin_key_e{i} / lin_value_e{i} on the lattice-length features: [Bg, 3C] views
      CartnetGemmArgs a = gargs(prec, Bg, C, C, 3 * C, C, 3 * C);
      a.ngroups = 3;
      for (int i = 0; i < 3; ++i) {
        a.A[i] = w.NLt + i * C; a.B[i] = kv ? q.value_e_w[i] : q.key_e_w[i]; a.bias[i] = kv ? q.value_e_b[i] : q.key_e_b[i];
        a.C[i] = (kv ? w.VY : w.KY) + i * C;
      }
      RUN(cartnet_gemm(&a, st));
    }
    {  // per-edge term
      CartnetGemmArgs a = gargs(prec, E, C, C, 3 * C, C, 2 * C);
      a.ngroups = 2; fwd_form(a, cw, F_K1I, C, w.use_img);
      a.A[0] = w.QKV[l] + C; a.A[1] = w.QKV[l] + 2 * C; a.C[0] = w.Ka; a.C[1] = w.Ka + C;
      fwd_operand(a, 0, cw, F_K1I, w.use_img); fwd_operand(a, 1, cw, F_M1I, w.use_img);
      a.bias[0] = q.key0_b; a.bias[1] = q.msg0_b;
      RUN(cartnet_gemm(&a, st));
    }
    {  // per (crystal, lattice vector) term: a handful of rows, the plain NT form
      CartnetGemmArgs a = gargs(prec, Bg * 3, C, C, C, 3 * C, 2 * C);
      a.ngroups = 2;
      a.A[0] = w.KY; a.A[1] = w.VY; a.B[0] = q.key0_w + C; a.B[1] = q.msg0_w + C; a.C[0] = w.KYb; a.C[1] = w.KYb + C;
      RUN(cartnet_gemm(&a, st));
    }
    Att t{3 * E, E, w.ptr3, w.idx_edge, w.idx_gl, w.Ka, w.KYb, w.QKV[l], 3 * C, 3LL * b.E, w.gp_e, w.sp_e, E > 3 * Bg ? E : 3 * Bg};
    RUN(att_forward(m, q, m.edge_bn_att, cw, t, l, w.NA, w, training, st));
    RUN(cartnet_eltwise(3, q.concate_b, nullptr, w.bias3, 1, C, C, 0, C, 3.0f, st));
    {
      CartnetGemmArgs a = gargs(prec, E, C, C, C, C, C);
      fwd_form(a, cw, F_CAT, C, w.use_img);
      a.A[0] = w.aggr[l]; a.C[0] = w.o[l]; a.bias[0] = w.bias3; fwd_operand(a, 0, cw, F_CAT, w.use_img);
      a.colsum[0] = w.cs; a.colsq[0] = w.cq;
      RUN(cartnet_gemm(&a, st));
    }
    RUN(cartnet_bn_finalize(w.cs, w.cq, w.tiles_e, b.E, C, m.bn_eps, m.bn_momentum, training, m.edge_bn.mean, m.edge_bn.var,
                            m.edge_bn.nbt, w.mr2[l], nullptr, 1, 0, st));
    return cartnet_softplus_update_fwd(w.o[l], e, w.mr2[l], q.bn_w, q.bn_b, E, C, y, st);
  };

  RUN(conv(0, w.x0, w.e0, w.y[0]));
  RUN(conv_edge(w.e0, w.y[4]));
  RUN(conv(1, w.y[0], w.y[4], w.y[1]));
  RUN(conv(2, w.y[1], w.y[4], w.y[2]));
  RUN(conv(3, w.y[2], w.y[4], x_out));

  // ---- Cholesky head (shared with CartNet)
  {
    CartnetGemmArgs a = gargs(prec, N, H, C, C, C, H);
    a.A[0] = x_out; a.B[0] = P.head0_w; a.C[0] = w.hid; a.bias[0] = P.head0_b;
    RUN(cartnet_gemm(&a, st));
  }
  RUN(cartnet_mask_index(b.non_h_mask, N, w.idx, nullptr, st));
  RUN(cartnet_cholesky_head_fwd(w.hid, w.idx, P.head2_w, P.head2_b, N, H, w.p6, pred, st));
  // join (the side stream carried only the weight forms)
  if (S.main_waits(S.mark_side()) != 0) { cartnet_set_error("cartnet_icomformer_forward: stream join failed"); return 2; }
  return 0;
}

extern "C" int cartnet_icomformer_backward(const CartnetIcfModel* model, const CartnetBatch* batch, void* workspace,
                                           size_t workspace_bytes, int32_t training, const float* dpred,
                                           const float* x_out, const CartnetIcfParams* grads, void* stream,
                                           void* aux_stream) {
  RUN(check_icf(model, batch, "cartnet_icomformer_backward"));
  const CartnetIcfModel& m = *model;
  const CartnetBatch& b = *batch;
  const CartnetIcfParams& P = m.p;
  CN_CHECK(workspace && dpred && x_out && grads, "cartnet_icomformer_backward: null argument");
  const CartnetIcfParams& G = *grads;
  size_t need = 0;
  IWork w = icarve(m, b.N, b.E, b.Bg, b.M, static_cast<char*>(workspace), &need);
  CN_CHECK(workspace_bytes >= need, "cartnet_icomformer_backward: workspace %zu < required %zu bytes", workspace_bytes, need);
  const int C = m.C, H = C / 2, N = b.N, Bg = b.Bg, prec = m.gemm_precision;
  const int E = (int)b.E;
  for (int l = 0; l < 5; ++l) {        // the views (the forms themselves were built by forward and are still in the workspace)
    Jobs dummy;
    plan_conv(l < 4 ? P.att[l] : P.edge, w.Fk[l], w.Fm[l], C, prec, false, w.cw[l], dummy);
  }
  g_icf_events.next = 0;
  Streams S{(hipStream_t)stream, aux_stream ? (hipStream_t)aux_stream : (hipStream_t)stream,
            aux_stream != nullptr && aux_stream != stream, &g_icf_events};
  void* st = stream;
  void* sw = (void*)S.side;
#define FORK() do { if (S.fork() != 0) { cartnet_set_error("cartnet_icomformer_backward: stream fork failed"); return 2; } } while (0)
#define JOIN() do { if (S.main_waits(S.mark_side()) != 0) { cartnet_set_error("cartnet_icomformer_backward: stream join failed"); return 2; } } while (0)
  // weight gradient on the side stream: everything it reads was queued on the main stream before this point
  auto wg = [&](std::initializer_list<const float*> dY, int ldy, std::initializer_list<const float*> X, int ldx,
                std::initializer_list<float*> outs, int ldo, long long K, int M_, int N_, bool b_act = false) -> int {
    FORK();
    return iwgrad(prec, dY.begin(), ldy, X.begin(), ldx, outs.begin(), ldo, K, M_, N_, (int)dY.size(), b_act, w, sw);
  };
  // dX (+ resid) = dY W with W [K = out, N = in] (leading dimension ldw) and its backward image if there is one
  auto dgemm = [&](const float* dY, int ldy, const float* W_, int ldw, const char* img, float* dX, int ldx, long long rows,
                   int Kout, int Nin, const float* resid, int ldr, void* s_, double* colparts = nullptr) -> int {
    CartnetGemmArgs a = gargs(prec, (int)rows, Nin, Kout, ldy, ldw, ldx);
    a.b_kstrided = 1; a.A[0] = dY; a.B[0] = W_; a.C[0] = dX; a.resid[0] = resid; a.ldr = ldr;
    a.colsum[0] = colparts;                                   // [tiles_m(rows)][Nin] partial column sums of dX
    if (img && rows >= 2048) a.b_split[0] = img;
    return cartnet_gemm(&a, s_);
  };

  // ---- head
  const int head_row = 7 * H + 8;
  RUN(cartnet_cholesky_head_bwd(w.hid, w.idx, P.head2_w, w.p6, dpred, N, H, w.dhid, w.head_parts, st));
  RUN(cartnet_colsum_finalize_f32(w.head_parts, w.nparts_n, head_row, w.head_tot, st));
  if (hipMemcpyAsync(G.head2_w, w.head_tot, sizeof(float) * 6 * H, hipMemcpyDeviceToDevice, (hipStream_t)st) != hipSuccess ||
      hipMemcpyAsync(G.head2_b, w.head_tot + 6 * H, sizeof(float) * 6, hipMemcpyDeviceToDevice, (hipStream_t)st) != hipSuccess ||
      hipMemcpyAsync(G.head0_b, w.head_tot + 6 * H + 8, sizeof(float) * H, hipMemcpyDeviceToDevice, (hipStream_t)st) != hipSuccess) {
    cartnet_set_error("cartnet_icomformer_backward: head gradient copy failed");
    return 2;
  }
  RUN(wg({w.dhid}, H, {x_out}, C, {G.head0_w}, C, N, H, C));
  RUN(dgemm(w.dhid, H, P.head0_w, C, (w.use_img && H % 16 == 0) ? w.head0B : nullptr, w.dx_head, C, N, H, C, nullptr, 0, st));

  // backward of y = softplus(x_in + bn(o)): d_o[l], dres[l]; the BatchNorm affine gradients are the two column sums
  // (round 5: the bias gradients that are column sums of a tensor an element-wise pass of this file writes come out of
  //  that pass -- d_o here, dkey / dq in cartnet_rowmul_bwd_sums, the RBF branches' dpre -- and those of a tensor a GEMM
  //  writes out of its epilogue; they were 34 cartnet_colsum_partial passes per step, 1.5 ms of the main stream's work)
#ifdef CN_ICF_NO_BIAS_FUSE
  const bool bias_fuse = false;
#else
  const bool bias_fuse = C <= 256;
#endif
#ifdef CN_ICF_NO_RBF_FUSE
  const bool rbf_fuse = false;
#else
  const bool rbf_fuse = bias_fuse;
#endif
  auto softplus_bwd = [&](int l, const CartnetIcfConv& q, const CartnetIcfConv& g, const float* dy, int rows,
                          const float* x_in, float* concate_b) -> int {
    RUN(cartnet_softplus_update_bwd_stats(w.o[l], x_in, dy, w.mr2[l], q.bn_w, q.bn_b, rows, C, w.pa, w.pb, st));
    double* parts[2] = {w.pa, w.pb};
    float* outs[2] = {w.sums2[l], w.sums2[l] + C};
    float* gr[2] = {g.bn_b, g.bn_w};
    RUN(cartnet_colsum_finalize2(parts, outs, gr, 2, cartnet_segment_nparts(rows), C, st));
    if (!bias_fuse) {
      RUN(cartnet_softplus_update_bwd_apply(w.o[l], x_in, dy, w.mr2[l], q.bn_w, q.bn_b, w.sums2[l], training, rows, C,
                                            w.d_o[l], nullptr, w.dres[l], st));
      return colsum(w.d_o[l], C, rows, C, w.pa, concate_b, st);
    }
    RUN(cartnet_softplus_update_bwd_apply_sums(w.o[l], x_in, dy, w.mr2[l], q.bn_w, q.bn_b, w.sums2[l], training, rows, C,
                                               w.d_o[l], nullptr, w.dres[l], w.pa, st));
    double* bp[1] = {w.pa};
    float* bo[1] = {concate_b};
    return cartnet_colsum_finalize(bp, bo, 1, cartnet_segment_nparts(rows), C, st);
  };

  // backward of the attention block: consumes gs[l]; leaves dpr[l] (gradient at the first Linears' pre-activation,
  // key | msg), dq in dq_out (leading dimension 3C) and the gradients of the second Linears / bn_att / the row blocks
  auto att_backward = [&](int l, const CartnetIcfConv& q, const CartnetIcfConv& g, const Att& t, const float* daggr,
                          float* dq_out) -> int {
    const ConvW& cw = w.cw[l];
    float* gs = w.gs[l];
    // sum(dbn), sum(dbn ghat): no edge residual here, so with the per-target sums of the forward pass (conv layers) both
    // are sums over the TARGETS -- 25 MB instead of a 363 MB pass over gs
    const bool fused_sums = w.bc[l] != nullptr && (training || icf_alpha_free(C));
    if (fused_sums)
      RUN(cartnet_coldot_bc_partial(daggr, C, w.bc[l], t.S, C, w.pa, w.pb, st));
    else
      RUN(cartnet_gate_scatter_bwd_stats(gs, nullptr, daggr, nullptr, t.segptr, w.mr1[l], q.bn_att_w, q.bn_att_b, t.S, C, w.pa,
                                         w.pb, nullptr, st));
    {
      double* parts[2] = {w.pa, w.pb};
      float* outs[2] = {w.sums1[l], w.sums1[l] + C};
      float* gr[2] = {g.bn_att_b, g.bn_att_w};
      RUN(cartnet_colsum_finalize2(parts, outs, gr, 2, fused_sums ? cartnet_segment_nparts(t.S) : t.gparts, C, st));
    }
    const float scale = 1.0f / sqrtf((float)C);
#ifdef CN_ICF_NO_GATE_ROWMUL
    const bool gate_rowmul = false;
#else
    const bool gate_rowmul = bias_fuse;
#endif
    CN_CHECK(gate_rowmul || !icf_alpha_free(C), "cartnet_icomformer_backward: the alpha-free forward needs the fused gate backward");
    if (gate_rowmul) {
      // gate backward + query x key backward + the three bias gradients in one pass: gs = [dkey | dmsg], dq
      RUN(cartnet_att_gate_bwd_apply(gs, icf_alpha_free(C) ? nullptr : w.keyb[l], 2 * C, t.q, t.ldq, daggr, t.segptr, w.mr1[l], q.bn_att_w, q.bn_att_b,
                                     w.sums1[l], t.count, training, scale, t.S, C, dq_out, 3 * C, w.pa, w.pd, w.pb, st));
      double* parts[3] = {w.pa, w.pd, w.pb};
      float* outs[3] = {g.key2_b, g.msg2_b, g.query_b};
      RUN(cartnet_colsum_finalize(parts, outs, 3, cartnet_segment_nparts(t.S), C, st));
    } else {
      RUN(cartnet_gate_scatter_bwd_apply(gs, nullptr, daggr, nullptr, t.segptr, w.mr1[l], q.bn_att_w, q.bn_att_b, w.sums1[l],
                                         t.count, training, t.S, C, w.pc, w.pd, nullptr, st));     // gs = [dalpha | dmsg]
      {
        double* parts[1] = {w.pd};
        float* outs[1] = {g.msg2_b};
        RUN(cartnet_colsum_finalize(parts, outs, 1, t.gparts, C, st));
      }
      if (bias_fuse) {    // gs = [dkey | dmsg]; key_update.2's and lin_query's bias gradients from the same pass
        RUN(cartnet_rowmul_bwd_sums(gs, 2 * C, w.keyb[l], 2 * C, t.q, t.ldq, t.segptr, t.S, C, scale, dq_out, 3 * C, w.pa, w.pb, st));
        double* parts[2] = {w.pa, w.pb};
        float* outs[2] = {g.key2_b, g.query_b};
        RUN(cartnet_colsum_finalize(parts, outs, 2, cartnet_segment_nparts(t.S), C, st));
      } else {
        RUN(cartnet_rowmul_bwd(gs, 2 * C, w.keyb[l], 2 * C, t.q, t.ldq, t.segptr, t.S, C, scale, dq_out, 3 * C, st));
        RUN(colsum(gs, 2 * C, t.R, C, w.pa, g.key2_b, st));
        RUN(colsum(dq_out, 3 * C, t.S, C, w.pa, g.query_b, st));
      }
    }
    if (w.act[l]) RUN(wg({gs, gs + C}, 2 * C, {w.act[l], w.act[l] + C}, 2 * C, {g.key2_w, g.msg2_w}, C, t.R, C, C, false));
    else RUN(wg({gs, gs + C}, 2 * C, {w.pr[l], w.pr[l] + C}, 2 * C, {g.key2_w, g.msg2_w}, C, t.R, C, C, true));
    {  // dpr = (gs W2) * silu'(pr), column sums = bias gradients of the first Linears
      CartnetGemmArgs a = gargs(prec, t.R, C, C, 2 * C, C, 2 * C);
      a.ngroups = 2; a.b_kstrided = 1;
      a.A[0] = gs; a.A[1] = gs + C; a.B[0] = q.key2_w; a.B[1] = q.msg2_w;
      a.C[0] = w.dpr[l]; a.C[1] = w.dpr[l] + C; a.dact[0] = w.pr[l]; a.dact[1] = w.pr[l] + C; a.ldd = 2 * C;
      a.colsum[0] = w.cs; a.colsum[1] = w.cq;
      if (w.use_img) { a.b_split[0] = cw.B[B_K2]; a.b_split[1] = cw.B[B_M2]; }
      RUN(cartnet_gemm(&a, st));
      double* parts[2] = {w.cs, w.cq};
      float* outs[2] = {g.key0_b, g.msg0_b};
      RUN(cartnet_colsum_finalize(parts, outs, 2, tiles_m(t.R), C, st));
    }
    return 0;
  };

  // The folded lin_edge: d(rows) = [dpr_k | dpr_m] [Fk; Fm] (+ resid), dF = dpr^T rows (the one row-sized weight gradient),
  // then the C x C chain rule back to W1[:, 2C:], We and be.  All on the side stream.
  // rows_ready[l]: d(rows) of layer l is complete on the side stream -- what the main stream waits for where it reads it
  // (a full join there also waited for the C x C gradient products queued behind it: 1.0 + 1.5 ms of the step)
  hipEvent_t rows_ready[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // sp_pre (round 5): the rows are the output of an RBF branch, rows = softplus(sp_pre), and nothing but that branch's
  // backward reads d(rows): the product writes d(sp_pre) = d(rows) * sigmoid(sp_pre) (CartnetGemmArgs.dact_kind = 1) and the
  // column partials of it -- the branch's bias gradient sp_bias -- so the element-wise pass and the column-sum pass over
  // [R, C] (3E rows for the angle branch) are gone from the main stream.
  auto fold_backward = [&](int l, const CartnetIcfConv& q, const CartnetIcfConv& g, const float* rows_in, long long R,
                           float* d_rows, const float* resid, const float* sp_pre = nullptr, float* sp_bias = nullptr) -> int {
    const ConvW& cw = w.cw[l];
    float* dpr = w.dpr[l];
    FORK();
    {
      CartnetGemmArgs a = gargs(prec, (int)R, C, C, 2 * C, C, C);
      a.nsegs = 2; a.b_kstrided = 1;
      a.A[0] = dpr; a.A[1] = dpr + C; a.B[0] = w.Fk[l]; a.B[1] = w.Fm[l]; a.C[0] = d_rows; a.resid[0] = resid; a.ldr = C;
      if (w.use_img) a.b_split_folded = cw.B[B_E1];
      if (sp_pre) { a.dact[0] = sp_pre; a.ldd = C; a.dact_kind = 1; a.colsum[0] = w.cs_side; }
      // (round 6: this side-stream launch is persistent and owns every CU for ~390 us while atom-sized kernels of the main
      //  stream wait for a slot; forcing the tile kernels here (tile_policy = 256) measured 26.00 against 25.73-25.76 ms)
      RUN(cartnet_gemm(&a, sw));
      rows_ready[l] = S.mark_side();
      if (sp_pre) {
        double* parts[1] = {w.cs_side};
        float* outs[1] = {sp_bias};
        RUN(cartnet_colsum_finalize(parts, outs, 1, tiles_m(R), C, sw));
      }
    }
    {
      const float* dY[2] = {dpr, dpr + C};
      const float* X[2] = {rows_in, rows_in};
      float* o[2] = {w.dFk[l], w.dFm[l]};
      RUN(iwgrad(prec, dY, 2 * C, X, C, o, C, R, C, C, 2, false, w, sw));
    }
    {  // dW1{k,m}[:, 2C:] = dF We^T
      CartnetGemmArgs a = gargs(0, C, C, C, C, C, 3 * C);
      a.ngroups = 2; a.tile_policy = 0;
      a.A[0] = w.dFk[l]; a.A[1] = w.dFm[l]; a.B[0] = q.edge_w; a.B[1] = q.edge_w;
      a.C[0] = g.key0_w + 2 * C; a.C[1] = g.msg0_w + 2 * C;
      RUN(cartnet_gemm(&a, sw));
    }
    if (q.edge_b) {   // ... + db be^T  (db = the bias gradients of the first Linears, already summed on the main stream)
      hipLaunchKernelGGL(cn_icf_rank1_kernel, dim3(C, 2), dim3(256), 0, (hipStream_t)sw, g.key0_w + 2 * C, g.msg0_w + 2 * C, 3 * C,
                         g.key0_b, g.msg0_b, q.edge_b, C);
      CN_LAUNCH_CHECK("cartnet_icomformer_backward (rank-1 update)");
    }
    {  // dWe = W1k[:, 2C:]^T dFk + W1m[:, 2C:]^T dFm
      CartnetGemmArgs a = gargs(0, C, C, C, 3 * C, C, C);
      a.nsegs = 2; a.a_kstrided = 1; a.b_kstrided = 1; a.tile_policy = 0;
      a.A[0] = q.key0_w + 2 * C; a.A[1] = q.msg0_w + 2 * C; a.B[0] = w.dFk[l]; a.B[1] = w.dFm[l]; a.C[0] = g.edge_w;
      RUN(cartnet_gemm(&a, sw));
    }
    if (q.edge_b) {   // dbe = W1k[:, 2C:]^T db_k + W1m[:, 2C:]^T db_m   (a row times W)
      hipLaunchKernelGGL(cn_icf_rowvec2_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)sw, g.edge_b, g.key0_b,
                         q.key0_w + 2 * C, g.msg0_b, q.msg0_w + 2 * C, 3 * C, C);
      CN_LAUNCH_CHECK("cartnet_icomformer_backward (dbe)");
    }
    return 0;
  };

  // d(inp) + resid of the query / key / value Linears that share the input `inp`
  // (lin_query's bias gradient: att_backward; lin_key's / lin_value's: the callers, from the products that write dk / dv)
  auto linear3_bwd = [&](int l, const CartnetIcfConv& q, const CartnetIcfConv& g, const float* inp, int rows, float* d_in) -> int {
    float* dQ = w.dQKV[l];
    RUN(wg({dQ, dQ + C, dQ + 2 * C}, 3 * C, {inp, inp, inp}, C, {g.query_w, g.key_w, g.value_w}, C, rows, C, C));
    CartnetGemmArgs a = gargs(prec, rows, C, C, 3 * C, C, C);
    a.nsegs = 3; a.b_kstrided = 1;
    a.A[0] = dQ; a.A[1] = dQ + C; a.A[2] = dQ + 2 * C; a.B[0] = q.query_w; a.B[1] = q.key_w; a.B[2] = q.value_w;
    a.C[0] = d_in; a.resid[0] = w.dres[l]; a.ldr = C;
    if (w.use_img && rows >= 2048) a.b_split_folded = w.cw[l].B[B_QKV];
    return cartnet_gemm(&a, st);
  };

  // ---- ComformerConv backward: dx[l], de[l] (de on the side stream; de_acc is added in its last product's epilogue)
  auto conv_bwd = [&](int l, const float* dy, const float* de_acc) -> int {
    const CartnetIcfConv& q = P.att[l];
    const CartnetIcfConv& g = G.att[l];
    const ConvW& cw = w.cw[l];
    const float* x_in = l == 0 ? w.x0 : w.y[l - 1];
    const float* e_in = l == 0 ? w.e0 : w.y[4];
    RUN(softplus_bwd(l, q, g, dy, N, x_in, g.concate_b));
    RUN(wg({w.d_o[l]}, C, {w.aggr[l]}, C, {g.concate_w}, C, N, C, C));
    RUN(dgemm(w.d_o[l], C, q.concate_w, C, w.use_img ? cw.B[B_CAT] : nullptr, w.daggr[l], C, N, C, C, nullptr, 0, st));
    Att t{E, N, w.rowptr, w.tgt32, w.src32, w.KPi[l], w.KPj[l], w.QKV[l], 3 * C, (long long)b.E, w.gp_n, w.sp_n, N};
    RUN(att_backward(l, q, g, t, w.daggr[l], w.dQKV[l]));
    float* dpr = w.dpr[l];
    // lin_edge (folded into the row block): nothing on the chain of atom gradients reads de before the edge layer's
    // backward (or the end), so d(e) and the C x C gradients run on the side stream
    if (l == 0 && rbf_fuse) RUN(fold_backward(l, q, g, e_in, b.E, w.dpre_e, de_acc, w.pre_e, w.gb1));   // e0 = softplus(pre_e)
    else RUN(fold_backward(l, q, g, e_in, b.E, w.de[l], de_acc));
    // node terms: reduce dpr over incoming (target) / outgoing (source) edges
    // ... into dKP = [i key | j key | i msg | j msg] (leading dimension 4C): the two K-segments of dk (and of dv) are then
    // adjacent column blocks, and the sum over them is ONE product over K = 2C on the DMA-fed kernel (b_split_folded) --
    // as two-segment products without images they ran on the general kernel, 0.2 ms each on the main stream
    float* dKP = w.dKP[l];
    if (C == 256) {
      // round 5: all four sums in ONE launch (cartnet_segment_sum_pair: a node's by-target and by-source sums adjacent, the
      // second read of a crystal's rows from cache; chunk j of 256 columns = key / msg lands at column j * 2C): 28.91 / 29.01 vs
      // 29.04 / 29.05 ms per step, same box)
      RUN(cartnet_segment_sum_pair(dpr, 2 * C, w.rowptr, w.colptr, w.perm, N, 2 * C, dKP, dKP + C, 4 * C, 2 * C, st));
    } else {
      RUN(cartnet_segment_sum(dpr, 2 * C, w.rowptr, nullptr, N, C, dKP, 4 * C, st));
      RUN(cartnet_segment_sum(dpr + C, 2 * C, w.rowptr, nullptr, N, C, dKP + 2 * C, 4 * C, st));
      RUN(cartnet_segment_sum(dpr, 2 * C, w.colptr, w.perm, N, C, dKP + C, 4 * C, st));
      RUN(cartnet_segment_sum(dpr + C, 2 * C, w.colptr, w.perm, N, C, dKP + 3 * C, 4 * C, st));
    }
    const float* k = w.QKV[l] + C; const float* v = w.QKV[l] + 2 * C;
    RUN(wg({dKP, dKP + 2 * C, dKP + C, dKP + 3 * C}, 4 * C, {k, v, k, v}, 3 * C,
           {g.key0_w, g.msg0_w, g.key0_w + C, g.msg0_w + C}, 3 * C, N, C, C));
    for (int km = 0; km < 2; ++km) {   // dk = dKPi_k W1k_i + dKPj_k W1k_j ; dv likewise
      CartnetGemmArgs a = gargs(prec, N, C, C, 4 * C, 3 * C, 3 * C);
      a.nsegs = 2; a.b_kstrided = 1;
      const float* W1 = km ? q.msg0_w : q.key0_w;
      a.A[0] = dKP + km * 2 * C; a.A[1] = a.A[0] + C; a.B[0] = W1; a.B[1] = W1 + C;
      a.C[0] = w.dQKV[l] + (1 + km) * C;
      if (w.use_img && N >= 2048) a.b_split_folded = cw.B[km ? B_M1I : B_K1I];
      if (bias_fuse) a.colsum[0] = km ? w.cq : w.cs;          // lin_key's / lin_value's bias gradients
      RUN(cartnet_gemm(&a, st));
    }
    if (bias_fuse) {
      double* parts[2] = {w.cs, w.cq};
      float* outs[2] = {g.key_b, g.value_b};
      RUN(cartnet_colsum_finalize(parts, outs, 2, w.tiles_n, C, st));
    } else {
      RUN(colsum(w.dQKV[l] + C, 3 * C, N, C, w.pa, g.key_b, st));
      RUN(colsum(w.dQKV[l] + 2 * C, 3 * C, N, C, w.pa, g.value_b, st));
    }
    return linear3_bwd(l, q, g, x_in, N, w.dx[l]);
  };

  // ---- ComformerConv_edge backward: de_old, dNL3 [Bg, 3C], dNA [3E, C] (dNA on the side stream)
  auto conv_edge_bwd = [&](const float* dy) -> int {
    const int l = 4;
    const CartnetIcfConv& q = P.edge;
    const CartnetIcfConv& g = G.edge;
    const ConvW& cw = w.cw[l];
    RUN(softplus_bwd(l, q, g, dy, E, w.e0, w.tmpb));
    RUN(cartnet_eltwise(3, w.tmpb, nullptr, g.concate_b, 1, C, C, 0, C, 3.0f, st));     // the bias entered three times
    RUN(wg({w.d_o[l]}, C, {w.aggr[l]}, C, {g.concate_w}, C, b.E, C, C));
    RUN(dgemm(w.d_o[l], C, q.concate_w, C, w.use_img ? cw.B[B_CAT] : nullptr, w.daggr[l], C, b.E, C, C, nullptr, 0, st));
    Att t{3 * E, E, w.ptr3, w.idx_edge, w.idx_gl, w.Ka, w.KYb, w.QKV[l], 3 * C, 3LL * b.E, w.gp_e, w.sp_e, E > 3 * Bg ? E : 3 * Bg};
    RUN(att_backward(l, q, g, t, w.daggr[l], w.dQKV[l]));
    float* dpr = w.dpr[l];
    // angle branch (lin_edge folded, no bias): only the RBF backward at the very end reads dNA
    if (rbf_fuse) RUN(fold_backward(l, q, g, w.NA, 3LL * b.E, w.dpre_na, nullptr, w.pre_na, G.rbf_angle_b));
    else RUN(fold_backward(l, q, g, w.NA, 3LL * b.E, w.dNA, nullptr));
    // per-edge term (sum over the three lattice vectors) and per-(crystal, lattice vector) term
    // (dpr is [3E, 2C] = [E, 3, 2C]: the per-edge term is the sum of each edge's three rows, the per-(crystal, lattice
    //  vector) term the sum over the crystal's edges of the [E, 6C] view.  Through cartnet_segment_sum -- one wave per
    //  segment -- the second was 64 x 2 waves on the whole chip, three times: 3 ms of the step; the first ran at 1.5 TB/s.)
    // (round 5: ONE pass over dpr for both -- cartnet_segment_sum_chunked_fold3 -- instead of a three-row-sum kernel + the
    //  chunked sums, 1.09 GB each at the benchmark batch: 28.68 / 28.82 vs 28.86 / 29.06 ms per step, same box)
    RUN(cartnet_segment_sum_chunked_fold3(dpr, 6 * C, w.gedge_ptr, Bg, E, 6 * C, w.seg_tmp_e, w.dKYb, 6 * C, w.dKa, st));
    RUN(wg({w.dKa, w.dKa + C}, 2 * C, {w.QKV[l] + C, w.QKV[l] + 2 * C}, 3 * C, {g.key0_w, g.msg0_w}, 3 * C, b.E, C, C));
    RUN(wg({w.dKYb, w.dKYb + C}, 2 * C, {w.KY, w.VY}, C, {g.key0_w + C, g.msg0_w + C}, 3 * C, (long long)Bg * 3, C, C));
    RUN(dgemm(w.dKa, 2 * C, q.key0_w, 3 * C, w.use_img ? cw.B[B_K1I] : nullptr, w.dQKV[l] + C, 3 * C, b.E, C, C, nullptr, 0, st,
              bias_fuse ? w.cs : nullptr));
    RUN(dgemm(w.dKa + C, 2 * C, q.msg0_w, 3 * C, w.use_img ? cw.B[B_M1I] : nullptr, w.dQKV[l] + 2 * C, 3 * C, b.E, C, C, nullptr, 0, st,
              bias_fuse ? w.cq : nullptr));
    if (bias_fuse) {
      double* parts[2] = {w.cs, w.cq};
      float* outs[2] = {g.key_b, g.value_b};
      RUN(cartnet_colsum_finalize(parts, outs, 2, w.tiles_e, C, st));
    } else {
      RUN(colsum(w.dQKV[l] + C, 3 * C, E, C, w.pa, g.key_b, st));
      RUN(colsum(w.dQKV[l] + 2 * C, 3 * C, E, C, w.pa, g.value_b, st));
    }
    RUN(dgemm(w.dKYb, 2 * C, q.key0_w + C, 3 * C, nullptr, w.dKY, C, (long long)Bg * 3, C, C, nullptr, 0, st));
    RUN(dgemm(w.dKYb + C, 2 * C, q.msg0_w + C, 3 * C, nullptr, w.dVY, C, (long long)Bg * 3, C, C, nullptr, 0, st));
    // lin_key_e{i} / lin_value_e{i} on the lattice-length features ([Bg, 3C] views)
    for (int i = 0; i < 3; ++i) {
      for (int kv = 0; kv < 2; ++kv) {
        const float* dT = (kv ? w.dVY : w.dKY) + i * C;
        RUN(colsum(dT, 3 * C, Bg, C, w.pa, kv ? g.value_e_b[i] : g.key_e_b[i], st));
        RUN(wg({dT}, 3 * C, {w.NLt + i * C}, 3 * C, {kv ? g.value_e_w[i] : g.key_e_w[i]}, C, Bg, C, C));
      }
      CartnetGemmArgs a = gargs(prec, Bg, C, C, 3 * C, C, 3 * C);
      a.nsegs = 2; a.b_kstrided = 1;
      a.A[0] = w.dKY + i * C; a.A[1] = w.dVY + i * C; a.B[0] = q.key_e_w[i]; a.B[1] = q.value_e_w[i]; a.C[0] = w.dNL3 + i * C;
      RUN(cartnet_gemm(&a, st));
    }
    return linear3_bwd(l, q, g, w.e0, E, w.de_old);
  };

  // ---- layers in reverse: att 3, 2, 1 (all read the updated edge features), edge layer, att 0
  const float* dx = w.dx_head;
  const float* de_new = nullptr;
  for (int l = 3; l >= 1; --l) {
    RUN(conv_bwd(l, dx, de_new));
    dx = w.dx[l];
    de_new = w.de[l];
  }
#define WAIT_ROWS(l) do { if (S.main_waits(rows_ready[l]) != 0) { cartnet_set_error("cartnet_icomformer_backward: stream wait failed"); return 2; } } while (0)
  WAIT_ROWS(1);                                      // de_new = de[1] (accumulated from layers 3 and 2 on the side stream)
  RUN(conv_edge_bwd(de_new));
  RUN(conv_bwd(0, dx, w.de_old));
  dx = w.dx[0];
  if (!rbf_fuse) {
    WAIT_ROWS(0);                                    // de[0]
    WAIT_ROWS(4);                                    // dNA
  }
#undef WAIT_ROWS

  // ---- RBF branches: out = softplus(pre), pre = rbf W^T + b; rbf.1 is shared by the distance and the lattice-length
  //      features, so its gradients add up
  auto rbf_bwd = [&](const float* r, const float* pre, const float* dout, long long n, float* dpre, float* gw, float* gb) -> int {
    if (!bias_fuse) {
      RUN(cartnet_eltwise(1, dout, pre, dpre, n, C, C, C, C, 1.0f, st));
      RUN(wg({dpre}, C, {r}, C, {gw}, C, n, C, C));
      return colsum(dpre, C, n, C, w.pa, gb, st);
    }
    RUN(cartnet_softplus_bwd_sums(dout, C, pre, C, dpre, C, (int)n, C, w.pa, st));
    RUN(wg({dpre}, C, {r}, C, {gw}, C, n, C, C));
    double* parts[1] = {w.pa};
    float* outs[1] = {gb};
    return cartnet_colsum_finalize(parts, outs, 1, cartnet_segment_nparts((int)n), C, st);
  };
  if (rbf_fuse) {     // dpre_e / dpre_na and the two bias gradients came out of the products above (side stream)
    RUN(wg({w.dpre_e}, C, {w.r_e}, C, {w.gw1}, C, b.E, C, C));
    RUN(rbf_bwd(w.r_nl, w.pre_nl, w.dNL3, (long long)Bg * 3, w.dpre_nl, w.gw2, w.gb2));
    RUN(wg({w.dpre_na}, C, {w.r_na}, C, {G.rbf_angle_w}, C, 3LL * b.E, C, C));
  } else {
    RUN(rbf_bwd(w.r_e, w.pre_e, w.de[0], b.E, w.dpre_e, w.gw1, w.gb1));
    RUN(rbf_bwd(w.r_nl, w.pre_nl, w.dNL3, (long long)Bg * 3, w.dpre_nl, w.gw2, w.gb2));
    RUN(rbf_bwd(w.r_na, w.pre_na, w.dNA, 3LL * b.E, w.dpre_na, G.rbf_angle_w, G.rbf_angle_b));
  }
  JOIN();                                            // gw1 / gw2 come from the side stream
  RUN(cartnet_eltwise(2, w.gw1, w.gw2, G.rbf_w, C, C, C, C, C, 1.0f, st));
  RUN(cartnet_eltwise(2, w.gb1, w.gb2, G.rbf_b, 1, C, C, C, C, 1.0f, st));

  // ---- atom embedding and temperature projection
  RUN(cartnet_node_embed_bwd(b.batch, b.temperature, dx, N, C, Bg, w.pa, w.pb, st));
  {
    double* parts[2] = {w.pa, w.pb};
    float* outs[2] = {G.temp_w, G.temp_b};
    RUN(cartnet_colsum_finalize(parts, outs, 2, w.nparts_n, C, st));
  }
  RUN(cartnet_sort_by_key(b.z, N, m.n_types, w.zperm, w.zptr, w.zstatus, st));
  RUN(cartnet_segment_sum_long(dx, C, w.zptr, w.zperm, m.n_types, N, C, w.seg_tmp, G.embedding, C, st));
  JOIN();
#undef FORK
#undef JOIN
  return 0;
}
