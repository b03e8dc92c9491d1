/*
 * cartnet_hip.h -- C ABI of libcartnet_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the CartNet
 * message-passing hot path (reference: models/cartnet.py:65-73 CartNet.forward and its autograd backward).
 *
 * The reference has no native boundary of its own (pure Python on PyG / pytorch-scatter / ATen); each entry
 * point below names the reference call site whose ATen/PyG kernels it replaces.  Conventions:
 *   - every pointer is a borrowed DEVICE pointer (the caller, e.g. torch, owns all memory); float = fp32
 *     row-major with an explicit leading dimension in elements; indices are int32 on the device
 *     (cartnet_csr_build converts the int64 edge_index of the PyG API once per batch);
 *   - `stream` is a hipStream_t passed as void*; kernels are only enqueued, no host synchronisation, no
 *     allocation -> safe to capture in a hipGraph;
 *   - state: the library reads NO environment variable and keeps no tunable global.  Every choice a caller can make
 *     is a field of CartnetGemmArgs / CartnetModel.  What it does keep is per THREAD: the message behind
 *     cartnet_last_error(), and, inside cartnet_model_forward / _backward, a pool of hipEvent_t (created once per
 *     thread, disable-timing, reused by every call of that thread to order its two streams) -- so two host threads
 *     may drive two models concurrently, one thread must not interleave two calls.  The one process-global is the
 *     opt-in launch timer (cartnet_profile_gemm*), off unless bench.py switches it on;
 *   - return value 0 = ok, non-zero = error; cartnet_last_error() returns a thread-local message.
 *   - shapes are validated on the host before launch (a bad shape returns an error, it never launches).
 */
#ifndef CARTNET_HIP_H
#define CARTNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CARTNET_MAX_GROUPS 4

const char* cartnet_last_error(void);
int cartnet_abi_version(void);
/* sizeof of every struct below, in header order (CartnetGemmArgs, CartnetShard, CartnetCollated, CartnetGemmProfile,
 * CartnetGroups, CartnetLayerParams, CartnetLayerBuffers, CartnetParams, CartnetModel, CartnetBatch,
 * CartnetGateGemmArgs, CartnetIcfConv, CartnetIcfParams, CartnetIcfModel); returns the number of structs.  A binding checks its mirrors against these when it loads the
 * library, and cartnet_abi_version() against the version it was written for (12: CartnetGemmArgs.gather_rows closes the struct -- the persistent kernel's node-term gather form; 11: CartnetGemmArgs.tile_policy = 3, the persistent activation x weight kernel; no layout change; 10: dact_kind closes CartnetGemmArgs, the *_sums / cartnet_att_gate_bwd_apply entry points; 9: gst_* in CartnetGemmArgs (8 also carried seg_*: per-target sums in an epilogue, measured and removed); 7: tile_policy in CartnetGemmArgs,
 * aux_stream in cartnet_model_forward, CartnetGateGemmArgs in the size table; cartnet_gemm_tile_policy() is gone). */
int cartnet_abi_struct_sizes(size_t* out, int32_t capacity);

/* ------------------------------------------------------------------------------------------------------
 * Dense per-row GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 accumulate).
 * Replaces every pyg_nn.Linear on the path (models/cartnet.py:119,126,133,135,188-195,289-291,319-321) and the
 * matching autograd mm/addmm backward kernels.  One launch computes `ngroups` independent problems of the
 * same shape (e.g. the gate and sender MLPs of a layer) or one problem summed over `nsegs` K-segments
 * (e.g. dX = sum_s dY_s @ W_s).  Index the pointer arrays by group when ngroups > 1, by segment otherwise.
 *
 *   C[g][m, n] = epilogue( sum_s sum_k opA(A[s])[m, k] * opB(B[s])[k, n] )
 *
 * a_kstrided = 0: A[m,k] at A + m*lda + k      (activations, k contiguous)
 * a_kstrided = 1: A[m,k] at A + k*lda + m      (transposed use: weight gradients, reduction over rows)
 * b_kstrided = 0: B[k,n] at B + n*ldb + k      (weights stored [out,in]: Y = X W^T)
 * b_kstrided = 1: B[k,n] at B + k*ldb + n      (dX = dY W ; weight gradients)
 * a_act / b_act = 1 applies SiLU to the operand as it is staged (fuses the activation between two Linears).
 *
 * Epilogue, applied in this order to v = accumulator:
 *   v += bias[g][n]; v += gather_i[g][tgt[m]*ldg + n] + gather_j[g][src[m]*ldg + n]; v += resid[g][m*ldr+n];
 *   v *= silu'(dact[g][m*ldd + n]);            (dact_kind = 1: v *= sigmoid(dact[g][m*ldd + n]), softplus')
 *   if colsum[g]: per-column partial sums of v (and v*v into colsq[g]) over this block's rows are written to
 *                 colsum[g][tile_m * N + n]  (tile_m in [0, ceil(M/128)); reduce with cartnet_colsum_finalize
 *                 or cartnet_bn_finalize -- deterministic, no atomics);
 *   if cpre[g]:   cpre[g][m*ldc + n] = v;      (pre-activation kept for backward)
 *   if out_act:   v = silu(v);                 (dact_kind = 1: v = softplus(v))
 *   C[g][m*ldc + n] = v.
 * splitk > 1 (only with a_kstrided = b_kstrided = 1, no epilogue): the K range is cut into `splitk` chunks and
 *   chunk s writes its raw partial tile to C[g] + s*M*ldc; reduce with cartnet_splitk_reduce.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct CartnetGemmArgs {
  const float* A[CARTNET_MAX_GROUPS];
  const float* B[CARTNET_MAX_GROUPS];
  float* C[CARTNET_MAX_GROUPS];
  float* cpre[CARTNET_MAX_GROUPS];
  const float* bias[CARTNET_MAX_GROUPS];
  const float* gather_i[CARTNET_MAX_GROUPS];
  const float* gather_j[CARTNET_MAX_GROUPS];
  const float* resid[CARTNET_MAX_GROUPS];
  const float* dact[CARTNET_MAX_GROUPS];
  double* colsum[CARTNET_MAX_GROUPS];
  double* colsq[CARTNET_MAX_GROUPS];
  const int32_t* tgt;
  const int32_t* src;
  int32_t M, N, K;
  int32_t lda, ldb, ldc, ldg, ldr, ldd;
  int32_t ngroups, nsegs, splitk;
  int32_t a_kstrided, b_kstrided, a_act, b_act, out_act;
  int32_t precision;   /* 0: fp32 MFMA (exact fp32 products).  1: bf16x3 split operands, 6 bf16 MFMAs per product,
                          fp32 accumulate (fp32-level accuracy, see gemm_kernel.h); falls back to 0 where no
                          such kernel exists (narrow tiles, ragged K tail, unaligned operands).
                          2: plain bf16 operands (round-to-nearest), ONE bf16 MFMA per product, fp32 accumulate and
                          fp32 storage -- the reduced-precision mode of BASELINE configs[2]; only the pre-split /
                          transposing-read kernels implement it (256-wide tiles), every other shape runs precision 0. */
  const void* b_split[CARTNET_MAX_GROUPS];
                       /* optional, b_kstrided = 1 and a_kstrided = 0: B[i] pre-arranged as the kernel's LDS image --
                          by cartnet_gemm_split_b for precision 1 / 2 (bf16 planes; the weight operand is split once
                          per step instead of once per tile), by cartnet_gemm_pack_b for precision 0 (swizzled fp32
                          rows of 16 k).  The image must match the precision of the call.  B[i] must still be given
                          (fallback paths read it). */
  const void* b_split_folded;
                       /* optional, nsegs > 1 and N == 256: the segments' images one after the other (segment order).
                          Used when the A segments are adjacent column blocks of one matrix (A[s] == A[0] + s*K):
                          the sum over segments is then one product over K*nsegs. */
  float* a_act_out[CARTNET_MAX_GROUPS];
                       /* optional, needs a_act = 1, a k-contiguous A and no K-segments: silu(A[g]) is also WRITTEN, with
                          A's row stride.  The pre-packed 256-wide fp32 kernel (precision 0, b_split) writes it as a
                          by-product of staging A (each element is activated exactly once per column tile anyway); any
                          other launch runs an elementwise pass first.  The second Linear's weight gradient
                          dW = dY^T silu(pre) then reads it as a plain operand and takes the all-DMA kernel. */
  int32_t a_half, b_half, c_half, dact_half;
                       /* precision 2 only ("bf16 storage / fp32 accumulate"): the operand lives in
                          memory as bf16 -- A[g] / B[g] / C[g] / dact[g] then point at bf16 elements (leading dimensions
                          stay in ELEMENTS).  Honoured by the half-storage kernels (csrc/gemm_h.h): activation x weight
                          products with a pre-split weight image (b_split) and weight gradients (both operands
                          k-strided); every other launch with one of these flags set is refused. */
  /* Gate statistics in the epilogue (precision 0 or 1 with a weight image, ngroups == 1, resid optional, nothing else: the dE
   * product of CartNet's backward).  v = the value written to C (de_out of the layer BELOW, models/cartnet.py:225).  With
   * g = gst_g[m*gst_ld + n] (that layer's gate pre-activation, cartnet.py:237), ghat = (g - mean[n]) * rstd[n] with
   * mean_rstd = [mean | rstd] (2N floats), w = gst_env[m] * s * (1 - s), s = sigmoid(ghat * gamma[n] + beta[n])
   * (cartnet.py:238-241), the per-column partial sums over this block's rows of  v * w  -> colsum[0][tile_m*N + n]  and
   * v * w * ghat -> colsq[0][tile_m*N + n] (both must be given): the edge-residual share of the BatchNorm backward's
   * sum(dbn) and sum(dbn * ghat), taken where de_out is in registers instead of by a pass over gs and de_out.
   * gst_env may be NULL (no envelope).  Any other launch with gst_g set is refused. */
  const float* gst_g;
  const float* gst_env;
  const float* gst_mean_rstd;
  const float* gst_gamma;
  const float* gst_beta;
  int32_t gst_ld;
  int32_t tile_policy; /* which DMA-fed fp32 activation x weight kernel takes the launch (precision 0, b_split given).
                          0: the library's choice per launch -- the PERSISTENT kernel (one workgroup per CU walks its tiles,
                          the epilogue of a tile inside the next tile's MFMA chain: csrc/gemm_f32p.h) for K = 256 / 512 / 768
                          launches of at least 4 tiles per CU whose epilogue it has a form for; otherwise 128 x 256 tiles
                          on two workgroups per CU, or 128 x 128 on three for the node-term gather epilogue, for
                          single-group N = 256 products and for launches with few tiles.  1: grouped N = 256 products
                          take the narrow tile too where the persistent kernel has no form (the iComformer path).
                          3: the persistent kernel for every launch it has the form for, whatever its size (tests, A/B
                          runs).  128 / 256: force one of the tile kernels, never the persistent one (A/B runs). */
  int32_t dact_kind;   /* The activation family of the launch's epilogue (dact and out_act).
                          0: v *= silu'(dact) (the SiLU between two Linears, models/cartnet.py:127,136).  1: v *= sigmoid(dact),
                          the derivative of softplus (iComformer's RBF branches, models/comformer.py:93-105: the product that
                          yields d(softplus output) writes d(pre-activation) and, with colsum, the bias gradient -- no
                          element-wise pass, no column-sum pass); out_act then applies softplus (threshold 20) instead of
                          SiLU: with cpre the forward of such a branch in one launch.  Not with the bf16-storage flags. */
  int32_t gather_rows; /* optional (ABI 12): an upper bound of the rows of the gather_i / gather_j tables (every tgt[m] and
                          src[m] is below it); 0 = not stated.  The persistent kernel addresses the tables through 32-bit
                          offsets and takes a gather launch only when it knows gather_rows * ldg * 4 < 2^32; every other
                          kernel ignores the field. */
} CartnetGemmArgs;

int cartnet_gemm(const CartnetGemmArgs* args, void* stream);
/* 1 if a launch with these arguments (gst_* set) takes the kernel that carries the gate-statistics epilogue; cartnet_gemm
 * refuses such a launch otherwise.  Host only, no launch. */
int cartnet_gemm_gate_stats_ok(const CartnetGemmArgs* args);

/* bf16x3 pre-split of a k-strided GEMM operand B [K, N] (element (k, n) = src[k*stride_k + n*stride_n]; a weight
 * W [out, in] used as B = W^T has stride_k = 1, stride_n = ld) into the image cartnet_gemm reads through
 * CartnetGemmArgs.b_split: per 256-column tile and 16-deep K-step, three planes (high / middle / low bf16 piece) of
 * [256][16] bf16 in the kernel's LDS order.  K % 16 == 0 and N % 256 == 0; dst: cartnet_gemm_split_b_bytes(K, N)
 * = 6*K*N bytes (0 when the shape has no such image), 16-byte aligned.  njobs matrices per call (host arrays). */
size_t cartnet_gemm_split_b_bytes(int32_t K, int32_t N);
/* The precision-0 counterpart: fp32 image (per 256-column tile and K-step [256][16] floats, 16-byte slots of a row
 * XOR-swizzled), 4*K*N bytes. */
size_t cartnet_gemm_pack_b_bytes(int32_t K, int32_t N);
int cartnet_gemm_pack_b(const float* const* src, void* const* dst, const int32_t* K, const int32_t* N,
                        const int32_t* stride_k, const int32_t* stride_n, int32_t njobs, void* stream);
int cartnet_gemm_split_b(const float* const* src, void* const* dst, const int32_t* K, const int32_t* N,
                         const int32_t* stride_k, const int32_t* stride_n, int32_t njobs, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * iComformer-only pieces (models/comformer.py:75-132, models/comformer_conv.py:21-193); the dense work of that
 * model reuses cartnet_gemm, its gated aggregation cartnet_gate_scatter_* (e_in = e_out = NULL: no edge residual).
 * ---------------------------------------------------------------------------------------------------- */
/* Gaussian RBF (models/utils.py:125-129): out[r, k] = exp(-gamma (v[r] - centers[k])^2). */
int cartnet_rbf_expand(const float* v, int64_t n, const float* centers, int32_t bins, float gamma, float* out,
                       int32_t ldo, void* stream);
/* models/comformer.py:117-120: edge_feat[e] = -0.75/dist[e]; nei_len[g,i] = -0.75/|cell[g,i]|;
 * nei_cos[e,i] = clamp(cos(cell[batch[src[e]], i], cart_dir[e]), -1, 1). */
int cartnet_lattice_features(const float* cell, const int64_t* batch, const int32_t* src32, const float* cart_dist,
                             const float* cart_dir, int64_t E, int32_t Bg, float* edge_feat, float* nei_len,
                             float* nei_cos, void* stream);
/* op 0: out = softplus(a); 1: out = a * sigmoid(b) (softplus backward); 2: out = a + b; 3: out = a * scale. */
int cartnet_eltwise(int32_t op, const float* a, const float* b, float* out, int64_t rows, int32_t cols, int32_t lda,
                    int32_t ldb, int32_t ldo, float scale, void* stream);
/* alpha[r] = key[r] * q[s] * scale for rows r in [ptr[s], ptr[s+1])  (query_i * key / sqrt(C), comformer_conv.py:95)
 * + fp64 column partial sums of alpha / alpha^2 [cartnet_segment_nparts(S)][C] for bn_att.  alpha == NULL: the statistics
 * only (alpha is then recomputed by cartnet_att_gate_fwd and cartnet_att_gate_bwd_apply and never written). */
int cartnet_segment_nparts(int32_t S);
int cartnet_rowmul_fwd(const float* key, int32_t ldk, const float* q, int32_t ldq, const int32_t* ptr, int32_t S,
                       int32_t C, float scale, float* alpha, int32_t lda, double* parts_sum, double* parts_sq,
                       void* stream);
/* Backward, in place: dalpha <- dkey = dalpha * q[s] * scale;  dq[s] = scale * sum_r dalpha[r] * key[r]. */
int cartnet_rowmul_bwd(float* dalpha, int32_t lda, const float* key, int32_t ldk, const float* q, int32_t ldq,
                       const int32_t* ptr, int32_t S, int32_t C, float scale, float* dq, int32_t lddq, void* stream);
/* The same pass, also leaving the fp64 column partials of dkey and of dq ([cartnet_segment_nparts(S)][C] each, C <= 256 ->
 * cartnet_colsum_finalize): the bias gradients of key_update.2 and lin_query (comformer_conv.py:45-47,72), which autograd
 * takes as column sums of those two tensors -- here without a pass of their own. */
int cartnet_rowmul_bwd_sums(float* dalpha, int32_t lda, const float* key, int32_t ldk, const float* q, int32_t ldq,
                            const int32_t* ptr, int32_t S, int32_t C, float scale, float* dq, int32_t lddq,
                            double* parts_dkey, double* parts_dq, void* stream);
/* The attention gate forward from the KEY rows (comformer_conv.py:90-99): gs = [key | msg] [R, 2D], alpha = key * q[s] * scale
 * recomputed per row, aggr[s] = sum_r sigmoid(bn_att(alpha)) * msg with mean_rstd from cartnet_rowmul_fwd's statistics;
 * bc != NULL: the per-segment sums B | C [S, 2D] of cartnet_gate_scatter_fwd_bc. */
int cartnet_att_gate_fwd(const float* gs, const float* q, int32_t ldq, const int32_t* ptr, const float* mean_rstd,
                         const float* gamma, const float* beta, float scale, int32_t S, int32_t D, float* aggr, float* bc,
                         void* stream);
/* The attention gate's backward (cartnet_gate_scatter_bwd_apply with e_out = env = NULL) and cartnet_rowmul_bwd_sums in ONE
 * pass (comformer_conv.py:90-99 backward): gs = [alpha | msg] [R, 2D] in place -> [dkey | dmsg], dq [S, D] (leading
 * dimension lddq); sums = [sum dbn | sum dbn ahat] (the finalised BatchNorm-backward sums of bn_att), count = the rows they
 * were taken over; fp64 column partials of dkey, dmsg and dq ([cartnet_segment_nparts(S)][D] each).
 * key == NULL: gs = [key | msg] (the forward pass of cartnet_att_gate_fwd: alpha is recomputed from the key rows). */
int cartnet_att_gate_bwd_apply(float* gs, const float* key, int32_t ldk, const float* q, int32_t ldq, const float* daggr,
                               const int32_t* ptr, const float* mean_rstd, const float* gamma, const float* beta,
                               const float* sums, int64_t count, int32_t training, float scale, int32_t S, int32_t D,
                               float* dq, int32_t lddq, double* parts_dkey, double* parts_dmsg, double* parts_dq,
                               void* stream);
/* y = softplus(x + bn(o)) (comformer_conv.py:88,193) and its backward in the two-pass BatchNorm form:
 * stats: du = dy * sigmoid(x + bn(o)); partial sums of du and du*ohat -> parts [cartnet_segment_nparts(N)][D];
 * apply: d_o = gamma*rstd*(du - sum_a/N - ohat*sum_b/N) (mean terms dropped when training == 0), dx = du (+ dx_add). */
int cartnet_softplus_update_fwd(const float* o, const float* x, const float* mean_rstd, const float* gamma,
                                const float* beta, int64_t N, int32_t D, float* y, void* stream);
int cartnet_softplus_update_bwd_stats(const float* o, const float* x, const float* dy, const float* mean_rstd,
                                      const float* gamma, const float* beta, int32_t N, int32_t D, double* parts_a,
                                      double* parts_b, void* stream);
int cartnet_softplus_update_bwd_apply(const float* o, const float* x, const float* dy, const float* mean_rstd,
                                      const float* gamma, const float* beta, const float* sums, int32_t training,
                                      int32_t N, int32_t D, float* d_o, const float* dx_add, float* dx, void* stream);
/* apply + the fp64 column partials of d_o ([cartnet_segment_nparts(N)][D]): the bias gradient of lin_concate
 * (comformer_conv.py:87,190), whose output gradient d_o is. */
int cartnet_softplus_update_bwd_apply_sums(const float* o, const float* x, const float* dy, const float* mean_rstd,
                                           const float* gamma, const float* beta, const float* sums, int32_t training,
                                           int32_t N, int32_t D, float* d_o, const float* dx_add, float* dx,
                                           double* parts_do, void* stream);
/* out = a * sigmoid(b) over [R, C] views (cartnet_eltwise op 1: the softplus of the RBF branches backward,
 * comformer.py:93-105) + the fp64 column partials of out ([cartnet_segment_nparts(R)][C]): the bias gradient of the
 * Linear in front of the softplus. */
int cartnet_softplus_bwd_sums(const float* a, int32_t lda, const float* b, int32_t ldb, float* out, int32_t ldo,
                              int32_t R, int32_t C, double* parts, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * eComformer's equivariant update (models/comformer_conv.py:197-280: ComformerConvEqui = two TensorProductConvLayer
 * over e3nn's FullyConnectedTensorProduct with per-edge weights, irreps 64x0e -> 64x0e + 8x1o + 8x2e -> 64x0e,
 * spherical harmonics 1x0e + 1x1o + 1x2e, "component" normalisation, scatter-mean over edge_index[0]).
 * Restated without e3nn (oracle/ecomformer_ref.py has the derivation; parity with e3nn itself is unpinned).
 *   w        [E, 5120] per-edge weights from the edge MLP (layer 1: [64x64 | 64x8 | 64x8], layer 2: [80 x 64])
 *   colptr / perm   edges grouped by SOURCE atom (cartnet_csr_build); tgt [E] int32 target of every edge
 *   tp1: x0 [N, 64] -> h1 [N, 128] = mean_{edges leaving j} [t0 | t1 (x) Y1 | t2 (x) Y2] + pad(x0),  t = x0[tgt] W / 8
 *   tp2: h1 [N, 128] -> o2 [N, 64] = mean_{edges leaving j} [s | <v1,Y1>/sqrt3 | <v2,Y2>/sqrt5][tgt] W / sqrt(80)
 * Backward: dw [E, 5120] and the per-edge gradient of the gathered rows (dxe [E, 64] / dhe [E, 128]), which the
 * caller reduces over targets with cartnet_segment_sum (+ dh1[:, :64] for layer 1's residual).
 * ---------------------------------------------------------------------------------------------------- */
int cartnet_equi_tp1_fwd(const float* x0, const float* w, const float* cart_dir, const int32_t* colptr,
                         const int32_t* perm, const int32_t* tgt, int32_t N, float* h1, void* stream);
int cartnet_equi_tp1_bwd(const float* x0, const float* w, const float* cart_dir, const int32_t* colptr,
                         const int32_t* perm, const int32_t* tgt, const float* dh1, int32_t N, float* dw, float* dxe,
                         void* stream);
int cartnet_equi_tp2_fwd(const float* h1, const float* w, const float* cart_dir, const int32_t* colptr,
                         const int32_t* perm, const int32_t* tgt, int32_t N, float* o2, void* stream);
int cartnet_equi_tp2_bwd(const float* h1, const float* w, const float* cart_dir, const int32_t* colptr,
                         const int32_t* perm, const int32_t* tgt, const float* do2, int32_t N, float* dw, float* dhe,
                         void* stream);
/* fp64 partial column sums and sums of squares of x [R, C] -> parts [cartnet_colstats_nparts(R)][C] each (BatchNorm
 * statistics of a tensor no GEMM epilogue produced); finalise with cartnet_bn_finalize. */
int cartnet_colstats_nparts(int32_t R);
int cartnet_colstats_partial(const float* x, int32_t ld, int32_t R, int32_t C, double* parts_sum, double* parts_sq,
                             void* stream);

/* parts[p][c] = partial column sums of the [R, C] view x (p < cartnet_segment_nparts(R)); finalise with
 * cartnet_colsum_finalize (bias gradients of Linears whose output gradient is not produced by a GEMM epilogue). */
int cartnet_colsum_partial(const float* x, int32_t ld, int32_t R, int32_t C, double* parts, void* stream);
/* parts_a[p][c] / parts_b[p][c] = partial column sums of d[r, c] * bc[r, c] / d[r, c] * bc[r, C + c] (d: an [R, C] view, bc
 * [R, 2C] from cartnet_gate_scatter_fwd_bc, p < cartnet_segment_nparts(R)): sum_t daggr[t] B[t] and sum_t daggr[t] C[t], the
 * whole of the gate's two BatchNorm-backward sums when there is no edge residual (iComformer's attention gate,
 * models/comformer_conv.py:90-99) -- no statistics pass over the edges. */
int cartnet_coldot_bc_partial(const float* d, int32_t ld, const float* bc, int32_t R, int32_t C, double* parts_a,
                              double* parts_b, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Periodic radius graph on the GPU (reference: dataset/utils.py:57-237 radius_graph_pbc as used by
 * dataset/figshare_dataset.py:65-68; pairs with d^2 <= 1e-4 dropped).  Edges come out in the
 * reference's order (target, source, periodic image), i.e. edge_index[1] ascending.
 *   count: reps = caller's scratch of 15 * Bg 4-byte words, kept for fill: int32 [Bg,3] periodic repetitions per
 *          lattice direction, then fp32 [Bg,12] reciprocal lattice vectors and radius |b_d| (the per-pair image box:
 *          only images that can lie within the radius are tested); deg[N] = in-degree of every atom.
 *   fill:  rowptr[N+1] = exclusive prefix sum of deg (int64), E = rowptr[N]; writes edge_index [2,E] (int64,
 *          row 0 source, row 1 target), cart_dist [E], cart_dir [E,3] = (pos_target - (pos_source + offset)) / dist,
 *          and, if cart_dist_sq is not NULL, the squared distances [E] the neighbour cap ranks by.
 * Neighbour cap (dataset/utils.py:240-360 get_max_neighbors_mask with enforce_max_strictly = False, as applied at
 * dataset/utils.py:216-233 when figshare_dataset.py passes max_neigh): a target atom with more than max_neighbors
 * edges keeps those with d^2 <= (max_neighbors+1)-th smallest d^2 of its row + tolerance (0.01 in the reference), in
 * their original order; rows with <= max_neighbors edges are kept whole.
 *   cap_count: rowptr[N+1] of the uncapped graph, dist_sq[E] -> cutoff[N] (fp32, +inf for whole rows), deg[N].
 *   cap_fill:  rowptr_out[N+1] = exclusive prefix sum of deg, E_out = rowptr_out[N]; compacts edge_index [2,E] ->
 *              [2,E_out], cart_dist, cart_dir.
 * ---------------------------------------------------------------------------------------------------- */
int cartnet_radius_graph_count(const float* pos, const float* cell, const int64_t* graph_ptr, const int64_t* batch,
                               int32_t N, int32_t Bg, float radius, int32_t* reps, int32_t* deg, void* stream);
int cartnet_radius_graph_fill(const float* pos, const float* cell, const int64_t* graph_ptr, const int64_t* batch,
                              const int32_t* reps, const int64_t* rowptr, int32_t N, int32_t Bg, float radius,
                              int64_t E, int64_t* edge_index, float* cart_dist, float* cart_dir, float* cart_dist_sq,
                              void* stream);
int cartnet_neighbor_cap_count(const int64_t* rowptr, const float* dist_sq, int32_t N, int32_t max_neighbors,
                               float tolerance, float* cutoff, int32_t* deg, void* stream);
int cartnet_neighbor_cap_fill(const int64_t* rowptr, const int64_t* rowptr_out, const float* cutoff,
                              const float* dist_sq, const int64_t* edge_index, const float* cart_dist,
                              const float* cart_dir, int32_t N, int64_t E, int64_t E_out, int64_t* edge_index_out,
                              float* cart_dist_out, float* cart_dir_out, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * ADP evaluation metrics (reference: train/metrics.py, called per test batch from
 * train/metrics.py:201-214 and main.py:47-49,101-102).  pred, truth: [M,3,3] fp32 symmetric positive definite.
 *   volume_error[M]     = |V(pred) - V(truth)| / (V(pred) + 1e-8), V = 4/3 pi sqrt(det)      (metrics.py:30-58)
 *   similarity_index[M] = 100 (1 - 2^(3/2) det(T^-1 P^-1)^(1/4) / det(T^-1 + P^-1)^(1/2))    (metrics.py:76-94)
 *   iou[M]              = voxel IoU of the two ellipsoids {x : x^T S^-1 x < 1}, S = matrix / max(|P|_F, |T|_F),
 *                         over the grid[num_points]^3 lattice (the reference: linspace(-1, 1, 64)) (metrics.py:96-180)
 * Any of the three outputs may be NULL (skipped); grid / num_points are only read for the IoU.
 * ---------------------------------------------------------------------------------------------------- */
int cartnet_adp_metrics(const float* pred, const float* truth, int32_t M, const float* grid, int32_t num_points,
                        float* volume_error, float* similarity_index, float* iou, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * Training loss (reference: train/metrics.py:15-28, called from train/train.py:173-178): L1Loss and MSELoss with mean
 * reduction over the n = M*9 (or Bg) elements of pred / truth, both from one pair of small launches (slices, then
 * their sum), and their gradient from one more:
 *   out2[0] = mean |pred - truth|,  out2[1] = mean (pred - truth)^2          (sums in fp64, fixed order)
 *   dpred   = g_mae[0] * sign(pred - truth) / n + g_mse[0] * 2 (pred - truth) / n   (g_* device scalars; NULL = 0)
 * Replaces eight eager launches per step (sub, abs, mean; fill, div, sign, mul, mul in backward) by three.
 * ---------------------------------------------------------------------------------------------------- */
int32_t cartnet_loss_nparts(int64_t n);      /* parts: workspace of 2 * cartnet_loss_nparts(n) doubles */
int cartnet_loss_fwd(const float* pred, const float* truth, int64_t n, double* parts, float* out2, void* stream);
int cartnet_loss_bwd(const float* pred, const float* truth, int64_t n, const float* g_mae, const float* g_mse,
                     float* dpred, void* stream);
/* Small inputs (n <= 4096: the scalar targets of a batch, scripts/train_cartnet_jarvis.sh) in ONE launch, which also
 * writes the gradients of either loss for an upstream gradient of exactly 1 -- what loss.backward() of a training step
 * (train/train.py:183) feeds in: unit[0..n) = sign(pred - truth) / n, unit[n..2n) = 2 (pred - truth) / n, bit for bit
 * what cartnet_loss_bwd writes for g_mae[0] = 1 (resp. g_mse[0] = 1).  The caller that knows its seed is 1 then needs
 * no backward launch.  out2 as cartnet_loss_fwd. */
int cartnet_loss_fwd_unit(const float* pred, const float* truth, int64_t n, float* out2, float* unit, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * Device-side batching from a packed shard resident in HBM (replaces torch.load of one pickled
 * Data per structure, dataset/datasetADP.py:41-42, PyG's collate in loader/loader.py:114-124, and the per-sample
 * CPU augmentation / temperature standardisation of dataset/datasetADP.py:33-39,43-45,76-77).
 * A shard is the concatenation of G crystals in CSR form; all pointers are device pointers:
 *   atom_ptr, edge_ptr, y_ptr [G+1] int64;  z [N] int32;  pos [N,3];  non_h_mask [N] u8;
 *   edge_src, edge_tgt [E] int32 (atom index INSIDE its crystal, edge_tgt ascending per crystal);
 *   cart_dist [E];  cart_dir [E,3];  cell [G,9];  temperature [G] (raw kelvin or pre-standardised);
 *   y [Y, y_width]: y_width = 9 -> one 3x3 ADP tensor per non-hydrogen atom, otherwise per-crystal targets.
 * pos, non_h_mask, cell, temperature may be NULL when the matching output is NULL.
 * cartnet_collate gathers crystals sel[0..B) into one batch: out_*_ptr [B+1] are the exclusive prefix sums of the
 * selected crystals' atom / edge / target counts (N, E, M = their last entries).  Outputs: x [N] int64, pos [N,3],
 * non_h_mask [N] u8 (0/1), batch [N] int64, ptr [B+1] int64, edge_index [2,E] int64 (batch numbering), cart_dist
 * [E], cart_dir [E,3], cell [B,9], temperature [B] = (T - temp_mean) / temp_std, y [M, y_width].
 * rot: NULL, or [B,9] one rotation per crystal: cart_dir <- cart_dir R, cell <- cell R, y <- R^T y R (y_width 9).
 * Without rot every output is a bit-exact copy (integers rebased); one launch, no host synchronisation.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct CartnetShard {
  const int64_t* atom_ptr;
  const int64_t* edge_ptr;
  const int64_t* y_ptr;
  const int32_t* z;
  const float* pos;
  const uint8_t* non_h_mask;
  const int32_t* edge_src;
  const int32_t* edge_tgt;
  const float* cart_dist;
  const float* cart_dir;
  const float* cell;
  const float* temperature;
  const float* y;
  int32_t y_width;
} CartnetShard;

typedef struct CartnetCollated {
  int64_t* x;
  float* pos;
  uint8_t* non_h_mask;
  int64_t* batch;
  int64_t* ptr;
  int64_t* edge_index;
  float* cart_dist;
  float* cart_dir;
  float* cell;
  float* temperature;
  float* y;
} CartnetCollated;

int cartnet_collate(const CartnetShard* shard, const int64_t* sel, const int64_t* out_atom_ptr,
                    const int64_t* out_edge_ptr, const int64_t* out_y_ptr, int32_t B, int64_t N, int64_t E, int64_t M,
                    const float* rot, float temp_mean, float temp_std, const CartnetCollated* out, void* stream);

/* Opt-in timing of cartnet_gemm launches (the only PROCESS-global state in the library; used by bench.py, off by default
 * and not for concurrent use from several threads):
 * while enabled, every cartnet_gemm call -- also those issued inside cartnet_model_forward/backward -- is bracketed
 * by HIP events on its launch stream.  cartnet_profile_gemm_read waits for the events and returns per-variant totals
 * (variant: bit0 a_kstrided, bit1 b_kstrided, bit2 a_act, bit3 b_act, bits 4.. = tile width / 64). */
typedef struct CartnetGemmProfile {
  int32_t variant;
  int64_t launches;
  double flops;      /* executed: sum of 2*M*N*K*(groups or segments) */
  double ms;
} CartnetGemmProfile;
int cartnet_profile_gemm(int32_t enable);
/* Restrict the timing to launches of one variant (the value CartnetGemmProfile.variant reports; < 0: all variants):
 * bench.py prices every variant during its warm-up steps and only the dominant one inside the timed region, so the
 * event pairs of the other ~50 launches per step do not sit in the measured steps. */
int cartnet_profile_gemm_only(int32_t variant);
/* Of the launches that qualify, time every n-th (n >= 1; resets the count).  An event pair in front of and behind a
 * launch costs the stream ~4 us; bench.py samples one in four launches of the dominant variant inside its timed region
 * (four is coprime with the nine such launches of a step, so the sample walks over all of them). */
int cartnet_profile_gemm_every(int32_t n);
int cartnet_profile_gemm_read(CartnetGemmProfile* out, int32_t max_entries);

/* For each job j < njobs (<= 4; host arrays of device pointers):
 * outs[j][m*ldo + n] = sum_{s<splitk} slabs[j][s*M*N + m*N + n]  (fixed order s = 0..splitk-1). */
int cartnet_splitk_reduce(const float* const* slabs, float* const* outs, int32_t njobs, int32_t splitk, int32_t M,
                          int32_t N, int32_t ldo, void* stream);

/* Partial sums.  Kernels that reduce over rows (BatchNorm statistics, bias / affine gradients) never use atomics:
 * every workgroup writes one fp64 row of per-column partial sums (fp64 so that var = E[v^2] - mean^2 keeps its
 * digits), and a finalise kernel adds the rows in fixed order.
 * For each job j < njobs (<= 8; host arrays of device pointers, all with the same nparts and N):
 * outs[j][n] = (float) sum_{p<nparts} parts[j][p*N + n].
 * The partial rows are CONSUMED: when nparts > 256 they are first folded in place to 32 rows (row r <- rows r, r+32,
 * r+64, ... in that order, one launch over many workgroups), so that the few finalising blocks have 32 rows left.
 * The _f32 form (single job) reads the fp32 partial rows of the head kernels. */
int cartnet_colsum_finalize(double* const* parts, float* const* outs, int32_t njobs, int32_t nparts, int32_t N,
                            void* stream);
/* The same with an optional second destination per job (outs2, or outs2[j], may be NULL): the sums land in a scratch
 * row a following kernel reads AND in the gradient tensor, without a device-to-device copy in between. */
int cartnet_colsum_finalize2(double* const* parts, float* const* outs, float* const* outs2, int32_t njobs,
                             int32_t nparts, int32_t N, void* stream);
int cartnet_colsum_finalize_f32(const float* parts, int32_t nparts, int32_t N, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Graph layout.  Replaces PyG MessagePassing's per-layer index_select bookkeeping (called at
 * models/cartnet.py:218) with a once-per-batch CSR/CSC build.  edge_index is the PyG tensor [2,E] int64,
 * row 0 = source j, row 1 = target i, row 1 sorted ascending (dataset/utils.py:235).
 *   src32/tgt32 [E]  int32 copies;   rowptr [N+1]: edges of target t are [rowptr[t], rowptr[t+1]);
 *   colptr [N+1] / perm [E]: edges with source j are perm[colptr[j] .. colptr[j+1]), in ascending edge order
 *   (stable), so by-source reductions are bitwise reproducible;
 *   status[0] is set non-zero on the device if row 1 is not sorted or an index is outside [0,N).
 * graph_ptr [Bg+1] int64 node offsets of each crystal (Batch.ptr); crystals never share edges, which the
 * CSC build uses to work one crystal per workgroup.
 * ---------------------------------------------------------------------------------------------------- */
int cartnet_csr_build(const int64_t* edge_index, int64_t E, int32_t N, const int64_t* graph_ptr, int32_t Bg,
                      int32_t* src32, int32_t* tgt32, int32_t* rowptr, int32_t* colptr, int32_t* perm,
                      int32_t* status, void* stream);
/* The CSC half of cartnet_csr_build on its own (colptr [N+1], perm [E]: the stable permutation of the edges by source),
 * from the src32 / rowptr a cartnet_csr_build call with colptr = perm = NULL produced: only backward reads it, so
 * cartnet_model_forward queues it on its second stream.  ORs the same status bits (4, 8) into `status`, which the
 * csr call must have reset first (stream order). */
int cartnet_csc_build(const int32_t* src32, const int32_t* rowptr, const int64_t* graph_ptr, int32_t Bg, int32_t N,
                      int64_t E, int32_t* colptr, int32_t* perm, int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Cartesian edge encoding (models/cartnet.py:159 + models/utils.py:56-61,87-91):
 *   feat[e, 0:R]   = cutoff(d_e; radius) * exp(-betas[k] * (exp(-(5/radius) d_e) - means[k])^2)
 *   feat[e, R:R+3] = cart_dir[e]   (omitted when invariant != 0);   columns up to ldf are zero-filled
 *   env[e]         = 0.5 (cos(pi d_e / env_radius) + 1) (d_e < env_radius)    (layer envelope, cartnet.py:201,241)
 * ---------------------------------------------------------------------------------------------------- */
int cartnet_edge_features(const float* cart_dist, const float* cart_dir, const float* means, const float* betas,
                          int64_t E, int32_t R, int32_t invariant, float radius, float env_radius,
                          float* feat, int32_t ldf, float* env, void* stream);

/* Node embedding (models/cartnet.py:145-151):
 *   x0[n, c] = emb[z[n], c] (if emb) + (temperature[batch[n]] * wt[c] + bt[c]) (if wt) + bias[c] (if bias).
 * emb has n_types rows; z == NULL with a table means row 0 for every atom (the single learned row of the ablation
 * without atom types and temperature, cartnet.py:150-151).  nn.Embedding raises on an index outside the table
 * (cartnet.py:145); here such an atomic number -- and a batch id outside [0, Bg) -- is clamped (no out-of-bounds read)
 * and reported: status[0] |= 16 (atomic number) / 32 (batch id).  status may be NULL (clamp only). */
int cartnet_node_embed(const int64_t* z, const int64_t* batch, const float* temperature, const float* emb,
                       const float* wt, const float* bt, const float* bias, int32_t N, int32_t C, int32_t n_types,
                       int32_t Bg, int32_t* status, float* x0, void* stream);
/* Backward of the above, temperature projection / bias part: parts_w[p][c] / parts_b[p][c] = per-block partial sums
 * over atoms of T[batch[n]]*dx0[n,c] and dx0[n,c] (p < cartnet_node_nparts(N)); reduce with cartnet_colsum_finalize
 * into the gradients of temperature_proj_atom.weight and .bias (or encoder.bias).
 * The embedding-table gradient demb[a] = sum_{n: z[n]=a} dx0[n] is cartnet_sort_by_key(z) once per batch +
 * cartnet_segment_sum_long (fixed order, no atomics). */
int cartnet_node_nparts(int32_t N);
int cartnet_node_embed_bwd(const int64_t* batch, const float* temperature, const float* dx0, int32_t N, int32_t C,
                           int32_t Bg, double* parts_w, double* parts_b, void* stream);

/* Stable counting sort of N items by key in [0, nkeys), nkeys <= 512: items with key k are
 * perm[ptr[k] .. ptr[k+1]) in ascending item order.  A key outside the range is clamped into it (perm is always a full
 * permutation of 0..N-1) and reported: status[0] |= 16. */
int cartnet_sort_by_key(const int64_t* keys, int32_t N, int32_t nkeys, int32_t* perm, int32_t* ptr, int32_t* status,
                        void* stream);

/* ------------------------------------------------------------------------------------------------------
 * BatchNorm statistics (nn.BatchNorm1d at models/cartnet.py:198-199, used at :238 over E rows and :269 over
 * N rows).  parts_sum / parts_sq hold `nparts` per-column partial sums.  training != 0: batch mean and biased
 * variance -> mean_rstd[0:C] = mean, mean_rstd[C:2C] = 1/sqrt(var + eps); running stats updated in place with
 * `momentum` and the unbiased variance; num_batches_tracked += 1.  training == 0: mean_rstd from running stats.
 * The partial rows are consumed (folded in place when nparts > 256, see cartnet_colsum_finalize).
 * ---------------------------------------------------------------------------------------------------- */
/* BatchNorm groups.  The reference's ADP recipe trains on micro-batches of 4 crystals and accumulates 16 of them per
 * optimiser step (scripts/train_cartnet_adp.sh:4, train/train.py:183-189): every micro-batch is a forward call of its
 * own, so BatchNorm normalises over the 4 crystals of that call.  Launch-bound on any GPU.  Here the 16 micro-batches
 * can travel through the network as ONE batch of 64 crystals whose consecutive crystals form G "groups": every
 * statistics-carrying kernel then reduces per group (its nodes [node_gptr[g], node_gptr[g+1]), its edges
 * [edge_gptr[g], edge_gptr[g+1]) -- edges are sorted by target, so a group's edges are contiguous), normalises every
 * row with its group's statistics, and the running statistics take the G momentum updates in group order.  Results
 * equal G separate calls.  A NULL CartnetGroups pointer means one group = the whole batch. */
typedef struct CartnetGroups {
  const int32_t* node_gptr;   /* [G+1] first node of every group (device)                                       */
  const int32_t* edge_gptr;   /* [G+1] first edge of every group (device) = rowptr[node_gptr[g]]                */
  int32_t G;                  /* number of groups                                                                */
  int32_t edge_parts;         /* workgroups per group of the per-edge kernels: their partial sums are [G][edge_parts][D] */
  int32_t node_parts;         /* the same for the per-node kernels                                              */
} CartnetGroups;
/* node_gptr[g] = graph_ptr[min(g*group_size, Bg)], edge_gptr[g] = rowptr[node_gptr[g]] for g = 0..G, G = ceil(Bg/group_size) */
int cartnet_group_ptrs(const int64_t* graph_ptr, int32_t Bg, int32_t group_size, const int32_t* rowptr, int32_t G,
                       int32_t* node_gptr, int32_t* edge_gptr, void* stream);
/* Per-group column sums and sums of squares of x [rows, C] (row stride ld) over the groups' EDGE ranges, as fp64 partial
 * rows [G][edge_parts][C] for cartnet_bn_finalize: the BatchNorm statistics of the gate pre-activation, which a
 * single-group run takes from the GEMM epilogue (a 128-row tile may straddle two groups). */
int cartnet_colstats_grouped(const float* x, int32_t ld, int32_t C, const CartnetGroups* groups, double* parts_sum,
                             double* parts_sq, void* stream);
/* Backward statistics per group: sums[g][0:D] / sums[g][D:2D] = column sums of group g's rows of parts_a / parts_b
 * ([G][parts][D], parts = edge_parts if over_edges else node_parts); grad_a / grad_b (may be NULL) = the sums over all
 * groups (the BatchNorm affine gradients).  The partial rows are consumed (row 0 of every group is overwritten). */
int cartnet_group_sums_finalize(double* parts_a, double* parts_b, int32_t D, const CartnetGroups* groups,
                                int32_t over_edges, float* sums, float* grad_a, float* grad_b, void* stream);

/* groups != NULL: parts are [G][parts][C] with parts = edge_parts if parts_over_edges (rows written by a per-edge kernel)
 * else node_parts; the row counts are the groups' edge counts if count_over_edges else their node counts (nparts and
 * count are ignored); mean_rstd is [G][2C]; the running statistics receive G updates in group order and
 * num_batches_tracked += G. */
int cartnet_bn_finalize(double* parts_sum, double* parts_sq, int32_t nparts, int64_t count, int32_t C,
                        float eps, float momentum, int32_t training, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* mean_rstd, const CartnetGroups* groups,
                        int32_t parts_over_edges, int32_t count_over_edges, void* stream);

/* Sync-BatchNorm across data-parallel ranks (optional; the reference's BatchNorm1d layers at
 * models/cartnet.py:198-199,238,269 see one process's batch): the column sums behind every BatchNorm statistic are
 * summed over the ranks before the statistic is formed, so that N ranks with a shard each compute what one process
 * would on the union batch.  Three small steps around the caller's all-reduce:
 *   cartnet_bn_sync_gather:   row[0:C] = column sums of parts_a, row[C:2C] = of parts_b, row[2C] = local_count (fp64);
 *                             out_a / out_b (optional, fp32 [C]) receive the LOCAL sums (backward: the BatchNorm affine
 *                             gradients stay local -- the gradient all-reduce adds them up like every other gradient)
 *   (caller: in-place SUM all-reduce of the 2C+1 doubles of `row`)
 *   cartnet_bn_finalize_row:  forward: mean / rstd / running statistics from the summed row (count = row[2C])
 *   cartnet_bn_sync_scale:    backward: sums[0:2C] = row[0:2C] * local_count / row[2C] -- the apply kernels divide by
 *                             the local count, so the pre-scaled sums give them sum_global / count_global. */
int cartnet_bn_sync_gather(const double* parts_a, const double* parts_b, int32_t nparts, int32_t C, int64_t local_count,
                           double* row, float* out_a, float* out_b, void* stream);
int cartnet_bn_finalize_row(const double* row, int32_t C, float eps, float momentum, float* running_mean,
                            float* running_var, int64_t* num_batches_tracked, float* mean_rstd, void* stream);
int cartnet_bn_sync_scale(const double* row, int32_t C, int64_t local_count, float* sums, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Neighbour-equalised gate + aggregation (models/cartnet.py:238-243 message, :259 scatter-sum, :225 edge
 * residual).  gs [E, 2D]: columns 0:D = pre-BatchNorm gate g, D:2D = sender s.  One wavefront walks the edges of
 * one target node in edge order (CSR), so the sum order equals the CPU scatter_add_ order:
 *   sigma = env[e] * sigmoid((g - mean) * rstd * gamma + beta);  e_out = e_in + sigma;
 *   aggr[t] = sum_{e in row t} sigma * s[e].
 * Also writes per-block column partial sums of aggr / aggr^2 (BatchNorm over nodes) to parts_sum / parts_sq
 * [nparts = cartnet_gate_scatter_nparts(N)][D].
 * ---------------------------------------------------------------------------------------------------- */
int cartnet_gate_scatter_nparts(int32_t N);
int cartnet_gate_scatter_fwd(const float* gs, const float* e_in, const float* env, const int32_t* rowptr,
                             const float* mean_rstd, const float* gamma, const float* beta, int32_t N, int32_t D,
                             float* e_out, float* aggr, double* parts_sum, double* parts_sq,
                             const CartnetGroups* groups, void* stream);

/* The same, and per target t also  bc[t, 0:D] = sum_e s w,  bc[t, D:2D] = sum_e s w ghat  with w = env z (1 - z),
 * z = sigmoid(bn(g)), ghat = (g - mean) rstd, the sums over the edges of row t (bc: [N, 2D] fp32, 16-byte aligned).  With
 * them the BatchNorm backward of the gate needs no pass over the edges for its two sums (cartnet.py:238 backward):
 *   sum_e dbn      = sum_e de_out w      + sum_t daggr[t] bc[t, 0:D]
 *   sum_e dbn ghat = sum_e de_out w ghat + sum_t daggr[t] bc[t, D:2D]
 * -- the first terms come from the epilogue of the product that writes de_out (CartnetGemmArgs.gst_*), the second from
 * cartnet_node_update_bwd_apply_bc, which has daggr in registers.  One BatchNorm group only. */
int cartnet_gate_scatter_fwd_bc(const float* gs, const float* e_in, const float* env, const int32_t* rowptr,
                                const float* mean_rstd, const float* gamma, const float* beta, int32_t N, int32_t D,
                                float* e_out, float* aggr, double* parts_sum, double* parts_sq, float* bc, void* stream);

/* de_out may be NULL in both backward passes (last layer: the head does not read the edge features).
 * Backward, pass 1 (statistics): with dm = daggr[tgt], z = sigmoid(bn(g)), dbn = (dm*s + de_out) * env * z(1-z):
 * column partial sums of dbn and dbn * ghat (ghat = (g-mean)*rstd) -> parts_a, parts_b [nparts][D]. */
int cartnet_gate_scatter_bwd_stats(const float* gs, const float* de_out, const float* daggr, const float* env,
                                   const int32_t* rowptr, const float* mean_rstd, const float* gamma,
                                   const float* beta, int32_t N, int32_t D, double* parts_a, double* parts_b,
                                   const CartnetGroups* groups, void* stream);
/* Backward, pass 2 (apply), in place on gs: g <- dg = gamma*rstd*(dbn - sum_dbn/E - ghat*sum_dbn_ghat/E)
 * (the two mean terms are dropped when training == 0), s <- ds = dm * sigma.  sums[0:D] = sum dbn,
 * sums[D:2D] = sum dbn*ghat (from cartnet_colsum_finalize).  Column partial sums of dg and ds (bias gradients
 * of the second Linears) -> parts_dg, parts_ds [nparts][D]. */
int cartnet_gate_scatter_bwd_apply(float* gs, const float* de_out, const float* daggr, const float* env,
                                   const int32_t* rowptr, const float* mean_rstd, const float* gamma,
                                   const float* beta, const float* sums, int64_t E, int32_t training, int32_t N,
                                   int32_t D, double* parts_dg, double* parts_ds, const CartnetGroups* groups,
                                   void* stream);

/* Half-storage forms ("bf16 storage / fp32 accumulate"; used by the model when
 * CartnetModel.half_storage is set at gemm_precision 2): the same three kernels with gs [E, 2D] kept in memory as bf16
 * (the apply pass writes dg / ds back as bf16); everything else -- e, aggr, sums, statistics -- stays fp32 / fp64. */
int cartnet_gate_scatter_fwd_h(const void* gs_bf16, const float* e_in, const float* env, const int32_t* rowptr,
                               const float* mean_rstd, const float* gamma, const float* beta, int32_t N, int32_t D,
                               float* e_out, float* aggr, double* parts_sum, double* parts_sq,
                               const CartnetGroups* groups, void* stream);
int cartnet_gate_scatter_bwd_stats_h(const void* gs_bf16, const float* de_out, const float* daggr, const float* env,
                                     const int32_t* rowptr, const float* mean_rstd, const float* gamma,
                                     const float* beta, int32_t N, int32_t D, double* parts_a, double* parts_b,
                                     const CartnetGroups* groups, void* stream);
int cartnet_gate_scatter_bwd_apply_h(void* gs_bf16, const float* de_out, const float* daggr, const float* env,
                                     const int32_t* rowptr, const float* mean_rstd, const float* gamma,
                                     const float* beta, const float* sums, int64_t E, int32_t training, int32_t N,
                                     int32_t D, double* parts_dg, double* parts_ds, const CartnetGroups* groups,
                                     void* stream);

/* Row-segment sums: out[t, :] = sum_{k in [ptr[t], ptr[t+1])} rows[(perm ? perm[k] : k), :]  (fixed order).
 * Backward of the two index_selects PyG performs per layer (x_i by target: perm = NULL; x_j by source: CSC). */
int cartnet_segment_sum(const float* rows, int32_t ld, const int32_t* ptr, const int32_t* perm, int32_t N,
                        int32_t W, float* out, int32_t ldo, void* stream);
/* rows kept as bf16 (ld in elements), fp32 sums */
int cartnet_segment_sum_h(const void* rows_bf16, int32_t ld, const int32_t* ptr, const int32_t* perm, int32_t N,
                          int32_t W, float* out, int32_t ldo, void* stream);
/* Both index_select backwards of a layer in one launch: out_t[t, :] = the by-target sum (rowptr, rows in place), out_s[t, :]
 * = the by-source sum (colptr + perm), same values and summation orders as two cartnet_segment_sum calls.  The work items
 * of a node's two sums are adjacent, so the rows of a crystal are read twice within microseconds and the second read
 * comes from cache instead of HBM.  `ochunk`: where the 256-column chunk j of a sum lands in its output row --
 * out[t * ldo + j * ochunk + 0..255]; 256 = the columns keep their places (CartNet: dPn = [by target | by source]); iComformer
 * with C = 256 passes 2C and out_s = out_t + C to interleave them as [i key | j key | i msg | j msg]. */
int cartnet_segment_sum_pair(const float* rows, int32_t ld, const int32_t* rowptr, const int32_t* colptr, const int32_t* perm,
                             int32_t N, int32_t W, float* out_t, float* out_s, int32_t ldo, int32_t ochunk, void* stream);
int cartnet_segment_sum_pair_h(const void* rows_bf16, int32_t ld, const int32_t* rowptr, const int32_t* colptr,
                               const int32_t* perm, int32_t N, int32_t W, float* out_t, float* out_s, int32_t ldo,
                               int32_t ochunk, void* stream);
/* Same sums for FEW, very uneven segments (atoms grouped by element): sorted positions [0,total) are cut into
 * 128-row chunks summed by independent wavefronts into tmp [total, W] (one partial row per run), then each segment
 * adds its partial rows in position order -- evenly loaded and still bitwise reproducible. */
int cartnet_segment_sum_long(const float* rows, int32_t ld, const int32_t* ptr, const int32_t* perm, int32_t nseg,
                             int32_t total, int32_t W, float* tmp, float* out, int32_t ldo, void* stream);
/* The same two passes with the partial rows NUMBERED (a run that starts a chunk: position / chunk; any other run starts
 * its segment: number of chunks + segment) instead of stored at their position: tmp is
 * [cartnet_segment_chunked_rows(nseg, total), W] instead of [total, W].  For sums over ALL edges per crystal -- iComformer's
 * per-(crystal, lattice vector) terms, comformer_conv.py:160-170 backward -- where `total` rows would be a gigabyte and
 * one wave per segment (cartnet_segment_sum) 64 waves on the whole chip. */
int32_t cartnet_segment_chunked_rows(int32_t nseg, int32_t total);
int cartnet_segment_sum_chunked(const float* rows, int32_t ld, const int32_t* ptr, const int32_t* perm, int32_t nseg,
                                int32_t total, int32_t W, float* tmp, float* out, int32_t ldo, void* stream);
/* The same for rows that are THREE pieces of W / 3 columns side by side (iComformer's edge layer: every edge's three
 * lattice-vector rows [E, 3, 2C] seen as [E, 6C]; perm = NULL), with the pieces' sum per row as a by-product:
 * fold_out[p, c] = (rows[p, c] + rows[p, W/3 + c]) + rows[p, 2 W/3 + c]  ([total, W/3] contiguous) -- each row is read once for
 * the per-crystal sums over all edges AND the per-edge sum over the lattice vectors (comformer_conv.py:160-193 backward). */
int cartnet_segment_sum_chunked_fold3(const float* rows, int32_t ld, const int32_t* ptr, int32_t nseg, int32_t total, int32_t W,
                                      float* tmp, float* out, int32_t ldo, float* fold_out, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Node update (models/cartnet.py:269 norm2, :223 SiLU + residual):  x_out = silu(bn(aggr)) + x_in.
 * ---------------------------------------------------------------------------------------------------- */
int cartnet_node_update_fwd(const float* aggr, const float* x_in, const float* mean_rstd, const float* gamma,
                            const float* beta, int32_t N, int32_t D, float* x_out, const CartnetGroups* groups,
                            void* stream);
/* Backward pass 1: dxn = dx_out * silu'(xn); column partial sums of dxn and dxn*ahat -> parts_a/parts_b
 * [cartnet_node_nparts(N)][D]. */
int cartnet_node_update_bwd_stats(const float* aggr, const float* dx_out, const float* mean_rstd,
                                  const float* gamma, const float* beta, int32_t N, int32_t D, double* parts_a,
                                  double* parts_b, const CartnetGroups* groups, void* stream);
/* Backward pass 2: daggr = gamma*rstd*(dxn - sum_a/N - ahat*sum_b/N) (mean terms dropped when training == 0). */
int cartnet_node_update_bwd_apply(const float* aggr, const float* dx_out, const float* mean_rstd,
                                  const float* gamma, const float* beta, const float* sums, int32_t training,
                                  int32_t N, int32_t D, float* daggr, const CartnetGroups* groups, void* stream);
/* The same (one BatchNorm group), and the column partial sums of daggr * bc[:, 0:D] -> parts_a and daggr * bc[:, D:2D] ->
 * parts_b [cartnet_node_nparts(N)][D] (bc from cartnet_gate_scatter_fwd_bc): the atoms' share of the gate's
 * BatchNorm-backward sums, taken while daggr is in registers. */
int cartnet_node_update_bwd_apply_bc(const float* aggr, const float* dx_out, const float* mean_rstd,
                                     const float* gamma, const float* beta, const float* sums, int32_t training,
                                     int32_t N, int32_t D, float* daggr, const float* bc, double* parts_a,
                                     double* parts_b, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Cholesky ADP head (models/cartnet.py:293-305).  hid [N, H] = pre-activation of head.MLP.0 for every atom;
 * atoms with mask != 0 get an output row (in atom order): p = W2 silu(hid) + b2 (6 values),
 * diag = softplus(p[0:3]), L upper-triangular (L01 = p3, L02 = p4, L12 = p5), pred = L^T L  [M,3,3].
 * out_index [N] receives the output row of each masked atom (-1 otherwise).
 * ---------------------------------------------------------------------------------------------------- */
int cartnet_mask_index(const uint8_t* mask, int32_t N, int32_t* out_index, int32_t* count, void* stream);
int cartnet_cholesky_head_fwd(const float* hid, const int32_t* out_index, const float* W2, const float* b2,
                              int32_t N, int32_t H, float* p6, float* pred, void* stream);
/* Backward: dhid [N,H] (zero rows for unmasked atoms); per-block partial sums -> parts [cartnet_node_nparts(N)][7*H+8],
 * row layout: 6*H of dW2, 6 of db2, 2 pad, H column sums of dhid (= gradient of head.MLP.0.bias). */
int cartnet_cholesky_head_bwd(const float* hid, const int32_t* out_index, const float* W2, const float* p6,
                              const float* dpred, int32_t N, int32_t H, float* dhid, float* parts, void* stream);

/* Scalar head (models/cartnet.py:323-327): v[n] = w2 . silu(hid[n]) + b2 ; out[g] = mean of v over the atoms of
 * crystal g (graph_ptr int64 [Bg+1]). */
int cartnet_scalar_head_fwd(const float* hid, const float* w2, const float* b2, const int64_t* graph_ptr,
                            int32_t Bg, int32_t H, float* out, void* stream);
int cartnet_scalar_head_bwd(const float* hid, const float* w2, const int64_t* graph_ptr, const int64_t* batch,
                            const float* dout, int32_t N, int32_t Bg, int32_t H, float* dhid, float* parts,
                            void* stream);

/* dst[j][c*ldd[j] + r] = src[j][r*lds[j] + c] for njobs <= 40 matrices (host arrays).  The model transposes its weights
 * once per forward so that forward GEMMs read them with b_kstrided = 1 (coalesced rows). */
int cartnet_transpose(const float* const* src, float* const* dst, const int32_t* rows, const int32_t* cols,
                      const int32_t* lds, const int32_t* ldd, int32_t njobs, void* stream);


/* ------------------------------------------------------------------------------------------------------
 * Whole-network entry points: CartNet.forward (models/cartnet.py:65-73) and its backward as ONE host call each.
 * They enqueue the kernels above in the order the reference executes its layers; nothing else happens on the host
 * (no allocation, no synchronisation), so a training step costs two ABI calls and can be captured in a hipGraph.
 *
 * CartnetModel: borrowed pointers to every parameter / buffer in the reference's state_dict layout.
 * CartnetGrads: one output pointer per parameter (same fields), each receiving a FRESH gradient (not accumulated).
 * CartnetBatch: the PyG batch attributes the model reads + sizes.
 * Workspace: one caller-owned device buffer of cartnet_workspace_bytes(...) bytes, 256-byte aligned.  It holds every
 *   intermediate including what backward needs, so it must stay untouched between forward and backward of a batch.
 * ---------------------------------------------------------------------------------------------------- */
#define CARTNET_MAX_LAYERS 16

typedef struct CartnetLayerParams {
  float* gate0_w; float* gate0_b; float* gate2_w; float* gate2_b;     /* MLP_gate.0 [D,3D], MLP_gate.2 [D,D] */
  float* aggr0_w; float* aggr0_b; float* aggr2_w; float* aggr2_b;     /* MLP_aggr.*                           */
  float* norm_w; float* norm_b; float* norm2_w; float* norm2_b;       /* BatchNorm affine                     */
} CartnetLayerParams;

typedef struct CartnetLayerBuffers {
  float* norm_mean; float* norm_var; int64_t* norm_nbt;               /* running statistics (updated in training) */
  float* norm2_mean; float* norm2_var; int64_t* norm2_nbt;
} CartnetLayerBuffers;

typedef struct CartnetParams {                  /* used both for parameters (inputs) and for gradients (outputs) */
  float* embedding;                             /* [n_types, 2D] or NULL                 */
  float* temp_w; float* temp_b;                 /* temperature_proj_atom [2D,1], [2D] or NULL */
  float* enc_bias;                              /* encoder.bias [2D] or NULL             */
  float* atom_w; float* atom_b;                 /* encoder_atom.1 [D,2D], [D]            */
  float* edge0_w; float* edge0_b;               /* encoder_edge.0 [2D,R(+3)], [2D]       */
  float* edge2_w; float* edge2_b;               /* encoder_edge.2 [D,2D], [D]            */
  CartnetLayerParams layer[CARTNET_MAX_LAYERS];
  float* head0_w; float* head0_b;               /* head.MLP.0 [D/2,D], [D/2]             */
  float* head2_w; float* head2_b;               /* head.MLP.2 [6 or 1, D/2]              */
} CartnetParams;

typedef int (*CartnetAllReduceFn)(void* user, double* buf, int64_t count, void* stream);
/* Gradient buckets ("all-reduce ... overlapped with the tail of backward"; the reference's accumulation
 * boundary is train/train.py:186-189).  cartnet_model_backward calls this once per bucket, as soon as every kernel that
 * writes the bucket's gradients has been ENQUEUED; `stream` is ordered after all of them (the weight-gradient stream
 * for the layer and head buckets, the caller's stream for the encoder bucket, which completes last), so the callee may
 * enqueue an all-reduce of that slice there while backward goes on.  bucket: 0..L-1 = layer l (called in the order
 * L-1 .. 0), L = head (called with layer L-1), L+1 = encoder (last, after the streams have joined). */
typedef int (*CartnetGradReadyFn)(void* user, int32_t bucket, void* stream);

typedef struct CartnetModel {
  int32_t D, R, L;                              /* dim_in, dim_rbf, num_layers                       */
  int32_t invariant, use_temperature, atom_types, cholesky, n_types;
  int32_t use_envelope[CARTNET_MAX_LAYERS];
  float radius, env_radius, bn_eps, bn_momentum;
  int32_t gemm_precision;                       /* CartnetGemmArgs.precision for every GEMM of the network */
  int32_t bn_group_size;                        /* > 0: consecutive crystals form BatchNorm groups of this size
                                                   (CartnetGroups: the reference's micro-batches in one pass);
                                                   <= 0: one group, the whole batch                          */
  const float* rbf_means; const float* rbf_betas;
  CartnetParams p;
  CartnetLayerBuffers buf[CARTNET_MAX_LAYERS];
  /* Sync-BatchNorm (training mode only, not with bn_group_size): when set, called once per BatchNorm in forward and
     once in backward with a device buffer of `count` doubles; must enqueue an in-place SUM all-reduce over the ranks,
     ordered after the work already queued on `stream` and before what is queued next.  Non-zero return = error. */
  CartnetAllReduceFn bn_allreduce;
  void* bn_allreduce_user;
  /* gemm_precision == 2 only: the three edge-sized tensors of every layer that are kept for backward / handed from
     kernel to kernel -- pre [E, 2D], gs / dgs [E, 2D], dpre [E, 2D] -- and the edge encoder's pre-activation [E, 2D]
     live in the workspace as bf16 (the MFMA operands
     are bf16 at this precision anyway; accumulation, BatchNorm statistics, the residual streams x / e and every
     gradient of a parameter stay fp32).  SURVEY.md 8d config 3: "bf16 storage / fp32 accumulate".  Not with
     bn_group_size (the per-group statistics pass reads gs as fp32). */
  int32_t half_storage;
  /* optional, cartnet_model_backward only: see CartnetGradReadyFn.  Non-zero return = error. */
  CartnetGradReadyFn grad_ready;
  void* grad_ready_user;
} CartnetModel;

typedef struct CartnetBatch {
  const int64_t* z;            /* [N] atomic numbers                                  */
  const int64_t* batch;        /* [N] crystal of each atom (sorted)                   */
  const int64_t* graph_ptr;    /* [Bg+1] atom offsets per crystal                     */
  const int64_t* edge_index;   /* [2,E], row 1 (target) sorted ascending              */
  const float* temperature;    /* [Bg] or NULL                                        */
  const float* cart_dist;      /* [E]                                                 */
  const float* cart_dir;       /* [E,3] or NULL (invariant)                           */
  const uint8_t* non_h_mask;   /* [N] or NULL (scalar head)                           */
  int32_t N, Bg, M;            /* atoms, crystals, masked atoms (rows of pred)        */
  int64_t E;
} CartnetBatch;

/* Inference-mode fusion of a layer's second Linears with the gate (reference models/cartnet.py:230-262 in eval mode, where
 * the edge BatchNorm uses its running statistics): g = silu(pre[:, :D]) W2g^T + b, s = silu(pre[:, D:]) W2a^T + b,
 * sigma = env * sigmoid(gamma (g - mean) rstd + beta), e_out = e_in + sigma, aggr[t] = sum over the edges of target t of
 * sigma * s (edge order; edges sorted by target) -- gs [E, 2D] never reaches memory.  fp32 MFMA; D a multiple of 256;
 * img_*: cartnet_gemm_pack_b images of the [K = D, N = D] operands W2^T; bnd: cartnet_gate_gemm_eval_workspace(E, D)
 * bytes; aggr rows of atoms without edges are set to zero.  Two launches (product + boundary fix-up), no atomics. */
typedef struct CartnetGateGemmArgs {
  const float* pre;                      /* [E, >= 2D] pre-activations [gate | sender], row stride ldp */
  int32_t ldp;
  const void *img_gate, *img_aggr;
  const float *bias_gate, *bias_aggr;    /* [D] */
  const float* mean_rstd;                /* [2D]: mean | rstd of the edge BatchNorm */
  const float *gamma, *beta;             /* [D] */
  const float* env;                      /* [E] envelope or NULL */
  const float* e_in;                     /* [E, D] */
  float* e_out;                          /* [E, D] */
  const int32_t* tgt;                    /* [E] ascending */
  const int32_t* rowptr;                 /* [N + 1] */
  float* aggr;                           /* [N, D] */
  float* bnd;                            /* workspace */
  int64_t E;
  int32_t N, D;
} CartnetGateGemmArgs;
size_t cartnet_gate_gemm_eval_workspace(int64_t E, int32_t D);
int cartnet_gate_gemm_eval(const CartnetGateGemmArgs* args, void* stream);

size_t cartnet_workspace_bytes(const CartnetModel* model, int32_t N, int64_t E, int32_t Bg, int32_t M,
                               int32_t need_backward);
/* pred [M,3,3] (Cholesky head) or [Bg] (scalar head); x_out [N,D] and e_out [E,D] receive the final node / edge
 * features (what the reference leaves in batch.x / batch.edge_attr).  status[0] (device int32) reports graph-layout
 * problems (cartnet_csr_build bits).  need_backward = 0 lets the layers reuse one set of activation buffers.
 * aux_stream (optional second hipStream_t, may be NULL): the work that does not depend on the graph (weight transposes
 * and DMA images) and the atom branch of the encoder with the first layer's node terms are enqueued there, next to
 * the layout build and the edge encoder on `stream`; ordered with events, joined before the first layer, so the
 * caller sees single-stream semantics on `stream`. */
int cartnet_model_forward(const CartnetModel* model, const CartnetBatch* batch, void* workspace, size_t workspace_bytes,
                          int32_t training, int32_t need_backward, float* pred, float* x_out, float* e_out,
                          int32_t* status, void* stream, void* aux_stream);
/* dpred: gradient of the loss w.r.t. pred.  grads: where each parameter's gradient is written (every non-NULL
 * parameter of the model must have a destination).  Consumes the workspace of the matching forward call.
 * aux_stream (optional second hipStream_t): when given, the parameter-gradient work (weight-gradient GEMMs, split-K
 * and bias reductions -- a third of the FLOPs, none of it on the dx/de dependency chain) is enqueued there and
 * overlaps the main stream; the call orders the two streams with events and joins them before returning to the
 * caller's stream order, so the caller sees single-stream semantics on `stream`. */
int cartnet_model_backward(const CartnetModel* model, const CartnetBatch* batch, void* workspace, size_t workspace_bytes,
                           int32_t training, const float* dpred, const float* x_out, const CartnetParams* grads,
                           void* stream, void* aux_stream);

/* ------------------------------------------------------------------------------------------------------
 * iComformer (BASELINE configs[4]; models/comformer.py:75-132, models/comformer_conv.py:21-193) as ONE host call per
 * direction, like CartNet above: the embeddings and RBF branches, attention layer 0, the edge-update layer on 3E rows
 * (edge x lattice vector), attention layers 1-3 and the Cholesky head are sequenced here on the kernels of this library;
 * every weight image / transposed copy of the step is made by two batched launches at the start of forward.
 * CartnetIcfConv: one ComformerConv (att_layers.l) or the ComformerConv_edge (edge_update_layer: lin_edge has no bias,
 *   key_e / value_e are lin_key_e{1,2,3} / lin_value_e{1,2,3}; the two parameters the reference declares and never uses,
 *   lemb and lin_edge_len, have no field and get no gradient).  CartnetIcfParams is used for parameters and gradients.
 * cell [Bg,9] comes next to the CartnetBatch (comformer.py:118,120 is the only reader); batch->cart_dir is required.
 * Workspace: cartnet_icomformer_workspace_bytes(...), untouched between forward and backward of a batch.
 * aux_stream (optional second hipStream_t): graph-independent preparation in forward; weight gradients and the two
 * branches nothing on the atom-gradient chain waits for (lin_edge backward, the angle branch) in backward; joined before
 * the calls return.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct CartnetIcfConv {
  float *query_w, *query_b, *key_w, *key_b, *value_w, *value_b;      /* lin_query / lin_key / lin_value [C,C], [C]      */
  float *edge_w, *edge_b;                                            /* lin_edge [C,C], [C] (edge layer: bias NULL)      */
  float *concate_w, *concate_b;                                      /* lin_concate                                      */
  float *key0_w, *key0_b, *key2_w, *key2_b;                          /* key_update.0 [C,3C] / .2 [C,C]                   */
  float *msg0_w, *msg0_b, *msg2_w, *msg2_b;                          /* lin_msg_update.0 / .2                            */
  float *bn_w, *bn_b, *bn_att_w, *bn_att_b;                          /* BatchNorm affine                                 */
  float *key_e_w[3], *key_e_b[3], *value_e_w[3], *value_e_b[3];      /* edge layer only, else NULL                       */
} CartnetIcfConv;

typedef struct CartnetIcfBn {            /* running statistics of one BatchNorm1d (updated in training)                */
  float* mean; float* var; int64_t* nbt;
} CartnetIcfBn;

typedef struct CartnetIcfParams {
  float *embedding;                      /* [n_types, C]                                                               */
  float *temp_w, *temp_b;                /* temperature_proj_atom [C,1], [C]                                           */
  float *rbf_w, *rbf_b;                  /* rbf.1 [C, bins = C], [C]       (distances and lattice lengths)             */
  float *rbf_angle_w, *rbf_angle_b;      /* rbf_angle.1                                                                */
  CartnetIcfConv att[4];
  CartnetIcfConv edge;
  float *head0_w, *head0_b, *head2_w, *head2_b;     /* cholesky.MLP.0 [C/2,C], .2 [6,C/2]                             */
} CartnetIcfParams;

typedef struct CartnetIcfModel {
  int32_t C, n_types, gemm_precision, reserved;
  float gamma_rbf, gamma_angle;          /* RBFExpansion gammas (models/utils.py:118-119)                              */
  float bn_eps, bn_momentum;
  const float *rbf_centers, *rbf_angle_centers;     /* [C] each                                                       */
  CartnetIcfParams p;
  CartnetIcfBn att_bn[4], att_bn_att[4], edge_bn, edge_bn_att;
} CartnetIcfModel;

size_t cartnet_icomformer_workspace_bytes(const CartnetIcfModel* model, int32_t N, int64_t E, int32_t Bg, int32_t M);
/* pred [M,3,3]; x_out [N,C] receives the final atom features (what the reference leaves in data.x). */
int cartnet_icomformer_forward(const CartnetIcfModel* model, const CartnetBatch* batch, const float* cell,
                               void* workspace, size_t workspace_bytes, int32_t training, float* pred, float* x_out,
                               int32_t* status, void* stream, void* aux_stream);
/* x_out: what forward wrote; grads: one destination per parameter (fresh values, not accumulated). */
int cartnet_icomformer_backward(const CartnetIcfModel* model, const CartnetBatch* batch, void* workspace,
                                size_t workspace_bytes, int32_t training, const float* dpred, const float* x_out,
                                const CartnetIcfParams* grads, void* stream, void* aux_stream);

/* Fused Adam step over a flat fp32 parameter buffer (torch.optim.Adam semantics, reference main.py:208):
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr * (m / (1-b1^t)) / (sqrt(v / (1-b2^t)) + eps).
 * grad_scale multiplies g first (1/world_size after the RCCL gradient all-reduce). */
int cartnet_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                      float beta1, float beta2, float eps, int32_t step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * PROTOTYPE (not on the model's path): the forward of ONE CartNet layer
 * (/root/reference/models/cartnet.py:204-274: propagate -> message -> aggregate -> update, training-mode BatchNorm,
 * envelope on) as ONE cooperative launch for the small batches of BASELINE configs[2] -- 256 workgroups, five phases,
 * four grid barriers (csrc/coop_layer.hip).  Precision 2 arithmetic (bf16 operands, fp32 accumulate and storage), D = 256,
 * E <= 16,384, every atom with at least one incoming edge; running statistics are not updated.
 * wn / w1e / w2: bf16 [4D][D] (gate_i | aggr_i | gate_j | aggr_j rows of MLP_*.0.weight), [2D][D] (its edge columns),
 * [2D][D] (MLP_gate.2 | MLP_aggr.2); b1 / b2 [2D]; Pn [N,4D], pre / gs [E,2D], e_out [E,D], aggr / x_out [N,D];
 * work: cartnet_coop_layer_workspace_floats(N, E) floats; bar: 3 x 8 x 32 zeroed words, epoch = launches on it so far;
 * status: one word, set to 1 if a barrier gave up (results are then invalid).
 * ---------------------------------------------------------------------------------------------------- */
size_t cartnet_coop_layer_workspace_floats(int32_t N, int32_t E);
int cartnet_coop_layer_fwd(const float* x, const float* e, const int32_t* tgt, const int32_t* src, const int32_t* rowptr,
                           const float* env, const void* wn_bf16, const void* w1e_bf16, const void* w2_bf16, const float* b1,
                           const float* b2, const float* bn1_w, const float* bn1_b, const float* bn2_w, const float* bn2_b,
                           int32_t N, int32_t E, float eps, float* Pn, float* pre, float* gs, float* e_out, float* aggr,
                           float* x_out, float* work, uint32_t* bar, uint32_t epoch, uint32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CARTNET_HIP_H */
