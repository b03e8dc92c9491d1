// Host side of the GEMM entry points + the small deterministic reduction kernels (split-K slabs, partial sums).
// The MFMA kernel itself is in gemm_kernel.h, instantiated per tile width in gemm_bn{256,128,64}.hip.
#include "gemm_x3.h"
#include <stdlib.h>
#include <vector>

namespace cn_gemm {
thread_local bool g_half_launched = false;
// Which DMA-fed fp32 kernel takes a prepacked activation x weight product: the 256-wide one (gemm_f32.h, two workgroups
// per CU), or the 128-wide one (gemm_f32w128.h, three per CU) for the launches with the node-term gather epilogue (layer
// GEMM 1) -- the one variant where a third resident workgroup pays: 402 vs 425 us sustained at the benchmark shape, the
// training step 15.40 vs 15.57 ms (same box, interleaved).  Every other variant is equal or slower on the narrow tile
// (SiLU on the A operand: -15 %, its prologue runs once per column tile).
bool use_f32nn128(const CartnetGemmArgs& a) {
  if (a.tile_policy == 128) return true;
  if (a.tile_policy == 256) return false;
  if (a.gather_i[0] != nullptr && !a.a_act) return true;
  // One 256-wide column tile per row tile, one group, one K-segment (iComformer's edge-sized C x C products): 1,384 row
  // tiles are 2.7 rounds of the 256-wide kernel's 512 resident workgroups but 3.6 of the 128-wide one's 768, whose third
  // workgroup per CU also keeps two main loops overlapping -- isolated 365 -> 346 us, the iComformer step 36.72 -> 36.52
  // ms (same-box A B A B, end of round 3).  CartNet's dE products (two folded K-segments) were faster alone too (370.6 ->
  // 362.5 us) but the training step, where they run next to the weight-gradient products, was not (14.86 -> 14.92 ms):
  // they stay on the wide kernel.  Grouped launches of that shape too when the caller asks for it per call
  // (CartnetGemmArgs.tile_policy = 1: the iComformer path, whose step gains another 1.8 %, 36.39 -> 35.74 ms; CartNet's
  // two-group layer products do not).
  if (a.N == 256 && (a.ngroups == 1 || a.tile_policy == 1) && a.nsegs == 1 && !a.a_act) return true;
  // few row tiles (atom-sized M): 128-wide tiles put twice as many workgroups on the chip (the folded dX product of the
  // node terms, M = 12,416, K = 1024: 97 tiles of 128 x 256 would use 97 of 256 CUs)
  const long long tiles = (long long)((a.M + 127) / 128) * (a.N / 256) * a.ngroups;
  return tiles < 200;
}
extern template bool launch_bn<256>(const CartnetGemmArgs&, const GemmFlags&, hipStream_t);
extern template bool launch_bn<128>(const CartnetGemmArgs&, const GemmFlags&, hipStream_t);
extern template bool launch_bn<64>(const CartnetGemmArgs&, const GemmFlags&, hipStream_t);
}

namespace {

struct ReduceJobs {
  const float* slabs[CARTNET_MAX_GROUPS];
  float* out[CARTNET_MAX_GROUPS];
};

__global__ void cn_splitk_reduce_kernel(const ReduceJobs jobs, int splitk, int M, int N, int ldo) {
  const float* __restrict__ slabs = jobs.slabs[blockIdx.y];
  float* __restrict__ out = jobs.out[blockIdx.y];
  const size_t total = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int s = 0; s < splitk; ++s) acc += slabs[(size_t)s * total + i];
    const size_t m = i / N, n = i % N;
    out[m * ldo + n] = acc;
  }
}

// N % 4 == 0, 16-byte aligned slabs / rows: 64 float4 elements per block, the slabs dealt round-robin to four
// 64-thread groups (eight independent 16-byte loads in flight per thread), group sums added in group order.
__global__ __launch_bounds__(256) void cn_splitk_reduce4_kernel(const ReduceJobs jobs, int splitk, int M, int N,
                                                                int ldo) {
  __shared__ f32x4 red[4][64];
  const float* __restrict__ slabs = jobs.slabs[blockIdx.y];
  float* __restrict__ out = jobs.out[blockIdx.y];
  const size_t total4 = (size_t)M * N / 4;
  const int sg = threadIdx.x >> 6, el = threadIdx.x & 63;
  const size_t i = (size_t)blockIdx.x * 64 + el;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (i < total4) {
    const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(slabs) + i;
    int s = sg;
    for (; s + 28 < splitk; s += 32) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(s + 4 * u) * total4];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < splitk; s += 4) acc += p[(size_t)s * total4];
  }
  red[sg][el] = acc;
  __syncthreads();
  if (sg == 0 && i < total4) {
    f32x4 t = red[0][el];
    t += red[1][el];
    t += red[2][el];
    t += red[3][el];
    const size_t e0 = i * 4, m = e0 / N, n = e0 % N;
    *reinterpret_cast<f32x4*>(out + m * ldo + n) = t;
  }
}

struct FinalizeJobs {
  const double* parts[8];
  float* out[8];
  float* out2[8];     // optional second destination of the same sums (a gradient tensor next to a scratch copy)
};

__global__ __launch_bounds__(1024) void cn_colsum_finalize_kernel(const FinalizeJobs jobs, int nparts, int N) {
  __shared__ double red[64 * CN_SUM_COLS];
  const double* __restrict__ parts = jobs.parts[blockIdx.y];
  float* __restrict__ out = jobs.out[blockIdx.y];
  const double tot = cn_block_colsum(parts, nparts, N, blockIdx.x * CN_SUM_COLS, red);
  const int c = blockIdx.x * CN_SUM_COLS + threadIdx.x;
  if (threadIdx.x < CN_SUM_COLS && c < N) {
    out[c] = (float)tot;
    if (jobs.out2[blockIdx.y]) jobs.out2[blockIdx.y][c] = (float)tot;
  }
}

// fp32 partial rows -> fp32 column sums, accumulated in fp64 in a fixed order: 32 columns x 32 row groups per block
// (a thread walks rows rg, rg + 32, ... with four independent loads in flight; the 32 partial sums of a column are
// added in row-group order).  The one-thread-per-column form took 260 us for the head's 256 x 904 partial matrix
// (a dependent load-add chain per thread, eight blocks on the whole chip) and sat alone on the weight-gradient stream.
__global__ __launch_bounds__(1024) void cn_colsum_finalize_f32_kernel(const float* __restrict__ parts, int nparts, int N,
                                                                      float* __restrict__ out) {
  __shared__ double red[32][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + cl;
  double acc = 0.0;
  if (n < N) {
    int q = rg;
    for (; q + 96 < nparts; q += 128) {
      const float v0 = parts[(size_t)q * N + n], v1 = parts[(size_t)(q + 32) * N + n],
                  v2 = parts[(size_t)(q + 64) * N + n], v3 = parts[(size_t)(q + 96) * N + n];
      acc += (double)v0;
      acc += (double)v1;
      acc += (double)v2;
      acc += (double)v3;
    }
    for (; q < nparts; q += 32) acc += (double)parts[(size_t)q * N + n];
  }
  red[rg][cl] = acc;
  __syncthreads();
  if (rg == 0 && n < N) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += red[r][cl];
    out[n] = (float)t;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// out[g][r, c] = silu(in[g][r, c]) with the GEMM kernels' own SiLU: CartnetGemmArgs.a_act_out for the launches whose
// kernel does not write it in passing
struct ActJobs {
  const float* in[CARTNET_MAX_GROUPS];
  float* out[CARTNET_MAX_GROUPS];
};
__global__ void cn_act_rows_kernel(const ActJobs jobs, int M, int K, int ld) {
  const float* __restrict__ in = jobs.in[blockIdx.y];
  float* __restrict__ out = jobs.out[blockIdx.y];
  const size_t total = (size_t)M * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / K, c = i % K;
    out[r * ld + c] = cn_gemm::fast_silu(in[r * ld + c]);
  }
}

}  // namespace

// every row any epilogue operand touches is 16-byte aligned -> vector epilogue (GemmFlags.wide)
static bool epilogue_rows_aligned(const CartnetGemmArgs& a) {
  bool w = (a.ldc % 4 == 0) && (a.N % 4 == 0);
  for (int gI = 0; gI < a.ngroups && w; ++gI) {
    w = w && aligned16(a.C[gI]) && (!a.bias[gI] || aligned16(a.bias[gI])) && (!a.cpre[gI] || aligned16(a.cpre[gI]));
    if (a.gather_i[gI]) w = w && aligned16(a.gather_i[gI]) && aligned16(a.gather_j[gI]) && (a.ldg % 4 == 0);
    if (a.resid[gI]) w = w && aligned16(a.resid[gI]) && (a.ldr % 4 == 0);
    if (a.dact[gI]) w = w && aligned16(a.dact[gI]) && (a.ldd % 4 == 0);
  }
  return w;
}

// K-segments that are adjacent column blocks of one matrix, with the image of the concatenated weight operand at hand:
// one product over the concatenated K on the DMA-fed kernels.
static bool segments_fold(const CartnetGemmArgs& a) {
  if (!(a.nsegs > 1 && a.b_split_folded && a.N == cn_gemm::X3_BN && !a.a_kstrided && a.b_kstrided && a.splitk == 1 &&
        a.ngroups == 1 && a.M > 0))
    return false;
  bool fold = epilogue_rows_aligned(a) && a.lda % 4 == 0 && a.ldb % 4 == 0 && a.K % cn_gemm::BK == 0 &&
              (long long)a.K * a.nsegs <= a.lda && (double)a.M * a.lda * 4.0 < 4294967296.0;
  for (int s = 0; s < a.nsegs && fold; ++s)
    fold = a.A[s] && a.B[s] && aligned16(a.A[s]) && aligned16(a.B[s]) &&
           reinterpret_cast<const char*>(a.A[s]) ==
               reinterpret_cast<const char*>(a.A[0]) + (size_t)s * a.K * (a.a_half ? 2 : 4);
  return fold;
}

// precision 0, an activation x weight product whose weight image is at hand and whose shape the DMA-fed fp32 kernels take
static bool f32_image_path(const CartnetGemmArgs& a) {
  if (a.precision != 0 || a.a_kstrided || !a.b_kstrided || a.splitk != 1 || a.M <= 0 || a.K <= 0 || a.N % 256 != 0)
    return false;
  if (a.nsegs > 1) return segments_fold(a);
  bool ok = epilogue_rows_aligned(a) && a.lda % 4 == 0 && a.ldb % 4 == 0 && a.K % cn_gemm::BK == 0 &&
            (double)a.M * a.lda * 4.0 < 4294967296.0;
  for (int gI = 0; gI < a.ngroups && ok; ++gI)
    ok = a.b_split[gI] && a.A[gI] && a.B[gI] && aligned16(a.A[gI]) && aligned16(a.B[gI]);
  return ok;
}

static int choose_bn(const CartnetGemmArgs& a);

// CartnetGemmArgs.gst_*: the launch reaches a kernel that carries the gate statistics epilogue -- precision 0:
// cn_gemm_f32p_kernel<false, false, 530, 32> (edge-sized launches) or cn_gemm_f32nn128_kernel<false, 1 / 2>; precision 1 (bf16x3): cn_gemm_x3nn16_kernel<false, 1 / 2>.  (The same predicates the
// dispatch below and launch_variant apply, in their order; the two DMA-fed families take the same shapes.)
static bool gate_stats_launch_ok(const CartnetGemmArgs& a) {
  if (!(a.gst_g && a.gst_mean_rstd && a.gst_gamma && a.gst_beta && a.gst_ld >= a.N && a.gst_ld % 4 == 0)) return false;
  if (!(aligned16(a.gst_g) && aligned16(a.gst_mean_rstd) && aligned16(a.gst_gamma) && aligned16(a.gst_beta))) return false;
  if (!((a.precision == 0 || a.precision == 1) && a.ngroups == 1 && !a.a_act && !a.b_act && !a.out_act && !a.dact[0] &&
        !a.gather_i[0] && !a.cpre[0] && !a.bias[0] && a.colsum[0] && a.colsq[0] && !a.a_act_out[0] && a.N == 256))
    return false;
  CartnetGemmArgs q = a;
  q.precision = 0;                       // (the image-path conditions do not depend on the operand format)
  if (!f32_image_path(q) || choose_bn(a) != 256 || cn_gemm::any_half(a)) return false;
  CartnetGemmArgs f = a;
  if (segments_fold(a)) { f.K *= f.nsegs; f.nsegs = 1; }
#ifdef CN_EXPERIMENTAL_Q
  if (a.precision == 0 && cn_gemm::use_f32nnq(f)) return false;     // (launch_variant asks the quad kernel first: no GST code there)
#endif
  // (launch_variant asks the persistent kernel first: it has the form -- gemm_f32p.h, KIND 530)
  return f.nsegs == 1 && (a.precision == 1 || cn_gemm::use_f32p(f) || cn_gemm::use_f32nn128(f));
}

extern "C" int cartnet_gemm_gate_stats_ok(const CartnetGemmArgs* args) { return args && gate_stats_launch_ok(*args) ? 1 : 0; }

// Column-tile width of a launch (see the comment in cartnet_gemm_impl): 256 / 128 / 64 by N, narrower for launches
// with few tiles.  Shared with the launch timer so that its variant names the kernel family that really runs.
static int choose_bn(const CartnetGemmArgs& a) {
  int bn = a.N > 128 ? 256 : (a.N > 64 ? 128 : 64);
  // (transposed-A launches: only the K-segment form -- iComformer's 256 x 256 x 512 chain-rule product dWe ran as TWO
  //  workgroups of the 256-wide general kernel, 110-150 us on the weight-gradient stream, five times per step; weight
  //  gradients proper carry split-K or take the DMA-fed kernel and are left alone)
  if (bn == 256 && a.precision <= 1 && (!a.a_kstrided || a.nsegs > 1) && a.splitk == 1 && a.N % 64 == 0) {
    const long long tiles = (long long)((a.M + 127) / 128) * ((a.N + 255) / 256) * a.ngroups;
    // bf16x3 tiles are ~2x shorter: switch later (96: the benchmark batch's 97-row-tile dX product, K = 1024, stays on
    // the DMA-fed bf16x3 kernel -- +0.4 % on the step in a same-box A/B)
    const long long few = a.precision == 0 ? 200 : 96;
    // fp32 with a weight image: from 64 tiles up the 128-wide DMA-fed kernel (use_f32nn128: 2 x tiles workgroups) beats the
    // register-staged narrow kernels -- the folded dX product of the node terms 85 -> 68 us sustained
    if (tiles < few && !(tiles >= 64 && f32_image_path(a))) bn = (2 * tiles >= few) ? 128 : 64;
  }
  return bn;
}

// ---- opt-in launch timing (bench.py): HIP events on the launch stream around every cartnet_gemm call ------------
namespace {
struct GemmRecord {
  int variant;       // bit0 a_kstrided, bit1 b_kstrided, bit2 a_act, bit3 b_act, bits 4.. tile width / 64
  double flops;
  hipEvent_t e0, e1;
};
bool g_prof_on = false;
int g_prof_only = -1;     // >= 0: only launches of this variant are timed
int g_prof_every = 1;     // of the launches that qualify, every n-th is timed (cartnet_profile_gemm_every)
long long g_prof_seen = 0;
std::vector<GemmRecord> g_prof;
}  // namespace

static int cartnet_gemm_impl(const CartnetGemmArgs* args, void* stream);

extern "C" int cartnet_profile_gemm(int32_t enable) {
  if (enable && !g_prof_on) {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
  }
  g_prof_on = enable != 0;
  return 0;
}

extern "C" int cartnet_profile_gemm_only(int32_t variant) {
  g_prof_only = variant;
  return 0;
}

extern "C" int cartnet_profile_gemm_every(int32_t n) {
  CN_CHECK(n >= 1, "cartnet_profile_gemm_every: n=%d must be >= 1", n);
  g_prof_every = n;
  g_prof_seen = 0;
  return 0;
}

extern "C" int cartnet_profile_gemm_read(CartnetGemmProfile* out, int32_t max_entries) {
  CN_CHECK(out && max_entries > 0, "cartnet_profile_gemm_read: bad arguments");
  int n = 0;
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.e1) != hipSuccess) { cartnet_set_error("cartnet_profile_gemm_read: event sync failed"); return -1; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    int slot = -1;
    for (int i = 0; i < n; ++i) if (out[i].variant == r.variant) slot = i;
    if (slot < 0) {
      if (n == max_entries) continue;
      slot = n++;
      out[slot].variant = r.variant;
      out[slot].launches = 0;
      out[slot].flops = 0.0;
      out[slot].ms = 0.0;
    }
    out[slot].launches += 1;
    out[slot].flops += r.flops;
    out[slot].ms += ms;
  }
  return n;
}

extern "C" int cartnet_gemm(const CartnetGemmArgs* args, void* stream) {
  if (!g_prof_on || !args) return cartnet_gemm_impl(args, stream);
  GemmRecord r;
  int bn = choose_bn(*args) / 64;
  // the variant names the kernel that runs: the 128-wide DMA-fed kernel reports tile width 128, the kernels that also
  // write silu(A) set bit 9
  const bool actout = args->a_act_out[0] != nullptr;
  if (bn == 4 && !actout && f32_image_path(*args) && cn_gemm::use_f32nn128(*args) &&
      !(args->nsegs == 1 && cn_gemm::use_f32p(*args)))
    bn = 2;
  // bit 8: the streamed dimension (rows of an activation x weight product, reduction length of a weight gradient) is
  // edge-sized; bits 10..: the other inner dimension / 16 (K of an NN product, M of a weight gradient), capped
  const long long streamed = args->a_kstrided ? args->K : args->M;
  const int inner = args->a_kstrided ? args->M : args->K;
  // bit 18: the persistent kernel takes the launch (gemm_f32p.hip; the same predicates launch_variant applies)
  const bool persistent = bn == 4 && args->nsegs == 1 && args->splitk == 1 && epilogue_rows_aligned(*args) &&
                          f32_image_path(*args) && cn_gemm::use_f32p(*args);
  r.variant = (args->a_kstrided ? 1 : 0) | (args->b_kstrided ? 2 : 0) | (args->a_act ? 4 : 0) | (args->b_act ? 8 : 0) |
              (bn << 4) | (streamed >= 32768 ? 256 : 0) | (actout ? 512 : 0) | ((inner > 4080 ? 255 : inner / 16) << 10) |
              (persistent ? (1 << 18) : 0);
  if (g_prof_only >= 0 && r.variant != g_prof_only) return cartnet_gemm_impl(args, stream);
  if (g_prof_every > 1 && (g_prof_seen++ % g_prof_every) != 0) return cartnet_gemm_impl(args, stream);
  const int nptr = args->ngroups > 1 ? args->ngroups : args->nsegs;
  r.flops = 2.0 * args->M * args->N * (double)args->K * nptr;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return cartnet_gemm_impl(args, stream);
  (void)hipEventRecord(r.e0, st);
  const int rc = cartnet_gemm_impl(args, stream);
  (void)hipEventRecord(r.e1, st);
  g_prof.push_back(r);
  return rc;
}

static int cartnet_gemm_impl(const CartnetGemmArgs* args, void* stream) {
  CN_CHECK(args != nullptr, "cartnet_gemm: null args");
  CartnetGemmArgs a = *args;
  CN_CHECK(a.M >= 0 && a.N >= 0 && a.K >= 0, "cartnet_gemm: negative shape M=%d N=%d K=%d", a.M, a.N, a.K);
  CN_CHECK(a.ngroups >= 1 && a.ngroups <= CARTNET_MAX_GROUPS, "cartnet_gemm: ngroups=%d out of range", a.ngroups);
  CN_CHECK(a.nsegs >= 1 && a.nsegs <= CARTNET_MAX_GROUPS, "cartnet_gemm: nsegs=%d out of range", a.nsegs);
  CN_CHECK(!(a.ngroups > 1 && a.nsegs > 1), "cartnet_gemm: groups and K-segments are mutually exclusive");
  CN_CHECK(a.splitk >= 1, "cartnet_gemm: splitk=%d", a.splitk);
  CN_CHECK(a.splitk == 1 || a.K >= 2 * cn_gemm::BK, "cartnet_gemm: split-K needs K >= %d", 2 * cn_gemm::BK);
  if (a.M == 0 || a.N == 0) return 0;
  const int nptr = a.ngroups > 1 ? a.ngroups : a.nsegs;
  bool vecA = (a.lda % 4 == 0), vecB = (a.ldb % 4 == 0);
  for (int i = 0; i < nptr; ++i) {
    CN_CHECK(a.A[i] && a.B[i], "cartnet_gemm: null operand %d", i);
    vecA = vecA && aligned16(a.A[i]);
    vecB = vecB && aligned16(a.B[i]);
  }
  for (int gI = 0; gI < a.ngroups; ++gI) {
    CN_CHECK(a.C[gI] != nullptr, "cartnet_gemm: null output %d", gI);
    CN_CHECK((a.gather_i[gI] == nullptr) == (a.gather_j[gI] == nullptr), "cartnet_gemm: gather_i/gather_j must pair");
    if (a.gather_i[gI]) CN_CHECK(a.tgt && a.src, "cartnet_gemm: gather epilogue needs tgt/src");
    if (a.colsq[gI]) CN_CHECK(a.colsum[gI] != nullptr, "cartnet_gemm: colsq needs colsum");
  }
  CN_CHECK(a.lda >= (a.a_kstrided ? a.M : a.K) || a.K == 0, "cartnet_gemm: lda=%d too small", a.lda);
  CN_CHECK(a.ldb >= (a.b_kstrided ? a.N : a.K) || a.K == 0, "cartnet_gemm: ldb=%d too small", a.ldb);
  CN_CHECK(a.ldc >= a.N, "cartnet_gemm: ldc=%d < N=%d", a.ldc, a.N);
  if (a.splitk > 1) {
    CN_CHECK(a.nsegs == 1, "cartnet_gemm: split-K with K-segments is not supported");
    for (int gI = 0; gI < a.ngroups; ++gI)
      CN_CHECK(!a.bias[gI] && !a.gather_i[gI] && !a.resid[gI] && !a.dact[gI] && !a.cpre[gI] && !a.colsum[gI] &&
                   !a.out_act,
               "cartnet_gemm: split-K writes raw partial slabs, no epilogue allowed");
  }
  CN_CHECK(!a.gst_g || gate_stats_launch_ok(a),
           "cartnet_gemm: gst_g is set but this launch does not reach the kernel with the gate-statistics epilogue "
           "(precision 0 / 1, N = 256, weight image, one group, colsum + colsq (+ resid) and nothing else, >= 64 / 96 row tiles: "
           "ask cartnet_gemm_gate_stats_ok first)");
  cn_gemm::GemmFlags fl;
  fl.tile_m0 = 0;
  fl.split0 = 0;
  fl.k_lo = 0;
  fl.k_hi = a.K;
  fl.wide = epilogue_rows_aligned(a) ? 1 : 0;
  CN_CHECK(a.precision >= 0 && a.precision <= 2, "cartnet_gemm: precision=%d (0 = fp32 MFMA, 1 = bf16x3 split, 2 = bf16)",
           a.precision);
  CN_CHECK(a.dact_kind == 0 || (a.dact_kind == 1 && !cn_gemm::any_half(a)),
           "cartnet_gemm: dact_kind=%d (0 = silu', 1 = sigmoid / softplus'; 1 not with bf16 storage)", a.dact_kind);
  CN_CHECK(a.tile_policy == 0 || a.tile_policy == 1 || a.tile_policy == 3 || a.tile_policy == 128 || a.tile_policy == 256,
           "cartnet_gemm: tile_policy=%d (0 = automatic, 1 = narrow tiles for grouped N = 256 products too, 3 = the persistent "
           "kernel wherever it has the form, 128 / 256 = force)",
           a.tile_policy);
  // Few row tiles (atom-sized M, small batches): 128 x 256 tiles would leave most of the 256 CUs idle and the launch
  // would last one tile's latency (16+ K-steps of a full tile); narrower column tiles (the general kernel's 128- and
  // 64-wide forms, exact fp32) spread the same work over 2-4x as many workgroups.  Not at precision 2 (the bf16 kernels
  // exist for 256-wide tiles only and their tiles are 6x shorter to begin with: measured a loss), and later at
  // precision 1.
  int bn = choose_bn(a);
  if (bn == 256 && segments_fold(a)) {
    // K-segments that are adjacent column blocks of one matrix: one product over the concatenated K.  Folded only when
    // the DMA-fed kernel is certain to take the launch (B[0] alone does not describe the folded operand).
    a.K *= a.nsegs;
    a.nsegs = 1;
    a.b_split[0] = a.b_split_folded;
  }
  fl.x3 = a.precision;
  fl.vecA = vecA ? 1 : 0;
  fl.vecB = vecB ? 1 : 0;
  int kchunk = (a.K + a.splitk - 1) / a.splitk;
  kchunk = ((kchunk + cn_gemm::BK - 1) / cn_gemm::BK) * cn_gemm::BK;
  if (kchunk == 0) kchunk = cn_gemm::BK;
  fl.kchunk = kchunk;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.a_act_out[0]) {
    CN_CHECK(a.a_act && !a.a_kstrided && a.nsegs == 1, "cartnet_gemm: a_act_out needs a_act = 1, a k-contiguous A and no K-segments");
    for (int gI = 0; gI < a.ngroups; ++gI) CN_CHECK(a.a_act_out[gI] != nullptr, "cartnet_gemm: a_act_out[%d] missing", gI);
    // the DMA-fed 256-wide kernels write the activated operand on its way into LDS (gemm_f32ao.h, gemm_x3ao.h); any
    // other launch (narrow tiles of a small batch, no weight image) gets it from an elementwise pass first
    bool fused = bn == 256 && !a.b_act && a.b_kstrided && a.splitk == 1 && fl.wide && vecA && vecB &&
                 a.N % 256 == 0 && a.K > 0 && a.K % cn_gemm::BK == 0 && (double)a.M * a.lda * 4.0 < 4294967296.0;
    for (int gI = 0; gI < a.ngroups; ++gI) fused = fused && a.b_split[gI] && aligned16(a.a_act_out[gI]);
    if (!fused) {
      ActJobs jobs;
      for (int gI = 0; gI < CARTNET_MAX_GROUPS; ++gI) {
        jobs.in[gI] = gI < a.ngroups ? a.A[gI] : nullptr;
        jobs.out[gI] = gI < a.ngroups ? a.a_act_out[gI] : nullptr;
        a.a_act_out[gI] = nullptr;
      }
      const long long total = (long long)a.M * a.K;
      if (total > 0) {
        const int blocks = (int)(total / 256 + 1 > 16384 ? 16384 : total / 256 + 1);
        hipLaunchKernelGGL(cn_act_rows_kernel, dim3(blocks, a.ngroups), dim3(256), 0, st, jobs, a.M, a.K, a.lda);
      }
    }
  }
  const bool half = cn_gemm::any_half(a);
  if (half) {
    CN_CHECK(a.precision == 2 && bn == 256 && a.K > 0,
             "cartnet_gemm: a_half / b_half / c_half / dact_half need precision 2 and a 256-wide launch");
    CN_CHECK(!a.a_act_out[0] && !(a.a_kstrided && (a.c_half || a.dact_half)) && !(!a.a_kstrided && a.b_half),
             "cartnet_gemm: half storage: a_act_out, a bf16 weight-gradient output and a bf16 weight operand are not supported");
    cn_gemm::g_half_launched = false;
  }
  bool ok;
  if (bn == 256) ok = cn_gemm::launch_bn<256>(a, fl, st);
  else if (bn == 128) ok = cn_gemm::launch_bn<128>(a, fl, st);
  else ok = cn_gemm::launch_bn<64>(a, fl, st);
  CN_CHECK(ok, "cartnet_gemm: unsupported layout/activation combination (a_ks=%d b_ks=%d a_act=%d b_act=%d)",
           a.a_kstrided, a.b_kstrided, a.a_act, a.b_act);
  CN_CHECK(!half || cn_gemm::g_half_launched,
           "cartnet_gemm: no half-storage kernel for this launch (needs the pre-split weight image of an activation x "
           "weight product, or a weight gradient with M %% 4 == 0; 16-byte aligned rows; a compiled operand combination: "
           "csrc/gemm_h.hip)");
  CN_LAUNCH_CHECK("cartnet_gemm");
  return 0;
}

extern "C" int cartnet_splitk_reduce(const float* const* slabs, float* const* outs, int32_t njobs, int32_t splitk,
                                     int32_t M, int32_t N, int32_t ldo, void* stream) {
  CN_CHECK(slabs && outs && njobs >= 1 && njobs <= CARTNET_MAX_GROUPS, "cartnet_splitk_reduce: njobs=%d out of range",
           njobs);
  CN_CHECK(splitk >= 1 && M >= 0 && N >= 0 && ldo >= N, "cartnet_splitk_reduce: bad shape");
  if (M == 0 || N == 0) return 0;
  ReduceJobs jobs;
  for (int j = 0; j < CARTNET_MAX_GROUPS; ++j) {
    jobs.slabs[j] = j < njobs ? slabs[j] : nullptr;
    jobs.out[j] = j < njobs ? outs[j] : nullptr;
    if (j < njobs) CN_CHECK(slabs[j] && outs[j], "cartnet_splitk_reduce: null pointer in job %d", j);
  }
  const size_t total = (size_t)M * N;
  bool vec4 = (N % 4 == 0) && (ldo % 4 == 0);
  for (int j = 0; j < njobs; ++j) vec4 = vec4 && aligned16(slabs[j]) && aligned16(outs[j]);
  if (vec4) {
    hipLaunchKernelGGL(cn_splitk_reduce4_kernel, dim3((unsigned)((total / 4 + 63) / 64), njobs), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), jobs, splitk, M, N, ldo);
    CN_LAUNCH_CHECK("cartnet_splitk_reduce");
    return 0;
  }
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cn_splitk_reduce_kernel, dim3(blocks, njobs), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     jobs, splitk, M, N, ldo);
  CN_LAUNCH_CHECK("cartnet_splitk_reduce");
  return 0;
}

extern "C" int cartnet_colsum_finalize(double* const* parts, float* const* outs, int32_t njobs, int32_t nparts,
                                       int32_t N, void* stream) {
  return cartnet_colsum_finalize2(parts, outs, nullptr, njobs, nparts, N, stream);
}

extern "C" int cartnet_colsum_finalize2(double* const* parts, float* const* outs, float* const* outs2, int32_t njobs,
                                        int32_t nparts, int32_t N, void* stream) {
  CN_CHECK(parts && outs && njobs >= 1 && njobs <= 8, "cartnet_colsum_finalize: njobs=%d out of range (1..8)", njobs);
  CN_CHECK(nparts >= 0 && N >= 0, "cartnet_colsum_finalize: bad shape");
  if (N == 0) return 0;
  FinalizeJobs jobs;
  for (int j = 0; j < 8; ++j) {
    jobs.parts[j] = j < njobs ? parts[j] : nullptr;
    jobs.out[j] = j < njobs ? outs[j] : nullptr;
    jobs.out2[j] = (j < njobs && outs2) ? outs2[j] : nullptr;
    if (j < njobs) CN_CHECK(parts[j] && outs[j], "cartnet_colsum_finalize: null pointer in job %d", j);
  }
  hipLaunchKernelGGL(cn_colsum_finalize_kernel, dim3(cn_ceil_div(N, CN_SUM_COLS), njobs), dim3(1024), 0,
                     reinterpret_cast<hipStream_t>(stream), jobs, nparts, N);
  CN_LAUNCH_CHECK("cartnet_colsum_finalize");
  return 0;
}

extern "C" int cartnet_colsum_finalize_f32(const float* parts, int32_t nparts, int32_t N, float* out, void* stream) {
  CN_CHECK(parts && out, "cartnet_colsum_finalize_f32: null pointer");
  CN_CHECK(nparts >= 0 && N >= 0, "cartnet_colsum_finalize_f32: bad shape");
  if (N == 0) return 0;
  hipLaunchKernelGGL(cn_colsum_finalize_f32_kernel, dim3(cn_ceil_div(N, 32)), dim3(1024), 0,
                     reinterpret_cast<hipStream_t>(stream), parts, nparts, N, out);
  CN_LAUNCH_CHECK("cartnet_colsum_finalize_f32");
  return 0;
}
