// Once-per-batch graph layout: int32 copies of edge_index, CSR row pointer by target (edge_index[1] is already
// sorted ascending, reference dataset/utils.py:235) and a stable CSC permutation by source for the backward
// by-source reduction.  Everything is deterministic: no ordering depends on atomics arrival.
#include "common.h"

namespace {

constexpr int MAX_GRAPH_NODES = 8192;  // LDS histogram capacity per crystal (32 KiB)

__global__ void cn_edge_convert_kernel(const int64_t* __restrict__ ei, long long E, int N,
                                       int* __restrict__ src32, int* __restrict__ tgt32,
                                       int* __restrict__ rowptr, int* __restrict__ status) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k <= E; k += stride) {
    long long tcur = N;  // sentinel for k == E
    if (k < E) {
      const long long s = ei[k];
      tcur = ei[E + k];
      int bad = 0;
      if (s < 0 || s >= N || tcur < 0 || tcur >= N) bad |= 2;
      // the int32 copies are what every later kernel gathers through: clamped, so that a malformed batch gives wrong
      // numbers and a status bit, never an out-of-bounds access
      src32[k] = (int)min(max(s, 0LL), (long long)max(N - 1, 0));
      tgt32[k] = (int)min(max(tcur, 0LL), (long long)max(N - 1, 0));
      if (bad) atomicOr(status, bad);
    }
    long long tprev = -1;
    if (k > 0) tprev = ei[E + k - 1];
    if (k < E && tprev > tcur) atomicOr(status, 1);  // not sorted by target
    // rowptr[t] = first edge whose target is >= t
    long long lo = tprev + 1, hi = (k < E) ? tcur : (long long)N;
    if (lo < 0) lo = 0;
    if (hi > N) hi = N;
    for (long long t = lo; t <= hi; ++t) rowptr[t] = (int)k;
  }
}

// One workgroup per crystal: counting sort of its edges by source, stable in edge order.
__global__ __launch_bounds__(256) void cn_csc_build_kernel(const int* __restrict__ src32,
                                                           const int* __restrict__ rowptr,
                                                           const int64_t* __restrict__ graph_ptr, int Bg, int N,
                                                           int E, int* __restrict__ colptr, int* __restrict__ perm,
                                                           int* __restrict__ status) {
  __shared__ int cnt[MAX_GRAPH_NODES];
  __shared__ int part[256];
  __shared__ int skey[256];
  const int g = blockIdx.x, tid = threadIdx.x;
  int n0 = 0, n1 = N;
  if (graph_ptr) {
    n0 = (int)graph_ptr[g];
    n1 = (int)graph_ptr[g + 1];
  }
  if (g == Bg - 1 && tid == 0) colptr[N] = E;
  const int n = n1 - n0;
  if (n <= 0) return;
  if (n > MAX_GRAPH_NODES) {
    if (tid == 0) atomicOr(status, 8);
    return;
  }
  const int e0 = rowptr[n0], e1 = rowptr[n1];
  for (int i = tid; i < n; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int k = e0 + tid; k < e1; k += 256) {
    const int j = src32[k] - n0;
    if (j < 0 || j >= n) atomicOr(status, 4);  // an edge leaves its crystal
    else atomicAdd(&cnt[j], 1);
  }
  __syncthreads();
  // exclusive scan of cnt[0..n): each thread owns a contiguous chunk
  const int per = (n + 255) / 256;
  const int c0 = min(n, tid * per), c1 = min(n, c0 + per);
  int sum = 0;
  for (int i = c0; i < c1; ++i) sum += cnt[i];
  part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) {
      const int v = part[i];
      part[i] = run;
      run += v;
    }
  }
  __syncthreads();
  int run = part[tid];
  for (int i = c0; i < c1; ++i) {
    const int v = cnt[i];
    cnt[i] = run;                 // becomes the running cursor of source i
    colptr[n0 + i] = e0 + run;
    run += v;
  }
  __syncthreads();
  // stable placement, 256 edges at a time in edge order
  for (int base = e0; base < e1; base += 256) {
    const int k = base + tid;
    int key = -1;
    if (k < e1) {
      key = src32[k] - n0;
      if (key < 0 || key >= n) key = -1;
    }
    skey[tid] = key;
    __syncthreads();
    int cur = 0, rank = 0;
    if (key >= 0) {
      cur = cnt[key];
      for (int t = 0; t < tid; ++t) rank += (skey[t] == key) ? 1 : 0;
    }
    __syncthreads();
    if (key >= 0) {
      perm[e0 + cur + rank] = k;
      atomicAdd(&cnt[key], 1);
    }
    __syncthreads();
  }
}

// Stable counting sort of N items by an int64 key in [0, nkeys), one workgroup of 16 waves (atom types: nkeys = 119).
// Wave w owns the contiguous item range [w * per, (w + 1) * per): pass 1 counts its keys into cnt[w][.]; the counts
// become offsets (exclusive prefix over the waves inside a key, exclusive prefix over the keys); pass 2 walks the range
// again in chunks of 64 and ranks the lanes of a chunk key by key with wave ballots (a crystal has a handful of
// elements, so a chunk needs a handful of rounds), so equal keys keep their input order.  The first version ranked
// every item with a serial scan over the preceding items of its 1024-chunk: 420 us for 12,416 atoms, alone on the
// weight-gradient stream.  Dynamic LDS: (SORT_WAVES + 1) * nkeys ints.
constexpr int SORT_WAVES = 16;
constexpr int SORT_BATCH = 8;      // 64-item chunks whose key loads are in flight together

__device__ __forceinline__ int cn_sort_key(const int64_t* __restrict__ keys, int i, int nkeys, int* status) {
  long long k = keys[i];
  if (k < 0 || k >= nkeys) {     // clamped like the forward gather (cn_node_embed_kernel), so perm stays a full permutation
    if (status) atomicOr(status, 16);
    k = k < 0 ? 0 : nkeys - 1;
  }
  return (int)k;
}

__global__ __launch_bounds__(64 * SORT_WAVES) void cn_sort_by_key_kernel(const int64_t* __restrict__ keys, int N, int nkeys,
                                                                         int* __restrict__ perm, int* __restrict__ ptr,
                                                                         int* __restrict__ status) {
  extern __shared__ int sort_lds[];
  int* cnt = sort_lds;                          // [SORT_WAVES][nkeys]
  int* base = sort_lds + SORT_WAVES * nkeys;    // [nkeys]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = ((N + SORT_WAVES - 1) / SORT_WAVES + 63) / 64 * 64;
  const int i0 = min(N, w * per), i1 = min(N, i0 + per);
  for (int i = tid; i < SORT_WAVES * nkeys; i += 64 * SORT_WAVES) cnt[i] = 0;
  __syncthreads();
  // (key loads in batches of SORT_BATCH chunks: the loads of a batch are independent, so a wave pays the memory latency
  //  once per batch instead of once per 64 items -- under a co-running GEMM that latency is microseconds)
  for (int c0 = i0; c0 < i1; c0 += 64 * SORT_BATCH) {
    int key[SORT_BATCH];
#pragma unroll
    for (int u = 0; u < SORT_BATCH; ++u) {
      const int i = c0 + 64 * u + lane;
      key[u] = i < i1 ? cn_sort_key(keys, i, nkeys, status) : -1;
    }
#pragma unroll
    for (int u = 0; u < SORT_BATCH; ++u)
      if (key[u] >= 0) atomicAdd(&cnt[w * nkeys + key[u]], 1);
  }
  __syncthreads();
  for (int k = tid; k < nkeys; k += 64 * SORT_WAVES) {     // per key: counts -> offsets of the waves inside the key
    int run = 0;
    for (int v = 0; v < SORT_WAVES; ++v) {
      const int c = cnt[v * nkeys + k];
      cnt[v * nkeys + k] = run;
      run += c;
    }
    base[k] = run;                                          // total of the key, turned into its start below
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 0; k < nkeys; ++k) {
      const int v = base[k];
      base[k] = run;
      ptr[k] = run;
      run += v;
    }
    ptr[nkeys] = run;
  }
  __syncthreads();
  volatile int* vcnt = cnt;     // lane `leader` writes what all lanes of the wave read in the next round
  for (int b0 = i0; b0 < i1; b0 += 64 * SORT_BATCH) {
    int keyb[SORT_BATCH];
#pragma unroll
    for (int u = 0; u < SORT_BATCH; ++u) {
      const int i = b0 + 64 * u + lane;
      keyb[u] = i < i1 ? cn_sort_key(keys, i, nkeys, nullptr) : -1;
    }
#pragma unroll
    for (int u = 0; u < SORT_BATCH; ++u) {
      const int i = b0 + 64 * u + lane;
      const int key = keyb[u];
      const bool valid = key >= 0;
      unsigned long long todo = __ballot(valid);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k0 = __shfl(key, leader);
        const unsigned long long same = __ballot(valid && key == k0);
        const int off = vcnt[w * nkeys + k0];
        if (valid && key == k0) perm[base[k0] + off + __popcll(same & ((1ull << lane) - 1ull))] = i;
        __builtin_amdgcn_wave_barrier();
        if (lane == leader) vcnt[w * nkeys + k0] = off + __popcll(same);
        __builtin_amdgcn_wave_barrier();
        todo &= ~same;
      }
    }
  }
}

struct ZeroJobs {
  int* p[4];
  long long n[4];
};
__global__ __launch_bounds__(256) void cn_zero_ints_kernel(const ZeroJobs z) {
  int* __restrict__ p = z.p[blockIdx.y];
  const long long n = z.n[blockIdx.y];
  if (!p) return;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = 0;
}

}  // namespace

extern "C" int cartnet_sort_by_key(const int64_t* keys, int32_t N, int32_t nkeys, int32_t* perm, int32_t* ptr,
                                   int32_t* status, void* stream) {
  CN_CHECK(N >= 0 && nkeys >= 1 && nkeys <= 512, "cartnet_sort_by_key: nkeys=%d out of range (1..512)", nkeys);
  CN_CHECK((keys || N == 0) && perm && ptr && status, "cartnet_sort_by_key: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(status, 0, sizeof(int32_t), st) != hipSuccess) {
    cartnet_set_error("cartnet_sort_by_key: memset failed");
    return 2;
  }
  hipLaunchKernelGGL(cn_sort_by_key_kernel, dim3(1), dim3(64 * SORT_WAVES), sizeof(int) * (SORT_WAVES + 1) * nkeys, st, keys, N,
                     nkeys, perm, ptr, status);
  CN_LAUNCH_CHECK("cartnet_sort_by_key");
  return 0;
}

extern "C" int cartnet_csc_build(const int32_t* src32, const int32_t* rowptr, const int64_t* graph_ptr, int32_t Bg, int32_t N,
                                 int64_t E, int32_t* colptr, int32_t* perm, int32_t* status, void* stream) {
  CN_CHECK(E >= 0 && N >= 0 && E < 2147483647LL, "cartnet_csc_build: bad sizes");
  CN_CHECK(src32 && rowptr && colptr && perm && status, "cartnet_csc_build: null pointer");
  CN_CHECK(graph_ptr == nullptr || Bg >= 1, "cartnet_csc_build: Bg=%d", Bg);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  ZeroJobs z;
  z.p[0] = colptr; z.n[0] = (long long)N + 1;
  z.p[1] = perm;   z.n[1] = (long long)E;
  z.p[2] = nullptr; z.n[2] = 0;
  z.p[3] = nullptr; z.n[3] = 0;
  long long most = z.n[0] > z.n[1] ? z.n[0] : z.n[1];
  int zb = (int)((most + 1023) / 1024);
  if (zb > 1024) zb = 1024;
  if (zb < 1) zb = 1;
  hipLaunchKernelGGL(cn_zero_ints_kernel, dim3(zb, 2), dim3(256), 0, st, z);
  CN_LAUNCH_CHECK("cartnet_csc_build/zero");
  const int ng = graph_ptr ? Bg : 1;
  hipLaunchKernelGGL(cn_csc_build_kernel, dim3(ng), dim3(256), 0, st, src32, rowptr, graph_ptr, ng, N, (int)E, colptr, perm,
                     status);
  CN_LAUNCH_CHECK("cartnet_csc_build");
  return 0;
}

extern "C" int cartnet_csr_build(const int64_t* edge_index, int64_t E, int32_t N, const int64_t* graph_ptr,
                                 int32_t Bg, int32_t* src32, int32_t* tgt32, int32_t* rowptr, int32_t* colptr,
                                 int32_t* perm, int32_t* status, void* stream) {
  CN_CHECK(E >= 0 && N >= 0, "cartnet_csr_build: negative sizes");
  CN_CHECK(E < 2147483647LL, "cartnet_csr_build: E=%lld does not fit int32", (long long)E);
  CN_CHECK((edge_index || E == 0) && src32 && tgt32 && rowptr && status, "cartnet_csr_build: null pointer");
  CN_CHECK((colptr == nullptr) == (perm == nullptr), "cartnet_csr_build: colptr and perm must pair");
  CN_CHECK(graph_ptr == nullptr || Bg >= 1, "cartnet_csr_build: Bg=%d", Bg);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // Defaults for entries a malformed batch (unsorted targets, edges across crystals, an oversized crystal) leaves
  // unwritten: empty segments and row 0 instead of whatever the buffers held; the status word starts at 0.  One launch
  // for the four buffers (four hipMemsetAsync calls were four ~6 us dispatches at the head of every forward pass).
  {
    ZeroJobs z;
    z.p[0] = status; z.n[0] = 1;
    z.p[1] = rowptr; z.n[1] = (long long)N + 1;
    z.p[2] = colptr; z.n[2] = colptr ? (long long)N + 1 : 0;
    z.p[3] = perm;   z.n[3] = perm ? (long long)E : 0;
    long long most = z.n[1] > z.n[3] ? z.n[1] : z.n[3];
    int zb = (int)((most + 1023) / 1024);
    if (zb > 1024) zb = 1024;
    if (zb < 1) zb = 1;
    hipLaunchKernelGGL(cn_zero_ints_kernel, dim3(zb, 4), dim3(256), 0, st, z);
    CN_LAUNCH_CHECK("cartnet_csr_build/zero");
  }
  int blocks = cn_ceil_div(E + 1, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cn_edge_convert_kernel, dim3(blocks), dim3(256), 0, st, edge_index, (long long)E, N, src32,
                     tgt32, rowptr, status);
  CN_LAUNCH_CHECK("cartnet_csr_build/convert");
  if (colptr) {
    const int ng = graph_ptr ? Bg : 1;
    hipLaunchKernelGGL(cn_csc_build_kernel, dim3(ng), dim3(256), 0, st, src32, rowptr, graph_ptr, ng, N, (int)E,
                       colptr, perm, status);
    CN_LAUNCH_CHECK("cartnet_csr_build/csc");
  }
  return 0;
}
