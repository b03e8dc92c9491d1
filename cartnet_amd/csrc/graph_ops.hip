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

// Stable counting sort of N items by an int64 key in [0, nkeys), one workgroup (atom types: nkeys = 119).
__global__ __launch_bounds__(1024) void cn_sort_by_key_kernel(const int64_t* __restrict__ keys, int N, int nkeys,
                                                              int* __restrict__ perm, int* __restrict__ ptr,
                                                              int* __restrict__ status) {
  __shared__ int cnt[1024];
  __shared__ int skey[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < nkeys; i += 1024) cnt[i] = 0;
  __syncthreads();
  for (int i = tid; i < N; i += 1024) {
    long long k = keys[i];
    if (k < 0 || k >= nkeys) {     // clamped like the forward gather (cn_node_embed_kernel), so perm stays a full permutation
      atomicOr(status, 16);
      k = k < 0 ? 0 : nkeys - 1;
    }
    atomicAdd(&cnt[(int)k], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < nkeys; ++i) {
      const int v = cnt[i];
      cnt[i] = run;
      ptr[i] = run;
      run += v;
    }
    ptr[nkeys] = run;
  }
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + tid;
    int key = -1;
    if (i < N) {
      const long long k = keys[i];
      key = k < 0 ? 0 : (k >= nkeys ? nkeys - 1 : (int)k);
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
      perm[cur + rank] = i;
      atomicAdd(&cnt[key], 1);
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int cartnet_sort_by_key(const int64_t* keys, int32_t N, int32_t nkeys, int32_t* perm, int32_t* ptr,
                                   int32_t* status, void* stream) {
  CN_CHECK(N >= 0 && nkeys >= 1 && nkeys <= 1024, "cartnet_sort_by_key: nkeys=%d out of range (1..1024)", nkeys);
  CN_CHECK((keys || N == 0) && perm && ptr && status, "cartnet_sort_by_key: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(status, 0, sizeof(int32_t), st) != hipSuccess) {
    cartnet_set_error("cartnet_sort_by_key: memset failed");
    return 2;
  }
  hipLaunchKernelGGL(cn_sort_by_key_kernel, dim3(1), dim3(1024), 0, st, keys, N, nkeys, perm, ptr, status);
  CN_LAUNCH_CHECK("cartnet_sort_by_key");
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
  if (hipMemsetAsync(status, 0, sizeof(int32_t), st) != hipSuccess) {
    cartnet_set_error("cartnet_csr_build: memset failed");
    return 2;
  }
  // Defaults for entries a malformed batch (unsorted targets, edges across crystals, an oversized crystal) leaves
  // unwritten: empty segments and row 0 instead of whatever the buffers held.
  if (hipMemsetAsync(rowptr, 0, sizeof(int32_t) * ((size_t)N + 1), st) != hipSuccess ||
      (colptr && hipMemsetAsync(colptr, 0, sizeof(int32_t) * ((size_t)N + 1), st) != hipSuccess) ||
      (perm && E > 0 && hipMemsetAsync(perm, 0, sizeof(int32_t) * (size_t)E, st) != hipSuccess)) {
    cartnet_set_error("cartnet_csr_build: memset failed");
    return 2;
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
