// Helpers shared by the translation units that sequence a whole network (model.hip: CartNet; icomformer.hip): workspace
// carving, the split-K rule of the weight-gradient products, the two-stream event plumbing.  Host code only.
#pragma once
#include "common.h"
#include <vector>

namespace cn_model {

constexpr int BM_TILE = 128;

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
inline int tiles_m(long long M) { return (int)((M + BM_TILE - 1) / BM_TILE); }

struct Carver {
  char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t count) {
    off = align_up(off);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

// Split-K of the weight-gradient products: ONE workgroup per CU (256), not two.  Round 4, same-box A B A B: the step
// 14.63 / 14.92 -> 14.08 / 14.14 ms (fp32), 10.68 -> 10.59 (bf16x3), 6.55 -> 6.32 (bf16 + bf16 storage).  These launches
// live on the weight-gradient stream for the whole of backward; with 512 long-lived workgroups they took BOTH slots of
// every CU whenever the main stream's kernel retired, and every kernel of the critical chain -- above all its ~30 small
// ones -- then waited for a slot (rocprof: 5 us finalisers reading 38 us in-step).  One resident workgroup per CU runs the
// matrix pipe as well as two (0.97 of it alone, profiles/r03_exp_phases.md) and leaves the other half of every CU's
// registers to the chain that the step's length depends on.  cn_gemm_f32tn_kernel enforces it with its LDS footprint
// (four stages = 96 KB: a second one does not fit, gemm_f32.h), which is also one more K-step of prefetch.
#ifndef CN_WGRAD_TARGET
#define CN_WGRAD_TARGET 256
#endif
inline int split_k(long long K, int tiles) {
  long long s = (CN_WGRAD_TARGET + tiles - 1) / tiles;
  if (s > K / 256) s = K / 256;
  if (s < 1) s = 1;
  return (int)s;
}
inline int wgrad_tiles(int groups, int M, int N) { return groups * ((M + 127) / 128) * (N > 128 ? (N + 255) / 256 : 1); }
inline size_t wgrad_slab_floats(int groups, long long K, int M, int N) {
  const int S = split_k(K, wgrad_tiles(groups, M, N));
  return S > 1 ? (size_t)groups * S * M * N : 0;
}
#ifdef CN_HOST_PROFILE
// Diagnostic build only (tools/host_profile.py): host time of every RUN(...) statement of the sequencing code, by source
// line, summed over calls; read and reset through cartnet_debug_host_profile.  The product build has none of this.
}  // namespace cn_model
#include <chrono>
namespace cn_model {
struct HostProf { double us[4096]; long n[4096]; const char* what[4096]; };
inline HostProf& host_prof() { static HostProf p{}; return p; }
#define RUN(call)                                                                                   \
  do {                                                                                              \
    const auto _t0 = std::chrono::steady_clock::now();                                              \
    int _rc = (call);                                                                               \
    const auto _t1 = std::chrono::steady_clock::now();                                              \
    auto& _p = cn_model::host_prof();                                                               \
    _p.us[__LINE__ & 4095] += std::chrono::duration<double, std::micro>(_t1 - _t0).count();         \
    _p.n[__LINE__ & 4095] += 1;                                                                     \
    _p.what[__LINE__ & 4095] = #call;                                                               \
    if (_rc != 0) return _rc;                                                                       \
  } while (0)
#else
#define RUN(call)            \
  do {                       \
    int _rc = (call);        \
    if (_rc != 0) return _rc; \
  } while (0)
#endif

// Events that order the weight-gradient stream against the main stream (created once per thread, timing disabled).
struct EventPool {
  std::vector<hipEvent_t> ev;
  size_t next = 0;
  hipEvent_t get() {
    if (next == ev.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      ev.push_back(e);
    }
    return ev[next++];
  }
};

#ifdef CN_HOST_PROFILE
struct HostProfScope {
  int slot; const char* what; std::chrono::steady_clock::time_point t0;
  HostProfScope(int s, const char* w) : slot(s), what(w), t0(std::chrono::steady_clock::now()) {}
  ~HostProfScope() {
    auto& p = host_prof();
    p.us[slot] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    p.n[slot] += 1; p.what[slot] = what;
  }
};
#define CN_PROF_SCOPE(slot, what) cn_model::HostProfScope _scope(slot, what)
#else
#define CN_PROF_SCOPE(slot, what)
#endif

struct Streams {
  hipStream_t main, side;
  bool dual;
  EventPool* pool;      // the calling thread's pool (thread_local in the translation unit that sequences the call)
  // after(main) -> side waits; returns 0 on success
  int fork() {
    CN_PROF_SCOPE(4001, "Streams::fork (record + wait)");
    if (!dual) return 0;
    hipEvent_t e = pool->get();
    if (!e || hipEventRecord(e, main) != hipSuccess || hipStreamWaitEvent(side, e, 0) != hipSuccess) return 2;
    return 0;
  }
  // the two halves of fork(), for call sites that put the main stream's next kernels in the queue BEFORE spending host
  // time on the side stream's launches (a batch of tiny crystals is host-paced: what is enqueued first starts first)
  hipEvent_t mark_main() {
    CN_PROF_SCOPE(4002, "Streams::mark_main (record)");
    if (!dual) return nullptr;
    hipEvent_t e = pool->get();
    if (!e || hipEventRecord(e, main) != hipSuccess) return nullptr;
    return e;
  }
  int side_waits(hipEvent_t e) {
    CN_PROF_SCOPE(4003, "Streams::side_waits (wait)");
    if (!dual) return 0;
    if (!e) return 2;
    return hipStreamWaitEvent(side, e, 0) == hipSuccess ? 0 : 2;
  }
  hipEvent_t mark_side() {
    CN_PROF_SCOPE(4004, "Streams::mark_side (record)");
    if (!dual) return nullptr;
    hipEvent_t e = pool->get();
    if (!e || hipEventRecord(e, side) != hipSuccess) return nullptr;
    return e;
  }
  int main_waits(hipEvent_t e) {
    CN_PROF_SCOPE(4005, "Streams::main_waits (wait)");
    if (!dual || !e) return 0;
    return hipStreamWaitEvent(main, e, 0) == hipSuccess ? 0 : 2;
  }
};

}  // namespace cn_model
