// Inference-mode fusion of a CartNet layer's second Linears with the gate (cartnet_gate_gemm_eval, include/cartnet_hip.h).
//
// In eval mode the edge BatchNorm normalises with its running statistics, so nothing about the gate needs the whole
// batch: sigma = env * sigmoid(bn(g)), sigma * sender, the sum over a target's edges and e_out = e_in + sigma can follow
// the two products g = silu(pre_g) W2g^T + b and s = silu(pre_s) W2a^T + b in the same kernel, and gs [E, 2D] never
// reaches memory (reference: models/cartnet.py:230-262 with `self.training == False`).  By the launch cost model of
// DESIGN.md section 5 that removes one [E, 2D] epilogue store (46 us), the gate kernel (166 us) and a launch per layer and
// adds one [E, D] read + one [E, D] write.
//
// One workgroup owns 128 edge rows x 128 channels and runs the 128-wide DMA-fed fp32 pipeline of gemm_f32w128.h TWICE
// (gate half, then sender half: the two accumulator sets, 32 + 32 VGPRs, live side by side).  Epilogue, all through one
// [128][132] fp32 tile in LDS: (A) sigma -> LDS -> e_out = e_in + sigma as full 512-byte row segments; (B) sigma * sender
// -> LDS -> per-target sums in edge order.  Edges are sorted by target, so a tile holds whole targets except possibly its
// first and last one: those two partial rows go to a boundary buffer [tile][2][D] and cn_gate_fixup_kernel adds the
// pieces of every such target in tile order (deterministic; targets without edges get zeros).
#include "gemm_f32w128.h"

namespace cn_gemm {

constexpr int GATE_LD = 132;                         // floats per row of the LDS tile (128 + 4)
constexpr int GATE_TILE_FLOATS = BM * GATE_LD;       // 16,896 floats = 67,584 B
constexpr int GATE_TAB_INTS = 4 * BM + 8;            // tgt per row, segment starts, segment targets, counters, envelope per row
static_assert(GATE_TILE_FLOATS * 4 >= 2 * W128_BUF_BYTES, "the LDS tile must cover the two pipeline stages it overlays");

struct GateArgs {
  const float* pre; int ldp;
  const char* img[2];
  const float* bias[2];
  const float *mean_rstd, *gamma, *beta, *env, *e_in;
  float* e_out;
  const int* tgt;
  float *aggr, *bnd;
  int E, D;
};

__global__ __launch_bounds__(NTHREADS, 4) void cn_gemm_f32gate_kernel(const GateArgs p) {
  using S = Shape<W128_BN>;
  static_assert(S::TM == 2 && S::TN == 1 && S::WGM == 2 && S::WGN == 4, "wave tile is 64 x 32");
  __shared__ __attribute__((aligned(16))) float smem[GATE_TILE_FLOATS + GATE_TAB_INTS];
  char* lds = reinterpret_cast<char*>(smem);
  int* tab = reinterpret_cast<int*>(smem + GATE_TILE_FLOATS);     // [0,128) tgt, [128,256) seg start, [256,384) seg tgt, [384..] counters

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / S::WGN, wn = wid % S::WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_n = p.D / W128_BN;
  int bx, gdummy;
  cn_block_map(bx, gdummy, tiles_n);            // the column tiles of a row tile read the same rows of `pre`: same XCD
  const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * W128_BN;
  const int nsteps = p.D / BK;                  // K = D

  f32x16 acc[2][1], accg[2];
  const int arow = tid >> 2, akq = tid & 3;
  const unsigned a_voff = ((unsigned)min(row0 + arow, p.E - 1) * (unsigned)p.ldp + akq * 4) * 4u;
  const int a_lds = (arow * KPAD + akq * 4) * 4;
  const unsigned b_voff = lane * 16;
  const float* a0 = nullptr;
  const char* b0 = nullptr;
  const unsigned lds_b = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + F32_A_BYTES + wid * 1024;

  auto a_issue = [&](f32x4& dst, int v) {
    const float* base = a0 + v * BK;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(a_voff), "s"(base) : "memory");
  };
  auto a_store = [&](f32x4 v, int buf) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = fast_silu(v[c]);
    *reinterpret_cast<f32x4*>(lds + buf * W128_BUF_BYTES + a_lds) = v;
  };
  auto b_issue = [&](int v, int buf) {
    const char* src = b0 + (size_t)v * F32_B_BYTES + wid * 1024;
    const unsigned dst = lds_b + buf * W128_BUF_BYTES;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(dst), "v"(b_voff), "s"(src) : "memory", "m0");
  };
  f32x4 af[2][2], bf[2];
  auto frags = [&](int buf, int kg) {
    const float* sA = reinterpret_cast<const float*>(lds + buf * W128_BUF_BYTES);
    const char* sB = lds + buf * W128_BUF_BYTES + F32_A_BYTES;
#pragma unroll
    for (int a = 0; a < 2; ++a)
      af[kg][a] = *reinterpret_cast<const f32x4*>(&sA[(wm * S::WM + a * 32 + li) * KPAD + kg * 8 + lh * 4]);
    bf[kg] = *reinterpret_cast<const f32x4*>(sB + f32_swz(wn * S::WN + li, kg * 2 + lh));
  };
  auto mma2 = [&](int kg, int j) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
      acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg][a][j], bf[kg][j], acc[a][0], 0, 0, 0);
  };
  auto step = [&](auto cur_c, int u, f32x4& r) {
    constexpr int CUR = decltype(cur_c)::value;
    frags(CUR, 0);
    frags(CUR, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < nsteps) {
      a_store(r, CUR ^ 1);
      b_issue(u + 1, CUR ^ 1);
    }
    if (u + 3 < nsteps) a_issue(r, u + 3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int j = 0; j < 4; ++j) mma2(kg, j);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 3 < nsteps) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto step_full = [&](auto cur_c, int u, f32x4& r) {      // steady state (u + 3 < nsteps), as in gemm_f32w128.h
    constexpr int CUR = decltype(cur_c)::value;
    frags(CUR, 0);
    frags(CUR, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = fast_silu(r[c]);
    __builtin_amdgcn_sched_barrier(0);
    mma2(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<f32x4*>(lds + (CUR ^ 1) * W128_BUF_BYTES + a_lds) = r;
    __builtin_amdgcn_sched_barrier(0);
    mma2(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    b_issue(u + 1, CUR ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    mma2(0, 2);
    __builtin_amdgcn_sched_barrier(0);
    a_issue(r, u + 3);
    __builtin_amdgcn_sched_barrier(0);
    mma2(0, 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) mma2(1, j);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // one product over K = D into `acc` (the pipeline of gemm_f32w128.h with the SiLU prologue)
  auto run_pass = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;
    // all of the pipeline head's loads in ONE memory round trip (K-step 0 into a third register set, 1 and 2 into the ring)
    f32x4 r0, r1, rt;
    a_issue(rt, 0);
    b_issue(0, 0);
    if (nsteps > 1) a_issue(r1, 1);
    if (nsteps > 2) a_issue(r0, 2);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(rt) :: "memory");
    a_store(rt, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int u = 0;
    for (; u + 4 < nsteps; u += 2) {
      step_full(std::integral_constant<int, 0>{}, u, r1);
      step_full(std::integral_constant<int, 1>{}, u + 1, r0);
    }
    for (; u < nsteps; u += 2) {
      step(std::integral_constant<int, 0>{}, u, r1);
      if (u + 1 < nsteps) step(std::integral_constant<int, 1>{}, u + 1, r0);
    }
  };

  // image of a [K = D, N = D] operand: per 256-column tile and K-step a [256][16] block of 16 KB; a 128-column tile is one half
  const size_t img_off = (size_t)(tile_n >> 1) * nsteps * F32_B_BYTES + (size_t)(tile_n & 1) * W128_B_BYTES;
  a0 = p.pre;
  b0 = p.img[0] + img_off;
  run_pass();
#pragma unroll
  for (int a = 0; a < 2; ++a) accg[a] = acc[a][0];
  a0 = p.pre + p.D;                          // the sender half of pre
  b0 = p.img[1] + img_off;
  run_pass();

  // per-row operands of the tile, once, through the table behind the tile (it does not overlap the pipeline stages):
  // target and envelope of row r
  const int nrows = min(BM, p.E - row0);
  float* envtab = reinterpret_cast<float*>(tab + 3 * BM + 8);
  if (tid < BM) {
    tab[tid] = tid < nrows ? p.tgt[row0 + tid] : -1;
    envtab[tid] = (p.env && tid < nrows) ? p.env[row0 + tid] : 1.0f;
  }
  __syncthreads();
  // ---- gate in the accumulator layout: lane = channel col0 + wn*32 + li; row = wm*64 + a*32 + (r&3) + 8*(r>>2) + 4*lh
  const int ch = col0 + wn * S::WN + li;
  const float bg = p.bias[0][ch], bs = p.bias[1][ch];
  const float mean = p.mean_rstd[ch], scale = p.mean_rstd[p.D + ch] * p.gamma[ch], shift = p.beta[ch];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = wm * S::WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float sig = envtab[rl] * fast_sigmoid((accg[a][r] + bg - mean) * scale + shift);
      accg[a][r] = sig;
      acc[a][0][r] = sig * (acc[a][0][r] + bs);
    }
  // (A) sigma -> LDS (the last K-step ended with a workgroup barrier: the pipeline stages are free)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      smem[(wm * S::WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * GATE_LD + wn * S::WN + li] = accg[a][r];
  {  // e_out = e_in + sigma: a row of the tile is 512 B = 32 lanes x 16 B; a wave instruction covers two rows.
    // All eight e_in rows of a thread are loaded first -- behind the barrier they fly across -- and the eight stores go
    // out back to back from eight register sets (the gate accumulators are dead by now): loads and stores retire
    // through one in-order counter, and a row-by-row loop waits for every store's acknowledgement (r03_exp_phases.md).
    const int c4 = (tid & 31) * 4, rsub = tid >> 5;         // 16 row slots
    constexpr int HALF = BM / 32;                           // 4 rows at a time: 8 spill the accumulators
    f32x4 ein[HALF];
    __builtin_amdgcn_sched_barrier(0);                      // not above the sigma stores: their registers are these
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
      const int rl = min(rsub + 16 * i, nrows - 1);         // rows past the end re-read the last one and are not stored
      ein[i] = *reinterpret_cast<const f32x4*>(p.e_in + (size_t)(row0 + rl) * p.D + col0 + c4);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < HALF; ++i)
        ein[i] += *reinterpret_cast<const f32x4*>(&smem[(rsub + 16 * (h * HALF + i)) * GATE_LD + c4]);
#pragma unroll
      for (int i = 0; i < HALF; ++i) {
        const int rl = rsub + 16 * (h * HALF + i);
        if (rl < nrows) *reinterpret_cast<f32x4*>(p.e_out + (size_t)(row0 + rl) * p.D + col0 + c4) = ein[i];
      }
      if (h == 0) {
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
          const int rl = min(rsub + 16 * (HALF + i), nrows - 1);
          ein[i] = *reinterpret_cast<const f32x4*>(p.e_in + (size_t)(row0 + rl) * p.D + col0 + c4);
        }
      }
    }
  }
  // segment starts: row r opens a segment if it is the tile's first row or its target differs from the row before
  // (tab: [0, BM) target per row, [BM, 2BM) segment starts, [2BM, 3BM) segment targets, [3BM], [3BM + 1] starts in waves 0 / 1)
  int rk = -1, tg = -1;
  if (tid < BM) {                     // waves 0 and 1, all lanes: rows 0-63 / 64-127
    tg = tab[tid];
    const bool start = tid < nrows && (tid == 0 || tab[tid - 1] != tg);
    const unsigned long long bal = __ballot(start);
    if (lane == 0) tab[3 * BM + wid] = __popcll(bal);
    if (start) rk = __popcll(bal & ((1ull << lane) - 1ull));
  }
  __syncthreads();      // every wave is done reading sigma from the tile; the per-wave counts are visible
  const int c0n = tab[3 * BM + 0], c1n = tab[3 * BM + 1];
  const int J = c0n + c1n;
  if (rk >= 0) {
    const int pos = rk + (wid == 1 ? c0n : 0);
    tab[BM + pos] = tid;
    tab[2 * BM + pos] = tg;
  }
  // (B) sigma * sender -> LDS
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      smem[(wm * S::WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * GATE_LD + wn * S::WN + li] = acc[a][0][r];
  __syncthreads();
  {  // per-target sums in edge order: thread = (channel c, target class q); consecutive threads walk consecutive columns
    const int c = tid & (W128_BN - 1), q = tid >> 7;
    for (int j = q; j < J; j += NTHREADS / W128_BN) {
      const int k0 = tab[BM + j], k1 = (j + 1 < J) ? tab[BM + j + 1] : nrows;
      float s = 0.f;
      for (int k = k0; k < k1; ++k) s += smem[k * GATE_LD + c];
      if (j > 0 && j < J - 1) p.aggr[(size_t)tab[2 * BM + j] * p.D + col0 + c] = s;       // a whole target
      else p.bnd[((size_t)tile_m * 2 + (j == 0 ? 0 : 1)) * p.D + col0 + c] = s;           // the tile's first / last one
    }
  }
}

// Targets that are not interior to a tile: sum their pieces (boundary rows) in tile order; targets without edges: zeros.
__global__ __launch_bounds__(256) void cn_gate_fixup_kernel(const int* __restrict__ rowptr, const float* __restrict__ bnd,
                                                            int N, int E, int D, float* __restrict__ aggr) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= N) return;
  const int k0 = rowptr[t], k1 = rowptr[t + 1];
  const int tile0 = k0 >> 7, tile1 = k1 > k0 ? (k1 - 1) >> 7 : tile0;
  const bool first = k0 == (tile0 << 7);
  const int tend = min((tile1 + 1) << 7, E);
  const bool last = k1 == tend;
  if (k1 > k0 && tile0 == tile1 && !first && !last) return;          // interior: the GEMM kernel wrote it
  for (int c = lane * 4; c < D; c += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (k1 > k0)
      for (int tile = tile0; tile <= tile1; ++tile) {
        const int slot = (tile > tile0 || first) ? 0 : 1;
        acc += *reinterpret_cast<const f32x4*>(bnd + ((size_t)tile * 2 + slot) * D + c);
      }
    *reinterpret_cast<f32x4*>(aggr + (size_t)t * D + c) = acc;
  }
}

}  // namespace cn_gemm

extern "C" size_t cartnet_gate_gemm_eval_workspace(int64_t E, int32_t D) {
  if (E <= 0 || D <= 0) return 0;
  return (size_t)((E + cn_gemm::BM - 1) / cn_gemm::BM) * 2 * (size_t)D * sizeof(float);
}

extern "C" int cartnet_gate_gemm_eval(const CartnetGateGemmArgs* a, void* stream) {
  CN_CHECK(a != nullptr, "cartnet_gate_gemm_eval: null arguments");
  CN_CHECK(a->D >= 256 && a->D % 256 == 0 && a->D <= 4096,
           "cartnet_gate_gemm_eval: D=%d must be a multiple of 256 (the weight images are built per 256-column tile)", a->D);
  CN_CHECK(a->N >= 0 && a->E >= 0 && a->E < 2147483647LL, "cartnet_gate_gemm_eval: bad sizes");
  CN_CHECK(a->ldp >= 2 * a->D && a->ldp % 4 == 0, "cartnet_gate_gemm_eval: ldp=%d", a->ldp);
  CN_CHECK((double)a->E * a->ldp * 4.0 < 4294967296.0, "cartnet_gate_gemm_eval: pre does not fit 32-bit byte offsets");
  CN_CHECK(a->rowptr && a->aggr, "cartnet_gate_gemm_eval: null rowptr / aggr");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->E > 0) {
    CN_CHECK(a->pre && a->img_gate && a->img_aggr && a->bias_gate && a->bias_aggr && a->mean_rstd && a->gamma && a->beta &&
                 a->e_in && a->e_out && a->tgt && a->bnd,
             "cartnet_gate_gemm_eval: null pointer");
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    CN_CHECK(al16(a->pre) && al16(a->e_in) && al16(a->e_out) && al16(a->aggr) && al16(a->bnd) && al16(a->img_gate) &&
                 al16(a->img_aggr),
             "cartnet_gate_gemm_eval: operands must be 16-byte aligned");
    cn_gemm::GateArgs g;
    g.pre = a->pre; g.ldp = a->ldp;
    g.img[0] = static_cast<const char*>(a->img_gate); g.img[1] = static_cast<const char*>(a->img_aggr);
    g.bias[0] = a->bias_gate; g.bias[1] = a->bias_aggr;
    g.mean_rstd = a->mean_rstd; g.gamma = a->gamma; g.beta = a->beta; g.env = a->env; g.e_in = a->e_in; g.e_out = a->e_out;
    g.tgt = a->tgt; g.aggr = a->aggr; g.bnd = a->bnd; g.E = (int)a->E; g.D = a->D;
    const int tiles = (int)((a->E + cn_gemm::BM - 1) / cn_gemm::BM) * (a->D / 128);
    hipLaunchKernelGGL(cn_gemm::cn_gemm_f32gate_kernel, dim3(tiles), dim3(cn_gemm::NTHREADS), 0, st, g);
    CN_LAUNCH_CHECK("cartnet_gate_gemm_eval");
  }
  if (a->N > 0) {
    hipLaunchKernelGGL(cn_gemm::cn_gate_fixup_kernel, dim3((a->N + 3) / 4), dim3(256), 0, st, a->rowptr, a->bnd, a->N, (int)a->E,
                       a->D, a->aggr);
    CN_LAUNCH_CHECK("cartnet_gate_gemm_eval/fixup");
  }
  return 0;
}
