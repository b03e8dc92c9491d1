// Micro-benchmark (round 3): the bf16x3 main loop's matrix work alone -- operands in LDS (the kernel's plane images, random
// fp32 values split h / m / l), no global traffic, two 512-thread workgroups per CU, one barrier per K-step of 16 -- in
// two MFMA shapes at the same 64 x 64 output tile per wave and the same six products per element:
//   mode 0: v_mfma_f32_32x32x16_bf16, 12 ds_read_b128 + 24 MFMAs per wave and K-step (the shipped kernel's loop body)
//   mode 1: v_mfma_f32_16x16x32_bf16 with TWO products per instruction: K slots 0-15 take one piece pair, 16-31 another
//           ([h|m] x [h|h] = hh + mh, [h|m] x [m|m] = hm + mm, [h|l] x [l|h] = hl + lh): 20 ds_read_b128 + 48 MFMAs
//   mode 2: mode 1's MFMAs with only 12 reads (B fragments reused: wrong arithmetic, isolates the shape from the LDS bytes)
// Prints wall time per K-step pair of a CU, the in-kernel clock (s_memtime / s_memrealtime) and the matrix-pipe occupancy.
// hipcc --offload-arch=gfx950 -O3 -o tools/experiments/micro/x3_shape tools/experiments/micro/x3_shape.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int A_PLANE = 128 * 32, B_PLANE = 256 * 32, BUF = 3 * (A_PLANE + B_PLANE);
__device__ __forceinline__ int off(int row, int kh) { return row * 32 + ((kh ^ ((row >> 4) & 1)) << 4); }

template <int MODE>
__global__ __launch_bounds__(512, 4) void k(const float* __restrict__ seed, float* out, unsigned long long* stamps, int nsteps) {
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 2, wn = wid & 3;
  // fill both buffers: (128 + 256) rows x 16 k per buffer, three planes
  for (int u = tid; u < 2 * 384 * 4; u += 512) {
    const int buf = u / (384 * 4), r = (u / 4) % 384, kq = u & 3;
    const f32x4 v = *reinterpret_cast<const f32x4*>(seed + ((size_t)(blockIdx.x & 63) * 2 * 384 * 4 + u) * 4);
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
    const bf16x4 m = __builtin_convertvector(r1, bf16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);
    const bf16x4 l = __builtin_convertvector(r2, bf16x4);
    const bool isA = r < 128;
    const int row = isA ? r : r - 128, plane = isA ? A_PLANE : B_PLANE;
    char* d = lds + buf * BUF + (isA ? 0 : 3 * A_PLANE) + off(row, kq >> 1) + (kq & 1) * 8;
    *reinterpret_cast<bf16x4*>(d) = h;
    *reinterpret_cast<bf16x4*>(d + plane) = m;
    *reinterpret_cast<bf16x4*>(d + 2 * plane) = l;
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 0) {
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int u = 0; u < nsteps; ++u) {
      const char* cA = lds + (u & 1) * BUF;
      const char* cB = cA + 3 * A_PLANE;
      bf16x8 ah[2], am[2], al[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const char* q = cA + off(wm * 64 + a * 32 + li, lh);
        ah[a] = *reinterpret_cast<const bf16x8*>(q);
        am[a] = *reinterpret_cast<const bf16x8*>(q + A_PLANE);
        al[a] = *reinterpret_cast<const bf16x8*>(q + 2 * A_PLANE);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const char* q = cB + off(wn * 64 + b * 32 + li, lh);
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(q);
        const bf16x8 bm = *reinterpret_cast<const bf16x8*>(q + B_PLANE);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(q + 2 * B_PLANE);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bm, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bh, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bm, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh, acc[a][b], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[(size_t)blockIdx.x * 512 + tid] = s;
  } else {
    const int li = lane & 15, lq = lane >> 4, kh = lq & 1, up = lq >> 1;
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    // per-lane plane choice of the concatenated operands
    const int a_hm = up ? A_PLANE : 0, a_hl = up ? 2 * A_PLANE : 0;
    const int b_hh = 0, b_mm = B_PLANE, b_lh = up ? 0 : 2 * B_PLANE;
    for (int u = 0; u < nsteps; ++u) {
      const char* cA = lds + (u & 1) * BUF;
      const char* cB = cA + 3 * A_PLANE;
      bf16x8 ahm[4], ahl[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const char* q = cA + off(wm * 64 + a * 16 + li, kh);
        ahm[a] = *reinterpret_cast<const bf16x8*>(q + a_hm);
        ahl[a] = *reinterpret_cast<const bf16x8*>(q + a_hl);
      }
      bf16x8 bhh, bmm, blh;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (MODE == 1 || b == 0) {
          const char* q = cB + off(wn * 64 + b * 16 + li, kh);
          bhh = *reinterpret_cast<const bf16x8*>(q + b_hh);
          bmm = *reinterpret_cast<const bf16x8*>(q + b_mm);
          blh = *reinterpret_cast<const bf16x8*>(q + b_lh);
        } else if (b == 2) {
          const char* q = cB + off(wn * 64 + b * 16 + li, kh);
          bhh = *reinterpret_cast<const bf16x8*>(q + b_hh);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahl[a], blh, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm[a], bmm, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm[a], bhh, acc[a][b], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
    out[(size_t)blockIdx.x * 512 + tid] = s;
  }
  if (tid == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - w0;
  }
}

template <int MODE>
static void run(const float* seed, float* out, unsigned long long* stamps, int nsteps, const char* what, bool zero) {
  const int grid = 512;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // hold the load for ~2.5 s, then time
  float ms = 0.f, total = 0.f;
  while (total < 2500.f) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<MODE><<<grid, 512>>>(seed, out, stamps, nsteps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    total += ms;
  }
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) k<MODE><<<grid, 512>>>(seed, out, stamps, nsteps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * grid);
  hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
  std::vector<double> ghz, cyc;
  for (int i = 0; i < grid; ++i) if (h[2 * i + 1]) { ghz.push_back(0.1 * h[2 * i] / h[2 * i + 1]); cyc.push_back((double)h[2 * i]); }
  std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
  const double us = 1e3 * ms / 10, per_step = cyc[cyc.size() / 2] / nsteps;
  // 2 x 177140 x 512 x 16 x 6 bf16 FLOP per K-step of the layer product correspond to 1384 x 2 tiles; here 512 tiles per launch
  const double tf = 512.0 * 128 * 256 * 16 * 2 * 6 * nsteps / (us * 1e-6) / 1e12;
  printf("%-44s %-6s: %8.1f us per launch, %6.1f bf16 TFLOP/s, clock %.3f GHz, %6.0f cycles per K-step (3072 = matrix pipe full)\n",
         what, zero ? "zeros" : "random", us, tf, ghz[ghz.size() / 2], per_step);
  fflush(stdout);
}

int main() {
  const size_t n = (size_t)64 * 2 * 384 * 16;
  std::vector<float> h(n);
  srand(1);
  for (size_t i = 0; i < n; ++i) {   // roughly normal
    float s = 0.f;
    for (int j = 0; j < 6; ++j) s += (float)rand() / RAND_MAX - 0.5f;
    h[i] = s * 1.4f;
  }
  float *seed, *zeros, *out; unsigned long long* stamps;
  hipMalloc(&seed, n * 4); hipMalloc(&zeros, n * 4); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&stamps, 2 * 512 * 8);
  hipMemcpy(seed, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(zeros, 0, n * 4);
  const int nsteps = 2048;
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(seed, out, stamps, nsteps, "32x32x16, 12 reads + 24 MFMAs per K-step", false);
    run<1>(seed, out, stamps, nsteps, "16x16x32 two products each, 20 reads + 48", false);
    run<2>(seed, out, stamps, nsteps, "16x16x32, 12 reads (B reused) + 48", false);
  }
  run<0>(zeros, out, stamps, nsteps, "32x32x16, 12 reads + 24 MFMAs per K-step", true);
  run<1>(zeros, out, stamps, nsteps, "16x16x32 two products each, 20 reads + 48", true);
  return 0;
}
