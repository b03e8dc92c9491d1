// Micro-benchmark (round 3): HBM write rate of the GEMM epilogue's store pattern (a wave instruction = 8 rows x 128 B,
// rows 2 KB apart) against full-row stores (a wave instruction = 1 KB contiguous), same bytes, same grid.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern tools/experiments/micro/store_pattern.hip && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// tile = 128 rows x 256 cols fp32 (ld = 512: one of two column groups), 8 waves (2 x 4), wave = 64 x 64, blocks of 32 x 32
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int ld, int tiles) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 2, wn = wid & 3;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    float* base = out + (size_t)t * 128 * ld;
    f32x4 v = {1.f, 2.f, 3.f, (float)t};
    if (MODE == 0) {            // epilogue pattern
      const int c4 = lane & 7, rsub = lane >> 3;
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(base + (size_t)(wm * 64 + a * 32 + rsub + 8 * i) * ld + wn * 64 + b * 32 + c4 * 4) = v;
    } else {                    // full rows: wave w writes rows w*16 .. w*16+15, 1 KB each
      for (int r = 0; r < 16; ++r)
        *reinterpret_cast<f32x4*>(base + (size_t)(wid * 16 + r) * ld + lane * 4) = v;
    }
  }
}
int main() {
  const long long E = 177140; const int ld = 512; const int tiles = (int)(E / 128);
  float* d; hipMalloc(&d, (size_t)E * ld * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int grid : {512, 1384}) {
      for (int it = 0; it < 3; ++it) { if (mode == 0) k<0><<<grid, 512>>>(d, ld, tiles); else k<1><<<grid, 512>>>(d, ld, tiles); }
      hipEventRecord(e0);
      for (int it = 0; it < 20; ++it) { if (mode == 0) k<0><<<grid, 512>>>(d, ld, tiles); else k<1><<<grid, 512>>>(d, ld, tiles); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)tiles * 128 * 256 * 4;
      printf("mode %d (%s) grid %4d: %7.1f us per pass, %.2f TB/s\n", mode, mode ? "full 1 KB rows" : "8 rows x 128 B", grid, 1e3 * ms / 20, bytes / (ms / 20 * 1e-3) / 1e12);
    }
  return 0;
}
