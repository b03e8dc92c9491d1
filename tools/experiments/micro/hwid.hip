// Which hardware wave slots do the waves of co-resident 512-thread workgroups get?  (round 3, for the asymmetric-priority experiment)
// hipcc --offload-arch=gfx950 -O3 -o tools/experiments/micro/hwid tools/experiments/micro/hwid.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
__global__ __launch_bounds__(512, 4) void k(unsigned* out, int spin) {
  __shared__ float lds[13000];          // ~52 KB like the GEMM kernels: two workgroups per CU
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = (float)hwid;
  float v = lds[(threadIdx.x * 7) % 512];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;          // stay resident for a while
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = hwid;
    out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (v == 12345.f ? 1 : 0);
  }
}
int main() {
  const int B = 1024;
  unsigned* d; hipMalloc(&d, B * 8 * 2 * 4);
  k<<<B, 512>>>(d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(B * 8 * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  for (int b : {0, 1, 2, 255, 256, 257, 511, 512, 513, 1023}) {
    printf("block %4d:", b);
    for (int w = 0; w < 8; ++w) {
      unsigned id = h[(b * 8 + w) * 2], xcc = h[(b * 8 + w) * 2 + 1] & 0xf;
      printf("  w%d[xcc%u se%u cu%2u simd%u slot%u]", w, xcc, (id >> 13) & 7, (id >> 8) & 15, (id >> 4) & 3, id & 15);
    }
    printf("\n");
  }
  // per (xcc, se, cu): which blocks and which slot sets
  std::map<unsigned, std::vector<int>> percu;
  for (int b = 0; b < B; ++b) {
    unsigned id = h[(b * 8) * 2], xcc = h[(b * 8) * 2 + 1] & 0xf;
    percu[(xcc << 16) | (id & 0xff00)].push_back(b);
  }
  int shown = 0;
  for (auto& kv : percu) {
    if (shown++ >= 6) break;
    printf("cu key %06x: blocks", kv.first);
    for (int b : kv.second) {
      unsigned slots = 0;
      for (int w = 0; w < 8; ++w) slots |= 1u << (h[(b * 8 + w) * 2] & 15);
      printf(" %d(slots %02x)", b, slots);
    }
    printf("\n");
  }
  printf("distinct CUs: %zu\n", percu.size());
  return 0;
}
