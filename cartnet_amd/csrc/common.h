// Shared helpers for the gfx950 kernels of libcartnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/cartnet_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WAVE 64

void cartnet_set_error(const char* fmt, ...);

#define CN_CHECK(cond, ...)                \
  do {                                     \
    if (!(cond)) {                         \
      cartnet_set_error(__VA_ARGS__);      \
      return 1;                            \
    }                                      \
  } while (0)

#define CN_LAUNCH_CHECK(name)                                                    \
  do {                                                                           \
    hipError_t _e = hipGetLastError();                                           \
    if (_e != hipSuccess) {                                                      \
      cartnet_set_error("%s: launch failed: %s", name, hipGetErrorString(_e));   \
      return 2;                                                                  \
    }                                                                            \
  } while (0)

__device__ __forceinline__ float cn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float cn_silu(float x) { return x / (1.0f + expf(-x)); }
// d/dx [x * sigmoid(x)] = s * (1 + x * (1 - s))
__device__ __forceinline__ float cn_dsilu(float x) {
  float s = cn_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}

static inline int cn_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
