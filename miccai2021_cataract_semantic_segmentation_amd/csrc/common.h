// Shared helpers for the gfx950 kernels of libcatseg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "catseg.h"
#include "catseg_debug.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void catseg_set_error(const char* fmt, ...);

#define CS_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      catseg_set_error(__VA_ARGS__);          \
      return CATSEG_EINVAL;                   \
    }                                         \
  } while (0)

#define CS_LAUNCH_CHECK()                                                        \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      catseg_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,            \
                       hipGetErrorString(e__));                                  \
      return CATSEG_EHIP;                                                        \
    }                                                                            \
  } while (0)

static inline bool cs_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline size_t cs_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// 64-wide wavefront reductions (CDNA: wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
