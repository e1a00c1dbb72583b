// Internal helpers shared by the gfx950 kernels of libmtam_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mtam_hip.h"

#define MTAM_WAVE 64

void mtam_set_error(const char *fmt, ...);

#define MTAM_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      mtam_set_error(__VA_ARGS__);           \
      return MTAM_E_ARG;                     \
    }                                        \
  } while (0)

#define MTAM_CHECK_LAUNCH(name)                                            \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      mtam_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return MTAM_E_LAUNCH;                                                \
    }                                                                      \
  } while (0)

static inline bool mtam_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- wave64 reductions (all 64 lanes get the result) -----------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
// reduce inside aligned groups of `width` lanes (width a power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int width) {
  for (int off = width >> 1; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
