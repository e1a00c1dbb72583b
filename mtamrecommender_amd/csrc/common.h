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
// reduce inside aligned groups of `width` lanes (width a power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int width) {
  for (int off = width >> 1; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// The same sums on the VALU's data-parallel primitives: __shfl_xor compiles to ds_bpermute_b32 -- an address register
// and a trip through the LDS crossbar per step (132 of them in the decoder's forward kernel, 5.2 of its 11.6 us in the
// per-key score phase, tools/attn_lab.hip) -- while a DPP operand comes from another lane of the same 16-lane row at
// register speed: quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror, then the row / half-wave swaps below.  Every lane of the group ends with the group's sum; the order
// of the additions differs from group_sum's (fp32 rounding).  All lanes of the group must be active.
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// across the rows: v_permlane16_swap / v_permlane32_swap (gfx950) exchange the odd 16-lane rows of one register with
// the even rows of another (the upper half wave with the lower): fed the same value twice they return (row 0, row 0,
// row 2, row 2) and (row 1, row 1, row 3, row 3) -- the two operands of the xor-16 (xor-32) step, on the VALU
__device__ __forceinline__ void rows_xor16(unsigned int u, unsigned int &a, unsigned int &b) {
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ void halves_xor32(unsigned int u, unsigned int &a, unsigned int &b) {
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  a = r[0]; b = r[1];
}
template <int WIDTH> __device__ __forceinline__ float group_sum_dpp(float v) {
  static_assert(WIDTH == 2 || WIDTH == 4 || WIDTH == 8 || WIDTH == 16 || WIDTH == 32 || WIDTH == 64, "power of two");
  if (WIDTH >= 2) v += dpp_f32<0xB1>(v);
  if (WIDTH >= 4) v += dpp_f32<0x4E>(v);
  if (WIDTH >= 8) v += dpp_f32<0x141>(v);
  if (WIDTH >= 16) v += dpp_f32<0x140>(v);
  if (WIDTH >= 32) {
    unsigned int a, b;
    rows_xor16(__builtin_bit_cast(unsigned int, v), a, b);
    v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  }
  if (WIDTH >= 64) {
    unsigned int a, b;
    halves_xor32(__builtin_bit_cast(unsigned int, v), a, b);
    v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  }
  return v;
}
template <int WIDTH> __device__ __forceinline__ float group_max_dpp(float v) {
  if (WIDTH >= 2) v = fmaxf(v, dpp_f32<0xB1>(v));
  if (WIDTH >= 4) v = fmaxf(v, dpp_f32<0x4E>(v));
  if (WIDTH >= 8) v = fmaxf(v, dpp_f32<0x141>(v));
  if (WIDTH >= 16) v = fmaxf(v, dpp_f32<0x140>(v));
  if (WIDTH >= 32) {
    unsigned int a, b;
    rows_xor16(__builtin_bit_cast(unsigned int, v), a, b);
    v = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
  }
  if (WIDTH >= 64) {
    unsigned int a, b;
    halves_xor32(__builtin_bit_cast(unsigned int, v), a, b);
    v = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
  }
  return v;
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)u, CTRL, 0xf, 0xf, true);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(u >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned int la, lb, ha, hb;
    rows_xor16((unsigned int)u, la, lb);
    rows_xor16((unsigned int)(u >> 32), ha, hb);
    v = __builtin_bit_cast(double, (unsigned long long)la | ((unsigned long long)ha << 32)) +
        __builtin_bit_cast(double, (unsigned long long)lb | ((unsigned long long)hb << 32));
  }
  {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned int la, lb, ha, hb;
    halves_xor32((unsigned int)u, la, lb);
    halves_xor32((unsigned int)(u >> 32), ha, hb);
    v = __builtin_bit_cast(double, (unsigned long long)la | ((unsigned long long)ha << 32)) +
        __builtin_bit_cast(double, (unsigned long long)lb | ((unsigned long long)hb << 32));
  }
  return v;
}
// the whole wave (every lane active): all 64 lanes end with the result
__device__ __forceinline__ float wave_sum(float v) { return group_sum_dpp<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max_dpp<64>(v); }
// width: a power of two, wave-uniform
__device__ __forceinline__ float group_sum_fast(float v, int width) {
  switch (width) {
    case 32: return group_sum_dpp<32>(v);
    case 16: return group_sum_dpp<16>(v);
    case 8: return group_sum_dpp<8>(v);
    case 4: return group_sum_dpp<4>(v);
    case 2: return group_sum_dpp<2>(v);
    case 64: return group_sum_dpp<64>(v);
    default: return v;
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Hardware-transcendental forms for the serial GRU / attention steps, where the accurate libm
// sequences (about 25 instructions for a sigmoid, 35 for a tanh) sit on the critical path.
// v_exp_f32 and v_rcp_f32 are 1-ulp instructions; the results below stay within a few ulp
// (relative) of the libm values, which the parity tolerances (>= 2e-5) absorb.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp(float x) { return fast_exp2(1.4426950408889634f * x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + fast_exp2(-1.4426950408889634f * x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  const float ax = fabsf(x), x2 = x * x;
  // |x| < 0.3: odd Taylor polynomial (truncation < 2e-8 relative); else (1 - e^-2|x|) / (1 + e^-2|x|)
  const float p = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 0.021869488f, -0.053968254f), 0.13333334f), -0.33333334f), 1.0f);
  const float t = fast_exp2(-2.8853900817779268f * ax);
  const float r = copysignf((1.0f - t) * fast_rcp(1.0f + t), x);
  return ax < 0.3f ? p : r;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
