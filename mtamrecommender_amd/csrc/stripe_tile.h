// Shared by the stripe kernels (csrc/seq_chain.hip) and the K/V roles co-scheduled with the GRU launches
// (csrc/tagru.hip): a 32 x 32 MFMA accumulator tile leaves through a wave-private LDS scratch as 16-byte row pieces.
#pragma once
#include <hip/hip_runtime.h>

namespace stripe {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 32;              // stripe height
constexpr int T_PITCH = 36;           // transposition scratch: 32 rows x (32 + 4) floats per wave
constexpr int X_PITCH = 128 + 4;      // floats per staged fp32 row of a [32, 128] stripe

// A 32 x 32 accumulator tile has its column on the lane (16 four-byte stores of 128-byte segments per lane):
// through a wave-private LDS scratch it leaves as 4 sixteen-byte stores per lane (8 rows x 128 B each).
__device__ __forceinline__ void store_tile(float *scratch, const f32x16 &v, float *out, long row0, int R, int ld,
                                           int col0, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 16; ++q) scratch[((q & 3) + 8 * (q >> 2) + 4 * h) * T_PITCH + r] = v[q];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 64 + lane, row = idx >> 3, c4 = idx & 7;
    const f32x4 t = *reinterpret_cast<const f32x4 *>(scratch + row * T_PITCH + 4 * c4);
    if (row0 + row < R) *reinterpret_cast<f32x4 *>(out + (row0 + row) * ld + col0 + 4 * c4) = t;
  }
}

}  // namespace stripe
