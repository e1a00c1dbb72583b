// The recurrent weights of the GRU in the order the forward kernel's lanes hold them (csrc/tagru.hip), kept beside the
// fp32 weights by the optimizer launch (csrc/optim.hip): 24 sixteen-byte pieces per thread of the 512, piece j of
// thread t at float4 index j * 512 + t.
#pragma once
#include <hip/hip_runtime.h>

namespace gru_image {

constexpr int UNITS = 128;
constexpr int FLOATS = 24 * 512 * 4;

// Where element (k, n) of the gate matrix Wg [128, 256] (which = 0) or of the candidate matrix Wc [128, 128]
// (which = 1) lies, in floats: lane kp = k / 16 of octet q2 = (n mod 128) / 2 holds rows 16 kp .. + 16 of its four gate
// columns (r and u of units 2 q2, 2 q2 + 1) and two candidate columns as f32x2 pairs of consecutive rows.
__host__ __device__ inline int pos(int which, int k, int n) {
  const int kp = k >> 4, kk = (k & 15) >> 1, par = k & 1;
  const int nn = n & (UNITS - 1), q2 = nn >> 1, odd = nn & 1;
  const int j = which ? 16 + kk : (n >= UNITS ? 8 + kk : kk);
  return ((j * 512 + 8 * q2 + kp) << 2) + 2 * odd + par;
}

}  // namespace gru_image
