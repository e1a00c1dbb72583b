// fp32 products on the bf16 matrix cores: x = x1 + x2 + x3 (three bf16, exact to 24 bits), a.b from the six partial
// products of weight >= 2^-16 accumulated in fp32 (v_mfma_f32_32x32x16_bf16 runs at 16 x the rate of
// v_mfma_f32_32x32x2_f32, so six of them cost 6/16 of the fp32 instruction; the result matches an fp32 product to
// fp32 rounding, tests/test_kernels_gpu.py).  Shared by csrc/score32.hip and csrc/gemm_f32.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace split_bf16 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, __bf16 &a, __bf16 &b, __bf16 &c) {
  a = (__bf16)x;
  const float r1 = x - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}
struct Tri {
  bf16x8 t[3];
};
__device__ __forceinline__ Tri split8(const float (&x)[8]) {
  Tri o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 a, b, c;
    split3(x[j], a, b, c);
    o.t[0][j] = a; o.t[1][j] = b; o.t[2][j] = c;
  }
  return o;
}
// four fp32 values -> one 8-byte piece of each of the three images
__device__ __forceinline__ void split4(const float (&x)[4], bf16x4 (&q)[3]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __bf16 a, b, c;
    split3(x[j], a, b, c);
    q[0][j] = a; q[1][j] = b; q[2][j] = c;
  }
}
// the six products of weight >= 2^-16, smallest first
__device__ __forceinline__ f32x16 mfma6(const Tri &a, const Tri &b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[2], b.t[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b.t[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b.t[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[0], acc, 0, 0, 0);
  return acc;
}
// An 8-element operand fragment gathered DOWN a column of a row-major LDS image: two ds_read_b64_tr_b16, each a
// 4-row x 16-column block per 16-lane group (lane 4 q + p of the group addresses row q, 8-byte piece p; lane i
// receives column i, row q in element q).  EXEC must be all ones; addresses 8-byte aligned.
__device__ __forceinline__ bf16x8 lds_tr8(const unsigned char *lo, const unsigned char *hi) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lo));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(hi));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// ---- B-operand images of a row-major fp32 weight matrix W [K, N] (K a multiple of 8), kept beside the fp32 master:
// three bf16 terms (W = W1 + W2 + W3), each stored [K / 8][N][8] -- the 8 consecutive k a lane of
// v_mfma_f32_32x32x16_bf16 supplies for its column sit in one 16-byte piece, and the 32 lanes of a half wave read
// 512 contiguous bytes.  Term t starts at t * K * N.  Written by the optimizer launch that updates W
// (csrc/optim.hip) or by mtam_split_weight_images; read by csrc/seq_chain.hip.
__host__ __device__ inline size_t wimg_elems(int K, int N) { return (size_t)3 * K * N; }
__host__ __device__ inline size_t wimg_off(int k, int n, int N) { return ((size_t)(k >> 3) * N + n) * 8 + (k & 7); }
__device__ __forceinline__ void wimg_store(uint16_t *img, int K, int N, int k, int n, float w) {
  __bf16 a, b, c;
  split3(w, a, b, c);
  const size_t o = wimg_off(k, n, N), term = (size_t)K * N;
  img[o] = __builtin_bit_cast(uint16_t, a);
  img[term + o] = __builtin_bit_cast(uint16_t, b);
  img[2 * term + o] = __builtin_bit_cast(uint16_t, c);
}
// the six products, two independent accumulator chains interleaved (a dependent MFMA waits for its predecessor's
// last pass; two chains keep the pipe issuing back to back)
__device__ __forceinline__ void mfma6x2(const Tri &a0, const Tri &b0, f32x16 &c0, const Tri &a1, const Tri &b1,
                                        f32x16 &c1) {
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.t[2], b0.t[0], c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.t[2], b1.t[0], c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.t[1], b0.t[1], c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.t[1], b1.t[1], c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.t[0], b0.t[2], c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.t[0], b1.t[2], c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.t[1], b0.t[0], c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.t[1], b1.t[0], c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.t[0], b0.t[1], c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.t[0], b1.t[1], c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.t[0], b0.t[0], c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.t[0], b1.t[0], c1, 0, 0, 0);
}

}  // namespace split_bf16
