// clip_by_global_norm + Adam: tf.clip_by_global_norm and
// tf.train.AdamOptimizer.apply_gradients as used by Model/base_model.py:290-297
// (beta1 0.9, beta2 0.999, eps 1e-8; TF 1.14 update formulas).
// All three kernels are HBM-bound streams: 16 B per lane, grid-stride-free
// (one 4096-float block per workgroup) so that V x D tables fill the chip.
#include "common.h"

namespace {

constexpr int NORM_BLOCK = 4096;   // floats per workgroup

__global__ __launch_bounds__(256) void sqnorm_kernel(const float *__restrict__ g, size_t n,
                                                     float *__restrict__ partial) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * NORM_BLOCK;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_BLOCK / 1024; ++i) {
    const size_t o = base + (size_t)(threadIdx.x + 256 * i) * 4;
    if (o + 3 < n) {
      const float4 v = *reinterpret_cast<const float4 *>(g + o);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (size_t q = o; q < n && q < o + 4; ++q) s += g[q] * g[q];
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void clip_scale_kernel(const float *__restrict__ partials, int n,
                                                         float clip, float *__restrict__ scale) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf((float)((red[0] + red[1]) + (red[2] + red[3])));
    // tf.clip_by_global_norm: t * clip_norm * min(1/global_norm, 1/clip_norm)
    scale[0] = clip * fminf(1.0f / norm, 1.0f / clip);
    scale[1] = norm;
  }
}

template <bool SPARSE_FORM>
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ m,
                                                   float *__restrict__ v, const float *__restrict__ g, size_t n,
                                                   const float *__restrict__ scale,
                                                   const float *__restrict__ hyper) {
  const float sc = scale[0];
  const float lr_t = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  auto step = [&](float &pp, float &mm, float &vv, float gg) {
    gg *= sc;
    if (SPARSE_FORM) {
      mm = mm * b1 + gg * omb1;
      vv = vv * b2 + (gg * gg) * omb2;
    } else {
      mm = mm + (gg - mm) * omb1;
      vv = vv + (gg * gg - vv) * omb2;
    }
    pp = pp - (lr_t * mm) / (sqrtf(vv) + eps);
  };
  const size_t base = (size_t)blockIdx.x * NORM_BLOCK;
#pragma unroll
  for (int i = 0; i < NORM_BLOCK / 1024; ++i) {
    const size_t o = base + (size_t)(threadIdx.x + 256 * i) * 4;
    if (o + 3 < n) {
      float4 pv = *reinterpret_cast<float4 *>(p + o), mv = *reinterpret_cast<float4 *>(m + o);
      float4 vv = *reinterpret_cast<float4 *>(v + o);
      const float4 gv = *reinterpret_cast<const float4 *>(g + o);
      step(pv.x, mv.x, vv.x, gv.x);
      step(pv.y, mv.y, vv.y, gv.y);
      step(pv.z, mv.z, vv.z, gv.z);
      step(pv.w, mv.w, vv.w, gv.w);
      *reinterpret_cast<float4 *>(p + o) = pv;
      *reinterpret_cast<float4 *>(m + o) = mv;
      *reinterpret_cast<float4 *>(v + o) = vv;
    } else {
      for (size_t q = o; q < n && q < o + 4; ++q) step(p[q], m[q], v[q], g[q]);
    }
  }
}

}  // namespace

extern "C" int mtam_sqnorm_blocks(size_t n) { return (int)((n + NORM_BLOCK - 1) / NORM_BLOCK); }

extern "C" int mtam_sqnorm_partial(const float *g, size_t n, float *partial, void *stream) {
  MTAM_CHECK_ARG(g && partial && n > 0, "sqnorm: bad arguments");
  MTAM_CHECK_ARG(mtam_aligned16(g), "sqnorm: gradient must be 16-byte aligned");
  hipLaunchKernelGGL(sqnorm_kernel, dim3(mtam_sqnorm_blocks(n)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g, n, partial);
  MTAM_CHECK_LAUNCH("sqnorm");
  return MTAM_OK;
}

extern "C" int mtam_clip_scale(const float *partials, int n_partials, float clip_norm, float *scale,
                               void *stream) {
  MTAM_CHECK_ARG(partials && scale && n_partials > 0 && clip_norm > 0.f, "clip_scale: bad arguments");
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), partials,
                     n_partials, clip_norm, scale);
  MTAM_CHECK_LAUNCH("clip_scale");
  return MTAM_OK;
}

extern "C" int mtam_adam(float *p, float *m, float *v, const float *g, size_t n, const float *scale,
                         const float *hyper, int sparse_form, void *stream) {
  MTAM_CHECK_ARG(p && m && v && g && scale && hyper && n > 0, "adam: bad arguments");
  MTAM_CHECK_ARG(mtam_aligned16(p) && mtam_aligned16(m) && mtam_aligned16(v) && mtam_aligned16(g),
                 "adam: buffers must be 16-byte aligned");
  dim3 grid(mtam_sqnorm_blocks(n));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (sparse_form) hipLaunchKernelGGL(adam_kernel<true>, grid, dim3(256), 0, s, p, m, v, g, n, scale, hyper);
  else hipLaunchKernelGGL(adam_kernel<false>, grid, dim3(256), 0, s, p, m, v, g, n, scale, hyper);
  MTAM_CHECK_LAUNCH("adam");
  return MTAM_OK;
}
