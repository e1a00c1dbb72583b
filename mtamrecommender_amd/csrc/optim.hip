// clip_by_global_norm + Adam: tf.clip_by_global_norm and
// tf.train.AdamOptimizer.apply_gradients as used by Model/base_model.py:290-297
// (beta1 0.9, beta2 0.999, eps 1e-8; TF 1.14 update formulas).
// All three kernels are HBM-bound streams: 16 B per lane, grid-stride-free
// (one 4096-float block per workgroup) so that V x D tables fill the chip.
#include "common.h"
#include "split_bf16.h"
#include "gru_image.h"
#include <stdlib.h>

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
                                                         float clip, float *__restrict__ scale,
                                                         const float *__restrict__ lr,
                                                         float *__restrict__ adam_state) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  s = wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf((float)((red[0] + red[1]) + (red[2] + red[3])));
    // tf.clip_by_global_norm: t * clip_norm * min(1/global_norm, 1/clip_norm)
    scale[0] = clip * fminf(1.0f / norm, 1.0f / clip);
    scale[1] = norm;
    if (adam_state) {
      // AdamOptimizer._prepare / _finish [TF1.14]: lr_t from the CURRENT beta powers, then advance them.
      const float b1 = adam_state[1], b2 = adam_state[2];
      const float b1p = adam_state[4], b2p = adam_state[5];
      adam_state[0] = lr[0] * sqrtf(1.0f - b2p) / (1.0f - b1p);
      adam_state[4] = b1p * b1;
      adam_state[5] = b2p * b2;
    }
  }
}

// Data-parallel form with a row-sharded item table (data_parallel.py): every rank sums the squares of what it
// OWNS into one double, the doubles are all-reduced, and the clip scale comes from the total.
__global__ __launch_bounds__(256) void partials_sum_kernel(const float *__restrict__ partials, int n, float weight,
                                                           double *__restrict__ out, int accumulate) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  s = wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (double)weight * ((red[0] + red[1]) + (red[2] + red[3]));
    out[0] = accumulate ? out[0] + t : t;
  }
}

__global__ void clip_scale_sq_kernel(const double *__restrict__ sq, int n, float clip, float *__restrict__ scale,
                                     const float *__restrict__ lr, float *__restrict__ adam_state) {
  if (threadIdx.x != 0) return;
  double t = 0.0;
  for (int i = 0; i < n; ++i) t += sq[i];
  const float norm = sqrtf((float)t);
  scale[0] = clip * fminf(1.0f / norm, 1.0f / clip);
  scale[1] = norm;
  if (adam_state) {
    const float b1 = adam_state[1], b2 = adam_state[2];
    const float b1p = adam_state[4], b2p = adam_state[5];
    adam_state[0] = lr[0] * sqrtf(1.0f - b2p) / (1.0f - b1p);
    adam_state[4] = b1p * b1;
    adam_state[5] = b2p * b2;
  }
}

// sqnorm + clip_scale in one launch: every workgroup writes its partial; the last one to arrive
// (agent-scope ticket; partials moved by sc1 stores / sc1 loads) sums ALL partials and does what
// clip_scale_kernel does.  `ticket` must be zero on entry and is left zero.
__global__ __launch_bounds__(256) void sqnorm_clip_kernel(const float *__restrict__ g, size_t n,
                                                          float *__restrict__ partials, int offset,
                                                          int n_total, float clip, float *__restrict__ scale,
                                                          const float *__restrict__ lr,
                                                          float *__restrict__ adam_state, unsigned int *ticket,
                                                          const float *__restrict__ l2_partial, int n_l2,
                                                          const float *__restrict__ ce, int B, float reg,
                                                          float ce_scale, float *__restrict__ loss) {
  __shared__ float red[4];
  __shared__ double dred[4];
  __shared__ int s_last;
  const size_t base = (size_t)blockIdx.x * NORM_BLOCK;
  // The loss inputs (the lookups' L2 partials, the per-row cross entropies) were written by earlier kernels of the
  // step: when they fit one batch of loads EVERY workgroup fetches them here, beside its gradient block, so that
  // the last one to arrive does not pay two more round trips after the ticket (a few KB of L2 reads per workgroup)
  const bool early_loss = loss && n_l2 <= 256 * 8 && B <= 256;
  float pre_l2[8], pre_ce = 0.f;
  if (early_loss) {
#pragma unroll
    for (int q = 0; q < 8; ++q) pre_l2[q] = l2_partial[min((int)threadIdx.x + 256 * q, max(n_l2 - 1, 0))];
    pre_ce = ce[min((int)threadIdx.x, B - 1)];
  }
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
  if (threadIdx.x == 0) {
    // write-through (sc1) store + drain + ticket; the last workgroup reads with sc1 loads (no fences)
    __hip_atomic_store(partials + offset + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int mine = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (mine == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // The last workgroup sums every partial (and, for the loss, the gather's L2 partials and the per-row
  // cross entropies).  Loads go out in batches of 8 per thread, unconditionally (clamped index, masked
  // value): a plain `for (i = tid; i < n; i += 256) t += load(i)` waits out one round trip per iteration
  // (19 of them for the 4,832 L2 partials: most of this kernel's former 13 us).
  auto sum_sc1 = [&](const float *src, int n) {
    double acc = 0.0;
    for (int base = 0; base < n; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        v[q] = __hip_atomic_load(src + min(base + (int)threadIdx.x + 256 * q, n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += (base + (int)threadIdx.x + 256 * q < n) ? (double)v[q] : 0.0;
    }
    return acc;
  };
  auto sum_plain = [&](const float *src, int n) {
    double acc = 0.0;
    for (int base = 0; base < n; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = src[min(base + (int)threadIdx.x + 256 * q, n - 1)];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += (base + (int)threadIdx.x + 256 * q < n) ? (double)v[q] : 0.0;
    }
    return acc;
  };
  double t = sum_sc1(partials, n_total);
  // the loss inputs were written by earlier kernels of the step: issue their loads before the reduction
  // of `t` needs a barrier
  double a_l2 = 0.0, c_ce = 0.0;
  if (early_loss) {
#pragma unroll
    for (int q = 0; q < 8; ++q) a_l2 += ((int)threadIdx.x + 256 * q < n_l2) ? (double)pre_l2[q] : 0.0;
    c_ce = ((int)threadIdx.x < B) ? (double)pre_ce : 0.0;
  } else if (loss) {
    a_l2 = sum_plain(l2_partial, n_l2);
    c_ce = sum_plain(ce, B);
  }
  t = wave_sum_f64(t);
  if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf((float)((dred[0] + dred[1]) + (dred[2] + dred[3])));
    scale[0] = clip * fminf(1.0f / norm, 1.0f / clip);
    scale[1] = norm;
    if (adam_state) {
      const float b1 = adam_state[1], b2 = adam_state[2];
      const float b1p = adam_state[4], b2p = adam_state[5];
      adam_state[0] = lr[0] * sqrtf(1.0f - b2p) / (1.0f - b1p);
      adam_state[4] = b1p * b1;
      adam_state[5] = b2p * b2;
    }
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (loss) {
    // step epilogue: the reported loss (Model/base_model.py:322-326) from the per-row cross entropies
    // and the gather's L2 partials -- both written by earlier kernels of the step
    double a = a_l2, c = c_ce;
    a = wave_sum_f64(a);
    c = wave_sum_f64(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = a;
    __syncthreads();
    const double l2 = 0.5 * ((dred[0] + dred[1]) + (dred[2] + dred[3]));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double ces = (dred[0] + dred[1]) + (dred[2] + dred[3]);
      loss[0] = (float)((double)reg * l2 + (double)ce_scale * ces);
      loss[1] = (float)l2;
      loss[2] = (float)((double)ce_scale * ces);
    }
  }
}

// sum of n floats in float64, by one 256-thread workgroup, the same order in every workgroup that runs it: per thread
// in batches of eight loads (clamped index, masked value), wave shuffle, four waves through LDS
__device__ __forceinline__ double block_sum_f64(const float *__restrict__ src, int n, double (&dred)[4]) {
  double acc = 0.0;
  // (24 loads in flight per thread and trip -- 6,144 values, the whole list at ml-1m sizes -- added in index order)
  for (int base = 0; base < n; base += 256 * 24) {
    float v[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) v[q] = src[min(base + (int)threadIdx.x + 256 * q, n - 1)];
#pragma unroll
    for (int q = 0; q < 24; ++q) acc += (base + (int)threadIdx.x + 256 * q < n) ? (double)v[q] : 0.0;
  }
  acc = wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = acc;
  __syncthreads();
  const double t = (dred[0] + dred[1]) + (dred[2] + dred[3]);
  __syncthreads();
  return t;
}

// The same step WITHOUT the ticket: workgroups 0 .. nb - 1 write their partial sums of squares and nothing else; one
// more workgroup advances the Adam state and reduces the reported loss (neither depends on the norm).  The norm itself
// is formed by the optimizer launch -- every one of its workgroups sums the partials (adam_kernel, norm_partials).
// 2.6 us against 7.0 for sqnorm_clip_kernel at ml-1m sizes (61 blocks): the arrival ticket is a same-address atomic per
// workgroup (~45 ns each, serialised) and the last workgroup's reduction a second dependent pass.
__global__ __launch_bounds__(256) void sqnorm_state_loss_kernel(const float *__restrict__ g, size_t n,
                                                                float *__restrict__ partials, int offset,
                                                                const float *__restrict__ lr,
                                                                float *__restrict__ adam_state,
                                                                const float *__restrict__ l2_partial, int n_l2,
                                                                const float *__restrict__ ce, int B, float reg,
                                                                float ce_scale, float *__restrict__ loss) {
  __shared__ float red[4];
  __shared__ double dred[4];
  if (blockIdx.x == gridDim.x - 1) {
    if (adam_state && threadIdx.x == 0) {
      const float b1 = adam_state[1], b2 = adam_state[2];
      const float b1p = adam_state[4], b2p = adam_state[5];
      adam_state[0] = lr[0] * sqrtf(1.0f - b2p) / (1.0f - b1p);
      adam_state[4] = b1p * b1;
      adam_state[5] = b2p * b2;
    }
    if (loss) {
      // the reported loss (Model/base_model.py:322-326) from the per-row cross entropies and the lookups' L2 partials
      const double l2 = 0.5 * block_sum_f64(l2_partial, n_l2, dred);
      const double ces = block_sum_f64(ce, B, dred);
      if (threadIdx.x == 0) {
        loss[0] = (float)((double)reg * l2 + (double)ce_scale * ces);
        loss[1] = (float)l2;
        loss[2] = (float)((double)ce_scale * ces);
      }
    }
    return;
  }
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
  if (threadIdx.x == 0) partials[offset + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// COPY: elements at or after copy_begin (the item table) are also written, rounded to nearest even, to the
// bf16 scoring copy (csrc/score16.hip) -- 2 more bytes per element instead of a separate 6-byte pass.
typedef __bf16 adam_bf16x4 __attribute__((ext_vector_type(4)));
// Weight matrices whose bf16 operand images (csrc/split_bf16.h, wimg_off) are re-written by the launch that updates
// them: the forward's fused projection kernel and the backward's stripe kernel (csrc/seq_chain.hip) read their B
// operands already split and laid out as MFMA fragments, and no per-step prepare launch exists.  By value: at most
// MTAM_MAX_WEIGHT_IMAGES matrices.
// The elements of such a matrix are updated by EXTRA workgroups appended to the grid (blockIdx >= n_linear), one
// thread per 4 rows x 4 columns: its reads and writes of p / m / v / g are 16-byte pieces coalesced over the lanes,
// it splits the updated values and stores them as 8-byte half pieces of the forward's images (4 of the 8
// consecutive k of a column) and 8-byte half pieces of the transpose's images.  The linear workgroups skip those
// elements.  Measured (launch at ml-1m sizes, 8.4 us without images): the linear threads writing their four
// elements' twelve 2-byte image entries themselves -- 48 scattered stores per thread in 28 of the 330 workgroups --
// 13.6 us; 8 x 4 elements per image thread 11.2; 4 x 4 (as many loads as a linear thread) 9.8-10.3; the image
// workgroups at the HEAD of the grid instead of its tail 11.3.
struct AdamImages {
  int n;
  unsigned n_linear;                          // workgroups of the linear sweep
  unsigned first_block[MTAM_MAX_WEIGHT_IMAGES + 1];   // image workgroups of matrix j: [first_block[j], first_block[j + 1])
  size_t begin[MTAM_MAX_WEIGHT_IMAGES], end[MTAM_MAX_WEIGHT_IMAGES];
  int K[MTAM_MAX_WEIGHT_IMAGES], N[MTAM_MAX_WEIGHT_IMAGES];
  uint16_t *img[MTAM_MAX_WEIGHT_IMAGES], *img_r[MTAM_MAX_WEIGHT_IMAGES];
  int gru_which[MTAM_MAX_WEIGHT_IMAGES];      // 1 / 2: the GRU forward's fp32 register-order image (csrc/tagru.hip)
  // mtam_adam_images_clip: non-NULL = EVERY workgroup derives the clip scale itself from these partial sums of
  // squares (written by earlier launches of the step) instead of reading scale[0]; workgroup 0 publishes it
  const float *norm_partials;
  int n_norm;
  float clip;
  float *scale_out;
  // mtam_adam_images_clip_feed: non-NULL = ONE more workgroup at the very end of the grid hands the NEXT step its
  // feed -- slot (cursor % feed_slots) of a ring of packed feed arenas already in HBM goes into the arena the step's
  // kernels read, and the cursor advances.  Nothing in this launch reads the arena (the learning rate was folded into
  // hyper[0] by the state advance of an earlier launch), so the 128 KB move in the shadow of the update instead of
  // being a copy in front of the next step's first kernel (4.9 us of a 226 us step at ml-1m sizes).
  const int32_t *feed_ring;
  int32_t *feed_arena;
  unsigned *feed_cursor;
  unsigned feed_slots, feed_words;            // words per slot (= the slot pitch)
};

template <bool COPY, bool NT>
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ m,
                                                   float *__restrict__ v, const float *__restrict__ g, size_t n,
                                                   const float *__restrict__ scale,
                                                   const float *__restrict__ hyper, size_t sparse_begin,
                                                   uint16_t *__restrict__ copy16, size_t copy_begin,
                                                   AdamImages wi) {
  __shared__ double dred[4];
  float sc = 0.f;
  // called by every thread of the workgroup AFTER its role has issued its own loads of p / m / v / g: the norm's
  // partials are summed while those are in flight
  auto clip_scale = [&]() {
    if (wi.norm_partials) {                    // launch-uniform
      const float norm = sqrtf((float)block_sum_f64(wi.norm_partials, wi.n_norm, dred));
      sc = wi.clip * fminf(1.0f / norm, 1.0f / wi.clip);
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        wi.scale_out[0] = sc;
        wi.scale_out[1] = norm;
      }
    } else {
      sc = scale[0];
    }
  };
  const float lr_t = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  const size_t base = (size_t)blockIdx.x * NORM_BLOCK;
  // elements at or after sparse_begin are table rows (IndexedSlices update form); a block never
  // straddles the boundary because sparse_begin is a multiple of the block size or 0 / n.
  const bool sparse_form = base >= sparse_begin;
  // Every product-sum is an EXPLICIT fma and contraction is off for the rest: the same element must come out bit
  // for bit whichever code path of this kernel updates it (linear sweep, image role), so nothing is left to the
  // compiler's contraction choices (__fmul_rn is a plain product in HIP: it does not stop a later fusion).
  auto step_form = [&](float &pp, float &mm, float &vv, float gg, bool sparse) {
#pragma clang fp contract(off)
    gg = gg * sc;
    if (sparse) {
      const float t1 = gg * omb1, t2 = (gg * gg) * omb2;
      mm = fmaf(mm, b1, t1);
      vv = fmaf(vv, b2, t2);
    } else {
      const float d1 = gg - mm;
      mm = fmaf(d1, omb1, mm);
      const float d2 = fmaf(gg, gg, -vv);
      vv = fmaf(d2, omb2, vv);
    }
    const float num = lr_t * mm, den = sqrtf(vv) + eps;
    pp = pp - num / den;
  };
  auto step = [&](float &pp, float &mm, float &vv, float gg) { step_form(pp, mm, vv, gg, sparse_form); };
  if (wi.feed_ring && blockIdx.x == gridDim.x - 1) {
    // ---- feed role: the next step's packed feed, ring slot -> arena (16-byte pieces, 16 in flight per thread)
    const unsigned c = *wi.feed_cursor;
    const int32_t *src = wi.feed_ring + (size_t)(c % wi.feed_slots) * wi.feed_words;
    typedef int feed_i4 __attribute__((ext_vector_type(4)));
    const unsigned n4 = wi.feed_words / 4;
    for (unsigned b0 = 0; b0 < n4; b0 += 256 * 16) {
      feed_i4 q[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const unsigned i = min(b0 + u * 256 + threadIdx.x, n4 - 1);
        q[u] = reinterpret_cast<const feed_i4 *>(src)[i];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const unsigned i = b0 + u * 256 + threadIdx.x;
        if (i < n4) reinterpret_cast<feed_i4 *>(wi.feed_arena)[i] = q[u];
      }
    }
    for (unsigned i = 4 * n4 + threadIdx.x; i < wi.feed_words; i += 256) wi.feed_arena[i] = src[i];
    __syncthreads();                           // every thread has read the cursor
    if (threadIdx.x == 0) *wi.feed_cursor = c + 1;
    return;
  }
  if (wi.n && blockIdx.x >= wi.n_linear) {
    // ---- image role (dense variables: the non-sparse update form): 4 rows x 4 columns of the matrix per thread
    typedef float adam_f4 __attribute__((ext_vector_type(4)));
    int j = 0;
    while (j + 1 < wi.n && blockIdx.x >= wi.n_linear + wi.first_block[j + 1]) ++j;
    const int K = wi.K[j], N = wi.N[j], n4 = N / 4;
    // (32-bit: a 64-bit division by a run-time value is over a hundred instructions)
    const unsigned unit = (blockIdx.x - wi.n_linear - wi.first_block[j]) * 256u + threadIdx.x;
    const bool live = unit < (unsigned)(K / 4) * (unsigned)n4;
    const int g4 = (int)(unit / (unsigned)n4), col = (int)(unit - (unsigned)g4 * (unsigned)n4) * 4;   // rows 4 g4 .. 4 g4 + 4
    const size_t e0 = wi.begin[j] + (size_t)g4 * 4 * N + col;
    adam_f4 pe[4], me[4], ve[4], ge[4];
    if (live) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t e = e0 + (size_t)i * N;
        pe[i] = *reinterpret_cast<const adam_f4 *>(p + e); me[i] = *reinterpret_cast<const adam_f4 *>(m + e);
        ve[i] = *reinterpret_cast<const adam_f4 *>(v + e); ge[i] = *reinterpret_cast<const adam_f4 *>(g + e);
      }
    }
    clip_scale();
    if (!live) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float a = pe[i][c], b = me[i][c], d = ve[i][c];
        step_form(a, b, d, ge[i][c], false);
        pe[i][c] = a; me[i][c] = b; ve[i][c] = d;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t e = e0 + (size_t)i * N;
      *reinterpret_cast<adam_f4 *>(p + e) = pe[i]; *reinterpret_cast<adam_f4 *>(m + e) = me[i];
      *reinterpret_cast<adam_f4 *>(v + e) = ve[i];
    }
    if (wi.gru_which[j]) {
      // the GRU forward's image: the same fp32 values in the order its lanes hold them (mtam_gru_weight_image_pos)
      float *gi = reinterpret_cast<float *>(wi.img[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) gi[gru_image::pos(wi.gru_which[j] - 1, g4 * 4 + i, col + c)] = pe[i][c];
      return;
    }
    const size_t term = (size_t)K * N;
    // forward images: half of a 16-byte piece (4 of its 8 consecutive k) per column and term
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x4[4] = {pe[0][c], pe[1][c], pe[2][c], pe[3][c]};
      split_bf16::bf16x4 q[3];
      split_bf16::split4(x4, q);
      uint16_t *dst = wi.img[j] + ((size_t)(g4 >> 1) * N + col + c) * 8 + 4 * (g4 & 1);
#pragma unroll
      for (int t = 0; t < 3; ++t) *reinterpret_cast<split_bf16::bf16x4 *>(dst + t * term) = q[t];
    }
    // images of the transpose (the backward's): element W[k][n] at ((n >> 3) K + k) 8 + (n & 7) -- 8 bytes per row
    // and term (4 consecutive n of one k)
    if (wi.img_r[j]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x4[4] = {pe[i][0], pe[i][1], pe[i][2], pe[i][3]};
        split_bf16::bf16x4 q[3];
        split_bf16::split4(x4, q);
        uint16_t *dst = wi.img_r[j] + ((size_t)(col >> 3) * K + g4 * 4 + i) * 8 + (col & 7);
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<split_bf16::bf16x4 *>(dst + t * term) = q[t];
      }
    }
    return;
  }
  // ---- linear sweep: all of the workgroup's 16-byte groups are fetched first, then the scale, then the updates
  typedef float adam_f4 __attribute__((ext_vector_type(4)));
  constexpr int NG = NORM_BLOCK / 1024;
  adam_f4 pq[NG], mq[NG], vq[NG], gq[NG];
  bool skip[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const size_t o = base + (size_t)(threadIdx.x + 256 * i) * 4;
    bool in_matrix = false;                    // (begin and size are multiples of 4: a group is inside or outside)
    for (int j = 0; j < wi.n; ++j) in_matrix |= o >= wi.begin[j] && o < wi.end[j];
    skip[i] = in_matrix;
    if (!in_matrix && o + 3 < n) {
      if (NT) {        // streamed once per step: nothing of the seven streams is worth a cache line
        pq[i] = __builtin_nontemporal_load(reinterpret_cast<const adam_f4 *>(p + o));
        mq[i] = __builtin_nontemporal_load(reinterpret_cast<const adam_f4 *>(m + o));
        vq[i] = __builtin_nontemporal_load(reinterpret_cast<const adam_f4 *>(v + o));
        gq[i] = __builtin_nontemporal_load(reinterpret_cast<const adam_f4 *>(g + o));
      } else {
        pq[i] = *reinterpret_cast<const adam_f4 *>(p + o); mq[i] = *reinterpret_cast<const adam_f4 *>(m + o);
        vq[i] = *reinterpret_cast<const adam_f4 *>(v + o); gq[i] = *reinterpret_cast<const adam_f4 *>(g + o);
      }
    }
  }
  clip_scale();
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const size_t o = base + (size_t)(threadIdx.x + 256 * i) * 4;
    if (skip[i]) continue;
    if (o + 3 < n) {
      adam_f4 pv = pq[i], mv = mq[i], vv = vq[i];
      const adam_f4 gv = gq[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pe = pv[k], me = mv[k], ve = vv[k];
        step(pe, me, ve, gv[k]);
        pv[k] = pe; mv[k] = me; vv[k] = ve;
      }
      if (NT) {
        __builtin_nontemporal_store(pv, reinterpret_cast<adam_f4 *>(p + o));
        __builtin_nontemporal_store(mv, reinterpret_cast<adam_f4 *>(m + o));
        __builtin_nontemporal_store(vv, reinterpret_cast<adam_f4 *>(v + o));
      } else {
        *reinterpret_cast<adam_f4 *>(p + o) = pv;
        *reinterpret_cast<adam_f4 *>(m + o) = mv;
        *reinterpret_cast<adam_f4 *>(v + o) = vv;
      }
      if (COPY && o >= copy_begin) {
        adam_bf16x4 c;
        c.x = (__bf16)pv[0]; c.y = (__bf16)pv[1]; c.z = (__bf16)pv[2]; c.w = (__bf16)pv[3];
        *reinterpret_cast<adam_bf16x4 *>(copy16 + (o - copy_begin)) = c;
      }
    } else {
      for (size_t q = o; q < n && q < o + 4; ++q) {
        step(p[q], m[q], v[q], g[q]);
        if (COPY && q >= copy_begin) *reinterpret_cast<__bf16 *>(copy16 + (q - copy_begin)) = (__bf16)p[q];
      }
    }
  }
}

// The reference's other optimizer choices (Model/base_model.py:71-80): GradientDescent, Adadelta
// (rho 0.95, eps 1e-8) and RMSProp (decay 0.9, momentum 0, eps 1e-10) with the TF 1.14 update
// formulas [TF1.14 training_ops].  Tables receive IndexedSlices gradients, so TF applies the
// sparse kernels to the rows that occur in the batch only: slots of rows outside the batch do not
// decay.  Elements in [sparse_begin, rowskip_end) are such tables (category, position, user): a
// 128-float row whose summed gradient is entirely zero is left untouched.  The item table (at and
// after rowskip_end) has every row in its IndexedSlices (dense scoring gradient).
enum { OPT_SGD = 0, OPT_ADADELTA = 1, OPT_RMSPROP = 2 };

template <int KIND>
__global__ __launch_bounds__(256) void opt_kernel(float *__restrict__ p, float *__restrict__ s1,
                                                  float *__restrict__ s2, const float *__restrict__ g, size_t n,
                                                  const float *__restrict__ scale, const float *__restrict__ lr_ptr,
                                                  size_t sparse_begin, size_t rowskip_end) {
  const float sc = scale[0];
  const float lr = lr_ptr[0];
  const size_t base = (size_t)blockIdx.x * NORM_BLOCK;
  const bool sparse_form = base >= sparse_begin;
  const bool rowskip = sparse_form && base < rowskip_end;
  auto step = [&](float &pp, float &a, float &b, float gg) {
    gg *= sc;
    if (KIND == OPT_SGD) {
      pp -= lr * gg;
    } else if (KIND == OPT_ADADELTA) {
      const float rho = 0.95f, eps = 1e-8f;
      a = a * rho + (gg * gg) * (1.0f - rho);
      const float upd = sqrtf(b + eps) * (1.0f / sqrtf(a + eps)) * gg;
      pp -= upd * lr;
      b = b * rho + (upd * upd) * (1.0f - rho);
    } else {
      const float rho = 0.9f, eps = 1e-10f;
      if (sparse_form) a = a * rho + (gg * gg) * (1.0f - rho);
      else a = a + (gg * gg - a) * (1.0f - rho);
      b = (gg * lr) / sqrtf(a + eps);          // momentum 0: mom = mom * 0 + lr * g / sqrt(ms + eps)
      pp -= b;
    }
  };
#pragma unroll
  for (int i = 0; i < NORM_BLOCK / 1024; ++i) {
    const size_t o = base + (size_t)(threadIdx.x + 256 * i) * 4;
    if (o + 3 < n) {
      const float4 gv = *reinterpret_cast<const float4 *>(g + o);
      if (rowskip) {       // 32 consecutive lanes hold one 128-float row
        const bool nz = gv.x != 0.f || gv.y != 0.f || gv.z != 0.f || gv.w != 0.f;
        const unsigned long long m = __ballot(nz);
        const unsigned half = (threadIdx.x & 32) ? (unsigned)(m >> 32) : (unsigned)m;
        if (half == 0u) continue;
      }
      float4 pv = *reinterpret_cast<float4 *>(p + o);
      float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv = av;
      if (KIND != OPT_SGD) {
        av = *reinterpret_cast<float4 *>(s1 + o);
        bv = *reinterpret_cast<float4 *>(s2 + o);
      }
      step(pv.x, av.x, bv.x, gv.x);
      step(pv.y, av.y, bv.y, gv.y);
      step(pv.z, av.z, bv.z, gv.z);
      step(pv.w, av.w, bv.w, gv.w);
      *reinterpret_cast<float4 *>(p + o) = pv;
      if (KIND != OPT_SGD) {
        *reinterpret_cast<float4 *>(s1 + o) = av;
        *reinterpret_cast<float4 *>(s2 + o) = bv;
      }
    } else {
      for (size_t q = o; q < n && q < o + 4; ++q) {
        float a = KIND != OPT_SGD ? s1[q] : 0.f, b = KIND != OPT_SGD ? s2[q] : 0.f;
        step(p[q], a, b, g[q]);
        if (KIND != OPT_SGD) { s1[q] = a; s2[q] = b; }
      }
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
                               const float *lr, float *adam_state, void *stream) {
  MTAM_CHECK_ARG(partials && scale && n_partials > 0 && clip_norm > 0.f, "clip_scale: bad arguments");
  MTAM_CHECK_ARG((lr == nullptr) == (adam_state == nullptr), "clip_scale: lr and adam_state go together");
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), partials,
                     n_partials, clip_norm, scale, lr, adam_state);
  MTAM_CHECK_LAUNCH("clip_scale");
  return MTAM_OK;
}

extern "C" int mtam_partials_sum(const float *partials, int n, float weight, double *out, int accumulate,
                                 void *stream) {
  MTAM_CHECK_ARG(partials && out && n > 0, "partials_sum: bad arguments");
  hipLaunchKernelGGL(partials_sum_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), partials, n, weight,
                     out, accumulate);
  MTAM_CHECK_LAUNCH("partials_sum");
  return MTAM_OK;
}

extern "C" int mtam_clip_scale_sq(const double *sq_total, int n, float clip_norm, float *scale, const float *lr,
                                  float *adam_state, void *stream) {
  MTAM_CHECK_ARG(sq_total && scale && n > 0 && n <= 64 && clip_norm > 0.f, "clip_scale_sq: bad arguments");
  MTAM_CHECK_ARG((lr == nullptr) == (adam_state == nullptr), "clip_scale_sq: lr and adam_state go together");
  hipLaunchKernelGGL(clip_scale_sq_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), sq_total, n,
                     clip_norm, scale, lr, adam_state);
  MTAM_CHECK_LAUNCH("clip_scale_sq");
  return MTAM_OK;
}

extern "C" int mtam_sqnorm_clip_scale(const float *g, size_t n, float *partials, int offset, int n_total,
                                      float clip_norm, float *scale, const float *lr, float *adam_state,
                                      unsigned int *ticket, const float *l2_partial, int n_l2,
                                      const float *ce, int B, float reg, float ce_scale, float *loss,
                                      void *stream) {
  MTAM_CHECK_ARG(g && partials && scale && ticket && n > 0 && clip_norm > 0.f, "sqnorm_clip_scale: bad arguments");
  MTAM_CHECK_ARG(!loss || (l2_partial && ce && B > 0 && n_l2 >= 0), "sqnorm_clip_scale: loss inputs missing");
  MTAM_CHECK_ARG(mtam_aligned16(g), "sqnorm_clip_scale: gradient must be 16-byte aligned");
  MTAM_CHECK_ARG((lr == nullptr) == (adam_state == nullptr), "sqnorm_clip_scale: lr and adam_state go together");
  const int blocks = mtam_sqnorm_blocks(n);
  MTAM_CHECK_ARG(offset >= 0 && n_total >= offset + blocks, "sqnorm_clip_scale: partial layout");
  hipLaunchKernelGGL(sqnorm_clip_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), g, n,
                     partials, offset, n_total, clip_norm, scale, lr, adam_state, ticket, l2_partial, n_l2, ce, B, reg,
                     ce_scale, loss);
  MTAM_CHECK_LAUNCH("sqnorm_clip_scale");
  return MTAM_OK;
}

// non-temporal accesses once the seven streams exceed the caches (MTAM_ADAM_NT_MIN_BYTES, default 256 MiB of
// parameters): 5.62 against 5.91 ms on the 10 M-row step (6.4 TB/s over the 35.8 GB)
static size_t adam_nt_min_bytes() {
  static const size_t v = [] {
    const char *e = getenv("MTAM_ADAM_NT_MIN_BYTES");
    return e ? (size_t)atoll(e) : ((size_t)1 << 28);
  }();
  return v;
}

extern "C" int mtam_adam_block(void) { return NORM_BLOCK; }

extern "C" int mtam_adam(float *p, float *m, float *v, const float *g, size_t n, const float *scale,
                         const float *hyper, size_t sparse_begin, void *stream) {
  MTAM_CHECK_ARG(p && m && v && g && scale && hyper && n > 0, "adam: bad arguments");
  MTAM_CHECK_ARG(mtam_aligned16(p) && mtam_aligned16(m) && mtam_aligned16(v) && mtam_aligned16(g),
                 "adam: buffers must be 16-byte aligned");
  MTAM_CHECK_ARG(sparse_begin >= n || sparse_begin % NORM_BLOCK == 0,
                 "adam: sparse_begin must be a multiple of %d (or >= n)", NORM_BLOCK);
  dim3 grid(mtam_sqnorm_blocks(n));
  if (n * 4 >= adam_nt_min_bytes())
    hipLaunchKernelGGL((adam_kernel<false, true>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), p, m, v, g, n,
                       scale, hyper, sparse_begin, static_cast<uint16_t *>(nullptr), n, AdamImages{});
  else
    hipLaunchKernelGGL((adam_kernel<false, false>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), p, m, v, g, n,
                       scale, hyper, sparse_begin, static_cast<uint16_t *>(nullptr), n, AdamImages{});
  MTAM_CHECK_LAUNCH("adam");
  return MTAM_OK;
}

extern "C" int mtam_adam_bf16copy(float *p, float *m, float *v, const float *g, size_t n, const float *scale,
                                  const float *hyper, size_t sparse_begin, uint16_t *copy16, size_t copy_begin,
                                  void *stream) {
  MTAM_CHECK_ARG(p && m && v && g && scale && hyper && copy16 && n > 0, "adam_bf16copy: bad arguments");
  MTAM_CHECK_ARG(mtam_aligned16(p) && mtam_aligned16(m) && mtam_aligned16(v) && mtam_aligned16(g) &&
                     (reinterpret_cast<uintptr_t>(copy16) & 7u) == 0,
                 "adam_bf16copy: buffers must be 16-byte aligned (the copy 8-byte)");
  MTAM_CHECK_ARG(sparse_begin >= n || sparse_begin % NORM_BLOCK == 0,
                 "adam_bf16copy: sparse_begin must be a multiple of %d (or >= n)", NORM_BLOCK);
  MTAM_CHECK_ARG(copy_begin <= n && copy_begin % 4 == 0, "adam_bf16copy: copy_begin must be a multiple of 4");
  dim3 grid(mtam_sqnorm_blocks(n));
  if (n * 4 >= adam_nt_min_bytes())
    hipLaunchKernelGGL((adam_kernel<true, true>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), p, m, v, g, n,
                       scale, hyper, sparse_begin, copy16, copy_begin, AdamImages{});
  else
    hipLaunchKernelGGL((adam_kernel<true, false>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), p, m, v, g, n,
                       scale, hyper, sparse_begin, copy16, copy_begin, AdamImages{});
  MTAM_CHECK_LAUNCH("adam_bf16copy");
  return MTAM_OK;
}

extern "C" int mtam_sqnorm_state_loss(const float *g, size_t n, float *partials, int offset, const float *lr,
                                      float *adam_state, const float *l2_partial, int n_l2, const float *ce, int B,
                                      float reg, float ce_scale, float *loss, void *stream) {
  MTAM_CHECK_ARG(g && partials && n > 0 && offset >= 0, "sqnorm_state_loss: bad arguments");
  MTAM_CHECK_ARG(!loss || (l2_partial && ce && B > 0 && n_l2 > 0), "sqnorm_state_loss: loss inputs missing");
  MTAM_CHECK_ARG(mtam_aligned16(g), "sqnorm_state_loss: gradient must be 16-byte aligned");
  MTAM_CHECK_ARG((lr == nullptr) == (adam_state == nullptr), "sqnorm_state_loss: lr and adam_state go together");
  hipLaunchKernelGGL(sqnorm_state_loss_kernel, dim3(mtam_sqnorm_blocks(n) + 1), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g, n, partials, offset, lr, adam_state, l2_partial, n_l2, ce, B,
                     reg, ce_scale, loss);
  MTAM_CHECK_LAUNCH("sqnorm_state_loss");
  return MTAM_OK;
}

// every workgroup of the optimizer launch sums the norm's partials itself: fine for thousands of them (ml-1m: 5,357,
// 21 KB of L2 reads per workgroup), not for the 312 k blocks of a 10 M-row table
extern "C" int mtam_adam_clip_max_partials(void) { return 1 << 14; }

struct FeedNext {
  const int32_t *ring = nullptr;
  int32_t *arena = nullptr;
  unsigned *cursor = nullptr;
  int slots = 0, words = 0;
};

static int adam_images_launch(float *p, float *m, float *v, const float *g, size_t n, const float *scale,
                              const float *norm_partials, int n_norm, float clip, float *scale_out,
                              const float *hyper, size_t sparse_begin, uint16_t *copy16, size_t copy_begin,
                              const MtamWeightImages *w, int n_w, void *stream, FeedNext feed = FeedNext());

extern "C" int mtam_adam_images(float *p, float *m, float *v, const float *g, size_t n, const float *scale,
                                const float *hyper, size_t sparse_begin, uint16_t *copy16, size_t copy_begin,
                                const MtamWeightImages *w, int n_w, void *stream) {
  MTAM_CHECK_ARG(scale, "adam_images: bad arguments");
  return adam_images_launch(p, m, v, g, n, scale, nullptr, 0, 0.f, nullptr, hyper, sparse_begin, copy16, copy_begin, w,
                            n_w, stream);
}

extern "C" int mtam_adam_images_clip(float *p, float *m, float *v, const float *g, size_t n,
                                     const float *norm_partials, int n_partials, float clip_norm, float *scale_out,
                                     const float *hyper, size_t sparse_begin, uint16_t *copy16, size_t copy_begin,
                                     const MtamWeightImages *w, int n_w, void *stream) {
  MTAM_CHECK_ARG(norm_partials && scale_out && clip_norm > 0.f && n_partials > 0 &&
                     n_partials <= mtam_adam_clip_max_partials(),
                 "adam_images_clip: 1 .. %d partials, a positive clip norm and a 2-float scale_out",
                 mtam_adam_clip_max_partials());
  return adam_images_launch(p, m, v, g, n, nullptr, norm_partials, n_partials, clip_norm, scale_out, hyper,
                            sparse_begin, copy16, copy_begin, w, n_w, stream);
}

extern "C" int mtam_adam_images_clip_feed(float *p, float *m, float *v, const float *g, size_t n,
                                          const float *norm_partials, int n_partials, float clip_norm,
                                          float *scale_out, const float *hyper, size_t sparse_begin, uint16_t *copy16,
                                          size_t copy_begin, const MtamWeightImages *w, int n_w,
                                          const int32_t *feed_ring, int feed_slots, int feed_words, int32_t *feed_arena,
                                          unsigned int *feed_cursor, void *stream) {
  MTAM_CHECK_ARG(norm_partials && scale_out && clip_norm > 0.f && n_partials > 0 &&
                     n_partials <= mtam_adam_clip_max_partials(),
                 "adam_images_clip_feed: 1 .. %d partials, a positive clip norm and a 2-float scale_out",
                 mtam_adam_clip_max_partials());
  MTAM_CHECK_ARG(feed_ring && feed_arena && feed_cursor && feed_slots > 0 && feed_words > 0,
                 "adam_images_clip_feed: ring, arena, cursor, slots > 0, words > 0");
  MTAM_CHECK_ARG(mtam_aligned16(feed_ring) && mtam_aligned16(feed_arena) && feed_words % 4 == 0,
                 "adam_images_clip_feed: ring and arena 16-byte aligned, the slot pitch a multiple of 4 words");
  {
    const char *r0 = reinterpret_cast<const char *>(feed_ring), *r1 = r0 + (size_t)feed_slots * feed_words * 4;
    const char *a0 = reinterpret_cast<const char *>(feed_arena), *a1 = a0 + (size_t)feed_words * 4;
    MTAM_CHECK_ARG(a1 <= r0 || r1 <= a0, "adam_images_clip_feed: the arena may not lie inside the ring");
  }
  FeedNext f;
  f.ring = feed_ring; f.arena = feed_arena; f.cursor = feed_cursor; f.slots = feed_slots; f.words = feed_words;
  return adam_images_launch(p, m, v, g, n, nullptr, norm_partials, n_partials, clip_norm, scale_out, hyper,
                            sparse_begin, copy16, copy_begin, w, n_w, stream, f);
}

static int adam_images_launch(float *p, float *m, float *v, const float *g, size_t n, const float *scale,
                              const float *norm_partials, int n_norm, float clip, float *scale_out,
                              const float *hyper, size_t sparse_begin, uint16_t *copy16, size_t copy_begin,
                              const MtamWeightImages *w, int n_w, void *stream, FeedNext feed) {
  MTAM_CHECK_ARG(p && m && v && g && hyper && n > 0, "adam_images: bad arguments");
  MTAM_CHECK_ARG(mtam_aligned16(p) && mtam_aligned16(m) && mtam_aligned16(v) && mtam_aligned16(g) &&
                     (reinterpret_cast<uintptr_t>(copy16) & 7u) == 0,
                 "adam_images: buffers must be 16-byte aligned (the bf16 copy 8-byte)");
  MTAM_CHECK_ARG(sparse_begin >= n || sparse_begin % NORM_BLOCK == 0,
                 "adam_images: sparse_begin must be a multiple of %d (or >= n)", NORM_BLOCK);
  MTAM_CHECK_ARG(!copy16 || (copy_begin <= n && copy_begin % 4 == 0), "adam_images: copy_begin must be a multiple of 4");
  MTAM_CHECK_ARG(n_w >= 0 && n_w <= MTAM_MAX_WEIGHT_IMAGES && (n_w == 0 || w), "adam_images: at most %d matrices",
                 MTAM_MAX_WEIGHT_IMAGES);
  AdamImages wi{};
  wi.n = n_w;
  wi.norm_partials = norm_partials; wi.n_norm = n_norm; wi.clip = clip; wi.scale_out = scale_out;
  wi.n_linear = (unsigned)mtam_sqnorm_blocks(n);
  const size_t dense_end = sparse_begin < n ? sparse_begin : n;
  for (int j = 0; j < n_w; ++j) {
    MTAM_CHECK_ARG(w[j].images && w[j].K > 0 && w[j].K % 8 == 0 && w[j].N > 0 && w[j].N % 4 == 0 && w[j].begin % 4 == 0 &&
                       w[j].begin + (size_t)w[j].K * w[j].N <= dense_end &&
                       (reinterpret_cast<uintptr_t>(w[j].images) & 15u) == 0,
                   "adam_images: matrix %d: K must be a multiple of 8, N and begin multiples of 4, inside the dense "
                   "(non-sparse) part [0, %zu)", j, dense_end);
    MTAM_CHECK_ARG((reinterpret_cast<uintptr_t>(w[j].images_r) & 15u) == 0 && (!w[j].images_r || w[j].N % 8 == 0),
                   "adam_images: matrix %d: images_r must be 16-byte aligned and N a multiple of 8", j);
    wi.begin[j] = w[j].begin; wi.end[j] = w[j].begin + (size_t)w[j].K * w[j].N;
    for (int i = 0; i < j; ++i)
      MTAM_CHECK_ARG(wi.end[i] <= wi.begin[j] || wi.end[j] <= wi.begin[i], "adam_images: matrices %d and %d overlap", i, j);
    MTAM_CHECK_ARG(w[j].gru_which == 0 || (w[j].K == MTAM_D && w[j].N == (w[j].gru_which == 1 ? 2 : 1) * MTAM_D &&
                                           w[j].gru_which <= 2),
                   "adam_images: matrix %d: gru_which 1 / 2 is the GRU's wh_g [128, 256] / wh_c [128, 128]", j);
    wi.K[j] = w[j].K; wi.N[j] = w[j].N; wi.img[j] = w[j].images; wi.img_r[j] = w[j].images_r;
    wi.gru_which[j] = w[j].gru_which;
    wi.first_block[j + 1] = wi.first_block[j] + (unsigned)(((size_t)(w[j].K / 4) * (w[j].N / 4) + 255) / 256);
  }
  wi.feed_ring = feed.ring; wi.feed_arena = feed.arena; wi.feed_cursor = feed.cursor;
  wi.feed_slots = (unsigned)feed.slots; wi.feed_words = (unsigned)feed.words;
  dim3 grid(wi.n_linear + wi.first_block[n_w] + (feed.ring ? 1u : 0u));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool nt = n * 4 >= adam_nt_min_bytes();
  const size_t cb = copy16 ? copy_begin : n;
  if (copy16 && nt)
    hipLaunchKernelGGL((adam_kernel<true, true>), grid, dim3(256), 0, st, p, m, v, g, n, scale, hyper, sparse_begin, copy16, cb, wi);
  else if (copy16)
    hipLaunchKernelGGL((adam_kernel<true, false>), grid, dim3(256), 0, st, p, m, v, g, n, scale, hyper, sparse_begin, copy16, cb, wi);
  else if (nt)
    hipLaunchKernelGGL((adam_kernel<false, true>), grid, dim3(256), 0, st, p, m, v, g, n, scale, hyper, sparse_begin, copy16, cb, wi);
  else
    hipLaunchKernelGGL((adam_kernel<false, false>), grid, dim3(256), 0, st, p, m, v, g, n, scale, hyper, sparse_begin, copy16, cb, wi);
  MTAM_CHECK_LAUNCH("adam_images");
  return MTAM_OK;
}

extern "C" int mtam_opt_update(int kind, float *p, float *slot1, float *slot2, const float *g, size_t n,
                               const float *scale, const float *lr, size_t sparse_begin, size_t rowskip_end,
                               void *stream) {
  MTAM_CHECK_ARG(kind >= OPT_SGD && kind <= OPT_RMSPROP, "opt_update: kind must be 0 (sgd), 1 (adadelta) or 2 (rmsprop)");
  MTAM_CHECK_ARG(p && g && scale && lr && n > 0, "opt_update: bad arguments");
  MTAM_CHECK_ARG(kind == OPT_SGD || (slot1 && slot2), "opt_update: adadelta / rmsprop need two slot buffers");
  MTAM_CHECK_ARG(mtam_aligned16(p) && mtam_aligned16(g) && mtam_aligned16(slot1) && mtam_aligned16(slot2),
                 "opt_update: buffers must be 16-byte aligned");
  MTAM_CHECK_ARG(sparse_begin >= n || sparse_begin % NORM_BLOCK == 0,
                 "opt_update: sparse_begin must be a multiple of %d (or >= n)", NORM_BLOCK);
  MTAM_CHECK_ARG(rowskip_end >= sparse_begin && (rowskip_end - sparse_begin) % 128 == 0,
                 "opt_update: the row-sparse region must be whole 128-float rows");
  dim3 grid(mtam_sqnorm_blocks(n));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (kind == OPT_SGD)
    hipLaunchKernelGGL(opt_kernel<OPT_SGD>, grid, dim3(256), 0, st, p, slot1, slot2, g, n, scale, lr, sparse_begin, rowskip_end);
  else if (kind == OPT_ADADELTA)
    hipLaunchKernelGGL(opt_kernel<OPT_ADADELTA>, grid, dim3(256), 0, st, p, slot1, slot2, g, n, scale, lr, sparse_begin, rowskip_end);
  else
    hipLaunchKernelGGL(opt_kernel<OPT_RMSPROP>, grid, dim3(256), 0, st, p, slot1, slot2, g, n, scale, lr, sparse_begin, rowskip_end);
  MTAM_CHECK_LAUNCH("opt_update");
  return MTAM_OK;
}
