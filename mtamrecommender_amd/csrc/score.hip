// Head of the model: contrib layer_norm, full-catalog softmax cross entropy and
// top-K over the catalog.
//   layer_norm : Model/Modules/net_utils.py:229-232 (tf.contrib.layers.layer_norm)
//   softmax CE : Model/base_model.py:316-322 (log_softmax + one_hot + reduce_mean)
//   top-K      : Model/base_model.py:194-200 (tf.nn.top_k x5 over pred @ table^T)
// The logits themselves come from mtam_gemm_f32 (pred @ table^T).
#include "common.h"

namespace {

constexpr int D = MTAM_D;
constexpr int CE_CHUNK = 4096;   // logits per workgroup in the softmax passes

// ------------------------------------------------------------- layer norm
__global__ __launch_bounds__(256) void layer_norm_fwd_kernel(const float *__restrict__ x,
                                                             const float *__restrict__ resid, const float *beta,
                                                             const float *gamma, float eps, int form, int rows,
                                                             float *__restrict__ y, float *__restrict__ save) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float x0 = x[(size_t)row * D + lane], x1 = x[(size_t)row * D + 64 + lane];
  if (resid) {
    x0 += resid[(size_t)row * D + lane];
    x1 += resid[(size_t)row * D + 64 + lane];
  }
  const float mean = wave_sum(x0 + x1) / (float)D;
  const float d0 = x0 - mean, d1 = x1 - mean;
  const float var = wave_sum(d0 * d0 + d1 * d1) / (float)D;
  float rstd;
  if (form == 0) {
    // tf.contrib.layers.layer_norm via nn.batch_normalization: x * inv + (beta - mean * inv)
    rstd = 1.0f / sqrtf(var + eps);
    const float i0 = rstd * gamma[lane], i1 = rstd * gamma[64 + lane];
    y[(size_t)row * D + lane] = x0 * i0 + (beta[lane] - mean * i0);
    y[(size_t)row * D + 64 + lane] = x1 * i1 + (beta[64 + lane] - mean * i1);
  } else {
    // Time_Aware_Attention.normalize: gamma * (x - mean) / sqrt(var + eps) + beta
    const float sd = sqrtf(var + eps);
    rstd = 1.0f / sd;
    y[(size_t)row * D + lane] = gamma[lane] * (d0 / sd) + beta[lane];
    y[(size_t)row * D + 64 + lane] = gamma[64 + lane] * (d1 / sd) + beta[64 + lane];
  }
  if (save) {
    save[(size_t)row * (D + 1) + lane] = d0 * rstd;
    save[(size_t)row * (D + 1) + 64 + lane] = d1 * rstd;
    if (lane == 0) save[(size_t)row * (D + 1) + D] = rstd;
  }
}

// A workgroup takes LN_BWD_ROWS rows (a wave: every fourth of them, all its loads issued before the first use) and
// adds its sums for d(beta), d(gamma) ONCE: with 4 rows per workgroup, 3,200 workgroups piled 256 float atomics each
// on the same 256 addresses at 12,800 rows (80 us; same-address atomics serialise at the memory side).
constexpr int LN_BWD_ROWS = 32;
__global__ __launch_bounds__(256) void layer_norm_bwd_kernel(const float *__restrict__ dy, const float *gamma,
                                                             const float *__restrict__ save, int rows,
                                                             float *__restrict__ dx, float *d_bg) {
  __shared__ float sb[4][D], sg[4][D];
  constexpr int IT = LN_BWD_ROWS / 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row0 = blockIdx.x * LN_BWD_ROWS + wv;
  const float gm0 = gamma[lane], gm1 = gamma[64 + lane];
  float y0[IT], y1[IT], h0[IT], h1[IT], rs[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const size_t row = min(row0 + 4 * i, rows - 1);          // clamped: rows past the end are masked below
    y0[i] = dy[row * D + lane]; y1[i] = dy[row * D + 64 + lane];
    h0[i] = save[row * (D + 1) + lane]; h1[i] = save[row * (D + 1) + 64 + lane];
    rs[i] = save[row * (D + 1) + D];
  }
  float b0 = 0.f, b1 = 0.f, g0 = 0.f, g1 = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int row = row0 + 4 * i;
    if (row < rows) {
      const float a0 = y0[i] * gm0, a1 = y1[i] * gm1;
      const float m1 = wave_sum(a0 + a1) / (float)D;
      const float m2 = wave_sum(a0 * h0[i] + a1 * h1[i]) / (float)D;
      dx[(size_t)row * D + lane] = rs[i] * (a0 - m1 - h0[i] * m2);
      dx[(size_t)row * D + 64 + lane] = rs[i] * (a1 - m1 - h1[i] * m2);
      b0 += y0[i]; b1 += y1[i]; g0 = fmaf(y0[i], h0[i], g0); g1 = fmaf(y1[i], h1[i], g1);
    }
  }
  sb[wv][lane] = b0; sb[wv][64 + lane] = b1; sg[wv][lane] = g0; sg[wv][64 + lane] = g1;
  __syncthreads();
  if (threadIdx.x < D) {
    const int c = threadIdx.x;
    atomicAdd(d_bg + c, sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
    atomicAdd(d_bg + D + c, sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
  }
}

// ------------------------------------------------------------- softmax CE
__device__ __forceinline__ float block_reduce_max(float v, float *red) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return v;
}
__device__ __forceinline__ float block_reduce_sum(float v, float *red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return v;
}

__global__ __launch_bounds__(256) void ce_partial_kernel(const float *__restrict__ logits, int ld, int V,
                                                         int chunks, float *__restrict__ partial) {
  __shared__ float red[4];
  const int b = blockIdx.y, c = blockIdx.x;
  const float *row = logits + (size_t)b * ld;
  const int v0 = c * CE_CHUNK, v1 = min(V, v0 + CE_CHUNK);
  float vals[CE_CHUNK / 256];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < CE_CHUNK / 256; ++i) {
    const int v = v0 + threadIdx.x + 256 * i;
    vals[i] = (v < v1) ? row[v] : -INFINITY;
    m = fmaxf(m, vals[i]);
  }
  m = block_reduce_max(m, red);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CE_CHUNK / 256; ++i) s += expf(vals[i] - m);   // exp(-inf) = 0 for the tail
  s = block_reduce_sum(s, red);
  if (threadIdx.x == 0) {
    partial[((size_t)b * chunks + c) * 2 + 0] = m;
    partial[((size_t)b * chunks + c) * 2 + 1] = s;
  }
}

__global__ __launch_bounds__(256) void ce_finish_kernel(const float *__restrict__ logits, int ld,
                                                        const int32_t *__restrict__ target, int V, int chunks,
                                                        const float *__restrict__ partial,
                                                        float *__restrict__ lse, float *__restrict__ ce) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float *pp = partial + (size_t)b * chunks * 2;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < chunks; c += 256) m = fmaxf(m, pp[2 * c]);
  m = block_reduce_max(m, red);
  float s = 0.f;
  for (int c = threadIdx.x; c < chunks; c += 256) s += pp[2 * c + 1] * expf(pp[2 * c] - m);
  s = block_reduce_sum(s, red);
  if (threadIdx.x == 0) {
    const float l = m + logf(s);
    const int t = min(max(target[b], 0), V - 1);
    lse[b] = l;
    ce[b] = l - logits[(size_t)b * ld + t];
  }
}

__global__ __launch_bounds__(256) void ce_grad_kernel(const float *__restrict__ logits, int ld,
                                                      const int32_t *__restrict__ target, int V,
                                                      const float *__restrict__ lse, float scale,
                                                      float *__restrict__ d_logits) {
  const int b = blockIdx.y;
  const float l = lse[b];
  const int t = target[b];
  const int v0 = blockIdx.x * CE_CHUNK;
#pragma unroll
  for (int i = 0; i < CE_CHUNK / 256; ++i) {
    const int v = v0 + threadIdx.x + 256 * i;
    if (v < V) {
      const float pr = expf(logits[(size_t)b * ld + v] - l);
      d_logits[(size_t)b * ld + v] = (pr - (v == t ? 1.0f : 0.0f)) * scale;
    }
  }
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float *__restrict__ l2_partial, int n_l2,
                                                          const float *__restrict__ ce, int B, float reg,
                                                          float ce_scale, float *__restrict__ loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n_l2; i += 256) s += l2_partial[i];
  const float l2 = 0.5f * block_reduce_sum(s, red);
  s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += ce[i];
  const float ces = block_reduce_sum(s, red);
  if (threadIdx.x == 0) {
    loss[0] = reg * l2 + ce_scale * ces;
    loss[1] = l2;
    loss[2] = ce_scale * ces;          // mean over the GLOBAL batch (ce_scale = 1 / global batch; = ces / B on one GPU)
  }
}

// Single-launch form for catalogs that fit one workgroup's registers (V <= 16384): one workgroup per
// row keeps its logits in registers (max, sum-exp, gradient in one read), and the LAST workgroup to
// finish (agent-scope ticket; payload moved by sc1 stores / sc1 loads) folds the per-row cross
// entropies and the L2 partials into the loss.  `ticket` must be zero on entry; the
// kernel leaves it zero.
constexpr int CE_ROW_MAX = 64;    // logits per thread
// NCH = logits per thread (16 / 32 / 64, chosen from V): every load of the row is issued up front,
// branch-free (clamped column, masked afterwards) -- with the loads inside `if (v < V)` blocks each
// one was waited for before the next was issued (15 serial round trips at V = 3,709: 14 us).
template <int NCH>
__global__ __launch_bounds__(256) void ce_row_loss_kernel(const float *__restrict__ logits, int ld,
                                                          const int32_t *__restrict__ target, int B, int V,
                                                          float grad_scale, float *__restrict__ lse,
                                                          float *__restrict__ ce, float *__restrict__ d_logits,
                                                          unsigned int *ticket, const float *__restrict__ l2_partial,
                                                          int n_l2, float reg, float ce_scale,
                                                          float *__restrict__ loss) {
  __shared__ float red[4];
  __shared__ int s_last;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *row = logits + (size_t)b * ld;
  const int t = min(max(target[b], 0), V - 1);
  const float target_logit = row[t];       // read before d_logits (which may alias logits) is written
  float vals[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) vals[i] = row[min(tid + 256 * i, V - 1)];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    vals[i] = (tid + 256 * i < V) ? vals[i] : -INFINITY;
    m = fmaxf(m, vals[i]);
  }
  m = block_reduce_max(m, red);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += expf(vals[i] - m);       // exp(-inf) = 0 for the masked tail
  s = block_reduce_sum(s, red);
  const float l = m + logf(s);
  if (d_logits) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int v = tid + 256 * i;
      if (v < V) d_logits[(size_t)b * ld + v] = (expf(vals[i] - l) - (v == t ? 1.0f : 0.0f)) * grad_scale;
    }
  }
  if (tid == 0) {
    lse[b] = l;
    // hand-off of ce[b] to whichever workgroup finishes last: write-through (sc1) store, drained,
    // then the ticket; the reader uses sc1 loads -- no L2 write-back fence (this workgroup has just
    // dirtied ~15 KB of gradient lines, which a release fence would have to flush first).
    __hip_atomic_store(ce + b, l - target_logit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = 0;
    if (loss) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int mine = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (mine == (unsigned int)(B - 1)) s_last = 1;
    }
  }
  __syncthreads();
  if (s_last) {
    float a = 0.f;
    for (int i = tid; i < n_l2; i += 256) a += l2_partial[i];
    const float l2 = 0.5f * block_reduce_sum(a, red);
    a = 0.f;
    for (int i = tid; i < B; i += 256) a += __hip_atomic_load(ce + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float ces = block_reduce_sum(a, red);
    if (tid == 0) {
      loss[0] = reg * l2 + ce_scale * ces;
      loss[1] = l2;
      loss[2] = ce_scale * ces;          // mean over the GLOBAL batch (ce_scale = 1 / global batch; = ces / B on one GPU)
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------ top-K
// Radix select on an order-preserving integer image of the float (3 digit
// passes of 11/11/10 bits), then one sweep that keeps everything above the
// threshold and the lowest-index ties, then a 64-wide bitonic sort of
// (value desc, index asc).  One workgroup per row.  Every pass streams the row
// with 8 loads in flight per thread (clamped index, masked): a plain strided loop
// waits out one round trip per element (8 ms per 128 x 1 M rows before).
__device__ __forceinline__ uint32_t order_key(float f) {
  uint32_t u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;                       // -0.0 == +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// Long rows are cut into gridDim.y segments of `seg` elements (one workgroup each, indices reported
// globally, missing entries of a short last segment filled with -inf / -1); a second launch of the same
// kernel over the [rows, segments * k] candidate values picks the final k -- equal values keep the lower
// POSITION, and positions are ordered by (segment, rank) = by global index among equal values -- and maps
// its positions back through `idx_map`.  One workgroup per row left 128 of the 256 CUs idle with one
// wave per SIMD: 80 ms for 128 rows of 50 M scores.
__global__ __launch_bounds__(256) void topk_kernel(const float *__restrict__ scores, long ld, int V_row, int k,
                                                   int32_t *__restrict__ idx_out, float *__restrict__ val_out,
                                                   int seg, const int32_t *__restrict__ idx_map, float fill,
                                                   int out_segs, int out_seg0, int idx_base) {
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t scan[256];
  __shared__ uint32_t sel_bin, sel_above;
  __shared__ uint32_t cand_key[64], cand_idx[64];
  __shared__ uint32_t n_gt, n_eq, eq_base;
  __shared__ uint32_t wave_cnt[4];
  constexpr int EQ_CAP = 1024;
  __shared__ uint32_t eq_list[EQ_CAP];       // indices of the elements equal to the threshold (unordered)

  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int seg0 = blockIdx.y * seg;                       // first element of this workgroup's segment
  const int V = min(seg, V_row - seg0);
  const float *s = scores + (size_t)row * ld + seg0;
  // candidate slot of this (row, segment): rows keep `out_segs` slots (0: gridDim.y), this launch fills the
  // slots from out_seg0 on; reported indices are offset by idx_base (a slab of a longer row)
  const size_t out_row = (size_t)row * (out_segs ? out_segs : gridDim.y) + out_seg0 + blockIdx.y;
  const int kk = max(0, min(k, V));
  // f(index, value) over the row: independent loads in flight per thread and trip -- four 16-byte
  // loads when the row starts on a 16-byte boundary (the model pads the logits row stride), else eight
  // 4-byte loads
  const bool vec = V >= 16384 && (reinterpret_cast<uintptr_t>(s) & 15u) == 0;   // short rows: 18 vs 16 us at V = 3,709
  auto for_each = [&](auto f) {
    int done = 0;
    if (vec) {
      const int n4 = V / 4;
      const float4 *s4 = reinterpret_cast<const float4 *>(s);
      for (int base = 0; base < n4; base += 256 * 4) {
        float4 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = s4[min(base + tid + 256 * q, n4 - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i4 = base + tid + 256 * q;
          if (i4 < n4) {
            f(4 * i4, x[q].x); f(4 * i4 + 1, x[q].y); f(4 * i4 + 2, x[q].z); f(4 * i4 + 3, x[q].w);
          }
        }
      }
      done = 4 * n4;
    }
    for (int base = done; base < V; base += 256 * 8) {
      float x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = s[min(base + tid + 256 * q, V - 1)];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int v = base + tid + 256 * q;
        if (v < V) f(v, x[q]);
      }
    }
  };

  uint32_t prefix = 0;        // selected high bits so far
  uint32_t need = kk;         // how many still to take from the current digit range
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    const int sh = shifts[pass], nb = 1 << bits[pass];
    for (int i = tid; i < 2048; i += 256) hist[i] = 0;
    __syncthreads();
    for_each([&](int, float x) {
      const uint32_t key = order_key(x);
      const bool match = (pass == 0) || ((key >> (sh + bits[pass])) == prefix);
      if (match) atomicAdd(&hist[(key >> sh) & (nb - 1)], 1u);
    });
    __syncthreads();
    // suffix sums over groups of 8 bins
    uint32_t c = 0;
    for (int i = 0; i < 8; ++i) c += hist[tid * 8 + i];
    scan[tid] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const uint32_t add = (tid + off < 256) ? scan[tid + off] : 0u;
      __syncthreads();
      scan[tid] += add;
      __syncthreads();
    }
    const uint32_t above = (tid + 1 < 256) ? scan[tid + 1] : 0u;   // elements in bins above my group
    if (above < need && scan[tid] >= need) {
      uint32_t acc = above;
      for (int i = 7; i >= 0; --i) {
        const uint32_t h = hist[tid * 8 + i];
        if (acc + h >= need) { sel_bin = tid * 8 + i; sel_above = acc; break; }
        acc += h;
      }
    }
    __syncthreads();
    prefix = (prefix << bits[pass]) | sel_bin;
    need -= sel_above;
    __syncthreads();
  }
  const uint32_t thr = prefix;          // key of the kk-th largest element; `need` ties to take (>= 1)

  if (tid == 0) { n_gt = 0; n_eq = 0; eq_base = 0; }
  if (tid < 64) { cand_key[tid] = 0; cand_idx[tid] = 0xffffffffu; }
  __syncthreads();
  const uint32_t gt_total = kk - need;  // number of elements strictly above the threshold
  // one sweep: everything above the threshold is a result (any order, sorted below); the elements equal
  // to it are listed, and the `need` lowest indices among them complete the result
  for_each([&](int v, float x) {
    const uint32_t key = order_key(x);
    if (key > thr) {
      const uint32_t slot = atomicAdd(&n_gt, 1u);
      cand_key[slot] = key;
      cand_idx[slot] = v;
    } else if (key == thr) {
      const uint32_t e = atomicAdd(&n_eq, 1u);
      if (e < EQ_CAP) eq_list[e] = v;
    }
  });
  __syncthreads();
  if (n_eq <= EQ_CAP) {
    const uint32_t ne = n_eq;
    for (uint32_t i = tid; i < ne; i += 256) {
      const uint32_t mine = eq_list[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < ne; ++j) rank += (eq_list[j] < mine) ? 1u : 0u;
      if (rank < need) {
        cand_key[gt_total + rank] = thr;
        cand_idx[gt_total + rank] = mine;
      }
    }
  } else {
    // more ties than the list holds (e.g. a constant row): index-ordered sweep over the ties only
    for (int v0 = 0; v0 < V; v0 += 256) {
      const int v = v0 + tid;
      const bool eq = (v < V) && order_key(s[v]) == thr;
      const unsigned long long bal = __ballot(eq);
      const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wave_cnt[wv] = __popcll(bal);
      __syncthreads();
      uint32_t base = eq_base;
      for (int q = 0; q < wv; ++q) base += wave_cnt[q];
      if (eq) {
        const uint32_t rank = base + before;
        if (rank < need) {
          cand_key[gt_total + rank] = thr;
          cand_idx[gt_total + rank] = v;
        }
      }
      __syncthreads();
      if (tid == 0) eq_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      __syncthreads();
      if (eq_base >= need) break;        // block-uniform (shared counter read after the barrier)
    }
  }
  __syncthreads();

  // bitonic sort of 64 (key, index) pairs, descending by key then ascending by index
  if (wv == 0) {
    unsigned long long item = ((unsigned long long)cand_key[lane] << 32) |
                              (unsigned long long)(0xffffffffu - cand_idx[lane]);
    for (int size = 2; size <= 64; size <<= 1) {
      for (int stride = size >> 1; stride >= 1; stride >>= 1) {
        const unsigned long long other = __shfl_xor(item, stride, 64);
        const bool up = ((lane & size) == 0);           // descending overall
        const bool lower = ((lane & stride) == 0);
        const bool keep_max = (up == lower);
        const unsigned long long mx = item > other ? item : other;
        const unsigned long long mn = item > other ? other : item;
        item = keep_max ? mx : mn;
      }
    }
    if (lane < k) {
      const uint32_t key = (uint32_t)(item >> 32);
      const uint32_t idx = 0xffffffffu - (uint32_t)(item & 0xffffffffu);
      int32_t out = -1;
      if (lane < kk) out = idx_map ? idx_map[(size_t)row * ld + idx] : (int32_t)idx + seg0 + idx_base;
      idx_out[out_row * k + lane] = out;
      if (val_out) val_out[out_row * k + lane] = (lane < kk) ? key_to_float(key) : fill;
    }
  }
}

}  // namespace

extern "C" int mtam_layer_norm_fwd(const float *x, const float *resid, const float *beta, const float *gamma,
                                   float eps, int form, int rows, float *y, float *save, void *stream) {
  MTAM_CHECK_ARG(x && beta && gamma && y && rows > 0 && (form == 0 || form == 1), "layer_norm_fwd: bad arguments");
  hipLaunchKernelGGL(layer_norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, resid, beta, gamma, eps, form, rows, y, save);
  MTAM_CHECK_LAUNCH("layer_norm_fwd");
  return MTAM_OK;
}

extern "C" int mtam_layer_norm_bwd(const float *d_y, const float *gamma, const float *save, int rows,
                                   float *d_x, float *d_bg, void *stream) {
  MTAM_CHECK_ARG(d_y && gamma && save && d_x && d_bg && rows > 0, "layer_norm_bwd: bad arguments");
  hipLaunchKernelGGL(layer_norm_bwd_kernel, dim3((rows + LN_BWD_ROWS - 1) / LN_BWD_ROWS), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d_y, gamma, save, rows, d_x, d_bg);
  MTAM_CHECK_LAUNCH("layer_norm_bwd");
  return MTAM_OK;
}

static int ce_chunks(int V) { return (V + CE_CHUNK - 1) / CE_CHUNK; }

extern "C" int mtam_softmax_ce_partials(int B, int V) { return B * ce_chunks(V) * 2; }

extern "C" int mtam_softmax_ce(const float *logits, int ld, const int32_t *target, int B, int V,
                               float grad_scale, float *lse, float *ce, float *d_logits, float *partial,
                               void *stream) {
  MTAM_CHECK_ARG(logits && target && lse && ce && partial, "softmax_ce: null argument");
  MTAM_CHECK_ARG(B > 0 && B <= 65535 && V > 0 && ld >= V, "softmax_ce: bad shape B=%d V=%d ld=%d", B, V, ld);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int chunks = ce_chunks(V);
  hipLaunchKernelGGL(ce_partial_kernel, dim3(chunks, B), dim3(256), 0, s, logits, ld, V, chunks, partial);
  hipLaunchKernelGGL(ce_finish_kernel, dim3(B), dim3(256), 0, s, logits, ld, target, V, chunks, partial, lse, ce);
  if (d_logits)
    hipLaunchKernelGGL(ce_grad_kernel, dim3(chunks, B), dim3(256), 0, s, logits, ld, target, V, lse,
                       grad_scale, d_logits);
  MTAM_CHECK_LAUNCH("softmax_ce");
  return MTAM_OK;
}

extern "C" int mtam_loss_reduce(const float *l2_partial, int n_l2, const float *ce, int B, float reg,
                                float ce_scale, float *loss, void *stream) {
  MTAM_CHECK_ARG(l2_partial && ce && loss && n_l2 >= 0 && B > 0, "loss_reduce: bad arguments");
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), l2_partial,
                     n_l2, ce, B, reg, ce_scale, loss);
  MTAM_CHECK_LAUNCH("loss_reduce");
  return MTAM_OK;
}

extern "C" int mtam_softmax_ce_loss(const float *logits, int ld, const int32_t *target, int B, int V,
                                    float grad_scale, float *lse, float *ce, float *d_logits, float *partial,
                                    const float *l2_partial, int n_l2, float reg, float ce_scale, float *loss,
                                    void *stream) {
  MTAM_CHECK_ARG(logits && target && lse && ce && partial && (l2_partial || !loss), "softmax_ce_loss: null argument");
  MTAM_CHECK_ARG(B > 0 && B <= 65535 && V > 0 && ld >= V && n_l2 >= 0, "softmax_ce_loss: bad shape");
  if (V <= 256 * CE_ROW_MAX) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned int *ticket = reinterpret_cast<unsigned int *>(partial);
    if (V <= 256 * 16)
      hipLaunchKernelGGL(ce_row_loss_kernel<16>, dim3(B), dim3(256), 0, st, logits, ld, target, B, V, grad_scale, lse,
                         ce, d_logits, ticket, l2_partial, n_l2, reg, ce_scale, loss);
    else if (V <= 256 * 32)
      hipLaunchKernelGGL(ce_row_loss_kernel<32>, dim3(B), dim3(256), 0, st, logits, ld, target, B, V, grad_scale, lse,
                         ce, d_logits, ticket, l2_partial, n_l2, reg, ce_scale, loss);
    else
      hipLaunchKernelGGL(ce_row_loss_kernel<64>, dim3(B), dim3(256), 0, st, logits, ld, target, B, V, grad_scale, lse,
                         ce, d_logits, ticket, l2_partial, n_l2, reg, ce_scale, loss);
    MTAM_CHECK_LAUNCH("softmax_ce_loss");
    return MTAM_OK;
  }
  int rc = mtam_softmax_ce(logits, ld, target, B, V, grad_scale, lse, ce, d_logits, partial + 4, stream);
  if (rc || !loss) return rc;
  return mtam_loss_reduce(l2_partial, n_l2, ce, B, reg, ce_scale, loss, stream);
}

// segments per row for the two-level form (1 = single pass)
static int topk_segments(int V) { return V < 262144 ? 1 : min(32, V / 65536); }
static int topk_seg_len(int V) {
  const int S = topk_segments(V);
  return ((V + S - 1) / S + 1023) / 1024 * 1024;      // multiple of 1024: segments keep the row's 16-byte alignment
}

extern "C" size_t mtam_topk_workspace_bytes(int rows, int V, int k) {
  if (rows <= 0 || V <= 0 || k <= 0 || topk_segments(V) == 1) return 0;
  const int seg = topk_seg_len(V), S = (V + seg - 1) / seg;
  return (size_t)rows * S * k * 8;
}

extern "C" int mtam_topk_ws(const float *scores, int ld, int rows, int V, int k, int32_t *idx_out, float *val_out,
                            void *workspace, void *stream) {
  MTAM_CHECK_ARG(scores && idx_out && rows > 0 && V > 0 && ld >= V, "topk: bad arguments");
  MTAM_CHECK_ARG(k >= 1 && k <= 64, "topk: k must be in [1, 64] (got %d)", k);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int seg = topk_seg_len(V), S = (V + seg - 1) / seg;
  if (!workspace || topk_segments(V) == 1 || S == 1) {
    hipLaunchKernelGGL(topk_kernel, dim3(rows), dim3(256), 0, st, scores, (long)ld, V, k, idx_out, val_out, V,
                       static_cast<const int32_t *>(nullptr), 0.f, 0, 0, 0);
  } else {
    MTAM_CHECK_ARG(rows <= 65535 * 1 && S <= 65535, "topk: too many segments");
    float *cand_val = static_cast<float *>(workspace);
    int32_t *cand_idx = reinterpret_cast<int32_t *>(cand_val + (size_t)rows * S * k);
    hipLaunchKernelGGL(topk_kernel, dim3(rows, S), dim3(256), 0, st, scores, (long)ld, V, k, cand_idx, cand_val, seg,
                       static_cast<const int32_t *>(nullptr), -INFINITY, 0, 0, 0);
    hipLaunchKernelGGL(topk_kernel, dim3(rows), dim3(256), 0, st, cand_val, (long)S * k, S * k, k, idx_out, val_out,
                       S * k, cand_idx, 0.f, 0, 0, 0);
  }
  MTAM_CHECK_LAUNCH("topk");
  return MTAM_OK;
}

// ---- top-K over a catalog that is scored slab by slab (evaluation without stored [rows, V] logits) ----------
// A slab = columns [col0, col0 + width) of the score matrix, held in a [rows, ld] scratch.  Each slab is cut into
// segments of MTAM_TOPK_STREAM_SEG columns; every (row, segment) keeps its k best as candidates in
// `workspace` (values, then indices: rows x total_segments x k each); mtam_topk_stream_finish picks the final
// k per row.  Slabs must start on a multiple of the segment length, so that a candidate's slot order is its
// global index order -- the tie rule (equal values: lower index first) then holds across slab and segment
// boundaries exactly as in mtam_topk.
static const int kStreamSeg = MTAM_TOPK_STREAM_SEG;
extern "C" int mtam_topk_stream_segments(int V) { return V <= 0 ? 0 : (V + kStreamSeg - 1) / kStreamSeg; }
extern "C" size_t mtam_topk_stream_workspace_bytes(int rows, int V, int k) {
  if (rows <= 0 || V <= 0 || k <= 0) return 0;
  return (size_t)rows * mtam_topk_stream_segments(V) * k * 8;
}

extern "C" int mtam_topk_stream_slab(const float *slab_scores, int ld, int rows, int col0, int width, int V, int k,
                                     void *workspace, void *stream) {
  MTAM_CHECK_ARG(slab_scores && workspace && rows > 0 && rows <= 65535, "topk_stream_slab: bad arguments");
  MTAM_CHECK_ARG(k >= 1 && k <= 64, "topk_stream_slab: k must be in [1, 64] (got %d)", k);
  MTAM_CHECK_ARG(col0 >= 0 && width > 0 && ld >= width && (long)col0 + width <= V && col0 % kStreamSeg == 0,
                 "topk_stream_slab: slab [%d, +%d) of %d columns must start on a multiple of %d", col0, width, V,
                 kStreamSeg);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int S = mtam_topk_stream_segments(V), nseg = (width + kStreamSeg - 1) / kStreamSeg;
  float *cand_val = static_cast<float *>(workspace);
  int32_t *cand_idx = reinterpret_cast<int32_t *>(cand_val + (size_t)rows * S * k);
  hipLaunchKernelGGL(topk_kernel, dim3(rows, nseg), dim3(256), 0, st, slab_scores, (long)ld, width, k, cand_idx,
                     cand_val, kStreamSeg, static_cast<const int32_t *>(nullptr), -INFINITY, S, col0 / kStreamSeg,
                     col0);
  MTAM_CHECK_LAUNCH("topk_stream_slab");
  return MTAM_OK;
}

extern "C" int mtam_topk_stream_finish(void *workspace, int rows, int V, int k, int32_t *idx_out, float *val_out,
                                       void *stream) {
  MTAM_CHECK_ARG(workspace && idx_out && rows > 0 && V > 0, "topk_stream_finish: bad arguments");
  MTAM_CHECK_ARG(k >= 1 && k <= 64, "topk_stream_finish: k must be in [1, 64] (got %d)", k);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int S = mtam_topk_stream_segments(V);
  float *cand_val = static_cast<float *>(workspace);
  int32_t *cand_idx = reinterpret_cast<int32_t *>(cand_val + (size_t)rows * S * k);
  hipLaunchKernelGGL(topk_kernel, dim3(rows), dim3(256), 0, st, cand_val, (long)S * k, S * k, k, idx_out, val_out,
                     S * k, cand_idx, 0.f, 0, 0, 0);
  MTAM_CHECK_LAUNCH("topk_stream_finish");
  return MTAM_OK;
}

extern "C" int mtam_topk(const float *scores, int ld, int rows, int V, int k, int32_t *idx_out,
                         float *val_out, void *stream) {
  return mtam_topk_ws(scores, ld, rows, V, k, idx_out, val_out, nullptr, stream);
}
