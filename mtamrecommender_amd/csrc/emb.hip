// Embedding lookups of the time-aware path and their sparse gradient.
//
// Forward  (mtam_emb_gather_fwd): tf.nn.embedding_lookup x4 +
//   concat(item, category) + the tf.nn.l2_loss sums
//   (Embedding/Behavior_embedding_time_aware_attention.py:68-95,
//    Model/base_model.py:302-307).  HBM-bound row copy: a 512-B row is moved by
//   a half wave as 32 x 16 B, four rows in flight per half wave.
// Backward (mtam_emb_scatter_add_bwd): the IndexedSlices gradient of the four
//   lookups, added into dense per-table gradient buffers with wave-level f32
//   atomics shaped as two 128-B row segments per wave instruction (the
//   full-rate shape, MI355X_MICROARCH.md "Global float atomics").
#include "common.h"

namespace {

constexpr int D = MTAM_D;
constexpr int SLOTS_PER_WAVE = 4;

__device__ __forceinline__ int clamp_id(int id, int rows) { return min(max(id, 0), rows - 1); }

struct GatherArgs {
  const float *item_table, *cat_table, *pos_table, *user_table;
  int item_rows, cat_rows, pos_rows, user_rows;
  const int32_t *item_ids, *cat_ids, *pos_ids, *user_ids;
  int B, L, with_user;
  float *ic_out, *pos_out, *user_out, *l2_partial;
};

// Slot s < R      : row r = s of the [item | category] concat, one wave (halves = item, category).
// Slot s >= R     : two rows of {position rows 0..R-1, user rows R..R+B-1}, one per half wave.
__global__ __launch_bounds__(256) void emb_gather_kernel(GatherArgs p) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, li = lane & 31;
  const int R = p.B * p.L;
  const int slots_ic = R;
  const int total = slots_ic + (R + p.B + 1) / 2;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int s0 = wave_id * SLOTS_PER_WAVE;

  const float *src[SLOTS_PER_WAVE];
  float *dst[SLOTS_PER_WAVE];
  bool count[SLOTS_PER_WAVE];
#pragma unroll
  for (int i = 0; i < SLOTS_PER_WAVE; ++i) {
    const int s = s0 + i;
    src[i] = nullptr;
    dst[i] = nullptr;
    count[i] = true;
    if (s < slots_ic) {
      const int id = half ? p.cat_ids[s] : p.item_ids[s];
      const float *tab = half ? p.cat_table : p.item_table;
      const int rows = half ? p.cat_rows : p.item_rows;
      src[i] = tab + (size_t)clamp_id(id, rows) * D + 4 * li;
      dst[i] = p.ic_out + (size_t)s * (2 * D) + half * D + 4 * li;
    } else if (s < total) {
      const int q = 2 * (s - slots_ic) + half;
      if (q < R) {
        src[i] = p.pos_table + (size_t)clamp_id(p.pos_ids[q], p.pos_rows) * D + 4 * li;
        dst[i] = p.pos_out + (size_t)q * D + 4 * li;
      } else if (q < R + p.B) {
        const int b = q - R;
        src[i] = p.user_table + (size_t)clamp_id(p.user_ids[b], p.user_rows) * D + 4 * li;
        dst[i] = p.user_out + (size_t)b * D + 4 * li;
        count[i] = p.with_user != 0;
      }
    }
  }
  float4 v[SLOTS_PER_WAVE];
#pragma unroll
  for (int i = 0; i < SLOTS_PER_WAVE; ++i)
    v[i] = src[i] ? *reinterpret_cast<const float4 *>(src[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < SLOTS_PER_WAVE; ++i) {
    if (dst[i]) {
      *reinterpret_cast<float4 *>(dst[i]) = v[i];
      if (count[i]) sq += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
  }
  sq = wave_sum(sq);
  if (lane == 0) p.l2_partial[wave_id] = sq;
}

struct ScatterArgs {
  const float *d_ic, *d_pos, *ic, *pos, *user;
  const int32_t *item_ids, *cat_ids, *pos_ids, *user_ids, *seq_len;
  int B, L, with_user;
  float reg;
  float *g_item, *g_cat, *g_pos, *g_user;
  int item_rows, cat_rows, pos_rows, user_rows;
  float *sq_partial;
  int work_blocks;
};

// Wave slot s -> two consecutive rows of ONE table (wave-uniform choice):
//   [0, P) item, [P, 2P) category, [2P, 3P) position with P = ceil(R/2); then ceil(B/2) user slots.
// One half wave per row; lane li adds floats li, li+32, li+64, li+96 of the row, so every
// atomic wave instruction covers two 128-B segments in two rows.
__device__ __forceinline__ float scatter_row(const float *__restrict__ d, const float *__restrict__ e,
                                             float *__restrict__ g, float reg, int li) {
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = li + 32 * q;
    v[q] = reg * e[c] + (d ? d[c] : 0.f);
  }
  float sq = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    atomicAdd(g + li + 32 * q, v[q]);
    sq += v[q] * v[q];
  }
  return sq;
}

__global__ __launch_bounds__(256) void emb_scatter_kernel(ScatterArgs p) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, li = lane & 31;
  const int R = p.B * p.L;
  const int wave_in_block = threadIdx.x >> 6;

  if ((int)blockIdx.x == p.work_blocks) {
    // Padded slots: every one of them holds row 0 of its table and a zero upstream
    // gradient, so their contributions collapse to n_pad * reg * row0 per table.
    if (wave_in_block != 0) return;
    int n_pad = 0, first = 0x7fffffff;
    for (int b = lane; b < p.B; b += 64) {
      const int sl = min(max(p.seq_len[b], 0), p.L);
      n_pad += p.L - sl;
      if (sl < p.L) first = min(first, b * p.L + sl);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      n_pad += __shfl_xor(n_pad, off, 64);
      first = min(first, __shfl_xor(first, off, 64));
    }
    float sq = 0.f;
    if (n_pad > 0) {
      const float w = p.reg * (float)n_pad;
      const int r = first;
      float *gi = p.g_item + (size_t)clamp_id(p.item_ids[r], p.item_rows) * D;
      float *gc = p.g_cat + (size_t)clamp_id(p.cat_ids[r], p.cat_rows) * D;
      float *gp = p.g_pos + (size_t)clamp_id(p.pos_ids[r], p.pos_rows) * D;
      for (int e = lane; e < D; e += 64) {
        const float vi = p.ic[(size_t)r * 2 * D + e], vc = p.ic[(size_t)r * 2 * D + D + e];
        const float vp = p.pos[(size_t)r * D + e];
        atomicAdd(gi + e, w * vi);
        atomicAdd(gc + e, w * vc);
        atomicAdd(gp + e, w * vp);
        const float a = p.reg * vi, b = p.reg * vc, c = p.reg * vp;
        sq += (float)n_pad * (a * a + b * b + c * c);
      }
    }
    sq = wave_sum(sq);
    if (lane < 4) p.sq_partial[p.work_blocks * 4 + lane] = (lane == 0) ? sq : 0.f;
    return;
  }

  const int wave_id = blockIdx.x * 4 + wave_in_block;
  const int P = (R + 1) / 2;
  const int total = 3 * P + (p.with_user ? (p.B + 1) / 2 : 0);
  float sq = 0.f;
  for (int i = 0; i < SLOTS_PER_WAVE; ++i) {
    const int s = wave_id * SLOTS_PER_WAVE + i;          // wave-uniform
    if (s >= total) break;
    if (s < 3 * P) {
      const int t = s / P;                                 // wave-uniform table
      const int r = 2 * (s - t * P) + half;
      bool live = r < R;
      if (live) {
        const int b = r / p.L;
        live = (r - b * p.L) < min(max(p.seq_len[b], 0), p.L);
      }
      if (live) {
        if (t == 0) {
          sq += scatter_row(p.d_ic + (size_t)r * 2 * D, p.ic + (size_t)r * 2 * D,
                            p.g_item + (size_t)clamp_id(p.item_ids[r], p.item_rows) * D, p.reg, li);
        } else if (t == 1) {
          sq += scatter_row(p.d_ic + (size_t)r * 2 * D + D, p.ic + (size_t)r * 2 * D + D,
                            p.g_cat + (size_t)clamp_id(p.cat_ids[r], p.cat_rows) * D, p.reg, li);
        } else {
          sq += scatter_row(p.d_pos + (size_t)r * D, p.pos + (size_t)r * D,
                            p.g_pos + (size_t)clamp_id(p.pos_ids[r], p.pos_rows) * D, p.reg, li);
        }
      }
    } else {
      const int b = 2 * (s - 3 * P) + half;
      if (b < p.B)
        sq += scatter_row(nullptr, p.user + (size_t)b * D,
                          p.g_user + (size_t)clamp_id(p.user_ids[b], p.user_rows) * D, p.reg, li);
    }
  }
  sq = wave_sum(sq);
  if (lane == 0) p.sq_partial[wave_id] = sq;
}

int gather_waves(int B, int L) {
  const int R = B * L;
  const int total = R + (R + B + 1) / 2;
  return (total + SLOTS_PER_WAVE - 1) / SLOTS_PER_WAVE;
}
int scatter_work_blocks(int B, int L) {
  const int slots = 3 * ((B * L + 1) / 2) + (B + 1) / 2;
  const int waves = (slots + SLOTS_PER_WAVE - 1) / SLOTS_PER_WAVE;
  return (waves + 3) / 4;
}

}  // namespace

extern "C" int mtam_emb_gather_partials(int B, int L) { return ((gather_waves(B, L) + 3) / 4) * 4; }

extern "C" int mtam_emb_gather_fwd(const float *item_table, int item_rows, const float *cat_table,
                                   int cat_rows, const float *pos_table, int pos_rows,
                                   const float *user_table, int user_rows, const int32_t *item_ids,
                                   const int32_t *cat_ids, const int32_t *pos_ids,
                                   const int32_t *user_ids, int B, int L, int with_user,
                                   float *item_cat_out, float *pos_out, float *user_out,
                                   float *l2_partial, void *stream) {
  MTAM_CHECK_ARG(B > 0 && L > 0, "emb_gather: B and L must be positive");
  MTAM_CHECK_ARG((long)B * L * 3 + B < 0x3fffffffL, "emb_gather: batch too large");
  MTAM_CHECK_ARG(item_table && cat_table && pos_table && user_table, "emb_gather: null table");
  MTAM_CHECK_ARG(item_rows > 0 && cat_rows > 0 && pos_rows > 0 && user_rows > 0, "emb_gather: empty table");
  MTAM_CHECK_ARG(item_ids && cat_ids && pos_ids && user_ids, "emb_gather: null ids");
  MTAM_CHECK_ARG(item_cat_out && pos_out && user_out && l2_partial, "emb_gather: null output");
  MTAM_CHECK_ARG(mtam_aligned16(item_table) && mtam_aligned16(cat_table) && mtam_aligned16(pos_table) &&
                     mtam_aligned16(user_table) && mtam_aligned16(item_cat_out) && mtam_aligned16(pos_out) &&
                     mtam_aligned16(user_out),
                 "emb_gather: tables and outputs must be 16-byte aligned");
  GatherArgs a{item_table, cat_table, pos_table, user_table, item_rows, cat_rows, pos_rows, user_rows,
               item_ids, cat_ids, pos_ids, user_ids, B, L, with_user,
               item_cat_out, pos_out, user_out, l2_partial};
  const int blocks = mtam_emb_gather_partials(B, L) / 4;
  hipLaunchKernelGGL(emb_gather_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("emb_gather");
  return MTAM_OK;
}

extern "C" int mtam_emb_scatter_partials(int B, int L) { return (scatter_work_blocks(B, L) + 1) * 4; }

extern "C" int mtam_emb_scatter_add_bwd(const float *d_item_cat, const float *d_pos, const float *item_cat,
                                        const float *pos, const float *user, const int32_t *item_ids,
                                        const int32_t *cat_ids, const int32_t *pos_ids,
                                        const int32_t *user_ids, const int32_t *seq_len, int B, int L,
                                        float reg, int with_user, float *g_item, int item_rows,
                                        float *g_cat, int cat_rows, float *g_pos, int pos_rows,
                                        float *g_user, int user_rows, float *slot_sq_partial,
                                        void *stream) {
  MTAM_CHECK_ARG(B > 0 && L > 0, "emb_scatter: B and L must be positive");
  MTAM_CHECK_ARG((long)B * L * 3 + B < 0x3fffffffL, "emb_scatter: batch too large");
  MTAM_CHECK_ARG(d_item_cat && d_pos && item_cat && pos && user, "emb_scatter: null gradient or gathered rows");
  MTAM_CHECK_ARG(item_ids && cat_ids && pos_ids && user_ids && seq_len, "emb_scatter: null ids");
  MTAM_CHECK_ARG(g_item && g_cat && g_pos && (g_user || !with_user) && slot_sq_partial, "emb_scatter: null output");
  MTAM_CHECK_ARG(item_rows > 0 && cat_rows > 0 && pos_rows > 0 && user_rows > 0, "emb_scatter: empty table");
  ScatterArgs a{d_item_cat, d_pos, item_cat, pos, user, item_ids, cat_ids, pos_ids, user_ids, seq_len,
                B, L, with_user, reg, g_item, g_cat, g_pos, g_user,
                item_rows, cat_rows, pos_rows, user_rows, slot_sq_partial, scatter_work_blocks(B, L)};
  hipLaunchKernelGGL(emb_scatter_kernel, dim3(a.work_blocks + 1), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("emb_scatter");
  return MTAM_OK;
}
