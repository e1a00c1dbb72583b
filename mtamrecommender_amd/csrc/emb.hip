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
  int n_rm, n_tr, n_user;      // chunks: row-major (item, category), transposed (position), user
  int n_partials;
};

// Scatter-add with a per-workgroup duplicate pre-reduction.
//
// Float atomics run at the memory side and serialise per 64-B line: with popularity-skewed ids
// the hottest item/category row (hundreds of hits per batch) and the position rows (every sample
// hits rows 0..len-1) set the kernel's duration, not the byte count.  So a workgroup owns a
// chunk of CH = 64 slots of ONE table, finds the slots of its chunk that share a row
// (leader = first slot with that id), sums those in LDS (ds_add_f32) and issues one global
// atomic row per distinct id; slots that are alone in their chunk go straight from registers to
// global atomics.  Chunks of the item and category tables are 64 consecutive (b, t) slots;
// position chunks are 64 samples at ONE time index t, where the reference's data has a single
// id (Prepare/mask_data_process.py:245-247).
// Every atomic wave instruction still covers two 128-B segments in two rows (the full-rate shape).
constexpr int CH = 64;
constexpr int SLOTS_PER_HALF = CH / 8;

__global__ __launch_bounds__(256) void emb_scatter_kernel(ScatterArgs p) {
  __shared__ float acc[CH][D];                 // 32 KB: sums of the rows that share an id
  __shared__ int s_row[CH], s_id[CH], s_lead[CH], s_cnt[CH];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int half = lane >> 5, li = lane & 31;
  const int wave_in_block = tid >> 6;
  const int R = p.B * p.L;
  const int n_work = 2 * p.n_rm + p.n_tr + p.n_user;

  if ((int)blockIdx.x == n_work) {
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
    for (int i = n_work * 4 + lane; i < p.n_partials; i += 64) p.sq_partial[i] = (i == n_work * 4) ? sq : 0.f;
    return;
  }

  // ---- which table and which chunk (block-uniform)
  int c = blockIdx.x, table;
  if (c < p.n_rm) table = 0;
  else if ((c -= p.n_rm) < p.n_rm) table = 1;
  else if ((c -= p.n_rm) < p.n_tr) table = 2;
  else { c -= p.n_tr; table = 3; }
  const float *d_base, *e_base;
  const int32_t *ids;
  float *g;
  int rows, stride;
  if (table == 0)      { d_base = p.d_ic;     e_base = p.ic;     ids = p.item_ids; g = p.g_item; rows = p.item_rows; stride = 2 * D; }
  else if (table == 1) { d_base = p.d_ic + D; e_base = p.ic + D; ids = p.cat_ids;  g = p.g_cat;  rows = p.cat_rows;  stride = 2 * D; }
  else if (table == 2) { d_base = p.d_pos;    e_base = p.pos;    ids = p.pos_ids;  g = p.g_pos;  rows = p.pos_rows;  stride = D; }
  else                 { d_base = nullptr;    e_base = p.user;   ids = p.user_ids; g = p.g_user; rows = p.user_rows; stride = D; }

  // ---- slot -> gathered-row index (or -1: dead) and clamped id
  if (tid < CH) {
    int r = -1;
    if (table <= 1) {
      const int q = c * CH + tid;
      if (q < R) {
        const int b = q / p.L;
        if (q - b * p.L < min(max(p.seq_len[b], 0), p.L)) r = q;
      }
    } else if (table == 2) {
      const int t = c % p.L, b = (c / p.L) * CH + tid;
      if (b < p.B && t < min(max(p.seq_len[b], 0), p.L)) r = b * p.L + t;
    } else {
      const int b = c * CH + tid;
      if (b < p.B) r = b;
    }
    s_row[tid] = r;
    s_id[tid] = r >= 0 ? clamp_id(ids[r], rows) : -1;
  }
  {
    float4 *z = reinterpret_cast<float4 *>(&acc[0][0]);
#pragma unroll
    for (int i = 0; i < CH * D / 4 / 256; ++i) z[tid + 256 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  // ---- every half wave loads its slots' rows (independent of the duplicate search below)
  const int hw = wave_in_block * 2 + half;
  float v[SLOTS_PER_HALF][4];
  int my_row[SLOTS_PER_HALF];
#pragma unroll
  for (int k = 0; k < SLOTS_PER_HALF; ++k) {
    const int s = hw + 8 * k;
    const int r = s_row[s];
    my_row[k] = r;
    if (r >= 0) {
      const float *e = e_base + (size_t)r * stride;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[k][q] = p.reg * e[li + 32 * q];
      if (d_base) {
        const float *d = d_base + (size_t)r * stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[k][q] += d[li + 32 * q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[k][q] = 0.f;
    }
  }
  // ---- leader (first slot of the chunk with the same id) and multiplicity
  if (tid < CH) {
    const int mine = s_id[tid];
    int lead = tid, cnt = 0;
    if (mine >= 0) {
#pragma unroll 8
      for (int j = 0; j < CH; ++j) {
        const bool same = s_id[j] == mine;
        cnt += same ? 1 : 0;
        lead = (same && j < lead) ? j : lead;
      }
    }
    s_lead[tid] = lead;
    s_cnt[tid] = cnt;
  }
  __syncthreads();

  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < SLOTS_PER_HALF; ++k) {
    if (my_row[k] < 0) continue;
    const int s = hw + 8 * k;
    const int lead = s_lead[s];
#pragma unroll
    for (int q = 0; q < 4; ++q) sq += v[k][q] * v[k][q];
    if (s_cnt[lead] > 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) atomicAdd(&acc[lead][li + 32 * q], v[k][q]);
    } else {
      float *gr = g + (size_t)s_id[s] * D;
#pragma unroll
      for (int q = 0; q < 4; ++q) atomicAdd(gr + li + 32 * q, v[k][q]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SLOTS_PER_HALF; ++k) {
    const int s = hw + 8 * k;
    if (my_row[k] < 0 || s_lead[s] != s || s_cnt[s] <= 1) continue;
    float *gr = g + (size_t)s_id[s] * D;
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(gr + li + 32 * q, acc[s][li + 32 * q]);
  }
  sq = wave_sum(sq);
  if (lane == 0) p.sq_partial[blockIdx.x * 4 + wave_in_block] = sq;
}

int gather_waves(int B, int L) {
  const int R = B * L;
  const int total = R + (R + B + 1) / 2;
  return (total + SLOTS_PER_WAVE - 1) / SLOTS_PER_WAVE;
}
int scatter_rm_chunks(int B, int L) { return (B * L + CH - 1) / CH; }
int scatter_tr_chunks(int B, int L) { return ((B + CH - 1) / CH) * L; }
int scatter_user_chunks(int B) { return (B + CH - 1) / CH; }
int scatter_work_blocks(int B, int L) {
  return 2 * scatter_rm_chunks(B, L) + scatter_tr_chunks(B, L) + scatter_user_chunks(B);
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
                item_rows, cat_rows, pos_rows, user_rows, slot_sq_partial,
                scatter_rm_chunks(B, L), scatter_tr_chunks(B, L), with_user ? scatter_user_chunks(B) : 0,
                mtam_emb_scatter_partials(B, L)};
  hipLaunchKernelGGL(emb_scatter_kernel, dim3(2 * a.n_rm + a.n_tr + a.n_user + 1), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("emb_scatter");
  return MTAM_OK;
}
