// Embedding lookups of the time-aware path and their sparse gradient.
//
// Forward  (mtam_emb_gather_fwd): tf.nn.embedding_lookup x4 +
//   concat(item, category) + the tf.nn.l2_loss sums
//   (Embedding/Behavior_embedding_time_aware_attention.py:68-95,
//    Model/base_model.py:302-307).  HBM-bound row copy: a 512-B row is moved by
//   a half wave as 32 x 16 B, two rows in flight per half wave.
// Backward (mtam_emb_scatter_add_bwd): the IndexedSlices gradient of the four
//   lookups, added into dense per-table gradient buffers.  Rows that share an id
//   inside a 128-slot chunk are summed on chip first (registers, then LDS), so
//   one f32 atomic row per distinct id leaves the workgroup, each atomic wave
//   instruction covering two 128-B row segments (the full-rate shape,
//   MI355X_MICROARCH.md "Global float atomics").
#include "common.h"

namespace {

constexpr int D = MTAM_D;
constexpr int SLOTS_PER_WAVE = 2;      // tools/gather_lab.hip at B=128: 1 -> 4.31, 2 -> 4.06, 4 -> 4.68, 8 -> 5.79 us

__device__ __forceinline__ int clamp_id(int id, int rows) { return min(max(id, 0), rows - 1); }

struct GatherArgs {
  const float *item_table, *cat_table, *pos_table, *user_table;
  int item_rows, cat_rows, pos_rows, user_rows;
  const int32_t *item_ids, *cat_ids, *pos_ids, *user_ids;
  int B, L, with_user;
  float *ic_out, *pos_out, *user_out, *l2_partial;
  // optional: two float ranges to clear (the step's gradient accumulators), spread over the grid.
  // The gather is the first kernel of a training step and is latency-bound, so the stores ride along.
  float4 *clear_a, *clear_b;
  size_t n_a4, n_b4;
  const uint16_t *item16;      // ITEM16: the item rows are read from this bf16 image of the table (256 B per row)
};

// Slot s < R      : row r = s of the [item | category] concat, one wave (halves = item, category).
// Slot s >= R     : two rows of {position rows 0..R-1, user rows R..R+B-1}, one per half wave.
// ITEM16 (mixed precision, BASELINE.json configs[4]): item rows come from the bf16 copy and are widened to
// fp32 on the way to the activations; a lane then needs 8 bytes of an item row or 16 of any other row, so
// every lane issues two 8-byte loads (the second one redundant for item lanes) -- uniform, unconditional.
template <bool ITEM16>
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
  bool narrow[SLOTS_PER_WAVE];
#pragma unroll
  for (int i = 0; i < SLOTS_PER_WAVE; ++i) {
    const int s = s0 + i;
    src[i] = nullptr;
    dst[i] = nullptr;
    count[i] = true;
    narrow[i] = false;
    if (s < slots_ic) {
      const int id = half ? p.cat_ids[s] : p.item_ids[s];
      const float *tab = half ? p.cat_table : p.item_table;
      const int rows = half ? p.cat_rows : p.item_rows;
      src[i] = tab + (size_t)clamp_id(id, rows) * D + 4 * li;
      if (ITEM16 && !half) {
        src[i] = reinterpret_cast<const float *>(p.item16 + (size_t)clamp_id(id, rows) * D + 4 * li);
        narrow[i] = true;
      }
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
  if (ITEM16) {
    uint2 lo[SLOTS_PER_WAVE], hi[SLOTS_PER_WAVE];
#pragma unroll
    for (int i = 0; i < SLOTS_PER_WAVE; ++i) {
      const uint2 *q = reinterpret_cast<const uint2 *>(src[i] ? src[i] : p.cat_table);
      lo[i] = q[0];
      hi[i] = q[narrow[i] ? 0 : 1];
    }
#pragma unroll
    for (int i = 0; i < SLOTS_PER_WAVE; ++i) {
      if (narrow[i])          // four bf16: element 0 in the low half of the first word
        v[i] = make_float4(__uint_as_float(lo[i].x << 16), __uint_as_float(lo[i].x & 0xffff0000u),
                           __uint_as_float(lo[i].y << 16), __uint_as_float(lo[i].y & 0xffff0000u));
      else
        v[i] = make_float4(__uint_as_float(lo[i].x), __uint_as_float(lo[i].y), __uint_as_float(hi[i].x),
                           __uint_as_float(hi[i].y));
      if (!src[i]) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
#pragma unroll
    for (int i = 0; i < SLOTS_PER_WAVE; ++i)
      v[i] = src[i] ? *reinterpret_cast<const float4 *>(src[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
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
  if (p.n_a4 | p.n_b4) {
    const size_t stride = (size_t)gridDim.x * 256, g0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = g0; i < p.n_a4; i += stride) p.clear_a[i] = z;
    for (size_t i = g0; i < p.n_b4; i += stride) p.clear_b[i] = z;
  }
}

constexpr int WPB = 16;                      // waves (= norm partials) per scatter workgroup

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
  // non-NULL: the looked-up position rows were never written out (the gather lives inside the forward's first
  // GEMM kernel, mtam_seq_chain_gather_fwd): the L2 term of a position slot reads its table row through the id
  const float *pos_table;
  // non-NULL (with d_ic NULL): the [item | category] gradient rows were never written either -- a chunk of the
  // item (category) table computes its own 128 x 128 block  d_z[chunk rows] . W4[table half]^T  on the matrix cores
  // (dense4emb's input gradient, Embedding/Behavior_embedding_time_aware_attention.py:95-101 under tf.gradients)
  const float *d_z, *W4;
  // data-parallel row-sharded scoring (mtam_emb_scatter_add_bwd_range): only item slots whose id lies in
  // [item_lo, item_hi) are added (the rank's own rows of the item gradient); n_cat = 0 drops the category chunks
  // (n_tr = n_user = 0 the others) when another rank's slots are applied to the item rows alone
  int item_lo, item_hi, n_cat;
  // mtam_emb_scatter_add_bwd_norm: workgroups past the padded-slot one do what csrc/optim.hip's
  // sqnorm_state_loss_kernel does (partial sums of squares of the DENSE gradient, complete by now; the Adam state;
  // the reported loss) on CUs this launch leaves idle -- one launch and its gap less per step.  nr.g NULL: none
  MtamNormRider nr;
  int nr_blocks;
};

// Scatter-add with a per-workgroup duplicate pre-reduction.
//
// Float atomics run at the memory side and serialise per 64-B line (about 25 ns per add on one
// line): with popularity-skewed ids, the mask-token row (once per sample) and the position rows
// (every sample hits rows 0..len-1) the hottest line sets the kernel's duration, not the byte
// count.  So a 1024-thread workgroup owns a chunk of CH = 128 slots of ONE table and issues one
// global atomic row per distinct id of its chunk:
//   1. every half wave owns 4 slots; all of the block's loads (ids, lengths, 8 x 128-B row
//      segments per lane pair) are issued up front, branch-free;
//   2. a half wave sums its own slots that share an id in registers;
//   3. what is left elects a leader per id through a 256-bucket LDS hash (ds_min_u64 of
//      slot:id); followers park their row in LDS with plain stores and set their bit in the
//      leader's member mask (LDS float atomics are not used: same-address ds_add_f32 measured
//      about 150 ns per follower row);
//   4. leaders (and the rare bucket-collision losers) sum registers + their members' rows and
//      add the result to the table gradient with global atomics, each wave instruction covering
//      two 128-B segments in two rows (the full-rate shape).
// Chunks of the item and category tables are 128 consecutive (b, t) slots; position chunks are
// 128 samples at ONE time index t, where the reference's data has a single id
// (Prepare/mask_data_process.py:245-247).  Phase costs were read with tools/scatter_lab.hip.
constexpr int CH = 128;                       // slots per workgroup
constexpr int SCATTER_THREADS = 1024;
constexpr int NHW = SCATTER_THREADS / 32;     // half waves per workgroup
constexpr int SPH = CH / NHW;                 // slots per half wave (4)
constexpr int NBUCKET = 256;

// sum of n floats in float64 by the whole 1024-thread workgroup (one fixed order)
__device__ __forceinline__ double rider_sum_f64(const float *__restrict__ src, int n, double *dred) {
  double acc = 0.0;
  for (int base = 0; base < n; base += SCATTER_THREADS * 4) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = src[min(base + (int)threadIdx.x + SCATTER_THREADS * q, n - 1)];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc += (base + (int)threadIdx.x + SCATTER_THREADS * q < n) ? (double)v[q] : 0.0;
  }
  acc = wave_sum_f64(acc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = acc;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < SCATTER_THREADS / 64; ++w) t += dred[w];
  return t;
}

// Rider workgroup r: four 4,096-float blocks of the dense gradient (one per 256 threads; the arithmetic of
// csrc/optim.hip's sqnorm kernels), or -- the last one -- the Adam state and the reported loss.
__device__ __forceinline__ void norm_rider(const ScatterArgs &p, int r, float *lds) {
  const MtamNormRider &nr = p.nr;
  constexpr int BLOCK = 4096;
  const int tid = threadIdx.x;
  if (r == (p.nr_blocks + 3) / 4) {
    if (nr.adam_state && tid == 0) {
      const float b1 = nr.adam_state[1], b2 = nr.adam_state[2];
      const float b1p = nr.adam_state[4], b2p = nr.adam_state[5];
      nr.adam_state[0] = nr.lr[0] * sqrtf(1.0f - b2p) / (1.0f - b1p);
      nr.adam_state[4] = b1p * b1;
      nr.adam_state[5] = b2p * b2;
    }
    if (nr.loss) {
      double *dred = reinterpret_cast<double *>(lds);
      const double l2 = 0.5 * rider_sum_f64(nr.l2_partial, nr.n_l2, dred);
      const double ces = rider_sum_f64(nr.ce, nr.B, dred);
      if (tid == 0) {
        nr.loss[0] = (float)((double)nr.reg * l2 + (double)nr.ce_scale * ces);
        nr.loss[1] = (float)l2;
        nr.loss[2] = (float)((double)nr.ce_scale * ces);
      }
    }
    return;
  }
  const int sub = tid >> 8, t = tid & 255, block = 4 * r + sub;
  const size_t base = (size_t)block * BLOCK;
  float s = 0.f;
  if (block < p.nr_blocks) {
#pragma unroll
    for (int i = 0; i < BLOCK / 1024; ++i) {
      const size_t o = base + (size_t)(t + 256 * i) * 4;
      if (o + 3 < nr.n) {
        const float4 v = *reinterpret_cast<const float4 *>(nr.g + o);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      } else {
        for (size_t q = o; q < nr.n && q < o + 4; ++q) s += nr.g[q] * nr.g[q];
      }
    }
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) lds[tid >> 6] = s;
  __syncthreads();
  if (t == 0 && block < p.nr_blocks)
    nr.partials[nr.offset + block] = (lds[4 * sub] + lds[4 * sub + 1]) + (lds[4 * sub + 2] + lds[4 * sub + 3]);
}

__global__ __launch_bounds__(SCATTER_THREADS) void emb_scatter_kernel(ScatterArgs p) {
  __shared__ float stage[CH][D];               // 64 KB: rows of the follower slots
  __shared__ unsigned long long hkey[NBUCKET]; // min over the bucket of (slot << 32 | id)
  __shared__ unsigned member[CH][CH / 32];     // per leader slot: bit mask of its follower slots
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int li = tid & 31;
  const int hw = tid >> 5;
  const int wave_in_block = tid >> 6;
  const int R = p.B * p.L;
  const int n_work = p.n_rm + p.n_cat + p.n_tr + p.n_user;

  if ((int)blockIdx.x > n_work) {
    norm_rider(p, blockIdx.x - n_work - 1, &stage[0][0]);
    return;
  }
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
      const int iid = clamp_id(p.item_ids[r], p.item_rows);
      const bool do_item = iid >= p.item_lo && iid < p.item_hi, do_cat = p.n_cat > 0, do_pos = p.n_tr > 0;
      float *gi = p.g_item + (size_t)iid * D;
      float *gc = do_cat ? p.g_cat + (size_t)clamp_id(p.cat_ids[r], p.cat_rows) * D : nullptr;
      float *gp = do_pos ? p.g_pos + (size_t)clamp_id(p.pos_ids[r], p.pos_rows) * D : nullptr;
      for (int e = lane; e < D; e += 64) {
        const float vi = do_item ? p.ic[(size_t)r * 2 * D + e] : 0.f, vc = do_cat ? p.ic[(size_t)r * 2 * D + D + e] : 0.f;
        const float vp = !do_pos ? 0.f
                         : p.pos_table ? p.pos_table[(size_t)clamp_id(p.pos_ids[r], p.pos_rows) * D + e]
                                       : p.pos[(size_t)r * D + e];
        if (do_item) atomicAdd(gi + e, w * vi);
        if (do_cat) atomicAdd(gc + e, w * vc);
        if (do_pos) atomicAdd(gp + e, w * vp);
        const float a = p.reg * vi, b = p.reg * vc, c = p.reg * vp;
        sq += (float)n_pad * (a * a + b * b + c * c);
      }
    }
    sq = wave_sum(sq);
    for (int i = n_work * WPB + lane; i < p.n_partials; i += 64) p.sq_partial[i] = (i == n_work * WPB) ? sq : 0.f;
    return;
  }

  // ---- which table and which chunk (block-uniform)
  int c = blockIdx.x, table;
  if (c < p.n_rm) table = 0;
  else if ((c -= p.n_rm) < p.n_cat) table = 1;
  else if ((c -= p.n_rm) < p.n_tr) table = 2;
  else { c -= p.n_tr; table = 3; }
  const float *d_base, *e_base;
  const int32_t *ids;
  float *g;
  int rows, stride;
  if (table == 0)      { d_base = p.d_ic;     e_base = p.ic;     ids = p.item_ids; g = p.g_item; rows = p.item_rows; stride = 2 * D; }
  else if (table == 1) { d_base = p.d_ic ? p.d_ic + D : nullptr; e_base = p.ic + D; ids = p.cat_ids;  g = p.g_cat;  rows = p.cat_rows;  stride = 2 * D; }
  else if (table == 2) { d_base = p.d_pos;    e_base = p.pos;    ids = p.pos_ids;  g = p.g_pos;  rows = p.pos_rows;  stride = D; }
  else                 { d_base = nullptr;    e_base = p.user;   ids = p.user_ids; g = p.g_user; rows = p.user_rows; stride = D; }

  // ---- all loads of the block up front, branch-free (clamped addresses, masked afterwards)
  // (b, t) of slot q = c * CH + i without a per-lane integer division: the chunk's first slot is divided
  // once on the scalar unit; lanes add i and divide the small remainder (< L + CH) by a multiply-shift
  // that is exact for L <= 256 (16 waves share 4 SIMDs here: every VALU instruction costs four-fold).
  const int q0 = c * CH, b0 = q0 / p.L, t0 = q0 - b0 * p.L;
  const unsigned inv_L = (1u << 20) / (unsigned)p.L + 1u;
  int cand[SPH], id[SPH], sl[SPH], tt[SPH];
#pragma unroll
  for (int k = 0; k < SPH; ++k) {
    const int i = hw + NHW * k;                   // slot of the chunk
    int bb = 0;
    cand[k] = -1;
    tt[k] = -1;                                   // < 0: always live (user)
    if (table <= 1) {
      const int q = q0 + i;
      if (q < R) {
        const int x = t0 + i;
        const int wraps = p.L <= 256 ? (int)(((unsigned)x * inv_L) >> 20) : x / p.L;
        cand[k] = q; bb = b0 + wraps; tt[k] = x - wraps * p.L;
      }
    } else if (table == 2) {
      tt[k] = c % p.L;
      bb = (c / p.L) * CH + i;
      if (bb < p.B) cand[k] = bb * p.L + tt[k];
    } else {
      bb = c * CH + i;
      if (bb < p.B) cand[k] = bb;
    }
    id[k] = ids[max(cand[k], 0)];
    sl[k] = p.seq_len[min(bb, p.B - 1)];
  }
  float v[SPH][4], dv[SPH][4];
  const bool fused_dic = table <= 1 && p.d_z != nullptr;                  // block-uniform
  const bool pos_from_table = table == 2 && p.pos_table != nullptr;      // block-uniform
#pragma unroll
  for (int k = 0; k < SPH; ++k) {
    const float *e = pos_from_table ? p.pos_table + (size_t)clamp_id(id[k], rows) * D + li
                                    : e_base + (size_t)max(cand[k], 0) * stride + li;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[k][q] = e[32 * q];
  }
  if (fused_dic) {
    // ---- this chunk's block of d[item | category] = d_z . W4^T, 128 slots x 128 columns, on 16 waves (one 32 x 32
    // tile each, K = 128 in two halves of 32 k-steps), through `stage` into the half waves' row layout
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int tm = wave_in_block >> 2, tn = wave_in_block & 3, r = lane & 31, h = lane >> 5;
    const float *arow = p.d_z + (size_t)min(q0 + 32 * tm + r, R - 1) * D + 64 * h;
    const float *brow = p.W4 + (size_t)(table * D + 32 * tn + r) * D + 64 * h;
    f32x16 acc = {0.f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float a[32], b[32];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 x = *reinterpret_cast<const float4 *>(arow + 32 * half + 4 * i);
        const float4 y = *reinterpret_cast<const float4 *>(brow + 32 * half + 4 * i);
        a[4 * i] = x.x; a[4 * i + 1] = x.y; a[4 * i + 2] = x.z; a[4 * i + 3] = x.w;
        b[4 * i] = y.x; b[4 * i + 1] = y.y; b[4 * i + 2] = y.z; b[4 * i + 3] = y.w;
      }
#pragma unroll
      for (int s_ = 0; s_ < 32; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_], b[s_], acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) stage[32 * tm + (q & 3) + 8 * (q >> 2) + 4 * h][32 * tn + r] = acc[q];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SPH; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) dv[k][q] = stage[hw + NHW * k][li + 32 * q];
    // (the barrier below, after the row loads, separates these reads from the followers' later writes to `stage`)
  }
  if (fused_dic) {
    // (dv came off the matrix cores just above, while the row loads issued before it were in flight)
  } else if (d_base) {                            // block-uniform
#pragma unroll
    for (int k = 0; k < SPH; ++k) {
      const float *d = d_base + (size_t)max(cand[k], 0) * stride + li;
#pragma unroll
      for (int q = 0; q < 4; ++q) dv[k][q] = d[32 * q];
    }
  } else {
#pragma unroll
    for (int k = 0; k < SPH; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) dv[k][q] = 0.f;
  }
  if (tid < NBUCKET) hkey[tid] = ~0ull;
  if (tid < CH * (CH / 32)) (&member[0][0])[tid] = 0u;
  __syncthreads();

  // ---- contributions, the un-deduplicated norm, then the half wave's own duplicates in registers
  float sq = 0.f;
  int n_in[SPH];                                  // own slots summed into slot k (0: dead or merged away)
#pragma unroll
  for (int k = 0; k < SPH; ++k) {
    id[k] = clamp_id(id[k], rows);
    const bool live = cand[k] >= 0 && tt[k] < min(max(sl[k], 0), p.L) &&
                      (table != 0 || (id[k] >= p.item_lo && id[k] < p.item_hi));
    n_in[k] = live ? 1 : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[k][q] = live ? fmaf(p.reg, v[k][q], dv[k][q]) : 0.f;
      sq += v[k][q] * v[k][q];
    }
  }
#pragma unroll
  for (int k = 1; k < SPH; ++k) {
#pragma unroll
    for (int k2 = 0; k2 < k; ++k2) {
      const bool m = n_in[k] > 0 && n_in[k2] > 0 && id[k] == id[k2];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[k2][q] += m ? v[k][q] : 0.f;
      n_in[k2] += m ? n_in[k] : 0;
      n_in[k] = m ? 0 : n_in[k];
    }
  }
  // ---- leader election per id: min slot of the bucket
  unsigned bucket[SPH];
#pragma unroll
  for (int k = 0; k < SPH; ++k) {
    bucket[k] = ((unsigned)id[k] * 2654435761u) >> 24;
    if (n_in[k] > 0 && li == 0)
      atomicMin(&hkey[bucket[k]], ((unsigned long long)(hw + NHW * k) << 32) | (unsigned)id[k]);
  }
  __syncthreads();
  bool direct[SPH];
#pragma unroll
  for (int k = 0; k < SPH; ++k) {
    const int s_ = hw + NHW * k;
    const unsigned long long key = hkey[bucket[k]];
    const int lead = (int)(key >> 32);
    const bool follower = n_in[k] > 0 && (int)(unsigned)key == id[k] && lead != s_;
    direct[k] = n_in[k] > 0 && !follower;
    if (follower) {
#pragma unroll
      for (int q = 0; q < 4; ++q) stage[s_][li + 32 * q] = v[k][q];
      if (li == 0) atomicOr(&member[lead][s_ >> 5], 1u << (s_ & 31));
    }
  }
  // LDS writes complete (lgkmcnt only: the barrier must not wait for global traffic)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // ---- one global atomic row per distinct id of the chunk (plus bucket-collision losers)
#pragma unroll
  for (int k = 0; k < SPH; ++k) {
    if (direct[k]) {
      const int s_ = hw + NHW * k;
#pragma unroll
      for (int w = 0; w < CH / 32; ++w) {
        unsigned bits = member[s_][w];
        while (bits) {
          const int j = 32 * w + __builtin_ctz(bits);
          bits &= bits - 1;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[k][q] += stage[j][li + 32 * q];
        }
      }
      float *gr = g + (size_t)id[k] * D + li;
#pragma unroll
      for (int q = 0; q < 4; ++q) atomicAdd(gr + 32 * q, v[k][q]);
    }
  }
  sq = wave_sum(sq);
  if (lane == 0) p.sq_partial[blockIdx.x * WPB + wave_in_block] = sq;
}

// Rows of a ROW RANGE of a table, by catalog row number: out[i] = table_rows[ids[i] - row0] if the range holds row
// ids[i], else zeros.  Data-parallel training with the item table sharded by rows (data_parallel.py, "sharded-table"):
// every rank runs this over the ids of ALL ranks and a reduce-scatter of the results hands each rank the rows of its
// own batch -- every row comes from exactly one owner, so the sum IS the row.  Half a wave per row, 16 bytes per lane.
__global__ __launch_bounds__(256) void rows_gather_range_kernel(const float *__restrict__ table_rows, int row0,
                                                                int nrows, const int32_t *__restrict__ ids, long n,
                                                                float *__restrict__ out) {
  const long i = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (i >= n) return;
  const int li = threadIdx.x & 31;
  const int t = ids[i] - row0;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t >= 0 && t < nrows) v = *reinterpret_cast<const float4 *>(table_rows + (size_t)t * D + 4 * li);
  *reinterpret_cast<float4 *>(out + (size_t)i * D + 4 * li) = v;
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

extern "C" int mtam_rows_gather_range(const float *table_rows, int row0, int nrows, const int32_t *ids, long n,
                                      float *out, void *stream) {
  MTAM_CHECK_ARG(table_rows && ids && out && nrows > 0 && row0 >= 0 && n > 0, "rows_gather_range: bad arguments");
  MTAM_CHECK_ARG(mtam_aligned16(table_rows) && mtam_aligned16(out), "rows_gather_range: rows must be 16-byte aligned");
  hipLaunchKernelGGL(rows_gather_range_kernel, dim3((unsigned)((n + 7) / 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), table_rows, row0, nrows, ids, n, out);
  MTAM_CHECK_LAUNCH("rows_gather_range");
  return MTAM_OK;
}

extern "C" int mtam_emb_gather_partials(int B, int L) { return ((gather_waves(B, L) + 3) / 4) * 4; }

extern "C" int mtam_emb_gather_fwd(const float *item_table, int item_rows, const float *cat_table,
                                   int cat_rows, const float *pos_table, int pos_rows,
                                   const float *user_table, int user_rows, const int32_t *item_ids,
                                   const int32_t *cat_ids, const int32_t *pos_ids,
                                   const int32_t *user_ids, int B, int L, int with_user,
                                   float *item_cat_out, float *pos_out, float *user_out,
                                   float *l2_partial, void *stream) {
  return mtam_emb_gather_fwd_clear(item_table, item_rows, cat_table, cat_rows, pos_table, pos_rows, user_table,
                                   user_rows, item_ids, cat_ids, pos_ids, user_ids, B, L, with_user, item_cat_out,
                                   pos_out, user_out, l2_partial, nullptr, 0, nullptr, 0, stream);
}

extern "C" int mtam_emb_gather_fwd_clear(const float *item_table, int item_rows, const float *cat_table,
                                         int cat_rows, const float *pos_table, int pos_rows,
                                         const float *user_table, int user_rows, const int32_t *item_ids,
                                         const int32_t *cat_ids, const int32_t *pos_ids,
                                         const int32_t *user_ids, int B, int L, int with_user,
                                         float *item_cat_out, float *pos_out, float *user_out,
                                         float *l2_partial, float *clear_a, size_t n_a, float *clear_b,
                                         size_t n_b, void *stream) {
  return mtam_emb_gather_fwd_item16(item_table, nullptr, item_rows, cat_table, cat_rows, pos_table, pos_rows,
                                    user_table, user_rows, item_ids, cat_ids, pos_ids, user_ids, B, L, with_user,
                                    item_cat_out, pos_out, user_out, l2_partial, clear_a, n_a, clear_b, n_b, stream);
}

extern "C" int mtam_emb_gather_fwd_item16(const float *item_table, const uint16_t *item16, int item_rows,
                                          const float *cat_table, int cat_rows, const float *pos_table,
                                          int pos_rows, const float *user_table, int user_rows,
                                          const int32_t *item_ids, const int32_t *cat_ids, const int32_t *pos_ids,
                                          const int32_t *user_ids, int B, int L, int with_user,
                                          float *item_cat_out, float *pos_out, float *user_out, float *l2_partial,
                                          float *clear_a, size_t n_a, float *clear_b, size_t n_b, void *stream) {
  MTAM_CHECK_ARG(!item16 || (reinterpret_cast<uintptr_t>(item16) & 7u) == 0, "emb_gather: item16 must be 8-byte aligned");
  MTAM_CHECK_ARG((n_a == 0 || (clear_a && mtam_aligned16(clear_a) && n_a % 4 == 0)) &&
                     (n_b == 0 || (clear_b && mtam_aligned16(clear_b) && n_b % 4 == 0)),
                 "emb_gather: clear ranges must be 16-byte aligned multiples of 4 floats");
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
               item_cat_out, pos_out, user_out, l2_partial,
               reinterpret_cast<float4 *>(clear_a), reinterpret_cast<float4 *>(clear_b), n_a / 4, n_b / 4, item16};
  const int blocks = mtam_emb_gather_partials(B, L) / 4;
  if (item16)
    hipLaunchKernelGGL(emb_gather_kernel<true>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(emb_gather_kernel<false>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("emb_gather");
  return MTAM_OK;
}

extern "C" int mtam_emb_scatter_partials(int B, int L) { return (scatter_work_blocks(B, L) + 1) * WPB; }

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
  return mtam_emb_scatter_add_bwd_postab(d_item_cat, d_pos, item_cat, pos, nullptr, user, item_ids, cat_ids, pos_ids,
                                         user_ids, seq_len, B, L, reg, with_user, g_item, item_rows, g_cat, cat_rows,
                                         g_pos, pos_rows, g_user, user_rows, slot_sq_partial, stream);
}

extern "C" int mtam_emb_scatter_add_bwd_postab(const float *d_item_cat, const float *d_pos, const float *item_cat,
                                               const float *pos, const float *pos_table, const float *user,
                                               const int32_t *item_ids, const int32_t *cat_ids,
                                               const int32_t *pos_ids, const int32_t *user_ids,
                                               const int32_t *seq_len, int B, int L, float reg, int with_user,
                                               float *g_item, int item_rows, float *g_cat, int cat_rows, float *g_pos,
                                               int pos_rows, float *g_user, int user_rows, float *slot_sq_partial,
                                               void *stream) {
  return mtam_emb_scatter_add_bwd_fused(d_item_cat, nullptr, nullptr, d_pos, item_cat, pos, pos_table, user, item_ids,
                                        cat_ids, pos_ids, user_ids, seq_len, B, L, reg, with_user, g_item, item_rows,
                                        g_cat, cat_rows, g_pos, pos_rows, g_user, user_rows, slot_sq_partial, stream);
}

extern "C" int mtam_emb_scatter_add_bwd_fused(const float *d_item_cat, const float *d_z, const float *W4,
                                              const float *d_pos, const float *item_cat, const float *pos,
                                              const float *pos_table, const float *user, const int32_t *item_ids,
                                              const int32_t *cat_ids, const int32_t *pos_ids,
                                              const int32_t *user_ids, const int32_t *seq_len, int B, int L, float reg,
                                              int with_user, float *g_item, int item_rows, float *g_cat, int cat_rows,
                                              float *g_pos, int pos_rows, float *g_user, int user_rows,
                                              float *slot_sq_partial, void *stream) {
  return mtam_emb_scatter_add_bwd_range(d_item_cat, d_z, W4, d_pos, item_cat, pos, pos_table, user, item_ids, cat_ids,
                                        pos_ids, user_ids, seq_len, B, L, reg, with_user, g_item, item_rows, g_cat,
                                        cat_rows, g_pos, pos_rows, g_user, user_rows, slot_sq_partial, 0, item_rows, 0,
                                        stream);
}

static int scatter_launch(const float *d_item_cat, const float *d_z, const float *W4, const float *d_pos,
                          const float *item_cat, const float *pos, const float *pos_table, const float *user,
                          const int32_t *item_ids, const int32_t *cat_ids, const int32_t *pos_ids,
                          const int32_t *user_ids, const int32_t *seq_len, int B, int L, float reg, int with_user,
                          float *g_item, int item_rows, float *g_cat, int cat_rows, float *g_pos, int pos_rows,
                          float *g_user, int user_rows, float *slot_sq_partial, int item_lo, int item_hi, int item_only,
                          const MtamNormRider *norm, void *stream);

extern "C" int mtam_emb_scatter_add_bwd_range(const float *d_item_cat, const float *d_z, const float *W4,
                                              const float *d_pos, const float *item_cat, const float *pos,
                                              const float *pos_table, const float *user, const int32_t *item_ids,
                                              const int32_t *cat_ids, const int32_t *pos_ids,
                                              const int32_t *user_ids, const int32_t *seq_len, int B, int L, float reg,
                                              int with_user, float *g_item, int item_rows, float *g_cat, int cat_rows,
                                              float *g_pos, int pos_rows, float *g_user, int user_rows,
                                              float *slot_sq_partial, int item_lo, int item_hi, int item_only,
                                              void *stream) {
  return scatter_launch(d_item_cat, d_z, W4, d_pos, item_cat, pos, pos_table, user, item_ids, cat_ids, pos_ids, user_ids,
                        seq_len, B, L, reg, with_user, g_item, item_rows, g_cat, cat_rows, g_pos, pos_rows, g_user,
                        user_rows, slot_sq_partial, item_lo, item_hi, item_only, nullptr, stream);
}

extern "C" int mtam_emb_scatter_add_bwd_norm(const float *d_item_cat, const float *d_z, const float *W4,
                                             const float *d_pos, const float *item_cat, const float *pos,
                                             const float *pos_table, const float *user, const int32_t *item_ids,
                                             const int32_t *cat_ids, const int32_t *pos_ids,
                                             const int32_t *user_ids, const int32_t *seq_len, int B, int L, float reg,
                                             int with_user, float *g_item, int item_rows, float *g_cat, int cat_rows,
                                             float *g_pos, int pos_rows, float *g_user, int user_rows,
                                             float *slot_sq_partial, const MtamNormRider *norm, void *stream) {
  MTAM_CHECK_ARG(norm && norm->g && norm->partials && norm->n > 0 && norm->offset >= 0 && mtam_aligned16(norm->g),
                 "emb_scatter (norm rider): g (16-byte aligned), n and partials are required");
  MTAM_CHECK_ARG(!norm->loss || (norm->l2_partial && norm->ce && norm->B > 0 && norm->n_l2 > 0),
                 "emb_scatter (norm rider): loss inputs missing");
  MTAM_CHECK_ARG((norm->lr == nullptr) == (norm->adam_state == nullptr),
                 "emb_scatter (norm rider): lr and adam_state go together");
  return scatter_launch(d_item_cat, d_z, W4, d_pos, item_cat, pos, pos_table, user, item_ids, cat_ids, pos_ids, user_ids,
                        seq_len, B, L, reg, with_user, g_item, item_rows, g_cat, cat_rows, g_pos, pos_rows, g_user,
                        user_rows, slot_sq_partial, 0, item_rows, 0, norm, stream);
}

static int scatter_launch(const float *d_item_cat, const float *d_z, const float *W4, const float *d_pos,
                          const float *item_cat, const float *pos, const float *pos_table, const float *user,
                          const int32_t *item_ids, const int32_t *cat_ids, const int32_t *pos_ids,
                          const int32_t *user_ids, const int32_t *seq_len, int B, int L, float reg, int with_user,
                          float *g_item, int item_rows, float *g_cat, int cat_rows, float *g_pos, int pos_rows,
                          float *g_user, int user_rows, float *slot_sq_partial, int item_lo, int item_hi, int item_only,
                          const MtamNormRider *norm, void *stream) {
  MTAM_CHECK_ARG(B > 0 && L > 0, "emb_scatter: B and L must be positive");
  MTAM_CHECK_ARG((long)B * L * 3 + B < 0x3fffffffL, "emb_scatter: batch too large");
  MTAM_CHECK_ARG(0 <= item_lo && item_lo <= item_hi && item_hi <= item_rows, "emb_scatter: bad item row range [%d, %d)",
                 item_lo, item_hi);
  if (item_only) {
    // another rank's slots applied to this rank's item rows: only the item halves and the item ids are read
    MTAM_CHECK_ARG(d_item_cat && item_cat && item_ids && seq_len && g_item && slot_sq_partial && item_rows > 0,
                   "emb_scatter (item only): null argument");
    ScatterArgs a{d_item_cat, nullptr, item_cat, nullptr, nullptr, item_ids, nullptr, nullptr, nullptr, seq_len,
                  B, L, 0, reg, g_item, nullptr, nullptr, nullptr, item_rows, 1, 1, 1, slot_sq_partial,
                  scatter_rm_chunks(B, L), 0, 0, mtam_emb_scatter_partials(B, L), nullptr, nullptr, nullptr,
                  item_lo, item_hi, 0, MtamNormRider{}, 0};
    hipLaunchKernelGGL(emb_scatter_kernel, dim3(a.n_rm + 1), dim3(SCATTER_THREADS), 0, static_cast<hipStream_t>(stream), a);
    MTAM_CHECK_LAUNCH("emb_scatter");
    return MTAM_OK;
  }
  MTAM_CHECK_ARG((d_item_cat || (d_z && W4)) && d_pos && item_cat && (pos || pos_table) && user,
                 "emb_scatter: null gradient or gathered rows");
  MTAM_CHECK_ARG(d_item_cat || (mtam_aligned16(d_z) && mtam_aligned16(W4)),
                 "emb_scatter: d_z and W4 must be 16-byte aligned");
  MTAM_CHECK_ARG(item_ids && cat_ids && pos_ids && user_ids && seq_len, "emb_scatter: null ids");
  MTAM_CHECK_ARG(g_item && g_cat && g_pos && (g_user || !with_user) && slot_sq_partial, "emb_scatter: null output");
  MTAM_CHECK_ARG(item_rows > 0 && cat_rows > 0 && pos_rows > 0 && user_rows > 0, "emb_scatter: empty table");
  ScatterArgs a{d_item_cat, d_pos, item_cat, pos ? pos : pos_table, user, item_ids, cat_ids, pos_ids, user_ids, seq_len,
                B, L, with_user, reg, g_item, g_cat, g_pos, g_user,
                item_rows, cat_rows, pos_rows, user_rows, slot_sq_partial,
                scatter_rm_chunks(B, L), scatter_tr_chunks(B, L), with_user ? scatter_user_chunks(B) : 0,
                mtam_emb_scatter_partials(B, L), pos ? nullptr : pos_table, d_item_cat ? nullptr : d_z,
                d_item_cat ? nullptr : W4, item_lo, item_hi, scatter_rm_chunks(B, L), MtamNormRider{}, 0};
  int riders = 0;
  if (norm) {
    a.nr = *norm;
    a.nr_blocks = (int)((norm->n + 4095) / 4096);
    riders = (a.nr_blocks + 3) / 4 + 1;
  }
  hipLaunchKernelGGL(emb_scatter_kernel, dim3(a.n_rm + a.n_cat + a.n_tr + a.n_user + 1 + riders),
                     dim3(SCATTER_THREADS), 0, static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("emb_scatter");
  return MTAM_OK;
}
