// The three sequence-side projections of the forward step in ONE launch:
//   z = [item | category] W4 ; zr = relu(z) ; x = zr + position     Embedding/Behavior_embedding_time_aware_attention.py:95-103
//   kv    = relu(x Wkv + bkv)        keys / values of every decoder block       Model/Modules/time_aware_attention.py:251-253
//   xproj = x Wx + bx                input halves of the GRU's gate / candidate products   Model/Modules/time_aware_rnn.py:243-256
// As three GEMM launches they took 11.1 + 10.9 + 13.4 us at B x L = 6,400 rows.  Here a workgroup owns a
// 32-row stripe: the stripe of x never leaves the CU between the first product and the other two -- 25.5 us.
// What that needed (tools lab, per launch): with every accumulator register stored as it lies (its column
// on the lane: 16 four-byte stores of 128-byte segments per tile and lane) the kernel took 34.3 us, 21.8 with
// the stores removed: store ISSUE, not bandwidth (22.9 MB).  Transposed through a wave-private LDS scratch
// a tile leaves as 4 sixteen-byte stores per lane and the stores cost 3.7 us.
//
// v_mfma_f32_32x32x2_f32: lane l (r = l & 31, h = l >> 5) supplies A[row r][k = h] and B[k = h][col r].
// k-step s of a K-long contraction pairs element s (lane half 0) with element s + K/2 (half 1); staged
// rows keep their halves one float apart ([K/2][gap][K/2][gap]) so that the 64 lanes of an A-fragment read
// hit 64 distinct LDS banks.  The B operands (weights, L2-resident) go straight from global memory to
// registers, a whole column block ahead of their use.  Two independent accumulator chains per tile.
#include "common.h"
#include "split_bf16.h"
#include "stripe_tile.h"

namespace {
using split_bf16::bf16x4;
using split_bf16::bf16x8;
using split_bf16::Tri;

constexpr int D = MTAM_D;
using stripe::f32x16;
using stripe::f32x4;
using stripe::ROWS;
using stripe::store_tile;
using stripe::T_PITCH;
constexpr int A_PITCH = 2 * D + 2;    // staged [item | category] row: [128][gap][128][gap]
constexpr int X_PITCH = D + 2;        // staged x row: [64][gap][64][gap]

struct ChainArgs {
  const float *ic, *W4, *pos;
  int R;
  const float *Wkv, *bkv;
  int n_kv;
  const float *Wx, *bx;
  int n_x;
  float *zr, *x, *kv, *xproj;
  // GATHER: the stripe's [item | category] and position rows come straight from the tables (the four
  // tf.nn.embedding_lookup of Embedding/Behavior_embedding_time_aware_attention.py:68-95 folded in, with the
  // tf.nn.l2_loss sums of Model/base_model.py:302-307): `ic` / `pos` above are not read
  const float *item_table, *cat_table, *pos_table, *user_table;
  int item_rows, cat_rows, pos_rows, user_rows;
  const int32_t *item_ids, *cat_ids, *pos_ids, *user_ids;
  int B, with_user, n_l2;
  float *ic_out, *user_out, *l2_partial;     // ic_out may be NULL (evaluation: nothing reads the rows again)
  float4 *clear_a, *clear_b;                 // the step's gradient accumulators, cleared on the side
  size_t n_a4, n_b4;
  // X3: the three bf16 images of W4, Wkv and Wx (split_bf16::wimg_off layout), written by the optimizer launch
  const uint16_t *img4, *imgkv, *imgx;
};

__device__ __forceinline__ int clamp_row(int id, int rows) { return min(max(id, 0), rows - 1); }

template <bool GATHER>
__global__ __launch_bounds__(256) void seq_chain_fwd_kernel(ChainArgs p) {
  __shared__ __attribute__((aligned(16))) float a_lds[ROWS * A_PITCH];
  __shared__ __attribute__((aligned(16))) float x_lds[ROWS * X_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const long row0 = (long)blockIdx.x * ROWS;
  const int R = p.R;

  // ---- stage 1 operands: this lane's 128 values of W4 column 32 w + r (k = s + 128 h), the stripe of ic
  float b1[D];
#pragma unroll
  for (int s = 0; s < D; ++s) b1[s] = p.W4[(size_t)(s + D * h) * D + 32 * w + r];
  float sq = 0.f;                          // GATHER: this thread's share of sum x^2 over the looked-up rows
  {
    f32x4 v[8];
    if (GATHER) {
      // a thread's 8 pieces are the same 16 bytes (column tid & 63) of 8 rows: all from one table
      const bool is_cat = (tid & 63) >= 32;
      const int32_t *ids = is_cat ? p.cat_ids : p.item_ids;
      const float *tab = is_cat ? p.cat_table : p.item_table;
      const int nrows = is_cat ? p.cat_rows : p.item_rows, off = ((tid & 63) - (is_cat ? 32 : 0)) * 4;
      int id[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) id[i] = ids[min(row0 + 4 * i + w, (long)R - 1)];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4 *>(tab + (size_t)clamp_row(id[i], nrows) * D + off);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long row = row0 + 4 * i + w;
        if (row < R) {
          sq += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
          if (p.ic_out) *reinterpret_cast<f32x4 *>(p.ic_out + row * (2 * D) + (tid & 63) * 4) = v[i];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = i * 256 + tid;
        const long row = min(row0 + (c >> 6), (long)R - 1);
        v[i] = *reinterpret_cast<const f32x4 *>(p.ic + row * (2 * D) + (c & 63) * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = i * 256 + tid, col = (c & 63) * 4;
      float *dst = a_lds + (c >> 6) * A_PITCH + col + (col >> 7);
      dst[0] = v[i].x; dst[1] = v[i].y; dst[2] = v[i].z; dst[3] = v[i].w;
    }
  }
  // position rows of this lane's output elements (column 32 w + r, rows (q & 3) + 8 (q >> 2) + 4 h)
  float pv[16];
  if (GATHER) {
    int pid[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) pid[q] = p.pos_ids[min(row0 + (q & 3) + 8 * (q >> 2) + 4 * h, (long)R - 1)];
#pragma unroll
    for (int q = 0; q < 16; ++q) pv[q] = p.pos_table[(size_t)clamp_row(pid[q], p.pos_rows) * D + 32 * w + r];
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (row0 + (q & 3) + 8 * (q >> 2) + 4 * h < R) sq = fmaf(pv[q], pv[q], sq);
    // the user rows of 32 samples ride with the first ceil(B / 32) stripes (only their sum of squares and, for
    // the scatter-add's L2 term, a copy are needed)
    if ((long)blockIdx.x * 32 < p.B) {
      int uid[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) uid[i] = p.user_ids[min((int)blockIdx.x * 32 + 8 * i + (tid >> 5), p.B - 1)];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ub = blockIdx.x * 32 + 8 * i + (tid >> 5);
        const f32x4 u = *reinterpret_cast<const f32x4 *>(p.user_table + (size_t)clamp_row(uid[i], p.user_rows) * D +
                                                        (tid & 31) * 4);
        if (ub < p.B) {
          *reinterpret_cast<f32x4 *>(p.user_out + (size_t)ub * D + (tid & 31) * 4) = u;
          if (p.with_user) sq += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
        }
      }
    }
    sq = wave_sum(sq);
    if (lane == 0) p.l2_partial[blockIdx.x * 4 + w] = sq;
    const size_t stride = (size_t)gridDim.x * 256, g0 = (size_t)blockIdx.x * 256 + tid;
    for (size_t i = (size_t)gridDim.x * 4 + g0; i < (size_t)p.n_l2; i += stride) p.l2_partial[i] = 0.f;   // unused tail
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = g0; i < p.n_a4; i += stride) p.clear_a[i] = z;
    for (size_t i = g0; i < p.n_b4; i += stride) p.clear_b[i] = z;
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const long row = min(row0 + (q & 3) + 8 * (q >> 2) + 4 * h, (long)R - 1);
      pv[q] = p.pos[row * D + 32 * w + r];
    }
  }
  __syncthreads();

  // ---- z = ic . W4 (K = 256): this wave's 32 columns
  f32x16 zr16, x16;
  {
    f32x16 a0 = {0.f}, a1 = {0.f};
    const float *a = a_lds + r * A_PITCH + (D + 1) * h;
#pragma unroll
    for (int s = 0; s < D; s += 2) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b1[s], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s + 1], b1[s + 1], a1, 0, 0, 0);
    }
    const f32x16 z = a0 + a1;
    const int col = 32 * w + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int rr = (q & 3) + 8 * (q >> 2) + 4 * h;
      zr16[q] = fmaxf(z[q], 0.f);
      x16[q] = zr16[q] + pv[q];
      x_lds[rr * X_PITCH + col + (col >> 6)] = x16[q];
    }
  }
  __syncthreads();
  float *scratch = a_lds + w * (ROWS * T_PITCH);      // the staged input stripe is dead: wave-private scratch
  store_tile(scratch, zr16, p.zr, row0, R, D, 32 * w, lane);
  store_tile(scratch, x16, p.x, row0, R, D, 32 * w, lane);

  // ---- kv = relu(x Wkv + bkv), xproj = x Wx + bx (K = 128): the A fragments of the stripe stay in
  // registers for every column block; column blocks of both outputs are dealt round-robin to the waves
  float af[64];
  {
    const float *a = x_lds + r * X_PITCH + 65 * h;
#pragma unroll
    for (int s = 0; s < 64; ++s) af[s] = a[s];
  }
  const int nb_kv = p.n_kv / 32, nb = nb_kv + p.n_x / 32;
  auto load_block = [&](int j, float (&bw)[64], float &bias) {
    const bool is_kv = j < nb_kv;
    const float *W = is_kv ? p.Wkv : p.Wx;
    const int ldw = is_kv ? p.n_kv : p.n_x, col = 32 * (is_kv ? j : j - nb_kv) + r;
#pragma unroll
    for (int s = 0; s < 64; ++s) bw[s] = W[(size_t)(s + 64 * h) * ldw + col];
    bias = (is_kv ? p.bkv : p.bx)[col];
  };
  auto run_block = [&](int j, const float (&bw)[64], float bias) {
    f32x16 a0 = {0.f}, a1 = {0.f};
#pragma unroll
    for (int s = 0; s < 64; s += 2) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bw[s], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s + 1], bw[s + 1], a1, 0, 0, 0);
    }
    const f32x16 acc = a0 + a1;
    const bool is_kv = j < nb_kv;
    float *out = is_kv ? p.kv : p.xproj;
    const int ldo = is_kv ? p.n_kv : p.n_x, col = 32 * (is_kv ? j : j - nb_kv) + r;
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float v = acc[q] + bias;
      o[q] = is_kv ? fmaxf(v, 0.f) : v;
    }
    store_tile(scratch, o, out, row0, R, ldo, col - r, lane);
  };
  // software pipeline over this wave's blocks j = w, w + 4, ...: block j + 4's weights load while j computes
  float bwA[64], bwB[64], biasA = 0.f, biasB = 0.f;
  if (w < nb) load_block(w, bwA, biasA);
  for (int j = w; j < nb; j += 8) {
    if (j + 4 < nb) load_block(j + 4, bwB, biasB);
    run_block(j, bwA, biasA);
    if (j + 4 < nb) {
      if (j + 8 < nb) load_block(j + 8, bwA, biasA);
      run_block(j + 4, bwB, biasB);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same launch on the bf16 matrix cores: every fp32 product as six v_mfma_f32_32x32x16_bf16 terms of operands
// split three ways (csrc/split_bf16.h; fp32-equivalent).  448 v_mfma_f32_32x32x2_f32 per wave (64 cycles each,
// ~12 us of the launch) become 336 bf16 instructions of 32 cycles (~4.5 us).  The weights arrive ALREADY split and
// laid out as B fragments (three images per matrix, written by the optimizer launch that updates them: no per-step
// prepare launch); the activations are split on the way into LDS (stage 1: the gathered [item | category] rows, as
// 8-byte pieces of three bf16 images) or on the way out of it (stage 2: each wave splits its A fragments of x).
constexpr int A3_PITCH = 2 * (2 * D) + 16;        // bytes per staged row of one image: 256 bf16 + 16 (bank rotation)
constexpr int A3_IMG = ROWS * A3_PITCH;           // one image of the stripe
constexpr int X3_PITCH = D + 4;                   // floats per staged x row

// tools/chain_lab.hip (-DMTAM_CHAIN_STAMPS): s_memrealtime (100 MHz) at the phase boundaries of the two split-bf16
// stripe kernels, wave 0 of the middle workgroup; the product build has none
#ifdef MTAM_CHAIN_STAMPS
__device__ unsigned long long g_chain_stamps[2][16];
#define CH_STAMP(k, i)                                                                             \
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {                                           \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    g_chain_stamps[k][i] = __builtin_amdgcn_s_memrealtime();                                       \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  }
#define CH_WAIT_VM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
#define CH_STAMP(k, i)
#define CH_WAIT_VM
#endif

template <bool GATHER>
__global__ __launch_bounds__(256) void seq_chain_x3_kernel(ChainArgs p) {
  // 50,688 B: the three images of the input stripe; once every wave is through the first product the same bytes
  // hold the fp32 stripe of x (16.9 KB) and, behind it, the four wave-private transposition scratches (18.4 KB)
  __shared__ __attribute__((aligned(16))) unsigned char a_img[3 * A3_IMG];
  float *const x_lds = reinterpret_cast<float *>(a_img);
  static_assert(3 * A3_IMG >= ROWS * X3_PITCH * 4 + 4 * ROWS * T_PITCH * 4, "x stripe + scratch must fit the dead images");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const long row0 = (long)blockIdx.x * ROWS;
  const int R = p.R;

  CH_STAMP(0, 0)
  // The prologue's loads in the order their data is NEEDED -- a wave's loads complete in issue order (vmcnt), so a
  // small dependent load issued behind 48 weight-fragment loads waits for all of them: ids (all of them) first; the
  // looked-up rows as soon as the ids are here; then the first product's B operand (the three images of W4 columns
  // 32 w + r, k = 16 s + 8 h .. + 8: 16 k-steps) and, behind it, this wave's first column block of the second stage,
  // which lands while the first product runs.  tools/chain_lab.hip, middle workgroup of 200: in the order B operand,
  // ids, rows the ids were back after 2.8 us and the first column block took 3.1 us behind its own fetch; now 1.0 and
  // 1.9 us -- but a wave holds at most 64 loads in flight and one CU moves 64 B a clock, so the rows now wait while the
  // 72 weight loads behind them are issued: 16.7 against 17.4 us per launch stand-alone (tools/chain_scale.py), the
  // same 0.2347 ms per step -- not the 3 us the stamps promised.
  const bool has_users = GATHER && (long)blockIdx.x * 32 < p.B;
  const bool is_cat = (tid & 63) >= 32;
  int id[8], pid[16], uid[4];
  if (GATHER) {
    const int32_t *ids = is_cat ? p.cat_ids : p.item_ids;
#pragma unroll
    for (int i = 0; i < 8; ++i) id[i] = ids[min(row0 + 4 * i + w, (long)R - 1)];
#pragma unroll
    for (int q = 0; q < 16; ++q) pid[q] = p.pos_ids[min(row0 + (q & 3) + 8 * (q >> 2) + 4 * h, (long)R - 1)];
    if (has_users) {
#pragma unroll
      for (int i = 0; i < 4; ++i) uid[i] = p.user_ids[min((int)blockIdx.x * 32 + 8 * i + (tid >> 5), p.B - 1)];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 v[8], urow[4];
  float pv[16];
  if (GATHER) {
    const float *tab = is_cat ? p.cat_table : p.item_table;
    const int nrows = is_cat ? p.cat_rows : p.item_rows, off = ((tid & 63) - (is_cat ? 32 : 0)) * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4 *>(tab + (size_t)clamp_row(id[i], nrows) * D + off);
#pragma unroll
    for (int q = 0; q < 16; ++q) pv[q] = p.pos_table[(size_t)clamp_row(pid[q], p.pos_rows) * D + 32 * w + r];
    if (has_users) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        urow[i] = *reinterpret_cast<const f32x4 *>(p.user_table + (size_t)clamp_row(uid[i], p.user_rows) * D + (tid & 31) * 4);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = i * 256 + tid;
      const long row = min(row0 + (c >> 6), (long)R - 1);
      v[i] = *reinterpret_cast<const f32x4 *>(p.ic + row * (2 * D) + (c & 63) * 4);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const long row = min(row0 + (q & 3) + 8 * (q >> 2) + 4 * h, (long)R - 1);
      pv[q] = p.pos[row * D + 32 * w + r];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  CH_STAMP(0, 1)        // ids in, rows requested
  Tri b1[16];
  {
    const uint16_t *base = p.img4 + ((size_t)h * D + 32 * w + r) * 8;
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        b1[s].t[t] = *reinterpret_cast<const bf16x8 *>(base + (size_t)t * (2 * D * D) + (size_t)s * (2 * D * 8));
  }
  const int nb_kv = p.n_kv / 32, nb = nb_kv + p.n_x / 32;
  auto load_block = [&](int j, Tri (&bw)[8], float &bias) {
    const bool is_kv = j < nb_kv;
    const uint16_t *img = is_kv ? p.imgkv : p.imgx;
    const int ldw = is_kv ? p.n_kv : p.n_x, col = 32 * (is_kv ? j : j - nb_kv) + r;
    const uint16_t *base = img + ((size_t)h * ldw + col) * 8;
    const size_t term = (size_t)D * ldw;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        bw[s].t[t] = *reinterpret_cast<const bf16x8 *>(base + t * term + (size_t)s * (2 * ldw * 8));
    bias = (is_kv ? p.bkv : p.bx)[col];
  };
  Tri bwA[8], bwB[8];
  float biasA = 0.f, biasB = 0.f;
  if (w < nb) load_block(w, bwA, biasA);
  __builtin_amdgcn_sched_barrier(0);
  float sq = 0.f;
  if (GATHER) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long row = row0 + 4 * i + w;
      if (row < R) {
        sq += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        if (p.ic_out) *reinterpret_cast<f32x4 *>(p.ic_out + row * (2 * D) + (tid & 63) * 4) = v[i];
      }
    }
  }
  // split on the way into LDS: four floats -> one 8-byte piece of each image (a wave writes 512 contiguous bytes)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = i * 256 + tid;
    const float x4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
    bf16x4 q[3];
    split_bf16::split4(x4, q);
    unsigned char *dst = a_img + (c >> 6) * A3_PITCH + (c & 63) * 8;
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<bf16x4 *>(dst + t * A3_IMG) = q[t];
  }
  CH_STAMP(0, 2)        // rows landed, split, written to LDS
  if (GATHER) {
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (row0 + (q & 3) + 8 * (q >> 2) + 4 * h < R) sq = fmaf(pv[q], pv[q], sq);
    if (has_users) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ub = blockIdx.x * 32 + 8 * i + (tid >> 5);
        const f32x4 u = urow[i];
        if (ub < p.B) {
          *reinterpret_cast<f32x4 *>(p.user_out + (size_t)ub * D + (tid & 31) * 4) = u;
          if (p.with_user) sq += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
        }
      }
    }
    sq = wave_sum(sq);
    if (lane == 0) p.l2_partial[blockIdx.x * 4 + w] = sq;
    const size_t stride = (size_t)gridDim.x * 256, g0 = (size_t)blockIdx.x * 256 + tid;
    for (size_t i = (size_t)gridDim.x * 4 + g0; i < (size_t)p.n_l2; i += stride) p.l2_partial[i] = 0.f;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = g0; i < p.n_a4; i += stride) p.clear_a[i] = z;
    for (size_t i = g0; i < p.n_b4; i += stride) p.clear_b[i] = z;
  }
  CH_STAMP(0, 3)        // position / user rows, clears
  __syncthreads();
  CH_STAMP(0, 4)        // barrier 1

  // ---- z = ic . W4 (K = 256 = 16 k-steps of 16): two accumulator chains, six terms per step
  f32x16 zr16, x16;
  {
    f32x16 a0 = {0.f}, a1 = {0.f};
    const unsigned char *ab = a_img + r * A3_PITCH + h * 16;
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
      Tri fa, fb;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        fa.t[t] = *reinterpret_cast<const bf16x8 *>(ab + t * A3_IMG + s * 32);
        fb.t[t] = *reinterpret_cast<const bf16x8 *>(ab + t * A3_IMG + (s + 1) * 32);
      }
      split_bf16::mfma6x2(fa, b1[s], a0, fb, b1[s + 1], a1);
    }
    const f32x16 z = a0 + a1;
    const int col = 32 * w + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      zr16[q] = fmaxf(z[q], 0.f);
      x16[q] = zr16[q] + pv[q];
    }
    CH_STAMP(0, 6)      // first product done
    // (the first product's B operand is dead: the second column block's weights go out before the x stripe is written)
    if (w + 4 < nb) load_block(w + 4, bwB, biasB);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                        // every wave has read its last fragment of the input images
#pragma unroll
    for (int q = 0; q < 16; ++q) x_lds[((q & 3) + 8 * (q >> 2) + 4 * h) * X3_PITCH + col] = x16[q];
  }
  __syncthreads();
  CH_STAMP(0, 7)        // x in LDS
  float *scratch = x_lds + ROWS * X3_PITCH + w * (ROWS * T_PITCH);
  store_tile(scratch, zr16, p.zr, row0, R, D, 32 * w, lane);
  store_tile(scratch, x16, p.x, row0, R, D, 32 * w, lane);

  // ---- kv = relu(x Wkv + bkv), xproj = x Wx + bx (K = 128 = 8 k-steps): this wave's A fragments of the stripe,
  // split once, stay in registers for every column block
  Tri af[8];
  {
    const float *a = x_lds + r * X3_PITCH + 8 * h;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(a + 16 * s), hi = *reinterpret_cast<const f32x4 *>(a + 16 * s + 4);
      const float x8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      af[s] = split_bf16::split8(x8);
    }
  }
  CH_STAMP(0, 8)        // zr, x stored; A fragments split
  auto run_block = [&](int j, const Tri (&bw)[8], float bias) {
    f32x16 a0 = {0.f}, a1 = {0.f};
#pragma unroll
    for (int s = 0; s < 8; s += 2) split_bf16::mfma6x2(af[s], bw[s], a0, af[s + 1], bw[s + 1], a1);
    const f32x16 acc = a0 + a1;
    const bool is_kv = j < nb_kv;
    float *out = is_kv ? p.kv : p.xproj;
    const int ldo = is_kv ? p.n_kv : p.n_x, col = 32 * (is_kv ? j : j - nb_kv) + r;
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float v = acc[q] + bias;
      o[q] = is_kv ? fmaxf(v, 0.f) : v;
    }
    store_tile(scratch, o, out, row0, R, ldo, col - r, lane);
  };
  for (int j = w; j < nb; j += 8) {
    run_block(j, bwA, biasA);
    CH_STAMP(0, 9 + (j >> 2))        // a column block done (9, 11, ..)
    if (j + 8 < nb) load_block(j + 8, bwA, biasA);
    if (j + 4 < nb) {
      run_block(j + 4, bwB, biasB);
      CH_STAMP(0, 10 + (j >> 2))     // (10, 12, ..)
      if (j + 12 < nb) load_block(j + 12, bwB, biasB);
    }
  }
  CH_WAIT_VM
  CH_STAMP(0, 15)       // stores drained
}

// ---------------------------------------------------------------------------------------------------------------
// The backward's sequence-side chain in ONE launch -- the mirror of the forward kernel above:
//   d_x  = d_x (the decoder's key gradient, already there) + d_xproj . Wx^T + d_kv . Wkv^T + d_xt
//   d_z  = d_x where relu(z) > 0                      (tf.gradients of Embedding/...attention.py:95-103)
//   d_ic = d_z . W4^T                                 (gradient of the looked-up [item | category] rows)
// As two GEMM launches (dual-source ACCUM2_MASK 19.5 us + 8.0 us at 6,400 rows) d_z made a round trip through HBM
// between them.  Here a workgroup owns a 32-row stripe: the concatenated [d_xproj | d_kv] stripe streams through LDS
// in 128-column chunks (split three ways on the way in, double buffered, one barrier per chunk), the d_z stripe
// stays on the CU for the second product, and every product runs as six bf16-MFMA terms.  B operands: the bf16 images
// of the weights' TRANSPOSES, written by the optimizer launch.  (Row-major bf16 copies of W were tried first -- W^T's
// fragments are 8-element runs of W's rows: every load instruction then touched 32 rows, a cache line served four
// k-steps and the 96 KB of lines a chunk keeps in flight thrashed the 32 KB L1: 34 us.)
constexpr int KC = 128;                           // columns per k-chunk (one set of B fragments)
constexpr int BWD_MAX_K = 640;                    // n_x + n_kv the staged stripe is sized for (one decoder block)
constexpr int BA_PITCH = BWD_MAX_K * 2 + 16;      // bytes per row of one image of the staged [d_xproj | d_kv] stripe
constexpr int BA_IMG = ROWS * BA_PITCH;
constexpr int BWD_LDS = 3 * BA_IMG;               // 124,416 B (dynamic)

struct ChainBwdArgs {
  const float *d_xproj, *d_kv, *d_xt, *zr;
  int R, n_x, n_kv;
  float *d_x, *d_z, *d_ic;
  const uint16_t *rimgx, *rimgkv, *rimg4;        // images of the transposes of Wx [D, n_x], Wkv [D, n_kv], W4 [2D, D]
};

// First form (kept in profiles/r03_seq_chain_bwd_chunked_retired.hip.txt): the stripe streamed through LDS in
// 128-column chunks, double buffered, one barrier per chunk -- 31.3 us against 27.5 for the two GEMMs: every chunk
// waited out its own operand round trip (0.64 us of multiplication per chunk against ~2 us of load latency) and
// the barrier tied the four waves to the slowest.  Here the whole stripe is staged ONCE (one barrier), the waves
// run free through the chunks, and a wave's B fragments travel two chunks ahead of their use.
__global__ __launch_bounds__(256) void seq_chain_bwd_kernel(ChainBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float *const z_lds = reinterpret_cast<float *>(lds);                        // after the first product
  static_assert(BWD_LDS >= ROWS * X3_PITCH * 4 + 4 * ROWS * T_PITCH * 4, "d_z stripe + scratch must fit");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const long row0 = (long)blockIdx.x * ROWS;
  const int R = p.R, K1 = p.n_x + p.n_kv, nchunk = K1 / KC;

  // B fragments of chunk c: B[k][n] = W[n][k] for rows n = 32 w + r of Wx (or Wkv), columns k = 128 c' + 16 s + 8 h
  // .. + 8 -- one 16-byte piece of the TRANSPOSE's image, contiguous over the lanes of a half wave
  auto load_b = [&](int c, Tri (&bw)[8]) {
    const bool from_x = c * KC < p.n_x;
    const int ld = from_x ? p.n_x : p.n_kv, k0 = from_x ? c * KC : c * KC - p.n_x;
    const uint16_t *base = (from_x ? p.rimgx : p.rimgkv) + ((size_t)(k0 / 8 + h) * D + 32 * w + r) * 8;
    const size_t term = (size_t)D * ld;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t) bw[s].t[t] = *reinterpret_cast<const bf16x8 *>(base + t * term + (size_t)s * (2 * D * 8));
  };
  Tri bw0[8], bw1[8], bw2[8];
  load_b(0, bw0);
  if (nchunk > 1) load_b(1, bw1);

  // ---- stage the [d_xproj | d_kv] stripe: 32 rows x K1 floats = K1 / 4 sixteen-byte pieces per row, split three
  // ways on the way into LDS (8-byte pieces of three bf16 images)
  {
    // every load first (20 sixteen-byte pieces per thread at K1 = 640), then the splits and the LDS writes
    const int p4 = K1 / 4, nx4 = p.n_x / 4, total = ROWS * p4;
    constexpr int NP = ROWS * (BWD_MAX_K / 4) / 256;
    // id / p4 by a multiply and a shift, exact for id < 32 p4 <= 5,120 (id p4 < 2^20): a division by a run-time value is
    // ~25 VALU instructions, and there were 60 of them per thread in this prologue
    const unsigned inv_p4 = (1u << 20) / (unsigned)p4 + 1u;
    f32x4 v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int id = min(i * 256 + tid, total - 1), row = (int)(((unsigned)id * inv_p4) >> 20), c4 = id - row * p4;
      const long gr = min(row0 + row, (long)R - 1);
      v[i] = c4 < nx4 ? *reinterpret_cast<const f32x4 *>(p.d_xproj + gr * p.n_x + c4 * 4)
                      : *reinterpret_cast<const f32x4 *>(p.d_kv + gr * p.n_kv + (c4 - nx4) * 4);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int id = i * 256 + tid;
      if (id < total) {
        const int row = (int)(((unsigned)id * inv_p4) >> 20), c4 = id - row * p4;
        const float x4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        bf16x4 q[3];
        split_bf16::split4(x4, q);
        unsigned char *dst = lds + row * BA_PITCH + c4 * 8;
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<bf16x4 *>(dst + t * BA_IMG) = q[t];
      }
    }
  }
  __syncthreads();

  f32x16 a0 = {0.f}, a1 = {0.f};
  auto compute = [&](int c, const Tri (&bw)[8]) {
    const unsigned char *ab = lds + r * BA_PITCH + c * (KC * 2) + h * 16;
#pragma unroll
    for (int s = 0; s < 8; s += 2) {
      Tri fa, fb;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        fa.t[t] = *reinterpret_cast<const bf16x8 *>(ab + t * BA_IMG + s * 32);
        fb.t[t] = *reinterpret_cast<const bf16x8 *>(ab + t * BA_IMG + (s + 1) * 32);
      }
      split_bf16::mfma6x2(fa, bw[s], a0, fb, bw[s + 1], a1);
    }
  };
  // ---- d_xproj . Wx^T + d_kv . Wkv^T: no barrier inside; chunk c + 2's fragments are requested before chunk c is
  // multiplied (three register sets in rotation)
  for (int c = 0; c < nchunk; c += 3) {
    if (c + 2 < nchunk) load_b(c + 2, bw2);
    compute(c, bw0);
    if (c + 1 < nchunk) {
      if (c + 3 < nchunk) load_b(c + 3, bw0);
      compute(c + 1, bw1);
    }
    if (c + 2 < nchunk) {
      if (c + 4 < nchunk) load_b(c + 4, bw1);
      compute(c + 2, bw2);
    }
  }
  // epilogue operands in ROW layout (rows 8 i + (lane >> 3), 4 columns of this wave's 32-column block)
  f32x4 cv[4], bv[4], av[4];
  const int gn = 32 * w + 4 * (lane & 7);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long gm = min(row0 + 8 * i + (lane >> 3), (long)R - 1);
    cv[i] = *reinterpret_cast<const f32x4 *>(p.d_x + gm * D + gn);
    bv[i] = *reinterpret_cast<const f32x4 *>(p.d_xt + gm * D + gn);
    av[i] = *reinterpret_cast<const f32x4 *>(p.zr + gm * D + gn);
  }
  // the second product's B fragments (W4 rows 64 w + 32 j + r) can travel now as well
  auto load_b4 = [&](int j, Tri (&bw)[8]) {
    const uint16_t *base = p.rimg4 + ((size_t)h * (2 * D) + 64 * w + 32 * j + r) * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        bw[s].t[t] = *reinterpret_cast<const bf16x8 *>(base + (size_t)t * (2 * D * D) + (size_t)s * (2 * 2 * D * 8));
  };
  load_b4(0, bw0);
  load_b4(1, bw1);
  __syncthreads();      // every wave is past its last fragment read: the stripe becomes d_z + the wave scratches
  float *scratch = z_lds + ROWS * X3_PITCH + w * (ROWS * T_PITCH);
  {
    const f32x16 acc = a0 + a1;
#pragma unroll
    for (int q = 0; q < 16; ++q) scratch[((q & 3) + 8 * (q >> 2) + 4 * h) * T_PITCH + r] = acc[q];
    f32x4 t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = *reinterpret_cast<const f32x4 *>(scratch + (8 * i + (lane >> 3)) * T_PITCH + 4 * (lane & 7));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = 8 * i + (lane >> 3);
      const f32x4 v = t[i] + (cv[i] + bv[i]);
      f32x4 z;
      z.x = av[i].x > 0.f ? v.x : 0.f; z.y = av[i].y > 0.f ? v.y : 0.f;
      z.z = av[i].z > 0.f ? v.z : 0.f; z.w = av[i].w > 0.f ? v.w : 0.f;
      *reinterpret_cast<f32x4 *>(z_lds + lr * X3_PITCH + gn) = z;
      if (row0 + lr < R) {
        *reinterpret_cast<f32x4 *>(p.d_x + (row0 + lr) * D + gn) = v;
        *reinterpret_cast<f32x4 *>(p.d_z + (row0 + lr) * D + gn) = z;
      }
    }
  }
  __syncthreads();

  // ---- d_ic = d_z . W4^T (K = 128): this wave's columns 64 w .. 64 w + 64 of the 256
  Tri af[8];
  {
    const float *a = z_lds + r * X3_PITCH + 8 * h;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(a + 16 * s), hi = *reinterpret_cast<const f32x4 *>(a + 16 * s + 4);
      const float x8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      af[s] = split_bf16::split8(x8);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    f32x16 c0 = {0.f}, c1 = {0.f};
    const Tri(&bw)[8] = j ? bw1 : bw0;
#pragma unroll
    for (int s = 0; s < 8; s += 2) split_bf16::mfma6x2(af[s], bw[s], c0, af[s + 1], bw[s + 1], c1);
    store_tile(scratch, c0 + c1, p.d_ic, row0, R, 2 * D, 64 * w + 32 * j, lane);
  }
}

// W [K, N] fp32 -> the three bf16 images of its TRANSPOSE (W^T [N, K] in the wimg_off layout: element W[k][n] at
// ((n >> 3) K + k) 8 + (n & 7)): the B operands of products with W^T (the backward's stripe kernel)
__global__ __launch_bounds__(256) void split_weight_rows_kernel(const float *__restrict__ W, int K, int N,
                                                                uint16_t *__restrict__ img) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (size_t)K * N) split_bf16::wimg_store(img, N, K, (int)(i % N), (int)(i / N), W[i]);
}

// W [K, N] fp32 -> its three bf16 images (split_bf16::wimg_off): one thread per element
__global__ __launch_bounds__(256) void split_weight_images_kernel(const float *__restrict__ W, int K, int N,
                                                                  uint16_t *__restrict__ img) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (size_t)K * N) split_bf16::wimg_store(img, K, N, (int)(i / N), (int)(i % N), W[i]);
}

// the one buffer the entry points take: [W4 images | Wx images | Wkv images], each 3 K N bf16 (Wkv last: where
// W4's and Wx's images lie does not depend on whether the launch computes the K/V projection)
void set_images(ChainArgs &a, const uint16_t *w_images) {
  if (!w_images) return;
  a.img4 = w_images;
  a.imgx = a.img4 + split_bf16::wimg_elems(2 * D, D);
  a.imgkv = a.imgx + split_bf16::wimg_elems(D, a.n_x);
}

}  // namespace

extern "C" size_t mtam_seq_chain_images_elems(int n_kv, int n_x) {
  return split_bf16::wimg_elems(2 * D, D) + split_bf16::wimg_elems(D, n_kv) + split_bf16::wimg_elems(D, n_x);
}

extern "C" size_t mtam_seq_chain_image_offset(int which, int n_x) {
  return which == 0 ? 0 : split_bf16::wimg_elems(2 * D, D) + (which == 2 ? 0 : split_bf16::wimg_elems(D, n_x));
}

extern "C" int mtam_split_weight_images(const float *W, int K, int N, uint16_t *images, void *stream) {
  MTAM_CHECK_ARG(W && images && K > 0 && N > 0 && K % 8 == 0, "split_weight_images: K must be a positive multiple of 8");
  MTAM_CHECK_ARG(mtam_aligned16(images), "split_weight_images: images must be 16-byte aligned");
  const size_t n = (size_t)K * N;
  hipLaunchKernelGGL(split_weight_images_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), W, K, N, images);
  MTAM_CHECK_LAUNCH("split_weight_images");
  return MTAM_OK;
}

extern "C" int mtam_split_weight_rows(const float *W, int K, int N, uint16_t *images_r, void *stream) {
  MTAM_CHECK_ARG(W && images_r && K > 0 && N > 0 && N % 8 == 0, "split_weight_rows: N must be a positive multiple of 8");
  MTAM_CHECK_ARG(mtam_aligned16(images_r), "split_weight_rows: images must be 16-byte aligned");
  const size_t n = (size_t)K * N;
  hipLaunchKernelGGL(split_weight_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), W, K, N, images_r);
  MTAM_CHECK_LAUNCH("split_weight_rows");
  return MTAM_OK;
}

extern "C" int mtam_seq_chain_bwd_max_k(void) { return BWD_MAX_K; }

extern "C" int mtam_seq_chain_bwd(const float *d_xproj, int n_x, const float *d_kv, int n_kv, const float *d_xt,
                                  const float *zr, int R, float *d_x, float *d_z, float *d_ic,
                                  const uint16_t *w_images_r, void *stream) {
  MTAM_CHECK_ARG(d_xproj && d_xt && zr && d_x && d_z && d_ic && w_images_r && R > 0, "seq_chain_bwd: null argument");
  MTAM_CHECK_ARG(n_x > 0 && n_x % KC == 0 && n_kv >= 0 && n_kv % KC == 0 && n_x + n_kv <= BWD_MAX_K,
                 "seq_chain_bwd: widths must be multiples of %d with n_x + n_kv <= %d (mtam_seq_chain_bwd_max_k)", KC,
                 BWD_MAX_K);
  MTAM_CHECK_ARG(n_kv == 0 || d_kv, "seq_chain_bwd: n_kv > 0 needs d_kv");
  MTAM_CHECK_ARG(mtam_aligned16(d_xproj) && mtam_aligned16(d_kv) && mtam_aligned16(d_xt) && mtam_aligned16(zr) &&
                     mtam_aligned16(d_x) && mtam_aligned16(d_z) && mtam_aligned16(d_ic) && mtam_aligned16(w_images_r),
                 "seq_chain_bwd: operands must be 16-byte aligned");
  ChainBwdArgs a{};
  a.d_xproj = d_xproj; a.d_kv = d_kv; a.d_xt = d_xt; a.zr = zr; a.R = R; a.n_x = n_x; a.n_kv = n_kv;
  a.d_x = d_x; a.d_z = d_z; a.d_ic = d_ic;
  a.rimg4 = w_images_r;
  a.rimgx = a.rimg4 + split_bf16::wimg_elems(2 * D, D);
  a.rimgkv = a.rimgx + split_bf16::wimg_elems(D, n_x);
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(seq_chain_bwd_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    MTAM_CHECK_ARG(e == hipSuccess, "seq_chain_bwd: cannot reserve %d bytes of LDS: %s", BWD_LDS, hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(seq_chain_bwd_kernel, dim3((R + ROWS - 1) / ROWS), dim3(256), BWD_LDS,
                     static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("seq_chain_bwd");
  return MTAM_OK;
}

extern "C" int mtam_seq_chain_fwd(const float *ic, const float *W4, const float *pos, int R, const float *Wkv,
                                  const float *bkv, int n_kv, const float *Wx, const float *bx, int n_x, float *zr,
                                  float *x, float *kv, float *xproj, const uint16_t *w_images, void *stream) {
  MTAM_CHECK_ARG(ic && W4 && pos && Wx && bx && zr && x && xproj && R > 0, "seq_chain_fwd: null argument");
  MTAM_CHECK_ARG(mtam_aligned16(w_images), "seq_chain_fwd: w_images must be 16-byte aligned");
  MTAM_CHECK_ARG(n_kv >= 0 && n_kv % 32 == 0 && n_x > 0 && n_x % 32 == 0, "seq_chain_fwd: widths must be multiples of 32");
  MTAM_CHECK_ARG(n_kv == 0 || (Wkv && bkv && kv), "seq_chain_fwd: n_kv > 0 needs Wkv, bkv and kv");
  MTAM_CHECK_ARG(mtam_aligned16(ic), "seq_chain_fwd: ic must be 16-byte aligned");
  ChainArgs a{};
  a.ic = ic; a.W4 = W4; a.pos = pos; a.R = R; a.Wkv = Wkv; a.bkv = bkv; a.n_kv = n_kv; a.Wx = Wx; a.bx = bx; a.n_x = n_x;
  a.zr = zr; a.x = x; a.kv = kv; a.xproj = xproj;
  set_images(a, w_images);
  if (w_images)
    hipLaunchKernelGGL(seq_chain_x3_kernel<false>, dim3((R + ROWS - 1) / ROWS), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(seq_chain_fwd_kernel<false>, dim3((R + ROWS - 1) / ROWS), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("seq_chain_fwd");
  return MTAM_OK;
}

extern "C" int mtam_seq_chain_gather_partials(int B, int L) { return 4 * ((B * L + ROWS - 1) / ROWS); }

extern "C" int mtam_seq_chain_gather_fwd(const float *item_table, int item_rows, const float *cat_table, int cat_rows,
                                         const float *pos_table, int pos_rows, const float *user_table,
                                         int user_rows, const int32_t *item_ids, const int32_t *cat_ids,
                                         const int32_t *pos_ids, const int32_t *user_ids, int B, int L, int with_user,
                                         const float *W4, const float *Wkv, const float *bkv, int n_kv,
                                         const float *Wx, const float *bx, int n_x, float *ic_out, float *user_out,
                                         float *l2_partial, int n_l2, float *zr, float *x, float *kv, float *xproj,
                                         float *clear_a, size_t n_clear_a, float *clear_b, size_t n_clear_b,
                                         const uint16_t *w_images, void *stream) {
  MTAM_CHECK_ARG(B > 0 && L > 0 && (long)B * L < 0x3fffffffL, "seq_chain_gather_fwd: bad batch shape");
  MTAM_CHECK_ARG(mtam_aligned16(w_images), "seq_chain_gather_fwd: w_images must be 16-byte aligned");
  const int R = B * L;
  MTAM_CHECK_ARG(item_table && cat_table && pos_table && user_table && item_ids && cat_ids && pos_ids && user_ids,
                 "seq_chain_gather_fwd: null table or ids");
  MTAM_CHECK_ARG(item_rows > 0 && cat_rows > 0 && pos_rows > 0 && user_rows > 0, "seq_chain_gather_fwd: empty table");
  MTAM_CHECK_ARG(W4 && Wx && bx && zr && x && xproj && user_out && l2_partial, "seq_chain_gather_fwd: null argument");
  MTAM_CHECK_ARG(n_kv >= 0 && n_kv % 32 == 0 && n_x > 0 && n_x % 32 == 0,
                 "seq_chain_gather_fwd: widths must be multiples of 32");
  MTAM_CHECK_ARG(n_kv == 0 || (Wkv && bkv && kv), "seq_chain_gather_fwd: n_kv > 0 needs Wkv, bkv and kv");
  MTAM_CHECK_ARG(n_l2 >= mtam_seq_chain_gather_partials(B, L), "seq_chain_gather_fwd: l2_partial too short");
  MTAM_CHECK_ARG((B + 31) / 32 <= (R + ROWS - 1) / ROWS, "seq_chain_gather_fwd: L must be at least 1");
  MTAM_CHECK_ARG(mtam_aligned16(item_table) && mtam_aligned16(cat_table) && mtam_aligned16(user_table) &&
                     mtam_aligned16(ic_out) && mtam_aligned16(user_out),
                 "seq_chain_gather_fwd: tables and row outputs must be 16-byte aligned");
  MTAM_CHECK_ARG(n_clear_a % 4 == 0 && n_clear_b % 4 == 0 && mtam_aligned16(clear_a) && mtam_aligned16(clear_b),
                 "seq_chain_gather_fwd: clear ranges must be 16-byte aligned multiples of 4 floats");
  ChainArgs a{};
  a.W4 = W4; a.R = R; a.Wkv = Wkv; a.bkv = bkv; a.n_kv = n_kv; a.Wx = Wx; a.bx = bx; a.n_x = n_x;
  a.zr = zr; a.x = x; a.kv = kv; a.xproj = xproj;
  a.item_table = item_table; a.cat_table = cat_table; a.pos_table = pos_table; a.user_table = user_table;
  a.item_rows = item_rows; a.cat_rows = cat_rows; a.pos_rows = pos_rows; a.user_rows = user_rows;
  a.item_ids = item_ids; a.cat_ids = cat_ids; a.pos_ids = pos_ids; a.user_ids = user_ids;
  a.B = B; a.with_user = with_user; a.n_l2 = n_l2;
  a.ic_out = ic_out; a.user_out = user_out; a.l2_partial = l2_partial;
  a.clear_a = reinterpret_cast<float4 *>(clear_a); a.n_a4 = clear_a ? n_clear_a / 4 : 0;
  a.clear_b = reinterpret_cast<float4 *>(clear_b); a.n_b4 = clear_b ? n_clear_b / 4 : 0;
  set_images(a, w_images);
  if (w_images)
    hipLaunchKernelGGL(seq_chain_x3_kernel<true>, dim3((R + ROWS - 1) / ROWS), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(seq_chain_fwd_kernel<true>, dim3((R + ROWS - 1) / ROWS), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("seq_chain_gather_fwd");
  return MTAM_OK;
}
