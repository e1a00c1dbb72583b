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

namespace {

constexpr int D = MTAM_D;
constexpr int ROWS = 32;              // stripe height
constexpr int A_PITCH = 2 * D + 2;    // staged [item | category] row: [128][gap][128][gap]
constexpr int X_PITCH = D + 2;        // staged x row: [64][gap][64][gap]

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int T_PITCH = 36;           // transposition scratch: 32 rows x (32 + 4) floats per wave
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

}  // namespace

extern "C" int mtam_seq_chain_fwd(const float *ic, const float *W4, const float *pos, int R, const float *Wkv,
                                  const float *bkv, int n_kv, const float *Wx, const float *bx, int n_x, float *zr,
                                  float *x, float *kv, float *xproj, void *stream) {
  MTAM_CHECK_ARG(ic && W4 && pos && Wx && bx && zr && x && xproj && R > 0, "seq_chain_fwd: null argument");
  MTAM_CHECK_ARG(n_kv >= 0 && n_kv % 32 == 0 && n_x > 0 && n_x % 32 == 0, "seq_chain_fwd: widths must be multiples of 32");
  MTAM_CHECK_ARG(n_kv == 0 || (Wkv && bkv && kv), "seq_chain_fwd: n_kv > 0 needs Wkv, bkv and kv");
  MTAM_CHECK_ARG(mtam_aligned16(ic), "seq_chain_fwd: ic must be 16-byte aligned");
  ChainArgs a{};
  a.ic = ic; a.W4 = W4; a.pos = pos; a.R = R; a.Wkv = Wkv; a.bkv = bkv; a.n_kv = n_kv; a.Wx = Wx; a.bx = bx; a.n_x = n_x;
  a.zr = zr; a.x = x; a.kv = kv; a.xproj = xproj;
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
                                         void *stream) {
  MTAM_CHECK_ARG(B > 0 && L > 0 && (long)B * L < 0x3fffffffL, "seq_chain_gather_fwd: bad batch shape");
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
  hipLaunchKernelGGL(seq_chain_fwd_kernel<true>, dim3((R + ROWS - 1) / ROWS), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("seq_chain_gather_fwd");
  return MTAM_OK;
}
