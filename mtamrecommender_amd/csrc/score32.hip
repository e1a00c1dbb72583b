// fp32 catalog scoring without stored logits: the training loss and both scoring gradients of base_model.output
// (Model/base_model.py:300-328, 290-297) in TWO passes over the item table instead of four launches (logits GEMM,
// softmax-CE, dE GEMM, d_pred GEMM) and without any [B, V] buffer.
//
//   lse   scores of a 32-row catalog slab, per-row running (max, sum-exp); a finish launch folds the ranges
//   bwd   recomputes the slab's scores, forms G = (softmax - onehot) * scale in registers, and produces dE (stored,
//         with its squared norm) and the workgroup's share of d_pred (atomics)
//
// Two forms behind the same entry points (mtam_score32_set_split_min_rows picks; the split form is the default):
//   * namespace x3 (second half of this file): every fp32 product as six bf16 MFMA terms of operands split three
//     ways (csrc/split_bf16.h), operands as swizzled bf16 images in LDS, the backward in three wave roles;
//   * the native form below, on v_mfma_f32_32x32x2_f32: lane l (r = l & 31, h = l >> 5) supplies A[row r][k = h] and
//     B[k = h][col r]; k-step s pairs element s (lane half 0) with element s + 64 (lane half 1) of a 128-long
//     contraction, so a lane's resident operand is 64 contiguous floats, and a staged row keeps its two halves one
//     float apart ([64][gap][64][gap], 130 floats): the 64 lanes of a fragment read then hit 64 distinct LDS banks.
//     Two independent accumulator chains per tile (a dependent fp32 MFMA issues every ~84 cycles, an independent one
//     every 64).
// Evaluation keeps the stored-logits GEMM (its k-ordered fmaf chain is the ranking contract).
#include "common.h"
#include "split_bf16.h"
#include <stdlib.h>

// In-kernel stamps for tools/score_lab.hip (a diagnostic build, -DMTAM_SCORE_STAMPS): timer ticks per segment of a
// slab iteration, summed per wave of the middle workgroup and written to a buffer of their own.  The product build
// has none.
#ifdef MTAM_SCORE_STAMPS
__device__ unsigned long long g_score_stamps[2][12][8];     // [kernel][wave][segment]; segment 7 = s_memrealtime ticks
__device__ unsigned long long g_score_wg[2][4096][3];       // [kernel][workgroup]: start, end (s_memrealtime), HW_ID | XCC_ID << 32
#define SC_STAMP_DECL unsigned long long st_last_ = 0, st_real_ = 0, st_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SC_STAMP_START                                                                 \
  st_real_ = __builtin_amdgcn_s_memrealtime();                                         \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last_)::"memory");
#define SC_STAMP(i)                                                                    \
  {                                                                                    \
    unsigned long long t_;                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    st_acc_[i] += t_ - st_last_;                                                       \
    st_last_ = t_;                                                                     \
  }
#define SC_WAIT_VM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define SC_STAMP_DUMP(k)                                                               \
  st_acc_[7] = __builtin_amdgcn_s_memrealtime() - st_real_;                            \
  if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {                      \
    g_score_wg[k][blockIdx.x][0] = st_real_;                                           \
    g_score_wg[k][blockIdx.x][1] = st_real_ + st_acc_[7];                              \
    g_score_wg[k][blockIdx.x][2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |      \
                                   ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); \
  }                                                                                    \
  if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && (threadIdx.x & 63) == 0)       \
    for (int i_ = 0; i_ < 8; ++i_) g_score_stamps[k][threadIdx.x >> 6][i_] = st_acc_[i_];
#else
#define SC_STAMP_DECL
#define SC_STAMP_START
#define SC_STAMP(i)
#define SC_WAIT_VM
#define SC_STAMP_DUMP(k)
#endif

namespace {

constexpr int D = MTAM_D;
constexpr int SLAB = 32;        // catalog rows per iteration
constexpr int BT = 128;         // batch rows per tile
constexpr int PITCH = 130;      // floats per staged 128-float row
constexpr int GT_PITCH = 34;    // floats per G^T row: [16][gap][16][gap]
constexpr float L2E = 1.4426950408889634f;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
__device__ __forceinline__ int gap64(int c) { return c + (c >> 6); }      // position of element c in a gapped 128-row
__device__ __forceinline__ int gap16(int c) { return c + (c >> 4); }      // ... in a gapped 32-row

struct Stage {
  f32x4 v[4];
};
// 32 rows x 512 B: 4 x 16 B per thread, every wave-instruction reads 1 KiB contiguous; rows past the end
// re-read the last row (masked later)
__device__ __forceinline__ void stage_load(Stage &st, const float *__restrict__ E, int v0, int V, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const long v = min(v0 + (c >> 5), V - 1);
    st.v[i] = *reinterpret_cast<const f32x4 *>(E + v * D + (c & 31) * 4);
  }
}
__device__ __forceinline__ void stage_store(const Stage &st, float *buf, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    float *dst = buf + (c >> 5) * PITCH + gap64((c & 31) * 4);
    dst[0] = st.v[i].x; dst[1] = st.v[i].y; dst[2] = st.v[i].z; dst[3] = st.v[i].w;
  }
}
// this lane's 64 contiguous floats of a 128-float row (elements 64 h .. 64 h + 63)
__device__ __forceinline__ void load_half_row(float (&p)[64], const float *__restrict__ row, int h) {
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const f32x4 x = *reinterpret_cast<const f32x4 *>(row + 64 * h + 4 * q);
    p[4 * q] = x.x; p[4 * q + 1] = x.y; p[4 * q + 2] = x.z; p[4 * q + 3] = x.w;
  }
}
// S[v][b] = sum_d E[v][d] P[b][d] for the slab's 32 rows and this wave's 32 batch rows
__device__ __forceinline__ f32x16 slab_scores(const float *e_lds, const float (&p1)[64], int r, int h) {
  f32x16 a0 = {0.f}, a1 = {0.f};
  const float *e = e_lds + r * PITCH + 65 * h;
#pragma unroll
  for (int s = 0; s < 64; s += 2) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(e[s], p1[s], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(e[s + 1], p1[s + 1], a1, 0, 0, 0);
  }
  return a0 + a1;
}

// ------------------------------------------------------------------ forward: log-sum-exp without logits
__global__ __launch_bounds__(256) void score32_lse_kernel(const float *__restrict__ E, const float *__restrict__ P,
                                                          int V, int B, int slabs_per_wg,
                                                          float *__restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float e_lds[SLAB * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int chunks = gridDim.x, c = blockIdx.x;
  const long b = (long)blockIdx.y * BT + 32 * w + r;
  const int nslab = (V + SLAB - 1) / SLAB;
  const int slab0 = min(c * slabs_per_wg, nslab), slab1 = min(nslab, slab0 + slabs_per_wg);
  float p1[64];
  load_half_row(p1, P + min(b, (long)B - 1) * D, h);
  float m = -INFINITY, ssum = 0.f;
  Stage st;
  if (slab0 < slab1) stage_load(st, E, slab0 * SLAB, V, tid);
  for (int sl = slab0; sl < slab1; ++sl) {
    stage_store(st, e_lds, tid);
    __syncthreads();
    if (sl + 1 < slab1) stage_load(st, E, (sl + 1) * SLAB, V, tid);
    f32x16 acc = slab_scores(e_lds, p1, r, h);
    const int vbase = sl * SLAB + 4 * h;
    const int vlim = (sl * SLAB + SLAB <= V) ? 0x7fffffff : V;
    float mx = m;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int v = vbase + (q & 3) + 8 * (q >> 2);
      const float x = (v < vlim) ? acc[q] : -INFINITY;
      acc[q] = x;
      mx = fmaxf(mx, x);
    }
    const float ref = (mx == -INFINITY) ? 0.f : mx;
    const float nref = -ref * L2E;
    float add = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) add += fast_exp2(fmaf(acc[q], L2E, nref));
    ssum = ssum * fast_exp2(fmaf(m, L2E, nref)) + add;
    m = mx;
    __syncthreads();
  }
  const float m2 = __shfl_xor(m, 32, 64), s2 = __shfl_xor(ssum, 32, 64);
  const float mm = fmaxf(m, m2), ref = (mm == -INFINITY) ? 0.f : mm;
  const float ss = ssum * fast_exp2((m - ref) * L2E) + s2 * fast_exp2((m2 - ref) * L2E);
  if (h == 0 && b < B) {
    partial[((size_t)b * chunks + c) * 2 + 0] = mm;
    partial[((size_t)b * chunks + c) * 2 + 1] = ss;
  }
}

// row0 >= 0 (a ROW RANGE of the catalog, data-parallel row-sharded scoring): E is the range's first row, targets are
// catalog row numbers; a target outside [row0, row0 + V) has no row here -- its logit comes out 0 -- and `ce` receives
// the TARGET LOGIT (not lse - logit): the ranks' (lse, logit) pairs are combined by the caller.  row0 < 0: the whole
// catalog, targets clamped, ce = lse - logit.
__device__ __forceinline__ int local_target(int t, int row0, int V) {
  if (row0 < 0) return min(max(t, 0), V - 1);
  t -= row0;
  return (t >= 0 && t < V) ? t : -1;
}
__global__ __launch_bounds__(256) void score32_finish_kernel(const float *__restrict__ E, const float *__restrict__ P,
                                                             const int32_t *__restrict__ target, int V, int chunks,
                                                             const float *__restrict__ partial,
                                                             float *__restrict__ lse, float *__restrict__ ce,
                                                             int row0) {
  __shared__ float red[4], red2[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *pp = partial + (size_t)b * chunks * 2;
  // all loads first: the target row and this row's partials
  const int tl = local_target(target[b], row0, V);
  const long t = max(tl, 0);
  float dot = (tid < D && tl >= 0) ? P[(size_t)b * D + tid] * E[t * D + tid] : 0.f;
  float m = -INFINITY;
  for (int c = tid; c < chunks; c += 256) m = fmaxf(m, pp[2 * c]);
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < chunks; c += 256) s += pp[2 * c + 1] * expf(pp[2 * c] - m);
  s = wave_sum(s);
  dot = wave_sum(dot);
  if ((tid & 63) == 0) {
    red[tid >> 6] = s;
    red2[tid >> 6] = dot;
  }
  __syncthreads();
  if (tid == 0) {
    const float l = m + logf((red[0] + red[1]) + (red[2] + red[3]));
    lse[b] = l;
    const float logit = (red2[0] + red2[1]) + (red2[2] + red2[3]);
    ce[b] = row0 < 0 ? l - logit : logit;
  }
}

// ------------------------------------------------------------------ backward: G, d_pred and dE in one pass
struct BwdArgs {
  const float *E, *P;         // P, lse, target, d_pred: already moved to this launch's 128-row batch tile
  const float *lse;
  const int32_t *target;
  int V, Bt, slabs_per_wg;
  float scale;
  float *d_pred, *dE, *sq_partial;
  int row0;                   // >= 0: E is a row range starting at catalog row row0 (see local_target)
};

template <bool RMW>
__global__ __launch_bounds__(256) void score32_bwd_kernel(BwdArgs p) {
  __shared__ __attribute__((aligned(16))) float e_lds[SLAB * PITCH];
  __shared__ __attribute__((aligned(16))) float g_lds[SLAB * PITCH];      // G[v][b], gapped at b = 64
  __shared__ __attribute__((aligned(16))) float gt_lds[BT * GT_PITCH];    // G^T[b][v], gapped at v = 16
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int V = p.V;
  const int nslab = (V + SLAB - 1) / SLAB;
  const int slab0 = min((int)blockIdx.x * p.slabs_per_wg, nslab), slab1 = min(nslab, slab0 + p.slabs_per_wg);
  const int dcol = 32 * w + r;        // this lane's output column in both backward products
  const int bcol = 32 * w + r;        // the batch row (of the tile) whose scores sit on this lane
  const bool valid_b = bcol < p.Bt;
  const int brow = min(bcol, p.Bt - 1);
  // a batch row that does not exist gets c_b = -inf and no target: G = 0 without a mask
  const float c_b = valid_b ? fmaf(-p.lse[brow], L2E, log2f(p.scale)) : -INFINITY;
  const int t_b = valid_b ? local_target(p.target[brow], p.row0, V) : -1;
  float p1[64], p2[64];
  load_half_row(p1, p.P + (size_t)brow * D, h);
  // pred[b = s + 64 h][d = dcol]: the k operand of dE (rows past the tile: any valid row, their G is 0)
#pragma unroll
  for (int s = 0; s < 64; ++s) p2[s] = p.P[(size_t)min(s + 64 * h, p.Bt - 1) * D + dcol];
  f32x16 dp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dp[i] = f32x16{0.f};
  float sq = 0.f;

  Stage st;
  if (slab0 < slab1) stage_load(st, p.E, slab0 * SLAB, V, tid);
  for (int sl = slab0; sl < slab1; ++sl) {
    stage_store(st, e_lds, tid);
    __syncthreads();
    if (sl + 1 < slab1) stage_load(st, p.E, (sl + 1) * SLAB, V, tid);
    const int vbase = sl * SLAB;
    const bool full = vbase + SLAB <= V;
    const int vlim = full ? 0x7fffffff : V;

    // ---- scores of this wave's 32 batch rows, G to LDS in both orientations
    {
      const f32x16 acc = slab_scores(e_lds, p1, r, h);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
        const int v = vbase + row;
        float g = fast_exp2(fmaf(acc[q], L2E, c_b)) - ((v == t_b) ? p.scale : 0.f);
        g = (v < vlim) ? g : 0.f;
        g_lds[row * PITCH + gap64(bcol)] = g;
        gt_lds[bcol * GT_PITCH + gap16(row)] = g;
      }
    }
    __syncthreads();

    // ---- dE[v][d] = sum_b G[v][b] pred[b][d]   (this wave: columns d = 32 w ..)
    {
      f32x16 a0 = {0.f}, a1 = {0.f};
      const float *g = g_lds + r * PITCH + 65 * h;
#pragma unroll
      for (int s = 0; s < 64; s += 2) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(g[s], p2[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(g[s + 1], p2[s + 1], a1, 0, 0, 0);
      }
      f32x16 acc = a0 + a1;
      float *const out = p.dE + ((size_t)vbase + 4 * h) * D + dcol;
      if (RMW) {
        float old[16];
#pragma unroll
        for (int q = 0; q < 16; ++q)
          old[q] = out[(long)min((q & 3) + 8 * (q >> 2), V - 1 - vbase - 4 * h) * D];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += old[q];
      }
      if (full) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          sq = fmaf(acc[q], acc[q], sq);
          out[(size_t)((q & 3) + 8 * (q >> 2)) * D] = acc[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2);
          if (vbase + row + 4 * h < V) {
            sq = fmaf(acc[q], acc[q], sq);
            out[(size_t)row * D] = acc[q];
          }
        }
      }
    }

    // ---- d_pred[b][d] += sum_v G[v][b] E[v][d]   (this wave: columns d = 32 w .., all 128 batch rows)
    {
      const float *eb = e_lds + (16 * h) * PITCH + gap64(dcol);
      const float *ga = gt_lds + r * GT_PITCH + 17 * h;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float bv = eb[s * PITCH];
#pragma unroll
        for (int mblk = 0; mblk < 4; ++mblk)
          dp[mblk] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[(32 * mblk) * GT_PITCH + s], bv, dp[mblk], 0, 0, 0);
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int mblk = 0; mblk < 4; ++mblk)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int b = 32 * mblk + acc_row(q, h);
      if (b < p.Bt) atomicAdd(p.d_pred + b * D + dcol, dp[mblk][q]);
    }
  if (p.sq_partial) {
    sq = wave_sum(sq);
    if (lane == 0) p.sq_partial[(size_t)blockIdx.x * 4 + w] = sq;
  }
}

// ==================================================================== large catalogs: fp32 through bf16 MFMA
// Above a few tens of thousands of rows the two kernels above are bound by the fp32 matrix rate: 4 x 2 B D V flops
// per step at 157 TFLOP/s (8.4 ms at 10 M rows; measured 11.8).  v_mfma_f32_32x32x16_bf16 runs 16x faster, and a
// product of two fp32 numbers can be put on it WITHOUT giving up fp32 accuracy: split every operand into three
// bf16 terms, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 3 x 8 significant bits =
// fp32's 24, the split is exact), and keep the six products of weight >= 2^-16,
//     a b  ~=  a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// each exact in fp32 (8 x 8 bits), accumulated in fp32 by the MFMA.  What is dropped (a2 b3 + a3 b2 + a3 b3) is
// <= ~2^-23 |a b|: the size of the rounding an fp32 fma commits anyway.  6/16 of the native fp32 MFMA time, and the
// passes become HBM-bound (the table is read as fp32: 512 B per row and pass).  Same slab / range / flush structure
// as above; operands are split on the fly (E per slab into three LDS images, pred once into registers, G per slab),
// fragment layouts as in csrc/score16.hip.  Results agree with the native-fp32 kernels to ~1e-7 relative (tested
// against float64); the evaluation path keeps the k-ordered fmaf chain (the top-K contract).
namespace x3 {

using namespace split_bf16;

// Which slabs a workgroup takes.  BLOCKED (slabs_per_wg > 0): a contiguous range.  CYCLIC (slabs_per_wg <= 0):
// slabs c, c + G, c + 2 G, ..: at any moment the grid reads ONE contiguous window of G x 16 KB instead of G streams a
// fixed multiple of 128 KB apart, which advance in step and so lean on the same memory channels together.
struct SlabWalk {
  int first, step, n;
};
__device__ __forceinline__ SlabWalk slab_walk(int c, int G, int slabs_per_wg, int nslab) {
  if (slabs_per_wg > 0) {
    const int s0 = min(c * slabs_per_wg, nslab);
    return {s0, 1, min(nslab, s0 + slabs_per_wg) - s0};
  }
  return {c, G, c < nslab ? (nslab - c + G - 1) / G : 0};
}


// the wave's 32 batch rows as an operand (row b, k = d = 16 s + 8 h + j), split
__device__ __forceinline__ void load_pred_rows(Tri (&p1)[8], const float *__restrict__ row, int h) {
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(row + 16 * s + 8 * h);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(row + 16 * s + 8 * h + 4);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    p1[s] = split8(x);
  }
}
// ---- LDS images of the split operands ([rows][128 x bf16], 256-byte rows; G^T: [128 b][32 v], 64-byte rows): one
// image serves row reads (ds_read_b128) and transposed reads (ds_read_b64_tr_b16) through a chunk swizzle
constexpr int E2_IMG = SLAB * 256;       // one of the three images of a 32-row slab
constexpr int GT2_IMG = BT * 64;         // G^T[b][v]: 128 rows of 32 v

__device__ __forceinline__ int e_off(int row, int ch) {      // 16-byte chunk ch (0..15) of image row `row`
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
__device__ __forceinline__ int gt_off(int b, int ch) {       // 16-byte chunk ch (0..3) of G^T row b
  return 64 * b + 16 * (ch ^ ((b >> 2) & 3));
}
// staged fp32 slab (4 x 16 B per thread of a 256-thread role) -> three swizzled bf16 images
__device__ __forceinline__ void stage_store_split_tr(const Stage &st, unsigned char *e_img, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid, row = c >> 5, piece = c & 31;
    const float x[4] = {st.v[i].x, st.v[i].y, st.v[i].z, st.v[i].w};
    bf16x4 q[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __bf16 a, b, cc;
      split3(x[j], a, b, cc);
      q[0][j] = a; q[1][j] = b; q[2][j] = cc;
    }
    unsigned char *const dst = e_img + e_off(row, piece >> 1) + 8 * (piece & 1);
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<bf16x4 *>(dst + t * E2_IMG) = q[t];
  }
}

// Forward: the log-sum-exp of the scores, one role.  4 waves take a slab in turn through split, barrier, 48 MFMAs,
// (max, sum-exp), barrier: 5,056 cycles a slab per wave with 1,536 of MFMA issue, 2-3 workgroups per CU; 2.27 ms at
// 10 M rows.  Two forms with the work cut by role (S waves scoring slab i with the fold of slab i - 1 between their
// MFMAs, one or two groups of loader waves splitting slab i + 1) measured the same 2.3-2.4 ms
// (profiles/r02_score32_lse_two_role_retired.hip.txt): the matrix pipe is 56-67 % busy in all of them and the clock
// sits at 1.7 GHz.
__global__ __launch_bounds__(256) void lse_kernel(const float *__restrict__ E, const float *__restrict__ P, int V,
                                                  int B, int slabs_per_wg, float *__restrict__ partial) {
  __shared__ __attribute__((aligned(16))) unsigned char e_img[3 * E2_IMG];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int chunks = gridDim.x, c = blockIdx.x;
  const long b = (long)blockIdx.y * BT + 32 * w + r;
  const int nslab = (V + SLAB - 1) / SLAB;
  const SlabWalk wk = slab_walk(c, chunks, slabs_per_wg, nslab);
  Tri p1[8];
  load_pred_rows(p1, P + min(b, (long)B - 1) * D, h);
  const int swz_r = ((r & 3) << 2) | ((r >> 2) & 3);
  float m = -INFINITY, ssum = 0.f;
  Stage st;
  SC_STAMP_DECL
  if (wk.n > 0) stage_load(st, E, wk.first * SLAB, V, tid);
  SC_STAMP_START
  for (int i = 0; i < wk.n; ++i) {
    const int sl = wk.first + i * wk.step;
    SC_WAIT_VM
    SC_STAMP(0)      // the slab's rows landed
    stage_store_split_tr(st, e_img, tid);
    SC_STAMP(1)      // split + LDS writes
    __syncthreads();
    SC_STAMP(2)      // barrier 1
    if (i + 1 < wk.n) stage_load(st, E, (sl + wk.step) * SLAB, V, tid);
    f32x16 acc = {0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      Tri a;
      const int o = 256 * r + 16 * ((2 * s + h) ^ swz_r);
#pragma unroll
      for (int t = 0; t < 3; ++t) a.t[t] = *reinterpret_cast<const bf16x8 *>(e_img + t * E2_IMG + o);
      acc = mfma6(a, p1[s], acc);
    }
    SC_STAMP(3)      // next loads issued, 48 MFMAs (+ fragment reads)
    const int vbase = sl * SLAB + 4 * h;
    if (sl * SLAB + SLAB > V) {              // the catalog's last slab (wave-uniform)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = (vbase + (q & 3) + 8 * (q >> 2) < V) ? acc[q] : -INFINITY;
    }
    float mx = m;
#pragma unroll
    for (int q = 0; q < 16; ++q) mx = fmaxf(mx, acc[q]);
    const float ref = (mx == -INFINITY) ? 0.f : mx;
    const float nref = -ref * L2E;
    float add = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) add += fast_exp2(fmaf(acc[q], L2E, nref));
    ssum = ssum * fast_exp2(fmaf(m, L2E, nref)) + add;
    m = mx;
    SC_STAMP(4)      // max, exp, sum
    __syncthreads();
    SC_STAMP(5)      // barrier 2
  }
  SC_STAMP_DUMP(0)
  const float m2 = __shfl_xor(m, 32, 64), s2 = __shfl_xor(ssum, 32, 64);
  const float mm = fmaxf(m, m2), ref = (mm == -INFINITY) ? 0.f : mm;
  const float ss = ssum * fast_exp2((m - ref) * L2E) + s2 * fast_exp2((m2 - ref) * L2E);
  if (h == 0 && b < B) {
    partial[((size_t)b * chunks + c) * 2 + 0] = mm;
    partial[((size_t)b * chunks + c) * 2 + 1] = ss;
  }
}

// ---- backward: G, d_pred and dE of a slab with the transposes done by the LDS read (ds_read_b64_tr_b16).
// Two earlier forms are kept, not built, in profiles/r02_score32_bwd_variants_retired.hip.txt: producer / consumer
// waves with G[v][b] and G^T[b][v] images, two-byte LDS accesses and three barriers a slab (9,400 cycles a slab against
// 4,608 of MFMA issue per SIMD, tools/score_lab.hip), and a two-role form of what follows (7,240).
//   * the score tile S[v][b] of a wave has its rows v in the 16 registers and its column b on the lane.
//     dE[v][d] = sum_b G[v][b] pred[b][d] sums over the COLUMN index: the lane writes its registers as 8-byte pieces of
//     a G^T[b][v] image (64-byte rows) and the dE role reads that image back transposed, as its A operand;
//   * d_pred[b][d] += sum_v G[v][b] E[v][d] takes A = row reads of the same G^T image and B = transposed reads of
//     the row-major E images;
//   * so there is no G[v][b] image, no two-byte LDS access, and ONE barrier per slab.
// Images ([rows][128 x bf16], 256-byte rows; G^T: 64-byte rows) serve row reads (ds_read_b128) and transposed reads
// alike through the chunk swizzles of e_off() / gt_off(); both kinds of read and the writes are bank-conflict free.
constexpr int T_PITCH = 36;             // per dE wave: a 32 x 32 fp32 tile on its way out, rows 36 floats apart
constexpr int OUT_SCRATCH = 4 * 32 * T_PITCH * 4;
// Three roles of 48 MFMAs a slab each, three waves per SIMD (768 threads, <= 168 registers each):
//     waves 0..3  (S)  scores(i) <- e[i % 3];  G(i) -> gt[i & 1];  split E(i + 1) -> e[(i + 1) % 3];  fetch E(i + 2)
//     waves 4..7  (D)  d_pred(i - 1) <- gt[(i - 1) & 1], e[(i - 1) % 3]
//     waves 8..11 (T)  dE(i - 1) <- gt[(i - 1) & 1]
// E images triple-buffered, G^T double (138 KB of LDS).  6,330 cycles a slab, the matrix pipe 76 % busy
// (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / 4, profiles/r02_score32_sq_counters.md); the staging placed with
// D, or shared by S and D, gives the same slab time.  The clock falls as the pipe fills: 1.83 GHz in the
// three-barrier form, 1.59 GHz here (s_memtime against s_memrealtime), and v_mfma_f32_32x32x16_bf16 alone on random
// operands holds 1.9 GHz = 1.87 PFLOP/s (tools/mfma_lab.hip), not the 2.5 of constant operands.
constexpr int TR3_LDS = 3 * 3 * E2_IMG + 2 * 3 * GT2_IMG + OUT_SCRATCH;      // 141,312 B

template <bool RMW>
__global__ __launch_bounds__(768) void bwd_tr3_kernel(BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *const e_buf = lds;                                   // [3][3][E2_IMG]
  unsigned char *const gt_buf = lds + 3 * 3 * E2_IMG;                 // [2][3][GT2_IMG]
  float *const t_scratch = reinterpret_cast<float *>(gt_buf + 2 * 3 * GT2_IMG);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int role = wave >> 2, w = wave & 3;
  const int V = p.V;
  const int nslab = (V + SLAB - 1) / SLAB;
  const SlabWalk wk = slab_walk(blockIdx.x, gridDim.x, p.slabs_per_wg, nslab);
  const int slab0 = wk.first, step = wk.step, n = wk.n;
  if (n <= 0) {                      // (a workgroup past the end of the catalog: nothing to add, nothing to sum)
    if (role == 2 && p.sq_partial && lane == 0) p.sq_partial[(size_t)blockIdx.x * 4 + w] = 0.f;
    return;
  }
  const int cg = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;      // transposed-read lane roles

  if (role == 0) {
    // ------------------------------------------------------------- S waves: scores and G (batch rows 32 w ..)
    const int bcol = 32 * w + r;
    const bool valid_b = bcol < p.Bt;
    const int brow = min(bcol, p.Bt - 1);
    const float c_b = valid_b ? fmaf(-p.lse[brow], L2E, log2f(p.scale)) : -INFINITY;
    const int t_b = valid_b ? local_target(p.target[brow], p.row0, V) : -1;
    Tri p1[8];
    load_pred_rows(p1, p.P + (size_t)brow * D, h);
    const int swz_r = ((r & 3) << 2) | ((r >> 2) & 3);
    int gt_wr[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) gt_wr[q4] = gt_off(bcol, q4) + 8 * h;
    Stage st;
    stage_load(st, p.E, slab0 * SLAB, V, tid);
    stage_store_split_tr(st, e_buf, tid);
    if (n > 1) stage_load(st, p.E, (slab0 + step) * SLAB, V, tid);
    SC_STAMP_DECL
    __syncthreads();       // (slab 0 staged)
    SC_STAMP_START
    for (int i = 0; i <= n; ++i) {
      if (i < n) {
        const unsigned char *const e_cur = e_buf + (i % 3) * 3 * E2_IMG;
        unsigned char *const gt_cur = gt_buf + (i & 1) * 3 * GT2_IMG;
        f32x16 acc = {0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          Tri a;
          const int o = 256 * r + 16 * ((2 * s + h) ^ swz_r);
#pragma unroll
          for (int t = 0; t < 3; ++t) a.t[t] = *reinterpret_cast<const bf16x8 *>(e_cur + t * E2_IMG + o);
          acc = mfma6(a, p1[s], acc);
        }
        SC_STAMP(0)
        const int vbase = (slab0 + i * step) * SLAB;
        // the target row of a lane lies in this slab for 128 of the catalog's slabs, and only the last slab is ragged:
        // both corrections are applied under a wave-uniform test, outside the exp / split chains
        float g[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) g[q] = fast_exp2(fmaf(acc[q], L2E, c_b));
        const int t_rel = t_b - vbase - 4 * h;                 // the target as a row of this lane's 16: (q & 3) + 8 (q >> 2)
        if (__builtin_amdgcn_ballot_w64(t_rel >= 0 && t_rel < SLAB - 4 * h) != 0 || vbase + SLAB > V) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2);
            g[q] -= (row == t_rel) ? p.scale : 0.f;
            g[q] = (vbase + row + 4 * h < V) ? g[q] : 0.f;
          }
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          bf16x4 gq[3];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            __bf16 a, b, c;
            split3(g[4 * q4 + k], a, b, c);
            gq[0][k] = a; gq[1][k] = b; gq[2][k] = c;
          }
#pragma unroll
          for (int t = 0; t < 3; ++t) *reinterpret_cast<bf16x4 *>(gt_cur + t * GT2_IMG + gt_wr[q4]) = gq[t];
        }
        SC_STAMP(1)
      }
      if (i + 1 < n) {
        SC_WAIT_VM
        SC_STAMP(6)
        stage_store_split_tr(st, e_buf + ((i + 1) % 3) * 3 * E2_IMG, tid);
        if (i + 2 < n) stage_load(st, p.E, (slab0 + (i + 2) * step) * SLAB, V, tid);
        SC_STAMP(2)
      }
      __syncthreads();
      SC_STAMP(3)
    }
    SC_STAMP_DUMP(1)
    return;
  }

  if (role == 1) {
    // ------------------------------------------------------------- D waves: staging and d_pred (columns d = 32 w ..)
    f32x16 dp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dp[i] = f32x16{0.f};
    // E^T fragment of (k-step s', half jj): rows v = 16 s' + 8 h + 4 jj + q, columns d = 32 w + 16 cg + 4 pp ..
    int et_off[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) et_off[jj] = e_off(8 * h + 4 * jj + tq, 4 * w + 2 * cg + (tp >> 1)) + 8 * (tp & 1);
    // G^T row fragment of (block mblk, k-step s'): chunk 2 s' + h of row b = 32 mblk + r
    int ga_off[2];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) ga_off[sp] = gt_off(r, 2 * sp + h);
    SC_STAMP_DECL
    __syncthreads();
    SC_STAMP_START
    for (int i = 0; i <= n; ++i) {
      if (i >= 1) {
        const unsigned char *const e_prev = e_buf + ((i - 1) % 3) * 3 * E2_IMG;
        const unsigned char *const gt_prev = gt_buf + ((i - 1) & 1) * 3 * GT2_IMG;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          Tri bf;
#pragma unroll
          for (int t = 0; t < 3; ++t)
            bf.t[t] = lds_tr8(e_prev + t * E2_IMG + 4096 * sp + et_off[0], e_prev + t * E2_IMG + 4096 * sp + et_off[1]);
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) {
            Tri a;
#pragma unroll
            for (int t = 0; t < 3; ++t)
              a.t[t] = *reinterpret_cast<const bf16x8 *>(gt_prev + t * GT2_IMG + 2048 * mb + ga_off[sp]);
            dp[mb] = mfma6(a, bf, dp[mb]);
          }
        }
      }
      SC_STAMP(0)
      __syncthreads();
      SC_STAMP(3)
    }
    SC_STAMP_DUMP(1)
    const int dcol = 32 * w + r;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int b = 32 * mb + acc_row(q, h);
        if (b < p.Bt) atomicAdd(p.d_pred + b * D + dcol, dp[mb][q]);
      }
    return;
  }

  // ------------------------------------------------------------------ T waves: dE (columns d = 32 w ..)
  const int dcol = 32 * w + r;
  Tri p2[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = p.P[(size_t)min(16 * s + 8 * h + j, p.Bt - 1) * D + dcol];
    p2[s] = split8(x);
  }
  // G^T fragment of (k-step s, half jj): rows b = 16 s + 8 h + 4 jj + q, columns v = 16 cg + 4 pp ..
  int gt_rd[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) gt_rd[jj] = gt_off(8 * h + 4 * jj + tq, 2 * cg + (tp >> 1)) + 8 * (tp & 1);
  float sq = 0.f;
  SC_STAMP_DECL
  __syncthreads();
  SC_STAMP_START
  for (int i = 0; i <= n; ++i) {
    if (i >= 1) {
      // dE[v][d] = sum_b G[v][b] pred[b][d] of slab i - 1
      const unsigned char *const gt_prev = gt_buf + ((i - 1) & 1) * 3 * GT2_IMG;
      const int vbase = (slab0 + (i - 1) * step) * SLAB;
      const bool full = vbase + SLAB <= V;
      f32x16 acc = {0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        Tri a;
#pragma unroll
        for (int t = 0; t < 3; ++t)
          a.t[t] = lds_tr8(gt_prev + t * GT2_IMG + 1024 * s + gt_rd[0], gt_prev + t * GT2_IMG + 1024 * s + gt_rd[1]);
        acc = mfma6(a, p2[s], acc);
      }
      SC_STAMP(0)
      float *const out = p.dE + ((size_t)vbase + 4 * h) * D + dcol;
      if (RMW) {
        float old[16];
#pragma unroll
        for (int q = 0; q < 16; ++q)
          old[q] = out[(long)min((q & 3) + 8 * (q >> 2), V - 1 - vbase - 4 * h) * D];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += old[q];
      }
      if (full && !RMW) {            // through a wave-private scratch: four 16-byte stores per lane
        float *const sc = t_scratch + w * (32 * T_PITCH);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          sq = fmaf(acc[q], acc[q], sq);
          sc[((q & 3) + 8 * (q >> 2) + 4 * h) * T_PITCH + r] = acc[q];
        }
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const int idx = i4 * 64 + lane, row = idx >> 3, c4 = idx & 7;
          const f32x4 t = *reinterpret_cast<const f32x4 *>(sc + row * T_PITCH + 4 * c4);
          *reinterpret_cast<f32x4 *>(p.dE + ((size_t)vbase + row) * D + 32 * w + 4 * c4) = t;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2);
          if (vbase + row + 4 * h < V) {
            sq = fmaf(acc[q], acc[q], sq);
            out[(size_t)row * D] = acc[q];
          }
        }
      }
      SC_STAMP(1)
    }
    __syncthreads();
    SC_STAMP(3)
  }
  SC_STAMP_DUMP(1)
  if (p.sq_partial) {
    sq = wave_sum(sq);
    if (lane == 0) p.sq_partial[(size_t)blockIdx.x * 4 + w] = sq;
  }
}

}  // namespace x3


// which form a catalog of V rows is scored with: the split-bf16 kernels from g_split_min_rows rows on (default 1:
// always -- 0.2722 against 0.2740 ms per step at 3,709 rows, and the native pair falls behind from there; 0 = never).
// MTAM_SCORE32_SPLIT_MIN_ROWS sets it at load, mtam_score32_set_split_min_rows() at run time (the tests run both forms).
long g_split_min_rows = -1;
bool use_split(int V) {
  if (g_split_min_rows < 0) {
    const char *e = getenv("MTAM_SCORE32_SPLIT_MIN_ROWS");
    g_split_min_rows = e ? atol(e) : 1L;
  }
  return g_split_min_rows > 0 && V >= g_split_min_rows;
}
int slabs_of(int V) { return (V + SLAB - 1) / SLAB; }
// Workgroups per batch tile: each owns a contiguous range of slabs and flushes its [128, 128] share of d_pred
// ONCE, by atomics (64 KB each at the ~1.3 TB/s float-atomic rate: 2,048 workgroups = 100 us, 512 = 25 us).
// The backward kernels run one workgroup per CU (registers, or 138 KB of LDS), so 512 ranges are two rounds over
// the 256 CUs.  MTAM_SCORE32_MAX_WGS overrides (read once).
int max_wgs() {
  static const int v = [] {
    const char *e = getenv("MTAM_SCORE32_MAX_WGS");
    const int n = e ? atoi(e) : 512;
    return n > 0 ? n : 512;
  }();
  return v;
}
int chunks_of(int V) { return max(1, min(slabs_of(V), max_wgs())); }
// the forward (lse) pass has nothing to flush: up to 2,048 ranges, so that three workgroups per CU keep loads in flight
int lse_chunks_of(int V) { return max(1, min(slabs_of(V), 2048)); }
// the split kernels walk the slabs cyclically (x3::slab_walk; slabs_per_wg = 0 says so) unless MTAM_SCORE32_CYCLIC=0
bool cyclic(int V) {
  static const bool on = [] {
    const char *e = getenv("MTAM_SCORE32_CYCLIC");
    return !(e && e[0] == '0');
  }();
  return on && use_split(V);
}
int lse_slabs_per_wg_of(int V) { return cyclic(V) ? 0 : (slabs_of(V) + lse_chunks_of(V) - 1) / lse_chunks_of(V); }
int lse_grid_of(int V) {
  return cyclic(V) ? lse_chunks_of(V) : (slabs_of(V) + lse_slabs_per_wg_of(V) - 1) / lse_slabs_per_wg_of(V);
}
int slabs_per_wg_of(int V) { return cyclic(V) ? 0 : (slabs_of(V) + chunks_of(V) - 1) / chunks_of(V); }
int grid_of(int V) { return cyclic(V) ? chunks_of(V) : (slabs_of(V) + slabs_per_wg_of(V) - 1) / slabs_per_wg_of(V); }

}  // namespace

extern "C" void mtam_score32_set_split_min_rows(long min_rows) { g_split_min_rows = min_rows < 0 ? 0 : min_rows; }

extern "C" int mtam_score32_partials(int B, int V) { return B * lse_grid_of(V) * 2; }
extern "C" int mtam_score32_sq_partials(int V) { return grid_of(V) * 4; }

extern "C" int mtam_score32_lse(const float *E, const float *pred, const int32_t *target, int B, int V,
                                float *partial, int n_partial, float *lse, float *ce, void *stream) {
  return mtam_score32_lse_range(E, pred, target, B, V, -1, partial, n_partial, lse, ce, stream);
}

extern "C" int mtam_score32_lse_range(const float *E, const float *pred, const int32_t *target, int B, int V, int row0,
                                      float *partial, int n_partial, float *lse, float *ce, void *stream) {
  MTAM_CHECK_ARG(E && pred && target && partial && lse && ce, "score32_lse: null argument");
  MTAM_CHECK_ARG(B > 0 && V > 0 && n_partial >= mtam_score32_partials(B, V),
                 "score32_lse: partial buffer holds %d floats, this form of the pass writes %d (sized before "
                 "mtam_score32_set_split_min_rows changed the form?)", n_partial, mtam_score32_partials(B, V));
  MTAM_CHECK_ARG(B > 0 && V > 0 && V < 0x7fffff00 && (B + BT - 1) / BT <= 65535, "score32_lse: bad shape B=%d V=%d", B, V);
  MTAM_CHECK_ARG(mtam_aligned16(E) && mtam_aligned16(pred), "score32_lse: operands must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = lse_grid_of(V);
  if (use_split(V))
    hipLaunchKernelGGL(x3::lse_kernel, dim3(grid, (B + BT - 1) / BT), dim3(256), 0, s, E, pred, V, B,
                       lse_slabs_per_wg_of(V), partial);
  else
    hipLaunchKernelGGL(score32_lse_kernel, dim3(grid, (B + BT - 1) / BT), dim3(256), 0, s, E, pred, V, B,
                       lse_slabs_per_wg_of(V), partial);
  hipLaunchKernelGGL(score32_finish_kernel, dim3(B), dim3(256), 0, s, E, pred, target, V, grid, partial, lse, ce, row0);
  MTAM_CHECK_LAUNCH("score32_lse");
  return MTAM_OK;
}

extern "C" int mtam_score32_bwd(const float *E, const float *pred, const float *lse, const int32_t *target, int B,
                                int V, float scale, float *d_pred, float *dE, float *sq_partial, int n_sq_partial,
                                void *stream) {
  return mtam_score32_bwd_range(E, pred, lse, target, B, V, -1, scale, d_pred, dE, sq_partial, n_sq_partial, stream);
}

extern "C" int mtam_score32_bwd_range(const float *E, const float *pred, const float *lse, const int32_t *target, int B,
                                      int V, int row0, float scale, float *d_pred, float *dE, float *sq_partial,
                                      int n_sq_partial, void *stream) {
  MTAM_CHECK_ARG(E && pred && lse && target && d_pred && dE, "score32_bwd: null argument");
  MTAM_CHECK_ARG(!sq_partial || (V > 0 && n_sq_partial == mtam_score32_sq_partials(V)),
                 "score32_bwd: sq_partial holds %d floats, this form of the pass writes %d (sized before "
                 "mtam_score32_set_split_min_rows changed the form?)", n_sq_partial, V > 0 ? mtam_score32_sq_partials(V) : 0);
  MTAM_CHECK_ARG(B > 0 && V > 0 && V < 0x7fffff00 && scale > 0.f, "score32_bwd: bad shape B=%d V=%d", B, V);
  MTAM_CHECK_ARG(mtam_aligned16(E) && mtam_aligned16(pred), "score32_bwd: operands must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool split = use_split(V);
  if (split) {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(x3::bwd_tr3_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, x3::TR3_LDS);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(x3::bwd_tr3_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, x3::TR3_LDS);
      MTAM_CHECK_ARG(e == hipSuccess, "score32_bwd: cannot reserve %d bytes of LDS: %s", x3::TR3_LDS,
                     hipGetErrorString(e));
      attr_set = true;
    }
  }
  // one launch per 128-row batch tile; a later tile adds onto the dE rows the earlier one stored
  const int ntile = (B + BT - 1) / BT;
  for (int tile = 0; tile < ntile; ++tile) {
    const long b0 = (long)tile * BT;
    BwdArgs a{E, pred + b0 * D, lse + b0, target + b0, V, (int)min((long)BT, B - b0), slabs_per_wg_of(V), scale,
              d_pred + b0 * D, dE, tile == ntile - 1 ? sq_partial : nullptr, row0};
    if (split && tile == 0)
      hipLaunchKernelGGL(x3::bwd_tr3_kernel<false>, dim3(grid_of(V)), dim3(768), x3::TR3_LDS, st, a);
    else if (split)
      hipLaunchKernelGGL(x3::bwd_tr3_kernel<true>, dim3(grid_of(V)), dim3(768), x3::TR3_LDS, st, a);
    else if (tile == 0)
      hipLaunchKernelGGL(score32_bwd_kernel<false>, dim3(grid_of(V)), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(score32_bwd_kernel<true>, dim3(grid_of(V)), dim3(256), 0, st, a);
  }
  MTAM_CHECK_LAUNCH("score32_bwd");
  return MTAM_OK;
}
