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

// tools/score_small_lab.hip (-DMTAM_SMALL_STAMPS): s_memrealtime (100 MHz) at the phase boundaries of the one-launch
// small-catalog kernel, first wave of each role of the middle workgroup
#ifdef MTAM_SMALL_STAMPS
__device__ unsigned long long g_small_stamps[3][16];
#define SM_STAMP(i)                                                                      \
  if (lane == 0 && w == 0 && c == G / 2) g_small_stamps[role][i] = __builtin_amdgcn_s_memrealtime();
#else
#define SM_STAMP(i)
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


// ---- small catalogs: the whole of training's scoring -- log-sum-exp, loss, G, d_pred, dE -- in ONE launch.
// At 3,709 rows the three launches above (lse 6.3 us, finish 4.7, backward 15.8) are chains of short phases on 116 of
// the 256 CUs: two launch gaps, the slab staged, split and multiplied twice, and 116 x 64 KB of float atomics for d_pred.
// When every slab can have its own RESIDENT workgroup (slabs <= CUs, one batch tile) the scores stay in the S waves'
// accumulators while the workgroups exchange what the softmax needs INSIDE the launch:
//     S waves   scores of the slab -> (max, sum-exp) of the slab per sample, published -> every workgroup folds the
//               G x 128 pairs itself (lse; the lane holding the target's score writes the cross entropy) -> G -> G^T image
//     D waves   the slab's share of d_pred, published to a [G][128][128] scratch (stores, no atomics) -> every workgroup
//               sums its 1/G of d_pred over the G shares in a fixed order (deterministic, unlike the atomics)
//     T waves   dE rows of the slab and their squares, as in bwd_tr3_kernel
// "Published": write-through (sc1) stores, read by sc1 loads (16-byte ones, by inline asm: the atomic builtins stop at
// 8 bytes).  There is no barrier object: the DATA is its own flag.  Both exchange buffers exist twice; a launch works in
// the copy of its epoch's parity and refills the OTHER copy with a sentinel (all-ones words, which no sum-exp and no
// finite product is), so a reader polls the very words it needs until none of them is the sentinel: every dependent
// step is one trip to memory (~1.4 us here), not store -> flag -> poll -> load.  Two earlier forms, measured with
// tools/score_small_lab.hip at 116 workgroups: arrival counter + generation word, 45 ns per workgroup on the
// same-address atomics, 6.1 + 4.3 us in barriers of a 26 us launch; one flag per workgroup polled by four waves,
// 3.0 + 2.8 us in barriers, 23.7 us.  A reader gives up after ~1 s of polling (it cannot hang the device); the launch
// then writes NaN.
constexpr int SMALL_MS_PARTS = 12;                      // 768 threads = 12 x 64 sample pairs
constexpr int SMALL_D_SCRATCH = 4 * 32 * T_PITCH * 4;   // the D waves' way out (as the T waves' t_scratch)
constexpr int SMALL_LDS = 3 * E2_IMG + 3 * GT2_IMG + OUT_SCRATCH + SMALL_D_SCRATCH + SMALL_MS_PARTS * BT * 8;
constexpr long SMALL_SPIN_LIMIT = 1L << 19;
constexpr int SMALL_MAX_G = 12 * 14;                    // one batch of 14 loads per thread folds the pairs: 5,376 rows
constexpr int SMALL_HEAD_WORDS = 64;                    // [0] epoch (launches so far on this buffer)
constexpr unsigned int SMALL_SENTINEL = 0xffffffffu;

struct SmallArgs {
  const float *E, *P;
  const int32_t *target;
  int V, Bt;
  float scale;
  unsigned int *head;          // SMALL_HEAD_WORDS words
  float *ms;                   // [2][G][128] (max, sum-exp) pairs
  float *dp_part;              // [2][G][128][128] shares of d_pred
  float *lse, *ce, *d_pred, *dE, *sq_partial;
  int red_U, red_groups;       // the d_pred reduction's split: pieces per workgroup, groups of threads per piece
  unsigned int red_inv_U;      // 2^20 / red_U + 1
};

// (a VMEM store of more than 8 bytes must not be followed at once by a VALU write of its data registers: the
// assembler's hazard pass does not see inside inline asm, hence the s_nop)
__device__ __forceinline__ void store16_sc1(float *p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// N 16-byte sc1 loads in flight, then the wait (the compiler does not count loads issued by inline asm)
__device__ __forceinline__ void load16x8_sc1(f32x4 (&v)[8], const float *const (&src)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %9, off sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc1\n\t"
      "global_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\t"
      "global_load_dwordx4 %7, %15, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]), "v"(src[7])
      : "memory");
}
// (base in SGPRs + a 32-bit byte offset per load: 14 + 14 + 1 operands)
__device__ __forceinline__ void load16x14_sc1(f32x4 (&v)[14], const float *base, const unsigned int (&off)[14]) {
  asm volatile(
      "global_load_dwordx4 %0, %14, %28 sc1\n\t"
      "global_load_dwordx4 %1, %15, %28 sc1\n\t"
      "global_load_dwordx4 %2, %16, %28 sc1\n\t"
      "global_load_dwordx4 %3, %17, %28 sc1\n\t"
      "global_load_dwordx4 %4, %18, %28 sc1\n\t"
      "global_load_dwordx4 %5, %19, %28 sc1\n\t"
      "global_load_dwordx4 %6, %20, %28 sc1\n\t"
      "global_load_dwordx4 %7, %21, %28 sc1\n\t"
      "global_load_dwordx4 %8, %22, %28 sc1\n\t"
      "global_load_dwordx4 %9, %23, %28 sc1\n\t"
      "global_load_dwordx4 %10, %24, %28 sc1\n\t"
      "global_load_dwordx4 %11, %25, %28 sc1\n\t"
      "global_load_dwordx4 %12, %26, %28 sc1\n\t"
      "global_load_dwordx4 %13, %27, %28 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
        "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13])
      : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]),
        "v"(off[8]), "v"(off[9]), "v"(off[10]), "v"(off[11]), "v"(off[12]), "v"(off[13]), "s"(base)
      : "memory");
}
__device__ __forceinline__ bool is_sentinel(float x) { return __float_as_uint(x) == SMALL_SENTINEL; }

// Every thread: fold the G (max, sum-exp) pairs of every sample -- 12 threads per PAIR of samples, each polling its own
// <= 14 pieces of 16 bytes until all of them have been published -- into LDS.
__device__ __forceinline__ void small_fold_pairs(const float *ms, int G, int tid, float2 *ms_lds) {
  const int bp = tid & 63, part = tid >> 6;
  f32x4 v[14];
  unsigned int off[14];       // bytes
#pragma unroll
  for (int j = 0; j < 14; ++j) off[j] = ((unsigned int)min(part + SMALL_MS_PARTS * j, G - 1) * BT + 2 * bp) * 8u;
  bool good = true;
  for (long spins = 0;; ++spins) {
    load16x14_sc1(v, ms, off);
    bool miss = false;
#pragma unroll
    for (int j = 0; j < 14; ++j) miss = miss || is_sentinel(v[j][1]) || is_sentinel(v[j][3]);
    if (__builtin_amdgcn_ballot_w64(miss) == 0) break;
    if (spins > SMALL_SPIN_LIMIT) { good = false; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  float M0 = -INFINITY, S0 = 0.f, M1 = -INFINITY, S1 = 0.f;
#pragma unroll
  for (int j = 0; j < 14; ++j) {
    if (part + SMALL_MS_PARTS * j < G) {
      float Mn = fmaxf(M0, v[j][0]);
      S0 = S0 * fast_exp2((M0 - Mn) * L2E) + v[j][1] * fast_exp2((v[j][0] - Mn) * L2E);
      M0 = Mn;
      Mn = fmaxf(M1, v[j][2]);
      S1 = S1 * fast_exp2((M1 - Mn) * L2E) + v[j][3] * fast_exp2((v[j][2] - Mn) * L2E);
      M1 = Mn;
    }
  }
  if (!good) S0 = S1 = __builtin_nanf("");
  *reinterpret_cast<f32x4 *>(&ms_lds[part * BT + 2 * bp]) = f32x4{M0, S0, M1, S1};
}

// Every thread: this workgroup's 1/G of d_pred, summed over the G shares in a fixed order; a thread polls its own
// pieces of the shares until all of them have been published.
__device__ __forceinline__ void small_reduce_d_pred(const SmallArgs &p, const float *dp_part, int G, int c, int tid,
                                                    f32x4 *red) {
  // (U, groups and 2^20 / U + 1 come from the host: a division by a run-time value is ~25 VALU instructions, and this
  // is the launch's last dependent phase)
  const int units = p.Bt * (D / 4);                 // 16-byte pieces of d_pred
  const int U = p.red_U;                            // ceil(units / G) <= 512 (G >= 8)
  const int groups = p.red_groups;                  // min(768 / U, G)
  const int grp = (int)(((unsigned)tid * p.red_inv_U) >> 20), ul = tid - grp * U;      // exact: tid U < 2^20
  const int unit = c * U + ul;
  if (grp < groups) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    const int uc = min(unit, units - 1);
    for (int g0 = grp; g0 < G; g0 += groups * 8) {
      f32x4 v[8];
      const float *src[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) src[j] = dp_part + (size_t)min(g0 + groups * j, G - 1) * BT * D + 4 * (size_t)uc;
      bool good = true;
      for (long spins = 0;; ++spins) {
        load16x8_sc1(v, src);
        bool miss = false;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          miss = miss || is_sentinel(v[j][0]) || is_sentinel(v[j][1]) || is_sentinel(v[j][2]) || is_sentinel(v[j][3]);
        if (!miss) break;
        if (spins > SMALL_SPIN_LIMIT) { good = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (g0 + groups * j < G) a += v[j];
      if (!good) a[0] = __builtin_nanf("");
    }
    red[grp * U + ul] = a;
  }
  __syncthreads();
  if (grp == 0 && unit < units) {
    f32x4 t = red[ul];
    for (int k = 1; k < groups; ++k) t += red[k * U + ul];
    f32x4 *const dst = reinterpret_cast<f32x4 *>(p.d_pred) + unit;
    *dst = *dst + t;
  }
}

// The three roles run the same sequence of workgroup barriers: [slab staged] [scored] fold [pairs in LDS] [G^T image]
// products [products done] reduce (one barrier inside).
__global__ __launch_bounds__(768) void train_small_kernel(SmallArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *const e_img = lds;                                   // [3][E2_IMG]
  unsigned char *const gt_img = lds + 3 * E2_IMG;                     // [3][GT2_IMG]
  float *const t_scratch = reinterpret_cast<float *>(gt_img + 3 * GT2_IMG);
  float *const d_scratch = t_scratch + OUT_SCRATCH / 4;
  float2 *const ms_lds = reinterpret_cast<float2 *>(d_scratch + SMALL_D_SCRATCH / 4);      // [12][128] (max, sum-exp)
  f32x4 *const red = reinterpret_cast<f32x4 *>(e_img);                // the reduction: [groups][U], <= 12 KB of 24
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int role = wave >> 2, w = wave & 3;
  const int V = p.V, G = gridDim.x, c = blockIdx.x;
  const int vbase = c * SLAB;
  const int cg = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;      // transposed-read lane roles
  const int dcol = 32 * w + r;
  // this launch's copies of the exchange buffers, and the copies to refill with the sentinel for the next one
  const unsigned int epoch = p.head[0];
  const size_t ms_copy = (size_t)G * BT * 2, dp_copy = (size_t)G * BT * D;
  float *const ms = p.ms + (epoch & 1) * ms_copy, *const ms_next = p.ms + ((epoch & 1) ^ 1) * ms_copy;
  float *const dp_part = p.dp_part + (epoch & 1) * dp_copy, *const dp_next = p.dp_part + ((epoch & 1) ^ 1) * dp_copy;
  const f32x4 sentinel4 = {__uint_as_float(SMALL_SENTINEL), __uint_as_float(SMALL_SENTINEL),
                           __uint_as_float(SMALL_SENTINEL), __uint_as_float(SMALL_SENTINEL)};
  SM_STAMP(0)

  if (role == 0) {
    // ------------------------------------------------------------------ S waves (batch rows 32 w ..)
    const int bcol = 32 * w + r;
    const bool valid_b = bcol < p.Bt;
    const int brow = min(bcol, p.Bt - 1);
    Stage st;
    stage_load(st, p.E, vbase, V, tid);
    Tri p1[8];
    load_pred_rows(p1, p.P + (size_t)brow * D, h);
    const int t_rel = local_target(p.target[brow], -1, V) - vbase - 4 * h;
    stage_store_split_tr(st, e_img, tid);
    __syncthreads();                               // [slab staged]
    SM_STAMP(1)
    f32x16 acc = {0.f};
    const int swz_r = ((r & 3) << 2) | ((r >> 2) & 3);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      Tri a;
      const int o = 256 * r + 16 * ((2 * s + h) ^ swz_r);
#pragma unroll
      for (int t = 0; t < 3; ++t) a.t[t] = *reinterpret_cast<const bf16x8 *>(e_img + t * E2_IMG + o);
      acc = mfma6(a, p1[s], acc);
    }
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool in = vbase + 4 * h + (q & 3) + 8 * (q >> 2) < V;
      m = fmaxf(m, in ? acc[q] : -INFINITY);
    }
    const float nref = -((m == -INFINITY) ? 0.f : m) * L2E;
    float ssum = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool in = vbase + 4 * h + (q & 3) + 8 * (q >> 2) < V;
      ssum += in ? fast_exp2(fmaf(acc[q], L2E, nref)) : 0.f;
    }
    const float m2 = __shfl_xor(m, 32, 64), s2 = __shfl_xor(ssum, 32, 64);
    const float mm = fmaxf(m, m2), rf = (mm == -INFINITY) ? 0.f : mm;
    const float ss = ssum * fast_exp2((m - rf) * L2E) + s2 * fast_exp2((m2 - rf) * L2E);
    if (h == 0) {                                  // one 8-byte write-through store: the pair appears whole
      const unsigned long long pair = (unsigned long long)__float_as_uint(mm) |
                                      ((unsigned long long)__float_as_uint(ss) << 32);
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(ms) + (size_t)c * BT + bcol, pair, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    SM_STAMP(2)
    __syncthreads();                               // [scored] (nobody polls before this workgroup has published)
    small_fold_pairs(ms, G, tid, ms_lds);
    __syncthreads();                               // [pairs in LDS]
    SM_STAMP(4)
    float M = -INFINITY, S = 0.f;
#pragma unroll
    for (int k = 0; k < SMALL_MS_PARTS; ++k) {
      const float2 q = ms_lds[k * BT + bcol];
      const float Mn = fmaxf(M, q.x);
      if (Mn != -INFINITY) S = S * fast_exp2((M - Mn) * L2E) + q.y * fast_exp2((q.x - Mn) * L2E);
      M = Mn;
    }
    const float lse = M + logf(S);
    if (c == 0 && h == 0 && valid_b) p.lse[bcol] = lse;
    const float c_b = valid_b ? fmaf(-lse, L2E, log2f(p.scale)) : -INFINITY;
    float g[16];
    float tl = 0.f;
    bool has = false;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = (q & 3) + 8 * (q >> 2);
      g[q] = fast_exp2(fmaf(acc[q], L2E, c_b));
      if (row == t_rel) {
        tl = acc[q];
        has = true;
        g[q] -= valid_b ? p.scale : 0.f;
      }
      g[q] = (vbase + row + 4 * h < V) ? g[q] : 0.f;
    }
    if (has && valid_b) p.ce[bcol] = lse - tl;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      bf16x4 gq[3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __bf16 a, b, cc;
        split3(g[4 * q4 + k], a, b, cc);
        gq[0][k] = a; gq[1][k] = b; gq[2][k] = cc;
      }
      const int o = gt_off(bcol, q4) + 8 * h;
#pragma unroll
      for (int t = 0; t < 3; ++t) *reinterpret_cast<bf16x4 *>(gt_img + t * GT2_IMG + o) = gq[t];
    }
    __syncthreads();                               // [G^T image]
    SM_STAMP(5)
    // refill the other copy of this workgroup's pairs for the next launch (plain stores: a launch boundary lies between)
    if (tid < BT / 2) *reinterpret_cast<f32x4 *>(ms_next + ((size_t)c * BT + 2 * tid) * 2) = sentinel4;
    __syncthreads();                               // [products done]
    small_reduce_d_pred(p, dp_part, G, c, tid, red);
    SM_STAMP(8)
    // (every workgroup has read the epoch before it published its pairs, and this one has seen all of them)
    if (c == 0 && tid == 0) p.head[0] = epoch + 1;
    return;
  }

  if (role == 1) {
    // ------------------------------------------------------------------ D waves: the slab's share of d_pred
    int et_off[2], ga_off[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) et_off[jj] = e_off(8 * h + 4 * jj + tq, 4 * w + 2 * cg + (tp >> 1)) + 8 * (tp & 1);
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) ga_off[sp] = gt_off(r, 2 * sp + h);
    __syncthreads();                               // [slab staged]
    __syncthreads();                               // [scored]
    small_fold_pairs(ms, G, tid, ms_lds);
    __syncthreads();                               // [pairs in LDS]
    __syncthreads();                               // [G^T image]
    SM_STAMP(5)
    f32x16 dp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dp[i] = f32x16{0.f};
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      Tri bf;
#pragma unroll
      for (int t = 0; t < 3; ++t)
        bf.t[t] = lds_tr8(e_img + t * E2_IMG + 4096 * sp + et_off[0], e_img + t * E2_IMG + 4096 * sp + et_off[1]);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        Tri a;
#pragma unroll
        for (int t = 0; t < 3; ++t)
          a.t[t] = *reinterpret_cast<const bf16x8 *>(gt_img + t * GT2_IMG + 2048 * mb + ga_off[sp]);
        dp[mb] = mfma6(a, bf, dp[mb]);
      }
    }
    SM_STAMP(9)
    // out through a wave-private scratch: 32 rows x 32 columns become four 16-byte write-through stores per lane
    float *const sc = d_scratch + w * (32 * T_PITCH);
    float *const out = dp_part + (size_t)c * BT * D + 32 * w;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
      for (int q = 0; q < 16; ++q) sc[acc_row(q, h) * T_PITCH + r] = dp[mb][q];
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int idx = i4 * 64 + lane, row = idx >> 3, c4 = idx & 7;
        store16_sc1(out + (size_t)(32 * mb + row) * D + 4 * c4, *reinterpret_cast<const f32x4 *>(sc + row * T_PITCH + 4 * c4));
      }
    }
    SM_STAMP(6)
    __syncthreads();                               // [products done]
    small_reduce_d_pred(p, dp_part, G, c, tid, red);
    SM_STAMP(8)
    // off the chain: refill the other copy of this workgroup's share for the next launch (plain stores: a launch
    // boundary lies between)
    {
      f32x4 *const dst = reinterpret_cast<f32x4 *>(dp_next + (size_t)c * BT * D);
#pragma unroll
      for (int i = 0; i < BT * D / 4 / 256; ++i) dst[i * 256 + (tid - 256)] = sentinel4;
    }
    return;
  }

  // -------------------------------------------------------------------- T waves: dE (columns d = 32 w ..)
  Tri p2[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = p.P[(size_t)min(16 * s + 8 * h + j, p.Bt - 1) * D + dcol];
    p2[s] = split8(x);
  }
  int gt_rd[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) gt_rd[jj] = gt_off(8 * h + 4 * jj + tq, 2 * cg + (tp >> 1)) + 8 * (tp & 1);
  __syncthreads();                                 // [slab staged]
  __syncthreads();                                 // [scored]
  small_fold_pairs(ms, G, tid, ms_lds);
  __syncthreads();                                 // [pairs in LDS]
  __syncthreads();                                 // [G^T image]
  SM_STAMP(5)
  f32x16 de = {0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    Tri a;
#pragma unroll
    for (int t = 0; t < 3; ++t)
      a.t[t] = lds_tr8(gt_img + t * GT2_IMG + 1024 * s + gt_rd[0], gt_img + t * GT2_IMG + 1024 * s + gt_rd[1]);
    de = mfma6(a, p2[s], de);
  }
  float sq = 0.f;
  if (vbase + SLAB <= V) {           // through a wave-private scratch: four 16-byte stores per lane
    float *const sc = t_scratch + w * (32 * T_PITCH);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      sq = fmaf(de[q], de[q], sq);
      sc[((q & 3) + 8 * (q >> 2) + 4 * h) * T_PITCH + r] = de[q];
    }
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int idx = i4 * 64 + lane, row = idx >> 3, c4 = idx & 7;
      const f32x4 t = *reinterpret_cast<const f32x4 *>(sc + row * T_PITCH + 4 * c4);
      *reinterpret_cast<f32x4 *>(p.dE + ((size_t)vbase + row) * D + 32 * w + 4 * c4) = t;
    }
  } else {
    float *const out = p.dE + ((size_t)vbase + 4 * h) * D + dcol;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = (q & 3) + 8 * (q >> 2);
      if (vbase + row + 4 * h < V) {
        sq = fmaf(de[q], de[q], sq);
        out[(size_t)row * D] = de[q];
      }
    }
  }
  if (p.sq_partial) {
    sq = wave_sum(sq);
    if (lane == 0) p.sq_partial[(size_t)c * 4 + w] = sq;
  }
  SM_STAMP(6)
  __syncthreads();                                 // [products done]
  small_reduce_d_pred(p, dp_part, G, c, tid, red);
  SM_STAMP(8)
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

// ---- training's scoring as ONE call: per-sample loss terms and both scoring gradients.
namespace {
int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return v;
  }();
  return n;
}
// the one-launch form (x3::train_small_kernel): one batch tile, every slab a resident workgroup of its own
// (MTAM_SCORE32_FUSED=0 turns it off; processes SHARING a GPU must turn it off -- their grids would wait on each
// other's CUs)
int g_fused = -1;               // -1: MTAM_SCORE32_FUSED (read once); mtam_score32_set_fused() overrides
bool small_form(int B, int V) {
  if (g_fused < 0) {
    const char *e = getenv("MTAM_SCORE32_FUSED");
    g_fused = (e && e[0] == '0') ? 0 : 1;
  }
  const bool on = g_fused != 0;
  const int G = slabs_of(V);
  return on && use_split(V) && B <= BT && G >= 8 && G <= min(cu_count() / 8 * 7, x3::SMALL_MAX_G);      // (CUs to spare)
}
}  // namespace

extern "C" void mtam_score32_set_fused(int on) { g_fused = on ? 1 : 0; }

extern "C" int mtam_score32_train_is_fused(int B, int V) { return (B > 0 && V > 0 && small_form(B, V)) ? 1 : 0; }

extern "C" long mtam_score32_train_work_floats(int B, int V) {
  if (B <= 0 || V <= 0) return 0;
  if (small_form(B, V)) return x3::SMALL_HEAD_WORDS + 2 * ((long)slabs_of(V) * BT * 2 + (long)slabs_of(V) * BT * D);
  return mtam_score32_partials(B, V);
}

extern "C" int mtam_score32_train_work_init(float *work, long n_work, int B, int V, void *stream) {
  MTAM_CHECK_ARG(work && B > 0 && V > 0 && n_work >= mtam_score32_train_work_floats(B, V),
                 "score32_train_work_init: work holds %ld floats, B=%d V=%d needs %ld", n_work, B, V,
                 mtam_score32_train_work_floats(B, V));
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(work, 0, (size_t)n_work * 4, s);
  if (e == hipSuccess && small_form(B, V))       // the exchange buffers start as "nothing published"
    e = hipMemsetAsync(work + x3::SMALL_HEAD_WORDS, 0xff, (size_t)(n_work - x3::SMALL_HEAD_WORDS) * 4, s);
  MTAM_CHECK_ARG(e == hipSuccess, "score32_train_work_init: %s", hipGetErrorString(e));
  return MTAM_OK;
}

extern "C" int mtam_score32_train(const float *E, const float *pred, const int32_t *target, int B, int V, float scale,
                                  float *work, long n_work, float *lse, float *ce, float *d_pred, float *dE,
                                  float *sq_partial, int n_sq_partial, void *stream) {
  MTAM_CHECK_ARG(E && pred && target && work && lse && ce && d_pred && dE, "score32_train: null argument");
  MTAM_CHECK_ARG(B > 0 && V > 0 && V < 0x7fffff00 && scale > 0.f, "score32_train: bad shape B=%d V=%d", B, V);
  MTAM_CHECK_ARG(n_work >= mtam_score32_train_work_floats(B, V),
                 "score32_train: work holds %ld floats, this form needs %ld", n_work,
                 mtam_score32_train_work_floats(B, V));
  if (!small_form(B, V)) {
    const int rc = mtam_score32_lse(E, pred, target, B, V, work, (int)min(n_work, (long)0x7fffffff), lse, ce, stream);
    if (rc != MTAM_OK) return rc;
    return mtam_score32_bwd(E, pred, lse, target, B, V, scale, d_pred, dE, sq_partial, n_sq_partial, stream);
  }
  MTAM_CHECK_ARG(!sq_partial || n_sq_partial == mtam_score32_sq_partials(V),
                 "score32_train: sq_partial holds %d floats, this form of the pass writes %d", n_sq_partial,
                 mtam_score32_sq_partials(V));
  MTAM_CHECK_ARG(mtam_aligned16(E) && mtam_aligned16(pred) && mtam_aligned16(work) && mtam_aligned16(d_pred) &&
                 mtam_aligned16(dE), "score32_train: operands must be 16-byte aligned");
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(x3::train_small_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, x3::SMALL_LDS);
    MTAM_CHECK_ARG(e == hipSuccess, "score32_train: cannot reserve %d bytes of LDS: %s", x3::SMALL_LDS,
                   hipGetErrorString(e));
    attr_set = true;
  }
  const int G = slabs_of(V);
  x3::SmallArgs a{E, pred, target, V, B, scale,
                  reinterpret_cast<unsigned int *>(work), work + x3::SMALL_HEAD_WORDS,
                  work + x3::SMALL_HEAD_WORDS + 2 * (long)G * BT * 2, lse, ce, d_pred, dE, sq_partial, 0, 0, 0u};
  a.red_U = (B * (D / 4) + G - 1) / G;
  a.red_groups = min(768 / a.red_U, G);
  a.red_inv_U = (1u << 20) / (unsigned)a.red_U + 1u;
  hipLaunchKernelGGL(x3::train_small_kernel, dim3(G), dim3(768), x3::SMALL_LDS, static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("score32_train");
  return MTAM_OK;
}
