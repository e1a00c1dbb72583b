// fp32 catalog scoring without stored logits, for catalogs small enough that the step is bound by launches
// rather than bytes (ml-1m: 3,709 rows): the training loss and both scoring gradients of
// base_model.output (Model/base_model.py:300-328, 290-297) in TWO launches instead of four
// (logits GEMM, softmax-CE, dE GEMM, d_pred GEMM) and without the [B, V] round trips between them.
//
//   score32_lse   scores of a 32-row catalog slab on v_mfma_f32_32x32x2_f32, per-row running (max, sum-exp)
//   score32_bwd   recomputes the slab's scores, forms G = (softmax - onehot) * scale in registers, and
//                 produces dE (stored, with its squared norm) and the workgroup's share of d_pred (atomics)
//
// Same structure as csrc/score16.hip with fp32 operands: lane l (r = l & 31, h = l >> 5) of the 32x32x2
// instruction supplies A[row r][k = h] and B[k = h][col r]; k-step s pairs element s (lane half 0) with
// element s + 64 (lane half 1) of a 128-long contraction, so a lane's resident operand is 64 contiguous
// floats, and a staged row keeps its two halves one float apart ([64][gap][64][gap], 130 floats): the
// 64 lanes of a fragment read then hit 64 distinct LDS banks.  fp32 products, fp32 accumulation (two
// independent accumulator chains per tile: a dependent MFMA issues every ~84 cycles, an independent one
// every 64).  Evaluation keeps the stored-logits GEMM (its k-ordered fmaf chain is the ranking contract).
#include "common.h"
#include <stdlib.h>

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

__global__ __launch_bounds__(256) void score32_finish_kernel(const float *__restrict__ E, const float *__restrict__ P,
                                                             const int32_t *__restrict__ target, int V, int chunks,
                                                             const float *__restrict__ partial,
                                                             float *__restrict__ lse, float *__restrict__ ce) {
  __shared__ float red[4], red2[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *pp = partial + (size_t)b * chunks * 2;
  // all loads first: the target row and this row's partials
  const long t = min(max(target[b], 0), V - 1);
  float dot = (tid < D) ? P[(size_t)b * D + tid] * E[t * D + tid] : 0.f;
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
    ce[b] = l - ((red2[0] + red2[1]) + (red2[2] + red2[3]));
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
  int lab;      // developer switches (MTAM_SCORE32_LAB, tools/score32_time.py): parts of x3::bwd_pc_kernel cut out to
                // time the rest -- 1: no dE stores, 2: no G[v][b] element writes, 4: no E^T gathers for d_pred,
                // 8: no exp.  0 in every product run.
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
  const int t_b = valid_b ? min(max(p.target[brow], 0), V - 1) : -1;
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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int E_PITCH = 272;             // bytes per staged row image: 128 bf16 + 16
constexpr int E_IMG = SLAB * E_PITCH;    // one of the three images of a 32-row slab
constexpr int G_PITCH = 272;             // G[v][b]: 128 b x 2 B + 16
constexpr int G_IMG = SLAB * G_PITCH;
constexpr int GT_PITCH = 80;             // G^T[b][v]: 32 v x 2 B + 16
constexpr int GT_IMG = BT * GT_PITCH;

__device__ __forceinline__ void split3(float x, __bf16 &a, __bf16 &b, __bf16 &c) {
  a = (__bf16)x;
  const float r1 = x - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}
struct Tri {
  bf16x8 t[3];
};
__device__ __forceinline__ Tri split8(const float (&x)[8]) {
  Tri o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 a, b, c;
    split3(x[j], a, b, c);
    o.t[0][j] = a; o.t[1][j] = b; o.t[2][j] = c;
  }
  return o;
}
// the six products of weight >= 2^-16, smallest first
__device__ __forceinline__ f32x16 mfma6(const Tri &a, const Tri &b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[2], b.t[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b.t[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b.t[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[0], acc, 0, 0, 0);
  return acc;
}
// staged fp32 slab (4 x 16 B per thread) -> three bf16 images in LDS
__device__ __forceinline__ void stage_store_split(const Stage &st, unsigned char *e_img, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid, row = c >> 5, col = (c & 31) * 4;
    const float x[4] = {st.v[i].x, st.v[i].y, st.v[i].z, st.v[i].w};
    bf16x4 q[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __bf16 a, b, cc;
      split3(x[j], a, b, cc);
      q[0][j] = a; q[1][j] = b; q[2][j] = cc;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<bf16x4 *>(e_img + t * E_IMG + row * E_PITCH + col * 2) = q[t];
  }
}
// A fragment (rows v = r, k = d = 16 s + 8 h ..) of the three images
__device__ __forceinline__ Tri e_frag(const unsigned char *e_img, int s, int r, int h) {
  Tri o;
#pragma unroll
  for (int t = 0; t < 3; ++t)
    o.t[t] = *reinterpret_cast<const bf16x8 *>(e_img + t * E_IMG + r * E_PITCH + 32 * s + 16 * h);
  return o;
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
__device__ __forceinline__ f32x16 slab_scores(const unsigned char *e_img, const Tri (&p1)[8], int r, int h) {
  f32x16 acc = {0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) acc = mfma6(e_frag(e_img, s, r, h), p1[s], acc);
  return acc;
}

__global__ __launch_bounds__(256) void lse_kernel(const float *__restrict__ E, const float *__restrict__ P, int V,
                                                  int B, int slabs_per_wg, float *__restrict__ partial) {
  __shared__ __attribute__((aligned(16))) unsigned char e_img[3 * E_IMG];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int chunks = gridDim.x, c = blockIdx.x;
  const long b = (long)blockIdx.y * BT + 32 * w + r;
  const int nslab = (V + SLAB - 1) / SLAB;
  const int slab0 = min(c * slabs_per_wg, nslab), slab1 = min(nslab, slab0 + slabs_per_wg);
  Tri p1[8];
  load_pred_rows(p1, P + min(b, (long)B - 1) * D, h);
  float m = -INFINITY, ssum = 0.f;
  Stage st;
  if (slab0 < slab1) stage_load(st, E, slab0 * SLAB, V, tid);
  for (int sl = slab0; sl < slab1; ++sl) {
    stage_store_split(st, e_img, tid);
    __syncthreads();
    if (sl + 1 < slab1) stage_load(st, E, (sl + 1) * SLAB, V, tid);
    f32x16 acc = slab_scores(e_img, p1, r, h);
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

constexpr int BWD_LDS = 3 * E_IMG + 3 * G_IMG + 3 * GT_IMG;      // 82,944 B

template <bool RMW>
__global__ __launch_bounds__(256) void bwd_kernel(BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *const e_img = lds, *const g_img = lds + 3 * E_IMG, *const gt_img = g_img + 3 * G_IMG;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int V = p.V;
  const int nslab = (V + SLAB - 1) / SLAB;
  const int slab0 = min((int)blockIdx.x * p.slabs_per_wg, nslab), slab1 = min(nslab, slab0 + p.slabs_per_wg);
  const int dcol = 32 * w + r;        // this lane's output column in both backward products
  const int bcol = 32 * w + r;        // the batch row (of the tile) whose scores sit on this lane
  const bool valid_b = bcol < p.Bt;
  const int brow = min(bcol, p.Bt - 1);
  const float c_b = valid_b ? fmaf(-p.lse[brow], L2E, log2f(p.scale)) : -INFINITY;
  const int t_b = valid_b ? min(max(p.target[brow], 0), V - 1) : -1;
  Tri p1[8], p2[8];
  load_pred_rows(p1, p.P + (size_t)brow * D, h);
  // pred[k = b = 16 s + 8 h + j][n = d = dcol]: the B operand of dE (rows past the tile: any valid row, their G is 0)
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = p.P[(size_t)min(16 * s + 8 * h + j, p.Bt - 1) * D + dcol];
    p2[s] = split8(x);
  }
  f32x16 dp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dp[i] = f32x16{0.f};
  float sq = 0.f;

  unsigned char *const g_wr = g_img + (4 * h) * G_PITCH + (32 * w + r) * 2;       // G[v][b], one element
  unsigned char *const gt_wr = gt_img + (32 * w + r) * GT_PITCH + (4 * h) * 2;    // G^T[b][v], four rows of v
  const unsigned char *const g_rd = g_img + r * G_PITCH + 16 * h;
  const unsigned char *const gt_rd = gt_img + r * GT_PITCH + 16 * h;

  Stage st;
  if (slab0 < slab1) stage_load(st, p.E, slab0 * SLAB, V, tid);
  for (int sl = slab0; sl < slab1; ++sl) {
    stage_store_split(st, e_img, tid);
    __syncthreads();
    if (sl + 1 < slab1) stage_load(st, p.E, (sl + 1) * SLAB, V, tid);
    const int vbase = sl * SLAB;
    const bool full = vbase + SLAB <= V;
    const int vlim = full ? 0x7fffffff : V;

    // ---- scores of this wave's 32 batch rows; G, split, to LDS in both orientations
    {
      const f32x16 acc = slab_scores(e_img, p1, r, h);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        bf16x4 gq[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 8 * q4 + i;                      // + 4 h: catalog row of the slab
          const int v = vbase + row + 4 * h;
          float g = fast_exp2(fmaf(acc[4 * q4 + i], L2E, c_b)) - ((v == t_b) ? p.scale : 0.f);
          g = (v < vlim) ? g : 0.f;
          __bf16 a, b, c;
          split3(g, a, b, c);
          gq[0][i] = a; gq[1][i] = b; gq[2][i] = c;
          *reinterpret_cast<__bf16 *>(g_wr + row * G_PITCH) = a;
          *reinterpret_cast<__bf16 *>(g_wr + G_IMG + row * G_PITCH) = b;
          *reinterpret_cast<__bf16 *>(g_wr + 2 * G_IMG + row * G_PITCH) = c;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<bf16x4 *>(gt_wr + t * GT_IMG + (8 * q4) * 2) = gq[t];
      }
    }
    __syncthreads();

    // ---- dE[v][d] = sum_b G[v][b] pred[b][d]   (this wave: columns d = 32 w ..)
    {
      f32x16 acc = {0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        Tri a;
#pragma unroll
        for (int t = 0; t < 3; ++t) a.t[t] = *reinterpret_cast<const bf16x8 *>(g_rd + t * G_IMG + 32 * s);
        acc = mfma6(a, p2[s], acc);
      }
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
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Tri bfrag;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        uint16_t u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          u[j] = *reinterpret_cast<const uint16_t *>(e_img + t * E_IMG + (16 * s + 8 * h + j) * E_PITCH + dcol * 2);
        const u32x4 pk = {(uint32_t)u[0] | ((uint32_t)u[1] << 16), (uint32_t)u[2] | ((uint32_t)u[3] << 16),
                          (uint32_t)u[4] | ((uint32_t)u[5] << 16), (uint32_t)u[6] | ((uint32_t)u[7] << 16)};
        bfrag.t[t] = __builtin_bit_cast(bf16x8, pk);
      }
#pragma unroll
      for (int mblk = 0; mblk < 4; ++mblk) {
        Tri a;
#pragma unroll
        for (int t = 0; t < 3; ++t)
          a.t[t] = *reinterpret_cast<const bf16x8 *>(gt_rd + t * GT_IMG + (32 * mblk) * GT_PITCH + 32 * s);
        dp[mblk] = mfma6(a, bfrag, dp[mblk]);
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

// ---- the same pass with the work cut by ROLE instead of by tile: 8 waves, two per SIMD.
// bwd_kernel above keeps one wave per SIMD busy with everything in turn -- split E, scores, exp + split G, two
// products -- and its 356 registers per lane allow no second workgroup on the CU: the matrix pipe idles through
// every VALU / LDS stretch (6.4 ms at 10 M rows against 2.35 ms of MFMA issue).  Here waves 0..3 PRODUCE (split the
// next slab, score it, turn the scores into the three G images) while waves 4..7, their SIMD partners, CONSUME
// (dE and d_pred of the current slab): vector work of one role runs beside matrix work of the other.
//   phase a   P: split E(s + 1) -> e[next]                 C: dE(s)            <- g
//   phase b   P: scores(s + 1)  <- e[next]                  C: d_pred(s), block 0      <- gt[cur], e[cur]
//   phase c   P: G(s + 1) -> g, gt[next]                    C: d_pred(s), blocks 1..3  <- gt[cur], e[cur]
// E images and G^T images are double-buffered, G single (140 KB of LDS); three barriers per slab, as before.
constexpr int T_PITCH = 36;             // per consumer wave: a 32 x 32 fp32 tile on its way out, rows 36 floats apart
constexpr int PC_SCRATCH = 4 * 32 * T_PITCH * 4;
constexpr int PC_LDS = 2 * 3 * E_IMG + 3 * G_IMG + 2 * 3 * GT_IMG + PC_SCRATCH;      // 158,208 B

template <bool RMW>
__global__ __launch_bounds__(512) void bwd_pc_kernel(BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *const e_buf = lds;                                   // [2][3][E_IMG]
  unsigned char *const g_img = lds + 2 * 3 * E_IMG;                   // [3][G_IMG]
  unsigned char *const gt_buf = g_img + 3 * G_IMG;                    // [2][3][GT_IMG]
  float *const t_scratch = reinterpret_cast<float *>(gt_buf + 2 * 3 * GT_IMG);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const bool producer = wave < 4;
  const int w = wave & 3, ptid = tid & 255;
  const int V = p.V;
  const int nslab = (V + SLAB - 1) / SLAB;
  const int slab0 = min((int)blockIdx.x * p.slabs_per_wg, nslab), slab1 = min(nslab, slab0 + p.slabs_per_wg);
  const int n = slab1 - slab0;
  if (n <= 0) {                      // (a workgroup past the end of the catalog: nothing to add, nothing to sum)
    if (!producer && p.sq_partial && lane == 0) p.sq_partial[(size_t)blockIdx.x * 4 + w] = 0.f;
    return;
  }

  if (producer) {
    // ---------------------------------------------------------------- producer waves
    const int bcol = 32 * w + r;      // the batch row (of the tile) whose scores sit on this lane
    const bool valid_b = bcol < p.Bt;
    const int brow = min(bcol, p.Bt - 1);
    const float c_b = valid_b ? fmaf(-p.lse[brow], L2E, log2f(p.scale)) : -INFINITY;
    const int t_b = valid_b ? min(max(p.target[brow], 0), V - 1) : -1;
    Tri p1[8];
    load_pred_rows(p1, p.P + (size_t)brow * D, h);
    unsigned char *const g_wr = g_img + (4 * h) * G_PITCH + (32 * w + r) * 2;
    Stage st;
    f32x16 acc;
    auto make_g = [&](int sl, unsigned char *gt_img) {      // scores in acc -> G(sl), split, both orientations
      unsigned char *const gt_wr = gt_img + (32 * w + r) * GT_PITCH + (4 * h) * 2;
      const int vbase = sl * SLAB;
      const int vlim = (vbase + SLAB <= V) ? 0x7fffffff : V;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        bf16x4 gq[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 8 * q4 + i;
          const int v = vbase + row + 4 * h;
          float g = ((p.lab & 8) ? acc[4 * q4 + i] * 1e-3f : fast_exp2(fmaf(acc[4 * q4 + i], L2E, c_b))) -
                    ((v == t_b) ? p.scale : 0.f);
          g = (v < vlim) ? g : 0.f;
          __bf16 a, b, c;
          split3(g, a, b, c);
          gq[0][i] = a; gq[1][i] = b; gq[2][i] = c;
          if (!(p.lab & 2)) {
            *reinterpret_cast<__bf16 *>(g_wr + row * G_PITCH) = a;
            *reinterpret_cast<__bf16 *>(g_wr + G_IMG + row * G_PITCH) = b;
            *reinterpret_cast<__bf16 *>(g_wr + 2 * G_IMG + row * G_PITCH) = c;
          }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<bf16x4 *>(gt_wr + t * GT_IMG + (8 * q4) * 2) = gq[t];
      }
    };
    // prologue: slab 0 scored and turned into G before the first consumer phase
    stage_load(st, p.E, slab0 * SLAB, V, ptid);
    stage_store_split(st, e_buf, ptid);
    if (n > 1) stage_load(st, p.E, (slab0 + 1) * SLAB, V, ptid);
    __syncthreads();
    acc = slab_scores(e_buf, p1, r, h);
    make_g(slab0, gt_buf);
    __syncthreads();
    for (int i = 0; i < n; ++i) {
      const bool more = i + 1 < n;
      unsigned char *const e_next = e_buf + ((i + 1) & 1) * 3 * E_IMG;
      // phase a: split slab i + 1 (its fp32 rows arrived during the previous phases), fetch slab i + 2
      if (more) stage_store_split(st, e_next, ptid);
      if (i + 2 < n) stage_load(st, p.E, (slab0 + i + 2) * SLAB, V, ptid);
      __syncthreads();
      // phase b: scores of slab i + 1
      if (more) acc = slab_scores(e_next, p1, r, h);
      __syncthreads();
      // phase c: G of slab i + 1
      if (more) make_g(slab0 + i + 1, gt_buf + ((i + 1) & 1) * 3 * GT_IMG);
      __syncthreads();
    }
    return;
  }

  // ------------------------------------------------------------------ consumer waves
  const int dcol = 32 * w + r;        // this lane's output column in both backward products
  Tri p2[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = p.P[(size_t)min(16 * s + 8 * h + j, p.Bt - 1) * D + dcol];
    p2[s] = split8(x);
  }
  f32x16 dp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dp[i] = f32x16{0.f};
  float sq = 0.f;
  const unsigned char *const g_rd = g_img + r * G_PITCH + 16 * h;
  auto dpred_blocks = [&](const unsigned char *e_img, const unsigned char *gt_img, const int m0, const int m1) {
    const unsigned char *const gt_rd = gt_img + r * GT_PITCH + 16 * h;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Tri bfrag;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (p.lab & 4) {
          bfrag.t[t] = *reinterpret_cast<const bf16x8 *>(e_img + t * E_IMG + r * E_PITCH + 32 * s + 16 * h);
          continue;
        }
        uint16_t u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          u[j] = *reinterpret_cast<const uint16_t *>(e_img + t * E_IMG + (16 * s + 8 * h + j) * E_PITCH + dcol * 2);
        const u32x4 pk = {(uint32_t)u[0] | ((uint32_t)u[1] << 16), (uint32_t)u[2] | ((uint32_t)u[3] << 16),
                          (uint32_t)u[4] | ((uint32_t)u[5] << 16), (uint32_t)u[6] | ((uint32_t)u[7] << 16)};
        bfrag.t[t] = __builtin_bit_cast(bf16x8, pk);
      }
#pragma unroll
      for (int mblk = 0; mblk < 4; ++mblk) {
        if (mblk < m0 || mblk >= m1) continue;
        Tri a;
#pragma unroll
        for (int t = 0; t < 3; ++t)
          a.t[t] = *reinterpret_cast<const bf16x8 *>(gt_rd + t * GT_IMG + (32 * mblk) * GT_PITCH + 32 * s);
        dp[mblk] = mfma6(a, bfrag, dp[mblk]);
      }
    }
  };
  __syncthreads();       // (producer prologue: slab 0 staged)
  __syncthreads();       // (producer prologue: G of slab 0 in place)
  for (int i = 0; i < n; ++i) {
    const int vbase = (slab0 + i) * SLAB;
    const bool full = vbase + SLAB <= V;
    const unsigned char *const e_cur = e_buf + (i & 1) * 3 * E_IMG;
    const unsigned char *const gt_cur = gt_buf + (i & 1) * 3 * GT_IMG;
    // phase a: dE[v][d] = sum_b G[v][b] pred[b][d]   (this wave: columns d = 32 w ..)
    {
      f32x16 acc = {0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        Tri a;
#pragma unroll
        for (int t = 0; t < 3; ++t) a.t[t] = *reinterpret_cast<const bf16x8 *>(g_rd + t * G_IMG + 32 * s);
        acc = mfma6(a, p2[s], acc);
      }
      float *const out = p.dE + ((size_t)vbase + 4 * h) * D + dcol;
      if (RMW) {
        float old[16];
#pragma unroll
        for (int q = 0; q < 16; ++q)
          old[q] = out[(long)min((q & 3) + 8 * (q >> 2), V - 1 - vbase - 4 * h) * D];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += old[q];
      }
      // The tile has its column on the lane: stored as it lies that is 16 four-byte stores of 128-byte segments per
      // lane (store ISSUE, 0.6 ms of the pass at 10 M rows).  Through a wave-private LDS scratch it leaves as four
      // 16-byte stores per lane (8 rows x 128 B per wave instruction).
      if (full && !RMW) {
        float *const sc = t_scratch + w * (32 * T_PITCH);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          sq = fmaf(acc[q], acc[q], sq);
          sc[((q & 3) + 8 * (q >> 2) + 4 * h) * T_PITCH + r] = acc[q];
        }
        if (!(p.lab & 1)) {
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) {
            const int idx = i4 * 64 + lane, row = idx >> 3, c4 = idx & 7;
            const f32x4 t = *reinterpret_cast<const f32x4 *>(sc + row * T_PITCH + 4 * c4);
            *reinterpret_cast<f32x4 *>(p.dE + ((size_t)vbase + row) * D + 32 * w + 4 * c4) = t;
          }
        }
      } else if (full) {
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
    __syncthreads();
    // phases b, c: d_pred[b][d] += sum_v G[v][b] E[v][d]   (this wave: columns d = 32 w .., two 32-row blocks each)
    // (one block beside the producers' 48 score MFMAs, three beside their exp + split: 60 / 36 MFMAs per SIMD in
    // phases b / c instead of 72 / 24)
    dpred_blocks(e_cur, gt_cur, 0, 1);
    __syncthreads();
    dpred_blocks(e_cur, gt_cur, 1, 4);
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

}  // namespace x3

// which form a catalog of V rows is scored with: the split-bf16 kernels from MTAM_SCORE32_SPLIT_MIN_ROWS rows on
// (default 65,536; 0 = never).  Below that a step is bound by launches, not by the matrix rate.
bool use_split(int V) {
  static const long min_rows = [] {
    const char *e = getenv("MTAM_SCORE32_SPLIT_MIN_ROWS");
    return e ? atol(e) : 65536L;
  }();
  return min_rows > 0 && V >= min_rows;
}
int lab_bits() {
  const char *e = getenv("MTAM_SCORE32_LAB");      // read at every call: the lab flips it between timings
  return e ? atoi(e) : 0;
}
bool use_pc() {
  static const bool on = [] {
    const char *e = getenv("MTAM_SCORE32_PC");
    return !(e && e[0] == '0');
  }();
  return on;
}

int slabs_of(int V) { return (V + SLAB - 1) / SLAB; }
// Workgroups per batch tile: each owns a contiguous range of slabs and flushes its [128, 128] share of d_pred
// ONCE, by atomics (64 KB each at the ~1.3 TB/s float-atomic rate: 2,048 workgroups = 100 us, 512 = 25 us).
// The backward kernel holds 336 registers per lane = one workgroup per CU, so 512 ranges are two rounds over
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
int lse_slabs_per_wg_of(int V) { return (slabs_of(V) + lse_chunks_of(V) - 1) / lse_chunks_of(V); }
int lse_grid_of(int V) { return (slabs_of(V) + lse_slabs_per_wg_of(V) - 1) / lse_slabs_per_wg_of(V); }
int slabs_per_wg_of(int V) { return (slabs_of(V) + chunks_of(V) - 1) / chunks_of(V); }
int grid_of(int V) { return (slabs_of(V) + slabs_per_wg_of(V) - 1) / slabs_per_wg_of(V); }

}  // namespace

extern "C" int mtam_score32_partials(int B, int V) { return B * lse_grid_of(V) * 2; }
extern "C" int mtam_score32_sq_partials(int V) { return grid_of(V) * 4; }

extern "C" int mtam_score32_lse(const float *E, const float *pred, const int32_t *target, int B, int V,
                                float *partial, float *lse, float *ce, void *stream) {
  MTAM_CHECK_ARG(E && pred && target && partial && lse && ce, "score32_lse: null argument");
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
  hipLaunchKernelGGL(score32_finish_kernel, dim3(B), dim3(256), 0, s, E, pred, target, V, grid, partial, lse, ce);
  MTAM_CHECK_LAUNCH("score32_lse");
  return MTAM_OK;
}

extern "C" int mtam_score32_bwd(const float *E, const float *pred, const float *lse, const int32_t *target, int B,
                                int V, float scale, float *d_pred, float *dE, float *sq_partial, void *stream) {
  MTAM_CHECK_ARG(E && pred && lse && target && d_pred && dE, "score32_bwd: null argument");
  MTAM_CHECK_ARG(B > 0 && V > 0 && V < 0x7fffff00 && scale > 0.f, "score32_bwd: bad shape B=%d V=%d", B, V);
  MTAM_CHECK_ARG(mtam_aligned16(E) && mtam_aligned16(pred), "score32_bwd: operands must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool split = use_split(V);
  if (split) {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(x3::bwd_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, x3::BWD_LDS);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(x3::bwd_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, x3::BWD_LDS);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(x3::bwd_pc_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, x3::PC_LDS);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(x3::bwd_pc_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, x3::PC_LDS);
      MTAM_CHECK_ARG(e == hipSuccess, "score32_bwd: cannot reserve %d bytes of LDS: %s", x3::BWD_LDS,
                     hipGetErrorString(e));
      attr_set = true;
    }
  }
  // one launch per 128-row batch tile; a later tile adds onto the dE rows the earlier one stored
  const int ntile = (B + BT - 1) / BT;
  for (int tile = 0; tile < ntile; ++tile) {
    const long b0 = (long)tile * BT;
    BwdArgs a{E, pred + b0 * D, lse + b0, target + b0, V, (int)min((long)BT, B - b0), slabs_per_wg_of(V), scale,
              d_pred + b0 * D, dE, tile == ntile - 1 ? sq_partial : nullptr, lab_bits()};
    if (split && use_pc() && tile == 0)
      hipLaunchKernelGGL(x3::bwd_pc_kernel<false>, dim3(grid_of(V)), dim3(512), x3::PC_LDS, st, a);
    else if (split && use_pc())
      hipLaunchKernelGGL(x3::bwd_pc_kernel<true>, dim3(grid_of(V)), dim3(512), x3::PC_LDS, st, a);
    else if (split && tile == 0)
      hipLaunchKernelGGL(x3::bwd_kernel<false>, dim3(grid_of(V)), dim3(256), x3::BWD_LDS, st, a);
    else if (split)
      hipLaunchKernelGGL(x3::bwd_kernel<true>, dim3(grid_of(V)), dim3(256), x3::BWD_LDS, st, a);
    else if (tile == 0)
      hipLaunchKernelGGL(score32_bwd_kernel<false>, dim3(grid_of(V)), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(score32_bwd_kernel<true>, dim3(grid_of(V)), dim3(256), 0, st, a);
  }
  MTAM_CHECK_LAUNCH("score32_bwd");
  return MTAM_OK;
}
