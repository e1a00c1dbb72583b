// Full-catalog scoring with a bf16 copy of the item table and bf16 MFMA (fp32 accumulation), without
// ever materialising the [B, V] logits in training: base_model.output (Model/base_model.py:300-328)
//   logits = pred . E^T ; loss_b = logsumexp(logits_b) - logits_b[target_b]
// and its gradient (Model/base_model.py:290-297 via tf.gradients)
//   G = (softmax(logits) - onehot(target)) / B ;  d_pred = G . E ;  dE = G^T . pred
// for BASELINE.json configs[4] (50 M items: the fp32 logits alone are 25.6 GB per pass).
//
//   score16_lse     one pass over the table: per-row running (max, sum-exp) -> lse, cross entropy
//   score16_bwd     second pass: recomputes the scores of a 64-row slab, forms G in registers from the
//                   saved lse, and produces BOTH products from it (d_pred accumulated per workgroup and
//                   flushed once by fp32 atomics; dE stored, its squared norm summed on the way out)
//   score16_logits  evaluation only: the same tile product, stored as fp32 logits for top-K
//
// HBM traffic per training step: the bf16 table twice (2 x V x 256 B) + the fp32 gradient once
// (V x 512 B) -- against seven [B, V] fp32 passes plus two table reads in the fp32 path.  Both
// operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32), products are exact in fp32
// and accumulate in fp32; G is rounded to bf16 before the two backward products.
//
// One workgroup = 4 waves = one 128-row batch tile x a contiguous range of 64-row catalog slabs.
// v_mfma_f32_32x32x16_bf16 throughout: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j],
// B[k = 8h + j][col r]; the result has its column on the lane and row (reg & 3) + 8 (reg >> 2) + 4 h
// in register reg.  Wave w owns batch rows 32w .. 32w+31 for the score tile S[v][b] (so the softmax
// state of a batch row lives in one lane) and output columns d = 32w .. 32w+31 for both backward
// products; G crosses waves through LDS in both orientations ([v][b] feeds dE, [b][v] feeds d_pred).
#include "common.h"

namespace {

constexpr int D = MTAM_D;
constexpr int SLAB = 64;        // catalog rows per iteration
constexpr int BT = 128;         // batch rows per tile
constexpr int E_PITCH = 272;    // bytes per staged table row: 256 + 16 (ds_read_b128 of 16 rows hit 16 distinct bank groups)
constexpr int G_PITCH = 272;    // G[v][b]: 128 b x 2 B + 16
constexpr int GT_PITCH = 144;   // G^T[b][v]: 64 v x 2 B + 16
constexpr int E_BYTES = SLAB * E_PITCH;
constexpr int G_BYTES = SLAB * G_PITCH;
constexpr int GT_BYTES = BT * GT_PITCH;
constexpr int BWD_LDS = 2 * E_BYTES + G_BYTES + GT_BYTES;
constexpr float L2E = 1.4426950408889634f;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t u) { return __uint_as_float((uint32_t)u << 16); }

// One slab of the table (64 rows x 256 B) in flight between global memory and LDS: 4 x 16 B per thread,
// each wave-instruction reading 1 KiB contiguous.  Rows past the end re-read the last row (their
// scores are masked), so the loads are unconditional.
struct Stage {
  u32x4 v[4];
};
__device__ __forceinline__ void stage_load(Stage &st, const uint16_t *__restrict__ E, long v0, int V, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const long v = min(v0 + (c >> 4), (long)V - 1);
    st.v[i] = *reinterpret_cast<const u32x4 *>(E + v * D + (c & 15) * 8);
  }
}
__device__ __forceinline__ void stage_store(const Stage &st, unsigned char *buf, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    *reinterpret_cast<u32x4 *>(buf + (c >> 4) * E_PITCH + (c & 15) * 16) = st.v[i];
  }
}
// fragment of table rows 32 mb .. 32 mb + 31, k-step s (d = 16 s + 8 h ..)
__device__ __forceinline__ bf16x8 e_frag(const unsigned char *buf, int mb, int s, int r, int h) {
  return *reinterpret_cast<const bf16x8 *>(buf + (32 * mb + r) * E_PITCH + 32 * s + 16 * h);
}
// the wave's 32 batch rows as an operand (rows b, k = d): the same registers serve as A (rows) or B (columns)
__device__ __forceinline__ void load_pred_rows(bf16x8 (&p1)[8], const uint16_t *__restrict__ P, long b, int h) {
#pragma unroll
  for (int s = 0; s < 8; ++s) p1[s] = *reinterpret_cast<const bf16x8 *>(P + b * D + 16 * s + 8 * h);
}

// ------------------------------------------------------------------ conversions
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float *__restrict__ src, size_t n_src4,
                                                          uint16_t *__restrict__ dst, size_t n_dst4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_dst4; i += stride) {
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (i < n_src4) x = *reinterpret_cast<const f32x4 *>(src + 4 * i);
    bf16x4 y;
    y.x = (__bf16)x.x; y.y = (__bf16)x.y; y.z = (__bf16)x.z; y.w = (__bf16)x.w;
    *reinterpret_cast<bf16x4 *>(dst + 4 * i) = y;
  }
}

// ------------------------------------------------------------------ forward: log-sum-exp without logits
__global__ __launch_bounds__(256) void score16_lse_kernel(const uint16_t *__restrict__ E,
                                                          const uint16_t *__restrict__ P, int V, int slabs_per_wg,
                                                          float *__restrict__ partial) {
  __shared__ __attribute__((aligned(16))) unsigned char e_lds[2][E_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int chunks = gridDim.x, c = blockIdx.x;
  const long b = (long)blockIdx.y * BT + 32 * w + r;
  const int nslab = (V + SLAB - 1) / SLAB;
  const int slab0 = min(c * slabs_per_wg, nslab), slab1 = min(nslab, slab0 + slabs_per_wg);
  bf16x8 p1[8];
  load_pred_rows(p1, P, b, h);
  float m = -INFINITY, ssum = 0.f;
  // `ready` holds the next slab, `issue` receives the one after it: two slabs in flight per workgroup
  // (one in flight left the kernel waiting on HBM latency: 4.8 TB/s of table bytes at 3 workgroups per CU)
  auto step = [&](const int sl, const unsigned char *e_cur, unsigned char *e_next, Stage &ready, Stage &issue) {
    if (sl + 2 < slab1) stage_load(issue, E, (long)(sl + 2) * SLAB, V, tid);
    f32x16 acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      acc[mb] = f32x16{0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s)
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e_frag(e_cur, mb, s, r, h), p1[s], acc[mb], 0, 0, 0);
    }
    const int vbase = sl * SLAB + 4 * h;
    const int vlim = (sl * SLAB + SLAB <= V) ? 0x7fffffff : V;
    float mx = m;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int v = vbase + 32 * mb + (q & 3) + 8 * (q >> 2);
        const float x = (v < vlim) ? acc[mb][q] : -INFINITY;
        acc[mb][q] = x;
        mx = fmaxf(mx, x);
      }
    const float ref = (mx == -INFINITY) ? 0.f : mx;
    const float nref = -ref * L2E;
    float add = 0.f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int q = 0; q < 16; ++q) add += fast_exp2(fmaf(acc[mb][q], L2E, nref));
    ssum = ssum * fast_exp2(fmaf(m, L2E, nref)) + add;
    m = mx;
    if (sl + 1 < slab1) stage_store(ready, e_next, tid);
    __syncthreads();
  };
  Stage sa, sb;
  if (slab0 < slab1) {
    stage_load(sa, E, (long)slab0 * SLAB, V, tid);
    stage_store(sa, e_lds[0], tid);
    if (slab0 + 1 < slab1) stage_load(sa, E, (long)(slab0 + 1) * SLAB, V, tid);
  }
  __syncthreads();
  for (int sl = slab0; sl < slab1; sl += 2) {
    step(sl, e_lds[0], e_lds[1], sa, sb);
    if (sl + 1 < slab1) step(sl + 1, e_lds[1], e_lds[0], sb, sa);
  }
  // the two lane halves hold different catalog rows of the same batch row
  const float m2 = __shfl_xor(m, 32, 64), s2 = __shfl_xor(ssum, 32, 64);
  const float mm = fmaxf(m, m2), ref = (mm == -INFINITY) ? 0.f : mm;
  const float ss = ssum * fast_exp2((m - ref) * L2E) + s2 * fast_exp2((m2 - ref) * L2E);
  if (h == 0) {
    partial[((size_t)b * chunks + c) * 2 + 0] = mm;
    partial[((size_t)b * chunks + c) * 2 + 1] = ss;
  }
}

__global__ __launch_bounds__(256) void score16_finish_kernel(const uint16_t *__restrict__ E,
                                                             const uint16_t *__restrict__ P,
                                                             const int32_t *__restrict__ target, int V, int chunks,
                                                             const float *__restrict__ partial,
                                                             float *__restrict__ lse, float *__restrict__ ce) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *pp = partial + (size_t)b * chunks * 2;
  float m = -INFINITY;
  for (int c = tid; c < chunks; c += 256) m = fmaxf(m, pp[2 * c]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < chunks; c += 256) s += pp[2 * c + 1] * expf(pp[2 * c] - m);
  // target logit: the same bf16 operands, fp32 products and sum
  const long t = min(max(target[b], 0), V - 1);
  float dot = 0.f;
  if (tid < D) dot = bf16_bits_to_f32(P[(size_t)b * D + tid]) * bf16_bits_to_f32(E[t * D + tid]);
  s = wave_sum(s);
  dot = wave_sum(dot);
  __shared__ float red2[4];
  if ((tid & 63) == 0) {
    red[tid >> 6] = s;
    red2[tid >> 6] = dot;
  }
  __syncthreads();
  if (tid == 0) {
    const float l = m + logf(red[0] + red[1] + red[2] + red[3]);
    lse[b] = l;
    ce[b] = l - (red2[0] + red2[1] + red2[2] + red2[3]);
  }
}

// ------------------------------------------------------------------ evaluation: fp32 logits for top-K
__global__ __launch_bounds__(256) void score16_logits_kernel(const uint16_t *__restrict__ E,
                                                             const uint16_t *__restrict__ P, int V, int B,
                                                             int slabs_per_wg, float *__restrict__ logits, long ld) {
  __shared__ __attribute__((aligned(16))) unsigned char e_lds[2][E_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const long b0 = (long)blockIdx.y * BT + 32 * w;
  const int nslab = (V + SLAB - 1) / SLAB;
  const int slab0 = min((int)blockIdx.x * slabs_per_wg, nslab), slab1 = min(nslab, slab0 + slabs_per_wg);
  bf16x8 p1[8];
  load_pred_rows(p1, P, b0 + r, h);
  Stage st;
  if (slab0 < slab1) {
    stage_load(st, E, (long)slab0 * SLAB, V, tid);
    stage_store(st, e_lds[0], tid);
  }
  __syncthreads();
  for (int sl = slab0; sl < slab1; ++sl) {
    const int cur = (sl - slab0) & 1;
    const bool more = sl + 1 < slab1;
    if (more) stage_load(st, E, (long)(sl + 1) * SLAB, V, tid);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      // operands swapped: rows = batch, column (on the lane) = catalog row -> 128-byte contiguous stores
      f32x16 acc = {0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1[s], e_frag(e_lds[cur], mb, s, r, h), acc, 0, 0, 0);
      const long v = (long)sl * SLAB + 32 * mb + r;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const long b = b0 + acc_row(q, h);
        if (b < B && v < V) logits[b * ld + v] = acc[q];
      }
    }
    if (more) stage_store(st, e_lds[cur ^ 1], tid);
    __syncthreads();
  }
}

// ------------------------------------------------------------------ backward: G, d_pred and dE in one pass
struct BwdArgs {
  const uint16_t *E, *P;      // P, lse, target, d_pred: already moved to this launch's 128-row batch tile
  const float *lse;
  const int32_t *target;
  int V, Bt, slabs_per_wg;    // Bt = batch rows of this tile that exist (1 .. 128)
  float scale;
  float *d_pred, *dE, *sq_partial;
};

#ifndef S16_BWD_WAVES_PER_EU
#define S16_BWD_WAVES_PER_EU 2      // 256 registers: two workgroups per CU (1.65 ms vs 2.24 ms at 10 M rows)
#endif
// S16_LAB_* are switches of the developer lab (tools/score16_lab.hip) that cut parts of the kernel out to
// time the rest; the library is built with none of them.
//
// RMW: a batch tile after the first adds onto the dE rows the previous launch stored (B > 128).
// Everything indexed by catalog row is 32-bit and branch-free: the first build of this kernel used 64-bit
// row numbers and `valid && v < V` conditions, which the compiler turned into one exec-mask branch per
// element (10,000 cycles per slab, 2.6 TB/s of the kernel's algorithmic bytes at one wave per SIMD).
template <bool RMW>
__global__ __launch_bounds__(256, S16_BWD_WAVES_PER_EU) void score16_bwd_kernel(BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *const e_lds0 = lds, *const e_lds1 = lds + E_BYTES;
  unsigned char *const g_lds = lds + 2 * E_BYTES, *const gt_lds = g_lds + G_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int V = p.V;
  const int nslab = (V + SLAB - 1) / SLAB;
  const int slab0 = min((int)blockIdx.x * p.slabs_per_wg, nslab), slab1 = min(nslab, slab0 + p.slabs_per_wg);
  const int dcol = 32 * w + r;        // this lane's output column in both backward products
  const int bcol = 32 * w + r;        // the batch row (of the tile) whose scores sit on this lane
  const bool valid_b = bcol < p.Bt;
  // G = exp2(score * log2 e + c_b) - [v == target] * scale, c_b = -lse * log2 e + log2 scale; a batch row
  // that does not exist gets c_b = -inf and no target: G = 0 without a mask
  const float c_b = valid_b ? fmaf(-p.lse[min(bcol, p.Bt - 1)], L2E, log2f(p.scale)) : -INFINITY;
  const int t_b = valid_b ? min(max(p.target[min(bcol, p.Bt - 1)], 0), V - 1) : -1;
  bf16x8 p1[8], p2[8];
  load_pred_rows(p1, p.P, bcol, h);
  // pred[k = b][n = d] fragments for dE (k-step s covers batch rows 16 s .. 16 s + 15 of the tile)
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    uint16_t u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = p.P[(16 * s + 8 * h + j) * D + dcol];
    u32x4 pk = {(uint32_t)u[0] | ((uint32_t)u[1] << 16), (uint32_t)u[2] | ((uint32_t)u[3] << 16),
                (uint32_t)u[4] | ((uint32_t)u[5] << 16), (uint32_t)u[6] | ((uint32_t)u[7] << 16)};
    p2[s] = __builtin_bit_cast(bf16x8, pk);
  }
  f32x16 dp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dp[i] = f32x16{0.f};
  float sq = 0.f;

  // lane-constant LDS addresses (the 4 h of the accumulator row map folded in)
  unsigned char *const g_wr = g_lds + (4 * h) * G_PITCH + (32 * w + r) * 2;       // G[v][b], one element
  unsigned char *const gt_wr = gt_lds + (32 * w + r) * GT_PITCH + (4 * h) * 2;    // G^T[b][v], four rows of v
  const unsigned char *const g_rd = g_lds + r * G_PITCH + 16 * h;
  const unsigned char *const gt_rd = gt_lds + r * GT_PITCH + 16 * h;

  // One slab.  `ready` holds the NEXT slab (loaded one step ago), `issue` receives the one after it: two
  // slabs (32 KiB per workgroup) are in flight while this one is computed.
  auto step = [&](const int sl, const unsigned char *e_cur, unsigned char *e_next, Stage &ready, Stage &issue) {
    if (sl + 2 < slab1) stage_load(issue, p.E, (long)(sl + 2) * SLAB, V, tid);
    const int vbase = sl * SLAB;
    const bool full = vbase + SLAB <= V;
    const int vlim = full ? 0x7fffffff : V;

    // ---- S[v][b] for this wave's 32 batch rows, then G to LDS as bf16 in both orientations
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x16 acc = {0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e_frag(e_cur, mb, s, r, h), p1[s], acc, 0, 0, 0);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        bf16x4 gq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 32 * mb + 8 * q4 + i;           // + 4 h: catalog row of the slab
          const int v = vbase + row + 4 * h;
#ifdef S16_LAB_NO_EXP
          float g = acc[4 * q4 + i] - ((v == t_b) ? p.scale : 0.f);
#else
          float g = fast_exp2(fmaf(acc[4 * q4 + i], L2E, c_b)) - ((v == t_b) ? p.scale : 0.f);
#endif
          g = (v < vlim) ? g : 0.f;
          const __bf16 gb = (__bf16)g;
          gq[i] = gb;
          *reinterpret_cast<__bf16 *>(g_wr + row * G_PITCH) = gb;
        }
        // registers 4 q4 .. 4 q4 + 3 are catalog rows 8 q4 + 4 h + 0..3: contiguous in G^T[b][v]
        *reinterpret_cast<bf16x4 *>(gt_wr + (32 * mb + 8 * q4) * 2) = gq;
      }
    }
    __syncthreads();

    // ---- dE[v][d] = sum_b G[v][b] pred[b][d]   (this wave: columns d = 32 w ..)
#ifndef S16_LAB_NO_DE
    float *const out = p.dE + ((size_t)vbase + 4 * h) * D + dcol;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x16 acc = {0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(g_rd + (32 * mb) * G_PITCH + 32 * s);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p2[s], acc, 0, 0, 0);
      }
      if (RMW) {
        float old[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = 32 * mb + (q & 3) + 8 * (q >> 2);
          old[q] = out[(long)min(row, V - 1 - vbase - 4 * h) * D];      // clamped: rows past the end are not stored
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += old[q];
      }
      if (full) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          sq = fmaf(acc[q], acc[q], sq);
#ifndef S16_LAB_NO_DE_STORE
          out[(size_t)(32 * mb + (q & 3) + 8 * (q >> 2)) * D] = acc[q];
#endif
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = 32 * mb + (q & 3) + 8 * (q >> 2);
          if (vbase + row + 4 * h < V) {
            sq = fmaf(acc[q], acc[q], sq);
            out[(size_t)row * D] = acc[q];
          }
        }
      }
    }
#endif

    // ---- d_pred[b][d] += sum_v G[v][b] E[v][d]   (this wave: columns d = 32 w .., all 128 batch rows)
#ifndef S16_LAB_NO_DPRED
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint16_t u[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        u[j] = *reinterpret_cast<const uint16_t *>(e_cur + (16 * s + 8 * h + j) * E_PITCH + dcol * 2);
      u32x4 pk = {(uint32_t)u[0] | ((uint32_t)u[1] << 16), (uint32_t)u[2] | ((uint32_t)u[3] << 16),
                  (uint32_t)u[4] | ((uint32_t)u[5] << 16), (uint32_t)u[6] | ((uint32_t)u[7] << 16)};
      const bf16x8 bfrag = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
      for (int mblk = 0; mblk < 4; ++mblk) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(gt_rd + (32 * mblk) * GT_PITCH + 32 * s);
        dp[mblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag, dp[mblk], 0, 0, 0);
      }
    }
#endif
    if (sl + 1 < slab1) stage_store(ready, e_next, tid);
    __syncthreads();
  };

  Stage sa, sb;
  if (slab0 < slab1) {
    stage_load(sa, p.E, (long)slab0 * SLAB, V, tid);
    stage_store(sa, e_lds0, tid);
    if (slab0 + 1 < slab1) stage_load(sa, p.E, (long)(slab0 + 1) * SLAB, V, tid);
  }
  __syncthreads();
  for (int sl = slab0; sl < slab1; sl += 2) {
    step(sl, e_lds0, e_lds1, sa, sb);
    if (sl + 1 < slab1) step(sl + 1, e_lds1, e_lds0, sb, sa);
  }

  // flush this workgroup's share of d_pred
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

int slabs_of(int V) { return (V + SLAB - 1) / SLAB; }
// workgroups along the catalog: every one gets a contiguous range of slabs; enough of them to fill
// the chip twice over, few enough that per-workgroup set-up and the d_pred flush stay small
int chunks_of(int V) { return max(1, min(slabs_of(V), 2048)); }
int slabs_per_wg_of(int V) { return (slabs_of(V) + chunks_of(V) - 1) / chunks_of(V); }
int grid_of(int V) { return (slabs_of(V) + slabs_per_wg_of(V) - 1) / slabs_per_wg_of(V); }
int bpad_of(int B) { return (B + BT - 1) / BT * BT; }

}  // namespace

extern "C" int mtam_f32_to_bf16(const float *src, size_t n_src, uint16_t *dst, size_t n_dst, void *stream) {
  MTAM_CHECK_ARG(src && dst && n_dst >= n_src && n_src % 4 == 0 && n_dst % 4 == 0,
                 "f32_to_bf16: counts must be multiples of 4 with n_dst >= n_src");
  MTAM_CHECK_ARG(mtam_aligned16(src) && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0, "f32_to_bf16: alignment");
  if (n_dst == 0) return MTAM_OK;
  const size_t n4 = n_dst / 4;
  const int blocks = (int)min((size_t)8192, (n4 + 255) / 256);
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                     n_src / 4, dst, n4);
  MTAM_CHECK_LAUNCH("f32_to_bf16");
  return MTAM_OK;
}

extern "C" int mtam_score16_batch_pad(int B) { return bpad_of(B); }
extern "C" int mtam_score16_partials(int B, int V) { return bpad_of(B) * grid_of(V) * 2; }
extern "C" int mtam_score16_sq_partials(int V) { return grid_of(V) * 4; }

extern "C" int mtam_score16_lse(const uint16_t *E16, const uint16_t *P16, const int32_t *target, int B, int V,
                                float *partial, float *lse, float *ce, void *stream) {
  MTAM_CHECK_ARG(E16 && P16 && target && partial && lse && ce, "score16_lse: null argument");
  MTAM_CHECK_ARG(B > 0 && V > 0 && bpad_of(B) / BT <= 65535, "score16_lse: bad shape B=%d V=%d", B, V);
  MTAM_CHECK_ARG(mtam_aligned16(E16) && mtam_aligned16(P16), "score16_lse: operands must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_of(V);
  hipLaunchKernelGGL(score16_lse_kernel, dim3(grid, bpad_of(B) / BT), dim3(256), 0, s, E16, P16, V,
                     slabs_per_wg_of(V), partial);
  hipLaunchKernelGGL(score16_finish_kernel, dim3(B), dim3(256), 0, s, E16, P16, target, V, grid, partial, lse, ce);
  MTAM_CHECK_LAUNCH("score16_lse");
  return MTAM_OK;
}

extern "C" int mtam_score16_logits(const uint16_t *E16, const uint16_t *P16, int B, int V, float *logits, long ld,
                                   void *stream) {
  MTAM_CHECK_ARG(E16 && P16 && logits, "score16_logits: null argument");
  MTAM_CHECK_ARG(B > 0 && V > 0 && ld >= V && bpad_of(B) / BT <= 65535, "score16_logits: bad shape");
  MTAM_CHECK_ARG(mtam_aligned16(E16) && mtam_aligned16(P16), "score16_logits: operands must be 16-byte aligned");
  hipLaunchKernelGGL(score16_logits_kernel, dim3(grid_of(V), bpad_of(B) / BT), dim3(256), 0,
                     static_cast<hipStream_t>(stream), E16, P16, V, B, slabs_per_wg_of(V), logits, ld);
  MTAM_CHECK_LAUNCH("score16_logits");
  return MTAM_OK;
}

extern "C" int mtam_score16_bwd(const uint16_t *E16, const uint16_t *P16, const float *lse, const int32_t *target,
                                int B, int V, float scale, float *d_pred, float *dE, float *sq_partial,
                                void *stream) {
  MTAM_CHECK_ARG(E16 && P16 && lse && target && d_pred && dE, "score16_bwd: null argument");
  MTAM_CHECK_ARG(B > 0 && V > 0 && V < 0x7fffff00 && scale > 0.f, "score16_bwd: bad shape B=%d V=%d", B, V);
  MTAM_CHECK_ARG(mtam_aligned16(E16) && mtam_aligned16(P16), "score16_bwd: operands must be 16-byte aligned");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(score16_bwd_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(score16_bwd_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    MTAM_CHECK_ARG(e == hipSuccess, "score16_bwd: cannot reserve %d bytes of LDS: %s", BWD_LDS, hipGetErrorString(e));
    attr_set = true;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // one launch per 128-row batch tile; launches on one stream run in order, so a later tile's
  // read-modify-write of dE sees the earlier one's stores.  The squared norm comes from the last tile.
  const int ntile = (B + BT - 1) / BT;
  for (int tile = 0; tile < ntile; ++tile) {
    const long b0 = (long)tile * BT;
    BwdArgs a{E16, P16 + b0 * D, lse + b0, target + b0, V, (int)min((long)BT, B - b0), slabs_per_wg_of(V), scale,
              d_pred + b0 * D, dE, tile == ntile - 1 ? sq_partial : nullptr};
    if (tile == 0)
      hipLaunchKernelGGL(score16_bwd_kernel<false>, dim3(grid_of(V)), dim3(256), BWD_LDS, st, a);
    else
      hipLaunchKernelGGL(score16_bwd_kernel<true>, dim3(grid_of(V)), dim3(256), BWD_LDS, st, a);
  }
  MTAM_CHECK_LAUNCH("score16_bwd");
  return MTAM_OK;
}
