// Time-aware GRU (TimeAwareGRUCell_decay_new under dynamic_rnn) forward and
// backward-through-time: Model/Modules/time_aware_rnn.py:186-269,
// Model/Modules/gru.py:69-77, plus gather_indexes(seq_len - 2)
// (Model/Modules/net_utils.py:82-92, Model/MTAMRec_model.py:75-79).
//
// The recurrence is a chain of seq_len-1 dependent steps per sample, each a
// [1,128]x[128,256] and a [1,128]x[128,128] product: latency-bound, not
// bandwidth- or MFMA-bound.  The input halves of both products are hoisted out
// of the loop into one GEMM (xproj); what stays serial is the recurrent half.
// One 512-thread workgroup (8 waves) owns one sample and keeps the recurrent
// weights (128 x 384 fp32 = 196 KB, more than the 160 KB LDS) in registers for
// the whole sequence -- 96 per lane.
//
// Round 2 layout (two barriers per step instead of four): the weights are cut by OUTPUT column, and a
// column's contraction is split over the 8 lanes of an octet (lane kp of the octet owns k in
// [16 kp, 16 kp + 16)).  The 8 partial sums meet inside the wave through DPP adds (quad_perm, row_shl:4),
// so no partial sum crosses waves through LDS: only r*h and u (after the gate phase) and the new state
// (after the candidate phase) do.  Wave w owns gate columns [32 w, 32 w + 32) (octet o: 4 of them) and
// candidate columns [16 w, 16 w + 16) (octet o: 2 of them); the lane that ends up with a candidate column's
// sum also owns that column's time gate and state update, which therefore never touch LDS.  A step reads
// the 128 state values as 4 ds_read_b128 per lane (8 distinct 64-byte chunks per instruction, laid out
// [64][4 pad][64] so that chunks kp and kp + 4 fall on different banks).  Round 1 cut the weights by
// k-slice per wave: 8 partial sums per column crossed waves through LDS, 4 barriers and ~2,400 cycles per step.
#include "common.h"

// In-kernel stamps for tools/gru_lab.hip (a diagnostic build, -DMTAM_GRU_STAMPS): cycles per segment of a
// step, summed per wave of workgroup 0 and written to a buffer of their own.  The product build has none.
#ifdef MTAM_GRU_STAMPS
__device__ unsigned long long g_gru_stamps[2][8][8];      // [kernel][wave][segment]
#define GRU_STAMP_DECL unsigned long long st_last_, st_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define GRU_STAMP_START asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last_)::"memory");
#define GRU_STAMP(i)                                                                   \
  {                                                                                    \
    unsigned long long t_;                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    st_acc_[i] += t_ - st_last_;                                                       \
    st_last_ = t_;                                                                     \
  }
#define GRU_STAMP_DUMP(k)                                                              \
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                      \
    for (int i_ = 0; i_ < 8; ++i_) g_gru_stamps[k][threadIdx.x >> 6][i_] = st_acc_[i_];
#else
#define GRU_STAMP_DECL
#define GRU_STAMP_START
#define GRU_STAMP(i)
#define GRU_STAMP_DUMP(k)
#endif

namespace {

constexpr int D = MTAM_D;

// tvec rows
enum { KW1 = 0, KB1, HW1, W1, B1, KW2, W12, B12, NTV };

// DPP lane exchanges inside a row of 16 lanes (all lanes active at every use)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1;     // quad_perm [1, 0, 3, 2]
constexpr int DPP_XOR2 = 0x4E;     // quad_perm [2, 3, 0, 1]
constexpr int DPP_SHL4 = 0x104;    // row_shl:4 -- lane i reads lane i + 4 (0 past the row)

// Four per-lane partial sums (columns 0..3 of the octet) -> on lane kp < 4 of the octet the full 8-lane sum of
// column 2 (kp & 1) + (kp >> 1).  A reduce-scatter: 4 adds instead of 12.
__device__ __forceinline__ float octet_reduce4(float s0, float s1, float s2, float s3, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float keepA = b0 ? s2 : s0, sendA = b0 ? s0 : s2;
  float keepB = b0 ? s3 : s1, sendB = b0 ? s1 : s3;
  keepA += dpp_f<DPP_XOR1>(sendA);
  keepB += dpp_f<DPP_XOR1>(sendB);
  float keep = b1 ? keepB : keepA;
  const float send = b1 ? keepA : keepB;
  keep += dpp_f<DPP_XOR2>(send);
  keep += dpp_f<DPP_SHL4>(keep);
  return keep;
}
__device__ __forceinline__ int octet_col4(int kp) { return 2 * (kp & 1) + ((kp >> 1) & 1); }
// Two per-lane partial sums -> on lane kp < 2 the full sum of column kp.
__device__ __forceinline__ float octet_reduce2(float s0, float s1, int lane) {
  const bool b0 = lane & 1;
  float keep = b0 ? s1 : s0;
  const float send = b0 ? s0 : s1;
  keep += dpp_f<DPP_XOR1>(send);
  keep += dpp_f<DPP_XOR2>(keep);
  keep += dpp_f<DPP_SHL4>(keep);
  return keep;
}
// position of element i of a broadcast vector in LDS: 4 floats of padding after every 64
__device__ __forceinline__ int padpos(int i) { return i + 4 * (i >> 6); }

// ldx = floats per xproj row: 3 D (gate | gate | candidate), or 5 D for the T-SeqRec cell
// (TimeAwareGRUCell_sigmoid, Model/Modules/time_aware_rnn.py:19-131), whose two time gates do not depend
// on the state: their pre-activations  x Wk + tanh(t w + b) Wt + bias  are hoisted into columns 3 D .. 5 D
// of the same projection, and the step is  h' = u h sigmoid(now) + (1 - u) c sigmoid(last).
// The saved state then has a sixth slot (the `now` gate).
struct FwdArgs {
  const float *xproj, *x, *timelast;
  const int32_t *seq_len;
  const float *wh_g, *wh_c, *tvec;
  int B, L, ldx;
  float *hs, *short_out, *save;
};

__global__ __launch_bounds__(512) void tagru_fwd_kernel(FwdArgs p) {
  __shared__ __attribute__((aligned(16))) float h_s[D + 8];
  __shared__ __attribute__((aligned(16))) float rh_s[D + 8];
  __shared__ float u_s[D];

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kp = lane & 7, o = lane >> 3;
  const int steps = min(max(p.seq_len[b] - 1, 0), p.L);
  const size_t row0 = (size_t)b * p.L;
  const bool seqrec = p.ldx == 5 * D;
  // tvec == nullptr: the plain tf GRUCell (Model/Modules/gru.py:13-39) -- no time gate, T = 1
  const bool plain = p.tvec == nullptr && !seqrec;
  const int ldx = p.ldx, nsave = seqrec ? 6 : 5;

  // ---- recurrent weights -> registers: for this lane's k range [16 kp, 16 kp + 16), pairs along k
  const int gc0 = 32 * w + 4 * o;            // first of the octet's 4 gate columns
  const int cc0 = 16 * w + 2 * o;            // first of its 2 candidate columns
  f32x2 wg[4][8], wc[2][8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const float4 a = *reinterpret_cast<const float4 *>(p.wh_g + (size_t)(16 * kp + 2 * kk) * (2 * D) + gc0);
    const float4 c = *reinterpret_cast<const float4 *>(p.wh_g + (size_t)(16 * kp + 2 * kk + 1) * (2 * D) + gc0);
    wg[0][kk] = f32x2{a.x, c.x}; wg[1][kk] = f32x2{a.y, c.y};
    wg[2][kk] = f32x2{a.z, c.z}; wg[3][kk] = f32x2{a.w, c.w};
    const float2 e = *reinterpret_cast<const float2 *>(p.wh_c + (size_t)(16 * kp + 2 * kk) * D + cc0);
    const float2 f = *reinterpret_cast<const float2 *>(p.wh_c + (size_t)(16 * kp + 2 * kk + 1) * D + cc0);
    wc[0][kk] = f32x2{e.x, f.x}; wc[1][kk] = f32x2{e.y, f.y};
  }
  // owners: lane kp < 4 finishes gate column gcol (waves 0..3: r, waves 4..7: u);
  //         lane kp < 2 finishes candidate column ccol and owns its time gate and state update
  const bool g_own = kp < 4, c_own = kp < 2;
  const int gcol = gc0 + octet_col4(kp);
  const int ccol = cc0 + (kp & 1);
  const bool is_r = w < 4;
  float tv[NTV];
#pragma unroll
  for (int i = 0; i < NTV; ++i) tv[i] = (c_own && !plain && !seqrec) ? p.tvec[i * D + ccol] : 0.f;

  if (tid < D + 8) h_s[tid] = 0.f;

  // software prefetch of step t's inputs (independent of the recurrence).  Every lane loads, owner or not
  // (valid addresses, values unused): a load under a lane-dependent branch is waited for inside the branch
  const float *pa = seqrec ? p.xproj + 3 * D + ccol : (plain ? p.xproj + ccol : p.x + ccol);
  const float *pb = seqrec ? p.xproj + 4 * D + ccol : (plain ? p.xproj : p.timelast);
  const size_t sa = (seqrec || plain) ? ldx : D, sb = (seqrec || plain) ? ldx : 1;
  float n_g = 0.f, n_c = 0.f, n_a = 0.f, n_b = 0.f;
  auto prefetch = [&](int t) {
    const size_t r = row0 + t;
    n_g = p.xproj[r * ldx + gcol];                  // gate pre-activation, input half
    n_c = p.xproj[r * ldx + 2 * D + ccol];          // candidate pre-activation, input half
    n_a = pa[r * sa];                               // x_t (or the hoisted `now` gate)
    n_b = pb[r * sb];                               // dt  (or the hoisted `last` gate)
  };
  if (steps > 0) prefetch(0);
  __syncthreads();

  const int vpos = 16 * kp + 4 * (kp >> 2);        // this lane's 16 values of a padded 128-vector
  float h_own = 0.f;                               // state of column ccol (owner lanes)
  GRU_STAMP_DECL
  GRU_STAMP_START
  for (int t = 0; t < steps; ++t) {
    const float in_g = n_g, in_c = n_c, in_a = n_a, in_b = n_b;
    if (t + 1 < steps) prefetch(t + 1);
    const size_t row = row0 + t;

    // ---- phase 1: gate pre-activations (4 columns x 16 k per lane), reduced inside the octet
    float4 hq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) hq[q] = *reinterpret_cast<const float4 *>(&h_s[vpos + 4 * q]);
    const float h_g = h_s[padpos(gcol & (D - 1))];            // r owners: h of their column
    // the time gate of column ccol needs only x_t, dt and the OLD state: computed in the shadow of the reads
    float T = 1.f, N = 1.f;
    if (seqrec) {                  // (wave-uniform branches; non-owner lanes compute on zeros)
      N = fast_sigmoid(in_a);
      T = fast_sigmoid(in_b);
    } else if (!plain) {
      const float tw = fmaxf(in_a * tv[KW1] + tv[KB1] + h_own * tv[HW1], 0.f);
      const float ts = fmaxf(tv[W1] * in_b + tv[B1], 0.f);
      T = fast_sigmoid(tv[KW2] * tw + tv[W12] * ts + tv[B12]);
    }
    float gsum;
    {
      f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 lo = {hq[q].x, hq[q].y}, hi = {hq[q].z, hq[q].w};
        a0 = pk_fma(lo, wg[0][2 * q], a0); a1 = pk_fma(lo, wg[1][2 * q], a1);
        a2 = pk_fma(lo, wg[2][2 * q], a2); a3 = pk_fma(lo, wg[3][2 * q], a3);
        a0 = pk_fma(hi, wg[0][2 * q + 1], a0); a1 = pk_fma(hi, wg[1][2 * q + 1], a1);
        a2 = pk_fma(hi, wg[2][2 * q + 1], a2); a3 = pk_fma(hi, wg[3][2 * q + 1], a3);
      }
      GRU_STAMP(0)      // state reads, time gate, gate FMAs
      gsum = octet_reduce4(a0.x + a0.y, a1.x + a1.y, a2.x + a2.y, a3.x + a3.y, lane);
    }
    float r_keep = 0.f;
    if (g_own) {
      const float sg = fast_sigmoid(gsum + in_g);
      if (is_r) {
        r_keep = sg;
        rh_s[padpos(gcol)] = sg * h_g;
      } else {
        u_s[gcol - D] = sg;
      }
    }
    GRU_STAMP(1)        // octet reduce, sigmoid, LDS writes
    __syncthreads();
    GRU_STAMP(2)        // barrier 1

    // ---- phase 2: candidate pre-activation on r*h (2 columns x 16 k per lane), state update on the owner lane
    float csum;
    {
      float4 rq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) rq[q] = *reinterpret_cast<const float4 *>(&rh_s[vpos + 4 * q]);
      f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 lo = {rq[q].x, rq[q].y}, hi = {rq[q].z, rq[q].w};
        a0 = pk_fma(lo, wc[0][2 * q], a0); a1 = pk_fma(lo, wc[1][2 * q], a1);
        a0 = pk_fma(hi, wc[0][2 * q + 1], a0); a1 = pk_fma(hi, wc[1][2 * q + 1], a1);
      }
      csum = octet_reduce2(a0.x + a0.y, a1.x + a1.y, lane);
    }
    GRU_STAMP(3)        // r*h reads, candidate FMAs, reduce
    const float u = u_s[ccol];
    const float c = fast_tanh(csum + in_c);
    const float hn = u * h_own * N + (1.f - u) * c * T;
    if (c_own) h_s[padpos(ccol)] = hn;
    // Land the prefetched inputs of step t + 1 HERE, before this step's stores are issued: vmcnt counts loads
    // and stores in one queue, so a wait placed after the stores (where the compiler would put it: at the
    // loop's back edge) also waits out the stores' round trip on the serial path.  The loads have had the whole
    // step to arrive, and the previous step's stores are long done.
    GRU_STAMP(4)        // tanh, state update
    asm volatile("" : "+v"(n_g), "+v"(n_c), "+v"(n_a), "+v"(n_b));
    GRU_STAMP(5)        // prefetched inputs landed (vmcnt)
    if (g_own && is_r && p.save) p.save[row * (nsave * D) + gcol] = r_keep;
    if (c_own) {
      p.hs[row * D + ccol] = hn;
      if (p.save) {
        float *sv = p.save + row * (nsave * D) + ccol;
        sv[D] = u; sv[2 * D] = c; sv[3 * D] = T; sv[4 * D] = h_own;
        if (seqrec) sv[5 * D] = N;
      }
      h_own = hn;
    }
    GRU_STAMP(6)        // stores issued
    __syncthreads();
    GRU_STAMP(7)        // barrier 2
  }
  GRU_STAMP_DUMP(0)

  if (c_own) {
    p.short_out[(size_t)b * D + ccol] = (steps > 0) ? h_own : 0.f;
    for (int t = steps; t < p.L; ++t) p.hs[(row0 + t) * D + ccol] = 0.f;   // dynamic_rnn zero-fills dead steps
  }
}

struct BwdArgs {
  const float *d_short, *d_hs, *x, *timelast;
  const int32_t *seq_len;
  const float *wh_g, *wh_c, *tvec, *save;
  int B, L, ldx;
  float *d_xproj, *rh, *d_xt, *d_tvec_partial;
};

// Backward through time with the same cut: output k of both transposed products (d(r*h) = dcpre . Wc_h^T over
// 128 n, dh_prev += dgpre . Wg_h^T over 256 n) belongs to ONE lane (kp < 2 of octet o of wave w: k = 16 w + 2 o +
// kp), which also carries dh, the element-wise chain and the time-gate parameter gradients of that column in
// registers from step to step; the contraction index n is split over the octet's 8 lanes.
__global__ __launch_bounds__(512) void tagru_bwd_kernel(BwdArgs p) {
  __shared__ __attribute__((aligned(16))) float dc_s[D + 8];
  __shared__ __attribute__((aligned(16))) float dg_s[2 * D + 16];

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kp = lane & 7, o = lane >> 3;
  const int steps = min(max(p.seq_len[b] - 1, 0), p.L);
  const size_t row0 = (size_t)b * p.L;
  const bool own = kp < 2;
  const int k0 = 16 * w + 2 * o;              // the octet's two output rows k0, k0 + 1 of both kernels
  const int kcol = k0 + (kp & 1);

  // transposed weights -> registers: wcT[c][j] = Wc_h[k0 + c][16 kp + (2 j, 2 j + 1)],
  //                                   wgT[c][j] = Wg_h[k0 + c][32 kp + (2 j, 2 j + 1)]
  f32x2 wcT[2][8], wgT[2][16];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(&p.wh_c[(size_t)(k0 + c) * D + 16 * kp + 4 * q]);
      wcT[c][2 * q] = f32x2{v.x, v.y}; wcT[c][2 * q + 1] = f32x2{v.z, v.w};
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(&p.wh_g[(size_t)(k0 + c) * (2 * D) + 32 * kp + 4 * q]);
      wgT[c][2 * q] = f32x2{v.x, v.y}; wgT[c][2 * q + 1] = f32x2{v.z, v.w};
    }
  }
  const bool seqrec = p.ldx == 5 * D;         // T-SeqRec cell: both time gates come hoisted in xproj
  const bool plain = p.tvec == nullptr;       // plain GRUCell (T = 1) or T-SeqRec: no in-loop time-gate parameters
  const int ldx = p.ldx, nsave = seqrec ? 6 : 5;
  float tv[NTV], gtv[NTV];
#pragma unroll
  for (int i = 0; i < NTV; ++i) {
    tv[i] = (own && !plain) ? p.tvec[i * D + kcol] : 0.f;
    gtv[i] = 0.f;
  }

  // zero-fill the dead steps of the outputs
  for (int t = steps; t < p.L; ++t) {
    const size_t r = row0 + t;
    for (int c = tid; c < ldx; c += 512) p.d_xproj[r * ldx + c] = 0.f;
    if (tid < D) {
      p.rh[r * D + tid] = 0.f;
      p.d_xt[r * D + tid] = 0.f;
    }
  }

  float dh = (own && steps > 0) ? p.d_short[(size_t)b * D + kcol] : 0.f;

  float n_r = 0.f, n_u = 0.f, n_c = 0.f, n_T = 0.f, n_hp = 0.f, n_x = 0.f, n_dl = 0.f, n_dhs = 0.f, n_N = 1.f;
  auto prefetch = [&](int t) {      // every lane loads (see the forward kernel); only owners use the values
    const size_t r = row0 + t;
    const float *sv = p.save + r * (nsave * D) + kcol;
    n_r = sv[0]; n_u = sv[D]; n_c = sv[2 * D]; n_T = sv[3 * D]; n_hp = sv[4 * D];
    if (seqrec) n_N = sv[5 * D];
    n_x = p.x[r * D + kcol];
    n_dl = p.timelast[r];
    if (p.d_hs) n_dhs = p.d_hs[r * D + kcol];       // gradient on the step's OUTPUT (decoder keys = GRU outputs)
  };
  if (steps > 0) prefetch(steps - 1);
  const int vpos = 16 * kp + 4 * (kp >> 2);          // 16 values of the padded 128-vector dc_s
  const int gpos = 32 * kp + 4 * (kp >> 1);          // 32 values of the padded 256-vector dg_s
  GRU_STAMP_DECL
  GRU_STAMP_START

  for (int t = steps - 1; t >= 0; --t) {
    const float r_ = n_r, u = n_u, c = n_c, T = n_T, hp = n_hp, xt = n_x, dl = n_dl, N = n_N;
    dh += n_dhs;
    if (t > 0) prefetch(t - 1);
    const size_t row = row0 + t;
    // element-wise chain of column kcol (computed by every lane, stored by the owners)
    const float du = dh * (hp * N - c * T);
    const float dc = dh * (1.f - u) * T;
    const float dT = dh * (1.f - u) * c;
    float dhp = dh * u * N;
    const float dcpre = dc * (1.f - c * c);
    // time gate T = sigmoid(kw2*tw + w12*ts + b12), tw = relu(x*kw1 + kb1 + h*hw1), ts = relu(w1*dl + b1)
    const float twp = xt * tv[KW1] + tv[KB1] + hp * tv[HW1];
    const float tsp = tv[W1] * dl + tv[B1];
    const float tw = fmaxf(twp, 0.f), ts = fmaxf(tsp, 0.f);
    const float dTp = plain ? 0.f : dT * T * (1.f - T);
    gtv[KW2] += dTp * tw;
    gtv[W12] += dTp * ts;
    gtv[B12] += dTp;
    const float dtw = (twp > 0.f) ? dTp * tv[KW2] : 0.f;
    const float dts = (tsp > 0.f) ? dTp * tv[W12] : 0.f;
    gtv[KW1] += dtw * xt;
    gtv[KB1] += dtw;
    gtv[HW1] += dtw * hp;
    gtv[W1] += dts * dl;
    gtv[B1] += dts;
    dhp += dtw * tv[HW1];
    const float d_now = dh * u * hp * N * (1.f - N), d_last = dT * T * (1.f - T);   // T-SeqRec's hoisted gates
    if (own) dc_s[padpos(kcol)] = dcpre;
    GRU_STAMP(0)        // element-wise chain
    __syncthreads();
    GRU_STAMP(1)        // barrier 1
    // phase A: d(r*h)[k] = sum_n dcpre[n] Wc_h[k][n]
    float drh;
    {
      float4 dq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) dq[q] = *reinterpret_cast<const float4 *>(&dc_s[vpos + 4 * q]);
      f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 lo = {dq[q].x, dq[q].y}, hi = {dq[q].z, dq[q].w};
        a0 = pk_fma(lo, wcT[0][2 * q], a0); a1 = pk_fma(lo, wcT[1][2 * q], a1);
        a0 = pk_fma(hi, wcT[0][2 * q + 1], a0); a1 = pk_fma(hi, wcT[1][2 * q + 1], a1);
      }
      drh = octet_reduce2(a0.x + a0.y, a1.x + a1.y, lane);
    }
    GRU_STAMP(2)        // phase A: reads, FMAs, reduce
    const float dr = drh * hp;
    dhp += drh * r_;
    const float dgr = dr * r_ * (1.f - r_);
    const float dgu = du * u * (1.f - u);
    if (own) {
      dg_s[padpos(kcol)] = dgr;
      dg_s[padpos(D + kcol)] = dgu;
    }
    GRU_STAMP(3)        // gate gradients, LDS writes
    __syncthreads();
    GRU_STAMP(4)        // barrier 2
    // phase B: dh_prev[k] += sum_n dgpre[n] Wg_h[k][n]   (n over r and u gates: 256)
    {
      float4 dq[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) dq[q] = *reinterpret_cast<const float4 *>(&dg_s[gpos + 4 * q]);
      f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x2 lo = {dq[q].x, dq[q].y}, hi = {dq[q].z, dq[q].w};
        a0 = pk_fma(lo, wgT[0][2 * q], a0); a1 = pk_fma(lo, wgT[1][2 * q], a1);
        a0 = pk_fma(hi, wgT[0][2 * q + 1], a0); a1 = pk_fma(hi, wgT[1][2 * q + 1], a1);
      }
      const float s = octet_reduce2(a0.x + a0.y, a1.x + a1.y, lane);
      dh = dhp + s;
    }
    GRU_STAMP(5)        // phase B: reads, FMAs, reduce
    // the step's stores come last, after the prefetched values of step t - 1 have landed (see the forward kernel)
    asm volatile("" : "+v"(n_r), "+v"(n_u), "+v"(n_c), "+v"(n_T), "+v"(n_hp), "+v"(n_x), "+v"(n_dl), "+v"(n_dhs),
                 "+v"(n_N));
    if (own) {
      float *dx = p.d_xproj + row * ldx + kcol;
      dx[0] = dgr; dx[D] = dgu; dx[2 * D] = dcpre;
      if (seqrec) {
        dx[3 * D] = d_now;
        dx[4 * D] = d_last;
      }
      p.rh[row * D + kcol] = r_ * hp;
      p.d_xt[row * D + kcol] = dtw * tv[KW1];
    }
    GRU_STAMP(6)        // prefetch landed, stores issued
  }
  GRU_STAMP_DUMP(1)

  if (own) {
#pragma unroll
    for (int i = 0; i < NTV; ++i) p.d_tvec_partial[((size_t)b * NTV + i) * D + kcol] = gtv[i];
  }
}

}  // namespace

extern "C" int mtam_tagru_fwd(const float *xproj, const float *x, const float *timelast,
                              const int32_t *seq_len, const float *wh_g, const float *wh_c,
                              const float *tvec, int B, int L, float *hs, float *short_out,
                              float *save, void *stream) {
  MTAM_CHECK_ARG(B > 0 && L > 0, "tagru_fwd: B and L must be positive");
  MTAM_CHECK_ARG(xproj && x && timelast && seq_len && wh_g && wh_c && hs && short_out,
                 "tagru_fwd: null argument");        // tvec may be NULL: plain GRUCell
  MTAM_CHECK_ARG(mtam_aligned16(wh_g) && mtam_aligned16(wh_c), "tagru_fwd: weights must be 16-byte aligned");
  FwdArgs a{xproj, x, timelast, seq_len, wh_g, wh_c, tvec, B, L, 3 * D, hs, short_out, save};
  hipLaunchKernelGGL(tagru_fwd_kernel, dim3(B), dim3(512), 0, static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("tagru_fwd");
  return MTAM_OK;
}

extern "C" int mtam_tagru_bwd(const float *d_short, const float *d_hs, const float *x, const float *timelast,
                              const int32_t *seq_len, const float *wh_g, const float *wh_c,
                              const float *tvec, const float *save, int B, int L, float *d_xproj,
                              float *rh, float *d_xt, float *d_tvec_partial, void *stream) {
  MTAM_CHECK_ARG(B > 0 && L > 0, "tagru_bwd: B and L must be positive");
  MTAM_CHECK_ARG(d_short && x && timelast && seq_len && wh_g && wh_c && save && d_xproj && rh && d_xt &&
                     d_tvec_partial,
                 "tagru_bwd: null argument");        // tvec (plain GRUCell) and d_hs may be NULL
  MTAM_CHECK_ARG(mtam_aligned16(wh_g) && mtam_aligned16(wh_c), "tagru_bwd: weights must be 16-byte aligned");
  BwdArgs a{d_short, d_hs, x, timelast, seq_len, wh_g, wh_c, tvec, save, B, L, 3 * D, d_xproj, rh, d_xt,
            d_tvec_partial};
  hipLaunchKernelGGL(tagru_bwd_kernel, dim3(B), dim3(512), 0, static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("tagru_bwd");
  return MTAM_OK;
}

// ---- T-SeqRec cell (TimeAwareGRUCell_sigmoid): the recurrence above with the two gates read from
// columns 3 D .. 5 D of xproj [B*L, 5 D]; save [B*L, 6 D]; d_xproj [B*L, 5 D].
extern "C" int mtam_tagru_seqrec_fwd(const float *xproj5, const int32_t *seq_len, const float *wh_g,
                                     const float *wh_c, int B, int L, float *hs, float *short_out, float *save6,
                                     void *stream) {
  MTAM_CHECK_ARG(B > 0 && L > 0, "tagru_seqrec_fwd: B and L must be positive");
  MTAM_CHECK_ARG(xproj5 && seq_len && wh_g && wh_c && hs && short_out, "tagru_seqrec_fwd: null argument");
  MTAM_CHECK_ARG(mtam_aligned16(wh_g) && mtam_aligned16(wh_c), "tagru_seqrec_fwd: weights must be 16-byte aligned");
  FwdArgs a{xproj5, xproj5, xproj5, seq_len, wh_g, wh_c, nullptr, B, L, 5 * D, hs, short_out, save6};
  hipLaunchKernelGGL(tagru_fwd_kernel, dim3(B), dim3(512), 0, static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("tagru_seqrec_fwd");
  return MTAM_OK;
}

extern "C" int mtam_tagru_seqrec_bwd(const float *d_short, const float *d_hs, const int32_t *seq_len,
                                     const float *wh_g, const float *wh_c, const float *save6, int B, int L,
                                     float *d_xproj5, float *rh, float *d_xt, float *d_tvec_partial, void *stream) {
  MTAM_CHECK_ARG(B > 0 && L > 0, "tagru_seqrec_bwd: B and L must be positive");
  MTAM_CHECK_ARG(d_short && seq_len && wh_g && wh_c && save6 && d_xproj5 && rh && d_xt && d_tvec_partial,
                 "tagru_seqrec_bwd: null argument");
  MTAM_CHECK_ARG(mtam_aligned16(wh_g) && mtam_aligned16(wh_c), "tagru_seqrec_bwd: weights must be 16-byte aligned");
  // x / timelast are only read (their values unused) in this mode: any readable [B*L, D] / [B*L] buffer does
  BwdArgs a{d_short, d_hs, save6, save6, seq_len, wh_g, wh_c, nullptr, save6, B, L, 5 * D, d_xproj5, rh, d_xt,
            d_tvec_partial};
  hipLaunchKernelGGL(tagru_bwd_kernel, dim3(B), dim3(512), 0, static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("tagru_seqrec_bwd");
  return MTAM_OK;
}

// ---- the cell's time inputs: tin[r, 0:D] = tanh(timenow[r] w1 + b1), tin[r, D:2D] = tanh(timelast[r] w2 + b2)
// (tvec4 rows: w1, b1, w2, b2), and their gradient products whose column sums are d(w1, b1, w2, b2):
// out[r] = (g_now timenow[r] | g_now | g_last timelast[r] | g_last), g = d_tin (1 - tin^2).
namespace {
__global__ __launch_bounds__(256) void tsr_time_fwd_kernel(const float *__restrict__ timenow,
                                                           const float *__restrict__ timelast,
                                                           const float *__restrict__ tvec4, int R,
                                                           float *__restrict__ tin) {
  const int c = threadIdx.x;                 // 0 .. 2 D - 1
  const bool last = c >= D;
  const float wv = tvec4[(last ? 2 : 0) * D + (c & (D - 1))], bv = tvec4[(last ? 3 : 1) * D + (c & (D - 1))];
  for (int r = blockIdx.x; r < R; r += gridDim.x) {
    const float t = last ? timelast[r] : timenow[r];
    tin[(size_t)r * (2 * D) + c] = tanhf(t * wv + bv);
  }
}
__global__ __launch_bounds__(256) void tsr_time_bwd_kernel(const float *__restrict__ d_tin,
                                                           const float *__restrict__ tin,
                                                           const float *__restrict__ timenow,
                                                           const float *__restrict__ timelast, int R,
                                                           float *__restrict__ out) {
  const int c = threadIdx.x;
  const bool last = c >= D;
  for (int r = blockIdx.x; r < R; r += gridDim.x) {
    const float a = tin[(size_t)r * (2 * D) + c];
    const float g = d_tin[(size_t)r * (2 * D) + c] * (1.f - a * a);
    const float t = last ? timelast[r] : timenow[r];
    float *o = out + (size_t)r * (4 * D) + (last ? 2 * D : 0) + (c & (D - 1));
    o[0] = g * t;
    o[D] = g;
  }
}
}  // namespace

extern "C" int mtam_tsr_time_inputs_fwd(const float *timenow, const float *timelast, const float *tvec4, int R,
                                        float *tin, void *stream) {
  MTAM_CHECK_ARG(timenow && timelast && tvec4 && tin && R > 0, "tsr_time_inputs_fwd: bad arguments");
  hipLaunchKernelGGL(tsr_time_fwd_kernel, dim3(min(R, 2048)), dim3(2 * D), 0, static_cast<hipStream_t>(stream),
                     timenow, timelast, tvec4, R, tin);
  MTAM_CHECK_LAUNCH("tsr_time_inputs_fwd");
  return MTAM_OK;
}

extern "C" int mtam_tsr_time_inputs_bwd(const float *d_tin, const float *tin, const float *timenow,
                                        const float *timelast, int R, float *out, void *stream) {
  MTAM_CHECK_ARG(d_tin && tin && timenow && timelast && out && R > 0, "tsr_time_inputs_bwd: bad arguments");
  hipLaunchKernelGGL(tsr_time_bwd_kernel, dim3(min(R, 2048)), dim3(2 * D), 0, static_cast<hipStream_t>(stream),
                     d_tin, tin, timenow, timelast, R, out);
  MTAM_CHECK_LAUNCH("tsr_time_inputs_bwd");
  return MTAM_OK;
}
