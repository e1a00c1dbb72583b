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
// the whole sequence: wave w owns k in [16w, 16w+16) and every output column
// (4 gate + 2 candidate columns per lane, 96 VGPRs).  Per step a wave reads its
// 16 h values from LDS as 4 broadcast ds_read_b128, multiplies, and the 8
// k-slices are summed through LDS.  No weight byte is re-read per step.
#include "common.h"

namespace {

constexpr int D = MTAM_D;
constexpr int NW = 8;  // waves per workgroup

// tvec rows
enum { KW1 = 0, KB1, HW1, W1, B1, KW2, W12, B12, NTV };

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
  __shared__ __attribute__((aligned(16))) float h_s[D];
  __shared__ __attribute__((aligned(16))) float rh_s[D];
  __shared__ float u_s[D];
  __shared__ float T_s[D];
  __shared__ float N_s[D];
  __shared__ float pg[NW][2 * D];
  __shared__ float pc[NW][D];

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int steps = min(max(p.seq_len[b] - 1, 0), p.L);
  const size_t row0 = (size_t)b * p.L;

  // recurrent weights -> registers (coalesced: lanes run along the output column), packed in pairs
  // of columns so that the inner products issue as v_pk_fma_f32
  f32x2 wg[16][2], wc[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const float *rg = p.wh_g + (size_t)(16 * w + kk) * (2 * D) + lane;
    const float *rc = p.wh_c + (size_t)(16 * w + kk) * D + lane;
    wg[kk][0] = f32x2{rg[0], rg[64]};
    wg[kk][1] = f32x2{rg[128], rg[192]};
    wc[kk] = f32x2{rc[0], rc[64]};
  }
  // Roles of the finalize phases (one column per thread):
  //   threads   0..127  r gate (critical path), then the candidate / new state
  //   threads 128..255  u gate
  //   threads 256..383  time gate T (needs only x_t, dt and the OLD state: off the critical path)
  const int col = tid & (D - 1);
  // tvec == nullptr: the plain tf GRUCell (Model/Modules/gru.py:13-39) -- no time gate, T = 1
  const bool seqrec = p.ldx == 5 * D;
  const bool plain = p.tvec == nullptr && !seqrec;
  const bool is_T = !plain && (tid >= 2 * D) && (tid < 3 * D);
  const int ldx = p.ldx, nsave = seqrec ? 6 : 5;
  float tv[NTV];
#pragma unroll
  for (int i = 0; i < NTV; ++i) tv[i] = (is_T && !seqrec) ? p.tvec[i * D + col] : 0.f;

  if (tid < D) {
    h_s[tid] = 0.f;
    T_s[tid] = 1.f;
    N_s[tid] = 1.f;
  }

  // software prefetch of step t's inputs (independent of the recurrence)
  float n_a = 0.f, n_b = 0.f;
  auto prefetch = [&](int t) {
    const size_t r = row0 + t;
    if (tid < 2 * D) n_a = p.xproj[r * ldx + tid];             // gate pre-activation, input half
    if (tid < D) n_b = p.xproj[r * ldx + 2 * D + tid];         // candidate pre-activation, input half
    if (is_T) {
      n_a = seqrec ? p.xproj[r * ldx + 3 * D + col] : p.x[r * D + col];
      n_b = seqrec ? p.xproj[r * ldx + 4 * D + col] : p.timelast[r];
    }
  };
  if (steps > 0) prefetch(0);
  __syncthreads();

  for (int t = 0; t < steps; ++t) {
    const float in_a = n_a, in_b = n_b;
    if (t + 1 < steps) prefetch(t + 1);

    // phase 1: gate pre-activations, recurrent half (all waves: k slice 16w..16w+15, 4 columns per lane)
    {
      float hv[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(&h_s[16 * w + 4 * q]);
        hv[4 * q] = v.x; hv[4 * q + 1] = v.y; hv[4 * q + 2] = v.z; hv[4 * q + 3] = v.w;
      }
      f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const f32x2 h2 = {hv[kk], hv[kk]};
        a0 = pk_fma(h2, wg[kk][0], a0);
        a1 = pk_fma(h2, wg[kk][1], a1);
      }
      pg[w][lane] = a0.x; pg[w][lane + 64] = a0.y; pg[w][lane + 128] = a1.x; pg[w][lane + 192] = a1.y;
    }
    __syncthreads();
    float r_keep = 0.f;
    if (tid < 2 * D) {
      float g = in_a;
#pragma unroll
      for (int q = 0; q < NW; ++q) g += pg[q][tid];
      const float s = fast_sigmoid(g);
      if (tid < D) {
        r_keep = s;
        rh_s[tid] = s * h_s[tid];
      } else {
        u_s[tid - D] = s;
      }
    } else if (is_T) {
      if (seqrec) {
        N_s[col] = fast_sigmoid(in_a);
        T_s[col] = fast_sigmoid(in_b);
      } else {
        const float h = h_s[col];
        const float tw = fmaxf(in_a * tv[KW1] + tv[KB1] + h * tv[HW1], 0.f);
        const float ts = fmaxf(tv[W1] * in_b + tv[B1], 0.f);
        T_s[col] = fast_sigmoid(tv[KW2] * tw + tv[W12] * ts + tv[B12]);
      }
    }
    __syncthreads();
    // phase 2: candidate pre-activation, recurrent half on r*h (2 columns per lane)
    {
      float rv[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(&rh_s[16 * w + 4 * q]);
        rv[4 * q] = v.x; rv[4 * q + 1] = v.y; rv[4 * q + 2] = v.z; rv[4 * q + 3] = v.w;
      }
      f32x2 a = {0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) a = pk_fma(f32x2{rv[kk], rv[kk]}, wc[kk], a);
      pc[w][lane] = a.x; pc[w][lane + 64] = a.y;
    }
    __syncthreads();
    if (tid < D) {
      float cp = in_b;
#pragma unroll
      for (int q = 0; q < NW; ++q) cp += pc[q][tid];
      const float c = fast_tanh(cp);
      const float h = h_s[tid], u = u_s[tid], T = T_s[tid], N = N_s[tid];
      const float hn = u * h * N + (1.f - u) * c * T;
      h_s[tid] = hn;
      const size_t r = row0 + t;
      p.hs[r * D + tid] = hn;
      if (p.save) {
        float *sv = p.save + r * (nsave * D) + tid;
        sv[0] = r_keep; sv[D] = u; sv[2 * D] = c; sv[3 * D] = T; sv[4 * D] = h;
        if (seqrec) sv[5 * D] = N;
      }
    }
    __syncthreads();
  }

  if (tid < D) {
    p.short_out[(size_t)b * D + tid] = (steps > 0) ? h_s[tid] : 0.f;
    for (int t = steps; t < p.L; ++t) p.hs[(row0 + t) * D + tid] = 0.f;   // dynamic_rnn zero-fills dead steps
  }
}

struct BwdArgs {
  const float *d_short, *d_hs, *x, *timelast;
  const int32_t *seq_len;
  const float *wh_g, *wh_c, *tvec, *save;
  int B, L, ldx;
  float *d_xproj, *rh, *d_xt, *d_tvec_partial;
};

__global__ __launch_bounds__(512) void tagru_bwd_kernel(BwdArgs p) {
  __shared__ __attribute__((aligned(16))) float dc_s[D];
  __shared__ __attribute__((aligned(16))) float dg_s[2 * D];
  __shared__ float pA[NW][D];
  __shared__ float pB[NW][D];

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int steps = min(max(p.seq_len[b] - 1, 0), p.L);
  const size_t row0 = (size_t)b * p.L;

  // transposed recurrent weights -> registers: lane owns outputs k = lane, lane + 64;
  // wave w owns n in [16w,16w+16) of the candidate kernel and [32w,32w+32) of the gate kernel.
  // (pairs {k = lane, k = lane + 64} packed for v_pk_fma_f32)
  f32x2 wcT[16], wgT[32];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v0 = *reinterpret_cast<const float4 *>(&p.wh_c[(size_t)lane * D + 16 * w + 4 * q]);
    const float4 v1 = *reinterpret_cast<const float4 *>(&p.wh_c[(size_t)(lane + 64) * D + 16 * w + 4 * q]);
    wcT[4 * q] = f32x2{v0.x, v1.x}; wcT[4 * q + 1] = f32x2{v0.y, v1.y};
    wcT[4 * q + 2] = f32x2{v0.z, v1.z}; wcT[4 * q + 3] = f32x2{v0.w, v1.w};
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 v0 = *reinterpret_cast<const float4 *>(&p.wh_g[(size_t)lane * (2 * D) + 32 * w + 4 * q]);
    const float4 v1 = *reinterpret_cast<const float4 *>(&p.wh_g[(size_t)(lane + 64) * (2 * D) + 32 * w + 4 * q]);
    wgT[4 * q] = f32x2{v0.x, v1.x}; wgT[4 * q + 1] = f32x2{v0.y, v1.y};
    wgT[4 * q + 2] = f32x2{v0.z, v1.z}; wgT[4 * q + 3] = f32x2{v0.w, v1.w};
  }
  const bool seqrec = p.ldx == 5 * D;         // T-SeqRec cell: both time gates come hoisted in xproj
  const bool plain = p.tvec == nullptr;       // plain GRUCell (T = 1) or T-SeqRec: no in-loop time-gate parameters
  const int ldx = p.ldx, nsave = seqrec ? 6 : 5;
  float tv[NTV], gtv[NTV];
#pragma unroll
  for (int i = 0; i < NTV; ++i) {
    tv[i] = (tid < D && !plain) ? p.tvec[i * D + tid] : 0.f;
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

  float dh = (tid < D && steps > 0) ? p.d_short[(size_t)b * D + tid] : 0.f;

  float n_r = 0.f, n_u = 0.f, n_c = 0.f, n_T = 0.f, n_hp = 0.f, n_x = 0.f, n_dl = 0.f, n_dhs = 0.f, n_N = 1.f;
  auto prefetch = [&](int t) {
    if (tid < D) {
      const size_t r = row0 + t;
      const float *sv = p.save + r * (nsave * D) + tid;
      n_r = sv[0]; n_u = sv[D]; n_c = sv[2 * D]; n_T = sv[3 * D]; n_hp = sv[4 * D];
      if (seqrec) n_N = sv[5 * D];
      n_x = p.x[r * D + tid];
      n_dl = p.timelast[r];
      if (p.d_hs) n_dhs = p.d_hs[r * D + tid];      // gradient on the step's OUTPUT (decoder keys = GRU outputs)
    }
  };
  if (steps > 0) prefetch(steps - 1);

  for (int t = steps - 1; t >= 0; --t) {
    const float r_ = n_r, u = n_u, c = n_c, T = n_T, hp = n_hp, xt = n_x, dl = n_dl, N = n_N;
    dh += n_dhs;
    if (t > 0) prefetch(t - 1);
    const size_t row = row0 + t;
    float du = 0.f, dhp = 0.f, dcpre = 0.f;
    if (tid < D) {
      du = dh * (hp * N - c * T);
      const float dc = dh * (1.f - u) * T;
      const float dT = dh * (1.f - u) * c;
      dhp = dh * u * N;
      if (seqrec) {       // gradients of the two hoisted gate pre-activations
        float *dx = p.d_xproj + row * ldx + tid;
        dx[3 * D] = dh * u * hp * N * (1.f - N);
        dx[4 * D] = dT * T * (1.f - T);
      }
      dcpre = dc * (1.f - c * c);
      dc_s[tid] = dcpre;
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
      p.d_xt[row * D + tid] = dtw * tv[KW1];
      dhp += dtw * tv[HW1];
    }
    __syncthreads();
    // phase A: d(r*h) = dcpre . Wc_h^T
    {
      float dv[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(&dc_s[16 * w + 4 * q]);
        dv[4 * q] = v.x; dv[4 * q + 1] = v.y; dv[4 * q + 2] = v.z; dv[4 * q + 3] = v.w;
      }
      f32x2 acc = {0.f, 0.f};
#pragma unroll
      for (int nn = 0; nn < 16; ++nn) acc = pk_fma(f32x2{dv[nn], dv[nn]}, wcT[nn], acc);
      pA[w][lane] = acc.x; pA[w][lane + 64] = acc.y;
    }
    __syncthreads();
    if (tid < D) {
      float drh = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) drh += pA[q][tid];
      const float dr = drh * hp;
      dhp += drh * r_;
      const float dgr = dr * r_ * (1.f - r_);
      const float dgu = du * u * (1.f - u);
      dg_s[tid] = dgr;
      dg_s[D + tid] = dgu;
      float *dx = p.d_xproj + row * ldx + tid;
      dx[0] = dgr; dx[D] = dgu; dx[2 * D] = dcpre;
      p.rh[row * D + tid] = r_ * hp;
    }
    __syncthreads();
    // phase B: dh_prev += dgpre . Wg_h^T
    {
      float dv[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(&dg_s[32 * w + 4 * q]);
        dv[4 * q] = v.x; dv[4 * q + 1] = v.y; dv[4 * q + 2] = v.z; dv[4 * q + 3] = v.w;
      }
      f32x2 acc = {0.f, 0.f};
#pragma unroll
      for (int nn = 0; nn < 32; ++nn) acc = pk_fma(f32x2{dv[nn], dv[nn]}, wgT[nn], acc);
      pB[w][lane] = acc.x; pB[w][lane + 64] = acc.y;
    }
    __syncthreads();
    if (tid < D) {
#pragma unroll
      for (int q = 0; q < NW; ++q) dhp += pB[q][tid];
      dh = dhp;
    }
  }

  if (tid < D) {
#pragma unroll
    for (int i = 0; i < NTV; ++i) p.d_tvec_partial[((size_t)b * NTV + i) * D + tid] = gtv[i];
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
