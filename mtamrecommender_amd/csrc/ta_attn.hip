// Time-aware multi-head attention, decoder form (T_q = 1), one block of
// vanilla_attention: Model/Modules/time_aware_attention.py:215-456 as wired by
// Model/MTAMRec_model.py:83-90, including residual and normalize() (:7-34).
//
// With one query per sample, Q.K^T is a 1xL row per head: there
// is no dense contraction to put on MFMA, the work is streaming the sample's
// L key / value / raw-key rows once (3 x L x 512 B) and a few wave reductions.
// One 256-thread workgroup per sample:
//   q -> [Q | q.Wt] by an in-kernel mat-vec over the L2-resident [D,2D] weight,
//   half a wave per key row (32 lanes x 16 B) for the two dot products,
//   wavefront-reduced masked softmax per head, weighted V sum, residual, LN.
#include "common.h"

// tools/attn_lab.hip (-DMTAM_ATTN_STAMPS): s_memrealtime (100 MHz) at the phase boundaries of the two decoder kernels,
// thread 0 of the middle workgroup; the product build has none
#ifdef MTAM_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[2][16];
#define AT_STAMP(k, i)                                                                             \
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {                                           \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    g_attn_stamps[k][i] = __builtin_amdgcn_s_memrealtime();                                        \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  }
#else
#define AT_STAMP(k, i)
#endif

namespace {

constexpr int D = MTAM_D;
constexpr int MAXL = 256;
constexpr int MAXH = 8;
// -2**32 + 1 (time_aware_attention.py:392); rounds to -2^32 in float32 exactly as in TF
constexpr float MASK_VALUE = -4294967295.0f;

__device__ __forceinline__ int save_floats(int L, int H) { return 3 * D + 3 * L + 2 * H * L + 1; }

struct FwdArgs {
  const float *dec_in, *x, *kv;
  int ld_kv, k_off, v_off;
  const float *t_query, *t_keys;
  const int32_t *seq_len;
  const float *wqt, *bq, *tparams, *ln_beta, *ln_gamma;
  int B, L, H;
  float *dec_out, *save;
  // optional (last decoder block): the model's head layer_norm (tf.contrib.layers.layer_norm, eps 1e-12,
  // Model/MTAMRec_model.py:91) applied to this block's output in the same launch
  const float *head_beta, *head_gamma;
  float *pred_out, *head_save;
};

__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

__global__ __launch_bounds__(256) void ta_attn_decode_fwd_kernel(FwdArgs p) {
  __shared__ __attribute__((aligned(16))) float q_s[D];
  __shared__ __attribute__((aligned(16))) float Q_s[D];
  __shared__ __attribute__((aligned(16))) float qt_s[D];
  __shared__ float sc_s[MAXH][MAXL];
  __shared__ float qk_s[MAXH][MAXL];
  __shared__ float a_s[MAXL], dk_s[MAXL], sg_s[MAXL];
  __shared__ __attribute__((aligned(16))) float o_part[8][D];
  __shared__ float red[4];

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int L = p.L, H = p.H;
  const int sl = min(max(p.seq_len[b], 0), L);
  const unsigned inv_L = (1u << 20) / (unsigned)L + 1u;      // i / L for i < H L <= 2,048 (i L < 2^20): exact
  const size_t row0 = (size_t)b * L;
  // H is 1, 2, 4 or 8 (checked by the host): shifts and masks, not run-time divisions (~25 instructions each)
  const int log_h = (H >= 2) + (H >= 4) + (H >= 8), lph_shift = 5 - log_h, lanes_per_head = 1 << lph_shift;
  const float inv_div = sqrtf((float)(D >> log_h));

  // Every global load that does not depend on the projected query is issued HERE, before the first
  // barrier: the sample's key / raw-key / value rows (half a wave per row, rows hw, hw+8, ... of the
  // sample), the key times and time-gate parameters, and the LN parameters.  The kernel is a chain of
  // short phases on one CU; written phase by phase it paid one global round trip per phase (~9 of them).
  AT_STAMP(0, 0)
  const int hw = tid >> 5, li = tid & 31;
  constexpr int KB = 8;                      // keys per half wave and trip: 64 keys per trip of the workgroup
  const float q_in = (tid < D) ? p.dec_in[(size_t)b * D + tid] : 0.f;
  const float ln_g = (tid < D) ? p.ln_gamma[tid] : 0.f, ln_b = (tid < D) ? p.ln_beta[tid] : 0.f;
  const float tq = p.t_query[b];
  float4 kq[KB], xq[KB], vq[KB];
  // the time-gate inputs of ONE key per lane: lane li of a half wave carries key li & 7 of the half wave's eight through
  // the gate chain (tanh, log, tanh, sigmoid) -- with all 32 lanes repeating every key's chain that phase was 2.3 us
  float tk_l, tp_l[5];
  auto load_keys = [&](int jb) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const int jc = min(jb + 8 * i, L - 1);      // clamped: padded keys are valid memory and are masked below
      kq[i] = *reinterpret_cast<const float4 *>(&p.kv[(row0 + jc) * p.ld_kv + p.k_off + 4 * li]);
      vq[i] = *reinterpret_cast<const float4 *>(&p.kv[(row0 + jc) * p.ld_kv + p.v_off + 4 * li]);
      xq[i] = *reinterpret_cast<const float4 *>(&p.x[(row0 + jc) * D + 4 * li]);
    }
    const int jl = min(jb + 8 * (li & 7), L - 1);
    tk_l = p.t_keys[row0 + jl];
#pragma unroll
    for (int q = 0; q < 5; ++q) tp_l[q] = p.tparams[q * L + jl];
  };
  load_keys(hw);
  AT_STAMP(0, 1)       // argument-only loads issued
  if (tid < D) q_s[tid] = q_in;
  __syncthreads();
  AT_STAMP(0, 2)       // q in LDS

  // [Q | qt] = q . [Wq | Wt]; thread = output column, coalesced weight reads, 16 loads in flight
  {
    float acc = 0.f;
    for (int k0 = 0; k0 < D; k0 += 16) {
      float wv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) wv[i] = p.wqt[(size_t)(k0 + i) * (2 * D) + tid];
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        const float4 qv = *reinterpret_cast<const float4 *>(&q_s[k0 + i]);
        acc = fmaf(qv.x, wv[i], acc);
        acc = fmaf(qv.y, wv[i + 1], acc);
        acc = fmaf(qv.z, wv[i + 2], acc);
        acc = fmaf(qv.w, wv[i + 3], acc);
      }
    }
    if (tid < D) Q_s[tid] = fmaxf(acc + p.bq[tid], 0.f);
    else qt_s[tid - D] = acc;
  }
  __syncthreads();
  AT_STAMP(0, 3)       // [Q | qt] projected

  // scores, masked softmax and the weighted value sum, 64 keys per trip (one trip for L <= 64).  The
  // value rows stay in registers from the load above until the softmax weights exist; for L > 64 the
  // softmax needs every score first, so later trips reload (scores in one pass, values in a second).
  const float4 Q4 = *reinterpret_cast<const float4 *>(&Q_s[4 * li]);
  const float4 T4 = *reinterpret_cast<const float4 *>(&qt_s[4 * li]);
  auto score_keys = [&](int jb) {
    // (1) both dot products of the half wave's eight keys, summed over the lanes (every lane ends with every sum)
    float dKv[KB], dAv[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      dKv[i] = group_sum_fast(dot4(kq[i], Q4), lanes_per_head);
      dAv[i] = group_sum_dpp<32>(dot4(xq[i], T4));
    }
    // (2) the gate chain, one key per lane
    {
      const int il = li & 7, jl = jb + 8 * il;
      float dA = dAv[0];
#pragma unroll
      for (int i = 1; i < KB; ++i) dA = (il == i) ? dAv[i] : dA;
      const float a = fast_tanh(dA);
      const float delta = logf(fabsf(tq - tk_l) + 1.0f);
      const float dk = fast_tanh(delta * tp_l[0] + tp_l[1]);
      const float g = tp_l[2] * dk + tp_l[3] * a + tp_l[4];
      const float sg = fast_sigmoid(g);
      if (li < 8 && jl < L) {
        const bool live = jl < sl;
        a_s[jl] = live ? a : 0.f;
        dk_s[jl] = live ? dk : 0.f;
        sg_s[jl] = live ? sg : 0.f;
      }
    }
    // (3) the scores: a head's first lane, with the key's gate back from LDS (written by a lane of this same wave)
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const int j = jb + 8 * i;
      if (j >= L) break;
      if (j < sl) {
        const float sg = sg_s[j];
        if ((li & (lanes_per_head - 1)) == 0) {
          const int h = li >> lph_shift;
          qk_s[h][j] = dKv[i];
          sc_s[h][j] = (dKv[i] * sg) / inv_div;
        }
      } else if (li < H) {
        qk_s[li][j] = 0.f;
        sc_s[li][j] = MASK_VALUE;
      }
    }
  };
  score_keys(hw);
  for (int jb = hw + 8 * KB; jb < L; jb += 8 * KB) {      // L > 64 only
    load_keys(jb);
    score_keys(jb);
  }
  __syncthreads();
  AT_STAMP(0, 4)       // scores

  // masked softmax over keys, one wave per head
  for (int h = w; h < H; h += 4) {
    float m = -INFINITY;
    for (int j = lane; j < L; j += 64) m = fmaxf(m, sc_s[h][j]);
    m = group_max_dpp<64>(m);
    float s = 0.f;
    for (int j = lane; j < L; j += 64) {
      const float e = fast_exp(sc_s[h][j] - m);
      sc_s[h][j] = e;
      s += e;
    }
    s = group_sum_dpp<64>(s);
    for (int j = lane; j < L; j += 64) sc_s[h][j] = sc_s[h][j] / s;
  }
  __syncthreads();
  AT_STAMP(0, 5)       // softmax

  // O = W . V: each half wave sums its own keys (4 channels per lane), the 8 partial rows meet in LDS
  {
    const int head_of_lane = li >> lph_shift;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add_values = [&](int jb) {
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        const int j = jb + 8 * i;
        if (j < sl) {
          const float wj = sc_s[head_of_lane][j];
          o.x = fmaf(wj, vq[i].x, o.x); o.y = fmaf(wj, vq[i].y, o.y);
          o.z = fmaf(wj, vq[i].z, o.z); o.w = fmaf(wj, vq[i].w, o.w);
        }
      }
    };
    if (L > 8 * KB) load_keys(hw);             // the first trip's rows were overwritten by later trips
    add_values(hw);
    for (int jb = hw + 8 * KB; jb < L; jb += 8 * KB) {
      load_keys(jb);
      add_values(jb);
    }
    *reinterpret_cast<float4 *>(&o_part[hw][4 * li]) = o;
  }
  __syncthreads();
  AT_STAMP(0, 6)       // weighted values

  // residual + normalize(eps = 1e-8): (y - mean) / sqrt(var + eps) * gamma + beta
  float y = 0.f;
  if (tid < D) {
    y = ((o_part[0][tid] + o_part[1][tid]) + (o_part[2][tid] + o_part[3][tid])) +
        ((o_part[4][tid] + o_part[5][tid]) + (o_part[6][tid] + o_part[7][tid])) + q_s[tid];
    const float s = group_sum_dpp<64>(y);
    if (lane == 0) red[w] = s;
  }
  __syncthreads();
  float diff = 0.f;
  if (tid < D) {
    const float mean = (red[0] + red[1]) / (float)D;
    diff = y - mean;
    const float s = group_sum_dpp<64>(diff * diff);
    if (lane == 0) red[2 + w] = s;
  }
  __syncthreads();
  float out = 0.f;
  if (tid < D) {
    const float var = (red[2] + red[3]) / (float)D;
    const float sd = sqrtf(var + 1e-8f);
    const float xhat = diff / sd;
    out = ln_g * xhat + ln_b;
    p.dec_out[(size_t)b * D + tid] = out;
    if (p.save) {
      float *sv = p.save + (size_t)b * save_floats(L, H);
      sv[tid] = Q_s[tid];
      sv[D + tid] = qt_s[tid];
      sv[2 * D + tid] = xhat;
      if (tid == 0) sv[3 * D + 3 * L + 2 * H * L] = 1.0f / sd;
    }
  }
  if (p.save) {
    float *sv = p.save + (size_t)b * save_floats(L, H) + 3 * D;
    for (int j = tid; j < L; j += 256) {
      sv[j] = a_s[j];
      sv[L + j] = dk_s[j];
      sv[2 * L + j] = sg_s[j];
    }
    for (int i = tid; i < H * L; i += 256) {
      const int h = (int)(((unsigned)i * inv_L) >> 20), j = i - h * L;
      sv[3 * L + i] = qk_s[h][j];
      sv[3 * L + H * L + i] = sc_s[h][j];
    }
  }
  AT_STAMP(0, 7)       // normalize, saves issued
  if (p.pred_out) {                    // block-uniform: fused head layer_norm
    __syncthreads();                   // red[] is reused
    if (tid < D) {
      const float s1 = group_sum_dpp<64>(out);
      if (lane == 0) red[w] = s1;
    }
    __syncthreads();
    float d2 = 0.f, mean2 = 0.f;
    if (tid < D) {
      mean2 = (red[0] + red[1]) / (float)D;
      d2 = out - mean2;
      const float s2 = group_sum_dpp<64>(d2 * d2);
      if (lane == 0) red[2 + w] = s2;
    }
    __syncthreads();
    if (tid < D) {
      const float var2 = (red[2] + red[3]) / (float)D;
      const float rstd2 = 1.0f / sqrtf(var2 + 1e-12f);
      const float inv = rstd2 * p.head_gamma[tid];
      p.pred_out[(size_t)b * D + tid] = out * inv + (p.head_beta[tid] - mean2 * inv);
      if (p.head_save) {
        p.head_save[(size_t)b * (D + 1) + tid] = d2 * rstd2;
        if (tid == 0) p.head_save[(size_t)b * (D + 1) + D] = rstd2;
      }
    }
  }
  AT_STAMP(0, 8)       // head layer_norm
}

struct BwdArgs {
  const float *d_out, *dec_in, *x, *kv;
  int ld_kv, k_off, v_off;
  const float *t_query, *t_keys;
  const int32_t *seq_len;
  const float *wqt, *tparams, *ln_gamma, *save;
  int B, L, H, accumulate_dx;
  float *d_dec_in, *d_kv, *d_x, *d_qt_pre, *d_tparams_partial, *d_ln_partial;
  // optional (last decoder block): backward of the fused head layer_norm -- d_out is then NOT read;
  // the gradient of the block output is derived from d_pred here
  const float *d_pred, *head_gamma, *head_save;
  float *d_head_partial;      // [B, 2D]: per-sample d beta | d gamma (caller column-sums)
};

__global__ __launch_bounds__(256) void ta_attn_decode_bwd_kernel(BwdArgs p) {
  __shared__ __attribute__((aligned(16))) float dO_s[D];
  __shared__ __attribute__((aligned(16))) float Q_s[D];
  __shared__ __attribute__((aligned(16))) float qt_s[D];
  __shared__ __attribute__((aligned(16))) float dqp_s[2 * D];
  __shared__ float w_s[MAXH][MAXL];    // softmax weights
  __shared__ float ds_s[MAXH][MAXL];   // dW, then dS, then d(QK)
  __shared__ float dap_s[MAXL];
  __shared__ __attribute__((aligned(16))) float partQ[8][D];
  __shared__ __attribute__((aligned(16))) float partT[8][D];
  __shared__ float red[4];

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int L = p.L, H = p.H;
  const int sl = min(max(p.seq_len[b], 0), L);
  const unsigned inv_L = (1u << 20) / (unsigned)L + 1u;      // i / L for i < H L <= 2,048 (i L < 2^20): exact
  const size_t row0 = (size_t)b * L;
  // H is 1, 2, 4 or 8 (checked by the host): shifts and masks, not run-time divisions (~25 instructions each)
  const int log_h = (H >= 2) + (H >= 4) + (H >= 8), lph_shift = 5 - log_h, lanes_per_head = 1 << lph_shift;
  const float inv_div = sqrtf((float)(D >> log_h));
  const float *sv = p.save + (size_t)b * save_floats(L, H);
  const float *sv_a = sv + 3 * D, *sv_dk = sv_a + L, *sv_sg = sv_dk + L;
  const float *sv_qk = sv_sg + L, *sv_w = sv_qk + H * L;
  const float rstd = sv_w[H * L];

  // As in the forward kernel, every global load that depends only on the kernel arguments is issued
  // before the first barrier: the sample's key / value / raw-key rows (and the running d_x rows when
  // accumulating), half a wave per row, and each key's gate inputs (thread j = key j).
  const int hw = tid >> 5, li = tid & 31;
  const int head_of_lane = li >> lph_shift;
  constexpr int KB = 8;
  float4 kq[KB], vq[KB], xq[KB], ox[KB];
  auto load_rows = [&](int jb) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const int jc = min(jb + 8 * i, L - 1);
      kq[i] = *reinterpret_cast<const float4 *>(&p.kv[(row0 + jc) * p.ld_kv + p.k_off + 4 * li]);
      vq[i] = *reinterpret_cast<const float4 *>(&p.kv[(row0 + jc) * p.ld_kv + p.v_off + 4 * li]);
      xq[i] = *reinterpret_cast<const float4 *>(&p.x[(row0 + jc) * D + 4 * li]);
    }
    if (p.accumulate_dx) {              // block-uniform
#pragma unroll
      for (int i = 0; i < KB; ++i)
        ox[i] = *reinterpret_cast<const float4 *>(&p.d_x[(row0 + min(jb + 8 * i, L - 1)) * D + 4 * li]);
    }
  };
  AT_STAMP(1, 0)
  load_rows(hw);
  AT_STAMP(1, 1)       // row loads issued
  const int jg = min(tid, L - 1);      // gate inputs of key `tid` (keys >= 256 are reloaded in the loop below)
  const float g_sg = sv_sg[jg], g_a = sv_a[jg], g_dk = sv_dk[jg];
  const float g_tk = p.t_keys[row0 + jg], g_tq = p.t_query[b];
  const float g_tp2 = p.tparams[2 * L + jg], g_tp3 = p.tparams[3 * L + jg];
  float g_qk[MAXH];
#pragma unroll
  for (int h = 0; h < MAXH; ++h) g_qk[h] = (h < H) ? sv_qk[h * L + jg] : 0.f;

  // ---- fused head layer_norm backward (last block only): dy = rstd * (a - mean(a) - h * mean(a h)), a = d_pred * gamma
  float dy_head = 0.f;
  if (p.d_pred) {                      // block-uniform
    float a = 0.f, hh = 0.f;
    if (tid < D) {
      const float yp = p.d_pred[(size_t)b * D + tid];
      hh = p.head_save[(size_t)b * (D + 1) + tid];
      a = yp * p.head_gamma[tid];
      p.d_head_partial[((size_t)b * 2 + 0) * D + tid] = yp;
      p.d_head_partial[((size_t)b * 2 + 1) * D + tid] = yp * hh;
      const float s1 = group_sum_dpp<64>(a), s2 = group_sum_dpp<64>(a * hh);
      if (lane == 0) { red[w] = s1; red[2 + w] = s2; }
    }
    __syncthreads();
    if (tid < D) {
      const float m1 = (red[0] + red[1]) / (float)D, m2 = (red[2] + red[3]) / (float)D;
      dy_head = p.head_save[(size_t)b * (D + 1) + D] * (a - m1 - hh * m2);
    }
    __syncthreads();                   // red[] is reused below
  }
  AT_STAMP(1, 2)       // head layer_norm backward
  // ---- normalize() backward
  float dxh = 0.f, xhat = 0.f;
  if (tid < D) {
    const float dy = p.d_pred ? dy_head : p.d_out[(size_t)b * D + tid];
    xhat = sv[2 * D + tid];
    p.d_ln_partial[((size_t)b * 2 + 0) * D + tid] = dy;
    p.d_ln_partial[((size_t)b * 2 + 1) * D + tid] = dy * xhat;
    dxh = dy * p.ln_gamma[tid];
    const float s1 = group_sum_dpp<64>(dxh), s2 = group_sum_dpp<64>(dxh * xhat);
    if (lane == 0) { red[w] = s1; red[2 + w] = s2; }
    Q_s[tid] = sv[tid];
    qt_s[tid] = sv[D + tid];
  }
  for (int i = tid; i < H * L; i += 256) {
    const int h = (int)(((unsigned)i * inv_L) >> 20), j = i - h * L;
    w_s[h][j] = sv_w[i];
  }
  __syncthreads();
  float dres = 0.f;
  if (tid < D) {
    const float m1 = (red[0] + red[1]) / (float)D, m2 = (red[2] + red[3]) / (float)D;
    dres = rstd * (dxh - m1 - xhat * m2);   // gradient of y = O + q
    dO_s[tid] = dres;
  }
  __syncthreads();

  AT_STAMP(1, 3)       // normalize backward, dO in LDS
  // ---- dW[h][j] = dO_h . V_j   (64 keys per trip; one trip for L <= 64, its rows are already here)
  {
    const float4 dO4 = *reinterpret_cast<const float4 *>(&dO_s[4 * li]);
    for (int jb = hw; jb < L; jb += 8 * KB) {
      if (jb != hw) load_rows(jb);
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        const int j = jb + 8 * i;
        if (j >= L) break;
        if (j < sl) {
          const float dW = group_sum_fast(dot4(vq[i], dO4), lanes_per_head);
          if ((li & (lanes_per_head - 1)) == 0) ds_s[head_of_lane][j] = dW;
        } else if (li < H) {
          ds_s[li][j] = 0.f;
        }
      }
    }
  }
  __syncthreads();
  AT_STAMP(1, 4)       // dW
  // ---- softmax backward: dS = W * (dW - sum_j W dW)
  for (int h = w; h < H; h += 4) {
    float s = 0.f;
    for (int j = lane; j < sl; j += 64) s += w_s[h][j] * ds_s[h][j];
    s = group_sum_dpp<64>(s);
    for (int j = lane; j < L; j += 64) ds_s[h][j] = (j < sl) ? w_s[h][j] * (ds_s[h][j] - s) : 0.f;
  }
  __syncthreads();
  AT_STAMP(1, 5)       // softmax backward
  // ---- gate backward, thread per key
  for (int j = tid; j < L; j += 256) {
    float g_w1 = 0.f, g_b1 = 0.f, g_ow1 = 0.f, g_ow2 = 0.f, g_ob = 0.f, dap = 0.f;
    if (j < sl) {
      const bool pre = j == tid;                   // always true for L <= 256
      const float sg = pre ? g_sg : sv_sg[j], a = pre ? g_a : sv_a[j], dk = pre ? g_dk : sv_dk[j];
      float dsg = 0.f;
#pragma unroll
      for (int h = 0; h < MAXH; ++h) {
        if (h < H) {
          const float dS = ds_s[h][j];
          dsg += dS * (pre ? g_qk[h] : sv_qk[h * L + j]);
          ds_s[h][j] = dS * sg / inv_div;          // d(Q_h . K_hj)
        }
      }
      dsg = dsg / inv_div;
      const float dG = dsg * sg * (1.f - sg);
      const float delta = logf(fabsf(g_tq - (pre ? g_tk : p.t_keys[row0 + j])) + 1.0f);
      g_ow1 = dG * dk;
      g_ow2 = dG * a;
      g_ob = dG;
      const float ddk = dG * (pre ? g_tp2 : p.tparams[2 * L + j]) * (1.f - dk * dk);
      g_w1 = ddk * delta;
      g_b1 = ddk;
      dap = dG * (pre ? g_tp3 : p.tparams[3 * L + j]) * (1.f - a * a);
    }
    dap_s[j] = dap;
    float *gp = p.d_tparams_partial + (size_t)b * 5 * L + j;
    gp[0] = g_w1; gp[L] = g_b1; gp[2 * L] = g_ow1; gp[3 * L] = g_ow2; gp[4 * L] = g_ob;
  }
  __syncthreads();
  AT_STAMP(1, 6)       // gate backward
  // ---- per key: dK, dV (relu-masked), raw-key gradient; accumulate dQ, d(qt)
  {
    const float4 dO4 = *reinterpret_cast<const float4 *>(&dO_s[4 * li]);
    const float4 Q4 = *reinterpret_cast<const float4 *>(&Q_s[4 * li]);
    const float4 T4 = *reinterpret_cast<const float4 *>(&qt_s[4 * li]);
    float4 aQ = make_float4(0.f, 0.f, 0.f, 0.f), aT = aQ;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int jb = hw; jb < L; jb += 8 * KB) {
      if (jb != hw || L > 8 * KB) load_rows(jb);       // L <= 64: the rows loaded at the top are still here
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        const int j = jb + 8 * i;
        if (j >= L) break;
        float *dk_row = p.d_kv + (row0 + j) * p.ld_kv + p.k_off + 4 * li;
        float *dv_row = p.d_kv + (row0 + j) * p.ld_kv + p.v_off + 4 * li;
        float *dx_row = p.d_x + (row0 + j) * D + 4 * li;
        if (j < sl) {
          const float wj = w_s[head_of_lane][j], dqk = ds_s[head_of_lane][j], dap = dap_s[j];
          float4 dv, dk;
          dv.x = vq[i].x > 0.f ? wj * dO4.x : 0.f; dv.y = vq[i].y > 0.f ? wj * dO4.y : 0.f;
          dv.z = vq[i].z > 0.f ? wj * dO4.z : 0.f; dv.w = vq[i].w > 0.f ? wj * dO4.w : 0.f;
          dk.x = kq[i].x > 0.f ? dqk * Q4.x : 0.f; dk.y = kq[i].y > 0.f ? dqk * Q4.y : 0.f;
          dk.z = kq[i].z > 0.f ? dqk * Q4.z : 0.f; dk.w = kq[i].w > 0.f ? dqk * Q4.w : 0.f;
          *reinterpret_cast<float4 *>(dv_row) = dv;
          *reinterpret_cast<float4 *>(dk_row) = dk;
          float4 dx = make_float4(dap * T4.x, dap * T4.y, dap * T4.z, dap * T4.w);
          if (p.accumulate_dx) { dx.x += ox[i].x; dx.y += ox[i].y; dx.z += ox[i].z; dx.w += ox[i].w; }
          *reinterpret_cast<float4 *>(dx_row) = dx;
          aQ.x = fmaf(dqk, kq[i].x, aQ.x); aQ.y = fmaf(dqk, kq[i].y, aQ.y);
          aQ.z = fmaf(dqk, kq[i].z, aQ.z); aQ.w = fmaf(dqk, kq[i].w, aQ.w);
          aT.x = fmaf(dap, xq[i].x, aT.x); aT.y = fmaf(dap, xq[i].y, aT.y);
          aT.z = fmaf(dap, xq[i].z, aT.z); aT.w = fmaf(dap, xq[i].w, aT.w);
        } else {
          *reinterpret_cast<float4 *>(dv_row) = zero4;
          *reinterpret_cast<float4 *>(dk_row) = zero4;
          if (!p.accumulate_dx) *reinterpret_cast<float4 *>(dx_row) = zero4;
        }
      }
    }
    *reinterpret_cast<float4 *>(&partQ[hw][4 * li]) = aQ;
    *reinterpret_cast<float4 *>(&partT[hw][4 * li]) = aT;
  }
  __syncthreads();
  AT_STAMP(1, 7)       // per-key gradients stored, partial dQ / d(qt) in LDS
  if (tid < D) {
    float dQ = 0.f, dT = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { dQ += partQ[q][tid]; dT += partT[q][tid]; }
    const float dQp = (Q_s[tid] > 0.f) ? dQ : 0.f;
    dqp_s[tid] = dQp;
    dqp_s[D + tid] = dT;
    p.d_qt_pre[(size_t)b * 2 * D + tid] = dQp;
    p.d_qt_pre[(size_t)b * 2 * D + D + tid] = dT;
  }
  __syncthreads();
  AT_STAMP(1, 8)       // dQ, d(qt) reduced
  // ---- d(dec_in)[c] = residual + [dQpre | dqt] . wqt[c, :]^T ; a wave per row, lanes along n
  {
    float dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[i] = dqp_s[lane + 64 * i];
    // the wave's 32 weight rows (128 loads) are all in flight before the first one is reduced; the
    // 32 wave sums run as four batches of eight independent chains (DPP adds inside the 16-lane rows)
    float wv[32][4];
#pragma unroll
    for (int r = 0; r < 32; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) wv[r][i] = p.wqt[(size_t)(w + 4 * r) * (2 * D) + lane + 64 * i];
#pragma unroll
    for (int r0 = 0; r0 < 32; r0 += 8) {
      float sv8[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s = fmaf(dq[i], wv[r0 + r][i], s);
        sv8[r] = s;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) sv8[r] = group_sum_dpp<64>(sv8[r]);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int c = w + 4 * (r0 + r);
        if (lane == 0) p.d_dec_in[(size_t)b * D + c] = sv8[r] + dO_s[c];
      }
    }
  }
  AT_STAMP(1, 9)       // d(dec_in)
}

}  // namespace

extern "C" int mtam_ta_attn_decode_save_floats(int L, int H) { return 3 * D + 3 * L + 2 * H * L + 1; }

static int check_attn_common(const char *name, int B, int L, int H, int ld_kv, int k_off, int v_off,
                             const float *kv, const float *x) {
  MTAM_CHECK_ARG(B > 0 && L > 0 && L <= MAXL, "%s: need 0 < L <= %d (got %d)", name, MAXL, L);
  MTAM_CHECK_ARG(H == 1 || H == 2 || H == 4 || H == 8, "%s: num_heads must be 1, 2, 4 or 8 (got %d)", name, H);
  MTAM_CHECK_ARG(ld_kv % 4 == 0 && k_off % 4 == 0 && v_off % 4 == 0 && k_off + D <= ld_kv && v_off + D <= ld_kv,
                 "%s: bad kv layout", name);
  MTAM_CHECK_ARG(mtam_aligned16(kv) && mtam_aligned16(x), "%s: kv and x must be 16-byte aligned", name);
  return MTAM_OK;
}

extern "C" int mtam_ta_attn_decode_fwd(const float *dec_in, const float *x, const float *kv, int ld_kv,
                                       int k_off, int v_off, const float *t_query, const float *t_keys,
                                       const int32_t *seq_len, const float *wqt, const float *bq,
                                       const float *tparams, const float *ln_beta, const float *ln_gamma,
                                       int B, int L, int H, float *dec_out, float *save, const float *head_beta,
                                       const float *head_gamma, float *pred_out, float *head_save,
                                       void *stream) {
  MTAM_CHECK_ARG(dec_in && x && kv && t_query && t_keys && seq_len && wqt && bq && tparams && ln_beta &&
                     ln_gamma && dec_out,
                 "ta_attn_decode_fwd: null argument");
  MTAM_CHECK_ARG(!pred_out || (head_beta && head_gamma), "ta_attn_decode_fwd: pred_out needs the head LN parameters");
  int rc = check_attn_common("ta_attn_decode_fwd", B, L, H, ld_kv, k_off, v_off, kv, x);
  if (rc) return rc;
  FwdArgs a{dec_in, x, kv, ld_kv, k_off, v_off, t_query, t_keys, seq_len, wqt, bq, tparams,
            ln_beta, ln_gamma, B, L, H, dec_out, save, head_beta, head_gamma, pred_out, head_save};
  hipLaunchKernelGGL(ta_attn_decode_fwd_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("ta_attn_decode_fwd");
  return MTAM_OK;
}

extern "C" int mtam_ta_attn_decode_bwd(const float *d_out, const float *dec_in, const float *x,
                                       const float *kv, int ld_kv, int k_off, int v_off,
                                       const float *t_query, const float *t_keys, const int32_t *seq_len,
                                       const float *wqt, const float *tparams, const float *ln_gamma,
                                       const float *save, int B, int L, int H, int accumulate_dx,
                                       float *d_dec_in, float *d_kv, float *d_x, float *d_qt_pre,
                                       float *d_tparams_partial, float *d_ln_partial, const float *d_pred,
                                       const float *head_gamma, const float *head_save, float *d_head_partial,
                                       void *stream) {
  MTAM_CHECK_ARG((d_out || d_pred) && dec_in && x && kv && t_query && t_keys && seq_len && wqt && tparams &&
                     ln_gamma && save && d_dec_in && d_kv && d_x && d_qt_pre && d_tparams_partial && d_ln_partial,
                 "ta_attn_decode_bwd: null argument");
  MTAM_CHECK_ARG(!d_pred || (head_gamma && head_save && d_head_partial),
                 "ta_attn_decode_bwd: d_pred needs head_gamma, head_save and d_head_partial");
  int rc = check_attn_common("ta_attn_decode_bwd", B, L, H, ld_kv, k_off, v_off, kv, x);
  if (rc) return rc;
  MTAM_CHECK_ARG(mtam_aligned16(d_kv) && mtam_aligned16(d_x), "ta_attn_decode_bwd: d_kv and d_x must be 16-byte aligned");
  BwdArgs a{d_out, dec_in, x, kv, ld_kv, k_off, v_off, t_query, t_keys, seq_len, wqt, tparams, ln_gamma,
            save, B, L, H, accumulate_dx, d_dec_in, d_kv, d_x, d_qt_pre, d_tparams_partial, d_ln_partial,
            d_pred, head_gamma, head_save, d_head_partial};
  hipLaunchKernelGGL(ta_attn_decode_bwd_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("ta_attn_decode_bwd");
  return MTAM_OK;
}
