// Time-aware multi-head attention, encoder form (T_q = T_k = L): the row-wise
// part of one self_attention block, Model/Modules/time_aware_attention.py:320-431
// as wired by Model/PISTRec_model.py:38-53.
//
// Here Q.K^T and (q.Wt).k^T are real L x L x D contractions; they
// and W.V run on the matrix cores through mtam_gemm_f32_batched.  What is left
// is per (sample, query row): the time gate, scaling, key mask, softmax over the
// L keys and the query mask -- one wave per row, lanes along the keys,
// wavefront-reduced max / sum.  The [L, L] gate parameters are indexed
// [query][key]; their gradients are summed over the batch with f32 atomics
// (contiguous 256 B per wave instruction).
#include "common.h"

namespace {

constexpr int MAXH = 8;
constexpr float MASK_VALUE = -4294967295.0f;   // -2**32 + 1 (time_aware_attention.py:392)

struct FwdArgs {
  const float *s_raw;     // [B, H, L, L]  Q_h . K_h^T
  float *a;               // [B, L, L]     in: (q Wt) k^T   out: tanh of it
  const float *t;         // [B, L]
  const int32_t *seq_len; // [B]
  const float *tparams;   // [5, L, L]
  int B, L, H, d;
  float *w;               // [B, H, L, L]  softmax weights (query rows >= seq_len zeroed)
  float *dk, *sg;         // [B, L, L]     saved tanh(decay), sigmoid(gate)
};

__global__ __launch_bounds__(256) void selfattn_gate_softmax_fwd_kernel(FwdArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);          // b * L + i
  if (row >= p.B * p.L) return;
  const int L = p.L, H = p.H;
  const int b = row / L, i = row - b * L;
  const int sl = min(max(p.seq_len[b], 0), L);
  const float ti = p.t[row];
  const float sqrt_d = sqrtf((float)p.d);
  const size_t LL = (size_t)L * L;
  float *a_row = p.a + (size_t)b * LL + (size_t)i * L;
  float *dk_row = p.dk + (size_t)b * LL + (size_t)i * L;
  float *sg_row = p.sg + (size_t)b * LL + (size_t)i * L;
  const float *tp = p.tparams + (size_t)i * L;
  const bool live_q = i < sl;                                   // query mask (:428-431)

  // gate per key (shared by the heads)
  for (int j = lane; j < L; j += 64) {
    float av = 0.f, dkv = 0.f, sgv = 0.f;
    if (live_q && j < sl) {
      av = fast_tanh(a_row[j]);
      const float delta = logf(fabsf(ti - p.t[(size_t)b * L + j]) + 1.0f);
      dkv = fast_tanh(delta * tp[j] + tp[LL + j]);
      sgv = fast_sigmoid(tp[2 * LL + j] * dkv + tp[3 * LL + j] * av + tp[4 * LL + j]);
    }
    a_row[j] = av;
    dk_row[j] = dkv;
    sg_row[j] = sgv;
  }
  for (int h = 0; h < H; ++h) {
    const float *s_row = p.s_raw + ((size_t)(b * H + h) * L + i) * L;
    float *w_row = p.w + ((size_t)(b * H + h) * L + i) * L;
    if (!live_q) {
      for (int j = lane; j < L; j += 64) w_row[j] = 0.f;
      continue;
    }
    float m = -INFINITY;
    for (int j = lane; j < L; j += 64) {
      const float v = (j < sl) ? (s_row[j] * sg_row[j]) / sqrt_d : MASK_VALUE;
      m = fmaxf(m, v);
    }
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) {
      const float v = (j < sl) ? (s_row[j] * sg_row[j]) / sqrt_d : MASK_VALUE;
      sum += fast_exp(v - m);
    }
    sum = wave_sum(sum);
    for (int j = lane; j < L; j += 64) {
      const float v = (j < sl) ? (s_row[j] * sg_row[j]) / sqrt_d : MASK_VALUE;
      w_row[j] = fast_exp(v - m) / sum;
    }
  }
}

struct BwdArgs {
  float *dw;              // [B, H, L, L]  in: d loss / d W   out: d loss / d (Q_h . K_h^T)
  const float *w, *s_raw; // [B, H, L, L]
  const float *a, *dk, *sg; // [B, L, L]
  const float *t;
  const int32_t *seq_len;
  const float *tparams;   // [5, L, L]
  int B, L, H, d;
  float *d_a;             // [B, L, L]     d loss / d ((q Wt) k^T)
  float *g_tparams;       // [5, L, L]     += (atomic)
};

__global__ __launch_bounds__(256) void selfattn_gate_softmax_bwd_kernel(BwdArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.B * p.L) return;
  const int L = p.L, H = p.H;
  const int b = row / L, i = row - b * L;
  const int sl = min(max(p.seq_len[b], 0), L);
  const float sqrt_d = sqrtf((float)p.d);
  const size_t LL = (size_t)L * L;
  const size_t ro = (size_t)b * LL + (size_t)i * L;
  float *da_row = p.d_a + ro;
  if (i >= sl) {                                                 // masked query: no gradient at all
    for (int j = lane; j < L; j += 64) da_row[j] = 0.f;
    for (int h = 0; h < H; ++h) {
      float *dw_row = p.dw + ((size_t)(b * H + h) * L + i) * L;
      for (int j = lane; j < L; j += 64) dw_row[j] = 0.f;
    }
    return;
  }
  const float ti = p.t[row];
  const float *tp = p.tparams + (size_t)i * L;
  float *gp = p.g_tparams + (size_t)i * L;
  // softmax backward per head: dSm = W (dW - sum W dW); dsg accumulates over the heads
  float dsg[4] = {0.f, 0.f, 0.f, 0.f};                           // keys lane, lane+64, lane+128, lane+192
  for (int h = 0; h < H; ++h) {
    const size_t ho = ((size_t)(b * H + h) * L + i) * L;
    const float *w_row = p.w + ho, *s_row = p.s_raw + ho;
    float *dw_row = p.dw + ho;
    float dot = 0.f;
    for (int j = lane; j < sl; j += 64) dot += w_row[j] * dw_row[j];
    dot = wave_sum(dot);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = lane + 64 * q;
      if (j < L) {
        float dqk = 0.f;
        if (j < sl) {
          const float dsm = w_row[j] * (dw_row[j] - dot);
          dqk = dsm * p.sg[ro + j] / sqrt_d;
          dsg[q] += dsm * s_row[j] / sqrt_d;
        }
        dw_row[j] = dqk;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = lane + 64 * q;
    if (j >= L) continue;
    float dap = 0.f;
    if (j < sl) {
      const float sg = p.sg[ro + j], a = p.a[ro + j], dk = p.dk[ro + j];
      const float dG = dsg[q] * sg * (1.f - sg);
      const float delta = logf(fabsf(ti - p.t[(size_t)b * L + j]) + 1.0f);
      const float ddk = dG * tp[2 * LL + j] * (1.f - dk * dk);
      atomicAdd(gp + j, ddk * delta);                            // _time_input_w1
      atomicAdd(gp + LL + j, ddk);                               // _time_input_b1
      atomicAdd(gp + 2 * LL + j, dG * dk);                       // time_output_w1
      atomicAdd(gp + 3 * LL + j, dG * a);                        // time_output_w2
      atomicAdd(gp + 4 * LL + j, dG);                            // time_output_b
      dap = dG * tp[3 * LL + j] * (1.f - a * a);
    }
    da_row[j] = dap;
  }
}

// gather_indexes(seq, seq_len + offset) and its gradient (Model/Modules/net_utils.py:82-92)
__global__ __launch_bounds__(256) void seq_row_gather_kernel(const float *__restrict__ src,
                                                             const int32_t *__restrict__ seq_len, int offset,
                                                             int B, int L, float *__restrict__ out) {
  const int b = blockIdx.x * 2 + (threadIdx.x >> 7), c = threadIdx.x & 127;
  if (b >= B) return;
  const int t = min(max(seq_len[b] + offset, 0), L - 1);
  out[(size_t)b * MTAM_D + c] = src[((size_t)b * L + t) * MTAM_D + c];
}
__global__ __launch_bounds__(256) void seq_row_scatter_kernel(const float *__restrict__ d_out,
                                                              const int32_t *__restrict__ seq_len, int offset,
                                                              int B, int L, float *__restrict__ d_src) {
  const size_t r = (size_t)blockIdx.x * 2 + (threadIdx.x >> 7);
  const int c = threadIdx.x & 127;
  if (r >= (size_t)B * L) return;
  const int b = (int)(r / L), t = (int)(r - (size_t)b * L);
  const int tt = min(max(seq_len[b] + offset, 0), L - 1);
  d_src[r * MTAM_D + c] = (t == tt) ? d_out[(size_t)b * MTAM_D + c] : 0.f;
}

// d[i] = y[i] > 0 ? d[i] : 0  (gradient through tf.nn.relu)
__global__ __launch_bounds__(256) void relu_bwd_kernel(float *__restrict__ d, const float *__restrict__ y, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 dv = reinterpret_cast<float4 *>(d)[i];
  const float4 yv = reinterpret_cast<const float4 *>(y)[i];
  dv.x = yv.x > 0.f ? dv.x : 0.f; dv.y = yv.y > 0.f ? dv.y : 0.f;
  dv.z = yv.z > 0.f ? dv.z : 0.f; dv.w = yv.w > 0.f ? dv.w : 0.f;
  reinterpret_cast<float4 *>(d)[i] = dv;
}

}  // namespace

extern "C" int mtam_ta_selfattn_gate_softmax_fwd(const float *s_raw, float *a, const float *t,
                                                 const int32_t *seq_len, const float *tparams, int B, int L,
                                                 int H, float *w, float *dk, float *sg, void *stream) {
  MTAM_CHECK_ARG(s_raw && a && t && seq_len && tparams && w && dk && sg, "selfattn_fwd: null argument");
  MTAM_CHECK_ARG(B > 0 && L > 0 && L <= 256 && H >= 1 && H <= MAXH && MTAM_D % H == 0, "selfattn_fwd: bad sizes");
  FwdArgs args{s_raw, a, t, seq_len, tparams, B, L, H, MTAM_D / H, w, dk, sg};
  hipLaunchKernelGGL(selfattn_gate_softmax_fwd_kernel, dim3((B * L + 3) / 4), dim3(256), 0,
                     static_cast<hipStream_t>(stream), args);
  MTAM_CHECK_LAUNCH("selfattn_fwd");
  return MTAM_OK;
}

extern "C" int mtam_ta_selfattn_gate_softmax_bwd(float *dw, const float *w, const float *s_raw, const float *a,
                                                 const float *dk, const float *sg, const float *t,
                                                 const int32_t *seq_len, const float *tparams, int B, int L,
                                                 int H, float *d_a, float *g_tparams, void *stream) {
  MTAM_CHECK_ARG(dw && w && s_raw && a && dk && sg && t && seq_len && tparams && d_a && g_tparams,
                 "selfattn_bwd: null argument");
  MTAM_CHECK_ARG(B > 0 && L > 0 && L <= 256 && H >= 1 && H <= MAXH && MTAM_D % H == 0, "selfattn_bwd: bad sizes");
  BwdArgs args{dw, w, s_raw, a, dk, sg, t, seq_len, tparams, B, L, H, MTAM_D / H, d_a, g_tparams};
  hipLaunchKernelGGL(selfattn_gate_softmax_bwd_kernel, dim3((B * L + 3) / 4), dim3(256), 0,
                     static_cast<hipStream_t>(stream), args);
  MTAM_CHECK_LAUNCH("selfattn_bwd");
  return MTAM_OK;
}

extern "C" int mtam_seq_row_gather(const float *src, const int32_t *seq_len, int offset, int B, int L,
                                   float *out, void *stream) {
  MTAM_CHECK_ARG(src && seq_len && out && B > 0 && L > 0, "seq_row_gather: bad arguments");
  hipLaunchKernelGGL(seq_row_gather_kernel, dim3((B + 1) / 2), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                     seq_len, offset, B, L, out);
  MTAM_CHECK_LAUNCH("seq_row_gather");
  return MTAM_OK;
}

extern "C" int mtam_seq_row_scatter(const float *d_out, const int32_t *seq_len, int offset, int B, int L,
                                    float *d_src, void *stream) {
  MTAM_CHECK_ARG(d_out && seq_len && d_src && B > 0 && L > 0, "seq_row_scatter: bad arguments");
  const size_t rows = (size_t)B * L;
  hipLaunchKernelGGL(seq_row_scatter_kernel, dim3((unsigned)((rows + 1) / 2)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d_out, seq_len, offset, B, L, d_src);
  MTAM_CHECK_LAUNCH("seq_row_scatter");
  return MTAM_OK;
}

extern "C" int mtam_relu_bwd_inplace(float *d, const float *y, size_t n, void *stream) {
  MTAM_CHECK_ARG(d && y && n > 0 && n % 4 == 0 && mtam_aligned16(d) && mtam_aligned16(y),
                 "relu_bwd: n must be a multiple of 4 and buffers 16-byte aligned");
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d, y, n4);
  MTAM_CHECK_LAUNCH("relu_bwd");
  return MTAM_OK;
}
