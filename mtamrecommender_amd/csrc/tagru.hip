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
// so no partial sum crosses waves through LDS: only r*h (after the gate phase) and the new state (after the
// candidate phase) do.  A step reads the 128 state values as 4 ds_read_b128 per lane (8 distinct 64-byte
// chunks per instruction, laid out [64][4 pad][64] so that chunks kp and kp + 4 fall on different banks).
// Round 1 cut the weights by k-slice per wave: 8 partial sums per column crossed waves through LDS, 4 barriers.
//
// What bounds a step (tools/gru_lab.hip, tools/valu_lab.hip; profiles/r02_gru_lab_*.txt): neither bandwidth nor
// FMA rate but the length of ONE wave's dependent instruction stream -- a wave issues a v_fma_f32 every ~5.5
// cycles, a v_pk_fma_f32 every ~8.5, a DEPENDENT one every ~15-17, and a step is ~45 dependent element-wise /
// reduction ops around two LDS round trips and two barriers: ~0.8-1.0 us at ~2 GHz in every layout tried
// (4 barriers / 8 waves: 1.0; 2 barriers / 8 waves: 0.95; 16 waves with one unit per octet: 1.0 -- four waves
// per SIMD starve the youngest; 8 waves, two units per octet, all per-step traffic staged in LDS: 0.8).
#include "common.h"
#include "split_bf16.h"
#include "stripe_tile.h"
#include "gru_image.h"

// In-kernel stamps for tools/gru_lab.hip (a diagnostic build, -DMTAM_GRU_STAMPS): cycles per segment of a
// step, summed per wave of workgroup 0 and written to a buffer of their own.  The product build has none.
#ifdef MTAM_GRU_STAMPS
__device__ unsigned long long g_gru_stamps[2][8][8];      // [kernel][wave][segment]
__device__ unsigned long long g_gru_real[2][8][2];        // [kernel][wave]: s_memtime ticks and s_memrealtime ticks (100 MHz) of the loop
#define GRU_STAMP_DECL unsigned long long st_last_, st_t0_ = 0, st_r0_ = 0, st_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define GRU_STAMP_START                                                                \
  st_r0_ = __builtin_amdgcn_s_memrealtime();                                           \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last_)::"memory");    \
  st_t0_ = st_last_;
#define GRU_STAMP(i)                                                                   \
  {                                                                                    \
    unsigned long long t_;                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    st_acc_[i] += t_ - st_last_;                                                       \
    st_last_ = t_;                                                                     \
  }
__device__ unsigned long long g_gru_phase[2][8][8];       // [kernel][wave][phase]: s_memrealtime ticks (10 ns) between phase marks
__device__ unsigned long long g_gru_wg[2][256][2];        // [kernel][workgroup]: s_memrealtime at its start and end
__device__ unsigned long long g_gru_span[2][64][2];       // [kernel][launch & 63]: earliest start, latest end over the launch's workgroups
__device__ unsigned int g_gru_launch[2];                  // launches so far
#define GRU_PHASE_DECL unsigned long long ph_last_ = __builtin_amdgcn_s_memrealtime(), ph_t0_ = ph_last_, ph_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
  const unsigned int ph_launch_ = __hip_atomic_load(&g_gru_launch[GRU_KERNEL_ID], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#define GRU_PHASE(i)                                                                   \
  {                                                                                    \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
    const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                    \
    ph_acc_[i] += t_ - ph_last_;                                                       \
    ph_last_ = t_;                                                                     \
  }
#define GRU_PHASE_DUMP(k)                                                              \
  if (threadIdx.x == 0 && blockIdx.x < 256) {                                          \
    const unsigned long long te_ = __builtin_amdgcn_s_memrealtime();                   \
    g_gru_wg[k][blockIdx.x][0] = ph_t0_;                                               \
    g_gru_wg[k][blockIdx.x][1] = te_;                                                  \
    atomicMin(&g_gru_span[k][ph_launch_ & 63][0], ph_t0_);                             \
    atomicMax(&g_gru_span[k][ph_launch_ & 63][1], te_);                                \
    if (blockIdx.x == 0) atomicAdd(&g_gru_launch[k], 1u);                              \
  }                                                                                    \
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                      \
    for (int i_ = 0; i_ < 8; ++i_) g_gru_phase[k][threadIdx.x >> 6][i_] = ph_acc_[i_];
#define GRU_STAMP_DUMP(k)                                                              \
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                    \
    for (int i_ = 0; i_ < 8; ++i_) g_gru_stamps[k][threadIdx.x >> 6][i_] = st_acc_[i_]; \
    g_gru_real[k][threadIdx.x >> 6][0] = st_last_ - st_t0_;                            \
    g_gru_real[k][threadIdx.x >> 6][1] = __builtin_amdgcn_s_memrealtime() - st_r0_;    \
  }
#else
#define GRU_STAMP_DECL
#define GRU_STAMP_START
#define GRU_STAMP(i)
#define GRU_STAMP_DUMP(k)
#define GRU_PHASE_DECL
#define GRU_PHASE(i)
#define GRU_PHASE_DUMP(k)
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
// Two per-lane partial sums -> the full sum of column 0 on lanes kp = 0, 1 and of column 1 on lanes kp = 2, 3.
__device__ __forceinline__ float octet_reduce2_pairs(float s0, float s1, int lane) {
  const bool b1 = lane & 2;
  float keep = b1 ? s1 : s0;
  const float send = b1 ? s0 : s1;
  keep += dpp_f<DPP_XOR2>(send);
  keep += dpp_f<DPP_XOR1>(keep);
  keep += dpp_f<DPP_SHL4>(keep);
  return keep;
}
// A [rows x cols] fp32 matrix (cols a multiple of 4, rows * cols / 4 a multiple of 512 * 8) from global memory to
// LDS with row pitch `pitch`, by 512 threads: 8 sixteen-byte loads in flight per thread before the first LDS
// write (a load inside a loop that consumes it is waited for one round trip at a time).
template <int ROWS, int COLS, int PITCH>
__device__ __forceinline__ void stage_matrix_512(const float *__restrict__ src, float *dst, int tid) {
  constexpr int N4 = ROWS * COLS / 4;
  static_assert(N4 % (512 * 8) == 0, "stage_matrix_512: size");
#pragma unroll 1
  for (int base = 0; base < N4; base += 512 * 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(src + 4 * (size_t)(base + u * 512 + tid));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 512 + tid, k = i / (COLS / 4), c4 = i - k * (COLS / 4);
      float *d = dst + k * PITCH + 4 * c4;
      d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
    }
  }
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
  // K/V role (workgroups blockIdx >= B of the same launch, on the CUs the recurrence leaves idle):
  // kv_out [B L, kv_n] = relu(x . Wkv + kv_bias) from the bf16 operand images of Wkv (kv_img); NULL = no role
  const uint16_t *kv_img;
  const float *kv_bias;
  float *kv_out;
  int kv_n;
  // the recurrent weights in the order the forward's lanes hold them (gru_wimg_pos): 24 sixteen-byte pieces per
  // thread, piece j of thread t at float4 index j * 512 + t; NULL = go through LDS (below)
  const float *w_img;
};

constexpr int GRU_WIMG_FLOATS = gru_image::FLOATS;
__device__ __forceinline__ int gru_wimg_pos(int which, int k, int n) { return gru_image::pos(which, k, n); }

__global__ __launch_bounds__(256) void gru_weight_image_kernel(const float *__restrict__ wh_g,
                                                               const float *__restrict__ wh_c, float *__restrict__ img) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < D * 2 * D) img[gru_wimg_pos(0, i / (2 * D), i % (2 * D))] = wh_g[i];
  else if (i < D * 3 * D) img[gru_wimg_pos(1, (i - D * 2 * D) / D, (i - D * 2 * D) % D)] = wh_c[i - D * 2 * D];
}

// ---------------------------------------------------------------------------------------------------------------
// Work co-scheduled with the recurrence.  A GRU launch keeps 128 of the 256 CUs busy for ~46 us (one sample per
// workgroup, 150 KB of LDS each -- or, in the backward, a dynamic LDS reservation of the same effect: nothing else
// fits beside a GRU workgroup on its CU).  The decoder's K/V work does not depend on the recurrence, so it rides in
// the SAME launches as extra workgroups (blockIdx >= B), which the dispatcher can only place on the idle CUs:
//   forward   kv = relu(x Wkv + bkv)            (Model/Modules/time_aware_attention.py:251-253) leaves the fused
//             lookups + projections kernel: 21.1 -> 16.6 us
//   backward  d_x += d_kv Wkv^T                 (its gradient towards x) leaves the backward stripe kernel: 20.5 -> 16.8 us
// As launches of their own these products take 7.8 and 8.1 us; here they end long before the recurrence does.
// Both are split-bf16 products on the weight images the optimizer launch writes (csrc/seq_chain.hip).
using split_bf16::bf16x4;
using split_bf16::bf16x8;
using split_bf16::Tri;
using stripe::f32x16;
using stripe::f32x4;

constexpr int KV_ROLE_LDS = stripe::ROWS * stripe::X_PITCH * 4 + 8 * stripe::ROWS * stripe::T_PITCH * 4;      // 53,760 B

// 512 threads, one 32-row stripe of x: wave w owns column blocks w, w + 8, ... of the kv_n / 32
__device__ __forceinline__ void kv_role(const FwdArgs &p, int stripe_id, float *lds) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const long row0 = (long)stripe_id * stripe::ROWS;
  const int R = p.B * p.L, n_kv = p.kv_n;
  constexpr int XP = stripe::X_PITCH;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = i * 512 + tid, row = id >> 5, c4 = id & 31;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(p.x + min(row0 + row, (long)R - 1) * D + c4 * 4);
    *reinterpret_cast<f32x4 *>(lds + row * XP + c4 * 4) = v;
  }
  __syncthreads();
  Tri af[8];
  {
    const float *a = lds + r * XP + 8 * h;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(a + 16 * s), hi = *reinterpret_cast<const f32x4 *>(a + 16 * s + 4);
      const float x8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      af[s] = split_bf16::split8(x8);
    }
  }
  float *scratch = lds + stripe::ROWS * XP + w * (stripe::ROWS * stripe::T_PITCH);
  const size_t term = (size_t)D * n_kv;
  for (int j = w; j < n_kv / 32; j += 8) {
    const int col = 32 * j + r;
    const uint16_t *base = p.kv_img + ((size_t)h * n_kv + col) * 8;
    Tri bw[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        bw[s].t[t] = *reinterpret_cast<const bf16x8 *>(base + t * term + (size_t)s * (2 * n_kv * 8));
    const float bias = p.kv_bias[col];
    f32x16 a0 = {0.f}, a1 = {0.f};
#pragma unroll
    for (int s = 0; s < 8; s += 2) split_bf16::mfma6x2(af[s], bw[s], a0, af[s + 1], bw[s + 1], a1);
    const f32x16 acc = a0 + a1;
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 16; ++q) o[q] = fmaxf(acc[q] + bias, 0.f);
    stripe::store_tile(scratch, o, p.kv_out, row0, R, n_kv, 32 * j, lane);
  }
}

// ---- forward ----------------------------------------------------------------------------------------------
//  * 8 waves (512 threads, two per SIMD); octet q2 owns hidden units 2 q2 and 2 q2 + 1 entirely: their r, u and
//    candidate columns (6 x 16 weights per lane), their time gates and states.  r, u, T and h stay in the owner
//    lanes' registers (kp 0: unit 2 q2, kp 2: unit 2 q2 + 1); only r*h and the new state cross lanes through LDS.
//  * NO global memory access inside the step loop: a chunk of up to TCH steps of every per-step input is
//    staged in LDS beforehand by all threads (16-byte loads 8 deep, pre-scaled by -log2 e so that a sigmoid is
//    add, exp2, add, rcp), the step's results overwrite the consumed inputs in place, and the chunk is written
//    back afterwards with 16-byte stores.  150 KB of the CU's 160 KB LDS.
//  * the input-only part of the time gate (x kw1 + kb1, w12 relu(w1 dt + b1) + b12) is computed in the staging
//    pass; the loop keeps relu(A + h hw1), one fma and a sigmoid.
//  * the recurrent weights go global -> LDS (whole 1-KB rows) -> registers: 48 eight-byte loads per lane straight
//    from global memory cost 9.5 us per launch, through LDS ~5.
// 47.2 us per launch at B = 128, L = 50 (the 2-barrier kernel with global loads and stores in the loop: 50.0).
constexpr int TCH = 50;            // steps per staged chunk
constexpr int ST = 6 * D;          // floats per staged step: G (r | u) 2 D, C D, A D, S D, H D
constexpr int OFF_C = 2 * D, OFF_A = 3 * D, OFF_S = 4 * D, OFF_H = 5 * D;
constexpr float L2E_ = 1.4426950408889634f;

__global__ __launch_bounds__(512) void tagru_fwd_kernel(FwdArgs p) {
  __shared__ __attribute__((aligned(16))) float stage[TCH * ST];
  __shared__ __attribute__((aligned(16))) float h_s[D + 8];
  __shared__ __attribute__((aligned(16))) float rh_s[D + 8];
  __shared__ float carry_s[D];             // the last state of the previous chunk (h_prev of a chunk's first step)

  if ((int)blockIdx.x >= p.B) {             // the K/V role: not a sample but a 32-row stripe of x (see kv_role)
    static_assert(sizeof(stage) >= KV_ROLE_LDS, "the K/V role's LDS lives in the staging buffer");
    kv_role(p, blockIdx.x - p.B, stage);
    return;
  }
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int kp = lane & 7, q2 = tid >> 3;  // octet q2 (0..63) owns hidden units 2 q2 and 2 q2 + 1
  const int steps = min(max(p.seq_len[b] - 1, 0), p.L);
  const size_t row0 = (size_t)b * p.L;
  const bool seqrec = p.ldx == 5 * D;
#define GRU_KERNEL_ID 0
  GRU_PHASE_DECL
#undef GRU_KERNEL_ID
  // tvec == nullptr: the plain tf GRUCell (Model/Modules/gru.py:13-39) -- no time gate, T = 1
  const bool plain = p.tvec == nullptr && !seqrec;
  const int ldx = p.ldx, nsave = seqrec ? 6 : 5;

  // ---- recurrent weights -> registers, pre-scaled: gates by -log2 e, candidate by -2 log2 e
  // gate columns in the order of octet_reduce4's result lanes: r(2 q2) on kp 0, u(2 q2) on kp 1, r(2 q2 + 1) on
  // kp 2, u(2 q2 + 1) on kp 3  <=>  reduce4 inputs (s0, s1, s2, s3) = (r_a, r_b, u_a, u_b)
  // The weights go global -> LDS (1-KB rows, 16-byte loads, every byte once) -> registers.  Straight from global
  // memory each lane issued 48 eight-byte loads of 8 rows per wave instruction: 9.5 us of a 52 us kernel
  // (tools/gru_lab.hip with the loads skipped).  Rows are staged 257 / 129 floats apart so that the 8 row
  // groups of a wave's read fall on different banks.
  // Round 3: with the image the optimizer launch keeps beside the weights (p.w_img: the same values in exactly this
  // register order) a lane's 96 weights are 24 coalesced 16-byte loads, requested HERE and scaled into place behind
  // the staging of the first chunk -- the ~5 us of the LDS route (three barriers, two LDS round trips) disappear
  // into the shadow of the staging loads.
  f32x2 wra[8], wrb[8], wua[8], wub[8], wca[8], wcb[8];
  if (p.w_img) {            // raw values into the registers they stay in; scaled in place behind the staging
    const float4 *wi = reinterpret_cast<const float4 *>(p.w_img) + tid;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float4 a = wi[kk * 512], b = wi[(8 + kk) * 512], c = wi[(16 + kk) * 512];
      wra[kk] = f32x2{a.x, a.y}; wrb[kk] = f32x2{a.z, a.w};
      wua[kk] = f32x2{b.x, b.y}; wub[kk] = f32x2{b.z, b.w};
      wca[kk] = f32x2{c.x, c.y}; wcb[kk] = f32x2{c.z, c.w};
    }
  } else {
    constexpr int GP = 2 * D + 1, CP = D + 1;
    stage_matrix_512<D, 2 * D, GP>(p.wh_g, stage, tid);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float *g0 = &stage[(16 * kp + 2 * kk) * GP + 2 * q2], *g1 = g0 + GP;
      wra[kk] = f32x2{g0[0], g1[0]} * (-L2E_); wrb[kk] = f32x2{g0[1], g1[1]} * (-L2E_);
      wua[kk] = f32x2{g0[D], g1[D]} * (-L2E_); wub[kk] = f32x2{g0[D + 1], g1[D + 1]} * (-L2E_);
    }
    __syncthreads();
    stage_matrix_512<D, D, CP>(p.wh_c, stage, tid);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float *c0 = &stage[(16 * kp + 2 * kk) * CP + 2 * q2], *c1 = c0 + CP;
      wca[kk] = f32x2{c0[0], c1[0]} * (-2.f * L2E_); wcb[kk] = f32x2{c0[1], c1[1]} * (-2.f * L2E_);
    }
  }
  // lanes kp 0, 1 work on unit 2 q2, lanes kp 2, 3 on unit 2 q2 + 1 (kp >= 4: copies, never stored)
  const int q = 2 * q2 + ((kp >> 1) & 1);
  // time-gate parameters of unit q (owner lane), pre-scaled where they feed the sigmoid
  float hw1 = 0.f, kw2s = 0.f;
  if (!plain && !seqrec) {
    hw1 = p.tvec[HW1 * D + q];
    kw2s = -L2E_ * p.tvec[KW2 * D + q];
  }
  if (tid < D + 8) h_s[tid] = 0.f;
  if (tid < D) carry_s[tid] = 0.f;

  const int vpos = 16 * kp + 4 * (kp >> 2);        // this lane's 16 values of a padded 128-vector
  const int gslot = (kp & 1) ? D + q : q;          // even lane of the pair: r of unit q, odd lane: u of unit q
  const bool owner = kp == 0 || kp == 2;
  float h_own = 0.f;                               // state of unit q (owner lane)
  GRU_STAMP_DECL
  GRU_PHASE(0)          // weights: global -> LDS -> registers

  for (int t0 = 0; t0 < steps; t0 += TCH) {
    const int nch = min(TCH, steps - t0);
    __syncthreads();                               // the previous chunk's write-back has read the stage
    GRU_PHASE(1)
    // ---- stage the chunk's inputs (all threads, 16-byte loads, pre-scaled).  Loads go out 8 per thread before
    // the first of them is used (clamped index, masked store): inside a run-time-bounded loop they are waited
    // for one round trip at a time -- 8.7 us of fixed cost per launch before this was batched.
    {
      const int ncol4 = seqrec ? 5 * D / 4 : 3 * D / 4;              // G | C (| A | S: the two hoisted gates)
      const int total = nch * ncol4;
#pragma unroll 1
      for (int base = 0; base < total; base += 512 * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = min(base + u * 512 + tid, total - 1), sidx = i / ncol4, c4 = i - sidx * ncol4;
          v[u] = *reinterpret_cast<const float4 *>(p.xproj + (row0 + t0 + sidx) * ldx + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + u * 512 + tid, sidx = i / ncol4, c4 = i - sidx * ncol4;
          const float sc = (c4 >= 2 * D / 4 && c4 < 3 * D / 4) ? -2.f * L2E_ : -L2E_;
          if (i < total)
            *reinterpret_cast<float4 *>(&stage[sidx * ST + 4 * c4]) =
                make_float4(v[u].x * sc, v[u].y * sc, v[u].z * sc, v[u].w * sc);
        }
      }
    }
    GRU_PHASE(2)        // x-projection rows staged
    if (!plain && !seqrec) {                                         // A = x kw1 + kb1 ; S = -log2 e (w12 relu(w1 dt + b1) + b12)
      const int j = tid & (D - 1);
      const float kw1 = p.tvec[KW1 * D + j], kb1 = p.tvec[KB1 * D + j], w1 = p.tvec[W1 * D + j],
                  b1 = p.tvec[B1 * D + j], w12 = p.tvec[W12 * D + j], b12 = p.tvec[B12 * D + j];
#pragma unroll 1
      for (int sb = tid >> 7; sb < nch; sb += 4 * 8) {
        float xv[8], dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const size_t r = row0 + t0 + min(sb + 4 * u, nch - 1);
          xv[u] = p.x[r * D + j];
          dv[u] = p.timelast[r];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int sidx = sb + 4 * u;
          if (sidx < nch) {
            stage[sidx * ST + OFF_A + j] = fmaf(xv[u], kw1, kb1);
            stage[sidx * ST + OFF_S + j] = -L2E_ * fmaf(w12, fmaxf(fmaf(w1, dv[u], b1), 0.f), b12);
          }
        }
      }
    }
    __syncthreads();
    GRU_PHASE(3)        // time-gate inputs staged
    if (p.w_img && t0 == 0) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        wra[kk] = wra[kk] * (-L2E_); wrb[kk] = wrb[kk] * (-L2E_);
        wua[kk] = wua[kk] * (-L2E_); wub[kk] = wub[kk] * (-L2E_);
        wca[kk] = wca[kk] * (-2.f * L2E_); wcb[kk] = wcb[kk] * (-2.f * L2E_);
      }
    }

    GRU_STAMP_START
    for (int s = 0; s < nch; ++s) {
      float *st = stage + s * ST;
      // ---- phase 1: r and u of unit q (2 columns x 16 k per lane), reduced inside the octet
      float4 hq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) hq[i] = *reinterpret_cast<const float4 *>(&h_s[vpos + 4 * i]);
      const float in_g = st[gslot], in_c = st[OFF_C + q], in_a = st[OFF_A + q], in_s = st[OFF_S + q];
      // the time gate needs only staged inputs and the OLD state: issued in the shadow of the state reads
      float T = 1.f, N = 1.f;
      if (seqrec) {
        N = fast_rcp(1.f + fast_exp2(in_a));
        T = fast_rcp(1.f + fast_exp2(in_s));
      } else if (!plain) {
        T = fast_rcp(1.f + fast_exp2(fmaf(kw2s, fmaxf(fmaf(h_own, hw1, in_a), 0.f), in_s)));
      }
      f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 lo = {hq[i].x, hq[i].y}, hi = {hq[i].z, hq[i].w};
        a0 = pk_fma(lo, wra[2 * i], a0); a1 = pk_fma(lo, wrb[2 * i], a1);
        a2 = pk_fma(lo, wua[2 * i], a2); a3 = pk_fma(lo, wub[2 * i], a3);
        a0 = pk_fma(hi, wra[2 * i + 1], a0); a1 = pk_fma(hi, wrb[2 * i + 1], a1);
        a2 = pk_fma(hi, wua[2 * i + 1], a2); a3 = pk_fma(hi, wub[2 * i + 1], a3);
      }
      GRU_STAMP(0)      // state reads, time gate, gate FMAs
      const float gsum = octet_reduce4(a0.x + a0.y, a1.x + a1.y, a2.x + a2.y, a3.x + a3.y, lane);
      const float sg = fast_rcp(1.f + fast_exp2(gsum + in_g));      // even lane of the pair: r, odd lane: u
      const float u = dpp_f<DPP_XOR1>(sg);                           // (on the owner lane)
      if (kp < 4) st[gslot] = sg;                                    // saved r | u take the place of their inputs
      if (owner) rh_s[padpos(q)] = sg * h_own;
      GRU_STAMP(1)      // octet reduce, sigmoid, LDS writes
      __syncthreads();
      GRU_STAMP(2)      // barrier 1

      // ---- phase 2: candidate of unit q on r*h (16 k per lane), state update on the owner lane
      float4 rq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) rq[i] = *reinterpret_cast<const float4 *>(&rh_s[vpos + 4 * i]);
      f32x2 ca = {0.f, 0.f}, cb = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 lo = {rq[i].x, rq[i].y}, hi = {rq[i].z, rq[i].w};
        ca = pk_fma(lo, wca[2 * i], ca); cb = pk_fma(lo, wcb[2 * i], cb);
        ca = pk_fma(hi, wca[2 * i + 1], ca); cb = pk_fma(hi, wcb[2 * i + 1], cb);
      }
      const float csum = octet_reduce2_pairs(ca.x + ca.y, cb.x + cb.y, lane);   // unit 2 q2 on kp 0, 1; 2 q2 + 1 on kp 2, 3
      GRU_STAMP(3)      // r*h reads, candidate FMAs, reduce
      // tanh(z) = 2 sigmoid(2 z) - 1; csum and in_c carry the factor -2 log2 e
      const float c = fmaf(2.f, fast_rcp(1.f + fast_exp2(csum + in_c)), -1.f);
      const float hn = u * h_own * N + (1.f - u) * c * T;
      if (owner) {
        h_s[padpos(q)] = hn;
        st[OFF_C + q] = c;
        st[OFF_S + q] = T;
        if (seqrec) st[OFF_A + q] = N;
        st[OFF_H + q] = hn;
      }
      h_own = hn;
      GRU_STAMP(4)      // tanh, state update, LDS writes
      __syncthreads();
      GRU_STAMP(7)      // barrier 2
    }

    GRU_PHASE(4)        // the recurrent steps
    // ---- write the chunk back: hs rows and the saved r | u | c | T | h_prev (| N) of every step
    for (int i = tid; i < nch * (D / 4); i += 512) {
      const int s = i / (D / 4), c4 = i - s * (D / 4);
      const size_t r = row0 + t0 + s;
      *reinterpret_cast<float4 *>(p.hs + r * D + 4 * c4) = *reinterpret_cast<const float4 *>(&stage[s * ST + OFF_H + 4 * c4]);
      if (p.save) {
        float *sv = p.save + r * (nsave * D) + 4 * c4;
        const float *sr = &stage[s * ST + 4 * c4];
        *reinterpret_cast<float4 *>(sv) = *reinterpret_cast<const float4 *>(sr);                       // r
        *reinterpret_cast<float4 *>(sv + D) = *reinterpret_cast<const float4 *>(sr + D);               // u
        *reinterpret_cast<float4 *>(sv + 2 * D) = *reinterpret_cast<const float4 *>(sr + OFF_C);       // c
        *reinterpret_cast<float4 *>(sv + 3 * D) =
            plain ? make_float4(1.f, 1.f, 1.f, 1.f) : *reinterpret_cast<const float4 *>(sr + OFF_S);   // T
        *reinterpret_cast<float4 *>(sv + 4 * D) = *reinterpret_cast<const float4 *>(
            s > 0 ? &stage[(s - 1) * ST + OFF_H + 4 * c4] : &carry_s[4 * c4]);                         // h_prev
        if (seqrec) *reinterpret_cast<float4 *>(sv + 5 * D) = *reinterpret_cast<const float4 *>(sr + OFF_A);   // N
      }
    }
    __syncthreads();
    GRU_PHASE(5)        // write-back
    if (tid < D) carry_s[tid] = stage[(nch - 1) * ST + OFF_H + tid];
  }
  GRU_STAMP_DUMP(0)
  GRU_PHASE_DUMP(0)

  if (owner) p.short_out[(size_t)b * D + q] = (steps > 0) ? h_own : 0.f;
  // dynamic_rnn zero-fills dead steps
  for (int i = tid; i < (p.L - steps) * (D / 4); i += 512)
    *reinterpret_cast<float4 *>(p.hs + (row0 + steps) * D + 4 * (size_t)i) = make_float4(0.f, 0.f, 0.f, 0.f);
}

struct BwdArgs {
  const float *d_short, *d_hs, *x, *timelast;
  const int32_t *seq_len;
  const float *wh_g, *wh_c, *tvec, *save;
  int B, L, ldx;
  float *d_xproj, *rh, *d_xt, *d_tvec_partial;
  // d_kv role (workgroups blockIdx >= B): d_x [B L, D] += d_kv [B L, 256] . Wkv^T from the images of Wkv's
  // transpose (dkv_img); NULL = no role
  const float *d_kv;
  const uint16_t *dkv_img;
  float *d_x;
};

constexpr int DKV_N = 2 * D;                                    // the role is built for one decoder block: d_kv [R, 256]
constexpr int DKV_PITCH = DKV_N * 2 + 16;                       // bytes per staged row of one image
constexpr int DKV_IMG = stripe::ROWS * DKV_PITCH;
constexpr int DKV_ROLE_LDS = 3 * DKV_IMG + 8 * stripe::ROWS * stripe::T_PITCH * 4;        // 87,552 B (dynamic)

// 512 threads, one 32-row stripe: wave w owns output column block w & 3 over k-steps 8 (w >> 2) .. + 8 of the 16;
// the two halves of a column block meet in LDS and waves 0..3 add them onto d_x (16-byte read-modify-write)
__device__ __forceinline__ void dkv_role(const BwdArgs &p, int stripe_id, unsigned char *lds) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const long row0 = (long)stripe_id * stripe::ROWS;
  const int R = p.B * p.L;
  const int cb = w & 3, kh = w >> 2;
  // B fragments: rows n = 32 cb + r of Wkv, columns k = 128 kh + 16 s + 8 h .. + 8 -- pieces of the transpose's image
  Tri bw[8];
  {
    const uint16_t *base = p.dkv_img + ((size_t)(16 * kh + h) * D + 32 * cb + r) * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        bw[s].t[t] = *reinterpret_cast<const bf16x8 *>(base + (size_t)t * (D * DKV_N) + (size_t)s * (2 * D * 8));
  }
  {
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = i * 512 + tid, row = id >> 6, c4 = id & 63;
      v[i] = *reinterpret_cast<const f32x4 *>(p.d_kv + min(row0 + row, (long)R - 1) * DKV_N + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = i * 512 + tid, row = id >> 6, c4 = id & 63;
      const float x4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
      bf16x4 q[3];
      split_bf16::split4(x4, q);
      unsigned char *dst = lds + row * DKV_PITCH + c4 * 8;
#pragma unroll
      for (int t = 0; t < 3; ++t) *reinterpret_cast<bf16x4 *>(dst + t * DKV_IMG) = q[t];
    }
  }
  __syncthreads();
  f32x16 a0 = {0.f}, a1 = {0.f};
  {
    const unsigned char *ab = lds + r * DKV_PITCH + kh * (8 * 32) + h * 16;
#pragma unroll
    for (int s = 0; s < 8; s += 2) {
      Tri fa, fb;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        fa.t[t] = *reinterpret_cast<const bf16x8 *>(ab + t * DKV_IMG + s * 32);
        fb.t[t] = *reinterpret_cast<const bf16x8 *>(ab + t * DKV_IMG + (s + 1) * 32);
      }
      split_bf16::mfma6x2(fa, bw[s], a0, fb, bw[s + 1], a1);
    }
  }
  float *scratch = reinterpret_cast<float *>(lds + 3 * DKV_IMG) + w * (stripe::ROWS * stripe::T_PITCH);
  {
    const f32x16 acc = a0 + a1;
#pragma unroll
    for (int q = 0; q < 16; ++q) scratch[((q & 3) + 8 * (q >> 2) + 4 * h) * stripe::T_PITCH + r] = acc[q];
  }
  __syncthreads();
  if (w < 4) {
    const float *mine = scratch, *other = scratch + 4 * (stripe::ROWS * stripe::T_PITCH);
    const int gn = 32 * cb + 4 * (lane & 7);
    f32x4 t[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = 8 * i + (lane >> 3);
      t[i] = *reinterpret_cast<const f32x4 *>(mine + lr * stripe::T_PITCH + 4 * (lane & 7)) +
             *reinterpret_cast<const f32x4 *>(other + lr * stripe::T_PITCH + 4 * (lane & 7));
      c[i] = *reinterpret_cast<const f32x4 *>(p.d_x + min(row0 + lr, (long)R - 1) * D + gn);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = 8 * i + (lane >> 3);
      if (row0 + lr < R) *reinterpret_cast<f32x4 *>(p.d_x + (row0 + lr) * D + gn) = c[i] + t[i];
    }
  }
}

// Backward through time, cut by output column (the 2-barrier form; an LDS-staged variant like the forward's
// measured 52.0 us against this kernel's 48.3: profiles/r02_tagru_v4b_batched_staging.hip.txt): output k of both transposed products (d(r*h) = dcpre . Wc_h^T over
// 128 n, dh_prev += dgpre . Wg_h^T over 256 n) belongs to ONE lane (kp < 2 of octet o of wave w: k = 16 w + 2 o +
// kp), which also carries dh, the element-wise chain and the time-gate parameter gradients of that column in
// registers from step to step; the contraction index n is split over the octet's 8 lanes.
__global__ __launch_bounds__(512) void tagru_bwd_kernel(BwdArgs p) {
  __shared__ __attribute__((aligned(16))) float dc_s[D + 8];
  __shared__ __attribute__((aligned(16))) float dg_s[2 * D + 16];
  // dynamic: DKV_ROLE_LDS bytes when the d_kv role rides along (its staging; for the recurrence's own workgroups a
  // reservation that keeps any second workgroup off their CU), else nothing
  extern __shared__ __attribute__((aligned(16))) unsigned char role_lds[];
  if ((int)blockIdx.x >= p.B) {
    dkv_role(p, blockIdx.x - p.B, role_lds);
    return;
  }

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

  // zero-fill the dead steps of the outputs (contiguous rows: 16-byte stores)
  {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t dead = (size_t)(p.L - steps);
    float4 *dx4 = reinterpret_cast<float4 *>(p.d_xproj + (row0 + steps) * ldx);
    for (size_t i = tid; i < dead * (ldx / 4); i += 512) dx4[i] = z4;
    float4 *rh4 = reinterpret_cast<float4 *>(p.rh + (row0 + steps) * D);
    float4 *xt4 = reinterpret_cast<float4 *>(p.d_xt + (row0 + steps) * D);
    for (size_t i = tid; i < dead * (D / 4); i += 512) {
      rh4[i] = z4;
      xt4[i] = z4;
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
  return mtam_tagru_fwd_kv(xproj, x, timelast, seq_len, wh_g, wh_c, tvec, B, L, hs, short_out, save, nullptr, nullptr,
                           0, nullptr, nullptr, stream);
}

extern "C" int mtam_tagru_fwd_kv(const float *xproj, const float *x, const float *timelast,
                                 const int32_t *seq_len, const float *wh_g, const float *wh_c,
                                 const float *tvec, int B, int L, float *hs, float *short_out,
                                 float *save, const uint16_t *wkv_images, const float *bkv, int n_kv, float *kv_out,
                                 const float *gru_w_image, void *stream) {
  MTAM_CHECK_ARG(B > 0 && L > 0, "tagru_fwd: B and L must be positive");
  MTAM_CHECK_ARG(mtam_aligned16(gru_w_image), "tagru_fwd_kv: gru_w_image must be 16-byte aligned");
  MTAM_CHECK_ARG(!wkv_images || (bkv && kv_out && n_kv > 0 && n_kv % 32 == 0 && mtam_aligned16(wkv_images) &&
                                 mtam_aligned16(kv_out) && mtam_aligned16(x)),
                 "tagru_fwd_kv: the K/V role needs bkv, kv_out and n_kv a multiple of 32 (16-byte aligned operands)");
  MTAM_CHECK_ARG(xproj && x && timelast && seq_len && wh_g && wh_c && hs && short_out,
                 "tagru_fwd: null argument");        // tvec may be NULL: plain GRUCell
  MTAM_CHECK_ARG(mtam_aligned16(wh_g) && mtam_aligned16(wh_c), "tagru_fwd: weights must be 16-byte aligned");
  MTAM_CHECK_ARG(mtam_aligned16(xproj) && mtam_aligned16(hs) && mtam_aligned16(save),
                 "tagru_fwd: xproj, hs and save must be 16-byte aligned");
  FwdArgs a{xproj, x, timelast, seq_len, wh_g, wh_c, tvec, B, L, 3 * D, hs, short_out, save, wkv_images, bkv, kv_out,
            wkv_images ? n_kv : 0, gru_w_image};
  const int stripes = wkv_images ? (B * L + stripe::ROWS - 1) / stripe::ROWS : 0;
  hipLaunchKernelGGL(tagru_fwd_kernel, dim3(B + stripes), dim3(512), 0, static_cast<hipStream_t>(stream), a);
  MTAM_CHECK_LAUNCH("tagru_fwd");
  return MTAM_OK;
}

extern "C" int mtam_gru_weight_image_floats(void) { return GRU_WIMG_FLOATS; }

extern "C" int mtam_gru_weight_image_pos(int which, int k, int n) {
  if (which < 0 || which > 1 || k < 0 || k >= D || n < 0 || n >= (which ? D : 2 * D)) return -1;
  return gru_image::pos(which, k, n);
}

extern "C" int mtam_gru_weight_image(const float *wh_g, const float *wh_c, float *image, void *stream) {
  MTAM_CHECK_ARG(wh_g && wh_c && image && mtam_aligned16(image), "gru_weight_image: bad arguments");
  hipLaunchKernelGGL(gru_weight_image_kernel, dim3(D * 3 * D / 256), dim3(256), 0, static_cast<hipStream_t>(stream), wh_g,
                     wh_c, image);
  MTAM_CHECK_LAUNCH("gru_weight_image");
  return MTAM_OK;
}

extern "C" int mtam_tagru_bwd(const float *d_short, const float *d_hs, const float *x, const float *timelast,
                              const int32_t *seq_len, const float *wh_g, const float *wh_c,
                              const float *tvec, const float *save, int B, int L, float *d_xproj,
                              float *rh, float *d_xt, float *d_tvec_partial, void *stream) {
  return mtam_tagru_bwd_dkv(d_short, d_hs, x, timelast, seq_len, wh_g, wh_c, tvec, save, B, L, d_xproj, rh, d_xt,
                            d_tvec_partial, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int mtam_tagru_bwd_dkv(const float *d_short, const float *d_hs, const float *x, const float *timelast,
                                  const int32_t *seq_len, const float *wh_g, const float *wh_c,
                                  const float *tvec, const float *save, int B, int L, float *d_xproj,
                                  float *rh, float *d_xt, float *d_tvec_partial, const float *d_kv, int n_kv,
                                  const uint16_t *wkv_images_t, float *d_x, void *stream) {
  MTAM_CHECK_ARG(B > 0 && L > 0, "tagru_bwd: B and L must be positive");
  MTAM_CHECK_ARG(!d_kv || (n_kv == DKV_N && wkv_images_t && d_x && mtam_aligned16(d_kv) && mtam_aligned16(d_x) &&
                           mtam_aligned16(wkv_images_t)),
                 "tagru_bwd_dkv: the d_kv role is built for n_kv = %d (one decoder block), 16-byte aligned operands", DKV_N);
  MTAM_CHECK_ARG(d_short && x && timelast && seq_len && wh_g && wh_c && save && d_xproj && rh && d_xt &&
                     d_tvec_partial,
                 "tagru_bwd: null argument");        // tvec (plain GRUCell) and d_hs may be NULL
  MTAM_CHECK_ARG(mtam_aligned16(wh_g) && mtam_aligned16(wh_c), "tagru_bwd: weights must be 16-byte aligned");
  MTAM_CHECK_ARG(mtam_aligned16(save) && mtam_aligned16(d_xproj) && mtam_aligned16(rh) && mtam_aligned16(d_xt),
                 "tagru_bwd: save, d_xproj, rh and d_xt must be 16-byte aligned");
  BwdArgs a{d_short, d_hs, x, timelast, seq_len, wh_g, wh_c, tvec, save, B, L, 3 * D, d_xproj, rh, d_xt,
            d_tvec_partial, d_kv, wkv_images_t, d_x};
  const int stripes = d_kv ? (B * L + stripe::ROWS - 1) / stripe::ROWS : 0;
  if (d_kv) {
    static bool attr_set = false;
    if (!attr_set) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tagru_bwd_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, DKV_ROLE_LDS);
      MTAM_CHECK_ARG(e == hipSuccess, "tagru_bwd_dkv: cannot reserve %d bytes of LDS: %s", DKV_ROLE_LDS, hipGetErrorString(e));
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(tagru_bwd_kernel, dim3(B + stripes), dim3(512), d_kv ? DKV_ROLE_LDS : 0,
                     static_cast<hipStream_t>(stream), a);
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
  MTAM_CHECK_ARG(mtam_aligned16(xproj5) && mtam_aligned16(hs) && mtam_aligned16(save6),
                 "tagru_seqrec_fwd: xproj5, hs and save6 must be 16-byte aligned");
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
  MTAM_CHECK_ARG(mtam_aligned16(save6) && mtam_aligned16(d_xproj5) && mtam_aligned16(rh) && mtam_aligned16(d_xt),
                 "tagru_seqrec_bwd: save6, d_xproj5, rh and d_xt must be 16-byte aligned");
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
