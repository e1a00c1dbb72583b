// Developer lab (timing only, no product code): the forward recurrence's step as a 4-wave workgroup -- ONE wave per SIMD,
// an octet owns FOUR hidden units (192 weights per lane) -- against the product's 8 waves / two units per octet, with the
// same instruction mix: LDS read of the state (16 floats per lane), gate FMAs, octet reduce-scatter, sigmoid, r*h
// through LDS, barrier, candidate FMAs, reduce, tanh, state update through LDS, barrier.  Prints cycles per step.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gru4_lab.hip -o tools/gru4_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_SHL4 = 0x104, DPP_HALF_MIRROR = 0x141;
__device__ __forceinline__ float fast_sig(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x)); }
constexpr int D = 128;
__device__ __forceinline__ int padpos(int i) { return i + 4 * (i >> 6); }

// UNITS hidden units per octet: 2 (512 threads, the product's layout) or 4 (256 threads)
template <int UNITS, int EXTRAS>
__global__ __launch_bounds__(1024 / UNITS) void step_kernel(const float *__restrict__ w, const float *__restrict__ inp, int steps,
                                                            float *__restrict__ out, unsigned long long *__restrict__ cyc) {
  __shared__ __attribute__((aligned(16))) float h_s[D + 8], rh_s[D + 8];
  __shared__ __attribute__((aligned(16))) float stage[50 * 4 * D];
  const int tid = threadIdx.x, lane = tid & 63, kp = lane & 7, oct = tid >> 3;
  f32x2 wg[UNITS * 2][8], wc[UNITS][8];
#pragma unroll
  for (int c = 0; c < UNITS * 2; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) wg[c][k] = f32x2{w[(c * 8 + k) * 1024 + tid], w[(c * 8 + k) * 1024 + 512 + tid]};
#pragma unroll
  for (int c = 0; c < UNITS; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) wc[c][k] = f32x2{w[(40 + c * 8 + k) * 1024 + tid], w[(40 + c * 8 + k) * 1024 + 512 + tid]};
  // (keep them in registers: without this the compiler re-loads some of the 192 from global memory inside the loop)
#pragma unroll
  for (int c = 0; c < UNITS * 2; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(wg[c][k]));
#pragma unroll
  for (int c = 0; c < UNITS; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(wc[c][k]));
  for (int i = tid; i < 50 * 4 * D; i += blockDim.x) stage[i] = inp[(size_t)blockIdx.x * 50 * 4 * D + i] * 0.01f;
  if (tid < D + 8) { h_s[tid] = 0.f; rh_s[tid] = 0.f; }
  __syncthreads();
  const int vpos = 16 * kp + 4 * (kp >> 2);
  // the lane's unit: UNITS == 2: lanes kp 0,1 -> unit 0 (r, u), 2,3 -> unit 1; UNITS == 4: lane kp < 4 owns unit kp (r there, u on kp + 4)
  const int q = UNITS * oct + (UNITS == 2 ? ((kp >> 1) & 1) : (kp & 3));
  float h_own = 0.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) {
    const float *st = stage + s * 4 * D;
    float4 hq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hq[i] = *reinterpret_cast<const float4 *>(&h_s[vpos + 4 * i]);
    const float in_g = st[(kp & 1 ? D : 0) + q], in_c = st[2 * D + q], in_t = st[3 * D + q];
    float T;
    if (EXTRAS >= 1) {
      const float in_s = st[3 * D + ((q + 64) & 127)];
      T = fast_sig(fmaf(0.7f, fmaxf(fmaf(h_own, 0.3f, in_t), 0.f), in_s));      // the product's time gate
    } else {
      T = fast_sig(in_t + h_own * 0.3f);
    }
    f32x2 a[UNITS * 2];
#pragma unroll
    for (int c = 0; c < UNITS * 2; ++c) a[c] = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 lo = {hq[i].x, hq[i].y}, hi = {hq[i].z, hq[i].w};
#pragma unroll
      for (int c = 0; c < UNITS * 2; ++c) a[c] = pk_fma(lo, wg[c][2 * i], a[c]);
#pragma unroll
      for (int c = 0; c < UNITS * 2; ++c) a[c] = pk_fma(hi, wg[c][2 * i + 1], a[c]);
    }
    float gsum;
    if (UNITS == 2) {
      const bool b0 = lane & 1, b1 = lane & 2;
      const float s0 = a[0].x + a[0].y, s1 = a[1].x + a[1].y, s2 = a[2].x + a[2].y, s3 = a[3].x + a[3].y;
      float keepA = b0 ? s2 : s0, sendA = b0 ? s0 : s2, keepB = b0 ? s3 : s1, sendB = b0 ? s1 : s3;
      keepA += dpp_f<DPP_XOR1>(sendA); keepB += dpp_f<DPP_XOR1>(sendB);
      float keep = b1 ? keepB : keepA;
      const float send = b1 ? keepA : keepB;
      keep += dpp_f<DPP_XOR2>(send);
      keep += dpp_f<DPP_SHL4>(keep);
      gsum = keep;
    } else {
      // 8 sums (r of units 0..3, u of units 0..3) -> lane kp < 4: r of unit kp, lane kp >= 4: u of unit kp - 4
      float sv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) sv[c] = a[c].x + a[c].y;
      const bool up = kp >= 4, b1 = kp & 2, b0 = kp & 1;
      float k4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) k4[j] = (up ? sv[4 + j] : sv[j]) + dpp_f<DPP_HALF_MIRROR>(up ? sv[j] : sv[4 + j]);
      // (lane kp's partner through the mirror is 7 - kp: the unit order of the upper half is reversed: bookkeeping only)
      float k2a = b1 ? k4[2] : k4[0], s2a = b1 ? k4[0] : k4[2], k2b = b1 ? k4[3] : k4[1], s2b = b1 ? k4[1] : k4[3];
      k2a += dpp_f<DPP_XOR2>(s2a); k2b += dpp_f<DPP_XOR2>(s2b);
      float k1 = b0 ? k2b : k2a;
      const float s1 = b0 ? k2a : k2b;
      k1 += dpp_f<DPP_XOR1>(s1);
      gsum = k1;
    }
    const float sg = fast_sig(gsum + in_g);
    const float u = UNITS == 2 ? dpp_f<DPP_XOR1>(sg) : dpp_f<DPP_HALF_MIRROR>(sg);
    const bool owner = UNITS == 2 ? (kp == 0 || kp == 2) : (kp < 4);
    if (EXTRAS >= 2 && kp < 4) stage[s * 4 * D + (kp & 1 ? D : 0) + q] = sg;          // saved r | u
    if (owner) rh_s[padpos(q)] = sg * h_own;
    __syncthreads();
    float4 rq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rq[i] = *reinterpret_cast<const float4 *>(&rh_s[vpos + 4 * i]);
    f32x2 cacc[UNITS];
#pragma unroll
    for (int c = 0; c < UNITS; ++c) cacc[c] = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 lo = {rq[i].x, rq[i].y}, hi = {rq[i].z, rq[i].w};
#pragma unroll
      for (int c = 0; c < UNITS; ++c) cacc[c] = pk_fma(lo, wc[c][2 * i], cacc[c]);
#pragma unroll
      for (int c = 0; c < UNITS; ++c) cacc[c] = pk_fma(hi, wc[c][2 * i + 1], cacc[c]);
    }
    float csum;
    if (UNITS == 2) {
      const bool b1 = lane & 2;
      const float s0 = cacc[0].x + cacc[0].y, s1 = cacc[1].x + cacc[1].y;
      float keep = b1 ? s1 : s0;
      keep += dpp_f<DPP_XOR2>(b1 ? s0 : s1);
      keep += dpp_f<DPP_XOR1>(keep);
      keep += dpp_f<DPP_SHL4>(keep);
      csum = keep;
    } else {
      const bool b0 = lane & 1, b1 = lane & 2;
      const float s0 = cacc[0].x + cacc[0].y, s1 = cacc[1].x + cacc[1].y, s2 = cacc[2].x + cacc[2].y, s3 = cacc[3].x + cacc[3].y;
      float keepA = b0 ? s2 : s0, sendA = b0 ? s0 : s2, keepB = b0 ? s3 : s1, sendB = b0 ? s1 : s3;
      keepA += dpp_f<DPP_XOR1>(sendA); keepB += dpp_f<DPP_XOR1>(sendB);
      float keep = b1 ? keepB : keepA;
      keep += dpp_f<DPP_XOR2>(b1 ? keepA : keepB);
      keep += dpp_f<DPP_SHL4>(keep);
      csum = keep;
    }
    const float c = fmaf(2.f, fast_sig(csum + in_c), -1.f);
    const float hn = u * h_own + (1.f - u) * c * T;
    if (owner) {
      h_s[padpos(q)] = hn;
      stage[s * 4 * D + 2 * D + q] = c;
      if (EXTRAS >= 2) { stage[s * 4 * D + 3 * D + q] = T; stage[s * 4 * D + 3 * D + ((q + 64) & 127)] = hn; }
    }
    h_own = hn;
    __syncthreads();
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[(size_t)blockIdx.x * blockDim.x + tid] = h_own;
}

template <int UNITS, int EXTRAS>
void run(const char *name) {
  const int B = 128, steps = 49;
  float *w, *inp, *out;
  unsigned long long *cyc;
  hipMalloc(&w, 80 * 1024 * 4); hipMalloc(&inp, (size_t)B * 50 * 4 * D * 4); hipMalloc(&out, B * 512 * 4); hipMalloc(&cyc, B * 8);
  std::vector<float> hw(80 * 1024), hi((size_t)B * 50 * 4 * D);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.05f * ((int)(i * 2654435761u % 2001) / 1000.f - 1.f);
  for (size_t i = 0; i < hi.size(); ++i) hi[i] = (int)(i * 40503u % 2001) / 1000.f - 1.f;
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(inp, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((step_kernel<UNITS, EXTRAS>), dim3(B), dim3(1024 / UNITS), 0, 0, w, inp, steps, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((step_kernel<UNITS, EXTRAS>), dim3(B), dim3(1024 / UNITS), 0, 0, w, inp, steps, out, cyc);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hc(B);
  hipMemcpy(hc.data(), cyc, B * 8, hipMemcpyDeviceToHost);
  printf("%s: %.1f us per launch, workgroup 0: %.0f cycles per step (s_memtime)\n", name, ms * 20.f, (double)hc[0] / steps);
}

int main() {
  run<2, 0>("8 waves, 2 units per octet");
  run<4, 0>("4 waves, 4 units per octet");
  run<2, 1>("8 waves, 2 units, + the product's time gate");
  run<2, 2>("8 waves, 2 units, + time gate + saved r|u, T, h rows to LDS");
  run<4, 2>("4 waves, 4 units, + time gate + saved rows");
  return 0;
}
