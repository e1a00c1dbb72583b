// Developer lab: issue cost of v_fma_f32 against v_pk_fma_f32 on gfx950 with one and two waves per SIMD
// (the serial GRU's inner products are VALU work confined to one CU per sample).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/valu_lab.hip -o tools/valu_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(unsigned long long *out, int iters, float *sink) {
  float a[16];
  f32x2 p[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
  float b = 1.0001f, c = 1e-7f;
  f32x2 b2 = {b, b}, c2 = {c, c};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {            // 16 independent v_fma_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    } else if (KIND == 1) {     // 8 independent v_pk_fma_f32 (= 16 FMAs per lane)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(c2));
    } else if (KIND == 2) {     // 16 v_fma_f32 in 2 dependent chains (latency-bound form)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(b), "v"(c));
      }
    } else {                    // 8 v_pk_fma_f32 in 2 dependent chains
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(b2), "v"(c2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[1]) : "v"(b2), "v"(c2));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  if (s == 123.456f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  unsigned long long *d, h[16 * 128];
  float *sink;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&sink, 4);
  const char *names[4] = {"16 x v_fma_f32, independent", "8 x v_pk_fma_f32, independent", "16 x v_fma_f32, 2 chains",
                          "8 x v_pk_fma_f32, 2 chains"};
  const int iters = 20000;
  for (int kind = 0; kind < 4; ++kind)
    for (int threads = 256; threads <= 1024; threads *= 2) {
      for (int rep = 0; rep < 2; ++rep) {
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(128), dim3(threads), 0, 0, d, iters, sink);
        if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(128), dim3(threads), 0, 0, d, iters, sink);
        if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(128), dim3(threads), 0, 0, d, iters, sink);
        if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(128), dim3(threads), 0, 0, d, iters, sink);
        (void)hipDeviceSynchronize();
      }
      (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      const double cyc = (double)h[0] / iters;       // cycles per iteration (= 16 FMAs per lane) seen by wave 0
      printf("%-32s %d waves/SIMD: %6.1f cycles per 16 FMAs per lane per wave -> %5.1f FMA-lanes per cycle per SIMD\n",
             names[kind], threads / 256, cyc, 16.0 * 64.0 * (threads / 256) / cyc);
    }
  return 0;
}
