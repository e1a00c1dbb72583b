// Developer lab: issue distance of DEPENDENT against independent v_mfma_f32_32x32x16_bf16 on gfx950 (the split-bf16
// scoring kernels accumulate six products per operand pair into one tile), one and two waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_lab.hip -o tools/mfma_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS, bool RANDOM>
__global__ void k(unsigned long long *out, int iters, float *sink) {
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (RANDOM) {      // operands with every bit toggling from lane to lane and element to element
      unsigned x = (threadIdx.x * 2654435761u) ^ (i * 40503u) ^ (blockIdx.x * 97u);
      x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
      a[i] = (__bf16)(((int)(x & 0xffff) - 32768) * 3.1e-5f);
      b[i] = (__bf16)(((int)(x >> 16) - 32768) * 3.1e-5f);
    } else {
      a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.f / (1 + i));
    }
  }
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = f32x16{0.f};
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % CHAINS], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 123.456f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) { out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0; out[blockIdx.x * 16 + 8 + (threadIdx.x >> 6)] = r1 - r0; }
}


typedef float f32x4 __attribute__((ext_vector_type(4)));
// the 16x16x32 form (4 passes = 16 cycles, 4 accumulator registers per tile) on random operands, 4 chains
__global__ void k16(unsigned long long *out, int iters, float *sink) {
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    unsigned x = (threadIdx.x * 2654435761u) ^ (i * 40503u) ^ (blockIdx.x * 97u);
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    a[i] = (__bf16)(((int)(x & 0xffff) - 32768) * 3.1e-5f);
    b[i] = (__bf16)(((int)(x >> 16) - 32768) * 3.1e-5f);
  }
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 123.456f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) { out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0; out[blockIdx.x * 16 + 8 + (threadIdx.x >> 6)] = r1 - r0; }
}

int main() {
  unsigned long long *d, h[16 * 128];
  float *sink;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&sink, 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int chains = 1; chains <= 4; ++chains)
    for (int threads = 256; threads <= 512; threads *= 2) {
      float ms = 0.f;
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (chains == 1) hipLaunchKernelGGL((k<1, false>), dim3(256), dim3(threads), 0, 0, d, iters, sink);
        if (chains == 2) hipLaunchKernelGGL((k<2, false>), dim3(256), dim3(threads), 0, 0, d, iters, sink);
        if (chains == 3) hipLaunchKernelGGL((k<2, true>), dim3(256), dim3(threads), 0, 0, d, iters * 10, sink);
        if (chains == 4) hipLaunchKernelGGL((k<2, true>), dim3(256), dim3(threads), 0, 0, d, iters * 100, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
      }
      (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      const int its = chains == 3 ? iters * 10 : chains == 4 ? iters * 100 : iters;
      const double per = (double)h[0] / its / 12.0;
      const double tf = 2.0 * 32 * 32 * 16 * 12.0 * its * (threads / 64) * 256 / (ms * 1e-3) / 1e12;
      const char *what[5] = {"", "1 chain, constant operands", "2 chains, constant operands", "2 chains, random operands, 10x longer",
                             "2 chains, random operands, 100x longer"};
      printf("%-40s %d wave(s)/SIMD: %6.1f s_memtime ticks per MFMA per wave; s_memtime / s_memrealtime = %.2f (x 100 MHz = shader clock); kernel %.3f ms -> %.0f TFLOP/s\n",
             what[chains], threads / 256, per, (double)h[0] / (double)h[8], ms, tf);
    }
  for (int threads = 256; threads <= 512; threads *= 2) {
    float ms = 0.f;
    const int its = iters * 100;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k16, dim3(256), dim3(threads), 0, 0, d, its, sink);
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const double tf = 2.0 * 16 * 16 * 32 * 24.0 * its * (threads / 64) * 256 / (ms * 1e-3) / 1e12;
    printf("%-40s %d wave(s)/SIMD: %6.1f s_memtime ticks per MFMA per wave; s_memtime / s_memrealtime = %.2f (x 100 MHz = shader clock); kernel %.3f ms -> %.0f TFLOP/s\n",
           "16x16x32, 4 chains, random operands, 100x longer", threads / 256, (double)h[0] / its / 24.0, (double)h[0] / (double)h[8], ms, tf);
  }
  return 0;
}
