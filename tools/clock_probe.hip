// Developer probe: shader clock actually held (s_memtime ticks per s_memrealtime 100 MHz tick).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long *out, int iters, float *sink) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) a = fmaf(a, b, 1e-7f);
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
  if (a == 123.456f) sink[0] = a;
}
int main() {
  unsigned long long *d, h[2 * 1024];
  float *sink;
  hipMalloc(&d, sizeof(h)); hipMalloc(&sink, 4);
  int grids[] = {1, 8, 128, 256, 1024};
  int iters[] = {2000, 20000, 200000, 2000000};
  for (int gi = 0; gi < 5; ++gi)
    for (int ii = 0; ii < 4; ++ii) {
      hipLaunchKernelGGL(spin, dim3(grids[gi]), dim3(256), 0, 0, d, iters[ii], sink);
      hipDeviceSynchronize();
      hipMemcpy(h, d, sizeof(unsigned long long) * 2 * grids[gi], hipMemcpyDeviceToHost);
      double mhz = 100.0 * (double)h[0] / (double)h[1];
      printf("blocks %4d iters %8d : %10llu shader ticks, %9llu x10ns -> %.0f MHz, %.1f cycles/fma\n", grids[gi], iters[ii], h[0], h[1], mhz, (double)h[0] / iters[ii]);
    }
  return 0;
}
