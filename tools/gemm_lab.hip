// Developer lab: where does a k-tile of mtam_gemm_f32 spend its cycles?  Includes the production
// source with stamps switched on (s_memtime per phase; one thread of one workgroup adds the phase
// lengths into a device array).  Stamps drain LDS reads, so read the SHARES, not the total.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_lab.hip mtamrecommender_amd/csrc/capi.hip -o tools/gemm_lab
#include <hip/hip_runtime.h>
__device__ unsigned long long g_stamp_sum[8];
#define GEMM_STAMP_DECL unsigned long long stamp_last_ = 0
#define GEMM_STAMP(i)                                                                       \
  {                                                                                         \
    unsigned long long t_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    if (blockIdx.x == 7 && blockIdx.y == 0 && threadIdx.x == 0) {                           \
      if ((i) > 0) g_stamp_sum[i] += t_ - stamp_last_;                                      \
      else { g_stamp_sum[0] += 1; if (stamp_last_) g_stamp_sum[5] += t_ - stamp_last_; }    \
    }                                                                                       \
    stamp_last_ = t_;                                                                       \
  }
#include "../mtamrecommender_amd/csrc/gemm_f32.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static void run(const char *name, int M, int N, int K, int ta, int tb, int epi, int split_k = 1) {
  float *A, *B, *C, *aux, *aux2, *bias;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMalloc(&aux, (size_t)M * N * 4)); CK(hipMalloc(&aux2, (size_t)M * N * 4)); CK(hipMalloc(&bias, (size_t)M * N * 4));
  CK(hipMemset(A, 0, (size_t)M * K * 4)); CK(hipMemset(B, 0, (size_t)N * K * 4)); CK(hipMemset(C, 0, (size_t)M * N * 4));
  CK(hipMemset(aux, 0, (size_t)M * N * 4)); CK(hipMemset(bias, 0, (size_t)M * N * 4));
  unsigned long long zero[8] = {0};
  for (int w = 0; w < 3; ++w)
    mtam_gemm_f32(ta, tb, M, N, K, A, ta ? M : K, B, tb ? K : N, C, N, epi, bias, aux, aux2, N, split_k, nullptr);
  CK(hipDeviceSynchronize());
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_sum), zero, sizeof(zero)));
  const int reps = 20;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int w = 0; w < reps; ++w)
    mtam_gemm_f32(ta, tb, M, N, K, A, ta ? M : K, B, tb ? K : N, C, N, epi, bias, aux, aux2, N, split_k, nullptr);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[8];
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp_sum), sizeof(h)));
  const double n = (double)h[0];
  // s_memtime counts at 100 MHz on gfx950?  print raw ticks per k-tile; the ratio is what matters
  // stamps 0..4 sit inside the k loop: per k-tile, cycles of  frag reads | MFMA issue | staging + prefetch | barrier | loop back
  printf("%-18s M=%d N=%d K=%d ta=%d tb=%d epi=0x%x split_k=%d: %.1f us per launch (stamped build); ktiles=%.0f | frag-read %.1f | mfma-issue %.1f | stage+prefetch %.1f | barrier %.1f | loop-back %.1f (cycles per k-tile; the split path books reads + MFMAs under mfma-issue)\n",
         name, M, N, K, ta, tb, epi, split_k, ms * 1000.f / reps, n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n);
  hipFree(A); hipFree(B); hipFree(C); hipFree(aux); hipFree(aux2); hipFree(bias);
}

int main() {
  const int X3 = MTAM_GEMM_SPLIT_BF16;
  run("dx accum2 NT", 6400, 128, 640, 0, 1, 7);
  run("dx accum2 NT x3", 6400, 128, 640, 0, 1, 7 | X3);
  run("d_ic NT", 6400, 256, 128, 0, 1, 0);
  run("d_ic NT x3", 6400, 256, 128, 0, 1, 0 | X3);
  run("dW TN atomic", 128, 384, 6400, 1, 0, 6, 16);
  run("dW TN atomic x3", 128, 384, 6400, 1, 0, 6 | X3, 16);
  run("xproj NN", 6400, 384, 128, 0, 0, 1);
  run("xproj NN x3", 6400, 384, 128, 0, 0, 1 | X3);
  return 0;
}
