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

static void run(const char *name, int M, int N, int K, int ta, int tb, int epi) {
  float *A, *B, *C, *aux, *aux2, *bias;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMalloc(&aux, (size_t)M * N * 4)); CK(hipMalloc(&aux2, (size_t)M * N * 4)); CK(hipMalloc(&bias, (size_t)M * N * 4));
  CK(hipMemset(A, 0, (size_t)M * K * 4)); CK(hipMemset(B, 0, (size_t)N * K * 4)); CK(hipMemset(C, 0, (size_t)M * N * 4));
  CK(hipMemset(aux, 0, (size_t)M * N * 4)); CK(hipMemset(bias, 0, (size_t)M * N * 4));
  unsigned long long zero[8] = {0};
  for (int w = 0; w < 3; ++w)
    mtam_gemm_f32(ta, tb, M, N, K, A, ta ? M : K, B, tb ? K : N, C, N, epi, bias, aux, aux2, N, 1, nullptr);
  CK(hipDeviceSynchronize());
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_sum), zero, sizeof(zero)));
  const int reps = 20;
  for (int w = 0; w < reps; ++w)
    mtam_gemm_f32(ta, tb, M, N, K, A, ta ? M : K, B, tb ? K : N, C, N, epi, bias, aux, aux2, N, 1, nullptr);
  CK(hipDeviceSynchronize());
  unsigned long long h[8];
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp_sum), sizeof(h)));
  const double n = (double)h[0];
  // s_memtime counts at 100 MHz on gfx950?  print raw ticks per k-tile; the ratio is what matters
  // stamps 0..4 sit inside the k loop: per k-tile, cycles of  frag reads | MFMA issue | staging + prefetch | barrier | loop back
  printf("%-18s M=%d N=%d K=%d ta=%d tb=%d epi=%d: ktiles=%.0f | frag-read %.1f | mfma-issue %.1f | stage+prefetch %.1f | barrier %.1f | loop-back %.1f (cycles per k-tile)\n",
         name, M, N, K, ta, tb, epi, n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n);
  hipFree(A); hipFree(B); hipFree(C); hipFree(aux); hipFree(aux2); hipFree(bias);
}

int main() {
  run("tall K=32", 6400, 128, 32, 0, 0, 0);
  run("tall K=32 NT", 6400, 128, 32, 0, 1, 0);
  run("tall K=512", 6400, 128, 512, 0, 0, 0);
  run("tall K=512 NT", 6400, 128, 512, 0, 1, 0);
  run("dense4emb", 6400, 128, 256, 0, 0, 3);
  run("dx accum_mask NT", 6400, 128, 384, 0, 1, 7);
  run("xproj", 6400, 384, 128, 0, 0, 1);
  run("big square", 4096, 4096, 4096, 0, 0, 0);
  return 0;
}
