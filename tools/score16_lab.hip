// Developer lab: csrc/score16.hip (included as is) timed at a large catalog, optionally with parts of the
// backward kernel cut out (-DS16_LAB_NO_DE, -DS16_LAB_NO_DE_STORE, -DS16_LAB_NO_DPRED, -DS16_LAB_NO_EXP,
// -DS16_BWD_WAVES_PER_EU=2).  Results of the cut variants are wrong by construction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../mtamrecommender_amd/csrc/score16.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv) {
  const int V = argc > 1 ? atoi(argv[1]) : 10000003, B = 128;
  float *E, *P, *lse, *ce, *partial, *d_pred, *dE, *sq; uint16_t *E16, *P16; int32_t *tgt;
  CK(hipMalloc(&E, (size_t)V * 512)); CK(hipMalloc(&E16, (size_t)V * 256)); CK(hipMalloc(&dE, (size_t)V * 512));
  CK(hipMalloc(&P, B * 512)); CK(hipMalloc(&P16, 128 * 256)); CK(hipMalloc(&lse, B * 4)); CK(hipMalloc(&ce, B * 4));
  CK(hipMalloc(&partial, (size_t)mtam_score16_partials(B, V) * 4)); CK(hipMalloc(&sq, (size_t)mtam_score16_sq_partials(V) * 4));
  CK(hipMalloc(&d_pred, B * 512)); CK(hipMalloc(&tgt, B * 4));
  CK(hipMemset(E, 0x3c, (size_t)V * 512)); CK(hipMemset(P, 0x3d, B * 512)); CK(hipMemset(tgt, 0, B * 4)); CK(hipMemset(d_pred, 0, B * 512));
  hipStream_t st; CK(hipStreamCreate(&st));
  if (mtam_f32_to_bf16(E, (size_t)V * 128, E16, (size_t)V * 128, st) || mtam_f32_to_bf16(P, B * 128, P16, 128 * 128, st)) { printf("%s\n", mtam_last_error()); return 1; }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int which = 0; which < 2; ++which) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(a, st));
      int rc = which == 0 ? mtam_score16_lse(E16, P16, tgt, B, V, partial, lse, ce, st)
                          : mtam_score16_bwd(E16, P16, lse, tgt, B, V, 1.0f / B, d_pred, dE, sq, st);
      if (rc) { printf("%s\n", mtam_last_error()); return 1; }
      CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = which == 0 ? (double)V * 256 : (double)V * 768;
    printf("%-12s V=%d  %8.3f ms  %6.0f GB/s\n", which == 0 ? "score16_lse" : "score16_bwd", V, best, bytes / best / 1e6);
  }
  return 0;
}
