// Developer lab: phase times of the one-launch small-catalog scoring kernel (x3::train_small_kernel), stamped build.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMTAM_SMALL_STAMPS tools/score_small_lab.hip mtamrecommender_amd/csrc/capi.hip -o tools/score_small_lab
//   tools/score_small_lab [V] [B]
#include "../mtamrecommender_amd/csrc/score32.hip"
#include <vector>

int main(int argc, char **argv) {
  const int V = argc > 1 ? atoi(argv[1]) : 3709, B = argc > 2 ? atoi(argv[2]) : 128;
  float *E, *P, *work, *lse, *ce, *d_pred, *dE, *sq;
  int32_t *tgt;
  const long nw = mtam_score32_train_work_floats(B, V);
  printf("V = %d, B = %d, fused %d, work %ld floats\n", V, B, mtam_score32_train_is_fused(B, V), nw);
  (void)hipMalloc(&E, (size_t)V * D * 4); (void)hipMalloc(&dE, (size_t)V * D * 4); (void)hipMalloc(&P, B * D * 4);
  (void)hipMalloc(&work, nw * 4); if (mtam_score32_train_work_init(work, nw, B, V, nullptr)) { printf("init: %s\n", mtam_last_error()); return 1; } (void)hipMalloc(&lse, B * 4);
  (void)hipMalloc(&ce, B * 4); (void)hipMalloc(&d_pred, B * D * 4); (void)hipMalloc(&sq, mtam_score32_sq_partials(V) * 4);
  (void)hipMalloc(&tgt, B * 4);
  {
    std::vector<float> h((size_t)V * D);
    srand(1);
    for (auto &v : h) v = 0.2165f * ((rand() % 2001) / 1000.f - 1.f);
    (void)hipMemcpy(E, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hp(B * D);
    for (auto &v : hp) v = (rand() % 2001) / 1000.f - 1.f;
    (void)hipMemcpy(P, hp.data(), B * D * 4, hipMemcpyHostToDevice);
    std::vector<int32_t> ht(B);
    for (auto &v : ht) v = rand() % V;
    (void)hipMemcpy(tgt, ht.data(), B * 4, hipMemcpyHostToDevice);
  }
  (void)hipMemset(d_pred, 0, B * D * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0.f;
  const int reps = 200;
  for (int round = 0; round < 2; ++round) {
    (void)hipEventRecord(e0, 0);
    for (int rep = 0; rep < reps; ++rep)
      if (mtam_score32_train(E, P, tgt, B, V, 1.f / B, work, nw, lse, ce, d_pred, dE, sq, mtam_score32_sq_partials(V), nullptr)) {
        printf("train: %s\n", mtam_last_error());
        return 1;
      }
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
  printf("one call: %.2f us (back to back, stamped build)\n", ms * 1000.f / reps);
#ifdef MTAM_SMALL_STAMPS
  unsigned long long st[3][16];
  (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_small_stamps), sizeof(st));
  const char *names[3] = {"S", "D", "T"};
  const char *pts[10] = {"start", "slab staged", "scores + pair stored", "-", "pairs folded", "G^T image",
                         "products + stores issued", "-", "d_pred reduced", "(D) MFMAs done"};
  for (int r = 0; r < 3; ++r) {
    printf("%s waves (us since start):", names[r]);
    for (int i = 0; i < 10; ++i)
      if (st[r][i]) printf("  %s %.2f", pts[i], (double)(st[r][i] - st[r][0]) / 100.0);
    printf("\n");
  }
#endif
  return 0;
}
