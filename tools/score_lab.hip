// Developer lab: where a slab iteration of the split-bf16 scoring kernels spends its time.  Builds csrc/score32.hip
// with in-kernel stamps (-DMTAM_SCORE_STAMPS: a diagnostic build, its run time is not quoted).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMTAM_SCORE_STAMPS tools/score_lab.hip mtamrecommender_amd/csrc/capi.hip -o tools/score_lab
//   tools/score_lab [V]
#include "../mtamrecommender_amd/csrc/score32.hip"
#include <vector>

int main(int argc, char **argv) {
  const int V = argc > 1 ? atoi(argv[1]) : 10000003, B = 128;
  float *E, *P, *partial, *lse, *ce, *d_pred, *dE, *sq;
  int32_t *tgt;
  (void)hipMalloc(&E, (size_t)V * D * 4); (void)hipMalloc(&dE, (size_t)V * D * 4); (void)hipMalloc(&P, B * D * 4);
  (void)hipMalloc(&partial, (size_t)mtam_score32_partials(B, V) * 4); (void)hipMalloc(&lse, B * 4);
  (void)hipMalloc(&ce, B * 4); (void)hipMalloc(&d_pred, B * D * 4); (void)hipMalloc(&sq, mtam_score32_sq_partials(V) * 4);
  (void)hipMalloc(&tgt, B * 4);
  {
    std::vector<float> h((size_t)1 << 24);
    srand(1);
    for (auto &v : h) v = 0.2165f * ((rand() % 2001) / 1000.f - 1.f);
    for (size_t off = 0; off < (size_t)V * D; off += h.size())
      (void)hipMemcpy(E + off, h.data(), std::min(h.size(), (size_t)V * D - off) * 4, hipMemcpyHostToDevice);
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
  float ms_l = 0.f, ms_b = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    if (mtam_score32_lse(E, P, tgt, B, V, partial, lse, ce, nullptr)) { printf("lse: %s\n", mtam_last_error()); return 1; }
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms_l, e0, e1);
    (void)hipEventRecord(e0, 0);
    if (mtam_score32_bwd(E, P, lse, tgt, B, V, 1.f / B, d_pred, dE, sq, nullptr)) { printf("bwd: %s\n", mtam_last_error()); return 1; }
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms_b, e0, e1);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
  printf("V = %d, B = %d (stamped build): lse pass %.3f ms, backward %.3f ms\n", V, B, ms_l, ms_b);
#ifdef MTAM_SCORE_STAMPS
  unsigned long long st[2][12][8];
  (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_score_stamps), sizeof(st));
  const int nslab = (V + 31) / 32;
  const char *ln[8] = {"rows landed (vmcnt)", "split + LDS writes", "barrier 1", "load issue + 48 MFMA", "max, exp, sum", "barrier 2", "-", "-"};
  const int lse_iters = (nslab + lse_grid_of(V) - 1) / lse_grid_of(V);
  printf("lse: timer ticks per slab by segment, middle workgroup, %d slabs\n", lse_iters);
  for (int s = 0; s < 6; ++s) {
    printf("  %-24s", ln[s]);
    for (int w = 0; w < 4; ++w) printf(" w%d %7.0f", w, (double)st[0][w][s] / lse_iters);
    printf("\n");
  }
  {
    double tot = 0;
    for (int s = 0; s < 6; ++s) tot += (double)st[0][0][s];
    printf("  wave 0: %.0f s_memtime ticks in %.0f s_memrealtime ticks (100 MHz): shader clock %.0f MHz\n", tot,
           (double)st[0][0][7], tot / (double)st[0][0][7] * 100.0);
  }
  const char *bn[8] = {"S scores / D d_pred / T dE MFMA", "S G / T dE epilogue", "S split + fetch", "barrier", "-", "-", "S: rows landed", "-"};
  const int bwd_iters = (nslab + grid_of(V) - 1) / grid_of(V);
  printf("backward: timer ticks per slab by segment, middle workgroup, %d slabs (waves 0-3 S, 4-7 D, 8-11 T)\n", bwd_iters);
  for (int s = 0; s < 7; ++s) {
    if (bn[s][0] == '-') continue;
    printf("  %-38s", bn[s]);
    for (int w = 0; w < 12; w += 1) printf(" %5.0f", (double)st[1][w][s] / bwd_iters);
    printf("\n");
  }
  {
    double tot = 0;
    for (int s = 0; s < 7; ++s) tot += (double)st[1][0][s];
    printf("  wave 0: %.0f s_memtime ticks in %.0f s_memrealtime ticks (100 MHz): shader clock %.0f MHz\n", tot,
           (double)st[1][0][7], tot / (double)st[1][0][7] * 100.0);
  }
  {   // the launch as the workgroups saw it: when each started and ended, and where it ran
    static unsigned long long wg[2][4096][3];
    (void)hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_score_wg), sizeof(wg));
    for (int k = 0; k < 2; ++k) {
      const int G = std::min(4096, k ? grid_of(V) : lse_grid_of(V));
      unsigned long long t0 = ~0ull, t1 = 0;
      double busy = 0, dmin = 1e30, dmax = 0;
      for (int i = 0; i < G; ++i) {
        t0 = std::min(t0, wg[k][i][0]); t1 = std::max(t1, wg[k][i][1]);
        const double d = (double)(wg[k][i][1] - wg[k][i][0]);
        busy += d; dmin = std::min(dmin, d); dmax = std::max(dmax, d);
      }
      printf("%s: %d workgroups, first start to last end %.1f us; a workgroup's main loop lasts %.1f us on average "
             "(%.1f .. %.1f): %.0f in flight on average\n", k ? "backward" : "lse", G, (t1 - t0) * 0.01, busy / G * 0.01,
             dmin * 0.01, dmax * 0.01, busy / (double)(t1 - t0));
      // start times by decile, and workgroups per (XCC, SE, SH, CU)
      int cnt[8][256] = {};
      for (int i = 0; i < G; ++i) {
        const unsigned hw = (unsigned)wg[k][i][2], xcc = (unsigned)(wg[k][i][2] >> 32) & 7;
        cnt[xcc][(hw >> 8) & 0xff]++;
      }
      int cus = 0, cmin = 1 << 30, cmax = 0;
      for (int x = 0; x < 8; ++x)
        for (int c = 0; c < 256; ++c)
          if (cnt[x][c]) { ++cus; cmin = std::min(cmin, cnt[x][c]); cmax = std::max(cmax, cnt[x][c]); }
      printf("  ran on %d distinct (XCC, SE/SH/CU) places, %d .. %d workgroups each\n", cus, cmin, cmax);
      printf("  start offsets (us) of workgroups 0, G/8, 2G/8, ..:");
      for (int j = 0; j < 8; ++j) printf(" %.0f", (wg[k][j * G / 8][0] - t0) * 0.01);
      printf("\n");
    }
  }
#endif
  return 0;
}
