// Developer lab: where a step of the serial GRU kernels spends its cycles.  Builds csrc/tagru.hip with in-kernel
// stamps (-DMTAM_GRU_STAMPS: a diagnostic build, its run time is not quoted) and, in the same binary, times the
// un-stamped entry points of the product library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMTAM_GRU_STAMPS tools/gru_lab.hip mtamrecommender_amd/csrc/capi.hip -o tools/gru_lab
#include "../mtamrecommender_amd/csrc/tagru.hip"
#include <stdlib.h>
#include <algorithm>
#include <vector>

int main() {
  const int B = 128, L = 50, R = B * L;
  std::vector<float> h_x(R * D), h_xp((size_t)R * 3 * D), h_tl(R), h_wg(D * 2 * D), h_wc(D * D), h_tv(8 * D), h_ds(B * D);
  srand(1);
  auto rnd = [](float s) { return s * ((rand() % 2001) / 1000.f - 1.f); };
  for (auto &v : h_x) v = rnd(0.3f);
  for (auto &v : h_xp) v = rnd(0.3f);
  for (auto &v : h_tl) v = (float)(rand() % 48);
  for (auto &v : h_wg) v = rnd(0.08f);
  for (auto &v : h_wc) v = rnd(0.08f);
  for (auto &v : h_tv) v = rnd(0.1f);
  for (auto &v : h_ds) v = rnd(0.1f);
  std::vector<int32_t> h_sl(B, L);
  float *x, *xp, *tl, *wg, *wc, *tv, *hs, *sh, *save, *ds, *dxp, *rh, *dxt, *dtv;
  int32_t *sl;
  hipMalloc(&x, R * D * 4); hipMalloc(&xp, (size_t)R * 3 * D * 4); hipMalloc(&tl, R * 4); hipMalloc(&wg, D * 2 * D * 4);
  hipMalloc(&wc, D * D * 4); hipMalloc(&tv, 8 * D * 4); hipMalloc(&hs, R * D * 4); hipMalloc(&sh, B * D * 4);
  hipMalloc(&save, (size_t)R * 5 * D * 4); hipMalloc(&sl, B * 4); hipMalloc(&ds, B * D * 4);
  hipMalloc(&dxp, (size_t)R * 3 * D * 4); hipMalloc(&rh, R * D * 4); hipMalloc(&dxt, R * D * 4); hipMalloc(&dtv, B * 8 * D * 4);
  hipMemcpy(x, h_x.data(), R * D * 4, hipMemcpyHostToDevice);
  hipMemcpy(xp, h_xp.data(), (size_t)R * 3 * D * 4, hipMemcpyHostToDevice);
  hipMemcpy(tl, h_tl.data(), R * 4, hipMemcpyHostToDevice);
  hipMemcpy(wg, h_wg.data(), D * 2 * D * 4, hipMemcpyHostToDevice);
  hipMemcpy(wc, h_wc.data(), D * D * 4, hipMemcpyHostToDevice);
  hipMemcpy(tv, h_tv.data(), 8 * D * 4, hipMemcpyHostToDevice);
  hipMemcpy(sl, h_sl.data(), B * 4, hipMemcpyHostToDevice);
  hipMemcpy(ds, h_ds.data(), B * D * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 5; ++i) {
      mtam_tagru_fwd(xp, x, tl, sl, wg, wc, tv, B, L, hs, sh, save, nullptr);
      mtam_tagru_bwd(ds, nullptr, x, tl, sl, wg, wc, tv, save, B, L, dxp, rh, dxt, dtv, nullptr);
    }
    hipDeviceSynchronize();
    float ms_f, ms_b;
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) mtam_tagru_fwd(xp, x, tl, sl, wg, wc, tv, B, L, hs, sh, save, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms_f, e0, e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) mtam_tagru_bwd(ds, nullptr, x, tl, sl, wg, wc, tv, save, B, L, dxp, rh, dxt, dtv, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms_b, e0, e1);
    printf("fwd %.1f us, bwd %.1f us per launch (B=%d, L=%d)\n", ms_f * 20.f, ms_b * 20.f, B, L);
  }
#ifdef MTAM_GRU_STAMPS
  {   // 40 forward launches back to back: when the first workgroup of each started and the last one ended
    static unsigned long long span[2][64][2];
    for (int i = 0; i < 64; ++i) { span[0][i][0] = span[1][i][0] = ~0ull; span[0][i][1] = span[1][i][1] = 0; }
    unsigned int zero2[2] = {0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_gru_span), span, sizeof(span));
    hipMemcpyToSymbol(HIP_SYMBOL(g_gru_launch), zero2, sizeof(zero2));
    for (int i = 0; i < 40; ++i) mtam_tagru_fwd(xp, x, tl, sl, wg, wc, tv, B, L, hs, sh, save, nullptr);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(span, HIP_SYMBOL(g_gru_span), sizeof(span));
    printf("forward, 40 launches back to back: [workgroups alive us | gap to the next launch's first workgroup us]\n ");
    for (int i = 20; i < 30; ++i)
      printf(" [%.1f | %.1f]", (span[0][i][1] - span[0][i][0]) * 0.01, (span[0][i + 1][0] - span[0][i][1]) * 0.01);
    printf("\n");
  }
#endif
  {   // the forward without its saved-activation writes (evaluation's form): 3.3 MB written instead of 19.7
    for (int i = 0; i < 5; ++i) mtam_tagru_fwd(xp, x, tl, sl, wg, wc, tv, B, L, hs, sh, nullptr, nullptr);
    hipDeviceSynchronize();
    float ms_f;
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) mtam_tagru_fwd(xp, x, tl, sl, wg, wc, tv, B, L, hs, sh, nullptr, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms_f, e0, e1);
    printf("fwd without the save writes: %.1f us per launch\n", ms_f * 20.f);
  }
  {   // the launches' fixed part: the same kernels on sequences of length 1 (no recurrent step at all)
    std::vector<int32_t> one(B, 1);
    int32_t *sl1;
    hipMalloc(&sl1, B * 4);
    hipMemcpy(sl1, one.data(), B * 4, hipMemcpyHostToDevice);
    for (int len : {1, 13, 26}) {
      std::vector<int32_t> v(B, len);
      hipMemcpy(sl1, v.data(), B * 4, hipMemcpyHostToDevice);
      for (int i = 0; i < 5; ++i) mtam_tagru_fwd(xp, x, tl, sl1, wg, wc, tv, B, L, hs, sh, save, nullptr);
      hipDeviceSynchronize();
      float ms_f, ms_b;
      hipEventRecord(e0, 0);
      for (int i = 0; i < 50; ++i) mtam_tagru_fwd(xp, x, tl, sl1, wg, wc, tv, B, L, hs, sh, save, nullptr);
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms_f, e0, e1);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 50; ++i) mtam_tagru_bwd(ds, nullptr, x, tl, sl1, wg, wc, tv, save, B, L, dxp, rh, dxt, dtv, nullptr);
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms_b, e0, e1);
      printf("seq_len %2d (%2d steps): fwd %.1f us, bwd %.1f us per launch\n", len, len - 1, ms_f * 20.f, ms_b * 20.f);
    }
  }
#ifdef MTAM_GRU_STAMPS
  // (the stamps below are those of the LAST launch: run the full-length pair again after the short-sequence runs)
  mtam_tagru_fwd(xp, x, tl, sl, wg, wc, tv, B, L, hs, sh, save, nullptr);
  mtam_tagru_bwd(ds, nullptr, x, tl, sl, wg, wc, tv, save, B, L, dxp, rh, dxt, dtv, nullptr);
  hipDeviceSynchronize();
  unsigned long long st[2][8][8];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_gru_stamps), sizeof(st));
  const char *fn[8] = {"reads+T+gate FMAs", "reduce+sigmoid+LDS write", "barrier 1", "rh reads+cand FMAs+reduce",
                       "tanh+update", "prefetch landed", "stores", "barrier 2"};
  const char *bn[8] = {"element-wise", "barrier 1", "phase A", "gate grads+LDS", "barrier 2", "phase B", "land+stores", "-"};
  for (int k = 0; k < 2; ++k) {
    printf("%s: cycles per step by segment (waves 0, 3, 4, 7 of workgroup 0)\n", k ? "backward" : "forward");
    for (int i = 0; i < 8; ++i)
      printf("  %-28s %7.0f %7.0f %7.0f %7.0f\n", k ? bn[i] : fn[i], st[k][0][i] / 49.0, st[k][3][i] / 49.0,
             st[k][4][i] / 49.0, st[k][7][i] / 49.0);
    double tot = 0;
    for (int i = 0; i < 8; ++i) tot += st[k][0][i] / 49.0;
    printf("  total (wave 0)               %7.0f cycles per step\n", tot);
    unsigned long long rl[2][8][2];
    hipMemcpyFromSymbol(rl, HIP_SYMBOL(g_gru_real), sizeof(rl));
    printf("  the loop of wave 0: %llu s_memtime ticks in %llu s_memrealtime ticks (100 MHz): shader clock %.0f MHz\n",
           rl[k][0][0], rl[k][0][1], 100.0 * (double)rl[k][0][0] / (double)rl[k][0][1]);
  }
  {
    unsigned long long ph[2][8][8];
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_gru_phase), sizeof(ph));
    const char *pn[6] = {"weights -> registers", "(chunk barrier)", "x-projection rows staged", "time-gate inputs staged",
                         "recurrent steps", "write-back"};
    printf("forward, workgroup 0, wave 0: microseconds by phase (s_memrealtime)\n");
    for (int i = 0; i < 6; ++i) printf("  %-28s %6.2f\n", pn[i], ph[0][0][i] * 0.01);
    static unsigned long long wg[2][256][2];
    hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_gru_wg), sizeof(wg));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < B; ++i) { t0 = std::min(t0, wg[0][i][0]); t1 = std::max(t1, wg[0][i][1]); }
    printf("forward: %d workgroups, first start to last end %.2f us; start offsets (us) of workgroups 0, 16, 32, ..:", B, (t1 - t0) * 0.01);
    for (int i = 0; i < B; i += 16) printf(" %.1f", (wg[0][i][0] - t0) * 0.01);
    printf("\n  durations (us) of the same:");
    for (int i = 0; i < B; i += 16) printf(" %.1f", (wg[0][i][1] - wg[0][i][0]) * 0.01);
    printf("\n");
  }
#endif
  return 0;
}
