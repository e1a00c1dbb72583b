// Developer lab: where a step of the serial GRU kernels spends its cycles.  Builds csrc/tagru.hip with in-kernel
// stamps (-DMTAM_GRU_STAMPS: a diagnostic build, its run time is not quoted) and, in the same binary, times the
// un-stamped entry points of the product library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMTAM_GRU_STAMPS tools/gru_lab.hip mtamrecommender_amd/csrc/capi.hip -o tools/gru_lab
#include "../mtamrecommender_amd/csrc/tagru.hip"
#include <stdlib.h>
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
  }
#endif
  return 0;
}
