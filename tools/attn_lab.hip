// Developer lab: phase times of the decoder's forward kernel (ta_attn_decode_fwd_kernel), stamped build.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMTAM_ATTN_STAMPS tools/attn_lab.hip mtamrecommender_amd/csrc/capi.hip -o tools/attn_lab
#include "../mtamrecommender_amd/csrc/ta_attn.hip"
#include <vector>

template <class T> static T *dev_alloc(size_t n, float scale = 0.1f, int mod = 0, int base = 0) {
  T *p; (void)hipMalloc(&p, n * sizeof(T));
  std::vector<T> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = mod ? (T)(base + rand() % mod) : (T)(scale * ((rand() % 2001) / 1000.f - 1.f));
  (void)hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice);
  return p;
}

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128, L = 50, H = 1, R = B * L;
  srand(1);
  float *dec_in = dev_alloc<float>(B * D), *x = dev_alloc<float>((size_t)R * D), *kv = dev_alloc<float>((size_t)R * 2 * D),
        *tq = dev_alloc<float>(B, 100.f), *tk = dev_alloc<float>(R, 100.f), *wqt = dev_alloc<float>(D * 2 * D), *bq = dev_alloc<float>(D),
        *tp = dev_alloc<float>(5 * L), *lb = dev_alloc<float>(D), *lg = dev_alloc<float>(D), *hb = dev_alloc<float>(D), *hg = dev_alloc<float>(D);
  int32_t *sl = dev_alloc<int32_t>(B, 0, 20, 31);
  float *dec_out = dev_alloc<float>(B * D), *save = dev_alloc<float>((size_t)B * (3 * D + 3 * L + 2 * H * L + 1)), *pred = dev_alloc<float>(B * D),
        *hsave = dev_alloc<float>(B * (D + 1));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0.f; const int reps = 100;
  for (int round = 0; round < 2; ++round) {
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
      if (mtam_ta_attn_decode_fwd(dec_in, x, kv, 2 * D, 0, D, tq, tk, sl, wqt, bq, tp, lb, lg, B, L, H, dec_out, save, hb, hg, pred, hsave, nullptr)) {
        printf("fwd: %s\n", mtam_last_error()); return 1;
      }
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  printf("B = %d: %.2f us per launch (stamped build, back to back)\n", B, ms * 1000.f / reps);
  // backward of the same block (head layer_norm fused: d_pred in, d_out unused)
  float *d_pred = dev_alloc<float>(B * D), *d_dec_in = dev_alloc<float>(B * D), *d_kv = dev_alloc<float>((size_t)R * 2 * D),
        *d_x = dev_alloc<float>((size_t)R * D), *d_qt = dev_alloc<float>(B * 2 * D), *d_tp = dev_alloc<float>((size_t)B * 5 * L),
        *d_ln = dev_alloc<float>(B * 2 * D), *d_head = dev_alloc<float>(B * 2 * D);
  for (int round = 0; round < 2; ++round) {
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
      if (mtam_ta_attn_decode_bwd(nullptr, dec_in, x, kv, 2 * D, 0, D, tq, tk, sl, wqt, tp, lg, save, B, L, H, 0, d_dec_in, d_kv, d_x, d_qt,
                                  d_tp, d_ln, d_pred, hg, hsave, d_head, nullptr)) { printf("bwd: %s\n", mtam_last_error()); return 1; }
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  printf("backward: %.2f us per launch (stamped build, back to back)\n", ms * 1000.f / reps);
#ifdef MTAM_ATTN_STAMPS
  unsigned long long st[2][16];
  (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_attn_stamps), sizeof(st));
  const char *pts[9] = {"start", "argument-only loads issued", "q in LDS", "[Q | qt] projected", "scores", "softmax", "weighted values",
                        "normalize, saves issued", "head layer_norm"};
  for (int i = 1; i < 9; ++i)
    if (st[0][i]) printf("  %-30s %6.2f us\n", pts[i], (double)(st[0][i] - st[0][0]) / 100.0);
  const char *ptb[10] = {"start", "row loads issued", "head layer_norm backward", "normalize backward", "dW", "softmax backward", "gate backward",
                         "per-key gradients", "dQ, d(qt) reduced", "d(dec_in)"};
  printf("backward:\n");
  for (int i = 1; i < 10; ++i)
    if (st[1][i]) printf("  %-30s %6.2f us\n", ptb[i], (double)(st[1][i] - st[1][0]) / 100.0);
#endif
  return 0;
}
