// Developer lab: phase times of the forward stripe kernel (seq_chain_x3_kernel<true>), stamped build.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMTAM_CHAIN_STAMPS tools/chain_lab.hip mtamrecommender_amd/csrc/capi.hip -o tools/chain_lab
//   tools/chain_lab [B]
#include "../mtamrecommender_amd/csrc/seq_chain.hip"
#include <vector>

template <class T> static T *dev_alloc(size_t n, float scale = 0.1f, int mod = 0) {
  T *p; (void)hipMalloc(&p, n * sizeof(T));
  std::vector<T> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = mod ? (T)(rand() % mod) : (T)(scale * ((rand() % 2001) / 1000.f - 1.f));
  (void)hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice);
  return p;
}

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128, L = 50, R = B * L, V = 3709, C = 304, U = 4835, NX = 384;
  srand(1);
  float *item = dev_alloc<float>((size_t)V * D), *cat = dev_alloc<float>((size_t)C * D), *pos = dev_alloc<float>((L + 3) * D),
        *user = dev_alloc<float>((size_t)U * D), *W4 = dev_alloc<float>(2 * D * D), *Wx = dev_alloc<float>(D * NX), *bx = dev_alloc<float>(NX);
  int32_t *iid = dev_alloc<int32_t>(R, 0, V), *cid = dev_alloc<int32_t>(R, 0, C), *pid = dev_alloc<int32_t>(R, 0, L), *uid = dev_alloc<int32_t>(B, 0, U);
  float *ic = dev_alloc<float>((size_t)R * 2 * D), *uo = dev_alloc<float>(B * D), *zr = dev_alloc<float>((size_t)R * D),
        *x = dev_alloc<float>((size_t)R * D), *xproj = dev_alloc<float>((size_t)R * NX), *l2 = dev_alloc<float>(mtam_seq_chain_gather_partials(B, L));
  uint16_t *img; (void)hipMalloc(&img, mtam_seq_chain_images_elems(0, NX) * 2);
  if (mtam_split_weight_images(W4, 2 * D, D, img + mtam_seq_chain_image_offset(0, NX), nullptr) ||
      mtam_split_weight_images(Wx, D, NX, img + mtam_seq_chain_image_offset(2, NX), nullptr)) { printf("images: %s\n", mtam_last_error()); return 1; }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0.f; const int reps = 100;
  for (int round = 0; round < 2; ++round) {
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
      if (mtam_seq_chain_gather_fwd(item, V, cat, C, pos, L + 3, user, U, iid, cid, pid, uid, B, L, 1, W4, nullptr, nullptr, 0, Wx, bx, NX,
                                    ic, uo, l2, mtam_seq_chain_gather_partials(B, L), zr, x, nullptr, xproj, nullptr, 0, nullptr, 0, img,
                                    nullptr)) { printf("fwd: %s\n", mtam_last_error()); return 1; }
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  printf("B = %d (%d workgroups): %.2f us per launch (stamped build, back to back)\n", B, (R + 31) / 32, ms * 1000.f / reps);
#ifdef MTAM_CHAIN_STAMPS
  unsigned long long st[2][16];
  (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_chain_stamps), sizeof(st));
  const char *pts[16] = {"start", "ids in, rows requested", "rows split into LDS", "pos/user rows, clears", "barrier 1", "all loads landed (W4 images)",
                         "first product", "x in LDS", "zr, x stored; A split", "block 1", "block 2", "block 3", "block 4", "block 5", "-", "stores drained"};
  for (int i = 1; i < 16; ++i)
    if (st[0][i]) printf("  %-34s %6.2f us\n", pts[i], (double)(st[0][i] - st[0][0]) / 100.0);
#endif
  return 0;
}
