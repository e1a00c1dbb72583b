// Developer lab: the PRODUCTION gather kernel (csrc/emb.hip included as is) timed in the gather_lab
// harness, to separate harness effects from kernel differences.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../mtamrecommender_amd/csrc/emb.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <typename K>
float time_graph(K launch, hipStream_t st) {
  for (int i = 0; i < 3; ++i) launch();
  CK(hipStreamSynchronize(st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 50; ++i) launch();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, st));
  for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3f / (50 * 20);
}
int main() {
  const int B = 128, L = 50, V = 3709, NC = 304, NP = 53, NU = 4835, R = B * L;
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<int32_t> hi(R), hc(R), hp(R), hu(B);
  srand(1234);
  for (int i = 0; i < R; ++i) { hi[i] = rand() % V; hc[i] = rand() % NC; hp[i] = i % L; }
  for (int i = 0; i < B; ++i) hu[i] = rand() % NU;
  float *item, *cat, *pos, *user, *ic, *po, *uo, *l2; int32_t *di, *dc, *dp, *du;
  CK(hipMalloc(&item, (size_t)V * 512)); CK(hipMalloc(&cat, NC * 512)); CK(hipMalloc(&pos, NP * 512)); CK(hipMalloc(&user, NU * 512));
  CK(hipMalloc(&ic, (size_t)R * 1024)); CK(hipMalloc(&po, (size_t)R * 512)); CK(hipMalloc(&uo, B * 512));
  CK(hipMalloc(&l2, (size_t)mtam_emb_gather_partials(B, L) * 4));
  CK(hipMalloc(&di, R * 4)); CK(hipMalloc(&dc, R * 4)); CK(hipMalloc(&dp, R * 4)); CK(hipMalloc(&du, B * 4));
  CK(hipMemset(item, 0, (size_t)V * 512)); CK(hipMemset(cat, 0, NC * 512)); CK(hipMemset(pos, 0, NP * 512)); CK(hipMemset(user, 0, NU * 512));
  CK(hipMemcpy(di, hi.data(), R * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), R * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, hp.data(), R * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(du, hu.data(), B * 4, hipMemcpyHostToDevice));
  const double bytes = (3.0 * L + 1) * (2 * 128 * 4 + 4) * B;
  float us = time_graph([&] { mtam_emb_gather_fwd(item, V, cat, NC, pos, NP, user, NU, di, dc, dp, du, B, L, 1, ic, po, uo, l2, st); }, st);
  printf("production gather via C ABI  %7.2f us  frac %.3f\n", us, bytes / us * 1e-3 / 8000.0);
  return 0;
}
