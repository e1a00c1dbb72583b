// Developer lab: latency anatomy of the B=128 embedding gather (19,328 rows of 512 B).
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_lab.hip -o tools/gather_lab ; run on the GPU box.
// Every variant is timed as 50 back-to-back launches captured in a hipGraph, replayed 20 times.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int D = 128;
struct Args {
  const float *item, *cat, *pos, *user;
  const int32_t *item_ids, *cat_ids, *pos_ids, *user_ids;
  int B, L;
  float *ic, *pos_out, *user_out, *l2;
};

__device__ __forceinline__ float wave_sum(float v) {
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ void k_empty(Args p) {}

__global__ __launch_bounds__(256) void k_ids_only(Args p) {
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  int s = wave_id * 4;
  int acc = 0;
  if (s + 3 < p.B * p.L) acc = p.item_ids[s] + p.item_ids[s + 1] + p.item_ids[s + 2] + p.item_ids[s + 3];
  if (acc == 0x7fffffff && lane == 0) p.l2[wave_id] = 1.f;
}

// SLOTS rows in flight per half wave; NT: non-temporal stores; SCALAR: ids through readfirstlane
template <int SLOTS, bool NT, bool SCALAR, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void k_gather(Args p) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, li = lane & 31;
  const int R = p.B * p.L;
  const int total = R + (R + p.B + 1) / 2;
  const int wave_id = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
  const int s0 = wave_id * SLOTS;
  const float *src[SLOTS];
  float *dst[SLOTS];
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int s = s0 + i;
    src[i] = nullptr;
    dst[i] = nullptr;
    if (s < R) {
      int id;
      if (SCALAR) {
        const int a = p.item_ids[__builtin_amdgcn_readfirstlane(s)];
        const int b = p.cat_ids[__builtin_amdgcn_readfirstlane(s)];
        id = half ? b : a;
      } else {
        id = half ? p.cat_ids[s] : p.item_ids[s];
      }
      const float *tab = half ? p.cat : p.item;
      src[i] = tab + (size_t)id * D + 4 * li;
      dst[i] = p.ic + (size_t)s * (2 * D) + half * D + 4 * li;
    } else if (s < total) {
      const int q = 2 * (s - R) + half;
      if (q < R) {
        src[i] = p.pos + (size_t)p.pos_ids[q] * D + 4 * li;
        dst[i] = p.pos_out + (size_t)q * D + 4 * li;
      } else if (q < R + p.B) {
        src[i] = p.user + (size_t)p.user_ids[q - R] * D + 4 * li;
        dst[i] = p.user_out + (size_t)(q - R) * D + 4 * li;
      }
    }
  }
  float4 v[SLOTS];
#pragma unroll
  for (int i = 0; i < SLOTS; ++i)
    v[i] = src[i] ? *reinterpret_cast<const float4 *>(src[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    if (dst[i]) {
      if (NT) {
        typedef float vf4 __attribute__((ext_vector_type(4)));
        vf4 t = {v[i].x, v[i].y, v[i].z, v[i].w};
        __builtin_nontemporal_store(t, reinterpret_cast<vf4 *>(dst[i]));
      }
      else *reinterpret_cast<float4 *>(dst[i]) = v[i];
      sq += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
  }
  sq = wave_sum(sq);
  if (lane == 0) p.l2[wave_id] = sq;
}

// full wave per row pair: lane handles 8 B?  no -- one wave moves TWO slots' worth as 64 x 16 B = 2 rows;
// this variant instead gives each wave ONE [item|cat] 1-KB row (1 slot), maximum wave count.
template <typename K>
float time_graph(K launch, hipStream_t st) {
  for (int i = 0; i < 3; ++i) launch();
  CK(hipStreamSynchronize(st));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 50; ++i) launch();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(hipEventRecord(a, st));
  for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return ms * 1e3f / (50 * 20);
}

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128, L = 50;
  const int V = argc > 2 ? atoi(argv[2]) : 3709, NC = 304, NP = 53, NU = 4835;
  const int R = B * L;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::vector<int32_t> hi(R), hc(R), hp(R), hu(B);
  srand(1234);
  for (int i = 0; i < R; ++i) { hi[i] = rand() % V; hc[i] = rand() % NC; hp[i] = i % L; }
  for (int i = 0; i < B; ++i) hu[i] = rand() % NU;
  Args a;
  float *item, *cat, *pos, *user, *ic, *po, *uo, *l2;
  int32_t *di, *dc, *dp, *du;
  CK(hipMalloc(&item, (size_t)V * D * 4)); CK(hipMalloc(&cat, NC * D * 4)); CK(hipMalloc(&pos, NP * D * 4));
  CK(hipMalloc(&user, NU * D * 4)); CK(hipMalloc(&ic, (size_t)R * 2 * D * 4)); CK(hipMalloc(&po, (size_t)R * D * 4));
  CK(hipMalloc(&uo, B * D * 4)); CK(hipMalloc(&l2, (size_t)(2 * R + 64) * 4));
  CK(hipMalloc(&di, R * 4)); CK(hipMalloc(&dc, R * 4)); CK(hipMalloc(&dp, R * 4)); CK(hipMalloc(&du, B * 4));
  CK(hipMemset(item, 0, (size_t)V * D * 4)); CK(hipMemset(cat, 0, NC * D * 4)); CK(hipMemset(pos, 0, NP * D * 4));
  CK(hipMemset(user, 0, NU * D * 4));
  CK(hipMemcpy(di, hi.data(), R * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), R * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, hp.data(), R * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(du, hu.data(), B * 4, hipMemcpyHostToDevice));
  a = Args{item, cat, pos, user, di, dc, dp, du, B, L, ic, po, uo, l2};
  const int total = R + (R + B + 1) / 2;
  const double bytes = (3.0 * L + 1) * (2 * D * 4 + 4) * B;
  auto report = [&](const char *name, float us) {
    printf("%-28s %7.2f us  %7.1f GB/s  frac %.3f\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
  };
  auto blocks = [&](int slots) { return ((total + slots - 1) / slots + 3) / 4; };
  report("empty", time_graph([&] { hipLaunchKernelGGL(k_empty, dim3(blocks(4)), dim3(256), 0, st, a); }, st));
  report("ids_only", time_graph([&] { hipLaunchKernelGGL(k_ids_only, dim3(blocks(4)), dim3(256), 0, st, a); }, st));
#define RUN(S, NT, SC) report("gather slots=" #S " nt=" #NT " scalar=" #SC, time_graph([&] { \
    hipLaunchKernelGGL((k_gather<S, NT, SC>), dim3(blocks(S)), dim3(256), 0, st, a); }, st))
  // launch floor against the number (and size) of workgroups
  for (int g : {128, 256, 604, 1208, 2416, 4832})
    for (int th : {64, 256, 512, 1024}) {
      char name[64];
      snprintf(name, sizeof name, "empty grid=%d threads=%d", g, th);
      report(name, time_graph([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(th), 0, st, a); }, st));
    }
#define RUNT(S, TH) report("gather slots=" #S " threads=" #TH, time_graph([&] { \
    hipLaunchKernelGGL((k_gather<S, false, false, TH>), dim3(((total + S - 1) / S + TH / 64 - 1) / (TH / 64)), dim3(TH), 0, st, a); }, st))
  RUNT(2, 128);
  RUNT(2, 512);
  RUNT(2, 1024);
  RUNT(1, 512);
  RUNT(1, 1024);
  RUNT(3, 256);
  RUNT(3, 512);
  RUN(1, false, false);
  RUN(2, false, false);
  RUN(4, false, false);
  RUN(8, false, false);
  RUN(4, true, false);
  RUN(4, false, true);
  RUN(2, false, true);
  RUN(2, true, true);
  RUN(8, true, true);
  return 0;
}
