// Stand-alone timing harness for the stock simple_tag register kernel (kernel-tuning aid).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DHDR='"..."' -DVARIANT='"..."' [-DMPE_SMALL_ATTR=...] tools/small_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef HDR
#define HDR "../marl_amd/csrc/mpe_small_kernel.h"
#endif
#include HDR
using namespace mpe;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#ifndef TT
#define TT float
#endif
#ifndef BENCH_THREADS
#define BENCH_THREADS 256
#endif
typedef TT T;
int main() {
#ifndef PAD
#define PAD 256
#endif
  const int64_t B = 1 << 20, LD = B + PAD;  // row stride off the power of two (core.py:padded_stride)
  constexpr int A = 4, L = 2, E = 6;
  KParams<T> p{};
  DevDesc<T> &d = p.d;
  d.kind = MPE_SIMPLE_TAG; d.A = A; d.L = L; d.E = E; d.dim_c = 2; d.info_w = 1; d.n_adv = 3; d.n_good = 1;
  d.dt = 0.1f; d.damp1 = 0.75f; d.cforce = 100.f; d.k = 1e-3f; d.inv_k = 1e3f; d.ncol_prefix = E;
  int row = 0;
  for (int a = 0; a < A; ++a) {
    const bool adv = a < 3;
    d.size[a] = adv ? 0.075f : 0.05f; d.mass[a] = 1.f; d.accel[a] = adv ? 3.f : 4.f; d.max_speed[a] = adv ? 1.f : 1.3f;
    d.movable[0] |= 1u << a; d.collide[0] |= 1u << a;
    if (adv) d.adversary[0] |= 1u << a;
    d.obs_row0[a] = row; row += 4 + 2 * L + 2 * (A - 1) + (adv ? 2 : 0);
  }
  for (int e = A; e < E; ++e) { d.size[e] = 0.2f; d.collide[0] |= 1u << e; }
  d.obs_total = row;
  std::vector<T> h((size_t)32 * LD);
  uint32_t s = 12345;
  for (auto &x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xFFFF) / 65536.f * 2.f - 1.f; }
  T *pos, *vel, *act, *obs, *rew, *info;
  const size_t es = sizeof(T);
  CK(hipMalloc(&pos, 2 * E * LD * es)); CK(hipMalloc(&vel, 2 * A * LD * es)); CK(hipMalloc(&act, 5 * A * LD * es));
  CK(hipMalloc(&obs, (size_t)row * LD * es)); CK(hipMalloc(&rew, A * LD * es)); CK(hipMalloc(&info, A * LD * es));
  CK(hipMemcpy(pos, h.data(), 2 * E * LD * es, hipMemcpyHostToDevice));
  CK(hipMemset(vel, 0, 2 * A * LD * es));
  CK(hipMemcpy(act, h.data(), 5 * A * LD * es, hipMemcpyHostToDevice));
  p.act = act; p.pos = pos; p.vel = vel; p.obs = obs; p.rew = rew; p.info = info; p.B = B; p.ld = LD;
  const dim3 grid((unsigned)(B / BENCH_THREADS)), block(BENCH_THREADS);
  auto kern = small_kernel<T, MPE_SIMPLE_TAG, 4, 2, 3, BENCH_THREADS, false, true, true>;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, grid, block, 0, 0, p);
  CK(hipEventRecord(e0));
  const int iters = 100;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, block, 0, 0, p);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)(5 * A + 2 * E + 2 * A + 4 * A + row + A + A) * es * B;
  printf("%-34s T=%4d %s: %7.1f us  %6.0f GB/s\n", VARIANT, BENCH_THREADS, sizeof(T) == 4 ? "f32" : "f64", ms / iters * 1e3, bytes / (ms / iters) / 1e6);
  return 0;
}
