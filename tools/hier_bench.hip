// Stand-alone timing harness for one hier_kernel instantiation (kernel-tuning aid; builds in seconds,
// unlike the whole library).  L1F2_3Targets_100observe (A=3, L=3, NF=2) or S1L2F2_10Targets_wide
// (A=5, L=10, NF=2) constants, random state / actions, fused step, HIP-event timing.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DBIG] [-D<variant macros>] tools/hier_bench.hip -o ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../marl_amd/csrc/mpe_hier_kernel.h"
using namespace mpe;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#ifdef BIG
constexpr int A = 5, L = 10, NF = 2;
#else
constexpr int A = 3, L = 3, NF = 2;
#endif
typedef float T;
int main() {
  const int64_t B = 1 << 20;
  constexpr int E = A + L;
  KParams<T> p{};
  DevDesc<T> &d = p.d;
  d.kind = MPE_HIER_PURSUIT; d.A = A; d.L = L; d.E = E; d.dim_c = 7; d.info_w = 1;
  d.dt = 0.1f; d.damp1 = 0.75f; d.cforce = 100.f; d.k = 1e-3f; d.inv_k = 1e3f;
  d.has_lforce = 1; d.n_target_obs = 3; d.ncol_prefix = E;
  int row = 0, arow = 0;
  for (int a = 0; a < A; ++a) {
    d.size[a] = 0.075f; d.mass[a] = 1.f; d.accel[a] = 3.f; d.max_speed[a] = 1.f;
    d.movable[0] |= 1u << a; d.collide[0] |= 1u << a; d.adversary[0] |= 1u << a;
    const bool follower = a < NF;
    d.force_id[a] = -1; d.forced_id[a] = -1;
    if (follower) { d.silent[0] |= 1u << a; d.forced_id[a] = (A == 5 && a == 0) ? 2 : 1; d.obs_r2[a] = 0.04f; }
    else {
      d.advleader[0] |= 1u << a;
      if (A == 5) { d.force_id[a] = a == 2 ? 2 : (a == 3 ? 1 : 3); d.forced_id[a] = a < 4 ? 3 : -1; if (a == 4) d.advsuper[0] |= 1u << a; d.obs_r2[a] = 0.04f; }
      else { d.force_id[a] = 1; d.obs_r2[a] = 0.5625f; }
    }
    d.obs_row0[a] = row; row += 4 + 4 * (A - 1) + (d.forced_id[a] >= 0 ? 2 : 0) + 6;
    d.act_row0[a] = arow; arow += follower ? 5 : 12;
  }
  for (int e = A; e < E; ++e) { d.size[e] = 0.05f; d.collide[e >> 5] |= 1u << (e & 31); }
  d.obs_total = row;
  std::vector<float> h((size_t)64 * B);
  uint32_t s = 12345;
  for (auto &x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xFFFF) / 65536.f * 2.f - 1.f; }
  T *pos, *vel, *act, *lf, *obs, *rew, *info;
  CK(hipMalloc(&pos, 2 * E * B * 4)); CK(hipMalloc(&vel, 2 * A * B * 4)); CK(hipMalloc(&act, (size_t)arow * B * 4));
  CK(hipMalloc(&lf, 2 * A * B * 4)); CK(hipMalloc(&obs, (size_t)row * B * 4)); CK(hipMalloc(&rew, A * B * 4)); CK(hipMalloc(&info, A * B * 4));
  CK(hipMemcpy(pos, h.data(), 2 * E * B * 4, hipMemcpyHostToDevice));
  CK(hipMemset(vel, 0, 2 * A * B * 4));
  CK(hipMemcpy(act, h.data(), (size_t)arow * B * 4, hipMemcpyHostToDevice));
  p.act = act; p.pos = pos; p.vel = vel; p.lforce = lf; p.obs = obs; p.rew = rew; p.info = info; p.B = B; p.ld = B;
  const dim3 grid((unsigned)(B / 256)), block(256);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((hier_kernel<T, A, L, NF, 256, true, true>), grid, block, 0, 0, p);
  CK(hipEventRecord(e0));
  const int iters = 50;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((hier_kernel<T, A, L, NF, 256, true, true>), grid, block, 0, 0, p);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)(arow + 2 * E + 2 * A + 4 * A + row + A + A + 2 * A) * 4 * B;
  printf("%-28s A=%d L=%d: %7.1f us  %6.0f GB/s\n", VARIANT, A, L, ms / iters * 1e3, bytes / (ms / iters) / 1e6);
  return 0;
}
