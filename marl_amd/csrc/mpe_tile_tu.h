// Translation-unit body of the LDS-tile kernel launchers: included by mpe_tile_f32.hip / mpe_tile_f64.hip with
// MPE_TU_T = float / double (one TU per arithmetic type so the library builds in parallel).
#include <hip/hip_runtime.h>

#include "mpe_launch.h"
#include "mpe_tile_kernel.h"

namespace mpe {

static bool tile_kind_is_ragged(int kind) {  // == mpe_hip.hip:kind_is_ragged
  return kind == MPE_HIER_PURSUIT || kind == MPE_HIER_MOVING || kind == MPE_SPEAKER_LISTENER ||
         kind == MPE_SIMPLE_REFERENCE || kind == MPE_SIMPLE_CRYPTO || kind == MPE_WORLD_COMM;
}

// Raises a kernel's dynamic-LDS limit once per (kernel instantiation, device) instead of on every launch: a
// per-device high-water mark of what has been granted (function-local, one per template instantiation).
template <typename K>
static int grant_lds(K kern, size_t lds, size_t *granted /*[kMaxDevices]*/) {
  constexpr int kMaxDevices = 64;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev >= 0 && dev < kMaxDevices && __atomic_load_n(&granted[dev], __ATOMIC_RELAXED) >= lds) return 0;
  e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  if (dev >= 0 && dev < kMaxDevices) __atomic_store_n(&granted[dev], lds, __ATOMIC_RELAXED);
  return 0;
}

// Workgroup size of the LGLOB form (its register budget: mpe_tile_kernel.h, MPE_TILE_ATTR)
#ifndef MPE_LGLOB_THREADS
#define MPE_LGLOB_THREADS 512
#endif

template <typename T, int THREADS, bool DO_STEP, bool DO_OBS, bool HIER, int KIND, bool LGLOB = false>
static int launch_tile_k(const mpe_scenario_desc *h, const KParams<T> &p, hipStream_t s) {
  constexpr int TILE = tile_envs<T>();
  const size_t lds = LGLOB ? tile_lds_bytes_lglob<T>(h, THREADS) : tile_lds_bytes<T>(h, THREADS);
  auto kern = tile_kernel<T, TILE, THREADS, DO_STEP, DO_OBS, HIER, KIND, LGLOB>;
  if (lds > 64 * 1024) {
    if (lds > 160 * 1024) return MPE_ERR_UNSUPPORTED;
    static size_t granted[64] = {};
    const int rc = grant_lds(kern, lds, granted);
    if (rc) return rc;
  }
  const dim3 grid((unsigned)tile_grid(p.B, TILE)), block(THREADS);
  hipLaunchKernelGGL(kern, grid, block, lds, s, p);
  return (int)hipGetLastError();
}

template <typename T, int THREADS, bool DO_STEP, bool DO_OBS>
static int launch_tile(const mpe_scenario_desc *h, const KParams<T> &p, hipStream_t s) {
  if (tile_kind_is_ragged(h->kind)) return launch_tile_k<T, THREADS, DO_STEP, DO_OBS, true, -1>(h, p, s);
  // the north-star scenarios get their own instantiation (mpe_tile_kernel.h: KIND)
  if (h->kind == MPE_SIMPLE_TAG) {
#ifndef MPE_TILE_NO_LGLOB
    // worlds whose full tile leaves room for fewer than four workgroups per CU keep the landmarks in global memory
    // (mpe_tile_kernel.h: LGLOB): the fused step only, f32 (the f64 tile is 32 envs: half the bytes per row already)
    if constexpr (DO_STEP && DO_OBS && sizeof(T) == 4) {
      bool fixed = h->n_landmarks > 0;  // (a movable landmark cannot stay in global memory: its row changes under the pair loops)
      for (int e = h->n_agents; e < h->n_agents + h->n_landmarks; ++e) fixed = fixed && !h->movable[e];
      if (fixed && tile_lds_bytes<T>(h, THREADS) > 40 * 1024)
        return launch_tile_k<T, MPE_LGLOB_THREADS, DO_STEP, DO_OBS, false, MPE_SIMPLE_TAG, true>(h, p, s);
    }
#endif
    return launch_tile_k<T, THREADS, DO_STEP, DO_OBS, false, MPE_SIMPLE_TAG>(h, p, s);
  }
  if (h->kind == MPE_SIMPLE_SPREAD) return launch_tile_k<T, THREADS, DO_STEP, DO_OBS, false, MPE_SIMPLE_SPREAD>(h, p, s);
  return launch_tile_k<T, THREADS, DO_STEP, DO_OBS, false, -1>(h, p, s);
}

template <typename T, bool DO_STEP, bool DO_OBS>
int launch_tile_world(const mpe_scenario_desc *h, const KParams<T> &p, hipStream_t s) {
  // Large worlds are LDS-limited to 1-2 workgroups per CU: give those workgroups more wavefronts so
  // the compute-heavy pair phase and the store-heavy observation phase still find work to overlap.
  const int E = h->n_agents + h->n_landmarks;
  if (E > 32) return launch_tile<T, 512, DO_STEP, DO_OBS>(h, p, s);
  return launch_tile<T, 256, DO_STEP, DO_OBS>(h, p, s);
}

template int launch_tile_world<MPE_TU_T, true, false>(const mpe_scenario_desc *, const KParams<MPE_TU_T> &, hipStream_t);
template int launch_tile_world<MPE_TU_T, false, true>(const mpe_scenario_desc *, const KParams<MPE_TU_T> &, hipStream_t);
template int launch_tile_world<MPE_TU_T, true, true>(const mpe_scenario_desc *, const KParams<MPE_TU_T> &, hipStream_t);

}  // namespace mpe
