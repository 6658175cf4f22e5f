// libmpe_hip.so -- C ABI (include/mpe_hip.h) over the gfx950 kernels.
// Host side: argument validation, descriptor conversion, kernel selection, launch.  Nothing here
// allocates, synchronises or keeps state.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "mpe_device.h"
#include "mpe_small_kernel.h"
#include "mpe_tile_kernel.h"

using namespace mpe;

static_assert(sizeof(KParams<double>) <= 4096, "kernarg segment must stay under 4 KiB");

// ------------------------------------------------------------------------------------------------
// descriptor helpers (host)
// ------------------------------------------------------------------------------------------------
static int obs_dim_of(const mpe_scenario_desc *h, int i) {
  const int A = h->n_agents, L = h->n_landmarks;
  if (h->kind == MPE_SIMPLE_TAG) {  // simple_tag.py:147
    int good_others = 0;
    for (int j = 0; j < A; ++j)
      if (j != i && !h->adversary[j]) ++good_others;
    return 4 + 2 * L + 2 * (A - 1) + 2 * good_others;
  }
  return 4 + 2 * L + 2 * (A - 1) + h->dim_c * (A - 1);  // simple_spread.py:99
}

extern "C" int mpe_abi_version(void) { return MPE_ABI_VERSION; }
extern "C" int64_t mpe_desc_sizeof(void) { return (int64_t)sizeof(mpe_scenario_desc); }

extern "C" const char *mpe_strerror(int code) {
  switch (code) {
    case MPE_OK: return "ok";
    case MPE_ERR_NULL: return "required pointer is NULL";
    case MPE_ERR_DESC: return "invalid scenario descriptor";
    case MPE_ERR_BATCH: return "bad batch size or row stride (need B >= 1, ld >= B)";
    case MPE_ERR_ALIGN: return "device pointer not aligned to its element type";
    case MPE_ERR_UNSUPPORTED: return "unsupported world (movable landmarks / unknown scenario kind)";
    case MPE_ERR_ARG: return "bad argument";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

extern "C" int mpe_desc_check(const mpe_scenario_desc *h) {
  if (!h) return MPE_ERR_NULL;
  if (h->kind != MPE_SIMPLE_TAG && h->kind != MPE_SIMPLE_SPREAD) return MPE_ERR_UNSUPPORTED;
  if (h->n_agents < 1 || h->n_agents > MPE_MAX_AGENTS) return MPE_ERR_DESC;
  if (h->n_landmarks < 0 || h->n_agents + h->n_landmarks > MPE_MAX_ENTITIES) return MPE_ERR_DESC;
  if (h->dim_c < 0 || h->dim_c > 16) return MPE_ERR_DESC;
  if (!(h->contact_margin > 0.0) || !std::isfinite(h->dt) || !std::isfinite(h->damping) ||
      !std::isfinite(h->contact_force))
    return MPE_ERR_DESC;
  const int E = h->n_agents + h->n_landmarks;
  for (int e = 0; e < E; ++e) {
    if (!std::isfinite(h->size[e]) || !(h->mass[e] > 0.0)) return MPE_ERR_DESC;
    if (e >= h->n_agents && h->movable[e]) return MPE_ERR_UNSUPPORTED;
  }
  for (int a = 0; a < h->n_agents; ++a)
    if (!std::isfinite(h->accel[a]) || std::isnan(h->max_speed[a]) || !std::isfinite(h->u_noise[a]))
      return MPE_ERR_DESC;
  return MPE_OK;
}

extern "C" int mpe_obs_dim(const mpe_scenario_desc *h, int agent) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  if (agent < 0 || agent >= h->n_agents) return MPE_ERR_ARG;
  return obs_dim_of(h, agent);
}
extern "C" int mpe_obs_row0(const mpe_scenario_desc *h, int agent) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  if (agent < 0 || agent >= h->n_agents) return MPE_ERR_ARG;
  int r = 0;
  for (int i = 0; i < agent; ++i) r += obs_dim_of(h, i);
  return r;
}
extern "C" int mpe_obs_total(const mpe_scenario_desc *h) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  int r = 0;
  for (int i = 0; i < h->n_agents; ++i) r += obs_dim_of(h, i);
  return r;
}
extern "C" int mpe_info_width(const mpe_scenario_desc *h) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  return h->kind == MPE_SIMPLE_TAG ? 1 : 4;
}
extern "C" int mpe_stats_width(const mpe_scenario_desc *h) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  return 2 * h->n_agents + 2;
}

template <typename T>
static void to_device_desc(const mpe_scenario_desc *h, DevDesc<T> &d) {
  std::memset(&d, 0, sizeof(d));
  d.kind = h->kind;
  d.A = h->n_agents;
  d.L = h->n_landmarks;
  d.E = d.A + d.L;
  d.dim_c = h->dim_c;
  d.info_w = h->kind == MPE_SIMPLE_TAG ? 1 : 4;
  d.dt = (T)h->dt;
  d.damp1 = (T)(1 - h->damping);  // core.py:214: (1 - self.damping) evaluated in float64
  d.cforce = (T)h->contact_force;
  d.k = (T)h->contact_margin;
  d.inv_k = (T)(1.0 / h->contact_margin);
  int row = 0;
  for (int e = 0; e < d.E; ++e) {
    d.size[e] = (T)h->size[e];
    if (h->movable[e]) d.movable[e >> 5] |= 1u << (e & 31);
    if (h->collide[e]) d.collide[e >> 5] |= 1u << (e & 31);
  }
  for (int a = 0; a < d.A; ++a) {
    d.mass[a] = (T)h->mass[a];
    d.accel[a] = (T)h->accel[a];
    d.max_speed[a] = (T)h->max_speed[a];
    d.u_noise[a] = (T)h->u_noise[a];
    if (h->kind == MPE_SIMPLE_TAG && h->adversary[a]) {
      d.adversary[a >> 5] |= 1u << (a & 31);
      d.adv_list[d.n_adv++] = (uint8_t)a;
    } else {
      d.good_list[d.n_good++] = (uint8_t)a;
    }
    d.obs_row0[a] = row;
    row += obs_dim_of(h, a);
  }
  d.obs_total = row;
  int ncol = 0;
  while (ncol < d.E && h->collide[ncol]) ++ncol;
  d.ncol_prefix = ncol;
  for (int e = ncol; e < d.E; ++e)
    if (h->collide[e]) d.ncol_prefix = -1;
}

// ------------------------------------------------------------------------------------------------
// kernel selection
// ------------------------------------------------------------------------------------------------
enum SmallId { SMALL_NONE = 0, SMALL_TAG_3_1_2, SMALL_SPREAD_3_3 };

// The register kernel hard-codes the stock role flags (mpe_small_kernel.h); anything else -> tile kernel.
static SmallId small_variant(const mpe_scenario_desc *h) {
  const int A = h->n_agents, L = h->n_landmarks;
  const bool tag = h->kind == MPE_SIMPLE_TAG;
  if (!((tag && A == 4 && L == 2) || (!tag && A == 3 && L == 3))) return SMALL_NONE;
  for (int a = 0; a < A; ++a) {
    if (!h->movable[a] || !h->collide[a]) return SMALL_NONE;
    if (tag ? !(h->max_speed[a] >= 0.0) : (h->max_speed[a] >= 0.0)) return SMALL_NONE;
    if (tag && (h->adversary[a] != 0) != (a < 3)) return SMALL_NONE;
  }
  for (int e = A; e < A + L; ++e)
    if ((h->collide[e] != 0) != tag) return SMALL_NONE;
  return tag ? SMALL_TAG_3_1_2 : SMALL_SPREAD_3_3;
}

static bool g_force_tile = false;  // test hook: route the stock worlds through the tile kernel too
extern "C" void mpe_debug_force_tile_kernel(int on) { g_force_tile = on != 0; }

template <typename T, int THREADS, bool DO_STEP, bool DO_OBS>
static int launch_tile(const mpe_scenario_desc *h, const KParams<T> &p, hipStream_t s) {
  constexpr int TILE = tile_envs<T>();
  const size_t lds = tile_lds_bytes<T>(h, THREADS);
  auto kern = tile_kernel<T, TILE, THREADS, DO_STEP, DO_OBS>;
  if (lds > 64 * 1024) {
    if (lds > 160 * 1024) return MPE_ERR_UNSUPPORTED;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const dim3 grid((unsigned)tile_grid(p.B, TILE)), block(THREADS);
  hipLaunchKernelGGL(kern, grid, block, lds, s, p);
  return (int)hipGetLastError();
}

template <typename T, int KIND, int A, int L, int NADV, int THREADS, bool DO_STEP, bool DO_OBS>
static int launch_small(const KParams<T> &p, bool noise, hipStream_t s) {
  const dim3 grid((unsigned)((p.B + THREADS - 1) / THREADS)), block(THREADS);
  if (noise)
    hipLaunchKernelGGL((small_kernel<T, KIND, A, L, NADV, THREADS, true, DO_STEP, DO_OBS>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((small_kernel<T, KIND, A, L, NADV, THREADS, false, DO_STEP, DO_OBS>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

template <typename T, bool DO_STEP, bool DO_OBS>
static int launch(const mpe_scenario_desc *h, const KParams<T> &p, hipStream_t s) {
  const SmallId sv = g_force_tile ? SMALL_NONE : small_variant(h);
  if (sv != SMALL_NONE) {
    bool noise = false;
    if (DO_STEP && p.noise != nullptr)
      for (int a = 0; a < h->n_agents; ++a) noise = noise || h->u_noise[a] != 0.0;
    if constexpr (sizeof(T) == 4) {
      if (small_threads(p.B, 4) == kSmallThreadsBig) {
        if (sv == SMALL_TAG_3_1_2)
          return launch_small<T, MPE_SIMPLE_TAG, 4, 2, 3, kSmallThreadsBig, DO_STEP, DO_OBS>(p, noise, s);
        return launch_small<T, MPE_SIMPLE_SPREAD, 3, 3, 0, kSmallThreadsBig, DO_STEP, DO_OBS>(p, noise, s);
      }
    }
    if (sv == SMALL_TAG_3_1_2)
      return launch_small<T, MPE_SIMPLE_TAG, 4, 2, 3, kSmallThreads, DO_STEP, DO_OBS>(p, noise, s);
    return launch_small<T, MPE_SIMPLE_SPREAD, 3, 3, 0, kSmallThreads, DO_STEP, DO_OBS>(p, noise, s);
  }
  // Large worlds are LDS-limited to 1-2 workgroups per CU: give those workgroups more wavefronts so
  // the compute-heavy pair phase and the store-heavy observation phase still find work to overlap.
  const int E = h->n_agents + h->n_landmarks;
  if (E > 32) return launch_tile<T, 512, DO_STEP, DO_OBS>(h, p, s);
  return launch_tile<T, 256, DO_STEP, DO_OBS>(h, p, s);
}

extern "C" int64_t mpe_stats_rows(const mpe_scenario_desc *h, int64_t B, int dtype_bytes) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  if (B < 1) return MPE_ERR_BATCH;
  if (dtype_bytes != 4 && dtype_bytes != 8) return MPE_ERR_ARG;
  if (!g_force_tile && small_variant(h) != SMALL_NONE) return (B + small_threads(B, dtype_bytes) - 1) / small_threads(B, dtype_bytes);
  return tile_grid(B, dtype_bytes == 4 ? 64 : 32);
}

template <typename T>
static bool misaligned(const void *ptr) { return ((uintptr_t)ptr % sizeof(T)) != 0; }

template <typename T>
static int check_common(const mpe_scenario_desc *h, int64_t B, int64_t ld) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  if (B < 1 || ld < B) return MPE_ERR_BATCH;
  return MPE_OK;
}

template <typename T>
static int world_step_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, const T *act, const T *noise,
                           T *pos, T *vel, T *comm, void *stream) {
  int rc = check_common<T>(h, B, ld);
  if (rc) return rc;
  if (!act || !pos || !vel) return MPE_ERR_NULL;
  if (misaligned<T>(act) || misaligned<T>(noise) || misaligned<T>(pos) || misaligned<T>(vel) || misaligned<T>(comm))
    return MPE_ERR_ALIGN;
  KParams<T> p{};
  p.act = act; p.noise = noise; p.pos = pos; p.vel = vel; p.comm = comm;
  p.B = B; p.ld = ld;
  to_device_desc<T>(h, p.d);
  return launch<T, true, false>(h, p, (hipStream_t)stream);
}

template <typename T>
static int observe_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, const T *pos, const T *vel, T *obs,
                        T *rew, T *info, void *stream) {
  int rc = check_common<T>(h, B, ld);
  if (rc) return rc;
  if (!pos || !vel) return MPE_ERR_NULL;
  if (!obs && !rew && !info) return MPE_ERR_NULL;
  if (misaligned<T>(pos) || misaligned<T>(vel) || misaligned<T>(obs) || misaligned<T>(rew) || misaligned<T>(info))
    return MPE_ERR_ALIGN;
  KParams<T> p{};
  p.pos = const_cast<T *>(pos); p.vel = const_cast<T *>(vel);
  p.obs = obs; p.rew = rew; p.info = info;
  p.B = B; p.ld = ld;
  to_device_desc<T>(h, p.d);
  return launch<T, false, true>(h, p, (hipStream_t)stream);
}

template <typename T>
static int env_step_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, const T *act, const T *noise, T *pos,
                         T *vel, T *comm, T *obs, T *rew, T *info, double *stats, void *stream) {
  int rc = check_common<T>(h, B, ld);
  if (rc) return rc;
  if (!act || !pos || !vel) return MPE_ERR_NULL;
  if (misaligned<T>(act) || misaligned<T>(noise) || misaligned<T>(pos) || misaligned<T>(vel) ||
      misaligned<T>(comm) || misaligned<T>(obs) || misaligned<T>(rew) || misaligned<T>(info) ||
      misaligned<double>(stats))
    return MPE_ERR_ALIGN;
  KParams<T> p{};
  p.act = act; p.noise = noise; p.pos = pos; p.vel = vel; p.comm = comm;
  p.obs = obs; p.rew = rew; p.info = info; p.stats = stats;
  p.B = B; p.ld = ld;
  to_device_desc<T>(h, p.d);
  return launch<T, true, true>(h, p, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// reset (Scenario.reset_world) and statistics reduce
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void reset_kernel(int64_t B, int64_t ld, int E, int A, int dim_c, double ra,
                                                     double rl, uint32_t k0, uint32_t k1, uint32_t episode,
                                                     uint64_t env_id0, const uint8_t *mask, T *pos, T *vel,
                                                     T *comm) {
  const int64_t env = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (env >= B) return;
  if (mask != nullptr && mask[env] == 0) return;
  const uint64_t gid = env_id0 + (uint64_t)env;
  for (int e = 0; e < E; ++e) {
    uint32_t w[4];
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), episode, (uint32_t)e, k0, k1, w);
    const double r = e < A ? ra : rl;
    pos[(int64_t)(2 * e) * ld + env] = (T)uniform_pm(w[0], w[1], r);
    pos[(int64_t)(2 * e + 1) * ld + env] = (T)uniform_pm(w[2], w[3], r);
  }
  for (int r = 0; r < 2 * A; ++r) vel[(int64_t)r * ld + env] = T(0);
  if (comm != nullptr)
    for (int r = 0; r < A * dim_c; ++r) comm[(int64_t)r * ld + env] = T(0);
}

template <typename T>
static int reset_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, uint64_t seed, uint32_t episode,
                      uint64_t env_id0, const uint8_t *mask, T *pos, T *vel, T *comm, void *stream) {
  int rc = check_common<T>(h, B, ld);
  if (rc) return rc;
  if (!pos || !vel) return MPE_ERR_NULL;
  if (misaligned<T>(pos) || misaligned<T>(vel) || misaligned<T>(comm)) return MPE_ERR_ALIGN;
  const dim3 grid((unsigned)((B + 255) / 256)), block(256);
  hipLaunchKernelGGL(reset_kernel<T>, grid, block, 0, (hipStream_t)stream, B, ld,
                     h->n_agents + h->n_landmarks, h->n_agents, h->dim_c, h->agent_pos_range,
                     h->landmark_pos_range, (uint32_t)seed, (uint32_t)(seed >> 32), episode, env_id0, mask, pos,
                     vel, comm);
  return (int)hipGetLastError();
}

// One workgroup per column; rows are summed in a fixed order (thread t takes rows t, t+256, ...
// sequentially, then a wavefront butterfly, then the 4 wavefront partials in a fixed order), so the
// result is bitwise reproducible for a given (rows, width).
__global__ __launch_bounds__(256) void stats_reduce_kernel(const double *stats, int64_t rows, int width,
                                                           double *out) {
  __shared__ double sp[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x;
  double s = 0.0;
  for (int64_t r = threadIdx.x; r < rows; r += 256) s += stats[r * width + c];
  s = group_sum(s, 64);
  if (lane == 0) sp[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = (sp[0] + sp[1]) + (sp[2] + sp[3]);
}

extern "C" int mpe_stats_reduce(const double *stats, int64_t rows, int width, double *out, void *stream) {
  if (!stats || !out) return MPE_ERR_NULL;
  if (rows < 1 || width < 1) return MPE_ERR_ARG;
  hipLaunchKernelGGL(stats_reduce_kernel, dim3((unsigned)width), dim3(256), 0, (hipStream_t)stream, stats, rows,
                     width, out);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// exported typed entry points
// ------------------------------------------------------------------------------------------------
#define MPE_EXPORT_TYPED(SUF, T)                                                                               \
  extern "C" int mpe_world_step_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, const T *act,         \
                                      const T *noise, T *pos, T *vel, T *comm, void *stream) {                 \
    return world_step_impl<T>(d, B, ld, act, noise, pos, vel, comm, stream);                                   \
  }                                                                                                            \
  extern "C" int mpe_observe_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, const T *pos,            \
                                   const T *vel, T *obs, T *rew, T *info, void *stream) {                      \
    return observe_impl<T>(d, B, ld, pos, vel, obs, rew, info, stream);                                        \
  }                                                                                                            \
  extern "C" int mpe_env_step_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, const T *act,           \
                                    const T *noise, T *pos, T *vel, T *comm, T *obs, T *rew, T *info,          \
                                    double *stats, void *stream) {                                             \
    return env_step_impl<T>(d, B, ld, act, noise, pos, vel, comm, obs, rew, info, stats, stream);              \
  }                                                                                                            \
  extern "C" int mpe_reset_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, uint64_t seed,             \
                                 uint32_t episode, uint64_t env_id0, const uint8_t *mask, T *pos, T *vel,      \
                                 T *comm, void *stream) {                                                      \
    return reset_impl<T>(d, B, ld, seed, episode, env_id0, mask, pos, vel, comm, stream);                      \
  }

MPE_EXPORT_TYPED(f32, float)
MPE_EXPORT_TYPED(f64, double)
