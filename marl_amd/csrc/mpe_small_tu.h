// Translation-unit body of the register-kernel launchers for the stock scenarios (simple_tag, simple_spread, simple,
// simple_adversary, simple_push, simple_speaker_listener, simple_reference, simple_crypto, simple_world_comm):
// included by mpe_small_f32.hip / mpe_small_f64.hip with MPE_TU_T = float / double.
#include <hip/hip_runtime.h>

#include "mpe_launch.h"
#include "mpe_comm_kernel.h"
#include "mpe_wcomm_kernel.h"
#include "mpe_small_kernel.h"
#include "mpe_split_kernel.h"

namespace mpe {

template <typename T, int KIND, int A, int L, int NADV, bool DO_STEP, bool DO_OBS>
static int launch_small(const KParams<T> &p, int noise, hipStream_t s) {
  constexpr int THREADS = kSmallThreads;
  const dim3 grid((unsigned)((p.B + THREADS - 1) / THREADS)), block(THREADS);
  if constexpr (DO_STEP) {  // noise: 0 none, 1 the caller's normals, 2 drawn in the kernel
    if (noise == 1) {
      hipLaunchKernelGGL((small_kernel<T, KIND, A, L, NADV, THREADS, 1, DO_STEP, DO_OBS>), grid, block, 0, s, p);
      return (int)hipGetLastError();
    }
    if (noise == 2) {
      hipLaunchKernelGGL((small_kernel<T, KIND, A, L, NADV, THREADS, 2, DO_STEP, DO_OBS>), grid, block, 0, s, p);
      return (int)hipGetLastError();
    }
  }
  if constexpr (DO_STEP && DO_OBS && L > 0) {  // the fused step has a variant that streams the landmark rows (residency_plan)
    if (resid_variant(p.resid) >= 3) {
      hipLaunchKernelGGL((small_kernel<T, KIND, A, L, NADV, THREADS, 0, DO_STEP, DO_OBS, 3>), grid, block, 0, s, p);
      return (int)hipGetLastError();
    }
  }
  hipLaunchKernelGGL((small_kernel<T, KIND, A, L, NADV, THREADS, 0, DO_STEP, DO_OBS>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

// small batches: one agent per wavefront, 64 envs per workgroup (mpe_split_kernel.h)
template <typename T, int KIND, int A, int L, int NADV, bool DO_STEP, bool DO_OBS>
static int launch_split(const KParams<T> &p, int noise, hipStream_t s) {
  const dim3 grid((unsigned)((p.B + kSplitEnvs - 1) / kSplitEnvs)), block(kSplitEnvs * A);
  if constexpr (DO_STEP) {
    if (noise == 1) {
      hipLaunchKernelGGL((split_kernel<T, KIND, A, L, NADV, 1, DO_STEP, DO_OBS>), grid, block, 0, s, p);
      return (int)hipGetLastError();
    }
    if (noise == 2) {
      hipLaunchKernelGGL((split_kernel<T, KIND, A, L, NADV, 2, DO_STEP, DO_OBS>), grid, block, 0, s, p);
      return (int)hipGetLastError();
    }
  }
  hipLaunchKernelGGL((split_kernel<T, KIND, A, L, NADV, 0, DO_STEP, DO_OBS>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

template <typename T, int KIND, int C, bool DO_STEP, bool DO_OBS>
static int launch_comm(const KParams<T> &p, hipStream_t s) {
  const dim3 grid((unsigned)((p.B + kSmallThreads - 1) / kSmallThreads)), block(kSmallThreads);
  hipLaunchKernelGGL((comm_kernel<T, KIND, C, kSmallThreads, DO_STEP, DO_OBS>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

template <typename T, bool DO_STEP, bool DO_OBS>
int launch_stock_id(SmallId sv, const KParams<T> &p, int noise, bool split, hipStream_t s) {
  const dim3 grid((unsigned)((p.B + kSmallThreads - 1) / kSmallThreads)), block(kSmallThreads);
  switch (sv) {
    case COMM_SPEAKER_LISTENER: return launch_comm<T, MPE_SPEAKER_LISTENER, 3, DO_STEP, DO_OBS>(p, s);
    case COMM_REFERENCE: return launch_comm<T, MPE_SIMPLE_REFERENCE, 10, DO_STEP, DO_OBS>(p, s);
    case WCOMM_4_2_1_2_2:
      if (DO_STEP && DO_OBS && resid_variant(p.resid) >= 3)
        hipLaunchKernelGGL((wcomm_kernel<T, kSmallThreads, DO_STEP, DO_OBS, (DO_STEP && DO_OBS) ? 3 : 0>), grid, block, 0, s, p);
      else
        hipLaunchKernelGGL((wcomm_kernel<T, kSmallThreads, DO_STEP, DO_OBS>), grid, block, 0, s, p);
      return (int)hipGetLastError();
    case COMM_CRYPTO:
      hipLaunchKernelGGL((crypto_kernel<T, kSmallThreads, DO_STEP, DO_OBS>), grid, block, 0, s, p);
      return (int)hipGetLastError();
    case SMALL_TAG_3_1_2:
      if (split) return launch_split<T, MPE_SIMPLE_TAG, 4, 2, 3, DO_STEP, DO_OBS>(p, noise, s);
      return launch_small<T, MPE_SIMPLE_TAG, 4, 2, 3, DO_STEP, DO_OBS>(p, noise, s);
    case SMALL_SPREAD_3_3:
      if (split) return launch_split<T, MPE_SIMPLE_SPREAD, 3, 3, 0, DO_STEP, DO_OBS>(p, noise, s);
      return launch_small<T, MPE_SIMPLE_SPREAD, 3, 3, 0, DO_STEP, DO_OBS>(p, noise, s);
    case SMALL_SIMPLE_1_1: return launch_small<T, MPE_SIMPLE, 1, 1, 0, DO_STEP, DO_OBS>(p, noise, s);
    case SMALL_ADVERSARY_3_2: return launch_small<T, MPE_SIMPLE_ADVERSARY, 3, 2, 1, DO_STEP, DO_OBS>(p, noise, s);
    case SMALL_PUSH_10_1:  // f64 needs > 256 VGPRs: the host routes it to the tile kernel (small_launch_threads)
      if constexpr (sizeof(T) == 4) return launch_small<T, MPE_SIMPLE_PUSH, 10, 1, 9, DO_STEP, DO_OBS>(p, noise, s);
      return MPE_ERR_UNSUPPORTED;
    default: return MPE_ERR_UNSUPPORTED;
  }
}

template int launch_stock_id<MPE_TU_T, true, false>(SmallId, const KParams<MPE_TU_T> &, int, bool, hipStream_t);
template int launch_stock_id<MPE_TU_T, false, true>(SmallId, const KParams<MPE_TU_T> &, int, bool, hipStream_t);
template int launch_stock_id<MPE_TU_T, true, true>(SmallId, const KParams<MPE_TU_T> &, int, bool, hipStream_t);

}  // namespace mpe
