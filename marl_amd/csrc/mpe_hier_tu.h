// Translation-unit body of the register-kernel launchers for the fork's command-hierarchy worlds (L1F* / S1L*
// Targets files and the two *_MovingTarget files): included by mpe_hier_f32.hip / mpe_hier_f64.hip with
// MPE_TU_T = float / double.
#include <hip/hip_runtime.h>

#include "mpe_launch.h"
#include "mpe_hier_kernel.h"
#include "mpe_hsplit_kernel.h"

namespace mpe {

template <typename T, int A, int L, int NF, bool DO_STEP, bool DO_OBS>
static int launch_hier(const KParams<T> &p, hipStream_t s) {
  constexpr int TH = hier_threads((int)sizeof(T));
  const dim3 grid((unsigned)((p.B + TH - 1) / TH)), block(TH);
  // the fused step has one variant per Infinity Cache plan (mpe_device.h:residency_plan)
  const int res = (DO_STEP && DO_OBS) ? resid_variant(p.resid) : 0;
  if constexpr (DO_STEP && DO_OBS) {
    if (res == 4) hipLaunchKernelGGL((hier_kernel<T, A, L, NF, TH, DO_STEP, DO_OBS, 4>), grid, block, 0, s, p);
    else if (res == 3) hipLaunchKernelGGL((hier_kernel<T, A, L, NF, TH, DO_STEP, DO_OBS, 3>), grid, block, 0, s, p);
    else if (res == 2) hipLaunchKernelGGL((hier_kernel<T, A, L, NF, TH, DO_STEP, DO_OBS, 2>), grid, block, 0, s, p);
    else if (res == 1) hipLaunchKernelGGL((hier_kernel<T, A, L, NF, TH, DO_STEP, DO_OBS, 1>), grid, block, 0, s, p);
  }
  if (res == 0) hipLaunchKernelGGL((hier_kernel<T, A, L, NF, TH, DO_STEP, DO_OBS, 0>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

template <typename T, int A, int NF, int C, bool DO_STEP, bool DO_OBS>
static int launch_moving(const KParams<T> &p, hipStream_t s) {
  constexpr int TH = hier_threads((int)sizeof(T));
  const dim3 grid((unsigned)((p.B + TH - 1) / TH)), block(TH);
  const int res = (DO_STEP && DO_OBS) ? resid_variant(p.resid) : 0;
  if constexpr (DO_STEP && DO_OBS) {  // no landmarks: variants 2 and 3 are the same kernel
    if (res == 4) hipLaunchKernelGGL((moving_kernel<T, A, NF, C, TH, DO_STEP, DO_OBS, 4>), grid, block, 0, s, p);
    else if (res >= 2) hipLaunchKernelGGL((moving_kernel<T, A, NF, C, TH, DO_STEP, DO_OBS, 2>), grid, block, 0, s, p);
    else if (res == 1) hipLaunchKernelGGL((moving_kernel<T, A, NF, C, TH, DO_STEP, DO_OBS, 1>), grid, block, 0, s, p);
  }
  if (res == 0) hipLaunchKernelGGL((moving_kernel<T, A, NF, C, TH, DO_STEP, DO_OBS, 0>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

template <typename T, int A, int L, int NF, int NG, int C, bool MOVING, bool DO_STEP, bool DO_OBS>
static int launch_hier_split(const KParams<T> &p, hipStream_t s) {
  constexpr int EW = hier_split_ew(A);
  const dim3 grid((unsigned)((p.B + 64 * EW - 1) / (64 * EW))), block(64 * EW * A);
  hipLaunchKernelGGL((hier_split_kernel<T, A, L, NF, NG, C, MOVING, EW, DO_STEP, DO_OBS>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

template <typename T, bool DO_STEP, bool DO_OBS>
static int launch_hier_split_id(SmallId sv, const KParams<T> &p, hipStream_t s) {
  constexpr int C = kHierDimC;
  switch (sv) {
    case HIER_3_3_2: return launch_hier_split<T, 3, 3, 2, 0, C, false, DO_STEP, DO_OBS>(p, s);
    case HIER_5_3_4: return launch_hier_split<T, 5, 3, 4, 0, C, false, DO_STEP, DO_OBS>(p, s);
    case HIER_2_10_1: return launch_hier_split<T, 2, 10, 1, 0, C, false, DO_STEP, DO_OBS>(p, s);
    case HIER_3_10_2: return launch_hier_split<T, 3, 10, 2, 0, C, false, DO_STEP, DO_OBS>(p, s);
    case HIER_3_10_1: return launch_hier_split<T, 3, 10, 1, 0, C, false, DO_STEP, DO_OBS>(p, s);
    case HIER_5_3_2: return launch_hier_split<T, 5, 3, 2, 0, C, false, DO_STEP, DO_OBS>(p, s);
    case HIER_5_10_2: return launch_hier_split<T, 5, 10, 2, 0, C, false, DO_STEP, DO_OBS>(p, s);
    case MOVING_6_4_3: return launch_hier_split<T, 6, 0, 4, 1, 3, true, DO_STEP, DO_OBS>(p, s);
    case MOVING_8_4_7: return launch_hier_split<T, 8, 0, 4, 1, 7, true, DO_STEP, DO_OBS>(p, s);
    default: return MPE_ERR_UNSUPPORTED;
  }
}

template <typename T, bool DO_STEP, bool DO_OBS>
int launch_hier_id(SmallId sv, const KParams<T> &p, bool split, hipStream_t s) {
  if (split) return launch_hier_split_id<T, DO_STEP, DO_OBS>(sv, p, s);
  switch (sv) {
    case HIER_3_3_2: return launch_hier<T, 3, 3, 2, DO_STEP, DO_OBS>(p, s);
    case HIER_5_3_4: return launch_hier<T, 5, 3, 4, DO_STEP, DO_OBS>(p, s);
    case HIER_2_10_1: return launch_hier<T, 2, 10, 1, DO_STEP, DO_OBS>(p, s);
    case HIER_3_10_2: return launch_hier<T, 3, 10, 2, DO_STEP, DO_OBS>(p, s);
    case HIER_3_10_1: return launch_hier<T, 3, 10, 1, DO_STEP, DO_OBS>(p, s);
    case HIER_5_3_2: return launch_hier<T, 5, 3, 2, DO_STEP, DO_OBS>(p, s);
    case HIER_5_10_2: return launch_hier<T, 5, 10, 2, DO_STEP, DO_OBS>(p, s);
    case MOVING_6_4_3: return launch_moving<T, 6, 4, 3, DO_STEP, DO_OBS>(p, s);
    case MOVING_8_4_7: return launch_moving<T, 8, 4, 7, DO_STEP, DO_OBS>(p, s);
    default: return MPE_ERR_UNSUPPORTED;
  }
}

template int launch_hier_id<MPE_TU_T, true, false>(SmallId, const KParams<MPE_TU_T> &, bool, hipStream_t);
template int launch_hier_id<MPE_TU_T, false, true>(SmallId, const KParams<MPE_TU_T> &, bool, hipStream_t);
template int launch_hier_id<MPE_TU_T, true, true>(SmallId, const KParams<MPE_TU_T> &, bool, hipStream_t);

}  // namespace mpe
