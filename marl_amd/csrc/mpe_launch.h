// Kernel selection ids, launch geometry and the launcher templates shared by the translation units of
// libmpe_hip.so.  The kernels are instantiated in their own .hip files (mpe_tile_f32.hip, mpe_tile_f64.hip,
// mpe_small_f32.hip, mpe_small_f64.hip) so that the library builds in parallel; mpe_hip.hip holds the C ABI,
// the validation and the selection logic and only calls the launchers declared here.
#pragma once
#include "mpe_device.h"

namespace mpe {

enum SmallId {
  SMALL_NONE = 0, SMALL_TAG_3_1_2, SMALL_SPREAD_3_3, SMALL_SIMPLE_1_1, SMALL_ADVERSARY_3_2, SMALL_PUSH_10_1,
  // command-hierarchy worlds, HIER_<agents>_<targets>_<followers> (mpe_hier_kernel.h)
  HIER_FIRST,
  HIER_3_3_2 = HIER_FIRST,  // L1F2_3Targets_*
  HIER_5_3_4,               // L1F4_3Targets_wide
  HIER_2_10_1,              // L1F1_10Targets_wide
  HIER_3_10_2,              // L1F2_10Targets_wide
  HIER_3_10_1,              // S1L1F1_10Targets_wide
  HIER_5_3_2,               // S1L2F2_3Targets_*
  HIER_5_10_2,              // S1L2F2_10Targets_wide(_rewardFixed)
  HIER_LAST = HIER_5_10_2,
  // speaking stock scenarios (mpe_comm_kernel.h)
  COMM_SPEAKER_LISTENER,
  COMM_REFERENCE,
  COMM_CRYPTO,
  WCOMM_4_2_1_2_2,  // simple_world_comm (mpe_wcomm_kernel.h)
  MOVING_6_4_3,     // L1F4_MovingTarget (agents, followers, dim_c; mpe_hier_kernel.h:moving_kernel)
  MOVING_8_4_7      // S1L2F4_MovingTarget
};

constexpr int kSmallThreads = 256;  // workgroup of the one-env-per-lane register kernels
// (simple_tag 1 Mi envs f32, r2: 128 lanes 86.6 us, 256 lanes 78.9, 512 lanes 85.8)
// Small batches of simple_tag / simple_spread (B <= MPE_SPLIT_MAX_ENVS, include/mpe_hip.h) run one AGENT per wavefront
// instead (mpe_split_kernel.h): a workgroup is 64 envs x A wavefronts.  Well below one wavefront per SIMD the
// one-env-per-lane kernel is bound by the latency of its own instruction stream, and the split form shortens it by ~A.
constexpr int kSplitEnvs = 64;
inline bool small_id_splits(SmallId sv, int64_t B, uint32_t flags) {
  if (sv != SMALL_TAG_3_1_2 && sv != SMALL_SPREAD_3_3) return false;
  if (flags & MPE_FLAG_SPLIT) return true;
  return !(flags & MPE_FLAG_WIDE) && B <= (int64_t)MPE_SPLIT_MAX_ENVS;
}
// Workgroup of the one-env-per-lane hierarchy / moving-target kernels in f64 (statistics rows follow it)
#ifndef MPE_HIER_F64_THREADS
#define MPE_HIER_F64_THREADS 256
#endif
constexpr int hier_threads(int dtype_bytes) { return dtype_bytes == 8 ? MPE_HIER_F64_THREADS : kSmallThreads; }
constexpr int kHierDimC = 7;        // stock shape of the L1F* / S1L* Targets files
constexpr int kHierTargets = 3;

// ---- LDS-tile kernel geometry (mpe_tile_kernel.h) ------------------------------------------------------------
#ifndef MPE_TILE_F32
#define MPE_TILE_F32 64
#endif
template <typename T>
constexpr int tile_envs() { return sizeof(T) == 4 ? MPE_TILE_F32 : 32; }

inline int64_t tile_grid(int64_t B, int tile) {
  const int64_t nt = (B + tile - 1) / tile;
  return nt < 16384 ? nt : 16384;
}

// Runs the generic LDS-tile kernel (any A <= 64, E <= 128) for this world.  Defined in mpe_tile_f32.hip /
// mpe_tile_f64.hip for (DO_STEP, DO_OBS) = (1,0), (0,1), (1,1).
template <typename T, bool DO_STEP, bool DO_OBS>
int launch_tile_world(const mpe_scenario_desc *h, const KParams<T> &p, hipStream_t s);

// Run the register kernel `sv` (one env per lane, stock shapes): the stock scenarios' kernels are defined in
// mpe_small_f32.hip / mpe_small_f64.hip, the command-hierarchy ones in mpe_hier_f32.hip / mpe_hier_f64.hip.
template <typename T, bool DO_STEP, bool DO_OBS>
int launch_stock_id(SmallId sv, const KParams<T> &p, int noise, bool split, hipStream_t s);
template <typename T, bool DO_STEP, bool DO_OBS>
int launch_hier_id(SmallId sv, const KParams<T> &p, bool split, hipStream_t s);
inline bool small_id_is_hier(SmallId sv) { return (sv >= HIER_FIRST && sv <= HIER_LAST) || sv == MOVING_6_4_3 || sv == MOVING_8_4_7; }
// The command-hierarchy worlds have a one-agent-per-wavefront form too (mpe_hsplit_kernel.h): 64 * kHierSplitEw envs x A
// wavefronts per workgroup.  MPE_FLAG_SPLIT asks for it, MPE_FLAG_WIDE for the one-env-per-lane form; otherwise it runs
// for B <= MPE_HIER_SPLIT_MAX_ENVS.  Kernel us per launch, MI355X r3, one env per lane -> one agent per wavefront
// (tools/hsplit_ab.sh, profiles/r3_hsplit_ab.txt):
//   S1L2F2_10Targets f64  4 096 envs 39.6 -> 14.8   16 384: 40.6 -> 17.1   65 536: 45.1 -> 37.4   262 144: 124.8 -> 135.0
//   S1L2F2_10Targets f32  4 096 envs 19.3 -> 10.0   16 384: 20.0 -> 11.0   65 536: 22.7 -> 18.4   262 144:  58.9 ->  65.6
//   L1F2_3Targets    f64  4 096 envs 13.5 ->  7.9   16 384: 14.3 ->  9.1   65 536: 17.3 -> 14.7   262 144:  47.5 ->  51.8
//   S1L2F4_Moving    f64  4 096 envs 31.1 -> 11.0   16 384: 32.3 -> 16.7   65 536: 50.3 -> 42.4   262 144: 154.1 -> 163.0
// and at 1 Mi envs 456 -> 515 us (S1L2F2_10Targets f64): the barriers lock the wavefronts of a workgroup into the same
// phase (all loading, all in the pair forces, all storing), so loads, arithmetic and stores overlap less than between the
// free-running wavefronts of the one-env-per-lane form.  Two env groups of 64 per workgroup (1 KiB row segments in f64)
// were worse still (699 us): fewer, larger phase-locked groups per CU.
#ifndef MPE_HSPLIT_EW
#define MPE_HSPLIT_EW 1
#endif
constexpr int kHierSplitEw = MPE_HSPLIT_EW;
constexpr int hier_split_ew(int agents) {  // env groups of 64 per workgroup: at most 1024 lanes
  return 64 * kHierSplitEw * agents <= 1024 ? kHierSplitEw : (128 * agents <= 1024 ? 2 : 1);
}
inline int hier_id_agents(SmallId sv) {
  switch (sv) {
    case HIER_2_10_1: return 2;
    case HIER_3_3_2: case HIER_3_10_2: case HIER_3_10_1: return 3;
    case HIER_5_3_4: case HIER_5_3_2: case HIER_5_10_2: return 5;
    case MOVING_6_4_3: return 6;
    default: return 8;
  }
}
inline int hier_split_envs(SmallId sv) { return 64 * hier_split_ew(hier_id_agents(sv)); }
inline bool hier_id_splits(SmallId sv, int64_t B, uint32_t flags) {
  if (!small_id_is_hier(sv)) return false;
  if (flags & MPE_FLAG_SPLIT) return true;
  return !(flags & MPE_FLAG_WIDE) && B <= (int64_t)MPE_HIER_SPLIT_MAX_ENVS;
}

}  // namespace mpe
