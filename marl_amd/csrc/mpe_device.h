// Device-side building blocks shared by the MPE kernels (gfx950 only; wave = 64 lanes).
//
// Arithmetic follows the reference's float64 operation order (core.py:233-249, :211-222;
// simple_tag.py:69-73,103-108) so that the f64 instantiation reproduces numpy to the last bits
// that the device exp/log1p allow.  The library is compiled with -ffp-contract=off: no FMA is
// formed unless written explicitly.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mpe_hip.h"

namespace mpe {

// Device copy of mpe_scenario_desc in the arithmetic type T, passed BY VALUE in the kernarg
// segment (uniform indices become scalar loads; no device allocation, no global state).
template <typename T>
struct DevDesc {
  int32_t kind, A, L, E, dim_c, obs_total, info_w, n_adv;
  int32_t n_good;       // agents that are not adversaries
  int32_t ncol_prefix;  // >= 0: exactly the entities [0, ncol_prefix) collide (fast pair loop); -1: irregular
  T dt, damp1 /* 1 - damping */, cforce, k, inv_k /* RN(1 / k) */;
  T size[MPE_MAX_ENTITIES];
  T mass[MPE_MAX_ENTITIES];       // agents; landmarks too when one of them is movable
  T accel[MPE_MAX_AGENTS];
  T max_speed[MPE_MAX_ENTITIES];  // < 0: None; entries >= A: movable landmarks (Entity.max_speed)
  int32_t lm_movable;             // some landmark is movable: vel rows exist for all E entities (p.lvel)
  T u_noise[MPE_MAX_AGENTS];    // 0: None
  int32_t obs_row0[MPE_MAX_AGENTS];
  uint32_t movable[MPE_MAX_ENTITIES / 32];
  uint32_t collide[MPE_MAX_ENTITIES / 32];
  uint32_t adversary[MPE_MAX_AGENTS / 32];
  uint8_t adv_list[MPE_MAX_AGENTS];   // agent indices of the adversaries, ascending
  uint8_t good_list[MPE_MAX_AGENTS];  // agent indices of the good agents, ascending
  // fork's command hierarchy (core.py:153-194; L1F*/S1L* scenarios)
  int32_t has_lforce;      // some agent has forced_id >= 0
  int32_t reward_variant, n_target_obs;
  int32_t n_food, n_forests;            // simple_world_comm: landmarks = plain, food, forests
  uint32_t obs_lf[MPE_MAX_AGENTS / 32]; // agents whose observation carries their leader_force
  T obs_r[MPE_MAX_AGENTS];              // observation radius (the *_MovingTarget files compare distances, not squares)
  T c_noise[MPE_MAX_AGENTS];            // agent.c_noise, 0 = None (core.py:229)
  uint32_t leader[MPE_MAX_AGENTS / 32];
  int32_t act_row0[MPE_MAX_AGENTS];  // first action row of each agent (5 rows if movable, + dim_c if not silent)
  int32_t cact_row0[MPE_MAX_AGENTS]; // first comm row inside the action (= act_row0 + 5 for a movable agent)
  uint32_t silent[MPE_MAX_AGENTS / 32];
  uint32_t advleader[MPE_MAX_AGENTS / 32];
  uint32_t advsuper[MPE_MAX_AGENTS / 32];
  int8_t force_id[MPE_MAX_AGENTS];
  int8_t forced_id[MPE_MAX_AGENTS];
  T obs_r2[MPE_MAX_AGENTS];  // obs_r * obs_r, evaluated in T (L1F2_3Targets_100observe.py:206)
};

// desc_i8 / desc_u8 (below) read the byte arrays as aligned 32-bit words
static_assert(offsetof(DevDesc<float>, adv_list) % 4 == 0 && offsetof(DevDesc<float>, good_list) % 4 == 0 &&
                  offsetof(DevDesc<float>, force_id) % 4 == 0 && offsetof(DevDesc<float>, forced_id) % 4 == 0 &&
                  offsetof(DevDesc<double>, adv_list) % 4 == 0 && offsetof(DevDesc<double>, good_list) % 4 == 0 &&
                  offsetof(DevDesc<double>, force_id) % 4 == 0 && offsetof(DevDesc<double>, forced_id) % 4 == 0 &&
                  MPE_MAX_AGENTS % 4 == 0,
              "byte arrays of the descriptor start on word boundaries");

// entity sizes alone (kernels that need nothing else of the descriptor)
template <typename T>
struct DevSizes {
  T size[MPE_MAX_ENTITIES];
};

template <typename T>
struct KParams {
  const T *act, *noise, *cnoise;
  const int32_t *goal;  // per-env goal landmark index (simple_adversary / simple_push), else unused
  T *pos, *vel, *comm, *lforce, *obs, *rew, *info;
  T *lvel;  // [L][2][ld] landmark velocities (worlds with a movable landmark), else nullptr
  double *stats;
  int64_t B, ld;
  uint32_t flags;  // mpe_buffers.flags (MPE_FLAG_*): the kernels read MPE_FLAG_SILENT_COMM_ZERO, MPE_FLAG_DEVICE_NOISE
  uint32_t resid;  // kResid*: which state rows of this launch bypass the Infinity Cache (residency_plan, below)
  // in-kernel u_noise / c_noise normals (MPE_FLAG_DEVICE_NOISE): Philox key, step counter (+ *nstep_dev when not
  // NULL: the word a HIP-graph replay moves), global id of env 0 of this call
  uint32_t nk0, nk1, nstep;
  const uint32_t *nstep_dev;
  uint64_t env_id0;
  DevDesc<T> d;
};

// Which state rows stay in the 256 MiB Infinity Cache between steps.  The env state is re-read by the next step; while
// all of it fits, every state load is a cache hit (simple_tag, 1 Mi envs f32: 84 MB), and rows a step only WRITES
// (leader_force: read by nobody but a later observe-only call) are overwritten in the cache step after step without
// ever reaching HBM.  Beyond the cache a sweep over all rows hits less and less, and it pays to send some rows past it
// with nontemporal accesses so that the others keep their place: first the write-only leader_force rows, then the
// READ-ONLY rows (fixed landmarks / targets), and the velocities when positions + velocities overflow the cache while
// the positions alone still fit (residency_plan, below).  Measured, one env per lane, plan on vs off on one box
// (tools/small_batch.sh, kernel us per step):
//   S1L2F2_10Targets f64 1 Mi envs (state 336 MB + 84 MB leader_force)  462 -> 427     2 Mi: 943 -> 894    f32 2 Mi: 420 -> 392
//   S1L2F4_MovingTarget f64 1 Mi envs (268 MB + 134 MB leader_force)     711 -> 637     2 Mi: 1479 -> 1413
//   simple_world_comm f64 1 Mi envs (285 MB)  444 -> 420      simple_tag f32 4 Mi envs (335 MB)  401 -> 326
//   simple_tag f64 2 Mi envs  383 -> 324       L1F2_3Targets f64 2 Mi envs  412 -> 394
// and where everything fits, streaming loses (S1L2F2_10Targets f64 512 Ki envs, targets streamed: 217.6 -> 226.0 us; f32
// 1 Mi envs, leader_force streamed: 186 -> 191; S1L2F4_MovingTarget f32 1 Mi envs: 305 -> 318), hence a plan per launch
// instead of a fixed policy.
constexpr uint32_t kResidNtLandmarks = 1u, kResidNtVel = 2u, kResidNtLforce = 4u, kResidNtComm = 8u;
#ifndef MPE_RESID_BUDGET_MB
#define MPE_RESID_BUDGET_MB 250
#endif
#ifndef MPE_RESID_SHED_MB
#define MPE_RESID_SHED_MB 300
#endif
// Kernel variants (template parameter RES of the fused one-env-per-lane kernels): 0 everything cached, 1 the speaking
// agents' comm rows streamed (write-only, like leader_force), 2 + leader_force, 3 + landmarks, 4 + velocities.
inline int resid_variant(uint32_t plan) {
  return (plan & kResidNtVel) ? 4 : ((plan & kResidNtLandmarks) ? 3 : ((plan & kResidNtLforce) ? 2 : ((plan & kResidNtComm) ? 1 : 0)));
}
inline uint32_t residency_plan(int A, int L, bool has_lforce, int comm_rows, int64_t ld, int elem_bytes) {
  const double row = (double)ld * elem_bytes, mib = 1048576.0;
  const double budget = MPE_RESID_BUDGET_MB * mib;  // what read-only and write-only rows may fill up
  const double shed = MPE_RESID_SHED_MB * mib;      // beyond this much read-write state the velocities are streamed
  const double pos = 2.0 * A * row, vel = 2.0 * A * row, lm = 2.0 * L * row;
  const double lf = has_lforce ? 2.0 * A * row : 0.0, comm = comm_rows * row;
  uint32_t plan = 0;
  // velocities: streamed when positions + velocities overflow the cache but the positions alone still (mostly) fit.
  // S1L2F4_MovingTarget f64, streamed vs cached: 1 Mi envs (134 + 134 MB) 679 vs 636 us -- the cache still serves most of
  // a 268 MB sweep --, 2 Mi envs (268 + 268 MB) 1413 vs 1479; S1L2F2_10Targets f64 2 Mi envs (168 + 168 MB) 894 vs 934;
  // positions alone past the cache (S1L2F4_MovingTarget f64 4 Mi envs, 537 MB): 3400 vs 3345, nothing left to protect.
  if (pos + vel > shed && pos <= shed) plan |= kResidNtVel;
  const double used = pos + ((plan & kResidNtVel) ? 0.0 : vel);
  if (L > 0 && used + lm > budget) plan |= kResidNtLandmarks;  // read-only rows before read-write ones
  const double cached = used + ((plan & kResidNtLandmarks) ? 0.0 : lm);
  // write-only rows go first of all: the comm rows (S1L2F2_10Targets f32: cached vs streamed 98.7 vs 101.6 us at 512 Ki envs,
  // where everything fits, 202 vs 188 at 1 Mi), then leader_force if that was not enough
  const bool shedding = plan != 0;  // landmarks or velocities are streamed already
  if (comm_rows > 0 && (shedding || cached + lf + comm > budget)) plan |= kResidNtComm;
  if (has_lforce && (shedding || cached + lf > budget)) plan |= kResidNtLforce;
  return plan;
}

// agent.state.c of a SILENT agent is zeros(dim_c) after every step (core.py:226-227).  A caller whose comm buffer already
// holds zeros in those rows says so with MPE_FLAG_SILENT_COMM_ZERO and the step leaves them alone: in the hierarchy worlds
// they are 5-10 % of the rows a step writes and, as default-policy stores next to the nontemporal observation stream, cost
// twice their share (S1L2F4_MovingTarget 1 Mi envs f32: 361 -> 289 us, L1F2_3Targets f64: 212 -> 172 us).
template <typename T>
__device__ __forceinline__ bool rewrite_silent_comm(const KParams<T> &p) {
  return (p.flags & MPE_FLAG_SILENT_COMM_ZERO) == 0;
}

// Streaming (write-once) outputs -- observations, rewards, info -- use nontemporal stores so they do
// not displace the env state (pos/vel, re-read every step) from L2 / Infinity Cache.  Measured on
// MI355X with the simple_tag access shape (40 read rows + 86 write rows, 1M envs): 4.8 -> 7.4 TB/s
// (tools/stream_probe.hip).
template <typename T>
__device__ __forceinline__ void st_stream(T *p, T v) {
#ifdef MPE_NO_NT_STORE
  *p = v;
#else
  __builtin_nontemporal_store(v, p);
#endif
}
// Read-once inputs (actions, noise).
template <typename T>
__device__ __forceinline__ T ld_stream(const T *p) {
#ifdef MPE_NO_NT_LOAD
  return *p;
#else
  return __builtin_nontemporal_load(p);
#endif
}

// Env state (pos / vel): re-read by the next step, default cache policy.
template <typename T>
__device__ __forceinline__ T ld_state(const T *p) {
#ifdef MPE_NT_STATE_LOAD
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <typename T>
__device__ __forceinline__ void st_state(T *p, T v) {
#ifdef MPE_NT_STATE_STORE
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// A window on one SoA array for one workgroup of a one-env-per-lane kernel: `base` points at
// (row 0, the workgroup's first column); element (row r, lane) is base[r * ld + lane].
//
// Tail lanes carry the sentinel kNoLane: their loads read column 0 of the window (always a valid env)
// and their stores are skipped inside st(), so kernel bodies have no `if (active)` blocks of their
// own.  That matters for register pressure: with the stores of every agent wrapped in conditional
// blocks, the compiler kept each block's inputs alive across all of them (the 5-agent / 10-target
// hierarchy kernel needed 256 VGPRs = 1 wave/SIMD; 147 without).
//
// Cache policy: actions / noise are read once and observations / rewards / info are written once ->
// nontemporal; pos / vel are re-read by the next step -> default.  Measured on MI355X with the
// simple_tag access shape (40 read rows + 86 write rows, 1Mi envs): 4.9 -> 7.3 TB/s
// (tools/stream_probe.hip).  The same accesses through raw buffer resources (one VGPR offset + one
// SGPR row offset, no 64-bit VALU address arithmetic) stream at the same rate in the probe but ran
// 4-6 % slower inside these kernels, so plain global accesses it is.
constexpr uint32_t kNoLane = 0xFFFFFFFFu;

template <typename T>
struct Win {
  T *base;
  int64_t ld;

  __device__ __forceinline__ Win(const T *base_, int64_t ld_) : base(const_cast<T *>(base_)), ld(ld_) {}
#ifdef MPE_WIN_SADDR
  // row base pinned in SGPRs + zero-extended 32-bit lane byte offset -> `global_* v, v_off, s[base]`
  __device__ __forceinline__ T *at(int row, uint32_t lane) const {
    const uint64_t b = reinterpret_cast<uint64_t>(base + (int64_t)row * ld);
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    typedef __attribute__((address_space(1))) char gchar;
    gchar *g = reinterpret_cast<gchar *>(((uint64_t)bhi << 32) | blo);
    return (T *)(g + (uint64_t)(lane * (uint32_t)sizeof(T)));
  }
#else
  __device__ __forceinline__ T *at(int row, uint32_t lane) const { return base + (int64_t)row * ld + lane; }
#endif
  __device__ __forceinline__ T ld_state(int row, uint32_t lane) const {
    return mpe::ld_state(at(row, lane == kNoLane ? 0u : lane));
  }
  __device__ __forceinline__ T ld_stream(int row, uint32_t lane) const {
    return mpe::ld_stream(at(row, lane == kNoLane ? 0u : lane));
  }
  __device__ __forceinline__ void st_state(int row, uint32_t lane, T v) const {
    if (lane != kNoLane) mpe::st_state(at(row, lane), v);
  }
  __device__ __forceinline__ void st_stream(int row, uint32_t lane, T v) const {
    if (lane != kNoLane) mpe::st_stream(at(row, lane), v);
  }
  // state rows whose cache policy the launch decides (residency_plan): NT is a compile-time constant of the kernel
  // variant.  (A run-time `nt ? ld_stream : ld_state` does not survive the compiler: the two accesses to one address are
  // merged into one plain access and the hint is gone.)
  template <bool NT>
  __device__ __forceinline__ T ld_state_nt(int row, uint32_t lane) const {
    if constexpr (NT) return ld_stream(row, lane);
    else return ld_state(row, lane);
  }
  template <bool NT>
  __device__ __forceinline__ void st_state_nt(int row, uint32_t lane, T v) const {
    if constexpr (NT) st_stream(row, lane, v);
    else st_state(row, lane, v);
  }
};

__device__ __forceinline__ bool bit(const uint32_t *w, int i) { return (w[i >> 5] >> (i & 31)) & 1u; }

// An entry of one of the descriptor's BYTE arrays (force_id, forced_id, adv_list, good_list) at a run-time index: read
// as the aligned 32-bit word that holds it.  With a plain byte access the compiler keeps `kernarg + i` as a scalar base
// and reuses it, plus 3 i, for the 4-byte arrays indexed by the same i (s_load_dword s, s[kernarg + i], s[3 i] offset:...),
// but the scalar memory unit drops the two low bits of the BASE before it adds the offsets (tools/smem_align_probe.hip,
// MI355X: base t + 1, offset 3 reads entry 0, not entry 1): agent 1 then got agent 0's 4-byte constants.  Seen in the f32
// observe-only instantiation of mpe_hsplit_kernel.h; every kernel reads these arrays through here since.
__device__ __forceinline__ int desc_i8(const int8_t *arr, int i) {
  const uint32_t w = reinterpret_cast<const uint32_t *>(arr)[i >> 2];
  return (int)(int8_t)(w >> (8 * (i & 3)));
}
__device__ __forceinline__ int desc_u8(const uint8_t *arr, int i) {
  const uint32_t w = reinterpret_cast<const uint32_t *>(arr)[i >> 2];
  return (int)((w >> (8 * (i & 3))) & 0xFFu);
}

template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_log1p(T x);
template <> __device__ __forceinline__ float t_log1p<float>(float x) { return log1pf(x); }
template <> __device__ __forceinline__ double t_log1p<double>(double x) { return log1p(x); }
// np.linalg.norm of a 2-vector (core.py:185,188) = sqrt(dot(x, x)); the BLAS dot behind it rounds as
// fma(y, y, x*x) (oracle/mpe_oracle.py:_norm2_blas); f32 mode: plain.
template <typename T> __device__ __forceinline__ T t_norm2(T x, T y);
template <> __device__ __forceinline__ float t_norm2<float>(float x, float y) { return __builtin_amdgcn_sqrtf(x * x + y * y); }
template <> __device__ __forceinline__ double t_norm2<double>(double x, double y) { return sqrt(fma(y, y, x * x)); }
template <typename T> __device__ __forceinline__ T t_abs(T x);
template <> __device__ __forceinline__ float t_abs<float>(float x) { return fabsf(x); }
template <> __device__ __forceinline__ double t_abs<double>(double x) { return fabs(x); }

// get_collision_force (core.py:239-246): force on entity a from the pair (a,b), d = pos_a - pos_b.
//   dist = sqrt(dx^2+dy^2); pen = logaddexp(0, -(dist-dmin)/k)*k; f = cforce*d/dist*pen
// np.logaddexp(0,x) == max(x,0) + log1p(exp(-|x|)).
//
// f64: the reference's operations in the reference's order (IEEE divide / sqrt, ocml exp / log1p).
template <typename T>
__device__ __forceinline__ void pair_force(T dx, T dy, T dmin, const DevDesc<T> &d, T &fx, T &fy) {
  const T dist = t_sqrt<T>(dx * dx + dy * dy);
  const T x = -(dist - dmin) / d.k;
  const T pen = ((x > T(0) ? x : T(0)) + t_log1p<T>(t_exp<T>(-t_abs<T>(x)))) * d.k;
  fx = (d.cforce * dx) / dist * pen;
  fy = (d.cforce * dy) / dist * pen;
}

// ---- f64 pair force without the library calls -------------------------------------------------------------------
// The generic form above costs ~270 instructions per pair in f64, half of them ocml's log1p (~140; exp ~50, the
// scaled sqrt 17, three IEEE divides 11 each): simple_tag f64 was 3 200 instructions per wavefront and ran at 0.51
// VALU utilisation beside the HBM stream, the 10-target hierarchy world (60 pairs) was compute-bound outright.  The
// pieces below use what this call site knows -- exp of a non-positive number, log1p of a number in [0, 1], a
// denominator that is a square root just computed, a divisor that is a launch constant -- and come to ~95.
// Accuracy (host emulation over 2e7 samples with 2^-23 seeds, tools/f64_math_check.c; GPU: tests/test_gpu_parity.py):
// the square root and both quotient forms are correctly rounded (equal to IEEE sqrt / divide on every sample),
// exp <= 0.88 ulp, log1p <= 0.83 ulp; numpy's own libm is within 1 ulp on the same two, so the distance to the
// reference stays where it was (~1e-15 after 25 steps; bar 1e-9).
#ifndef MPE_F64_RCP_NR
#define MPE_F64_RCP_NR 2  // Newton steps on v_rcp_f64 (1 would do if the seed is good to 2^-27; 2 is safe from 2^-14)
#endif
__device__ __forceinline__ double rcp_nr(double b) {
  double y = __builtin_amdgcn_rcp(b);
#pragma unroll
  for (int i = 0; i < MPE_F64_RCP_NR; ++i) {
    const double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
  }
  return y;
}
// exp(-a) for a >= 0.  n = rint(-a log2 e), r = -a - n ln2 (two-piece ln2, |r| <= 0.347), Taylor to r^13
// (truncation 4e-18), ldexp.  a is clamped at 800 (exp(-745.2) already rounds to 0; the clamp keeps n and r finite for
// a = inf and turns a NaN into 0: the caller's NaN reaches the force through the division by dist).
__device__ __forceinline__ double exp_neg_f64(double a) {
  const double y = -fmin(a, 800.0);
  const double n = rint(y * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, y);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}
// log1p(t) for t in [0, 1] (fdlibm's s_log1p.c scheme: u = 1 + t with its rounding error c carried as c / u, u = 2^k m
// with m in [sqrt(1/2), sqrt 2), f = m - 1, s = f / (2 + f), log m = f - f^2/2 + s (f^2/2 + R(s^2))).  Both quotients go
// through refined reciprocals: s is multiplied by an O(f^2) term, and c / u is the whole result when t < 2^-53.
__device__ __forceinline__ double log1p_unit_f64(double t) {
  const double u = 1.0 + t;
  const double c = (t - (u - 1.0)) * rcp_nr(u);
  const bool up = u >= 1.4142135623730951;
  const double kd = up ? 1.0 : 0.0;
  const double f = u * (up ? 0.5 : 1.0) - 1.0;
  const double hfsq = 0.5 * f * f;
  const double s = f * rcp_nr(2.0 + f), z = s * s;
  double R = fma(z, 1.479819860511658591e-01, 1.531383769920937332e-01);
  R = fma(z, R, 1.818357216161805012e-01);
  R = fma(z, R, 2.222219843214978396e-01);
  R = fma(z, R, 2.857142874366239149e-01);
  R = fma(z, R, 3.999999999940941908e-01);
  R = fma(z, R, 6.666666666666735130e-01);
  R = R * z;
  const double lo = fma(kd, 1.90821492927058770002e-10, c);
  return fma(kd, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, lo)) - f));
}
// sqrt(x) and 1 / sqrt(x) from one v_rsq_f64: the compiler's own sqrt expansion (one Goldschmidt step, two residual
// corrections) without its input scaling, which only matters below 2^-767; x = 0 gives NaN, which is what the caller's
// 0 / 0 gives the reference.  rinv is good to ~2^-50: a seed for quot().
__device__ __forceinline__ void sqrt_rcp_f64(double x, double &root, double &rinv) {
  const double y0 = __builtin_amdgcn_rsq(x);
  double g = x * y0, h = 0.5 * y0;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double e = fma(-g, g, x);
  g = fma(e, h, g);
  e = fma(-g, g, x);
  root = fma(e, h, g);
  rinv = h + h;
}
// n / b given y ~ 1 / b: one residual step (Markstein).  With y = RN(1/b) the result is the correctly rounded quotient
// (Brisebarre, Muller, Raina 2004: division by a constant); with a ~2^-50 seed it still was on every sample tried.
__device__ __forceinline__ double quot(double n, double b, double y) {
  const double q = n * y;
  return fma(fma(-q, b, n), y, q);
}
template <>
__device__ __forceinline__ void pair_force<double>(double dx, double dy, double dmin, const DevDesc<double> &d,
                                                   double &fx, double &fy) {
#ifdef MPE_F64_OCML
  const double dist = sqrt(dx * dx + dy * dy);
  const double x = -(dist - dmin) / d.k;
  const double pen = ((x > 0.0 ? x : 0.0) + log1p(exp(-fabs(x)))) * d.k;
  fx = (d.cforce * dx) / dist * pen;
  fy = (d.cforce * dy) / dist * pen;
#else
  double dist, rdist;
  sqrt_rcp_f64(dx * dx + dy * dy, dist, rdist);
  const double x = quot(-(dist - dmin), d.k, d.inv_k);
  // (log1p(e) == e bit for bit once e < 2^-53, i.e. |x| > 36.8, true for ~99 % of lanes; a wavefront-uniform branch around
  // log1p was tried in round 3: the 60 unrolled pairs of a kernel, once cut into basic blocks, made the register
  // allocator go from 168 to 256 VGPRs with spills, so the unconditional form stays.)
  const double pen = ((x > 0.0 ? x : 0.0) + log1p_unit_f64(exp_neg_f64(fabs(x)))) * d.k;
  fx = quot(d.cforce * dx, dist, rdist) * pen;
  fy = quot(d.cforce * dy, dist, rdist) * pen;
#endif
}

// f32 (throughput mode, stated tolerance 1e-4 relative): hardware rsq / exp2 / log2 (each ~1 ulp)
// instead of the ~230-instruction correctly-rounded sequence; one reciprocal shared by both
// components.  log(1+e) for e < 2^-24 returns 0 instead of e: an absolute softplus error <= 6e-8,
// i.e. a force error <= cforce*k*6e-8 = 6e-9.  dist == 0 still yields NaN like the reference
// (0 * inf).
template <>
__device__ __forceinline__ void pair_force<float>(float dx, float dy, float dmin, const DevDesc<float> &d,
                                                  float &fx, float &fy) {
  const float d2 = dx * dx + dy * dy;
  const float inv = __builtin_amdgcn_rsqf(d2);
  const float dist = d2 * inv;
  const float x = (dmin - dist) * d.inv_k;
  const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896341f);
  const float sp = fmaxf(x, 0.0f) + __builtin_amdgcn_logf(1.0f + e) * 0.693147180559945309f;
  const float s = d.cforce * inv * (sp * d.k);
  fx = dx * s;
  fy = dy * s;
}

template <typename T> __device__ __forceinline__ T t_div(T a, T b) { return a / b; }
// f32: a * rcp(b) (v_rcp_f32, ~1 ulp) instead of the IEEE divide sequence
template <> __device__ __forceinline__ float t_div<float>(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// (dx, dy) / np.linalg.norm((dx, dy)) (core.py:185-189).  f64: root and quotients from one v_rsq_f64 (sqrt_rcp_f64 / quot:
// the same doubles as the IEEE sqrt and divides, ~20 instructions instead of ~40; a zero vector gives NaN both ways).
template <typename T>
__device__ __forceinline__ void unit_vec(T dx, T dy, T &ux, T &uy) {
  const T n = t_norm2<T>(dx, dy);
  ux = t_div<T>(dx, n);
  uy = t_div<T>(dy, n);
}
template <>
__device__ __forceinline__ void unit_vec<double>(double dx, double dy, double &ux, double &uy) {
  double n, rn;
  sqrt_rcp_f64(fma(dy, dy, dx * dx), n, rn);
  ux = quot(dx, n, rn);
  uy = quot(dy, n, rn);
}

// F / mass (core.py:216).  f64: every entity of every scenario file has mass 1.0 (core.py:47,51-53), and the mass is a
// launch constant: a scalar branch skips the 11-instruction IEEE divide then (F / 1.0 == F).
template <typename T> __device__ __forceinline__ T over_mass(T F, T mass) { return t_div<T>(F, mass); }
template <> __device__ __forceinline__ double over_mass<double>(double F, double mass) {
  if (mass == 1.0) return F;
  return F / mass;
}

// The max_speed clamp of integrate_state (core.py:217-221): speed = sqrt(vx^2 + vy^2); if speed > max_speed:
// v = v / speed * max_speed (divide first, then multiply).  Branch-free (selects) in f32.
template <typename T>
__device__ __forceinline__ void speed_clamp(T &vx, T &vy, T max_speed) {
  const T sp = t_sqrt<T>(vx * vx + vy * vy);
  const T cx = t_div<T>(vx, sp) * max_speed, cy = t_div<T>(vy, sp) * max_speed;
  const bool clamp = sp > max_speed;
  vx = clamp ? cx : vx;
  vy = clamp ? cy : vy;
}
// f64: the square root and the two divides (~40 instructions) only run for lanes that can be over the limit.  sqrt is
// monotonic, so v2 <= max^2 (1 - 2^-50) implies sqrt(v2) <= max: those lanes (nearly all: a wavefront whose 64 envs all
// stay below skips the block) decide on the squares; the others evaluate the reference's expression, with the
// correctly rounded root and quotients of sqrt_rcp_f64 / quot.  NaN speeds are left alone both ways.
template <>
__device__ __forceinline__ void speed_clamp<double>(double &vx, double &vy, double max_speed) {
  const double v2 = vx * vx + vy * vy;
  if (v2 > max_speed * max_speed * (1.0 - 0x1p-50)) {
    double sp, rsp;
    sqrt_rcp_f64(v2, sp, rsp);
    const double cx = quot(vx, sp, rsp) * max_speed, cy = quot(vy, sp, rsp) * max_speed;
    const bool clamp = sp > max_speed;
    vx = clamp ? cx : vx;
    vy = clamp ? cy : vy;
  }
}

// integrate_state (core.py:214-221) for one movable entity; returns new velocity.
template <typename T>
__device__ __forceinline__ void integrate_vel(T &vx, T &vy, T Fx, T Fy, T mass, T damp1, T dt,
                                              T max_speed) {
  vx = vx * damp1;
  vy = vy * damp1;
  vx = vx + over_mass<T>(Fx, mass) * dt;
  vy = vy + over_mass<T>(Fy, mass) * dt;
  if (max_speed >= T(0)) speed_clamp<T>(vx, vy, max_speed);
}

// Same arithmetic, branch-free (selects), for the straight-line register kernel.  HAS_MAX is the
// compile-time form of `entity.max_speed is not None`.
template <typename T, bool HAS_MAX>
__device__ __forceinline__ void integrate_vel_sel(T &vx, T &vy, T Fx, T Fy, T mass, T damp1, T dt,
                                                  T max_speed) {
  vx = vx * damp1;
  vy = vy * damp1;
  vx = vx + over_mass<T>(Fx, mass) * dt;
  vy = vy + over_mass<T>(Fy, mass) * dt;
  if (HAS_MAX) speed_clamp<T>(vx, vy, max_speed);
}

// apply_leader_force (core.py:153-194) for agent a of the env in column `col`: every leader whose
// force_id equals a's forced_id, in agent order.  A super-leader's command is the raw c[0:2] and is
// only observed (core.py:162-163); a leader's command is argmax(c) (first maximum; dim_c == 3 maps
// to [0, 5, 6], core.py:169-171): 1-4 axis pushes, 5 / 6 the unit vector towards / away from the
// leader, added to the physical force and to leader_force (mag = 1, core.py:154,190-192).
// pos_of(entity, component) returns the pre-step position.
template <typename T, typename PosFn>
__device__ __forceinline__ void leader_force(const KParams<T> &p, int a, int64_t col, PosFn pos_of, T &Fx, T &Fy,
                                             T &lfx, T &lfy) {
  const DevDesc<T> &d = p.d;
  lfx = T(0);
  lfy = T(0);
  const int want = desc_i8(d.forced_id, a);
  if (want < 0) return;
  for (int l = 0; l < d.A; ++l) {
    if (desc_i8(d.force_id, l) != want) continue;
    const bool mute = bit(d.silent, l);  // action.c stays zeros(dim_c) (core.py:32-36)
    const T *c = p.act + (int64_t)d.cact_row0[l] * p.ld + col;
    if (bit(d.advsuper, l)) {
      lfx = mute ? T(0) : ld_stream(c);
      lfy = mute ? T(0) : ld_stream(c + p.ld);
      continue;
    }
    if (mute) continue;  // argmax(zeros) = 0: no push
    int cmd = 0;
    T best = ld_stream(c);
    for (int k = 1; k < d.dim_c; ++k) {
      const T v = ld_stream(c + (int64_t)k * p.ld);
      if (v > best || (v != v && best == best)) {  // np.argmax: first maximum, a NaN wins
        best = v;
        cmd = k;
      }
    }
    if (d.dim_c == 3) cmd = cmd == 0 ? 0 : cmd + 4;
    T fx = T(0), fy = T(0);
    if (cmd == 1) fx = -T(1);
    else if (cmd == 2) fx = T(1);
    else if (cmd == 3) fy = -T(1);
    else if (cmd == 4) fy = T(1);
    else if (cmd == 5 || cmd == 6) {
      T dx = pos_of(l, 0) - pos_of(a, 0), dy = pos_of(l, 1) - pos_of(a, 1);
      if (cmd == 6) { dx = pos_of(a, 0) - pos_of(l, 0); dy = pos_of(a, 1) - pos_of(l, 1); }
      unit_vec<T>(dx, dy, fx, fy);
    }
    Fx = Fx + fx;
    Fy = Fy + fy;
    lfx = lfx + fx;
    lfy = lfy + fy;
  }
}

// Philox4x32-10 (Salmon et al. 2011); restated in oracle/mpe_oracle.py:philox4x32_10.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 53-bit uniform in [0,1) from two words; p = -r + 2r*u (np.random.uniform's low + (high-low)*u).
__device__ __forceinline__ double uniform_pm(uint32_t hi, uint32_t lo, double r) {
  const double u = ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
  return -r + (2.0 * r) * u;
}

// ---- in-kernel standard normals for u_noise / c_noise (core.py:147-148,229: `np.random.randn(...) * noise`) -------------
// The reference draws from numpy's global stream, which no device code can follow; with MPE_FLAG_DEVICE_NOISE and no
// noise tensor from the caller the step kernels draw their own: Philox4x32-10 keyed like mpe_reset / the actor (counter =
// (global env id lo, hi, step, stream word), key = seed), so a draw depends on WHICH env, step, agent and component it
// is for and on nothing else -- not on the batch size, the sharding over GPUs or the kernel form -- then Box-Muller:
//     u1 = (k53 + 1) 2^-53 in (0, 1], u2 = k53' 2^-53 in [0, 1)   (k53 from two words each, like uniform_pm)
//     z0 = sqrt(-2 ln u1) cos(2 pi u2),  z1 = sqrt(-2 ln u1) sin(2 pi u2)
// Stream words: 0x40000000 | agent -> (z0, z1) = the agent's (x, y) u_noise normals; 0x60000000 | agent << 8 | j -> the
// c_noise normals of comm components 2j, 2j + 1.
// f64: ln / sin / cos are written out in IEEE +, -, *, /, sqrt only, in a fixed order with contraction off, so that
// oracle/mpe_oracle.py:device_normals restates them BIT FOR BIT in numpy (a libm call would not be the same function
// on both sides).  ln u1 = e ln2 + 2 atanh((m - 1) / (m + 1)) with u1 = 2^e m, m in [sqrt(1/2), sqrt 2) (odd series to
// s^21, |s| <= 0.172: truncation 1e-18); sin / cos of g pi/4, g in [0, 1], by Taylor polynomials (to a^17 / a^18:
// truncation 1e-19) after an exact octant reduction of 8 u2.  Accuracy ~2e-16: far more than a noise term needs.
// f32: the same Philox words, hardware v_log_f32 / v_sin_f32 / v_cos_f32 (the latter take revolutions: no 2 pi), u1 kept
// to 24 significant bits whatever its magnitude; within ~1e-6 of the f64 normal.
constexpr uint32_t kNoiseWordU = 0x40000000u, kNoiseWordC = 0x60000000u;

__device__ __forceinline__ void normal_pair(const uint32_t w[4], double &z0, double &z1) {
#pragma clang fp contract(off)
  const double u1 = ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6) + 1.0) * 0x1p-53;
  const double u2 = ((double)(w[2] >> 5) * 67108864.0 + (double)(w[3] >> 6)) * 0x1p-53;
  int e;
  double m = frexp(u1, &e);  // exact: m in [0.5, 1)
  const bool low = m < 0.70710678118654757;
  m = low ? m * 2.0 : m;
  e = low ? e - 1 : e;
  const double s = (m - 1.0) / (m + 1.0), z = s * s;
  double P = 1.0 / 21.0;
  P = P * z + 1.0 / 19.0;
  P = P * z + 1.0 / 17.0;
  P = P * z + 1.0 / 15.0;
  P = P * z + 1.0 / 13.0;
  P = P * z + 1.0 / 11.0;
  P = P * z + 1.0 / 9.0;
  P = P * z + 1.0 / 7.0;
  P = P * z + 1.0 / 5.0;
  P = P * z + 1.0 / 3.0;
  P = P * z + 1.0;
  const double lnu = (double)e * 0.69314718055994529 + 2.0 * (s * P);
  const double r = sqrt(-2.0 * lnu);
  const double t = u2 * 8.0;
  const int k = (int)t;  // octant 0..7
  const double f = t - (double)k;
  const double g = (k & 1) ? 1.0 - f : f;
  const double a = g * 0.78539816339744828, a2 = a * a;
  double S = -1.0 / 355687428096000.0;  // -1/17!
  S = S * a2 + 1.0 / 1307674368000.0;
  S = S * a2 - 1.0 / 6227020800.0;
  S = S * a2 + 1.0 / 39916800.0;
  S = S * a2 - 1.0 / 362880.0;
  S = S * a2 + 1.0 / 5040.0;
  S = S * a2 - 1.0 / 120.0;
  S = S * a2 + 1.0 / 6.0;   // sign folded below: S = 1/6 - a2/120 + ...
  S = a - a * (a2 * S);     // sin a = a - a^3 (1/6 - a^2/120 + ...)
  double Cc = -1.0 / 6402373705728000.0;  // -1/18!
  Cc = Cc * a2 + 1.0 / 20922789888000.0;
  Cc = Cc * a2 - 1.0 / 87178291200.0;
  Cc = Cc * a2 + 1.0 / 479001600.0;
  Cc = Cc * a2 - 1.0 / 3628800.0;
  Cc = Cc * a2 + 1.0 / 40320.0;
  Cc = Cc * a2 - 1.0 / 720.0;
  Cc = Cc * a2 + 1.0 / 24.0;
  Cc = Cc * a2 - 0.5;
  Cc = Cc * a2 + 1.0;       // cos a
  const bool swap = ((k + 1) >> 1) & 1;           // octants 1, 2, 5, 6: (cos, sin) = (+-sin a', +-cos a')
  const double c0 = swap ? S : Cc, s0 = swap ? Cc : S;
  z0 = r * ((((k + 2) >> 2) & 1) ? -c0 : c0);     // cos < 0 in octants 2..5
  z1 = r * (k >= 4 ? -s0 : s0);                   // sin < 0 in octants 4..7
}

__device__ __forceinline__ void normal_pair(const uint32_t w[4], float &z0, float &z1) {
  const float u1 = ((float)(w[0] >> 5) * 67108864.0f + (float)(w[1] >> 6) + 1.0f) * 0x1p-53f;
  const float u2 = (float)w[2] * 0x1p-32f;
  const float r = __builtin_amdgcn_sqrtf(-1.38629436111989062f * __builtin_amdgcn_logf(u1));  // -2 ln2 log2(u1)
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}

// Step counter of this launch (wave-uniform): the baked word plus the device word a graph replay moves.
template <typename T>
__device__ __forceinline__ uint32_t noise_step(const KParams<T> &p) {
  return p.nstep + (p.nstep_dev != nullptr ? *p.nstep_dev : 0u);
}
template <typename T>
__device__ __forceinline__ void device_normals(const KParams<T> &p, int64_t env, uint32_t step, uint32_t word, T &z0,
                                               T &z1) {
  const uint64_t gid = p.env_id0 + (uint64_t)env;
  uint32_t w[4];
  philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), step, word, p.nk0, p.nk1, w);
  normal_pair(w, z0, z1);
}
template <typename T>
__device__ __forceinline__ bool device_noise(const KParams<T> &p) { return (p.flags & MPE_FLAG_DEVICE_NOISE) != 0; }

// agent.state.c of a speaking agent as this step leaves it (core.py:226-230): the comm entries of its action,
// plus c_noise * standard normal when the agent has a c_noise (the caller's tensor, else with MPE_FLAG_DEVICE_NOISE the
// in-kernel draw: counter-based, so every reader of (agent, component, env) computes the same value); when only
// observing, the comm buffer.
template <typename T, bool DO_STEP>
__device__ __forceinline__ T state_c(const KParams<T> &p, int a, int k, int64_t env) {
  if (!DO_STEP) return p.comm[(int64_t)(a * p.d.dim_c + k) * p.ld + env];
  T v = ld_stream(p.act + (int64_t)(p.d.cact_row0[a] + k) * p.ld + env);
  if (p.d.c_noise[a] != T(0)) {
    if (p.cnoise != nullptr) {
      v = v + ld_stream(p.cnoise + (int64_t)(a * p.d.dim_c + k) * p.ld + env) * p.d.c_noise[a];
    } else if (device_noise(p)) {
      T z0, z1;
      device_normals<T>(p, env, noise_step(p), kNoiseWordC | ((uint32_t)a << 8) | (uint32_t)(k >> 1), z0, z1);
      v = v + ((k & 1) ? z1 : z0) * p.d.c_noise[a];
    }
  }
  return v;
}

// Which block of envs this workgroup owns.  Workgroups are dispatched round-robin over the 8 XCDs (each with
// its own L2); with MPE_XCD_SWIZZLE every XCD takes one contiguous eighth of the batch instead, so that the
// workgroups sharing an L2 write neighbouring segments of the same SoA rows.
__device__ __forceinline__ int64_t env_block() {
#ifdef MPE_XCD_SWIZZLE
  const uint32_t nb = gridDim.x, b = blockIdx.x, x = b & 7u, i = b >> 3, q = nb >> 3, r = nb & 7u;
  return (int64_t)(x * q + (x < r ? x : r) + i);
#else
  return (int64_t)blockIdx.x;
#endif
}

// np.sum(np.square(p - q)) for 2-vectors.
template <typename T>
__device__ __forceinline__ T sqdist(T px, T py, T qx, T qy) {
  const T dx = px - qx, dy = py - qy;
  return dx * dx + dy * dy;
}

__device__ __forceinline__ bool kind_has_goal(int kind) {
  return kind == MPE_SIMPLE_ADVERSARY || kind == MPE_SIMPLE_PUSH;
}
// the two speaking stock scenarios (per-agent goal_b + observed comm)
__device__ __forceinline__ bool kind_is_comm(int kind) {
  return kind == MPE_SPEAKER_LISTENER || kind == MPE_SIMPLE_REFERENCE;
}

// is_collision (simple_tag.py:69-73 / simple_spread.py:65-69): sqrt(d2) < dmin, strict, for d2 = dx^2 + dy^2.
template <typename T>
__device__ __forceinline__ bool touching_d2(T d2, T dmin) {
  return t_sqrt<T>(d2) < dmin;
}
// f64: the same boolean without the ~25-instruction correctly rounded square root.  sqrt is monotonic, so away from
// the threshold the squares decide: d2 < dmin^2 (1 - 2^-50) implies sqrt(d2) < dmin (1 - 2^-51.2), which rounds to a
// double below dmin; d2 > dmin^2 (1 + 2^-50) likewise rounds to one >= dmin.  Inside that band (a few ulps; about one
// contact test in 10^15) the reference's expression itself is evaluated.  NaN -> false both ways.
template <>
__device__ __forceinline__ bool touching_d2<double>(double d2, double dmin) {
  const double m2 = dmin * dmin;
  const double lo = m2 * (1.0 - 0x1p-50), hi = m2 * (1.0 + 0x1p-50);
  bool r = d2 < lo;
  if (d2 >= lo && d2 <= hi) {  // (written out: the compiler speculates a bare sqrt() call, 17 instructions, out of the branch)
    double root, rinv;
    sqrt_rcp_f64(d2, root, rinv);
    r = root < dmin;
  }
  return r;
}
template <typename T>
__device__ __forceinline__ bool touching(T dx, T dy, T dmin) {
  return touching_d2<T>(dx * dx + dy * dy, dmin);
}

// bound() of simple_tag.agent_reward (simple_tag.py:103-108), x = |pos|.
template <typename T>
__device__ __forceinline__ T tag_bound(T x) {
  if (x < T(0.9)) return T(0);
  if (x < T(1.0)) return (x - T(0.9)) * T(10);
  const T e = t_exp<T>(T(2) * x - T(2));
  return T(10) < e ? T(10) : e;  // Python min(e, 10): a NaN e stays NaN
}

template <typename T>
__device__ __forceinline__ T tag_bound_sel(T x) {
  // f64: exp is ~45 instructions and only agents outside the arena (|pos| >= 1, or NaN) need it: skipped by a
  // wavefront-level branch when no lane does; f32: one v_exp_f32, evaluated unconditionally
  T e = T(0);
  if (sizeof(T) == 4 || !(x < T(1.0))) e = t_exp<T>(T(2) * x - T(2));
  const T hi = T(10) < e ? T(10) : e;
  const T mid = (x - T(0.9)) * T(10);
  return x < T(0.9) ? T(0) : (x < T(1.0) ? mid : hi);
}

// The NT nearest visible targets of one observer as a list sorted by (d2, index), kept while the targets stream by in
// index order (L1F2_3Targets_100observe.py:242-245: a stable sort of the visible targets by squared distance).  Slots
// start at (+inf, -1); an invisible target enters as +inf and stops in front of nothing.  One candidate bubbles
// through the list: the strict comparison leaves an equal, earlier-index entry in front, like the stable sort, and
// from the slot it takes onwards the list shifts by one (the entry pushed out is an earlier index than an equal
// successor: comparing it again would let the successor pass it -- found by the f32 kernel-form fuzz on an exact tie of
// two targets behind a nearer third).  The distances move with v_min / v_max, only the indices need selects (no
// branches: the select form of the same insertion compiled to three exec-mask regions per target).
template <typename T> __device__ __forceinline__ T t_min(T a, T b);
template <> __device__ __forceinline__ float t_min<float>(float a, float b) { return __builtin_fminf(a, b); }
template <> __device__ __forceinline__ double t_min<double>(double a, double b) { return __builtin_fmin(a, b); }
template <typename T> __device__ __forceinline__ T t_max(T a, T b);
template <> __device__ __forceinline__ float t_max<float>(float a, float b) { return __builtin_fmaxf(a, b); }
template <> __device__ __forceinline__ double t_max<double>(double a, double b) { return __builtin_fmax(a, b); }
template <typename T, int NT>
__device__ __forceinline__ void nearest_init(T (&bd)[NT], int (&bl)[NT]) {
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    bd[k] = T(INFINITY);
    bl[k] = -1;  // empty slot
  }
}
template <typename T, int NT>
__device__ __forceinline__ void nearest_insert(T (&bd)[NT], int (&bl)[NT], T d2, bool seen, int l) {
  T x = seen ? d2 : T(INFINITY);  // (a NaN d2 is not seen: NaN < r2 is false)
  int xi = l;
  bool placed = false;  // once the candidate is in, x is the entry it pushed out: that one goes in front of an equal successor
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const bool front = placed || x < bd[k];
    placed = front;
    const T lo = t_min<T>(bd[k], x), hi = t_max<T>(bd[k], x);
    const int li = front ? xi : bl[k], hi_i = front ? bl[k] : xi;
    bd[k] = lo;
    bl[k] = li;
    x = hi;
    xi = hi_i;
  }
}

// Sum over the `width` consecutive lanes (width = 32 or 64) that contain this lane.
__device__ __forceinline__ double group_sum(double v, int width) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
    if (off < width) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
  return v;
}

}  // namespace mpe
