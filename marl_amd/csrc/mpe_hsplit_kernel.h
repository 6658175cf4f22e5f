// One-AGENT-per-wavefront form of the command-hierarchy step (mpe_hier_kernel.h holds the one-env-per-lane form and the
// description of the worlds).  A workgroup is 64 * EW envs x A wavefronts: wavefront (agent `me`, env group `half`), lane
// <-> env.
//
// Why: a batch of a few thousand envs is a fraction of a wavefront per SIMD, and the one-env-per-lane kernels of these
// worlds are 2 000 - 12 000 instructions long with 150 - 200 VGPRs in f64: such a launch runs at the latency of its own
// instruction stream.  Here every wavefront runs ONE agent: ~1/A of the instructions with the agent index a scalar, loops
// over the other entities as real loops over LDS columns, < 90 VGPRs, A times as many wavefronts on the chip.
// S1L2F2_10Targets f64: 4 096 envs 39.6 -> 14.8 us per step, 65 536 envs 45.1 -> 37.4; from 262 144 envs on the
// one-env-per-lane form wins (mpe_launch.h:hier_id_splits has the table and the reason).
//
// Exchange through LDS, column = env, so nobody but the lanes of one env column ever meet and the barriers are the only
// synchronisation:
//   sP  pre-step positions of every entity (each row loaded from HBM exactly once per workgroup)
//   sK / sC  the leaders' decoded command / raw c[0:2] (core.py:153-194), written by the leader's own wavefront
//   sN  post-step (pos.x, pos.y, vel.x, vel.y) of every agent, written by its own wavefront
//   sH  targets every agent touches (the leaders' rewards add up their followers')
// The arithmetic per agent is the one-env-per-lane kernel's, operation for operation (same helpers, same order), so the
// two forms agree bit for bit on state, observations, rewards and info (tests/test_gpu_hier.py); the per-workgroup
// statistics rows group 64 * EW envs instead of 256, so their f64 sums can differ in the last bits.
#pragma once
#include "mpe_device.h"
#include "mpe_launch.h"

namespace mpe {

#ifndef MPE_HSPLIT_ATTR
#define MPE_HSPLIT_ATTR
#endif

template <typename T, int A, int L, int NF, int NG, int C, bool MOVING, int EW, bool DO_STEP, bool DO_OBS>
__global__ __launch_bounds__(64 * EW * A) MPE_HSPLIT_ATTR void hier_split_kernel(const KParams<T> p) {
  constexpr int E = A + L, LE = A - NG, NL = LE - NF, ENVS = 64 * EW, W = 2 * A + 2;
  constexpr int NT = L < kHierTargets ? L : kHierTargets;
  constexpr int G = A - 1;  // MOVING: the prey
  static_assert(NF >= 0 && NL >= 1 && NG >= 0, "followers, at least one leader, good agents");
  static_assert(!MOVING || (L == 0 && NG == 1), "the *_MovingTarget files: no landmarks, one prey");
  static_assert(64 * EW * A <= 1024, "workgroup size");
  const DevDesc<T> &d = p.d;
  const int64_t ld = p.ld;
  const uint32_t lane = threadIdx.x & 63u;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int me = wave / EW, half = wave % EW;  // scalars
  const uint32_t col = (uint32_t)half * 64u + lane;
  const int64_t blk0 = env_block() * ENVS;  // wave-uniform
  const bool active = blk0 + col < p.B;
  const uint32_t lo = active ? col : kNoLane;  // tail lanes: see Win

  __shared__ T sP[2 * E][ENVS];
  __shared__ T sN[4 * A][ENVS];
  __shared__ T sC[2 * NL][ENVS];
  __shared__ int sK[NL][ENVS];
  __shared__ int sH[MOVING ? 1 : A][MOVING ? 1 : ENVS];
  __shared__ double s_stat[EW][W];

  // first action row of agent a: followers 5 rows, leaders 5 + C, good agents 5 (mpe_hier_kernel.h:hier_step)
  auto arow = [](int a) {
    return a < NF ? 5 * a : (a < LE ? 5 * NF + (5 + C) * (a - NF) : 5 * NF + (5 + C) * NL + 5 * (a - LE));
  };
  const bool speaks = me >= NF && me < LE;

  const Win<T> wpos(p.pos + blk0, ld), wvel(p.vel + blk0, ld);
  T mpx = wpos.ld_state(2 * me, lo), mpy = wpos.ld_state(2 * me + 1, lo);
  T mvx = wvel.ld_state(2 * me, lo), mvy = wvel.ld_state(2 * me + 1, lo);
  for (int r = 2 * A + me; r < 2 * E; r += A) sP[r][col] = wpos.ld_state(r, lo);  // the targets, shared out
  // the statistics row is fetched up front: the read-modify-write at the end must not add a memory round trip
  double stats_prev = 0.0;
  if (DO_OBS && p.stats != nullptr && threadIdx.x < W) stats_prev = p.stats[(int64_t)blockIdx.x * W + threadIdx.x];
  T a1 = T(0), a2 = T(0), a3 = T(0), a4 = T(0);
  if (DO_STEP) {
    const Win<T> wact(p.act + blk0, ld);
    const int ar = arow(me);
    a1 = wact.ld_stream(ar + 1, lo);  // a[0] (the no-op slot) is never read (environment.py:189-190)
    a2 = wact.ld_stream(ar + 2, lo);
    a3 = wact.ld_stream(ar + 3, lo);
    a4 = wact.ld_stream(ar + 4, lo);
    sP[2 * me][col] = mpx;
    sP[2 * me + 1][col] = mpy;
    const Win<T> wcomm(p.comm + blk0, ld);
    if (speaks) {
      T c[C];
#pragma unroll
      for (int k = 0; k < C; ++k) c[k] = wact.ld_stream(ar + 5 + k, lo);
      int k_best = 0;  // np.argmax: first maximum, a NaN wins
      T best = c[0];
#pragma unroll
      for (int k = 1; k < C; ++k) {
        const bool take = c[k] > best || (c[k] != c[k] && best == best);
        best = take ? c[k] : best;
        k_best = take ? k : k_best;
      }
      // three commands only: nothing / towards / away (core.py:169-171)
      sK[me - NF][col] = C == 3 ? (k_best == 0 ? 0 : (k_best == 1 ? 5 : 6)) : k_best;
      sC[2 * (me - NF)][col] = c[0];
      sC[2 * (me - NF) + 1][col] = c[1];
      if (p.comm != nullptr) {  // core.py:226-230 (agent.state.c = action.c); small batches: the rows stay in the cache
#pragma unroll
        for (int k = 0; k < C; ++k) wcomm.st_state(me * C + k, lo, c[k]);
      }
    } else if (p.comm != nullptr && rewrite_silent_comm(p)) {
      for (int k = 0; k < C; ++k) wcomm.st_state(me * C + k, lo, T(0));
    }
  } else if (DO_OBS) {
    sN[4 * me + 0][col] = mpx;
    sN[4 * me + 1][col] = mpy;
    sN[4 * me + 2][col] = mvx;
    sN[4 * me + 3][col] = mvy;
  }
  __syncthreads();

  T lfx = T(0), lfy = T(0);
  if (DO_STEP) {
    T Fx = (a1 - a2) * d.accel[me], Fy = (a3 - a4) * d.accel[me];  // environment.py:188-196
    // core.py:153-194 (same order as mpe_device.h:leader_force): leaders in agent order
    const int want = desc_i8(d.forced_id, me);
    if (want >= 0) {
      for (int l = NF; l < LE; ++l) {
        if (desc_i8(d.force_id, l) != want) continue;
        if (bit(d.advsuper, l)) {  // observed only (core.py:162-163)
          lfx = sC[2 * (l - NF)][col];
          lfy = sC[2 * (l - NF) + 1][col];
          continue;
        }
        const int k = sK[l - NF][col];
        const T dx = sP[2 * l][col] - mpx, dy = sP[2 * l + 1][col] - mpy;
        T ux, uy;  // cmd 6 is exactly the negation
        unit_vec<T>(dx, dy, ux, uy);
        T fx = k == 1 ? -T(1) : (k == 2 ? T(1) : T(0));
        T fy = k == 3 ? -T(1) : (k == 4 ? T(1) : T(0));
        fx = k == 5 ? ux : (k == 6 ? -ux : fx);
        fy = k == 5 ? uy : (k == 6 ? -uy : fy);
        Fx = Fx + fx;
        Fy = Fy + fy;
        lfx = lfx + fx;
        lfy = lfy + fy;
      }
    }
#pragma nounroll
    for (int b = me + 1; b < E; ++b) {  // core.py:199-205, b ascending; only `me` is pushed
      T fx, fy;
      pair_force<T>(mpx - sP[2 * b][col], mpy - sP[2 * b + 1][col], d.size[me] + d.size[b], d, fx, fy);
      Fx = fx + Fx;
      Fy = fy + Fy;
    }
    integrate_vel_sel<T, true>(mvx, mvy, Fx, Fy, d.mass[me], d.damp1, d.dt, d.max_speed[me]);  // core.py:211-222
    mpx = mpx + mvx * d.dt;
    mpy = mpy + mvy * d.dt;
    wvel.st_state(2 * me, lo, mvx);
    wvel.st_state(2 * me + 1, lo, mvy);
    wpos.st_state(2 * me, lo, mpx);
    wpos.st_state(2 * me + 1, lo, mpy);
    if (p.lforce != nullptr) {
      const Win<T> wlf(p.lforce + blk0, ld);
      wlf.st_state(2 * me, lo, lfx);
      wlf.st_state(2 * me + 1, lo, lfy);
    }
    if (DO_OBS) {
      sN[4 * me + 0][col] = mpx;
      sN[4 * me + 1][col] = mpy;
      sN[4 * me + 2][col] = mvx;
      sN[4 * me + 3][col] = mvy;
      __syncthreads();
    }
  } else if (DO_OBS && p.lforce != nullptr && desc_i8(d.forced_id, me) >= 0) {  // only commanded agents observe it
    const Win<T> wlf0(p.lforce + blk0, ld);
    lfx = wlf0.ld_state(2 * me, lo);
    lfy = wlf0.ld_state(2 * me + 1, lo);
  }

  if constexpr (DO_OBS) {
    const bool want_obs = p.obs != nullptr, want_rew = p.rew != nullptr, want_info = p.info != nullptr,
               want_stats = p.stats != nullptr;
    const Win<T> wobs(p.obs + blk0 + (int64_t)d.obs_row0[me] * ld, ld);  // this agent's rows
    const int fi = desc_i8(d.force_id, me), fdi = desc_i8(d.forced_id, me);
    T rew = T(0);
    int coll = 0, extra = 0;  // this agent's info counter; the env-wide counter of statistics entry 2A
    if constexpr (!MOVING) {
      // ---- L1F2_3Targets_100observe.py:196-253; the targets `me` touches (is_collision, :150-154) ----
      const T r2 = d.obs_r2[me];
      // the NT nearest visible targets as a sorted (d2, index) list while the targets stream by, in index order with
      // strict comparisons: a tie leaves the earlier index in front, like the reference's stable sort (:242-245)
      T bd[NT];
      int bl[NT];
      nearest_init<T, NT>(bd, bl);
      int n = 0;
#pragma nounroll
      for (int l = 0; l < L; ++l) {
        const T dx = sP[2 * (A + l)][col] - mpx, dy = sP[2 * (A + l) + 1][col] - mpy;
        const T d2 = dx * dx + dy * dy;
        nearest_insert<T, NT>(bd, bl, d2, d2 < r2, l);
        n += touching_d2<T>(d2, d.size[A + l] + d.size[me]) ? 1 : 0;
      }
      sH[me][col] = n;
      if (want_obs) {
        int r = 0;
        wobs.st_stream(r++, lo, mvx);
        wobs.st_stream(r++, lo, mvy);
        wobs.st_stream(r++, lo, mpx);
        wobs.st_stream(r++, lo, mpy);
        uint32_t vis = 0;
        for (int j = 0; j < A; ++j) {
          if (j == me) continue;
          const T dx = sN[4 * j][col] - mpx, dy = sN[4 * j + 1][col] - mpy;
          const bool linked = (fi >= 0 && fi == desc_i8(d.forced_id, j)) || (fdi >= 0 && fdi == desc_i8(d.force_id, j));
          const bool v = linked || (dx * dx + dy * dy < r2);
          vis |= (uint32_t)v << j;
          wobs.st_stream(r++, lo, v ? dx : T(0));
          wobs.st_stream(r++, lo, v ? dy : T(0));
        }
        for (int j = 0; j < A; ++j) {
          if (j == me) continue;
          const bool v = (vis >> j) & 1u;
          wobs.st_stream(r++, lo, v ? sN[4 * j + 2][col] : T(0));
          wobs.st_stream(r++, lo, v ? sN[4 * j + 3][col] : T(0));
        }
        if (fdi >= 0) {
          wobs.st_stream(r++, lo, lfx);
          wobs.st_stream(r++, lo, lfy);
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) {  // the same subtraction as above, fetched by index; then zeros
          const int l = bl[k] < 0 ? 0 : bl[k];
          const T tx = sP[2 * (A + l)][col] - mpx, ty = sP[2 * (A + l) + 1][col] - mpy;
          wobs.st_stream(r++, lo, bl[k] < 0 ? T(0) : tx);
          wobs.st_stream(r++, lo, bl[k] < 0 ? T(0) : ty);
        }
      }
      // ---- rewards (:164-194 / S1L2F2_10Targets_wide_rewardFixed.py:127-151), benchmark_data (:136-148) ----
      if (want_rew || want_info || want_stats) {
        __syncthreads();  // sH
        int follower_hits = 0;
        for (int o = 0; o < A; ++o) follower_hits += bit(d.advleader, o) ? 0 : sH[o][col];
        const bool lead = bit(d.advleader, me), super = bit(d.advsuper, me);
        const bool own = d.reward_variant == 0 ? !lead : (!super && !lead);
        int hits = 0;
        T unit = T(10);
        if (own) {
          hits = n;
        } else if (d.reward_variant == 0 && super) {
          hits = follower_hits;
        } else if (d.reward_variant == 1 && super) {
          unit = T(1);
          for (int o = 0; o < A; ++o) hits += (desc_i8(d.forced_id, o) == fi && T(0) < d.obs_r2[o]) ? L : 0;
        } else {
          for (int o = 0; o < A; ++o) hits += desc_i8(d.forced_id, o) == fi ? sH[o][col] : 0;
        }
        T r_i = unit * T(hits);
        r_i = r_i - tag_bound_sel<T>(t_abs<T>(mpx));
        r_i = r_i - tag_bound_sel<T>(t_abs<T>(mpy));
        rew = r_i;
        const bool i_adv = bit(d.adversary, me);
        for (int a = 0; a < A; ++a)
          if (bit(d.adversary, a) != i_adv)
            coll += touching<T>(sN[4 * a][col] - mpx, sN[4 * a + 1][col] - mpy, d.size[a] + d.size[me]) ? 1 : 0;
        extra = follower_hits;
      }
    } else {
      // ---- L1F4_MovingTarget.py:214-258: distance (not squared-distance) radius test; command links visible only in
      // the L1F4 text; the prey sees everybody in absolute coordinates ----
      if (want_obs) {
        int r = 0;
        wobs.st_stream(r++, lo, mvx);
        wobs.st_stream(r++, lo, mvy);
        wobs.st_stream(r++, lo, mpx);
        wobs.st_stream(r++, lo, mpy);
        uint32_t vis = 0;
        for (int j = 0; j < A; ++j) {
          if (j == me) continue;
          const T ox = sN[4 * j][col], oy = sN[4 * j + 1][col];
          const T dx = ox - mpx, dy = oy - mpy;
          const bool linked = d.reward_variant == 0 &&
                              ((fi >= 0 && fi == desc_i8(d.forced_id, j)) || (fdi >= 0 && fdi == desc_i8(d.force_id, j)));
          bool v = me == G || linked;
          if (!v) v = touching_d2<T>(dx * dx + dy * dy, d.obs_r[me]);  // (:233-236)
          vis |= (uint32_t)v << j;
          wobs.st_stream(r++, lo, me == G ? ox : (v ? dx : T(0)));  // the prey: absolute (:249-251)
          wobs.st_stream(r++, lo, me == G ? oy : (v ? dy : T(0)));
        }
        for (int j = 0; j < A; ++j) {
          if (j == me) continue;
          const bool v = (vis >> j) & 1u;
          wobs.st_stream(r++, lo, v ? sN[4 * j + 2][col] : T(0));
          wobs.st_stream(r++, lo, v ? sN[4 * j + 3][col] : T(0));
        }
        if (bit(d.obs_lf, me)) {
          wobs.st_stream(r++, lo, lfx);
          wobs.st_stream(r++, lo, lfy);
        }
      }
      if (want_rew || want_info || want_stats) {
        // benchmark_data (:128-140): adversary <-> prey contacts; every wavefront counts them all
        const T gx = sN[4 * G][col], gy = sN[4 * G + 1][col];
        int count = 0, mine = 0;
        for (int o = 0; o < G; ++o) {
          const int hit = touching<T>(gx - sN[4 * o][col], gy - sN[4 * o + 1][col], d.size[G] + d.size[o]) ? 1 : 0;
          count += hit;
          mine = o == me ? hit : mine;
        }
        coll = me == G ? count : mine;
        const T adv_r = T(10 * count + (count >= 4 ? 50 : 0));  // :186-199 (integers: exact)
        T r_i = me == G ? T(0) - T(count) : adv_r;               // :162-171
        r_i = r_i - tag_bound_sel<T>(t_abs<T>(mpx));
        r_i = r_i - tag_bound_sel<T>(t_abs<T>(mpy));
        rew = r_i;
        extra = count;
      }
    }
    if (want_rew || want_info || want_stats) {
      const Win<T> wrew(p.rew + blk0, ld), winfo(p.info + blk0, ld);
      if (want_rew) wrew.st_stream(me, lo, rew);
      if (want_info) winfo.st_stream(me, lo, T(coll));
      // ---- per-workgroup statistics: every wavefront owns its agent's two entries, agent 0 the shared ones ----
      if (want_stats) {
        const double s_r = group_sum(active ? (double)rew : 0.0, 64);
        // coll <= A - 1 <= 7 (the prey's: <= 7), extra <= A * L <= 50 per lane: 64 lanes fit 16 bits each
        const uint32_t s_c = wave_sum_u32(active ? ((uint32_t)coll | (uint32_t)extra << 16) : 0u);
        const double n_active = (double)__builtin_popcountll(__ballot(active));
        if (lane == 0) {
          s_stat[half][me] = s_r;
          s_stat[half][A + me] = (double)(s_c & 0xFFFFu);
          if (me == 0) {
            s_stat[half][2 * A] = (double)(s_c >> 16);
            s_stat[half][2 * A + 1] = n_active;
          }
        }
        __syncthreads();
        if (threadIdx.x < W) {
          double s = 0.0;
#pragma unroll
          for (int h = 0; h < EW; ++h) s += s_stat[h][threadIdx.x];
          p.stats[(int64_t)blockIdx.x * W + threadIdx.x] = stats_prev + s;
        }
      }
    }
  }
}

}  // namespace mpe
