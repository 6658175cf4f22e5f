// Register-resident env-step kernel for the fork's follower / leader / super-leader pursuit worlds
// (multiagent/scenarios/L1F*.py, S1L*.py) in their stock shapes: one env per lane, the agent /
// target counts and the number of followers are template constants, so the state (E positions, A
// velocities, the leaders' command entries) lives in VGPRs and every loop is unrolled.
//
//   agents 0..NF-1          followers: silent, 5 action rows each (L1F2_3Targets_100observe.py:36-46)
//   agents NF..A-1          leaders / super-leader: 5 + 7 action rows each (:48-70, environment.py:61-69)
//   entities A..A+L-1       targets (landmarks)
//
// Stock flags checked on the host (mpe_hip.hip:small_variant): dim_c == 7, n_target_obs == 3, every
// agent movable with a max_speed, every entity collides, no u_noise.  Who commands whom (force_id /
// forced_id), the roles, radii and sizes stay run-time values: they are wave-uniform, so the tests on
// them are scalar branches.  Anything else runs the LDS-tile kernel (same arithmetic, same order:
// tests/test_gpu_hier.py checks the two bit for bit).
#pragma once
#include "mpe_device.h"
#include "mpe_launch.h"

namespace mpe {

#ifndef MPE_HIER_LDS_TARGETS
#define MPE_HIER_LDS_TARGETS 10
#endif
constexpr int kHierLdsTargets = MPE_HIER_LDS_TARGETS;  // worlds with at least this many targets keep them in LDS
// The per-target loops of the LDS form are REAL loops in f64 (the targets are addressed by index in LDS, their sizes by
// index in the kernel arguments): the 5-agent / 10-target kernel shrinks from 16 013 to 8 315 instructions (128 -> 66 KB of
// code streamed once per wavefront) and from 470 to 457 us.  f32 keeps them unrolled (rolled: 191 -> 202 us; its body is a
// quarter as long and the loop overhead shows).  -DMPE_HIER_UNROLL_TARGETS: unrolled everywhere.
#ifdef MPE_HIER_UNROLL_TARGETS
constexpr bool kHierRollF64 = false;
#else
constexpr bool kHierRollF64 = true;
#endif

// Occupancy cap, see mpe_small_kernel.h (L1F2_3Targets f32: 94.2 us uncapped at 5 waves/SIMD, 84.5 us at 3)
#ifndef MPE_HIER_ATTR
#define MPE_HIER_ATTR __attribute__((amdgpu_waves_per_eu(1, 3)))
#endif

// World.step for one env per lane (shared by hier_kernel and moving_kernel): action decode, leader force,
// pair forces, integration, comm / leader_force write-back; or, when only observing, the leader_force load.
// Agent order: NF silent followers, then the speaking leaders, then NG silent good agents (the prey of the
// *_MovingTarget files; 0 in the Targets files).
//
// EREG: how many entity positions sit in the px / py registers: all E, or only the A agents when the targets are kept in
// LDS (hier_kernel, L >= kHierLdsTargets); then tgt(l, component) reads target l of this lane's env.
template <typename T, int A, int E, int NF, int NG, int C, bool DO_STEP, bool DO_OBS, int EREG, int RES, typename TgtFn>
__device__ __forceinline__ void hier_step(const KParams<T> &p, const int64_t blk0, const uint32_t lo, T (&px)[EREG],
                                          T (&py)[EREG], T (&vx)[A], T (&vy)[A], T (&lfx)[A], T (&lfy)[A], TgtFn tgt) {
  static_assert(EREG == E || EREG == A, "all entities in registers, or the agents only");
  const DevDesc<T> &d = p.d;
  const int64_t ld = p.ld;
  const Win<T> wpos(p.pos + blk0, ld), wvel(p.vel + blk0, ld);
  constexpr int LE = A - NG;  // leaders are agents NF..LE-1
  static_assert(NF >= 0 && NF < LE && NG >= 0, "followers, at least one leader, good agents");
  // first action row of agent a (compile-time)
  auto arow = [](int a) {
    return a < NF ? 5 * a : (a < LE ? 5 * NF + (5 + C) * (a - NF) : 5 * NF + (5 + C) * (LE - NF) + 5 * (a - LE));
  };
  if (DO_STEP) {
    const Win<T> wact(p.act + blk0, ld);
    T a1[A], a2[A], a3[A], a4[A], c[LE - NF][C];
#pragma unroll
    for (int a = 0; a < A; ++a) {  // a[0] (the no-op slot) is never read (environment.py:189-190)
      a1[a] = wact.ld_stream(arow(a) + 1, lo);
      a2[a] = wact.ld_stream(arow(a) + 2, lo);
      a3[a] = wact.ld_stream(arow(a) + 3, lo);
      a4[a] = wact.ld_stream(arow(a) + 4, lo);
    }
#pragma unroll
    for (int l = NF; l < LE; ++l)
#pragma unroll
      for (int k = 0; k < C; ++k) c[l - NF][k] = wact.ld_stream(arow(l) + 5 + k, lo);

    T Fx[A], Fy[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {  // environment.py:188-196
      Fx[a] = (a1[a] - a2[a]) * d.accel[a];
      Fy[a] = (a3[a] - a4[a]) * d.accel[a];
    }
    // core.py:153-194 (same order as mpe_device.h:leader_force): leaders in agent order
    int cmd[LE - NF];
#pragma unroll
    for (int l = NF; l < LE; ++l) {  // np.argmax: first maximum, a NaN wins
      int k_best = 0;
      T best = c[l - NF][0];
#pragma unroll
      for (int k = 1; k < C; ++k) {
        const T v = c[l - NF][k];
        const bool take = v > best || (v != v && best == best);
        best = take ? v : best;
        k_best = take ? k : k_best;
      }
      // three commands only: nothing / towards / away (core.py:169-171)
      cmd[l - NF] = C == 3 ? (k_best == 0 ? 0 : (k_best == 1 ? 5 : 6)) : k_best;
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {
      const int want = desc_i8(d.forced_id, a);
      if (want < 0) continue;
#pragma unroll
      for (int l = NF; l < LE; ++l) {  // silent agents command nobody: their c is zero (argmax 0)
        if (desc_i8(d.force_id, l) != want) continue;
        if (bit(d.advsuper, l)) {  // observed only (core.py:162-163)
          lfx[a] = c[l - NF][0];
          lfy[a] = c[l - NF][1];
          continue;
        }
        const int k = cmd[l - NF];
        const T dx = px[l] - px[a], dy = py[l] - py[a];
        T ux, uy;  // cmd 6 is exactly the negation
        unit_vec<T>(dx, dy, ux, uy);
        T fx = k == 1 ? -T(1) : (k == 2 ? T(1) : T(0));
        T fy = k == 3 ? -T(1) : (k == 4 ? T(1) : T(0));
        fx = k == 5 ? ux : (k == 6 ? -ux : fx);
        fy = k == 5 ? uy : (k == 6 ? -uy : fy);
        Fx[a] = Fx[a] + fx;
        Fy[a] = Fy[a] + fy;
        lfx[a] = lfx[a] + fx;
        lfy[a] = lfy[a] + fy;
      }
    }
    // core.py:226-230 (agent.state.c = action.c), written here so that the command entries die early.  Write-only in
    // this launch: nontemporal in the larger worlds (S1L2F2_10Targets 1 Mi f32: 209 -> 193 us; the 3-agent worlds lose
    // 0.5-1.5 % with it and keep the default policy).  The zero rows of silent agents: rewrite_silent_comm().
    if (p.comm != nullptr) {
      const Win<T> wcomm(p.comm + blk0, ld);
      const bool zeros = rewrite_silent_comm(p);
      // the fused step follows the launch's Infinity Cache plan (RES >= 1: streamed); world_step alone keeps round 2's rule
      constexpr bool NT_COMM = (DO_STEP && DO_OBS) ? (RES >= 1) : (A >= 5);
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const bool silent = a < NF || a >= LE;
        if (silent && !zeros) continue;
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const T v = silent ? T(0) : c[silent ? 0 : a - NF][k];
          wcomm.template st_state_nt<NT_COMM>(a * C + k, lo, v);
        }
      }
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {  // core.py:199-205, b ascending; only a is pushed
#pragma unroll
      for (int b = a + 1; b < EREG; ++b) {
        T fx, fy;
        pair_force<T>(px[a] - px[b], py[a] - py[b], d.size[a] + d.size[b], d, fx, fy);
        Fx[a] = fx + Fx[a];
        Fy[a] = fy + Fy[a];
      }
    }
    if constexpr (EREG < E) {
      // the targets, one at a time out of LDS: every agent's sum still grows in ascending b (its agent pairs above, then
      // the targets in order), and the A pair forces of one target are independent instruction streams
      auto one_target = [&](int b) {  // the A pair forces of one target
        const T bx = tgt(b - A, 0), by = tgt(b - A, 1);
#pragma unroll
        for (int a = 0; a < A; ++a) {
          T fx, fy;
          pair_force<T>(px[a] - bx, py[a] - by, d.size[a] + d.size[b], d, fx, fy);
          Fx[a] = fx + Fx[a];
          Fy[a] = fy + Fy[a];
        }
      };
      if constexpr (sizeof(T) == 8 && kHierRollF64) {
#pragma nounroll
        for (int b = A; b < E; ++b) one_target(b);
      } else {
#pragma unroll
        for (int b = A; b < E; ++b) one_target(b);
      }
    }
    const Win<T> wlf(p.lforce + blk0, ld);
    T npx[A], npy[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {  // core.py:211-222
      integrate_vel_sel<T, true>(vx[a], vy[a], Fx[a], Fy[a], d.mass[a], d.damp1, d.dt, d.max_speed[a]);
      npx[a] = px[a] + vx[a] * d.dt;
      npy[a] = py[a] + vy[a] * d.dt;
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {
      px[a] = npx[a];
      py[a] = npy[a];
      wvel.template st_state_nt<(RES >= 4)>(2 * a, lo, vx[a]);
      wvel.template st_state_nt<(RES >= 4)>(2 * a + 1, lo, vy[a]);
      wpos.st_state(2 * a, lo, px[a]);
      wpos.st_state(2 * a + 1, lo, py[a]);
      if (p.lforce != nullptr) {  // (read again only by an observe-only call)
        wlf.template st_state_nt<(RES >= 2)>(2 * a, lo, lfx[a]);
        wlf.template st_state_nt<(RES >= 2)>(2 * a + 1, lo, lfy[a]);
      }
    }
  } else if (DO_OBS && p.lforce != nullptr) {
    const Win<T> wlf0(p.lforce + blk0, ld);
#pragma unroll
    for (int a = 0; a < A; ++a) {
      if (desc_i8(d.forced_id, a) < 0) continue;  // only commanded agents observe it
      lfx[a] = wlf0.ld_state(2 * a, lo);
      lfy[a] = wlf0.ld_state(2 * a + 1, lo);
    }
  }

}

// RES: the launch's Infinity Cache plan (mpe_device.h:residency_plan / resid_variant)
template <typename T, int A, int L, int NF, int THREADS, bool DO_STEP, bool DO_OBS, int RES = 0>
__global__ __launch_bounds__(THREADS) MPE_HIER_ATTR void hier_kernel(const KParams<T> p) {
  constexpr int E = A + L;
  constexpr int C = kHierDimC;
  constexpr int NT = L < kHierTargets ? L : kHierTargets;
  static_assert(A <= 6, "counter packing holds 6 agents");
  const DevDesc<T> &d = p.d;
  const int64_t ld = p.ld;
  const int64_t blk0 = env_block() * THREADS;  // wave-uniform
  const bool active = blk0 + threadIdx.x < p.B;
  const uint32_t lo = active ? threadIdx.x : kNoLane;  // tail lanes: see Win
  const Win<T> wpos(p.pos + blk0, ld), wvel(p.vel + blk0, ld);
  // Ten-target worlds keep the (immovable) targets in LDS, column = lane: nobody but the lane itself touches its column,
  // so there is no barrier.  20 fewer position registers (40 in f64), and the three nearest targets are fetched by
  // INDEX after the selection passes instead of being dragged through them as two more selects per candidate.
  constexpr bool TLDS = L >= kHierLdsTargets;
  constexpr int EREG = TLDS ? A : E;
  __shared__ T sT[TLDS ? 2 * L : 1][TLDS ? THREADS : 1];
  T px[EREG], py[EREG], vx[A], vy[A], lfx[A], lfy[A];
#pragma unroll
  for (int e = 0; e < A; ++e) {
    px[e] = wpos.ld_state(2 * e, lo);
    py[e] = wpos.ld_state(2 * e + 1, lo);
  }
#pragma unroll
  for (int e = A; e < EREG; ++e) {
    px[e] = wpos.template ld_state_nt<(RES >= 3)>(2 * e, lo);
    py[e] = wpos.template ld_state_nt<(RES >= 3)>(2 * e + 1, lo);
  }
  if constexpr (TLDS) {
#pragma unroll
    for (int r = 0; r < 2 * L; ++r) sT[r][threadIdx.x] = wpos.template ld_state_nt<(RES >= 3)>(2 * A + r, lo);
  }
  auto tgt = [&](int l, int c) -> T { return sT[TLDS ? 2 * l + c : 0][TLDS ? threadIdx.x : 0]; };
#pragma unroll
  for (int a = 0; a < A; ++a) {
    vx[a] = wvel.template ld_state_nt<(RES >= 4)>(2 * a, lo);
    vy[a] = wvel.template ld_state_nt<(RES >= 4)>(2 * a + 1, lo);
    lfx[a] = lfy[a] = T(0);
  }

  hier_step<T, A, E, NF, 0, C, DO_STEP, DO_OBS, EREG, RES>(p, blk0, lo, px, py, vx, vy, lfx, lfy, tgt);

  if (DO_OBS) {
    const bool want_obs = p.obs != nullptr, want_rew = p.rew != nullptr, want_info = p.info != nullptr,
               want_stats = p.stats != nullptr;
    int hits[A];  // targets agent o touches (is_collision, L1F2_3Targets_100observe.py:150-154)
    // ---- observations (:196-253) and the per-agent target contacts, one agent at a time -------------
#pragma unroll
    for (int i = 0; i < A; ++i) {
      T tdx[TLDS ? 1 : L], tdy[TLDS ? 1 : L], td2[TLDS ? 1 : L];
      int n = 0;
      // TLDS: the NT nearest visible targets are kept as a sorted (d2, index) list while the targets stream by, in index
      // order with strict comparisons: a tie leaves the earlier index in front, like the reference's stable sort.
      // 20 instructions per target instead of 36 for three selection passes, and no td2[] array.
      const T r2 = d.obs_r2[i];
      T bd[NT];
      int bl[NT];
      nearest_init<T, NT>(bd, bl);
      auto one_target = [&](int l) {
        T dx, dy, d2;
        if constexpr (TLDS) {
          dx = tgt(l, 0) - px[i];
          dy = tgt(l, 1) - py[i];
          d2 = dx * dx + dy * dy;
          nearest_insert<T, NT>(bd, bl, d2, d2 < r2, l);
        } else {
          dx = tdx[l] = px[A + l] - px[i];
          dy = tdy[l] = py[A + l] - py[i];
          d2 = td2[l] = dx * dx + dy * dy;
        }
#ifdef MPE_HIER_SQRT_CONTACT
        n += t_sqrt<T>(d2) < d.size[A + l] + d.size[i] ? 1 : 0;
#else
        n += touching_d2<T>(d2, d.size[A + l] + d.size[i]) ? 1 : 0;
#endif
      };
      if constexpr (TLDS && sizeof(T) == 8 && kHierRollF64) {
#pragma nounroll
        for (int l = 0; l < L; ++l) one_target(l);
      } else {
#pragma unroll
        for (int l = 0; l < L; ++l) one_target(l);
      }
      hits[i] = n;
      if (!want_obs) continue;
      const Win<T> wobs(p.obs + blk0 + (int64_t)d.obs_row0[i] * ld, ld);  // this agent's rows
      int r = 0;
      wobs.st_stream(r++, lo, vx[i]);
      wobs.st_stream(r++, lo, vy[i]);
      wobs.st_stream(r++, lo, px[i]);
      wobs.st_stream(r++, lo, py[i]);
      const int fi = desc_i8(d.force_id, i), fdi = desc_i8(d.forced_id, i);
      bool vis[A];
#pragma unroll
      for (int j = 0; j < A; ++j) {
        if (j == i) continue;
        const T dx = px[j] - px[i], dy = py[j] - py[i];
        const bool linked = (fi >= 0 && fi == desc_i8(d.forced_id, j)) || (fdi >= 0 && fdi == desc_i8(d.force_id, j));
        vis[j] = linked || (dx * dx + dy * dy < r2);
        wobs.st_stream(r++, lo, vis[j] ? dx : T(0));
        wobs.st_stream(r++, lo, vis[j] ? dy : T(0));
      }
#pragma unroll
      for (int j = 0; j < A; ++j) {
        if (j == i) continue;
        wobs.st_stream(r++, lo, vis[j] ? vx[j] : T(0));
        wobs.st_stream(r++, lo, vis[j] ? vy[j] : T(0));
      }
      if (fdi >= 0) {
        wobs.st_stream(r++, lo, lfx[i]);
        wobs.st_stream(r++, lo, lfy[i]);
      }
      // visible targets, nearest first, ties by index (stable sort, :242-245), then zeros.  Three-target worlds: three
      // selection passes over registers.  (Round 2 tried the single pass there, with the relative positions carried
      // along: nine extra live doubles, 225 -> 274 VGPRs in f64.  With the targets in LDS the list holds (d2, index)
      // only and the positions are fetched afterwards.)
      T prev_d = T(0);
      int prev_l = -1;
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        T bx = T(0), by = T(0);
        if constexpr (TLDS) {  // the k-th nearest's relative position: the same subtraction as above, fetched by index
          const int l = bl[k] < 0 ? 0 : bl[k];
          const T tx = sT[2 * l][threadIdx.x] - px[i], ty = sT[2 * l + 1][threadIdx.x] - py[i];
          bx = bl[k] < 0 ? T(0) : tx;
          by = bl[k] < 0 ? T(0) : ty;
        } else {
          int best_l = -1;
          T best_d = T(0);
#pragma unroll
          for (int l = 0; l < L; ++l) {
            // k == 0: nothing chosen yet.  Later slots: strictly after the previous pick in (d2, index)
            // order; once a slot finds nothing (best_l = -1 -> prev_l = L) the rest find nothing either.
            const bool after = k == 0 || td2[l] > prev_d || (td2[l] == prev_d && l > prev_l);
            const bool take = td2[l] < r2 && after && (best_l < 0 || td2[l] < best_d);
            best_l = take ? l : best_l;
            best_d = take ? td2[l] : best_d;
            bx = take ? tdx[l] : bx;
            by = take ? tdy[l] : by;
          }
          prev_d = best_l < 0 ? r2 : best_d;  // nothing found: no later d2 < r2 can follow
          prev_l = best_l < 0 ? L : best_l;
        }
        wobs.st_stream(r++, lo, bx);
        wobs.st_stream(r++, lo, by);
      }
    }
    // ---- rewards (:164-194 / S1L2F2_10Targets_wide_rewardFixed.py:127-151), benchmark_data (:136-148) --
    if (want_rew || want_info || want_stats) {
      const Win<T> wrew(p.rew + blk0, ld), winfo(p.info + blk0, ld);
      T rew[A];
      int coll[A];
      int follower_hits = 0;
#pragma unroll
      for (int o = 0; o < A; ++o) follower_hits += bit(d.advleader, o) ? 0 : hits[o];
#pragma unroll
      for (int i = 0; i < A; ++i) {
        const bool lead = bit(d.advleader, i), super = bit(d.advsuper, i);
        const int fi = desc_i8(d.force_id, i);
        const bool own = d.reward_variant == 0 ? !lead : (!super && !lead);
        int n = 0;
        T unit = T(10);
        if (own) {
          n = hits[i];
        } else if (d.reward_variant == 0 && super) {
          n = follower_hits;
        } else if (d.reward_variant == 1 && super) {
          unit = T(1);
#pragma unroll
          for (int o = 0; o < A; ++o) n += (desc_i8(d.forced_id, o) == fi && T(0) < d.obs_r2[o]) ? L : 0;
        } else {
#pragma unroll
          for (int o = 0; o < A; ++o) n += desc_i8(d.forced_id, o) == fi ? hits[o] : 0;
        }
        T r_i = unit * T(n);
        r_i = r_i - tag_bound_sel<T>(t_abs<T>(px[i]));
        r_i = r_i - tag_bound_sel<T>(t_abs<T>(py[i]));
        rew[i] = r_i;
        int c_i = 0;
        const bool i_adv = bit(d.adversary, i);
#pragma unroll
        for (int a = 0; a < A; ++a)
          if (bit(d.adversary, a) != i_adv)
            c_i += touching<T>(px[a] - px[i], py[a] - py[i], d.size[a] + d.size[i]) ? 1 : 0;
        coll[i] = c_i;
        if (want_rew) wrew.st_stream(i, lo, r_i);
        if (want_info) winfo.st_stream(i, lo, T(c_i));
      }
      if (want_stats) {
        constexpr int W = 2 * A + 2;
        constexpr int NW = THREADS / 64;
        __shared__ double sred[NW][W];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        // contact counters (<= A-1 <= 4 per lane -> <= 256 per wavefront... 9 bits): two words
        uint32_t pk0 = 0, pk1 = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) {
          if (i < 3) pk0 |= (uint32_t)coll[i] << (10 * i);
          else pk1 |= (uint32_t)coll[i] << (10 * (i - 3));
        }
        pk0 = wave_sum_u32(active ? pk0 : 0u);
        if (A > 3) pk1 = wave_sum_u32(active ? pk1 : 0u);
        const uint32_t fh = wave_sum_u32(active ? (uint32_t)follower_hits : 0u);
        const double n_active = (double)__builtin_popcountll(__ballot(active));
        double rsum[A];
#pragma unroll
        for (int i = 0; i < A; ++i) rsum[i] = group_sum(active ? (double)rew[i] : 0.0, 64);
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < A; ++i) {
            sred[wave][i] = rsum[i];
            sred[wave][A + i] = (double)(((i < 3 ? pk0 : pk1) >> (10 * (i < 3 ? i : i - 3))) & 0x3FFu);
          }
          sred[wave][2 * A] = (double)fh;
          sred[wave][2 * A + 1] = n_active;
        }
        __syncthreads();
        if (threadIdx.x < W) {
          double s = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) s += sred[w][threadIdx.x];
          p.stats[(int64_t)blockIdx.x * W + threadIdx.x] += s;
        }
      }
    }
  }
}

// The two *_MovingTarget files (L1F4_MovingTarget.py: A = 6, NF = 4, C = 3; S1L2F4_MovingTarget.py: A = 8, NF = 4,
// C = 7): no landmarks, the last agent is the prey.  Same step as above; own observation (distance, not
// squared-distance, radius test; command links visible only in the L1F4 text; the prey sees everybody in
// absolute coordinates) and reward text (10 per (prey, anybody) contact, +50 from four contacts on, the
// same for every adversary).  Stock flags checked on the host (mpe_hip.hip:moving_variant).
#ifndef MPE_MOVING_CAP
#define MPE_MOVING_CAP 2  // S1L2F4_MovingTarget 1Mi f32: 388.8 us at 3, 376.0 at 2, 387.5 at 4 (477 rows per wavefront)
#endif
#ifndef MPE_MOVING_ATTR
#define MPE_MOVING_ATTR __attribute__((amdgpu_waves_per_eu(1, MPE_MOVING_CAP)))
#endif
template <typename T, int A, int NF, int C, int THREADS, bool DO_STEP, bool DO_OBS, int RES = 0>
__global__ __launch_bounds__(THREADS) MPE_MOVING_ATTR void moving_kernel(const KParams<T> p) {
  constexpr int E = A, G = A - 1;  // G: the prey
  const DevDesc<T> &d = p.d;
  const int64_t ld = p.ld;
  const int64_t blk0 = env_block() * THREADS;
  const bool active = blk0 + threadIdx.x < p.B;
  const uint32_t lo = active ? threadIdx.x : kNoLane;
  const Win<T> wpos(p.pos + blk0, ld), wvel(p.vel + blk0, ld);
  T px[E], py[E], vx[A], vy[A], lfx[A], lfy[A];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    px[e] = wpos.ld_state(2 * e, lo);
    py[e] = wpos.ld_state(2 * e + 1, lo);
  }
#pragma unroll
  for (int a = 0; a < A; ++a) {
    vx[a] = wvel.template ld_state_nt<(RES >= 4)>(2 * a, lo);
    vy[a] = wvel.template ld_state_nt<(RES >= 4)>(2 * a + 1, lo);
    lfx[a] = lfy[a] = T(0);
  }
  hier_step<T, A, E, NF, 1, C, DO_STEP, DO_OBS, E, RES>(p, blk0, lo, px, py, vx, vy, lfx, lfy, [](int, int) { return T(0); });

  if (DO_OBS) {
    const bool want_rew = p.rew != nullptr, want_info = p.info != nullptr, want_stats = p.stats != nullptr;
    if (p.obs != nullptr) {
      // who is inside whose radius (:233-236: np.sqrt(np.sum(np.square(delta))) < agent.obs_r): the distance of a pair
      // is the same from both ends ((-dx)^2 == dx^2 exactly), so each unordered pair is measured once and tested
      // against both radii; touching_d2 is that very comparison (f64: decided on the squares outside a 2^-50 band
      // around the radius, the reference's square root inside it).  Bit (8 i + j): j is within i's radius.
      static_assert(A <= 8, "visibility bits: 8 x 8");
      uint64_t near = 0;
#ifndef MPE_MOVING_SQRT_VIS
#pragma unroll
      for (int i = 0; i < A; ++i) {
#pragma unroll
        for (int j = i + 1; j < A; ++j) {
          const T dx = px[j] - px[i], dy = py[j] - py[i];
          const T d2 = dx * dx + dy * dy;
          if (i != G) near |= (uint64_t)touching_d2<T>(d2, d.obs_r[i]) << (8 * i + j);
          if (j != G) near |= (uint64_t)touching_d2<T>(d2, d.obs_r[j]) << (8 * j + i);
        }
      }
#endif
#pragma unroll
      for (int i = 0; i < A; ++i) {  // L1F4_MovingTarget.py:214-258
        const Win<T> wobs(p.obs + blk0 + (int64_t)d.obs_row0[i] * ld, ld);
        int r = 0;
        wobs.st_stream(r++, lo, vx[i]);
        wobs.st_stream(r++, lo, vy[i]);
        wobs.st_stream(r++, lo, px[i]);
        wobs.st_stream(r++, lo, py[i]);
        const int fi = desc_i8(d.force_id, i), fdi = desc_i8(d.forced_id, i);
        bool vis[A];
#pragma unroll
        for (int j = 0; j < A; ++j) {
          if (j == i) continue;
          const T dx = px[j] - px[i], dy = py[j] - py[i];
          const bool linked = d.reward_variant == 0 &&
                              ((fi >= 0 && fi == desc_i8(d.forced_id, j)) || (fdi >= 0 && fdi == desc_i8(d.force_id, j)));
#ifndef MPE_MOVING_SQRT_VIS
          vis[j] = i == G || linked || ((near >> (8 * i + j)) & 1);
#else
          vis[j] = i == G || linked || t_sqrt<T>(dx * dx + dy * dy) < d.obs_r[i];
#endif
          wobs.st_stream(r++, lo, i == G ? px[j] : (vis[j] ? dx : T(0)));  // the prey: absolute (:249-251)
          wobs.st_stream(r++, lo, i == G ? py[j] : (vis[j] ? dy : T(0)));
        }
#pragma unroll
        for (int j = 0; j < A; ++j) {
          if (j == i) continue;
          wobs.st_stream(r++, lo, vis[j] ? vx[j] : T(0));
          wobs.st_stream(r++, lo, vis[j] ? vy[j] : T(0));
        }
        if (bit(d.obs_lf, i)) {
          wobs.st_stream(r++, lo, lfx[i]);
          wobs.st_stream(r++, lo, lfy[i]);
        }
      }
    }
    if (want_rew || want_info || want_stats) {
      int coll[A];  // benchmark_data (:128-140): adversary <-> prey contacts
      int count = 0;
#pragma unroll
      for (int o = 0; o < G; ++o) {
        coll[o] = touching<T>(px[G] - px[o], py[G] - py[o], d.size[G] + d.size[o]) ? 1 : 0;
        count += coll[o];
      }
      coll[G] = count;
      const T adv_r = T(10 * count + (count >= 4 ? 50 : 0));  // :186-199 (integers: exact)
      T rew[A];
      const Win<T> wrew(p.rew + blk0, ld), winfo(p.info + blk0, ld);
#pragma unroll
      for (int i = 0; i < A; ++i) {
        T r_i = i == G ? T(0) - T(count) : adv_r;  // :162-171
        r_i = r_i - tag_bound_sel<T>(t_abs<T>(px[i]));
        r_i = r_i - tag_bound_sel<T>(t_abs<T>(py[i]));
        rew[i] = r_i;
        if (want_rew) wrew.st_stream(i, lo, r_i);
        if (want_info) winfo.st_stream(i, lo, T(coll[i]));
      }
      if (want_stats) {
        constexpr int W = 2 * A + 2;
        constexpr int NW = THREADS / 64;
        __shared__ double sred[NW][W];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        // A - 1 one-bit contact flags (<= 64 per wavefront: 7 bits) + their sum (<= 7 per lane: 9 bits)
        uint64_t pk = 0;
#pragma unroll
        for (int i = 0; i < G; ++i) pk |= (uint64_t)coll[i] << (7 * i);
        uint32_t pk_lo = (uint32_t)(pk & 0xFFFFFFFu), pk_hi = (uint32_t)(pk >> 28);
        pk_lo = wave_sum_u32(active ? pk_lo : 0u);
        pk_hi = wave_sum_u32(active ? pk_hi : 0u);
        const uint32_t caught = wave_sum_u32(active ? (uint32_t)count : 0u);
        const double n_active = (double)__builtin_popcountll(__ballot(active));
        double rsum[A];
#pragma unroll
        for (int i = 0; i < A; ++i) rsum[i] = group_sum(active ? (double)rew[i] : 0.0, 64);
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < A; ++i) {
            sred[wave][i] = rsum[i];
            sred[wave][A + i] = i == G ? (double)caught
                                       : (double)(((i < 4 ? pk_lo : pk_hi) >> (7 * (i < 4 ? i : i - 4))) & 0x7Fu);
          }
          sred[wave][2 * A] = (double)caught;
          sred[wave][2 * A + 1] = n_active;
        }
        __syncthreads();
        if (threadIdx.x < W) {
          double s = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) s += sred[w][threadIdx.x];
          p.stats[(int64_t)blockIdx.x * W + threadIdx.x] += s;
        }
      }
    }
  }
}

}  // namespace mpe
