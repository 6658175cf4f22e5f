// Register-resident env-step kernel for simple_world_comm in its stock shape
// (build/lib/multiagent/scenarios/simple_world_comm.py:11-61): 4 adversaries + 2 good agents, landmarks =
// 1 colliding obstacle, 2 food, 2 forests; agent 0 is the only speaker (dim_c 4: action rows
// [u0..u4 c0..c3 | u0..u4 | ... ]).  One env per lane, straight-line code, same access conventions as
// mpe_small_kernel.h (Win row windows, nontemporal streams, occupancy cap).  Stock flags are checked on the
// host (mpe_hip.hip:wcomm_variant): agents and the obstacle collide, food and forests do not, every agent
// has a max_speed, no u_noise.  Sizes, accel, mass, leader flags and the contact constants stay run-time
// values.  Other shapes run the tile kernel (same arithmetic: tests/test_gpu_world_comm.py checks the two
// bit for bit).
#pragma once
#include "mpe_device.h"

namespace mpe {

#ifndef MPE_WCOMM_CAP
#define MPE_WCOMM_CAP 2  // 1Mi envs f32: 217.3 us at 4, 208.9 at 3, 203.6 at 2 (320 rows per wavefront)
#endif
// RES: the launch's Infinity Cache plan (mpe_device.h:residency_plan); 3 = the landmark rows are streamed past the cache
template <typename T, int THREADS, bool DO_STEP, bool DO_OBS, int RES = 0>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, MPE_WCOMM_CAP))) void wcomm_kernel(const KParams<T> p) {
  constexpr int A = 6, NADV = 4, L = 5, E = A + L, ECOL = A + 1, C = 4;
  constexpr int FOOD0 = A + 1, FOOD1 = A + 3, FOREST0 = A + 3;
  const DevDesc<T> &d = p.d;
  const int64_t ld = p.ld;
  const int64_t blk0 = env_block() * THREADS;
  const bool active = blk0 + threadIdx.x < p.B;
  const uint32_t lo = active ? threadIdx.x : kNoLane;
  auto urow = [](int a) { return a == 0 ? 0 : 5 + C + 5 * (a - 1); };

  const Win<T> wpos(p.pos + blk0, ld), wvel(p.vel + blk0, ld);
  T px[E], py[E], vx[A], vy[A], c0[C];
#pragma unroll
  for (int e = 0; e < A; ++e) {
    px[e] = wpos.ld_state(2 * e, lo);
    py[e] = wpos.ld_state(2 * e + 1, lo);
  }
#pragma unroll
  for (int e = A; e < E; ++e) {  // fixed landmarks: read-only rows
    px[e] = wpos.template ld_state_nt<(RES >= 3)>(2 * e, lo);
    py[e] = wpos.template ld_state_nt<(RES >= 3)>(2 * e + 1, lo);
  }
#pragma unroll
  for (int a = 0; a < A; ++a) {
    vx[a] = wvel.ld_state(2 * a, lo);
    vy[a] = wvel.ld_state(2 * a + 1, lo);
  }

  if (DO_STEP) {
    const Win<T> wact(p.act + blk0, ld);
    T a1[A], a2[A], a3[A], a4[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {  // a[0] (the no-op slot) is never read (environment.py:189-190)
      a1[a] = wact.ld_stream(urow(a) + 1, lo);
      a2[a] = wact.ld_stream(urow(a) + 2, lo);
      a3[a] = wact.ld_stream(urow(a) + 3, lo);
      a4[a] = wact.ld_stream(urow(a) + 4, lo);
    }
#pragma unroll
    for (int k = 0; k < C; ++k) c0[k] = wact.ld_stream(5 + k, lo);
    if (p.comm != nullptr) {  // core.py:226-230: the leader's message, zeros for the silent agents
      const Win<T> wcomm(p.comm + blk0, ld);
#pragma unroll
      for (int k = 0; k < C; ++k) wcomm.st_state(k, lo, c0[k]);
      if (rewrite_silent_comm(p)) {
#pragma unroll
        for (int r = C; r < A * C; ++r) wcomm.st_state(r, lo, T(0));
      }
    }
    T Fx[A], Fy[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {  // environment.py:188-196
      Fx[a] = (a1[a] - a2[a]) * d.accel[a];
      Fy[a] = (a3[a] - a4[a]) * d.accel[a];
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {  // core.py:199-205, b ascending; only a is pushed
#pragma unroll
      for (int b = a + 1; b < ECOL; ++b) {
        T fx, fy;
        pair_force<T>(px[a] - px[b], py[a] - py[b], d.size[a] + d.size[b], d, fx, fy);
        Fx[a] = fx + Fx[a];
        Fy[a] = fy + Fy[a];
      }
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {  // core.py:211-222
      integrate_vel_sel<T, true>(vx[a], vy[a], Fx[a], Fy[a], d.mass[a], d.damp1, d.dt, d.max_speed[a]);
      px[a] = px[a] + vx[a] * d.dt;
      py[a] = py[a] + vy[a] * d.dt;
      wvel.st_state(2 * a, lo, vx[a]);
      wvel.st_state(2 * a + 1, lo, vy[a]);
      wpos.st_state(2 * a, lo, px[a]);
      wpos.st_state(2 * a + 1, lo, py[a]);
    }
  } else {  // observing only: the leader's state.c comes from the comm buffer
    const Win<T> wcomm(p.comm + blk0, ld);
#pragma unroll
    for (int k = 0; k < C; ++k) c0[k] = wcomm.ld_state(k, lo);
  }

  if (DO_OBS) {
    auto hit = [&](int a, int b) { return touching<T>(px[a] - px[b], py[a] - py[b], d.size[a] + d.size[b]); };
    auto dist = [&](int a, int b) { return t_sqrt<T>(sqdist<T>(px[a], py[a], px[b], py[b])); };
    if (p.obs != nullptr) {
      bool f0[A], f1[A];  // is_collision(agent, forests[0 / 1]) (simple_world_comm.py:240-245,258-259)
#pragma unroll
      for (int a = 0; a < A; ++a) {
        f0[a] = hit(a, FOREST0);
        f1[a] = hit(a, FOREST0 + 1);
      }
#pragma unroll
      for (int i = 0; i < A; ++i) {
        const Win<T> wobs(p.obs + blk0 + (int64_t)d.obs_row0[i] * ld, ld);
        int r = 0;
        wobs.st_stream(r++, lo, vx[i]);
        wobs.st_stream(r++, lo, vy[i]);
        wobs.st_stream(r++, lo, px[i]);
        wobs.st_stream(r++, lo, py[i]);
#pragma unroll
        for (int l = A; l < E; ++l) {
          wobs.st_stream(r++, lo, px[l] - px[i]);
          wobs.st_stream(r++, lo, py[l] - py[i]);
        }
        const bool lead = bit(d.leader, i);
        bool vis[A];  // same forest, or nobody in any forest, or the leader looks (:261)
#pragma unroll
        for (int j = 0; j < A; ++j) {
          if (j == i) continue;
          vis[j] = (f0[i] && f0[j]) || (f1[i] && f1[j]) || (!f0[i] && !f0[j] && !f1[i] && !f1[j]) || lead;
          wobs.st_stream(r++, lo, vis[j] ? px[j] - px[i] : T(0));
          wobs.st_stream(r++, lo, vis[j] ? py[j] - py[i] : T(0));
        }
        if (i >= NADV) {  // good agents: in_forest before other_vel (:313)
          wobs.st_stream(r++, lo, f0[i] ? T(1) : T(-1));
          wobs.st_stream(r++, lo, f1[i] ? T(1) : T(-1));
        }
#pragma unroll
        for (int j = NADV; j < A; ++j) {
          if (j == i) continue;
          wobs.st_stream(r++, lo, vis[j] ? vx[j] : T(0));
          wobs.st_stream(r++, lo, vis[j] ? vy[j] : T(0));
        }
        if (i < NADV) {  // adversaries: other_vel, in_forest, agent 0's state.c (:292,305-310)
          wobs.st_stream(r++, lo, f0[i] ? T(1) : T(-1));
          wobs.st_stream(r++, lo, f1[i] ? T(1) : T(-1));
#pragma unroll
          for (int k = 0; k < C; ++k) wobs.st_stream(r++, lo, c0[k]);
        }
      }
    }
    const bool want_rew = p.rew != nullptr, want_info = p.info != nullptr, want_stats = p.stats != nullptr;
    if (want_rew || want_info || want_stats) {
      // simple_world_comm.py:119-204, sums in the reference's order (the running reward is not an integer)
      bool h[A - NADV][NADV];
      int coll[NADV];
#pragma unroll
      for (int a = 0; a < NADV; ++a) coll[a] = 0;
#pragma unroll
      for (int g = NADV; g < A; ++g)
#pragma unroll
        for (int a = 0; a < NADV; ++a) {
          h[g - NADV][a] = hit(g, a);
          coll[a] += h[g - NADV][a] ? 1 : 0;
        }
      T rew[A];
#pragma unroll
      for (int i = 0; i < NADV; ++i) {
        T m = dist(NADV, i);
#pragma unroll
        for (int g = NADV + 1; g < A; ++g) {
          const T dg = dist(g, i);
          m = dg < m ? dg : m;
        }
        T r = T(0) - T(0.1) * m;
#pragma unroll
        for (int g = NADV; g < A; ++g)
#pragma unroll
          for (int a = 0; a < NADV; ++a) r = h[g - NADV][a] ? r + T(5) : r;
        rew[i] = r;
      }
#pragma unroll
      for (int i = NADV; i < A; ++i) {
        T r = T(0);
#pragma unroll
        for (int a = 0; a < NADV; ++a) r = h[i - NADV][a] ? r - T(5) : r;
        r = r - T(2) * tag_bound_sel<T>(t_abs<T>(px[i]));
        r = r - T(2) * tag_bound_sel<T>(t_abs<T>(py[i]));
        T m = T(0);
#pragma unroll
        for (int f = FOOD0; f < FOOD1; ++f) {
          r = hit(i, f) ? r + T(2) : r;
          const T df = dist(f, i);
          m = (f == FOOD0 || df < m) ? df : m;
        }
        rew[i] = r + T(0.05) * m;
      }
      if (want_rew) {
        const Win<T> wrew(p.rew + blk0, ld);
#pragma unroll
        for (int i = 0; i < A; ++i) wrew.st_stream(i, lo, rew[i]);
      }
      if (want_info) {
        const Win<T> winfo(p.info + blk0, ld);
#pragma unroll
        for (int i = 0; i < A; ++i) winfo.st_stream(i, lo, i < NADV ? T(coll[i]) : T(0));
      }
      if (want_stats) {
        constexpr int W = 2 * A + 2;
        constexpr int NW = THREADS / 64;
        __shared__ double sred[NW][W];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t packed = 0;  // four counters <= 2 per lane, <= 128 per wavefront: one byte each
#pragma unroll
        for (int a = 0; a < NADV; ++a) packed |= (uint32_t)coll[a] << (8 * a);
        packed = wave_sum_u32(active ? packed : 0u);
        const double n_active = (double)__builtin_popcountll(__ballot(active));
        double rsum[A];
#pragma unroll
        for (int i = 0; i < A; ++i) rsum[i] = group_sum(active ? (double)rew[i] : 0.0, 64);
        if (lane == 0) {
          double tot = 0.0;
#pragma unroll
          for (int i = 0; i < A; ++i) {
            const double c = i < NADV ? (double)((packed >> (8 * i)) & 0xFFu) : 0.0;
            sred[wave][i] = rsum[i];
            sred[wave][A + i] = c;
            tot += c;
          }
          sred[wave][2 * A] = tot;
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
