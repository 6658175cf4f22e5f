// Register-resident env-step kernels for the speaking stock scenarios in their stock shapes:
//   MPE_SPEAKER_LISTENER  agent 0 fixed + speaking (3 comm rows), agent 1 moving + silent (5 rows), dim_c 3
//   MPE_SIMPLE_REFERENCE  two moving + speaking agents (5 + 10 rows each), dim_c 10
// both with 3 landmarks and no collisions (simple_speaker_listener.py:10-34, simple_reference.py:6-25),
// and simple_crypto (crypto_kernel at the end of this file).
// One env per lane, everything in VGPRs, straight-line (see mpe_small_kernel.h for the access
// conventions: Win row windows, nontemporal streams, occupancy cap).  Stock flags checked on the host
// (mpe_hip.hip:comm_variant): nothing collides, no max_speed, no u_noise.  Other shapes run the tile
// kernel (same arithmetic: tests/test_gpu_comm.py checks the two bit for bit).
#pragma once
#include "mpe_device.h"

namespace mpe {

constexpr int comm_wave_cap(int kind) { return kind == MPE_SIMPLE_REFERENCE ? 3 : 4; }

template <typename T, int KIND, int C, int THREADS, bool DO_STEP, bool DO_OBS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, comm_wave_cap(KIND)))) void comm_kernel(
    const KParams<T> p) {
  constexpr bool REF = KIND == MPE_SIMPLE_REFERENCE;
  constexpr int A = 2, L = 3, E = A + L;
  constexpr int M0 = REF ? 0 : 1;  // first moving agent (the speaker of speaker_listener is fixed)
  const DevDesc<T> &d = p.d;
  const int64_t ld = p.ld;
  const int64_t blk0 = env_block() * THREADS;
  const bool active = blk0 + threadIdx.x < p.B;
  const uint32_t lo = active ? threadIdx.x : kNoLane;
  // action rows: speaker_listener [c0 c1 c2 | u0..u4]; reference [u0..u4 c0..c9 | u0..u4 c0..c9]
  auto urow = [](int a) { return REF ? a * (5 + C) : C; };
  auto crow = [](int a) { return REF ? a * (5 + C) + 5 : 0; };
  auto speaks = [](int a) { return REF || a == 0; };

  const Win<T> wpos(p.pos + blk0, ld), wvel(p.vel + blk0, ld);
  T px[E], py[E], vx[A], vy[A], c[A][C];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    px[e] = wpos.ld_state(2 * e, lo);
    py[e] = wpos.ld_state(2 * e + 1, lo);
  }
#pragma unroll
  for (int a = M0; a < A; ++a) {
    vx[a] = wvel.ld_state(2 * a, lo);
    vy[a] = wvel.ld_state(2 * a + 1, lo);
  }
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int k = 0; k < C; ++k) c[a][k] = T(0);

  if (DO_STEP) {
    const Win<T> wact(p.act + blk0, ld);
    T a1[A], a2[A], a3[A], a4[A];
#pragma unroll
    for (int a = M0; a < A; ++a) {  // a[0] (the no-op slot) is never read (environment.py:189-190)
      a1[a] = wact.ld_stream(urow(a) + 1, lo);
      a2[a] = wact.ld_stream(urow(a) + 2, lo);
      a3[a] = wact.ld_stream(urow(a) + 3, lo);
      a4[a] = wact.ld_stream(urow(a) + 4, lo);
    }
#pragma unroll
    for (int a = 0; a < A; ++a)
      if (speaks(a)) {
#pragma unroll
        for (int k = 0; k < C; ++k) c[a][k] = wact.ld_stream(crow(a) + k, lo);
      }
    if (p.comm != nullptr) {  // core.py:226-230
      const Win<T> wcomm(p.comm + blk0, ld);
#pragma unroll
      for (int a = 0; a < A; ++a) {
        if (!speaks(a) && !rewrite_silent_comm(p)) continue;
#pragma unroll
        for (int k = 0; k < C; ++k) wcomm.st_state(a * C + k, lo, c[a][k]);
      }
    }
#pragma unroll
    for (int a = M0; a < A; ++a) {  // environment.py:188-196, core.py:211-222 (nothing collides)
      const T Fx = (a1[a] - a2[a]) * d.accel[a], Fy = (a3[a] - a4[a]) * d.accel[a];
      integrate_vel_sel<T, false>(vx[a], vy[a], Fx, Fy, d.mass[a], d.damp1, d.dt, d.max_speed[a]);
      px[a] = px[a] + vx[a] * d.dt;
      py[a] = py[a] + vy[a] * d.dt;
      wvel.st_state(2 * a, lo, vx[a]);
      wvel.st_state(2 * a + 1, lo, vy[a]);
      wpos.st_state(2 * a, lo, px[a]);
      wpos.st_state(2 * a + 1, lo, py[a]);
    }
  } else if (DO_OBS) {  // observing only: agent.state.c comes from the comm buffer
    const Win<T> wcomm(p.comm + blk0, ld);
#pragma unroll
    for (int a = 0; a < A; ++a)
      if (speaks(a)) {
#pragma unroll
        for (int k = 0; k < C; ++k) c[a][k] = wcomm.ld_state(a * C + k, lo);
      }
  }

  if (DO_OBS) {
    const bool want_rew = p.rew != nullptr, want_stats = p.stats != nullptr;
    int g[A];  // agent.goal_b landmark index
    const int64_t gi = blk0 + (active ? threadIdx.x : 0);
#pragma unroll
    for (int a = 0; a < (REF ? A : 1); ++a) {
      g[a] = p.goal[(int64_t)a * ld + gi];
      g[a] = g[a] < 0 ? 0 : (g[a] >= L ? L - 1 : g[a]);
    }
    const T base = REF ? T(0.25) : T(0.15), hi = REF ? T(0.75) : T(0.65);  // landmark.color tables
    if (p.obs != nullptr) {
#pragma unroll
      for (int i = 0; i < A; ++i) {
        const Win<T> wobs(p.obs + blk0 + (int64_t)d.obs_row0[i] * ld, ld);
        int r = 0;
        if (!REF && i == 0) {  // speaker: the goal colour (simple_speaker_listener.py:89-91)
#pragma unroll
          for (int k = 0; k < 3; ++k) wobs.st_stream(r++, lo, k == g[0] ? hi : base);
          continue;
        }
        wobs.st_stream(r++, lo, vx[i]);
        wobs.st_stream(r++, lo, vy[i]);
#pragma unroll
        for (int l = A; l < E; ++l) {
          wobs.st_stream(r++, lo, px[l] - px[i]);
          wobs.st_stream(r++, lo, py[l] - py[i]);
        }
        if (REF) {
#pragma unroll
          for (int k = 0; k < 3; ++k) wobs.st_stream(r++, lo, k == g[i] ? hi : base);
        }
#pragma unroll
        for (int k = 0; k < C; ++k) wobs.st_stream(r++, lo, c[1 - i][k]);  // the other agent's state.c
      }
    }
    if (want_rew || want_stats) {
      const Win<T> wrew(p.rew + blk0, ld);
      T rew[A];
#pragma unroll
      for (int i = 0; i < A; ++i) {
        // simple_speaker_listener.py:66-70: listener to agent 0's goal, for both agents;
        // simple_reference.py:56-60: the OTHER agent to agent i's goal
        const int who = REF ? 1 - i : 1, gl = g[REF ? i : 0];
        T lx = px[A], ly = py[A];
#pragma unroll
        for (int l = 1; l < L; ++l) {
          lx = gl == l ? px[A + l] : lx;
          ly = gl == l ? py[A + l] : ly;
        }
        rew[i] = -sqdist<T>(px[who], py[who], lx, ly);
        if (want_rew) wrew.st_stream(i, lo, rew[i]);
      }
      if (want_stats) {
        constexpr int W = 2 * A + 2;
        constexpr int NW = THREADS / 64;
        __shared__ double sred[NW][A + 1];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < A; ++i) {
          const double s_r = group_sum(active ? (double)rew[i] : 0.0, 64);
          if (lane == 0) sred[wave][i] = s_r;
        }
        const double n_active = (double)__builtin_popcountll(__ballot(active));  // all lanes vote
        if (lane == 0) sred[wave][A] = n_active;
        __syncthreads();
        if (threadIdx.x <= A) {
          double s = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) s += sred[w][threadIdx.x];
          p.stats[(int64_t)blockIdx.x * W + (threadIdx.x < A ? threadIdx.x : 2 * A + 1)] += s;
        }
      }
    }
  }
}

// simple_crypto in its stock shape (3 fixed speaking agents, 2 landmarks, dim_c 4; simple_crypto.py:24-58):
// no physics at all -- the step copies the three messages into state.c and scores them against the goal
// colour.  Pure streaming: 12 action rows in, 12 comm + 20 obs + 3 reward (+ 24 info) rows out.
#ifndef MPE_CRYPTO_CAP
#define MPE_CRYPTO_CAP 3  // 2Mi envs f32: 102.4 us uncapped (8) or at 6, 97.4 at 5, 94.1 at 4, 93.3 at 3, 102.0 at 2
#endif
template <typename T, int THREADS, bool DO_STEP, bool DO_OBS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, MPE_CRYPTO_CAP))) void crypto_kernel(const KParams<T> p) {
  constexpr int A = 3, C = 4;
  const DevDesc<T> &d = p.d;
  const int64_t ld = p.ld;
  const int64_t blk0 = env_block() * THREADS;
  const bool active = blk0 + threadIdx.x < p.B;
  const uint32_t lo = active ? threadIdx.x : kNoLane;
  T c[A][C];
  if (DO_STEP) {
    const Win<T> wact(p.act + blk0, ld);  // action rows: [c0..c3] per agent (nobody moves)
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int k = 0; k < C; ++k) c[a][k] = wact.ld_stream(a * C + k, lo);
    if (p.comm != nullptr) {  // core.py:226-230
      const Win<T> wcomm(p.comm + blk0, ld);
#pragma unroll
      for (int a = 0; a < A; ++a)
#pragma unroll
        for (int k = 0; k < C; ++k) wcomm.st_state(a * C + k, lo, c[a][k]);
    }
  } else {
    const Win<T> wcomm(p.comm + blk0, ld);
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int k = 0; k < C; ++k) c[a][k] = wcomm.ld_state(a * C + k, lo);
  }
  if (DO_OBS) {
    const bool want_rew = p.rew != nullptr, want_stats = p.stats != nullptr;
    const int64_t gi = blk0 + (active ? threadIdx.x : 0);
    const int gl = p.goal[gi], key = p.goal[ld + gi];
    if (p.obs != nullptr) {
      const Win<T> wobs(p.obs + blk0, ld);  // rows: Eve [c2] | Bob [key, c2] | Alice [goal, key]
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const T kk = k == key ? T(1) : T(0);
        wobs.st_stream(k, lo, c[2][k]);
        wobs.st_stream(C + k, lo, kk);
        wobs.st_stream(2 * C + k, lo, c[2][k]);
        wobs.st_stream(3 * C + k, lo, k == gl ? T(1) : T(0));
        wobs.st_stream(4 * C + k, lo, kk);
      }
    }
    if (p.info != nullptr) {  // (agent.state.c, agent.goal_a.color), simple_crypto.py:78-80
      const Win<T> winfo(p.info + blk0, ld);
#pragma unroll
      for (int a = 0; a < A; ++a)
#pragma unroll
        for (int k = 0; k < C; ++k) {
          winfo.st_stream(2 * C * a + k, lo, c[a][k]);
          winfo.st_stream(2 * C * a + C + k, lo, k == gl ? T(1) : T(0));
        }
    }
    if (want_rew || want_stats) {
      // simple_crypto.py:96-121: squared error against the goal colour, summed left to right; an all-zero
      // message switches its term off
      T err[2];
      bool spoke[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        T s = T(0);
        spoke[a] = false;
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const T dlt = c[a][k] - (k == gl ? T(1) : T(0));
          s = k == 0 ? dlt * dlt : s + dlt * dlt;
          spoke[a] = spoke[a] || c[a][k] != T(0);
        }
        err[a] = s;
      }
      T rew[A];
      rew[0] = spoke[0] ? -err[0] : T(0);
      rew[1] = (spoke[0] ? err[0] : T(0)) + (spoke[1] ? -err[1] : T(0));  // adv_rew + good_rew
      rew[2] = rew[1];
      if (want_rew) {
        const Win<T> wrew(p.rew + blk0, ld);
#pragma unroll
        for (int i = 0; i < A; ++i) wrew.st_stream(i, lo, rew[i]);
      }
      if (want_stats) {
        constexpr int W = 2 * A + 2;
        constexpr int NW = THREADS / 64;
        __shared__ double sred[NW][A + 1];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < A; ++i) {
          const double s_r = group_sum(active ? (double)rew[i] : 0.0, 64);
          if (lane == 0) sred[wave][i] = s_r;
        }
        const double n_active = (double)__builtin_popcountll(__ballot(active));  // all lanes vote
        if (lane == 0) sred[wave][A] = n_active;
        __syncthreads();
        if (threadIdx.x <= A) {
          double s = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) s += sred[w][threadIdx.x];
          p.stats[(int64_t)blockIdx.x * W + (threadIdx.x < A ? threadIdx.x : 2 * A + 1)] += s;
        }
      }
    }
  }
  (void)d;
}

}  // namespace mpe
