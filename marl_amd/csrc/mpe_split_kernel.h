// Small-batch form of the simple_tag (3+1 agents, 2 landmarks) / simple_spread (3 agents, 3 landmarks) step:
// one AGENT per wavefront.  A workgroup is 64 envs x A wavefronts; lane <-> env as in mpe_small_kernel.h, wavefront
// index <-> agent.
//
// Why: a batch of a few thousand envs is a fraction of a wavefront per SIMD, and the one-env-per-lane kernel then runs
// at the latency of its own dependent instruction stream (~2 000 f64 instructions for simple_spread, a handful of them
// divide / sqrt / exp / log1p sequences).  Splitting an env's agents over wavefronts cuts the stream per wavefront by
// about A (agent a owns the pair forces (a, b > a), its integration, its observation rows, its reward) and puts A times
// as many wavefronts on the chip.  Per step inside a HIP-graph replay, MI355X r2 (wide -> split):
//   simple_spread f64   4 096 envs 8.64 -> 6.03 us   16 384: 9.49 -> 7.08   65 536: 11.7 -> 12.3   262 144: 41.1 -> 52.1
//   simple_tag    f32   4 096 envs 5.10 -> 4.29 us   16 384: 5.59 -> 4.83   65 536: 7.25 -> 7.28   1 Mi: 78.9 -> 102.2
// From 65 536 envs on the step is a bandwidth problem (5 TB/s at 65 536) and re-reading every position row A times
// costs more than the shorter stream saves: the default switches at MPE_SPLIT_MAX_ENVS = 32 768.
//
// The arithmetic per agent is the one-env-per-lane kernel's, operation for operation (same helpers, same order), so the
// two kernels agree bit for bit on state, observations, rewards and info (tests/test_gpu_split.py); the per-workgroup
// statistics rows group 64 envs instead of 256, so their f64 sums can differ in the last bits.
//
// Exchange: every wavefront integrates its own agent from the OLD positions of all entities (core.py:199-222 computes
// all forces before it moves anything), publishes the new position / velocity in LDS, and after one barrier reads the
// other agents' for its observation (simple_tag.py:131-147 / simple_spread.py:83-99) and reward.
#pragma once
#include "mpe_device.h"

namespace mpe {

template <typename T, int KIND, int A, int L, int NADV, int NOISE, bool DO_STEP, bool DO_OBS>
__global__ __launch_bounds__(64 * A) void split_kernel(const KParams<T> p) {
  static_assert(KIND == MPE_SIMPLE_TAG || KIND == MPE_SIMPLE_SPREAD, "north-star worlds only");
  constexpr int E = A + L;
  constexpr bool TAG = KIND == MPE_SIMPLE_TAG;
  constexpr int ECOL = TAG ? E : A;  // colliding entities (stock flags, checked on the host)
  constexpr int W = 2 * A + 2;
  const DevDesc<T> &d = p.d;
  const int64_t ld = p.ld;
  const uint32_t lane = threadIdx.x & 63u;
  const int me = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // this wavefront's agent: a scalar
  const int64_t blk0 = (int64_t)blockIdx.x * 64;
  const bool active = blk0 + lane < p.B;
  const uint32_t lo = active ? lane : kNoLane;  // tail lanes: see Win
  __shared__ T s_new[4 * A][64];                // new (pos.x, pos.y, vel.x, vel.y) of every agent
  __shared__ double s_stat[W];

  const Win<T> wpos(p.pos + blk0, ld), wvel(p.vel + blk0, ld);
  T px[E], py[E], vx[A], vy[A];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    px[e] = wpos.ld_state(2 * e, lo);
    py[e] = wpos.ld_state(2 * e + 1, lo);
  }
  T mvx = wvel.ld_state(2 * me, lo), mvy = wvel.ld_state(2 * me + 1, lo);
  // the statistics row is fetched up front: the read-modify-write at the end must not add a memory round trip
  double stats_prev = 0.0;
  if (DO_OBS && p.stats != nullptr && threadIdx.x < W) stats_prev = p.stats[(int64_t)blockIdx.x * W + threadIdx.x];
  T mpx = px[0], mpy = py[0];
#pragma unroll
  for (int i = 1; i < A; ++i) {
    mpx = me == i ? px[i] : mpx;
    mpy = me == i ? py[i] : mpy;
  }

  if (DO_STEP) {
    const Win<T> wact(p.act + blk0, ld), wnoise(p.noise + blk0, ld);
    const T a1 = wact.ld_stream(5 * me + 1, lo), a2 = wact.ld_stream(5 * me + 2, lo);  // environment.py:188-196
    const T a3 = wact.ld_stream(5 * me + 3, lo), a4 = wact.ld_stream(5 * me + 4, lo);
    T Fx = (a1 - a2) * d.accel[me], Fy = (a3 - a4) * d.accel[me];
    if (NOISE != 0) {  // core.py:147-148; u_noise == 0 encodes None.  1: the caller's normals, 2: drawn here
      T nx = T(0), ny = T(0);
      if (NOISE == 1) {
        nx = wnoise.ld_stream(2 * me, lo);
        ny = wnoise.ld_stream(2 * me + 1, lo);
      } else if (d.u_noise[me] != T(0)) {
        device_normals<T>(p, active ? blk0 + lane : p.B - 1, noise_step(p), kNoiseWordU | (uint32_t)me, nx, ny);
      }
      Fx = d.u_noise[me] != T(0) ? Fx + nx * d.u_noise[me] : Fx;
      Fy = d.u_noise[me] != T(0) ? Fy + ny * d.u_noise[me] : Fy;
    }
#pragma unroll
    for (int b = 1; b < ECOL; ++b) {  // core.py:199-205: b ascending, only `me` is pushed
      if (b > me) {
        T fx, fy;
        pair_force<T>(mpx - px[b], mpy - py[b], d.size[me] + d.size[b], d, fx, fy);
        Fx = fx + Fx;
        Fy = fy + Fy;
      }
    }
    integrate_vel_sel<T, TAG>(mvx, mvy, Fx, Fy, d.mass[me], d.damp1, d.dt, d.max_speed[me]);  // core.py:211-222
    mpx = mpx + mvx * d.dt;
    mpy = mpy + mvy * d.dt;
    wvel.st_state(2 * me, lo, mvx);
    wvel.st_state(2 * me + 1, lo, mvy);
    wpos.st_state(2 * me, lo, mpx);
    wpos.st_state(2 * me + 1, lo, mpy);
    if (p.comm != nullptr && rewrite_silent_comm(p)) {  // core.py:226-227
      const Win<T> wcomm(p.comm + blk0, ld);
      for (int r = me * d.dim_c; r < (me + 1) * d.dim_c; ++r) wcomm.st_state(r, lo, T(0));
    }
    if (DO_OBS) {
      s_new[4 * me + 0][lane] = mpx;
      s_new[4 * me + 1][lane] = mpy;
      s_new[4 * me + 2][lane] = mvx;
      s_new[4 * me + 3][lane] = mvy;
    }
  }

  if constexpr (DO_OBS) {
    if (DO_STEP) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < A; ++j) {
        px[j] = s_new[4 * j + 0][lane];
        py[j] = s_new[4 * j + 1][lane];
        vx[j] = s_new[4 * j + 2][lane];
        vy[j] = s_new[4 * j + 3][lane];
      }
    } else {
#pragma unroll
      for (int j = 0; j < A; ++j) {
        vx[j] = wvel.ld_state(2 * j, lo);
        vy[j] = wvel.ld_state(2 * j + 1, lo);
      }
    }
    // ---- observations (simple_tag.py:131-147 / simple_spread.py:83-99) ---------------------------
    if (p.obs != nullptr) {
      const Win<T> wobs(p.obs + blk0 + (int64_t)d.obs_row0[me] * ld, ld);  // this agent's rows
      int r = 0;
      wobs.st_stream(r++, lo, mvx);
      wobs.st_stream(r++, lo, mvy);
      wobs.st_stream(r++, lo, mpx);
      wobs.st_stream(r++, lo, mpy);
#pragma unroll
      for (int l = A; l < E; ++l) {
        wobs.st_stream(r++, lo, px[l] - mpx);
        wobs.st_stream(r++, lo, py[l] - mpy);
      }
#pragma unroll
      for (int j = 0; j < A; ++j) {
        if (j != me) {
          wobs.st_stream(r++, lo, px[j] - mpx);
          wobs.st_stream(r++, lo, py[j] - mpy);
        }
      }
      if (TAG) {
#pragma unroll
        for (int j = NADV; j < A; ++j) {
          if (j != me) {
            wobs.st_stream(r++, lo, vx[j]);
            wobs.st_stream(r++, lo, vy[j]);
          }
        }
      } else {
        for (int z = 0; z < d.dim_c * (A - 1); ++z) wobs.st_stream(r++, lo, T(0));
      }
    }
    // ---- reward / info ------------------------------------------------------------------------
    const bool want_rew = p.rew != nullptr, want_info = p.info != nullptr, want_stats = p.stats != nullptr;
    if (want_rew || want_info || want_stats) {
      const Win<T> wrew(p.rew + blk0, ld), winfo(p.info + blk0, ld);
      T rew;
      int coll = 0, extra = 0;
      if (TAG) {
        int total = 0, mine = 0;  // all (good, adversary) contacts of the env; those `me` takes part in
#pragma unroll
        for (int g = NADV; g < A; ++g) {
#pragma unroll
          for (int a = 0; a < NADV; ++a) {
            const int hit = touching<T>(px[g] - px[a], py[g] - py[a], d.size[g] + d.size[a]) ? 1 : 0;
            total += hit;
            mine += (a == me || g == me) ? hit : 0;
          }
        }
        if (me < NADV) {
          rew = T(10) * T(total);  // simple_tag.py:124-128
          coll = mine;             // simple_tag.py:57-64
        } else {
          T r = -T(10) * T(mine);                    // simple_tag.py:97-100
          r = r - tag_bound_sel<T>(t_abs<T>(mpx));  // simple_tag.py:109-111
          r = r - tag_bound_sel<T>(t_abs<T>(mpy));
          rew = r;
        }
        extra = total;
        if (want_info) winfo.st_stream(me, lo, T(coll));
      } else {
        T r0 = T(0), md = T(0);
        int occ = 0;
#pragma unroll
        for (int l = A; l < E; ++l) {  // simple_spread.py:74-76 / :51-56
          T m = T(0);
#pragma unroll
          for (int a = 0; a < A; ++a) {
            const T dx = px[a] - px[l], dy = py[a] - py[l];
            const T dist = t_sqrt<T>(dx * dx + dy * dy);
            m = (a == 0 || dist < m) ? dist : m;
          }
          md = md + m;
          r0 = r0 - m;
          occ += (m < T(0.1)) ? 1 : 0;
        }
        T r = r0;
#pragma unroll
        for (int a = 0; a < A; ++a) {  // includes a == me (simple_spread.py:78-80)
          const bool hit = touching<T>(px[a] - mpx, py[a] - mpy, d.size[a] + d.size[me]);
          r = hit ? r - T(1) : r;
          coll += hit ? 1 : 0;
        }
        rew = r;
        extra = occ;
        if (want_info) {
          winfo.st_stream(4 * me, lo, r);
          winfo.st_stream(4 * me + 1, lo, T(coll));
          winfo.st_stream(4 * me + 2, lo, md);
          winfo.st_stream(4 * me + 3, lo, T(occ));
        }
      }
      if (want_rew) wrew.st_stream(me, lo, rew);
      // ---- per-workgroup statistics: every wavefront owns its agent's two entries, agent 0 the shared ones ----
      if (want_stats) {
        const double s_r = group_sum(active ? (double)rew : 0.0, 64);
        // <= 3 per lane each, so <= 192 per wavefront: two counters in one word
        const uint32_t s_c = wave_sum_u32(active ? ((uint32_t)coll | (uint32_t)extra << 16) : 0u);
        const double n_active = (double)__builtin_popcountll(__ballot(active));
        if (lane == 0) {
          s_stat[me] = s_r;
          s_stat[A + me] = (double)(s_c & 0xFFFFu);
          if (me == 0) {
            s_stat[2 * A] = (double)(s_c >> 16);
            s_stat[2 * A + 1] = n_active;
          }
        }
        __syncthreads();
        if (threadIdx.x < W) p.stats[(int64_t)blockIdx.x * W + threadIdx.x] = stats_prev + s_stat[threadIdx.x];
      }
    }
  }
}

}  // namespace mpe
