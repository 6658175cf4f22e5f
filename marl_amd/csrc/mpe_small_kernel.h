// Register-resident env-step kernel for the stock small worlds (simple_tag 3+1 agents / 2
// landmarks, simple_spread 3 / 3): one env per lane, entity counts and the stock role flags are
// template constants, so the whole state (E positions, A velocities) lives in VGPRs, every loop is
// fully unrolled and the kernel is straight-line code:
//
//     issue ALL loads (36 rows) -> one wait -> ~700 VALU ops -> 86 streaming stores
//
// A wavefront = 64 consecutive envs, so each load/store instruction moves one contiguous
// 64*sizeof(T)-byte segment of an SoA row; row bases are wave-uniform (SGPR) and the lane offset
// is a 32-bit VGPR, so address arithmetic is almost free.
//
// "Stock flags" (checked on the host, mpe_hip.hip:small_variant): every agent movable and
// colliding; landmarks collide in simple_tag and do not in simple_spread; simple_tag agents all
// have a max_speed and the first NADV agents are the adversaries (simple_tag.py:16-33);
// simple_spread agents have none (simple_spread.py:15-25).  Sizes, accel, mass, dt, damping and
// the contact constants stay run-time values.  Anything else runs the LDS-tile kernel.
//
// Cache policy: actions / noise are read once (nontemporal loads) and observations / rewards / info
// are written once (nontemporal stores); pos / vel use the default policy so the env state, which
// is re-read every step, can stay in L2 / Infinity Cache.
#pragma once
#include "mpe_device.h"
#include "mpe_launch.h"

namespace mpe {

// Occupancy cap.  Measured on MI355X (simple_tag, 1Mi envs, f32, fused step): 93.9 us at the 5
// waves/SIMD that the kernel's VGPR count allows, 84.8 us capped at 3 and 84.6 us at 2.  Each
// wavefront streams ~126 SoA rows; fewer resident wavefronts mean fewer DRAM pages open at once.
// The hierarchy kernel has the same optimum (L1F2_3Targets: 94.2 us uncapped, 84.5 us at 3).
// (1024-lane workgroups, +3.5 % when uncapped, need 4 waves/SIMD and lose to this: dropped.)
// simple.py's world moves 20 elements per env step, not 126: it wants every wavefront it can get
// (5866 GB/s uncapped, 4803 capped at 3).
// f64 simple_tag is the exception the other way: its ~3 200 f64 instructions per wavefront (~120 us of VALU time at
// 1 Mi envs) weigh about as much as its 1 GB of traffic (~160 us), so it wants every wavefront the registers allow to
// overlap the two: 207.7 us capped at 3, 212.0 at 4, 197.4 at 5, 197.2 uncapped (r2, MI355X).  f64 simple_spread
// (65 536 envs = one wavefront per SIMD) is indifferent: 11.7 us at 3 or 4, 12.2 at 5.
constexpr int small_wave_cap(int kind, int elem_bytes) {
  return kind == MPE_SIMPLE ? 8 : (kind == MPE_SIMPLE_TAG && elem_bytes == 8 ? 5 : 3);
}
#ifdef MPE_ABL_NO_STATS  // ablation: what the per-workgroup statistics epilogue costs
#define MPE_WANT_STATS(x) false
#else
#define MPE_WANT_STATS(x) (x)
#endif
#ifndef MPE_SMALL_ATTR
#define MPE_SMALL_ATTR __attribute__((amdgpu_waves_per_eu(1, small_wave_cap(KIND, (int)sizeof(T)))))
#endif

// NOISE: 0 = no u_noise, 1 = the caller's normals (p.noise), 2 = drawn in the kernel (MPE_FLAG_DEVICE_NOISE, mpe_device.h)
// RES: the launch's Infinity Cache plan (mpe_device.h:residency_plan); 3 = the landmark rows are streamed past the cache
template <typename T, int KIND, int A, int L, int NADV, int THREADS, int NOISE, bool DO_STEP, bool DO_OBS, int RES = 0>
__global__ __launch_bounds__(THREADS) MPE_SMALL_ATTR void small_kernel(const KParams<T> p) {
  constexpr int E = A + L;
  constexpr bool TAG = KIND == MPE_SIMPLE_TAG;
  constexpr bool F1 = KIND >= MPE_SIMPLE;  // simple / simple_adversary / simple_push gathers
  constexpr bool HAS_GOAL = KIND == MPE_SIMPLE_ADVERSARY || KIND == MPE_SIMPLE_PUSH;
  // entities that collide (stock flags): everything in simple_tag, the agents in simple_spread and
  // simple_push, nothing in simple and simple_adversary
  constexpr int ECOL = TAG ? E : ((KIND == MPE_SIMPLE_SPREAD || KIND == MPE_SIMPLE_PUSH) ? A : 0);
  const DevDesc<T> &d = p.d;
  const int64_t ld = p.ld;
  const int64_t blk0 = env_block() * THREADS;  // wave-uniform
  const bool active = blk0 + threadIdx.x < p.B;
  const uint32_t lo = active ? threadIdx.x : kNoLane;  // tail lanes: see Win
  const Win<T> wpos(p.pos + blk0, ld), wvel(p.vel + blk0, ld);
  T px[E], py[E], vx[A], vy[A];
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
  // simple_tag / simple_spread: this block's statistics row is fetched with the state, so that the read-modify-write
  // at the very end of the kernel does not hold the workgroup's slot for one more memory round trip (simple_tag f32
  // 81.7 -> 79.8 us, f64 195.8 -> 192.9).  The same in the other register kernels is neutral (hierarchy, world_comm) or
  // costs 1-3 % (simple_adversary, speaker_listener, simple_crypto: A/B on one box, r2), so they keep the plain `+=`.
  double stats_prev = 0.0;
  if constexpr (DO_OBS && !F1) {
    if (p.stats != nullptr && threadIdx.x < 2 * A + 2) stats_prev = p.stats[(int64_t)blockIdx.x * (2 * A + 2) + threadIdx.x];
  }

  if (DO_STEP) {
    const Win<T> wact(p.act + blk0, ld), wnoise(p.noise + blk0, ld);
    T a1[A], a2[A], a3[A], a4[A], nx[A], ny[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {  // a[0] (the no-op slot) is never read (environment.py:189-190)
      a1[a] = wact.ld_stream(5 * a + 1, lo);
      a2[a] = wact.ld_stream(5 * a + 2, lo);
      a3[a] = wact.ld_stream(5 * a + 3, lo);
      a4[a] = wact.ld_stream(5 * a + 4, lo);
      if (NOISE == 1) {
        nx[a] = wnoise.ld_stream(2 * a, lo);
        ny[a] = wnoise.ld_stream(2 * a + 1, lo);
      }
    }
    if (NOISE == 2) {  // tail lanes draw env B - 1's normals and store nothing
      const int64_t env = active ? blk0 + threadIdx.x : p.B - 1;
      const uint32_t step = noise_step(p);
#pragma unroll
      for (int a = 0; a < A; ++a) {
        nx[a] = ny[a] = T(0);
        if (d.u_noise[a] != T(0)) device_normals<T>(p, env, step, kNoiseWordU | (uint32_t)a, nx[a], ny[a]);
      }
    }
    T Fx[A], Fy[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {  // environment.py:188-196, core.py:147-148
      Fx[a] = (a1[a] - a2[a]) * d.accel[a];
      Fy[a] = (a3[a] - a4[a]) * d.accel[a];
      if (NOISE != 0) {  // u_noise[a] == 0 encodes None: u + 0.0*n would differ only for n = inf/nan
        Fx[a] = d.u_noise[a] != T(0) ? Fx[a] + nx[a] * d.u_noise[a] : Fx[a];
        Fy[a] = d.u_noise[a] != T(0) ? Fy[a] + ny[a] * d.u_noise[a] : Fy[a];
      }
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
      integrate_vel_sel<T, TAG>(vx[a], vy[a], Fx[a], Fy[a], d.mass[a], d.damp1, d.dt, d.max_speed[a]);
      px[a] = px[a] + vx[a] * d.dt;
      py[a] = py[a] + vy[a] * d.dt;
      wvel.st_state(2 * a, lo, vx[a]);
      wvel.st_state(2 * a + 1, lo, vy[a]);
      wpos.st_state(2 * a, lo, px[a]);
      wpos.st_state(2 * a + 1, lo, py[a]);
    }
    if (p.comm != nullptr && rewrite_silent_comm(p)) {  // core.py:226-227
      const Win<T> wcomm(p.comm + blk0, ld);
      for (int r = 0; r < A * d.dim_c; ++r) wcomm.st_state(r, lo, T(0));
    }
  }

  if constexpr (DO_OBS && F1) {
    // ---- simple.py:41-50 / simple_adversary.py:58-140 / simple_push.py:66-104 ------------------------
    int goal = 0;
    T gx = px[A], gy = py[A];
    if (HAS_GOAL) {  // agent.goal_a: per-env landmark index
      goal = p.goal[blk0 + (active ? threadIdx.x : (uint32_t)(p.B - 1 - blk0))];  // tail lanes: last env
      goal = goal < 0 ? 0 : (goal >= L ? L - 1 : goal);
#pragma unroll
      for (int l = 1; l < L; ++l) {
        gx = goal == l ? px[A + l] : gx;
        gy = goal == l ? py[A + l] : gy;
      }
    }
    if (p.obs != nullptr) {
#pragma unroll
      for (int i = 0; i < A; ++i) {
        const Win<T> wobs(p.obs + blk0 + (int64_t)d.obs_row0[i] * ld, ld);  // this agent's rows
        int r = 0;
        if (KIND != MPE_SIMPLE_ADVERSARY) {
          wobs.st_stream(r++, lo, vx[i]);
          wobs.st_stream(r++, lo, vy[i]);
        }
        if (HAS_GOAL && i >= NADV) {
          wobs.st_stream(r++, lo, gx - px[i]);
          wobs.st_stream(r++, lo, gy - py[i]);
          if (KIND == MPE_SIMPLE_PUSH) {  // agent.color (simple_push.py:51-56)
#pragma unroll
            for (int k = 0; k < 3; ++k)
              wobs.st_stream(r++, lo, k == goal + 1 ? T(0.25) + T(0.5) : T(0.25));
          }
        }
#pragma unroll
        for (int l = A; l < E; ++l) {
          wobs.st_stream(r++, lo, px[l] - px[i]);
          wobs.st_stream(r++, lo, py[l] - py[i]);
        }
        if (KIND == MPE_SIMPLE_PUSH && i >= NADV) {  // landmark.color (simple_push.py:43-45)
#pragma unroll
          for (int l = 0; l < L; ++l)
#pragma unroll
            for (int k = 0; k < 3; ++k)
              wobs.st_stream(r++, lo, k == l + 1 ? T(0.1) + T(0.8) : T(0.1));
        }
        if (KIND != MPE_SIMPLE) {
#pragma unroll
          for (int j = 0; j < A; ++j) {
            if (j == i) continue;
            wobs.st_stream(r++, lo, px[j] - px[i]);
            wobs.st_stream(r++, lo, py[j] - py[i]);
          }
        }
      }
    }
    const bool want_rew = p.rew != nullptr, want_info = p.info != nullptr, want_stats = p.stats != nullptr;
    if (want_rew || want_info || want_stats) {
      const Win<T> wrew(p.rew + blk0, ld), winfo(p.info + blk0, ld);
      T m = T(0), adv_rew = T(0);
      if (HAS_GOAL) {
#pragma unroll
        for (int g = NADV; g < A; ++g) {
          const T dist = t_sqrt<T>(sqdist<T>(px[g], py[g], gx, gy));
          m = (g == NADV || dist < m) ? dist : m;
        }
        if (KIND == MPE_SIMPLE_ADVERSARY) {
#pragma unroll
          for (int a = 0; a < NADV; ++a) adv_rew = adv_rew + t_sqrt<T>(sqdist<T>(px[a], py[a], gx, gy));
        }
      }
      T rew[A];
#pragma unroll
      for (int i = 0; i < A; ++i) {
        if (KIND == MPE_SIMPLE) {
          rew[i] = -sqdist<T>(px[i], py[i], px[A], py[A]);
        } else if (KIND == MPE_SIMPLE_ADVERSARY) {
          rew[i] = i < NADV ? -sqdist<T>(px[i], py[i], gx, gy) : -m + adv_rew;
        } else {
          const T dg = t_sqrt<T>(sqdist<T>(px[i], py[i], gx, gy));
          rew[i] = i < NADV ? m - dg : -dg;
        }
        if (want_rew) wrew.st_stream(i, lo, rew[i]);
        if (KIND == MPE_SIMPLE_ADVERSARY && want_info) {  // simple_adversary.py:58-67
          const T dgoal = sqdist<T>(px[i], py[i], gx, gy);
#pragma unroll
          for (int l = 0; l < L; ++l)
            winfo.st_stream((L + 1) * i + l, lo,
                            i < NADV ? (l == 0 ? dgoal : T(0)) : sqdist<T>(px[i], py[i], px[A + l], py[A + l]));
          winfo.st_stream((L + 1) * i + L, lo, i < NADV ? T(0) : dgoal);
        }
      }
      if (want_stats) {
        constexpr int W = 2 * A + 2;
        constexpr int NW = THREADS / 64;
        __shared__ double sred1[NW][A + 1];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < A; ++i) {
          const double s_r = group_sum(active ? (double)rew[i] : 0.0, 64);
          if (lane == 0) sred1[wave][i] = s_r;
        }
        const double n_active = (double)__builtin_popcountll(__ballot(active));  // all lanes vote
        if (lane == 0) sred1[wave][A] = n_active;
        __syncthreads();
        if (threadIdx.x <= A) {
          double s = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) s += sred1[w][threadIdx.x];
          p.stats[(int64_t)blockIdx.x * W + (threadIdx.x < A ? threadIdx.x : 2 * A + 1)] += s;
        }
      }
    }
  }

  if constexpr (DO_OBS && !F1) {
#ifndef MPE_NO_SYNC_OBS
    // the four wavefronts of a workgroup start their store streams together: each row then gets its 1 KiB piece from
    // one workgroup at about the same time (simple_tag f32 79.6 -> 78.8 us, r2 A/B; tail lanes take no early exit)
    __syncthreads();
#endif
    // ---- observations (simple_tag.py:131-147 / simple_spread.py:83-99) ---------------------------
    if (p.obs != nullptr) {
#pragma unroll
      for (int i = 0; i < A; ++i) {
        const Win<T> wobs(p.obs + blk0 + (int64_t)d.obs_row0[i] * ld, ld);  // this agent's rows
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
#pragma unroll
        for (int j = 0; j < A; ++j) {
          if (j == i) continue;
          wobs.st_stream(r++, lo, px[j] - px[i]);
          wobs.st_stream(r++, lo, py[j] - py[i]);
        }
        if (TAG) {
#pragma unroll
          for (int j = NADV; j < A; ++j) {
            if (j == i) continue;
            wobs.st_stream(r++, lo, vx[j]);
            wobs.st_stream(r++, lo, vy[j]);
          }
        } else {
          for (int z = 0; z < d.dim_c * (A - 1); ++z) wobs.st_stream(r++, lo, T(0));
        }
      }
    }
    // ---- reward / info ------------------------------------------------------------------------
    const bool want_rew = p.rew != nullptr, want_info = p.info != nullptr, want_stats = MPE_WANT_STATS(p.stats != nullptr);
    if (want_rew || want_info || want_stats) {
      const Win<T> wrew(p.rew + blk0, ld), winfo(p.info + blk0, ld);
      T rew[A];
      int coll[A];
      int extra = 0;
      if (TAG) {
        int total = 0;
#pragma unroll
        for (int a = 0; a < NADV; ++a) coll[a] = 0;
#pragma unroll
        for (int g = NADV; g < A; ++g) {
          int n = 0;
#pragma unroll
          for (int a = 0; a < NADV; ++a) {
            const int hit = touching<T>(px[g] - px[a], py[g] - py[a], d.size[g] + d.size[a]) ? 1 : 0;
            n += hit;
            coll[a] += hit;
          }
          total += n;
          coll[g] = 0;
          T r = -T(10) * T(n);                                 // simple_tag.py:97-100
          r = r - tag_bound_sel<T>(t_abs<T>(px[g]));          // simple_tag.py:109-111
          r = r - tag_bound_sel<T>(t_abs<T>(py[g]));
          rew[g] = r;
        }
#pragma unroll
        for (int a = 0; a < NADV; ++a) rew[a] = T(10) * T(total);  // simple_tag.py:124-128
        extra = total;
        if (want_info) {
#pragma unroll
          for (int i = 0; i < A; ++i) winfo.st_stream(i, lo, T(coll[i]));
        }
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
#pragma unroll
        for (int i = 0; i < A; ++i) {
          T r = r0;
          int c = 0;
#pragma unroll
          for (int a = 0; a < A; ++a) {  // includes a == i (simple_spread.py:78-80)
            const bool hit = touching<T>(px[a] - px[i], py[a] - py[i], d.size[a] + d.size[i]);
            r = hit ? r - T(1) : r;
            c += hit ? 1 : 0;
          }
          rew[i] = r;
          coll[i] = c;
          if (want_info) {
            winfo.st_stream(4 * i, lo, r);
            winfo.st_stream(4 * i + 1, lo, T(c));
            winfo.st_stream(4 * i + 2, lo, md);
            winfo.st_stream(4 * i + 3, lo, T(occ));
          }
        }
        extra = occ;
      }
      if (want_rew) {
#pragma unroll
        for (int i = 0; i < A; ++i) wrew.st_stream(i, lo, rew[i]);
      }
      // ---- per-workgroup statistics: wavefront reductions, then one row update per block ---------
      // Small integer counters (<= 3 per lane, so <= 192 per wavefront) are packed four to a word
      // and summed with one integer butterfly; only the float rewards need a double butterfly.
      if (want_stats) {
        constexpr int W = 2 * A + 2;
        constexpr int NW = THREADS / 64;
        static_assert(A <= 4 && (!TAG || (A - NADV == 1 && NADV <= 3)), "counter packing assumes the stock worlds");
        __shared__ double sred[NW][W];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < (TAG ? NADV : A); ++i) packed |= (uint32_t)coll[i] << (8 * i);
        if (!TAG) packed |= (uint32_t)extra << 24;
        packed = wave_sum_u32(active ? packed : 0u);
        const double n_active = (double)__builtin_popcountll(__ballot(active));
        double rsum[A];
        if (TAG) {
          rsum[NADV] = group_sum(active ? (double)rew[NADV] : 0.0, 64);
        } else {
#pragma unroll
          for (int i = 0; i < A; ++i) rsum[i] = group_sum(active ? (double)rew[i] : 0.0, 64);
        }
        if (lane == 0) {
          double csum[A], tot = 0.0;
#pragma unroll
          for (int i = 0; i < A; ++i) {
            csum[i] = (TAG && i >= NADV) ? 0.0 : (double)((packed >> (8 * i)) & 0xFFu);
            tot += csum[i];
          }
#pragma unroll
          for (int i = 0; i < A; ++i) {
            // simple_tag adversaries all get 10 * total (exact in double)
            sred[wave][i] = (TAG && i < NADV) ? 10.0 * tot : rsum[i];
            sred[wave][A + i] = csum[i];
          }
          sred[wave][2 * A] = TAG ? tot : (double)(packed >> 24);
          sred[wave][2 * A + 1] = n_active;
        }
        __syncthreads();
        if (threadIdx.x < W) {
          double s = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) s += sred[w][threadIdx.x];
          p.stats[(int64_t)blockIdx.x * W + threadIdx.x] = stats_prev + s;
        }
      }
    }
  }
}

}  // namespace mpe
