// Register-resident env-step kernel for the stock small worlds (simple_tag 3+1 agents / 2
// landmarks, simple_spread 3 / 3): one env per lane, entity counts are template constants so the
// whole state (E positions, A velocities) lives in VGPRs and every loop is fully unrolled.
// A wavefront = 64 consecutive envs, so each load/store instruction moves one contiguous
// 64*sizeof(T)-byte segment of an SoA row.  simple_tag requires the stock role order: the first
// NADV agents are the adversaries (simple_tag.py:21).
#pragma once
#include "mpe_device.h"

namespace mpe {

constexpr int kSmallThreads = 256;

template <typename T, int KIND, int A, int L, int NADV, bool DO_STEP, bool DO_OBS>
__global__ __launch_bounds__(kSmallThreads) void small_kernel(const KParams<T> p) {
  constexpr int E = A + L;
  const DevDesc<T> &d = p.d;
  const int64_t ld = p.ld;
  const int64_t env0 = (int64_t)blockIdx.x * kSmallThreads + threadIdx.x;
  const bool active = env0 < p.B;
  const int64_t env = active ? env0 : p.B - 1;  // tail lanes shadow the last env and never store

  T px[E], py[E], vx[A], vy[A];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    px[e] = p.pos[(int64_t)(2 * e) * ld + env];
    py[e] = p.pos[(int64_t)(2 * e + 1) * ld + env];
  }
#pragma unroll
  for (int a = 0; a < A; ++a) {
    vx[a] = p.vel[(int64_t)(2 * a) * ld + env];
    vy[a] = p.vel[(int64_t)(2 * a + 1) * ld + env];
  }

  if (DO_STEP) {
    T Fx[A], Fy[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {  // environment.py:188-196, core.py:147-148
      const T *ar = p.act + (int64_t)(5 * a) * ld + env;
      Fx[a] = (ar[1 * ld] - ar[2 * ld]) * d.accel[a];
      Fy[a] = (ar[3 * ld] - ar[4 * ld]) * d.accel[a];
      if (p.noise != nullptr && d.u_noise[a] != T(0)) {
        Fx[a] = Fx[a] + p.noise[(int64_t)(2 * a) * ld + env] * d.u_noise[a];
        Fy[a] = Fy[a] + p.noise[(int64_t)(2 * a + 1) * ld + env] * d.u_noise[a];
      }
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {  // core.py:199-205, b ascending; only a is pushed
      if (!(bit(d.collide, a) && bit(d.movable, a))) continue;
#pragma unroll
      for (int b = a + 1; b < E; ++b) {
        if (!bit(d.collide, b)) continue;
        T fx, fy;
        pair_force<T>(px[a] - px[b], py[a] - py[b], d.size[a] + d.size[b], d.k, d.cforce, fx, fy);
        Fx[a] = fx + Fx[a];
        Fy[a] = fy + Fy[a];
      }
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {  // core.py:211-222
      if (!bit(d.movable, a)) continue;
      integrate_vel<T>(vx[a], vy[a], Fx[a], Fy[a], d.mass[a], d.damp1, d.dt, d.max_speed[a]);
      px[a] = px[a] + vx[a] * d.dt;
      py[a] = py[a] + vy[a] * d.dt;
      if (active) {
        p.vel[(int64_t)(2 * a) * ld + env] = vx[a];
        p.vel[(int64_t)(2 * a + 1) * ld + env] = vy[a];
        p.pos[(int64_t)(2 * a) * ld + env] = px[a];
        p.pos[(int64_t)(2 * a + 1) * ld + env] = py[a];
      }
    }
    if (p.comm != nullptr && active)  // core.py:226-227
      for (int r = 0; r < A * d.dim_c; ++r) p.comm[(int64_t)r * ld + env] = T(0);
  }

  if (DO_OBS) {
    // ---- observations (simple_tag.py:131-147 / simple_spread.py:83-99) ---------------------------
    if (p.obs != nullptr && active) {
#pragma unroll
      for (int i = 0; i < A; ++i) {
        T *o = p.obs + (int64_t)d.obs_row0[i] * ld + env;
        int r = 0;
        o[(r++) * ld] = vx[i];
        o[(r++) * ld] = vy[i];
        o[(r++) * ld] = px[i];
        o[(r++) * ld] = py[i];
#pragma unroll
        for (int l = A; l < E; ++l) {
          o[(r++) * ld] = px[l] - px[i];
          o[(r++) * ld] = py[l] - py[i];
        }
#pragma unroll
        for (int j = 0; j < A; ++j) {
          if (j == i) continue;
          o[(r++) * ld] = px[j] - px[i];
          o[(r++) * ld] = py[j] - py[i];
        }
        if (KIND == MPE_SIMPLE_TAG) {
#pragma unroll
          for (int j = NADV; j < A; ++j) {
            if (j == i) continue;
            o[(r++) * ld] = vx[j];
            o[(r++) * ld] = vy[j];
          }
        } else {
          for (int z = 0; z < d.dim_c * (A - 1); ++z) o[(r++) * ld] = T(0);
        }
      }
    }
    // ---- reward / info ------------------------------------------------------------------------
    const bool want_rew = p.rew != nullptr, want_info = p.info != nullptr, want_stats = p.stats != nullptr;
    if (want_rew || want_info || want_stats) {
      T rew[A];
      int coll[A];
      int extra = 0;
      if (KIND == MPE_SIMPLE_TAG) {
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
          T r = bit(d.collide, g) ? -T(10) * T(n) : T(0);  // simple_tag.py:97-100
          r = r - tag_bound<T>(t_abs<T>(px[g]));          // simple_tag.py:109-111
          r = r - tag_bound<T>(t_abs<T>(py[g]));
          rew[g] = r;
        }
#pragma unroll
        for (int a = 0; a < NADV; ++a) rew[a] = bit(d.collide, a) ? T(10) * T(total) : T(0);
        extra = total;
        if (want_info && active) {
#pragma unroll
          for (int i = 0; i < A; ++i) p.info[(int64_t)i * ld + env] = T(coll[i]);
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
          if (bit(d.collide, i)) {
#pragma unroll
            for (int a = 0; a < A; ++a) {  // includes a == i (simple_spread.py:78-80)
              if (touching<T>(px[a] - px[i], py[a] - py[i], d.size[a] + d.size[i])) {
                r = r - T(1);
                c += 1;
              }
            }
          }
          rew[i] = r;
          coll[i] = c;
          if (want_info && active) {
            T *ir = p.info + (int64_t)(4 * i) * ld + env;
            ir[0] = r; ir[ld] = T(c); ir[2 * ld] = md; ir[3 * ld] = T(occ);
          }
        }
        extra = occ;
      }
      if (want_rew && active) {
#pragma unroll
        for (int i = 0; i < A; ++i) p.rew[(int64_t)i * ld + env] = rew[i];
      }
      // ---- per-workgroup statistics: wavefront reduction, then one row update per block ------------
      if (want_stats) {
        constexpr int W = 2 * A + 2;
        __shared__ double sred[kSmallThreads / 64][W];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        double v[W];
#pragma unroll
        for (int i = 0; i < A; ++i) {
          v[i] = active ? (double)rew[i] : 0.0;
          v[A + i] = active ? (double)coll[i] : 0.0;
        }
        v[2 * A] = active ? (double)extra : 0.0;
        v[2 * A + 1] = active ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < W; ++k) {
          const double s = group_sum(v[k], 64);
          if (lane == 0) sred[wave][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < W) {
          double s = 0.0;
#pragma unroll
          for (int w = 0; w < kSmallThreads / 64; ++w) s += sred[w][threadIdx.x];
          p.stats[(int64_t)blockIdx.x * W + threadIdx.x] += s;
        }
      }
    }
  }
}

}  // namespace mpe
