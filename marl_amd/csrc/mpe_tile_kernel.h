// Generic (any A <= 64, E <= 128) env-step kernel: one workgroup = one tile of TILE env instances,
// entity state staged in LDS.
//
//   workgroup = 256 threads = NCH chunks x TILE lanes; lane <-> env (so every global access is a
//   coalesced 256-byte row segment of the SoA arrays), chunk <-> subset of agents (a % NCH).
//   TILE = 64 for f32 (chunk = one wavefront), 32 for f64 (chunk = half a wavefront).
//
//   LDS:  sPos[2E][TILE]  (sLf[2A][TILE] leader_force, command hierarchies only)  sVel[max(2A, aux)][TILE]
//         (the reward scratch sAux aliases sVel once the observations, the last reader of sVel, are
//         out: E=96/A=64 is exactly 80 KiB = 2 blocks/CU)
//
//   phase 0  cooperative load of pos rows -> sPos
//   phase 1  per (env, agent a): decode action, add u_noise, sum the one-sided contact forces over
//            b = a+1..E-1 in ascending order (reference order, core.py:199-205), integrate the
//            velocity; new vel -> sVel + HBM          (old positions still in sPos)
//   phase 2  pos_a += vel_a*dt -> sPos + HBM
//   phase 3  observation gather from sPos, sVel; then reward / info from sPos (+ sAux); optional
//            per-workgroup statistics via wavefront reductions.
//
// The O(E^2) pair sum is kept sequential in b inside one lane: that is the reference's summation
// order, it needs no cross-lane traffic at all, and the lanes are already full with envs.
#pragma once
#include "mpe_device.h"
#include "mpe_launch.h"

#ifdef MPE_ABL_NO_STATS  // ablation: what the statistics rows cost
#define MPE_TILE_WANT_STATS(x) false
#else
#define MPE_TILE_WANT_STATS(x) (x)
#endif

namespace mpe {

// 16-byte vector of T: 4 floats / 2 doubles = the envs one lane covers in the vectorised observation phase
template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec16<double> { typedef double type __attribute__((ext_vector_type(2))); };

// One 16-byte streaming store (the lane's VEC consecutive envs of one row), or, for a lane of the last tile that
// reaches past B / a row stride or base pointer that is not 16-byte aligned, its valid elements one by one.
template <typename T>
__device__ __forceinline__ void st_stream_vec(T *ptr, typename Vec16<T>::type v, int nvalid, bool full) {
  constexpr int VEC = 16 / sizeof(T);
  if (full) {
#ifdef MPE_NO_NT_STORE
    *reinterpret_cast<typename Vec16<T>::type *>(ptr) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<typename Vec16<T>::type *>(ptr));
#endif
  } else {
#pragma unroll
    for (int c = 0; c < VEC; ++c)
      if (c < nvalid) st_stream(ptr + c, v[c]);
  }
}

// One section of an agent's observation: rows kk = 0 .. nrows-1 copy the LDS rows src + kk (+ 2 from kk >= skip on:
// the observer's own two rows are left out of "other agents"), minus the observer's position when REL.  Four rows per
// trip: lane (r, q) handles row b0 + r, envs VEC q .. VEC q + VEC - 1, so one store instruction writes four 256-byte
// row segments and one ds_read_b128 per lane fetches them (1 KiB of LDS per wave-instruction, conflict-free).
// Sections start on even rows and b0 is a multiple of 4, so row parity (x / y) is r & 1 throughout.
// 16 bytes (the lane's VEC consecutive envs) of one source row: LDS rows and, with FULL, global rows are aligned.
template <typename T, bool FULL>
__device__ __forceinline__ typename Vec16<T>::type ld_vec(const T *ptr, int nvalid) {
  typedef typename Vec16<T>::type VT;
  constexpr int VEC = 16 / sizeof(T);
  if (FULL) return *reinterpret_cast<const VT *>(ptr);
  VT v;
#pragma unroll
  for (int c = 0; c < VEC; ++c) v[c] = c < nvalid ? ptr[c] : T(0);
  return v;
}

// The landmark section of an agent's observation with the landmark rows read from GLOBAL memory (LGLOB tiles keep only
// the agents in LDS): row kk of the section is position row 2A + kk of the SoA array; `lane_src` points at (row 2A + r,
// this lane's first env).  Same lane mapping as obs_section: four rows per store instruction.
template <typename T, bool FULL>
__device__ __forceinline__ void obs_section_global(T *lane_out, int64_t ld, const T *lane_src, int nrows, int r,
                                                   typename Vec16<T>::type pi, int nvalid) {
  typedef typename Vec16<T>::type VT;
  int b0 = 0;
  for (; b0 + 4 < nrows; b0 += 8) {  // two blocks per trip: both loads are in flight before the first subtraction
    const bool ok1 = b0 + 4 + r < nrows;
    VT v0 = ld_vec<T, FULL>(lane_src + (int64_t)b0 * ld, nvalid);
    VT v1 = ld_vec<T, FULL>(lane_src + (int64_t)(ok1 ? b0 + 4 : 0) * ld, nvalid);
    v0 = v0 - pi;
    v1 = v1 - pi;
    st_stream_vec<T>(lane_out + (int64_t)b0 * ld, v0, nvalid, FULL);
    if (ok1) st_stream_vec<T>(lane_out + (int64_t)(b0 + 4) * ld, v1, nvalid, FULL);
  }
  if (b0 < nrows) {
    const bool ok = b0 + r < nrows;
    VT v = ld_vec<T, FULL>(lane_src + (int64_t)(ok ? b0 : 0) * ld, nvalid);
    v = v - pi;
    if (ok) st_stream_vec<T>(lane_out + (int64_t)b0 * ld, v, nvalid, FULL);
  }
}

template <typename T, int TILE, bool REL, bool FULL>
__device__ __forceinline__ void obs_section(T *lane_out, int64_t ld, const T *lane_src, int nrows, int skip, int r,
                                            typename Vec16<T>::type pi, int nvalid) {
  typedef typename Vec16<T>::type VT;
  // two blocks per trip: both LDS reads are in flight before the first subtraction waits for its data
  int b0 = 0;
  for (; b0 + 4 < nrows; b0 += 8) {
    const int k0 = b0 + r, k1 = b0 + 4 + r;
    const bool ok1 = k1 < nrows;
    const int row0 = k0 >= skip ? b0 + 2 : b0;                   // k0 < nrows: only the last block of a section is partial
    const int row1 = ok1 ? (k1 >= skip ? b0 + 6 : b0 + 4) : 0;   // rows beyond the section: read row 0, store nothing
    VT v0 = *reinterpret_cast<const VT *>(lane_src + row0 * TILE);
    VT v1 = *reinterpret_cast<const VT *>(lane_src + row1 * TILE);
    if (REL) { v0 = v0 - pi; v1 = v1 - pi; }
    st_stream_vec<T>(lane_out + (int64_t)b0 * ld, v0, nvalid, FULL);
    if (ok1) st_stream_vec<T>(lane_out + (int64_t)(b0 + 4) * ld, v1, nvalid, FULL);
  }
  if (b0 < nrows) {
    const int kk = b0 + r;
    const bool ok = kk < nrows;
    const int row = ok ? (kk >= skip ? b0 + 2 : b0) : 0;
    VT v = *reinterpret_cast<const VT *>(lane_src + row * TILE);
    if (REL) v = v - pi;
    if (ok) st_stream_vec<T>(lane_out + (int64_t)b0 * ld, v, nvalid, FULL);
  }
}

// Observation rows of simple_tag (simple_tag.py:131-147) / simple_spread (simple_spread.py:83-99) for one tile, all
// agents: [vel_i, pos_i, landmarks - pos_i, other agents - pos_i, (tag) velocities of the other good agents /
// (spread) dim_c (A-1) zeros].  Wavefront w takes agents w, w + NW, ...; 94 % of the scaled world's HBM traffic is
// issued here, as 16-byte stores (global_store_dwordx4), 4 rows per instruction.
//
// LGLOB: the tile keeps only the agents' positions and the good agents' velocities in LDS (sVel row pair g = the g-th good
// agent); landmark rows come from global memory, and every agent's own velocity rows were already stored by phase 1.
template <typename T, int TILE, int THREADS, bool TAG, bool FULL, bool LGLOB>
__device__ __forceinline__ void obs_rows_vec_impl(const KParams<T> &p, const T *sPos, const T *sVel, int64_t tile) {
  constexpr int VEC = 16 / sizeof(T);
  static_assert(TILE == 16 * VEC, "a row segment of the tile is 16 lanes x 16 bytes");
  typedef typename Vec16<T>::type VT;
  constexpr int NW = THREADS / 64;
  const DevDesc<T> &d = p.d;
  const int A = d.A, L = d.L;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l64 = threadIdx.x & 63, q = l64 & 15, r = l64 >> 4;
  const int64_t ld = p.ld;
  const int64_t env0 = tile * TILE + VEC * q;
  const int64_t left = p.B - env0;
  const int nvalid = left >= VEC ? VEC : (left > 0 ? (int)left : 0);
  const int lcol = VEC * q;
  for (int i = wv; i < A; i += NW) {
    T *o = p.obs + (int64_t)(d.obs_row0[i] + r) * ld + env0;  // this lane's row of block 0
    const VT pi = *reinterpret_cast<const VT *>(sPos + (2 * i + (r & 1)) * TILE + lcol);
    if constexpr (LGLOB) {  // [vel_i (stored by phase 1), pos_i]
      if (r >= 2) st_stream_vec<T>(o, pi, nvalid, FULL);
      o += 4 * ld;
      obs_section_global<T, FULL>(o, ld, p.pos + (int64_t)(2 * A + r) * ld + env0, 2 * L, r, pi, nvalid);
    } else {  // [vel_i, pos_i]
      const T *s0 = (r < 2 ? sVel : sPos) + (2 * i + (r & 1)) * TILE + lcol;
      st_stream_vec<T>(o, *reinterpret_cast<const VT *>(s0), nvalid, FULL);
      o += 4 * ld;
      obs_section<T, TILE, true, FULL>(o, ld, sPos + (2 * A + r) * TILE + lcol, 2 * L, 0x7fffffff, r, pi, nvalid);
    }
    o += (int64_t)(2 * L) * ld;
    obs_section<T, TILE, true, FULL>(o, ld, sPos + r * TILE + lcol, 2 * (A - 1), 2 * i, r, pi, nvalid);
    o += (int64_t)(2 * (A - 1)) * ld;
    if (TAG) {
      // good agents other than i, in agent order: two per trip (scalar lookups in the descriptor's good list)
      const bool i_good = !bit(d.adversary, i);
      // position of i in the good list = number of good agents below i (no entry is skipped for an adversary)
      int rank = d.n_good;
      if (i_good) {
        const uint32_t lo = i < 32 ? ~d.adversary[0] & ((1u << (i & 31)) - 1u) : ~d.adversary[0];
        const uint32_t hi = i < 32 ? 0u : ~d.adversary[1] & ((1u << (i & 31)) - 1u);
        rank = __builtin_popcount(lo) + __builtin_popcount(hi);
      }
      const int nrows = 2 * (d.n_good - (i_good ? 1 : 0));
      for (int b0 = 0; b0 < nrows; b0 += 4) {
        const int ga = b0 >> 1, gb = ga + 1;
        // LGLOB: the velocity rows in LDS are indexed by position in the good list, not by agent
        const int sa = ga + (ga >= rank ? 1 : 0), sb = gb + (gb >= rank ? 1 : 0);
        const int ja = LGLOB ? sa : desc_u8(d.good_list, sa);
        const int jb = 2 * gb < nrows ? (LGLOB ? sb : desc_u8(d.good_list, sb)) : ja;
        const bool ok = b0 + r < nrows;
        const VT v = *reinterpret_cast<const VT *>(sVel + (2 * (r >> 1 ? jb : ja) + (r & 1)) * TILE + lcol);
        if (ok) st_stream_vec<T>(o + (int64_t)b0 * ld, v, nvalid, FULL);
      }
    } else {
      const int nrows = d.dim_c * (A - 1);  // other.state.c of silent agents (simple_spread.py:95)
      VT z;
#pragma unroll
      for (int c = 0; c < VEC; ++c) z[c] = T(0);
      for (int b0 = 0; b0 < nrows; b0 += 4)
        if (b0 + r < nrows) st_stream_vec<T>(o + (int64_t)b0 * ld, z, nvalid, FULL);
    }
  }
}

template <typename T, int TILE, int THREADS, bool TAG, bool LGLOB>
__device__ __forceinline__ void obs_rows_vec(const KParams<T> &p, const T *sPos, const T *sVel, int64_t tile) {
  constexpr int VEC = 16 / sizeof(T);
  // whole 16-byte accesses everywhere (workgroup-uniform test): every env of the tile exists, rows and bases are aligned
  const bool full = (tile + 1) * TILE <= p.B && (p.ld % VEC) == 0 && (reinterpret_cast<uintptr_t>(p.obs) % 16) == 0 &&
                    (!LGLOB || (reinterpret_cast<uintptr_t>(p.pos) % 16) == 0);
  if (full) obs_rows_vec_impl<T, TILE, THREADS, TAG, true, LGLOB>(p, sPos, sVel, tile);
  else obs_rows_vec_impl<T, TILE, THREADS, TAG, false, LGLOB>(p, sPos, sVel, tile);
}

// HIER: the world may carry ragged actions / speaking agents (kinds MPE_HIER_PURSUIT, the only one with
// force_id / forced_id, MPE_SPEAKER_LISTENER and MPE_SIMPLE_REFERENCE; mpe_hip.hip:mpe_desc_check).  A template flag so that the
// stock kinds do not pay for its branches in the per-agent loops (scaled simple_tag: 3.80 -> 3.50 ms).
//
// KIND: MPE_SIMPLE_TAG / MPE_SIMPLE_SPREAD compile the scenario in (the north-star worlds at any size: the other
// kinds' branches disappear, and with them most of the live scalars: the all-kinds build of the scaled simple_tag
// step needed 106 SGPRs + 134 spilled to VGPR lanes); -1 = the kind is read from the descriptor at run time.
//
// LGLOB (the fused step of the specialised simple_tag kind only): the landmarks stay in global memory.
// They never move and 94 % of this kernel's time is the observation store stream, whose rate depends on how many
// workgroups per CU are at a different point of their tile (load -> forces -> stores); LDS is what limits that number.
//   LDS:  sPos[2A][TILE] (agents only)  sVel[2 n_good][TILE] (the good agents' velocities, by position in the good
//         list)  -- E = 96 / A = 64 / 16 good: 40 KiB instead of 80 = 4 workgroups per CU
//   phase 1 reads landmark rows with coalesced global loads (L2 hits after the first chunk), stores the new velocity,
//           the new POSITION and the agent's own two velocity rows of its observation straight from registers;
//   phase 2 reloads the new positions of the chunk's agents into LDS (same thread, same addresses) once every pair
//           loop is done with the old ones;
//   phase 3 reads the landmark section of every observation from global memory (16 bytes per lane, four rows per
//           instruction as before); the reward scratch is 4 adversaries per int (counts <= n_good <= 64).
// Same arithmetic in the same order as the LDS form: bit-identical (tests/test_gpu_parity.py).
// Register budget.  The LGLOB form is compiled for 6 wavefronts per SIMD (80 VGPRs instead of the 111 the compiler takes
// when left alone): 24 wavefronts per CU = three 512-thread workgroups of 40 KiB.  Measured on the scaled world (r3,
// profiles/r3_tile_lglob.txt): LDS form 3.33 ms; LGLOB at 4 x 256 threads 3.41, at 2 x 512 (111 VGPRs) 3.18 vs 3.21 on
// another box, at 3 x 512 (this) 3.27 (2.98 on the other box), forced to 7 / 8 wavefronts (spills) 3.30 / 3.31.
constexpr int tile_min_waves(bool lglob) { return lglob ? 6 : 1; }
constexpr int tile_max_waves(bool lglob) { return lglob ? 6 : 8; }
#ifndef MPE_TILE_ATTR
#define MPE_TILE_ATTR __attribute__((amdgpu_waves_per_eu(tile_min_waves(LGLOB), tile_max_waves(LGLOB))))
#endif
template <typename T, int TILE, int THREADS, bool DO_STEP, bool DO_OBS, bool HIER, int KIND, bool LGLOB = false>
__global__ __launch_bounds__(THREADS) MPE_TILE_ATTR void tile_kernel(const KParams<T> p) {
  constexpr int NCH = THREADS / TILE;
  static_assert(!HIER || KIND < 0, "the ragged kinds are not specialised");
  static_assert(!LGLOB || (DO_STEP && DO_OBS && KIND == MPE_SIMPLE_TAG),
                "landmarks in global memory: the fused simple_tag step only");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const DevDesc<T> &d = p.d;
  const int kind = KIND >= 0 ? KIND : d.kind;
  const int A = d.A, L = d.L, E = d.E;
  T *sPos = reinterpret_cast<T *>(smem_raw);
  T *sLf = sPos + 2 * (LGLOB ? A : E) * TILE;
  const bool has_lf = HIER && d.has_lforce;
  T *sVel = sLf + (has_lf ? 2 * A * TILE : 0);
  T *sAux = sVel;  // valid only after the observation phase

  const int lane = threadIdx.x % TILE;
  // f32: a chunk is a whole wavefront, so the chunk index -- and with it every agent index, loop
  // bound and descriptor lookup below -- is wave-uniform.  Saying so (readfirstlane) turns the
  // per-agent loops into scalar loops with scalar descriptor loads; left as a per-lane value the
  // compiler ran them as exec-masked vector loops and fetched d.size[b] with a vector load + full
  // wait inside the pair loop.  f64 (two chunks per wavefront) keeps the per-lane form.
  const int chunk = TILE == 64 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x / TILE)) : (int)(threadIdx.x / TILE);
  const int64_t ld = p.ld;
  const int64_t ntiles = (p.B + TILE - 1) / TILE;

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t env = tile * TILE + lane;
    const bool active = env < p.B;
    const int64_t ge = active ? env : (p.B - 1);  // inactive lanes shadow the last env, never store

    // ---- phase 0 ------------------------------------------------------------------------------
    for (int r = chunk; r < 2 * (LGLOB ? A : E); r += NCH) sPos[r * TILE + lane] = ld_state(p.pos + (int64_t)r * ld + ge);
    // Inputs of one agent of one env: old velocity, the four action entries that matter, u_noise normals.  Fetched one
    // agent ahead of the pair loop (and the first one here, beside the position loads), so that the HBM latency of
    // these read-once rows -- several microseconds while the chip streams observations -- hides behind compute.
    struct AgentIn { T vx, vy, a1, a2, a3, a4, nx, ny; };
    auto fetch = [&](int a, AgentIn &in) {
      in.vx = ld_state(p.vel + (int64_t)(2 * a) * ld + ge);
      in.vy = ld_state(p.vel + (int64_t)(2 * a + 1) * ld + ge);
      in.a1 = in.a2 = in.a3 = in.a4 = in.nx = in.ny = T(0);
      if (bit(d.movable, a)) {
        const T *ar = p.act + (int64_t)(HIER ? d.act_row0[a] : 5 * a) * ld + ge;  // environment.py:188-190
        in.a1 = ld_stream(ar + ld);
        in.a2 = ld_stream(ar + 2 * ld);
        in.a3 = ld_stream(ar + 3 * ld);
        in.a4 = ld_stream(ar + 4 * ld);
        if (p.noise != nullptr && d.u_noise[a] != T(0)) {
          in.nx = ld_stream(p.noise + (int64_t)(2 * a) * ld + ge);
          in.ny = ld_stream(p.noise + (int64_t)(2 * a + 1) * ld + ge);
        }
      }
    };
    // u_noise normals drawn in the kernel (MPE_FLAG_DEVICE_NOISE, no tensor from the caller): at the point of use
    const bool draw_u = DO_STEP && p.noise == nullptr && device_noise(p);
    const uint32_t nstep = draw_u ? noise_step(p) : 0u;
    AgentIn nxt;
    if (DO_STEP && chunk < A) fetch(chunk, nxt);
    if (!DO_STEP) {
      for (int r = chunk; r < 2 * A; r += NCH) sVel[r * TILE + lane] = ld_state(p.vel + (int64_t)r * ld + ge);
      if (has_lf)
        for (int r = chunk; r < 2 * A; r += NCH) sLf[r * TILE + lane] = p.lforce[(int64_t)r * ld + ge];
    }
    __syncthreads();

    if (DO_STEP) {
      // ---- phase 1: forces + velocity ---------------------------------------------------------
      for (int a = chunk; a < A; a += NCH) {
        const AgentIn cur = nxt;
        if (a + NCH < A) fetch(a + NCH, nxt);
        T vx = cur.vx, vy = cur.vy;
        if (bit(d.movable, a)) {
          // environment.py:188-196
          T Fx = (cur.a1 - cur.a2) * d.accel[a];
          T Fy = (cur.a3 - cur.a4) * d.accel[a];
          // core.py:147-148
          if (p.noise != nullptr && d.u_noise[a] != T(0)) {
            Fx = Fx + cur.nx * d.u_noise[a];
            Fy = Fy + cur.ny * d.u_noise[a];
          } else if (draw_u && d.u_noise[a] != T(0)) {
            T nx, ny;
            device_normals<T>(p, ge, nstep, kNoiseWordU | (uint32_t)a, nx, ny);
            Fx = Fx + nx * d.u_noise[a];
            Fy = Fy + ny * d.u_noise[a];
          }
          // core.py:153-194
          if (has_lf) {
            T lfx, lfy;
            leader_force<T>(p, a, ge, [&](int e, int c) { return sPos[(2 * e + c) * TILE + lane]; }, Fx, Fy, lfx, lfy);
            sLf[(2 * a) * TILE + lane] = lfx;
            sLf[(2 * a + 1) * TILE + lane] = lfy;
            if (active) {
              p.lforce[(int64_t)(2 * a) * ld + env] = lfx;
              p.lforce[(int64_t)(2 * a + 1) * ld + env] = lfy;
            }
          }
          // core.py:199-205
          if (bit(d.collide, a)) {
            const T pax = sPos[(2 * a) * TILE + lane], pay = sPos[(2 * a + 1) * TILE + lane];
            const T sa = d.size[a];
            int b = a + 1;
            // positions of entity b: LDS; with LGLOB the landmarks (b >= A) come from global memory
            auto bx_lds = [&](int e) { return sPos[(2 * e) * TILE + lane]; };
            auto by_lds = [&](int e) { return sPos[(2 * e + 1) * TILE + lane]; };
            auto bx_glb = [&](int e) { return ld_state(p.pos + (int64_t)(2 * e) * ld + ge); };
            auto by_glb = [&](int e) { return ld_state(p.pos + (int64_t)(2 * e + 1) * ld + ge); };
            // regular worlds (all of [0, ncol) collide): U independent pair evaluations per trip for
            // instruction-level parallelism; the additions stay in ascending-b order.
            auto run = [&](int b0, int b1, auto bx, auto by) {
              constexpr int U = sizeof(T) == 4 ? 4 : 1;  // f64 pair code is long: unrolling it spills
              int q = b0;
              for (; q + U - 1 < b1; q += U) {
                T fx[U], fy[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                  pair_force<T>(pax - bx(q + u), pay - by(q + u), sa + d.size[q + u], d, fx[u], fy[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  Fx = fx[u] + Fx;
                  Fy = fy[u] + Fy;
                }
              }
              for (; q < b1; ++q) {
                T fx, fy;
                pair_force<T>(pax - bx(q), pay - by(q), sa + d.size[q], d, fx, fy);
                Fx = fx + Fx;
                Fy = fy + Fy;
              }
            };
            if (d.ncol_prefix >= 0) {
#ifdef MPE_ABL_NO_PAIRS
              const int bend = b + 1 < d.ncol_prefix ? b + 1 : d.ncol_prefix;
#else
              const int bend = d.ncol_prefix;
#endif
              if constexpr (LGLOB) {
                const int mid = bend < A ? bend : A;
                run(b, mid, bx_lds, by_lds);
                run(mid > b ? mid : b, bend, bx_glb, by_glb);
              } else {
                run(b, bend, bx_lds, by_lds);
              }
            } else {
              for (; b < E; ++b) {
                if (!bit(d.collide, b)) continue;
                T fx, fy;
                if (LGLOB && b >= A) pair_force<T>(pax - bx_glb(b), pay - by_glb(b), sa + d.size[b], d, fx, fy);
                else pair_force<T>(pax - bx_lds(b), pay - by_lds(b), sa + d.size[b], d, fx, fy);
                Fx = fx + Fx;
                Fy = fy + Fy;
              }
            }
          }
          integrate_vel<T>(vx, vy, Fx, Fy, d.mass[a], d.damp1, d.dt, d.max_speed[a]);
          if (active) {
            st_state(p.vel + (int64_t)(2 * a) * ld + env, vx);
            st_state(p.vel + (int64_t)(2 * a + 1) * ld + env, vy);
            if constexpr (LGLOB) {  // core.py:222 now, from registers; LDS keeps the old position until phase 2
              st_state(p.pos + (int64_t)(2 * a) * ld + env, sPos[(2 * a) * TILE + lane] + vx * d.dt);
              st_state(p.pos + (int64_t)(2 * a + 1) * ld + env, sPos[(2 * a + 1) * TILE + lane] + vy * d.dt);
            }
          }
        }
        if constexpr (LGLOB) {
          if (p.obs != nullptr && active) {  // the agent's own velocity: rows 0, 1 of its observation
            st_stream(p.obs + (int64_t)d.obs_row0[a] * ld + env, vx);
            st_stream(p.obs + (int64_t)(d.obs_row0[a] + 1) * ld + env, vy);
          }
          if (KIND == MPE_SIMPLE_TAG && !bit(d.adversary, a)) {  // slot = number of good agents below a
            const uint32_t lo = a < 32 ? ~d.adversary[0] & ((1u << (a & 31)) - 1u) : ~d.adversary[0];
            const uint32_t hi = a < 32 ? 0u : ~d.adversary[1] & ((1u << (a & 31)) - 1u);
            const int g = __builtin_popcount(lo) + __builtin_popcount(hi);
            sVel[(2 * g) * TILE + lane] = vx;
            sVel[(2 * g + 1) * TILE + lane] = vy;
          }
        } else {
          sVel[(2 * a) * TILE + lane] = vx;
          sVel[(2 * a + 1) * TILE + lane] = vy;
        }
      }
      // movable landmarks (no scenario file has one; core.py:211-222 integrates every movable entity): damped, pushed by
      // the colliding entities BEHIND them (one-sided response: pairs (e, b > e), i.e. later landmarks only), clamped,
      // moved.  Their velocity rows follow the agents' in LDS.
      if (!LGLOB && d.lm_movable) {
        for (int e = A + chunk; e < E; e += NCH) {
          if (!bit(d.movable, e)) continue;
          T vx = ld_state(p.lvel + (int64_t)(2 * (e - A)) * ld + ge), vy = ld_state(p.lvel + (int64_t)(2 * (e - A) + 1) * ld + ge);
          T Fx = T(0), Fy = T(0);
          if (bit(d.collide, e)) {
            const T pex = sPos[(2 * e) * TILE + lane], pey = sPos[(2 * e + 1) * TILE + lane];
            for (int b = e + 1; b < E; ++b) {
              if (!bit(d.collide, b)) continue;
              T fx, fy;
              pair_force<T>(pex - sPos[(2 * b) * TILE + lane], pey - sPos[(2 * b + 1) * TILE + lane], d.size[e] + d.size[b],
                            d, fx, fy);
              Fx = fx + Fx;
              Fy = fy + Fy;
            }
          }
          integrate_vel<T>(vx, vy, Fx, Fy, d.mass[e], d.damp1, d.dt, d.max_speed[e]);
          if (active) {
            st_state(p.lvel + (int64_t)(2 * (e - A)) * ld + env, vx);
            st_state(p.lvel + (int64_t)(2 * (e - A) + 1) * ld + env, vy);
          }
          sVel[(2 * e) * TILE + lane] = vx;
          sVel[(2 * e + 1) * TILE + lane] = vy;
        }
      }
      __syncthreads();
      // ---- phase 2: positions (core.py:222), comm (core.py:226-227) -----------------------------
      if (!LGLOB && d.lm_movable) {
        for (int e = A + chunk; e < E; e += NCH) {
          if (!bit(d.movable, e)) continue;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int r = 2 * e + c;
            const T np_ = sPos[r * TILE + lane] + sVel[r * TILE + lane] * d.dt;
            sPos[r * TILE + lane] = np_;
            if (active) st_state(p.pos + (int64_t)r * ld + env, np_);
          }
        }
      }
      for (int a = chunk; a < A; a += NCH) {
        if (bit(d.movable, a)) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int r = 2 * a + c;
            if constexpr (LGLOB) {  // what this very thread stored in phase 1 (inactive lanes shadow env B - 1, store nothing)
              sPos[r * TILE + lane] = ld_state(p.pos + (int64_t)r * ld + ge);
            } else {
              const T np_ = sPos[r * TILE + lane] + sVel[r * TILE + lane] * d.dt;
              sPos[r * TILE + lane] = np_;
              if (active) st_state(p.pos + (int64_t)r * ld + env, np_);
            }
          }
        }
        if (p.comm != nullptr && active) {  // core.py:226-230
          const bool mute = !HIER || bit(d.silent, a);
          if (mute && !rewrite_silent_comm(p)) continue;
          for (int c = 0; c < d.dim_c; ++c)
            p.comm[(int64_t)(a * d.dim_c + c) * ld + env] = mute ? T(0) : state_c<T, DO_STEP>(p, a, c, env);
        }
      }
      __syncthreads();
    }

    if (DO_OBS) {
      const bool want_rew = p.rew != nullptr, want_info = p.info != nullptr,
                 want_stats = MPE_TILE_WANT_STATS(p.stats != nullptr);
      double *srow = want_stats ? p.stats + (int64_t)blockIdx.x * (2 * A + 2) : nullptr;
      // goal landmark of this env (agent.goal_a), scenarios with a goal only
      const bool has_goal = kind_has_goal(kind);
      int goal = 0;
      T gx = T(0), gy = T(0);
      if (has_goal) {
        goal = p.goal[ge];
        goal = goal < 0 ? 0 : (goal >= L ? L - 1 : goal);
        gx = sPos[(2 * (A + goal)) * TILE + lane];
        gy = sPos[(2 * (A + goal) + 1) * TILE + lane];
      }
      // ---- phase 3a: observations (simple_tag.py:131-147 / simple_spread.py:83-99) -----------------
      constexpr bool VEC_OBS = KIND == MPE_SIMPLE_TAG || KIND == MPE_SIMPLE_SPREAD;
#ifdef MPE_ABL_NO_OBS
      if (p.obs != nullptr && active && p.B < 0) {
#else
      if (VEC_OBS && p.obs != nullptr) {
        obs_rows_vec<T, TILE, THREADS, KIND == MPE_SIMPLE_TAG, LGLOB>(p, sPos, sVel, tile);
      } else if (p.obs != nullptr && active) {
#endif
        // simple_world_comm: which agents stand in forest 0 / forest 1 (is_collision, :240-245,258-259)
        uint64_t in_f0 = 0, in_f1 = 0;
        if (HIER && kind == MPE_WORLD_COMM) {
          const int f0 = E - 2;
          const T f0x = sPos[(2 * f0) * TILE + lane], f0y = sPos[(2 * f0 + 1) * TILE + lane];
          const T f1x = sPos[(2 * f0 + 2) * TILE + lane], f1y = sPos[(2 * f0 + 3) * TILE + lane];
          for (int a = 0; a < A; ++a) {
            const T ax = sPos[(2 * a) * TILE + lane], ay = sPos[(2 * a + 1) * TILE + lane];
            in_f0 |= (uint64_t)touching<T>(ax - f0x, ay - f0y, d.size[a] + d.size[f0]) << a;
            in_f1 |= (uint64_t)touching<T>(ax - f1x, ay - f1y, d.size[a] + d.size[f0 + 1]) << a;
          }
        }
        for (int i = chunk; i < A; i += NCH) {
          T *o = p.obs + (int64_t)d.obs_row0[i] * ld + env;
          const T pix = sPos[(2 * i) * TILE + lane], piy = sPos[(2 * i + 1) * TILE + lane];
          if (HIER && kind == MPE_WORLD_COMM) {
            // simple_world_comm.py:230-313
            st_stream(o, sVel[(2 * i) * TILE + lane]);
            st_stream(o + ld, sVel[(2 * i + 1) * TILE + lane]);
            st_stream(o + 2 * ld, pix);
            st_stream(o + 3 * ld, piy);
            o += 4 * ld;
            for (int l = A; l < E; ++l) {
              st_stream(o, sPos[(2 * l) * TILE + lane] - pix);
              st_stream(o + ld, sPos[(2 * l + 1) * TILE + lane] - piy);
              o += 2 * ld;
            }
            const bool i0 = (in_f0 >> i) & 1, i1 = (in_f1 >> i) & 1, lead = bit(d.leader, i);
            uint64_t vis = 0;  // same forest, or nobody in any forest, or the leader looks (:261)
            for (int j = 0; j < A; ++j) {
              if (j == i) continue;
              const bool j0 = (in_f0 >> j) & 1, j1 = (in_f1 >> j) & 1;
              const bool v = (i0 && j0) || (i1 && j1) || (!i0 && !j0 && !i1 && !j1) || lead;
              vis |= (uint64_t)v << j;
              st_stream(o, v ? sPos[(2 * j) * TILE + lane] - pix : T(0));
              st_stream(o + ld, v ? sPos[(2 * j + 1) * TILE + lane] - piy : T(0));
              o += 2 * ld;
            }
            const bool i_adv = bit(d.adversary, i);
            if (!i_adv) {  // good agents: in_forest before other_vel (:313)
              st_stream(o, i0 ? T(1) : T(-1));
              st_stream(o + ld, i1 ? T(1) : T(-1));
              o += 2 * ld;
            }
            for (int j = 0; j < A; ++j) {
              if (j == i || bit(d.adversary, j)) continue;
              const bool v = (vis >> j) & 1;
              st_stream(o, v ? sVel[(2 * j) * TILE + lane] : T(0));
              st_stream(o + ld, v ? sVel[(2 * j + 1) * TILE + lane] : T(0));
              o += 2 * ld;
            }
            if (i_adv) {   // adversaries: other_vel, in_forest, agent 0's state.c (:292,305-310)
              st_stream(o, i0 ? T(1) : T(-1));
              st_stream(o + ld, i1 ? T(1) : T(-1));
              o += 2 * ld;
              for (int c = 0; c < d.dim_c; ++c) {
                T v = T(0);
                if (!bit(d.silent, 0))
                  v = state_c<T, DO_STEP>(p, 0, c, env);
                st_stream(o, v);
                o += ld;
              }
            }
            continue;
          }
          if (HIER && kind == MPE_SIMPLE_CRYPTO) {
            // simple_crypto.py:124-171: speaker [goal colour, key]; listener [key, speaker's c]; adversary [speaker's c]
            const int C = d.dim_c;
            const int gl = p.goal[ge], key = p.goal[ld + ge];
            if (i == 2) {
              for (int k = 0; k < C; ++k) { st_stream(o, k == gl ? T(1) : T(0)); o += ld; }
            }
            if (i != 0) {
              for (int k = 0; k < C; ++k) { st_stream(o, k == key ? T(1) : T(0)); o += ld; }
            }
            if (i != 2) {
              for (int k = 0; k < C; ++k) {
                st_stream(o, state_c<T, DO_STEP>(p, 2, k, env));
                o += ld;
              }
            }
            continue;
          }
          if (HIER && kind_is_comm(kind)) {
            // simple_speaker_listener.py:72-94 / simple_reference.py:62-89
            const bool ref = kind == MPE_SIMPLE_REFERENCE;
            int g = p.goal[(int64_t)(ref ? i : 0) * ld + ge];  // agent.goal_b
            g = g < 0 ? 0 : (g >= L ? L - 1 : g);
            const T base = ref ? T(0.25) : T(0.15), hi = ref ? T(0.75) : T(0.65);  // landmark.color tables
            if (!ref && !bit(d.movable, i)) {  // the speaker sees the goal colour only
              for (int k = 0; k < 3; ++k) { st_stream(o, k == g ? hi : base); o += ld; }
              continue;
            }
            st_stream(o, sVel[(2 * i) * TILE + lane]);
            st_stream(o + ld, sVel[(2 * i + 1) * TILE + lane]);
            o += 2 * ld;
            for (int l = A; l < E; ++l) {
              st_stream(o, sPos[(2 * l) * TILE + lane] - pix);
              st_stream(o + ld, sPos[(2 * l + 1) * TILE + lane] - piy);
              o += 2 * ld;
            }
            if (ref)
              for (int k = 0; k < 3; ++k) { st_stream(o, k == g ? hi : base); o += ld; }
            // other.state.c: what this step just set from the actions (core.py:226-230), or the comm
            // buffer when only observing
            for (int j = 0; j < A; ++j) {
              if (j == i) continue;
              for (int c = 0; c < d.dim_c; ++c) {
                T v = T(0);
                if (!bit(d.silent, j))
                  v = state_c<T, DO_STEP>(p, j, c, env);
                st_stream(o, v);
                o += ld;
              }
            }
            continue;
          }
          if (HIER && kind == MPE_HIER_MOVING) {
            // L1F4_MovingTarget.py:214-258 (S1L2F4_MovingTarget.py: radius only, :227)
            st_stream(o, sVel[(2 * i) * TILE + lane]);
            st_stream(o + ld, sVel[(2 * i + 1) * TILE + lane]);
            st_stream(o + 2 * ld, pix);
            st_stream(o + 3 * ld, piy);
            o += 4 * ld;
            const bool i_adv = bit(d.adversary, i);
            const T rad = d.obs_r[i];
            const int fi = desc_i8(d.force_id, i), fdi = desc_i8(d.forced_id, i);
            uint64_t vis = 0;
            for (int j = 0; j < A; ++j) {
              if (j == i) continue;
              const T pjx = sPos[(2 * j) * TILE + lane], pjy = sPos[(2 * j + 1) * TILE + lane];
              const T dx = pjx - pix, dy = pjy - piy;
              const bool linked = d.reward_variant == 0 &&
                                  ((fi >= 0 && fi == desc_i8(d.forced_id, j)) || (fdi >= 0 && fdi == desc_i8(d.force_id, j)));
              const bool v = !i_adv || linked || t_sqrt<T>(dx * dx + dy * dy) < rad;
              vis |= (uint64_t)v << j;
              // the prey sees everybody, in absolute coordinates (:249-251)
              st_stream(o, i_adv ? (v ? dx : T(0)) : pjx);
              st_stream(o + ld, i_adv ? (v ? dy : T(0)) : pjy);
              o += 2 * ld;
            }
            for (int j = 0; j < A; ++j) {
              if (j == i) continue;
              const bool v = (vis >> j) & 1;
              st_stream(o, v ? sVel[(2 * j) * TILE + lane] : T(0));
              st_stream(o + ld, v ? sVel[(2 * j + 1) * TILE + lane] : T(0));
              o += 2 * ld;
            }
            if (bit(d.obs_lf, i)) {  // (a world in which nobody is commanded keeps no leader_force rows: zeros, core.py:154)
              st_stream(o, has_lf ? sLf[(2 * i) * TILE + lane] : T(0));
              st_stream(o + ld, has_lf ? sLf[(2 * i + 1) * TILE + lane] : T(0));
            }
            continue;
          }
          if (HIER && kind == MPE_HIER_PURSUIT) {
            // L1F2_3Targets_100observe.py:198-253 (same text in every L1F*/S1L* file)
            st_stream(o, sVel[(2 * i) * TILE + lane]);
            st_stream(o + ld, sVel[(2 * i + 1) * TILE + lane]);
            st_stream(o + 2 * ld, pix);
            st_stream(o + 3 * ld, piy);
            o += 4 * ld;
            const T r2 = d.obs_r2[i];
            const int fi = desc_i8(d.force_id, i), fdi = desc_i8(d.forced_id, i);
            uint64_t vis = 0;  // others inside the radius or on the same command link (:209-216)
            for (int j = 0; j < A; ++j) {
              if (j == i) continue;
              const T dx = sPos[(2 * j) * TILE + lane] - pix, dy = sPos[(2 * j + 1) * TILE + lane] - piy;
              const bool linked = (fi >= 0 && fi == desc_i8(d.forced_id, j)) || (fdi >= 0 && fdi == desc_i8(d.force_id, j));
              const bool v = linked || (dx * dx + dy * dy < r2);
              vis |= (uint64_t)v << j;
              st_stream(o, v ? dx : T(0));
              st_stream(o + ld, v ? dy : T(0));
              o += 2 * ld;
            }
            for (int j = 0; j < A; ++j) {
              if (j == i) continue;
              const bool v = (vis >> j) & 1;
              st_stream(o, v ? sVel[(2 * j) * TILE + lane] : T(0));
              st_stream(o + ld, v ? sVel[(2 * j + 1) * TILE + lane] : T(0));
              o += 2 * ld;
            }
            if (fdi >= 0) {  // comm = [agent.leader_force] (:244-246)
              st_stream(o, sLf[(2 * i) * TILE + lane]);
              st_stream(o + ld, sLf[(2 * i + 1) * TILE + lane]);
              o += 2 * ld;
            }
            // visible targets, nearest first (stable sort by squared distance, :224-236), then zeros:
            // one selection pass per output slot over the (sq_dis, index) order.
            const int nt = L < d.n_target_obs ? L : d.n_target_obs;
            T prev_d = T(0);
            int prev_l = -1;
            bool more = true;
            for (int k = 0; k < nt; ++k) {
              int best_l = -1;
              T best_d = T(0), bx = T(0), by = T(0);
              if (more) {
                for (int l = 0; l < L; ++l) {
                  const T dx = sPos[(2 * (A + l)) * TILE + lane] - pix, dy = sPos[(2 * (A + l) + 1) * TILE + lane] - piy;
                  const T d2 = dx * dx + dy * dy;
                  const bool after = prev_l < 0 || d2 > prev_d || (d2 == prev_d && l > prev_l);
                  if (d2 < r2 && after && (best_l < 0 || d2 < best_d)) { best_l = l; best_d = d2; bx = dx; by = dy; }
                }
                more = best_l >= 0;
                prev_d = best_d;
                prev_l = best_l;
              }
              st_stream(o, bx);
              st_stream(o + ld, by);
              o += 2 * ld;
            }
            continue;
          }
          if (kind == MPE_SIMPLE_TAG || kind == MPE_SIMPLE_SPREAD) {
            st_stream(o, sVel[(2 * i) * TILE + lane]);
            st_stream(o + ld, sVel[(2 * i + 1) * TILE + lane]);
            st_stream(o + 2 * ld, pix);
            st_stream(o + 3 * ld, piy);
            o += 4 * ld;
          } else {
            // simple.py:45-50 / simple_adversary.py:122-140 / simple_push.py:84-104
            const bool i_good = !bit(d.adversary, i);
            if (kind != MPE_SIMPLE_ADVERSARY) {
              st_stream(o, sVel[(2 * i) * TILE + lane]);
              st_stream(o + ld, sVel[(2 * i + 1) * TILE + lane]);
              o += 2 * ld;
            }
            if (has_goal && i_good) {
              st_stream(o, gx - pix);
              st_stream(o + ld, gy - piy);
              o += 2 * ld;
              if (kind == MPE_SIMPLE_PUSH) {  // agent.color (simple_push.py:51-56)
                for (int k = 0; k < 3; ++k) { st_stream(o, k == goal + 1 ? T(0.25) + T(0.5) : T(0.25)); o += ld; }
              }
            }
          }
          for (int l = A; l < E; ++l) {
            st_stream(o, sPos[(2 * l) * TILE + lane] - pix);
            st_stream(o + ld, sPos[(2 * l + 1) * TILE + lane] - piy);
            o += 2 * ld;
          }
          if (kind == MPE_SIMPLE_PUSH && !bit(d.adversary, i)) {  // landmark.color (simple_push.py:43-45)
            for (int l = 0; l < L; ++l)
              for (int k = 0; k < 3; ++k) { st_stream(o, k == l + 1 ? T(0.1) + T(0.8) : T(0.1)); o += ld; }
          }
          if (kind != MPE_SIMPLE) {
            for (int j = 0; j < A; ++j) {
              if (j == i) continue;
              st_stream(o, sPos[(2 * j) * TILE + lane] - pix);
              st_stream(o + ld, sPos[(2 * j + 1) * TILE + lane] - piy);
              o += 2 * ld;
            }
          }
          if (kind == MPE_SIMPLE_TAG) {
            for (int j = 0; j < A; ++j) {
              if (j == i || bit(d.adversary, j)) continue;
              st_stream(o, sVel[(2 * j) * TILE + lane]);
              st_stream(o + ld, sVel[(2 * j + 1) * TILE + lane]);
              o += 2 * ld;
            }
          } else if (kind == MPE_SIMPLE_SPREAD) {
            for (int z = 0; z < d.dim_c * (A - 1); ++z) { st_stream(o, T(0)); o += ld; }
          }
        }
      }
      __syncthreads();
      // ---- phase 3b: reward / info (sAux aliases sVel from here on) ---------------------------------------------------------------
#ifdef MPE_ABL_NO_REW
      if ((want_rew || want_info || want_stats) && p.B < 0) {
#else
      if (want_rew || want_info || want_stats) {
#endif
        if (kind == MPE_SIMPLE_TAG) {
          // hit(g,a) = is_collision(good g, adversary a) (simple_tag.py:69-73), each pair evaluated
          // once by the chunk that owns g; per-adversary counts meet in LDS (ds_add, rare).
          // LGLOB: four adversaries per int (8 bits each; a count is at most n_good <= 64), so that the scratch fits
          // the good agents' velocity rows it aliases
          constexpr int PK = LGLOB ? 4 : 1;
          int *sCnt = reinterpret_cast<int *>(sAux);        // [ceil(n_adv / PK)][TILE]
          const int ncnt = (d.n_adv + PK - 1) / PK;
          int *sTot = sCnt + ncnt * TILE;                   // [NCH][TILE]
          for (int ai = chunk; ai < ncnt; ai += NCH) sCnt[ai * TILE + lane] = 0;
          __syncthreads();
          int part = 0;  // this chunk's share of sum_{g,a} hit
          for (int gi = chunk; gi < d.n_good; gi += NCH) {
            const int g = desc_u8(d.good_list, gi);
            const T pgx = sPos[(2 * g) * TILE + lane], pgy = sPos[(2 * g + 1) * TILE + lane];
            const T sg = d.size[g];
            int n = 0;
            for (int ai = 0; ai < d.n_adv; ++ai) {
              const int a = desc_u8(d.adv_list, ai);
              if (touching<T>(pgx - sPos[(2 * a) * TILE + lane], pgy - sPos[(2 * a + 1) * TILE + lane],
                              sg + d.size[a])) {
                ++n;
                atomicAdd(&sCnt[(ai / PK) * TILE + lane], 1 << (8 * (ai % PK)));
              }
            }
            part += n;
            // agent_reward (simple_tag.py:89-113)
            T r_g = bit(d.collide, g) ? -T(10) * T(n) : T(0);
            r_g = r_g - tag_bound<T>(t_abs<T>(pgx));
            r_g = r_g - tag_bound<T>(t_abs<T>(pgy));
            if (want_rew && active) st_stream(p.rew + (int64_t)g * ld + env, r_g);
            if (want_info && active) st_stream(p.info + (int64_t)g * ld + env, T(0));  // simple_tag.py:65-66
            if (want_stats) {
              const double s_r = group_sum(active ? (double)r_g : 0.0, TILE);
              if (lane == 0) srow[g] += s_r;
            }
          }
          sTot[chunk * TILE + lane] = part;
          __syncthreads();
          int total = 0;
#pragma unroll
          for (int c = 0; c < NCH; ++c) total += sTot[c * TILE + lane];
          // adversary_reward (simple_tag.py:115-129): same value for every adversary;
          // benchmark_data (simple_tag.py:57-64): goods touching this adversary
          for (int ai = chunk; ai < d.n_adv; ai += NCH) {
            const int i = desc_u8(d.adv_list, ai);
            const int n = LGLOB ? (sCnt[(ai / PK) * TILE + lane] >> (8 * (ai % PK))) & 0xFF : sCnt[ai * TILE + lane];
            const T r_i = bit(d.collide, i) ? T(10) * T(total) : T(0);
            if (want_rew && active) st_stream(p.rew + (int64_t)i * ld + env, r_i);
            if (want_info && active) st_stream(p.info + (int64_t)i * ld + env, T(n));
            if (want_stats) {
              const double s_r = group_sum(active ? (double)r_i : 0.0, TILE);
              const double s_c = group_sum(active ? (double)n : 0.0, TILE);
              if (lane == 0) { srow[i] += s_r; srow[A + i] += s_c; }
            }
          }
          if (want_stats && chunk == 0) {
            const double s_t = group_sum(active ? (double)total : 0.0, TILE);
            const double s_n = group_sum(active ? 1.0 : 0.0, TILE);
            if (lane == 0) { srow[2 * A] += s_t; srow[2 * A + 1] += s_n; }
          }
        } else if (HIER && kind == MPE_WORLD_COMM) {
          // simple_world_comm.py:119-204.  Every agent's sums are evaluated by the chunk that owns the agent, in
          // the reference's order (the running reward is not an integer here: -0.1 * min distance comes first).
          const int food0 = E - d.n_forests - d.n_food, food1 = E - d.n_forests;
          auto hit = [&](int a, int b) {
            return touching<T>(sPos[(2 * a) * TILE + lane] - sPos[(2 * b) * TILE + lane],
                               sPos[(2 * a + 1) * TILE + lane] - sPos[(2 * b + 1) * TILE + lane], d.size[a] + d.size[b]);
          };
          auto dist = [&](int a, int b) {
            return t_sqrt<T>(sqdist<T>(sPos[(2 * a) * TILE + lane], sPos[(2 * a + 1) * TILE + lane],
                                       sPos[(2 * b) * TILE + lane], sPos[(2 * b + 1) * TILE + lane]));
          };
          for (int i = chunk; i < A; i += NCH) {
            T r_i = T(0);
            int c_i = 0;
            if (bit(d.adversary, i)) {
              T m = T(0);
              for (int gi = 0; gi < d.n_good; ++gi) {
                const T dg = dist(desc_u8(d.good_list, gi), i);
                m = (gi == 0 || dg < m) ? dg : m;
              }
              r_i = r_i - T(0.1) * m;
              if (bit(d.collide, i))
                for (int gi = 0; gi < d.n_good; ++gi)
                  for (int ai = 0; ai < d.n_adv; ++ai)
                    r_i = hit(desc_u8(d.good_list, gi), desc_u8(d.adv_list, ai)) ? r_i + T(5) : r_i;
              for (int gi = 0; gi < d.n_good; ++gi) c_i += hit(desc_u8(d.good_list, gi), i) ? 1 : 0;
            } else {
              if (bit(d.collide, i))
                for (int ai = 0; ai < d.n_adv; ++ai) r_i = hit(desc_u8(d.adv_list, ai), i) ? r_i - T(5) : r_i;
              r_i = r_i - T(2) * tag_bound<T>(t_abs<T>(sPos[(2 * i) * TILE + lane]));
              r_i = r_i - T(2) * tag_bound<T>(t_abs<T>(sPos[(2 * i + 1) * TILE + lane]));
              T m = T(0);
              for (int f = food0; f < food1; ++f) {
                r_i = hit(i, f) ? r_i + T(2) : r_i;
                const T df = dist(f, i);
                m = (f == food0 || df < m) ? df : m;
              }
              r_i = r_i + T(0.05) * m;
            }
            if (want_rew && active) st_stream(p.rew + (int64_t)i * ld + env, r_i);
            if (want_info && active) st_stream(p.info + (int64_t)i * ld + env, T(c_i));
            if (want_stats) {
              const double s_r = group_sum(active ? (double)r_i : 0.0, TILE);
              const double s_c = group_sum(active ? (double)c_i : 0.0, TILE);
              if (lane == 0) { srow[i] += s_r; srow[A + i] += s_c; }
            }
          }
          if (want_stats && chunk == 0) {
            int total = 0;
            for (int gi = 0; gi < d.n_good; ++gi)
              for (int ai = 0; ai < d.n_adv; ++ai) total += hit(desc_u8(d.good_list, gi), desc_u8(d.adv_list, ai)) ? 1 : 0;
            const double s_t = group_sum(active ? (double)total : 0.0, TILE);
            const double s_n = group_sum(active ? 1.0 : 0.0, TILE);
            if (lane == 0) { srow[2 * A] += s_t; srow[2 * A + 1] += s_n; }
          }
        } else if (HIER && kind == MPE_SIMPLE_CRYPTO) {
          // simple_crypto.py:78-121: squared error of Eve's / Bob's message against the goal colour; an
          // all-zero message switches its term off.  Every chunk recomputes the two sums (tiny world).
          const int C = d.dim_c;
          const int gl = p.goal[ge];
          auto msg = [&](int a, int k) {
            return state_c<T, DO_STEP>(p, a, k, ge);
          };
          T err[2];
          bool spoke[2];
          for (int a = 0; a < 2; ++a) {
            T s = T(0);
            bool any = false;
            for (int k = 0; k < C; ++k) {
              const T c = msg(a, k), dlt = c - (k == gl ? T(1) : T(0));
              s = k == 0 ? dlt * dlt : s + dlt * dlt;  // np.sum: left to right
              any = any || c != T(0);
            }
            err[a] = s;
            spoke[a] = any;
          }
          const T eve = spoke[0] ? err[0] : T(0);
          const T r_good = eve + (spoke[1] ? -err[1] : T(0));  // adv_rew + good_rew (:114)
          for (int i = chunk; i < A; i += NCH) {
            const T r_i = i == 0 ? (spoke[0] ? -err[0] : T(0)) : r_good;
            if (want_rew && active) st_stream(p.rew + (int64_t)i * ld + env, r_i);
            if (want_info && active) {  // (agent.state.c, agent.goal_a.color)
              T *ir = p.info + (int64_t)(2 * C * i) * ld + env;
              for (int k = 0; k < C; ++k) st_stream(ir + (int64_t)k * ld, msg(i, k));
              for (int k = 0; k < C; ++k) st_stream(ir + (int64_t)(C + k) * ld, k == gl ? T(1) : T(0));
            }
            if (want_stats) {
              const double s_r = group_sum(active ? (double)r_i : 0.0, TILE);
              if (lane == 0) srow[i] += s_r;
            }
          }
          if (want_stats && chunk == 0) {
            const double s_n = group_sum(active ? 1.0 : 0.0, TILE);
            if (lane == 0) srow[2 * A + 1] += s_n;
          }
        } else if (HIER && kind_is_comm(kind)) {
          // simple_speaker_listener.py:66-70 (both agents: listener to agent 0's goal) /
          // simple_reference.py:56-60 (agent i: the OTHER agent to agent i's goal)
          const bool ref = kind == MPE_SIMPLE_REFERENCE;
          for (int i = chunk; i < A; i += NCH) {
            int g = p.goal[(int64_t)(ref ? i : 0) * ld + ge];
            g = g < 0 ? 0 : (g >= L ? L - 1 : g);
            const int who = ref ? 1 - i : 1;
            const T r_i = -sqdist<T>(sPos[(2 * who) * TILE + lane], sPos[(2 * who + 1) * TILE + lane],
                                     sPos[(2 * (A + g)) * TILE + lane], sPos[(2 * (A + g) + 1) * TILE + lane]);
            if (want_rew && active) st_stream(p.rew + (int64_t)i * ld + env, r_i);
            if (want_stats) {
              const double s_r = group_sum(active ? (double)r_i : 0.0, TILE);
              if (lane == 0) srow[i] += s_r;
            }
          }
          if (want_stats && chunk == 0) {
            const double s_n = group_sum(active ? 1.0 : 0.0, TILE);
            if (lane == 0) srow[2 * A + 1] += s_n;
          }
        } else if (HIER && kind == MPE_HIER_MOVING) {
          // L1F4_MovingTarget.py:157-212.  Every adversary gets the same sum: 10 per (good agent, anybody)
          // contact, +50 after each good agent once the running count has reached 4.  Integers: exact.
          auto hit = [&](int a, int b) {
            return touching<T>(sPos[(2 * a) * TILE + lane] - sPos[(2 * b) * TILE + lane],
                               sPos[(2 * a + 1) * TILE + lane] - sPos[(2 * b + 1) * TILE + lane], d.size[a] + d.size[b]);
          };
          int count = 0, bonus = 0, caught = 0;  // caught: (good, adversary) contacts
          for (int gi = 0; gi < d.n_good; ++gi) {
            const int g = desc_u8(d.good_list, gi);
            for (int o = 0; o < A; ++o) {
              if (o == g) continue;
              const int h = hit(g, o) ? 1 : 0;
              count += h;
              caught += bit(d.adversary, o) ? h : 0;
            }
            bonus += count >= 4 ? 1 : 0;
          }
          for (int i = chunk; i < A; i += NCH) {
            const bool i_adv = bit(d.adversary, i);
            int c_i = 0;  // benchmark_data (:128-140): touching agents of the other team
            for (int a = 0; a < A; ++a)
              if (bit(d.adversary, a) != i_adv) c_i += hit(a, i) ? 1 : 0;
            T r_i = i_adv ? T(10 * count + 50 * bonus) : (bit(d.collide, i) ? T(0) - T(c_i) : T(0));
            r_i = r_i - tag_bound<T>(t_abs<T>(sPos[(2 * i) * TILE + lane]));
            r_i = r_i - tag_bound<T>(t_abs<T>(sPos[(2 * i + 1) * TILE + lane]));
            if (want_rew && active) st_stream(p.rew + (int64_t)i * ld + env, r_i);
            if (want_info && active) st_stream(p.info + (int64_t)i * ld + env, T(c_i));
            if (want_stats) {
              const double s_r = group_sum(active ? (double)r_i : 0.0, TILE);
              const double s_c = group_sum(active ? (double)c_i : 0.0, TILE);
              if (lane == 0) { srow[i] += s_r; srow[A + i] += s_c; }
            }
          }
          if (want_stats && chunk == 0) {
            const double s_t = group_sum(active ? (double)caught : 0.0, TILE);
            const double s_n = group_sum(active ? 1.0 : 0.0, TILE);
            if (lane == 0) { srow[2 * A] += s_t; srow[2 * A + 1] += s_n; }
          }
        } else if (HIER && kind == MPE_HIER_PURSUIT) {
          // n_o = number of targets agent o touches (is_collision, L1F2_3Targets_100observe.py:150-154)
          int *sCnt = reinterpret_cast<int *>(sAux);  // [A][TILE]
          for (int o = chunk; o < A; o += NCH) {
            const T pox = sPos[(2 * o) * TILE + lane], poy = sPos[(2 * o + 1) * TILE + lane];
            int n = 0;
            for (int l = A; l < E; ++l)
              n += touching<T>(sPos[(2 * l) * TILE + lane] - pox, sPos[(2 * l + 1) * TILE + lane] - poy,
                               d.size[l] + d.size[o]) ? 1 : 0;
            sCnt[o * TILE + lane] = n;
          }
          __syncthreads();
          int follower_hits = 0;
          if (want_stats && chunk == 0)
            for (int o = 0; o < A; ++o) follower_hits += bit(d.advleader, o) ? 0 : sCnt[o * TILE + lane];
          for (int i = chunk; i < A; i += NCH) {
            const T pix = sPos[(2 * i) * TILE + lane], piy = sPos[(2 * i + 1) * TILE + lane];
            const bool lead = bit(d.advleader, i), super = bit(d.advsuper, i);
            const int fi = desc_i8(d.force_id, i);
            // every term of the reference's running sum is +10 (or +1): exact in T, order-free
            int n = 0;
            T unit = T(10);
            // variant 0 tests advleader first (:167-180), rewardFixed tests advsuperleader first (:127-150)
            const bool own = d.reward_variant == 0 ? !lead : (!super && !lead);
            if (own) {
              n = sCnt[i * TILE + lane];
            } else if (d.reward_variant == 0 && super) {
              for (int o = 0; o < A; ++o) n += bit(d.advleader, o) ? 0 : sCnt[o * TILE + lane];  // :176-180
            } else if (d.reward_variant == 1 && super) {
              // S1L2F2_10Targets_wide_rewardFixed.py:127-137: is_observed(target, target) is 0 < obs_r
              unit = T(1);
              for (int o = 0; o < A; ++o) n += (desc_i8(d.forced_id, o) == fi && T(0) < d.obs_r2[o]) ? L : 0;
            } else {
              for (int o = 0; o < A; ++o) n += desc_i8(d.forced_id, o) == fi ? sCnt[o * TILE + lane] : 0;  // :171-175
            }
            T r_i = unit * T(n);
            r_i = r_i - tag_bound<T>(t_abs<T>(pix));  // :183-193
            r_i = r_i - tag_bound<T>(t_abs<T>(piy));
            // benchmark_data (:136-148): touching agents of the other team
            int c_i = 0;
            const bool i_adv = bit(d.adversary, i);
            for (int a = 0; a < A; ++a)
              if (bit(d.adversary, a) != i_adv)
                c_i += touching<T>(sPos[(2 * a) * TILE + lane] - pix, sPos[(2 * a + 1) * TILE + lane] - piy,
                                   d.size[a] + d.size[i]) ? 1 : 0;
            if (want_rew && active) st_stream(p.rew + (int64_t)i * ld + env, r_i);
            if (want_info && active) st_stream(p.info + (int64_t)i * ld + env, T(c_i));
            if (want_stats) {
              const double s_r = group_sum(active ? (double)r_i : 0.0, TILE);
              const double s_c = group_sum(active ? (double)c_i : 0.0, TILE);
              if (lane == 0) { srow[i] += s_r; srow[A + i] += s_c; }
            }
          }
          if (want_stats && chunk == 0) {
            const double s_t = group_sum(active ? (double)follower_hits : 0.0, TILE);
            const double s_n = group_sum(active ? 1.0 : 0.0, TILE);
            if (lane == 0) { srow[2 * A] += s_t; srow[2 * A + 1] += s_n; }
          }
        } else if (kind != MPE_SIMPLE_SPREAD) {
          // simple.py:41-43 / simple_adversary.py:58-119 / simple_push.py:66-82.  The per-env scalars
          // (nearest good agent to the goal, summed adversary distances) are recomputed by every
          // chunk: these worlds are small.
          T m = T(0), adv_rew = T(0);
          if (has_goal) {
            for (int gi = 0; gi < d.n_good; ++gi) {
              const int g = desc_u8(d.good_list, gi);
              const T dist = t_sqrt<T>(sqdist<T>(sPos[(2 * g) * TILE + lane], sPos[(2 * g + 1) * TILE + lane], gx, gy));
              m = (gi == 0 || dist < m) ? dist : m;
            }
            if (kind == MPE_SIMPLE_ADVERSARY)
              for (int ai = 0; ai < d.n_adv; ++ai) {
                const int a = desc_u8(d.adv_list, ai);
                adv_rew = adv_rew + t_sqrt<T>(sqdist<T>(sPos[(2 * a) * TILE + lane], sPos[(2 * a + 1) * TILE + lane], gx, gy));
              }
          }
          for (int i = chunk; i < A; i += NCH) {
            const T pix = sPos[(2 * i) * TILE + lane], piy = sPos[(2 * i + 1) * TILE + lane];
            const bool i_adv = bit(d.adversary, i);
            T r_i;
            if (kind == MPE_SIMPLE) {
              r_i = -sqdist<T>(pix, piy, sPos[(2 * A) * TILE + lane], sPos[(2 * A + 1) * TILE + lane]);
            } else if (kind == MPE_SIMPLE_ADVERSARY) {
              r_i = i_adv ? -sqdist<T>(pix, piy, gx, gy) : -m + adv_rew;
            } else {
              const T dg = t_sqrt<T>(sqdist<T>(pix, piy, gx, gy));
              r_i = i_adv ? m - dg : -dg;
            }
            if (want_rew && active) st_stream(p.rew + (int64_t)i * ld + env, r_i);
            if (want_info && active && d.info_w > 0) {  // simple_adversary.py:58-67
              T *ir = p.info + (int64_t)(d.info_w * i) * ld + env;
              for (int l = 0; l < L; ++l) {
                const T v = sqdist<T>(pix, piy, sPos[(2 * (A + l)) * TILE + lane], sPos[(2 * (A + l) + 1) * TILE + lane]);
                st_stream(ir + (int64_t)l * ld, i_adv ? (l == 0 ? sqdist<T>(pix, piy, gx, gy) : T(0)) : v);
              }
              st_stream(ir + (int64_t)L * ld, i_adv ? T(0) : sqdist<T>(pix, piy, gx, gy));
            }
            if (want_stats) {
              const double s_r = group_sum(active ? (double)r_i : 0.0, TILE);
              if (lane == 0) srow[i] += s_r;
            }
          }
          if (want_stats && chunk == 0) {
            const double s_n = group_sum(active ? 1.0 : 0.0, TILE);
            if (lane == 0) srow[2 * A + 1] += s_n;
          }
        } else {
          // simple_spread: m_l = min_a dist(a, landmark l) (simple_spread.py:74-76), summed in
          // ascending l by every lane so the float order matches the reference.
          for (int l = chunk; l < L; l += NCH) {
            const T plx = sPos[(2 * (A + l)) * TILE + lane], ply = sPos[(2 * (A + l) + 1) * TILE + lane];
            T m = T(0);
            for (int a = 0; a < A; ++a) {
              const T dx = sPos[(2 * a) * TILE + lane] - plx, dy = sPos[(2 * a + 1) * TILE + lane] - ply;
              const T dist = t_sqrt<T>(dx * dx + dy * dy);
              m = (a == 0 || dist < m) ? dist : m;
            }
            sAux[l * TILE + lane] = m;
          }
          __syncthreads();
          T r0 = T(0), md = T(0);
          int occ = 0;
          for (int l = 0; l < L; ++l) {
            const T m = sAux[l * TILE + lane];
            md = md + m;
            r0 = r0 - m;
            occ += (m < T(0.1)) ? 1 : 0;
          }
          for (int i = chunk; i < A; i += NCH) {
            const T pix = sPos[(2 * i) * TILE + lane], piy = sPos[(2 * i + 1) * TILE + lane];
            T r_i = r0;
            int c_i = 0;
            if (bit(d.collide, i)) {
              for (int a = 0; a < A; ++a) {  // includes a == i (simple_spread.py:78-80)
                if (touching<T>(sPos[(2 * a) * TILE + lane] - pix, sPos[(2 * a + 1) * TILE + lane] - piy,
                                d.size[a] + d.size[i])) {
                  r_i = r_i - T(1);
                  c_i += 1;
                }
              }
            }
            if (active) {
              if (want_rew) st_stream(p.rew + (int64_t)i * ld + env, r_i);
              if (want_info) {
                T *ir = p.info + (int64_t)(4 * i) * ld + env;
                st_stream(ir, r_i); st_stream(ir + ld, T(c_i)); st_stream(ir + 2 * ld, md); st_stream(ir + 3 * ld, T(occ));
              }
            }
            if (want_stats) {
              const double s_r = group_sum(active ? (double)r_i : 0.0, TILE);
              const double s_c = group_sum(active ? (double)c_i : 0.0, TILE);
              if (lane == 0) { srow[i] += s_r; srow[A + i] += s_c; }
            }
          }
          if (want_stats && chunk == 0) {
            const double s_t = group_sum(active ? (double)occ : 0.0, TILE);
            const double s_n = group_sum(active ? 1.0 : 0.0, TILE);
            if (lane == 0) { srow[2 * A] += s_t; srow[2 * A + 1] += s_n; }
          }
        }
      }
    }
    __syncthreads();  // LDS is reused by the next tile
  }
}

// LDS bytes: pos rows + max(vel rows, aux rows); aux = (n_adv + NCH) int rows (tag) or L rows of T (spread).
template <typename T>
inline size_t tile_lds_bytes(const mpe_scenario_desc *h, int threads) {
  constexpr int TILE = tile_envs<T>();
  const int NCH = threads / TILE;
  const int E = h->n_agents + h->n_landmarks;
  size_t aux = h->kind == MPE_SIMPLE_TAG ? (size_t)(NCH + h->n_agents) * TILE * sizeof(int)
                                         : (size_t)h->n_landmarks * TILE * sizeof(T);  // hierarchy: A ints < vel rows
  bool lm_moves = false;  // a movable landmark: velocity rows for every entity
  for (int e = h->n_agents; e < E; ++e) lm_moves = lm_moves || h->movable[e];
  size_t velb = (size_t)(2 * (lm_moves ? E : h->n_agents)) * TILE * sizeof(T);
  if (velb < aux) velb = aux;
  size_t lfb = 0;
  for (int a = 0; a < h->n_agents; ++a)
    if (h->forced_id[a] >= 0) lfb = (size_t)(2 * h->n_agents) * TILE * sizeof(T);
  return (size_t)(2 * E) * TILE * sizeof(T) + lfb + velb;
}

// LDS bytes of the LGLOB form (simple_tag, fused step): agents' positions + max(good agents' velocity rows, reward scratch).
template <typename T>
inline size_t tile_lds_bytes_lglob(const mpe_scenario_desc *h, int threads) {
  constexpr int TILE = tile_envs<T>();
  const int NCH = threads / TILE;
  int n_good = 0;
  for (int a = 0; a < h->n_agents; ++a) n_good += h->adversary[a] ? 0 : 1;
  const int n_adv = h->n_agents - n_good;
  size_t velb = (size_t)(2 * n_good) * TILE * sizeof(T);
  const size_t aux = (size_t)((n_adv + 3) / 4 + NCH) * TILE * sizeof(int);
  if (velb < aux) velb = aux;
  return (size_t)(2 * h->n_agents) * TILE * sizeof(T) + velb;
}

}  // namespace mpe
