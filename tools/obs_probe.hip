// Store-phase probe for the scaled simple_tag tile kernel (48 adversaries + 16 good agents, 32 landmarks:
// 14 432 observation rows of [ld] floats per env tile): what HBM rate can the observation WRITE pattern itself
// reach on MI355X, as a function of how the rows are cut into store instructions?  Entity state sits in LDS
// (dummy values), nothing is loaded from HBM, every variant writes the same bytes to the same addresses.
//   hipcc -O3 --offload-arch=gfx950 tools/obs_probe.hip -o tools/obs_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int A = 64, L = 32, E = A + L, NADV = 48;
constexpr int ODIM_ADV = 4 + 2 * L + 2 * (A - 1) + 2 * (A - NADV);      // 226
constexpr int ODIM_GOOD = ODIM_ADV - 2;                                  // 224
constexpr int OBS_TOTAL = NADV * ODIM_ADV + (A - NADV) * ODIM_GOOD;      // 14432
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int obs_row0(int i) { return i < NADV ? i * ODIM_ADV : NADV * ODIM_ADV + (i - NADV) * ODIM_GOOD; }

// Synthetic stand-in for the load / pair-force / reward phases of a tile: `iters` trips of 16 independent FMAs per
// lane (pure VALU, nothing in memory), then a barrier, like the real kernel's phases in front of its stores.
__device__ __forceinline__ void compute_phase(int iters, float *sink) {
  if (iters <= 0) return;
  float a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = (float)(threadIdx.x + k);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = __builtin_fmaf(a[k], 1.0001f, 0.5f);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += a[k];
  if (s == 123.456f) *sink = s;
  __syncthreads();
}

template <bool NT, typename V>
__device__ __forceinline__ void st(V *p, V v) {
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// ---- V0: the shipped shape: lane = env, wave w owns agents w, w+NW, ...; one dword store per row -----------------
template <int TILE, int THREADS, bool NT, int UNROLL>
__global__ __launch_bounds__(THREADS) void obs_dword(float *__restrict__ obs, long ld, long ntiles, int citers, int do_store) {
  extern __shared__ float lds[];
  float *sPos = lds, *sVel = lds + 2 * E * TILE;
  constexpr int NW = THREADS / TILE;
  const int lane = threadIdx.x % TILE, w = __builtin_amdgcn_readfirstlane(threadIdx.x / TILE);
  for (int r = threadIdx.x; r < (2 * E + 2 * A) * TILE; r += THREADS) lds[r] = (float)(r & 1023) * 0.001f;
  __syncthreads();
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long env = tile * TILE + lane;
    compute_phase(citers, obs);
    if (do_store)
    for (int i = w; i < A; i += NW) {
      float *o = obs + (long)obs_row0(i) * ld + env;
      const float pix = sPos[(2 * i) * TILE + lane], piy = sPos[(2 * i + 1) * TILE + lane];
      st<NT>(o, sVel[(2 * i) * TILE + lane]); st<NT>(o + ld, sVel[(2 * i + 1) * TILE + lane]);
      st<NT>(o + 2 * ld, pix); st<NT>(o + 3 * ld, piy);
      o += 4 * ld;
#pragma unroll UNROLL
      for (int l = A; l < E; ++l) {
        st<NT>(o, sPos[(2 * l) * TILE + lane] - pix); st<NT>(o + ld, sPos[(2 * l + 1) * TILE + lane] - piy);
        o += 2 * ld;
      }
#pragma unroll UNROLL
      for (int j = 0; j < i; ++j) {
        st<NT>(o, sPos[(2 * j) * TILE + lane] - pix); st<NT>(o + ld, sPos[(2 * j + 1) * TILE + lane] - piy);
        o += 2 * ld;
      }
#pragma unroll UNROLL
      for (int j = i + 1; j < A; ++j) {
        st<NT>(o, sPos[(2 * j) * TILE + lane] - pix); st<NT>(o + ld, sPos[(2 * j + 1) * TILE + lane] - piy);
        o += 2 * ld;
      }
      for (int j = NADV; j < A; ++j) {
        if (j == i) continue;
        st<NT>(o, sVel[(2 * j) * TILE + lane]); st<NT>(o + ld, sVel[(2 * j + 1) * TILE + lane]);
        o += 2 * ld;
      }
    }
  }
}

// ---- V1: four rows per store instruction: lane = (row r = lane / 16, env quad q = lane % 16), dwordx4 -----------------
// Block b of an agent's rows = rows [4b, 4b + 4); source of obs row k of agent i: see src_row().  TILE = 64 only.
__device__ __forceinline__ int src_row(int i, int k, bool &is_rel, bool &is_vel) {
  // returns the LDS row index (into sPos rows 0..2E-1, or sVel rows when is_vel) that obs row k of agent i copies
  is_rel = false; is_vel = false;
  if (k < 2) { is_vel = true; return 2 * i + k; }
  if (k < 4) return 2 * i + (k - 2);
  k -= 4;
  if (k < 2 * L) { is_rel = true; return 2 * A + k; }
  k -= 2 * L;
  if (k < 2 * (A - 1)) { is_rel = true; return k < 2 * i ? k : k + 2; }
  k -= 2 * (A - 1);
  is_vel = true;
  int j = NADV + (k >> 1);
  if (i >= NADV && j >= i) ++j;
  return 2 * j + (k & 1);
}

template <int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void obs_x4(float *__restrict__ obs, long ld, long ntiles) {
  constexpr int TILE = 64;
  extern __shared__ float lds[];
  float *sPos = lds, *sVel = lds + 2 * E * TILE;
  constexpr int NW = THREADS / 64;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane & 15, r = lane >> 4;
  for (int x = threadIdx.x; x < (2 * E + 2 * A) * TILE; x += THREADS) lds[x] = (float)(x & 1023) * 0.001f;
  __syncthreads();
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long env0 = tile * TILE + 4 * q;
    for (int i = w; i < A; i += NW) {
      const int odim = i < NADV ? ODIM_ADV : ODIM_GOOD;
      float *o = obs + (long)obs_row0(i) * ld + env0;
      const f4 pi = *reinterpret_cast<const f4 *>(sPos + (2 * i + (r & 1)) * TILE + 4 * q);   // rows alternate x / y
      for (int k0 = 0; k0 < odim; k0 += 4) {
        const int k = k0 + r;
        bool rel, vel;
        const int row = src_row(i, k < odim ? k : odim - 1, rel, vel);
        f4 v = *reinterpret_cast<const f4 *>((vel ? sVel : sPos) + row * TILE + 4 * q);
        if (rel) v = v - pi;
        if (k < odim) st<NT>(reinterpret_cast<f4 *>(o + (long)k * ld), v);
      }
    }
  }
}

// ---- V2: same four-rows-per-instruction stores, but the 4 rows of an instruction come from a host-built plan ---------
// plan[i][b][r] = (lds float offset of the source row) | rel << 30 | valid << 31, read with one scalar-indexed load per
// block (wave-uniform address + r): removes the per-lane integer work of src_row() from the loop.
template <int THREADS, bool NT, int UNROLL>
__global__ __launch_bounds__(THREADS) void obs_x4_plan(float *__restrict__ obs, long ld, long ntiles, const unsigned *__restrict__ plan, int citers, int do_store) {
  constexpr int TILE = 64;
  extern __shared__ float lds[];
  constexpr int NW = THREADS / 64;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane & 15, r = lane >> 4;
  for (int x = threadIdx.x; x < (2 * E + 2 * A) * TILE; x += THREADS) lds[x] = (float)(x & 1023) * 0.001f;
  __syncthreads();
  constexpr int NB = (ODIM_ADV + 3) / 4;   // 57 blocks per agent
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long env0 = tile * TILE + 4 * q;
    compute_phase(citers, obs);
    if (do_store)
    for (int i = w; i < A; i += NW) {
      float *o = obs + ((long)obs_row0(i) + r) * ld + env0;
      const f4 pi = *reinterpret_cast<const f4 *>(lds + (2 * i + (r & 1)) * TILE + 4 * q);
      const unsigned *pl = plan + (i * NB) * 4 + r;
#pragma unroll UNROLL
      for (int b = 0; b < NB; ++b) {
        const unsigned e = pl[b * 4];
        f4 v = *reinterpret_cast<const f4 *>(lds + (e & 0x3FFFFFFFu) + 4 * q);
        const float m = (e & 0x40000000u) ? 1.0f : 0.0f;
        v = v - pi * m;
        if (e & 0x80000000u) st<NT>(reinterpret_cast<f4 *>(o + (long)(4 * b) * ld), v);
      }
    }
  }
}


// ---- V4: V2 + the tile's input rows (pos 192, vel 128, act 320 = 640 rows x 256 B at row stride ld) loaded into LDS in
// front of the stores, like the real kernel.  PREFETCH: persistent workgroups keep the NEXT tile's rows in flight in
// registers while they store the current tile (80 rows per wave = 80 VGPRs at 8 waves).
template <int THREADS, bool NT, int UNROLL, bool PREFETCH, int NIN>
__global__ __launch_bounds__(THREADS) void obs_x4_load(float *__restrict__ obs, long ld, long ntiles, const unsigned *__restrict__ plan,
                                                        const float *__restrict__ in, int citers,
                                                        unsigned long long *__restrict__ stamps) {
  constexpr int TILE = 64;
  extern __shared__ float lds[];
  constexpr int NW = THREADS / 64;
  constexpr int PER = NIN / NW;   // input rows per wave
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane & 15, r = lane >> 4;
  constexpr int NB = (ODIM_ADV + 3) / 4;
  float pre[PREFETCH ? PER : 1];
  if (PREFETCH) {
    const long env = (long)blockIdx.x * TILE + lane;
#pragma unroll
    for (int k = 0; k < PER; ++k) pre[k] = in[(long)(w + k * NW) * ld + env];
  }
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long env = tile * TILE + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (!PREFETCH) {
#pragma unroll 1
      for (int k0 = 0; k0 < PER; k0 += 16) {
        float tmp[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) tmp[k] = in[(long)(w + (k0 + k) * NW) * ld + env];
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[((w + (k0 + k) * NW) % (2 * E + 2 * A)) * TILE + lane] = tmp[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < PER; ++k) lds[((w + k * NW) % (2 * E + 2 * A)) * TILE + lane] = pre[k];
    }
    __syncthreads();
    if (PREFETCH && tile + gridDim.x < ntiles) {
      const long env2 = (tile + gridDim.x) * TILE + lane;
#pragma unroll
      for (int k = 0; k < PER; ++k) pre[k] = in[(long)(w + k * NW) * ld + env2];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    const bool inter = citers < 0;          // negative: the same compute, cut into one slice per agent between its stores
    const int cit = inter ? -citers : citers;
    if (!inter) compute_phase(cit, obs);
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    const long env0 = tile * TILE + 4 * q;
    for (int i = w; i < A; i += NW) {
      if (inter) {
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = (float)(threadIdx.x + k);
        for (int it = 0; it < cit / (A / NW); ++it) {
#pragma unroll
          for (int k = 0; k < 16; ++k) a[k] = __builtin_fmaf(a[k], 1.0001f, 0.5f);
        }
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sm += a[k];
        if (sm == 123.456f) *obs = sm;
      }
      float *o = obs + ((long)obs_row0(i) + r) * ld + env0;
      const f4 pi = *reinterpret_cast<const f4 *>(lds + (2 * i + (r & 1)) * TILE + 4 * q);
      const unsigned *pl = plan + (i * NB) * 4 + r;
#pragma unroll UNROLL
      for (int b = 0; b < NB; ++b) {
        const unsigned e = pl[b * 4];
        f4 v = *reinterpret_cast<const f4 *>(lds + (e & 0x3FFFFFFFu) + 4 * q);
        const float m = (e & 0x40000000u) ? 1.0f : 0.0f;
        v = v - pi * m;
        if (e & 0x80000000u) st<NT>(reinterpret_cast<f4 *>(o + (long)(4 * b) * ld), v);
      }
    }
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
    if (stamps != nullptr && threadIdx.x == 0) {
      stamps[tile * 5 + 0] = t0; stamps[tile * 5 + 1] = t1; stamps[tile * 5 + 2] = t2; stamps[tile * 5 + 3] = t3; stamps[tile * 5 + 4] = t4;
    }
  }
}

// ---- V3: contiguous per-agent records [env][odim] (what a layout change would buy): flat float4 stream ---------------
template <int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void obs_aos(float *__restrict__ obs, long B, long ntiles) {
  constexpr int TILE = 64;
  extern __shared__ float lds[];
  for (int x = threadIdx.x; x < (2 * E + 2 * A) * TILE; x += THREADS) lds[x] = (float)(x & 1023) * 0.001f;
  __syncthreads();
  constexpr int NW = THREADS / 64;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int i = w; i < A; i += NW) {
      const int odim = i < NADV ? ODIM_ADV : ODIM_GOOD;
      f4 *o = reinterpret_cast<f4 *>(obs + (long)obs_row0(i) * B + tile * TILE * odim);   // agent block, then env-major
      const int n4 = TILE * odim / 4;
      for (int f = lane; f < n4; f += 64) {
        f4 v = *reinterpret_cast<const f4 *>(lds + ((f * 4) & 16383));
        st<NT>(o + f, v);
      }
    }
  }
}

template <typename F>
static float time_it(F f, int iters = 5) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

static void report(const char *name, float ms, long B) {
  const double bytes = (double)OBS_TOTAL * B * 4;
  printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const long B = 1 << 18, ld = B + 256;
  float *obs;
  CK(hipMalloc(&obs, (size_t)OBS_TOTAL * ld * 4));
  CK(hipMemset(obs, 0, (size_t)OBS_TOTAL * ld * 4));
  // plan for V2
  constexpr int NB = (ODIM_ADV + 3) / 4;
  unsigned *hplan = (unsigned *)malloc(A * NB * 4 * sizeof(unsigned)), *dplan;
  for (int i = 0; i < A; ++i) {
    const int odim = i < NADV ? ODIM_ADV : ODIM_GOOD;
    for (int b = 0; b < NB; ++b)
      for (int r = 0; r < 4; ++r) {
        const int k = 4 * b + r;
        unsigned e = 0;
        if (k < odim) {
          int kk = k;
          bool rel = false, vel = false;
          int row;
          if (kk < 2) { vel = true; row = 2 * i + kk; }
          else if (kk < 4) row = 2 * i + kk - 2;
          else if ((kk -= 4) < 2 * L) { rel = true; row = 2 * A + kk; }
          else if ((kk -= 2 * L) < 2 * (A - 1)) { rel = true; row = kk < 2 * i ? kk : kk + 2; }
          else { kk -= 2 * (A - 1); vel = true; int j = NADV + (kk >> 1); if (i >= NADV && j >= i) ++j; row = 2 * j + (kk & 1); }
          e = (unsigned)((vel ? 2 * E + row : row) * 64) | (rel ? 0x40000000u : 0u) | 0x80000000u;
        }
        hplan[(i * NB + b) * 4 + r] = e;
      }
  }
  CK(hipMalloc(&dplan, A * NB * 4 * sizeof(unsigned)));
  CK(hipMemcpy(dplan, hplan, A * NB * 4 * sizeof(unsigned), hipMemcpyHostToDevice));

  const size_t lds64 = (size_t)(2 * E + 2 * A) * 64 * 4;
#define SETLDS(k, n) CK(hipFuncSetAttribute((const void *)(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(n)))
  const int citers = argc > 1 ? atoi(argv[1]) : 0;
  const long nt = B / 64;
  char name[128];
#define RUN_DWORD(THREADS, NTS, UNR, GRID)                                                                            \
  {                                                                                                                  \
    auto k = obs_dword<64, THREADS, NTS, UNR>; SETLDS(k, lds64);                                                     \
    for (int ds = 0; ds < 2; ++ds) {                                                                                 \
      if (!ds && !citers) continue;                                                                                  \
      snprintf(name, sizeof name, "V0 dword %dthr unroll%d grid %ld %s", THREADS, UNR, (long)(GRID), ds ? "compute+store" : "compute only"); \
      report(name, time_it([&] { hipLaunchKernelGGL(k, dim3(GRID), dim3(THREADS), lds64, 0, obs, ld, nt, citers, ds); }), B); \
    }                                                                                                                \
  }
#define RUN_PLAN(THREADS, NTS, UNR, GRID)                                                                             \
  {                                                                                                                  \
    auto k = obs_x4_plan<THREADS, NTS, UNR>; SETLDS(k, lds64);                                                       \
    for (int ds = 0; ds < 2; ++ds) {                                                                                 \
      if (!ds && !citers) continue;                                                                                  \
      snprintf(name, sizeof name, "V2 x4+plan %dthr unroll%d grid %ld %s", THREADS, UNR, (long)(GRID), ds ? "compute+store" : "compute only"); \
      report(name, time_it([&] { hipLaunchKernelGGL(k, dim3(GRID), dim3(THREADS), lds64, 0, obs, ld, nt, dplan, citers, ds); }), B); \
    }                                                                                                                \
  }
  float *in;
  constexpr int NIN = 640;
  unsigned long long *dstamps, *hstamps = (unsigned long long *)malloc(nt * 5 * 8);
  CK(hipMalloc(&dstamps, nt * 5 * 8));
  CK(hipMalloc(&in, (size_t)NIN * ld * 4));
  CK(hipMemset(in, 0, (size_t)NIN * ld * 4));
#define RUN_LOAD(NTS, PRE, GRID)                                                                                       \
  {                                                                                                                  \
    auto k = obs_x4_load<512, NTS, 4, PRE, NIN>; SETLDS(k, lds64);                                                   \
    snprintf(name, sizeof name, "V4 x4+plan+%d input rows %s %s grid %ld", NIN, NTS ? "nt" : "plain", PRE ? "prefetch" : "", (long)(GRID)); \
    report(name, time_it([&] { hipLaunchKernelGGL(k, dim3(GRID), dim3(512), lds64, 0, obs, ld, nt, dplan, in, citers, (unsigned long long *)nullptr); }), B); \
    hipLaunchKernelGGL(k, dim3(GRID), dim3(512), lds64, 0, obs, ld, nt, dplan, in, citers, dstamps);                \
    CK(hipMemcpy(hstamps, dstamps, nt * 5 * 8, hipMemcpyDeviceToHost));                                             \
    {                                                                                                                \
      double seg[4] = {0, 0, 0, 0}; unsigned long long lo = ~0ull, hi = 0;                                          \
      for (long t = 0; t < nt; ++t) {                                                                                \
        for (int j = 0; j < 4; ++j) seg[j] += (double)(hstamps[t * 5 + j + 1] - hstamps[t * 5 + j]);                 \
        if (hstamps[t * 5] < lo) lo = hstamps[t * 5];                                                                \
        if (hstamps[t * 5 + 4] > hi) hi = hstamps[t * 5 + 4];                                                        \
      }                                                                                                              \
      printf("    per tile (wave 0), us: load+barrier %.1f  compute %.1f  store issue %.1f  drain %.1f   | first start -> last end %.3f ms\n", \
             seg[0] / nt / 100, seg[1] / nt / 100, seg[2] / nt / 100, seg[3] / nt / 100, (double)(hi - lo) / 1e5);   \
    }                                                                                                                \
  }
  printf("synthetic compute phase: %d trips of 16 FMAs per lane per tile\n", citers);
  RUN_LOAD(true, false, nt)
  RUN_LOAD(false, false, nt)
  RUN_LOAD(true, false, 512)
  RUN_DWORD(512, true, 1, nt)
  RUN_DWORD(512, true, 8, nt)
  RUN_PLAN(512, true, 1, nt)
  RUN_PLAN(512, true, 4, nt)
  RUN_PLAN(512, true, 4, 512)
  RUN_PLAN(256, true, 4, nt)
  return 0;
}
