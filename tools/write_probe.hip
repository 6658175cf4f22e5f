// Why do the scaled world's observation stores run at 5.4-5.6 TB/s when a 15 GB torch fill_ runs at 6.9 TB/s?
// Separates the two suspects: occupancy (the tile kernel's 80 KiB of LDS allow 2 workgroups = 16 wavefronts per CU) and
// pattern (256-byte row segments, rows 1 MiB apart, vs one moving contiguous front).
//   hipcc -O3 --offload-arch=gfx950 tools/write_probe.hip -o tools/write_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr long ROWS = 14432, B = 1 << 18, LD = B + 256;

template <bool NT> __device__ __forceinline__ void st(f4 *p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// contiguous front: block b writes bytes [b * chunk, (b + 1) * chunk), 16 B per lane per store
template <bool NT>
__global__ __launch_bounds__(256) void fill_front(f4 *out, long n4, int per_thread) {
  extern __shared__ float lds[];
  const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  long i = (long)blockIdx.x * 256 * per_thread + threadIdx.x;
  for (int k = 0; k < per_thread; ++k, i += 256)
    if (i < n4) st<NT>(out + i, v);
}

// tile pattern: workgroup = tile of SEG/4 envs... lane (r = lane / (SEG / 16), q) writes 16 B of row k + r; a wavefront
// covers 1024 / SEG rows per store; wavefront w of the workgroup takes rows w, w + NW, ... in units of that many rows
template <bool NT, int SEG /* bytes of one row segment per workgroup: 256, 512, 1024 */, int THREADS>
__global__ __launch_bounds__(THREADS) void tile_rows(float *out, long ld, long rows) {
  extern __shared__ float lds[];
  constexpr int LPR = SEG / 16;          // lanes per row segment
  constexpr int RPW = 64 / LPR;          // rows per wave-instruction
  constexpr int NW = THREADS / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = lane % LPR, r = lane / LPR;
  const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  float *base = out + (long)blockIdx.x * (SEG / 4) + 4 * q;
  for (long k = (long)w * RPW; k < rows; k += (long)NW * RPW)
    if (k + r < rows) st<NT>(reinterpret_cast<f4 *>(base + (k + r) * ld), v);
}

// env-major records: agent i's observations are [B][odim] contiguous; a workgroup (tile of 64 envs) writes, agent after
// agent, its 64 x odim x 4 B = 57.8 KB chunk of that agent's block, all wavefronts side by side (a compact moving window
// of THREADS x 16 B), or (COOP = false) wavefront w writes the chunks of agents w, w + NW, ... on its own
template <bool NT, int THREADS, bool COOP>
__global__ __launch_bounds__(THREADS) void records(float *out, long nb) {
  extern __shared__ float lds[];
  constexpr int NW = THREADS / 64, A = 64, ODIM = 226;   // (good agents: 224; 226 for all here)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  const long chunk4 = 64L * ODIM / 4;                    // float4 per (tile, agent) chunk
  for (int i = COOP ? 0 : w; i < A; i += COOP ? 1 : NW) {
    f4 *o = reinterpret_cast<f4 *>(out) + ((long)i * nb + blockIdx.x) * chunk4;
    for (long f = COOP ? threadIdx.x : lane; f < chunk4; f += COOP ? THREADS : 64) st<NT>(o + f, v);
  }
}

template <typename F>
static float time_it(F f, int iters = 5) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f();
  (void)hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}
static void report(const char *name, float ms) {
  printf("%-72s %8.3f ms  %7.1f GB/s\n", name, ms, (double)ROWS * B * 4 / ms / 1e6);
  fflush(stdout);
}
#define SETLDS(k, n) CK(hipFuncSetAttribute((const void *)(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(n)))

int main() {
  float *out;
  CK(hipMalloc(&out, (size_t)(ROWS + 256) * LD * 4));   // + room for the records variant (64 agents x 226 columns)
  CK(hipMemset(out, 0, (size_t)ROWS * LD * 4));
  const long n4 = ROWS * B / 4;
  char name[160];
  for (int lds : {0, 40 * 1024, 80 * 1024})
    for (int pt : {4, 16, 64}) {
      auto k = fill_front<false>; SETLDS(k, 80 * 1024);
      const long blocks = (n4 + 256L * pt - 1) / (256L * pt);
      snprintf(name, sizeof name, "contiguous front, plain, 256 thr, %2d x 16 B per lane, dynamic LDS %5d B", pt, lds);
      report(name, time_it([&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, (f4 *)out, n4, pt); }));
    }
  {
    auto k = fill_front<true>; SETLDS(k, 80 * 1024);
    report("contiguous front, nt, 256 thr, 4 x 16 B per lane, no LDS", time_it([&] { hipLaunchKernelGGL(k, dim3((n4 + 1023) / 1024), dim3(256), 0, 0, (f4 *)out, n4, 4); }));
  }
#define TILE_RUN(NTS, SEG, THREADS, LDS)                                                                            \
  {                                                                                                                 \
    auto k = tile_rows<NTS, SEG, THREADS>; SETLDS(k, 160 * 1024);                                                  \
    snprintf(name, sizeof name, "row segments of %4d B, rows 1 MiB apart, %s, %4d thr, dynamic LDS %6d B", SEG, NTS ? "nt" : "plain", THREADS, LDS); \
    report(name, time_it([&] { hipLaunchKernelGGL(k, dim3(B * 4 / SEG), dim3(THREADS), LDS, 0, out, LD, ROWS); })); \
  }
#define REC_RUN(NTS, THREADS, COOP, LDS)                                                                           \
  {                                                                                                                 \
    auto k = records<NTS, THREADS, COOP>; SETLDS(k, 160 * 1024);                                                   \
    snprintf(name, sizeof name, "env-major records, 57.8 KB chunks, %s, %s, %4d thr, dynamic LDS %6d B", COOP ? "workgroup-cooperative" : "one agent per wavefront", NTS ? "nt" : "plain", THREADS, LDS); \
    report(name, time_it([&] { hipLaunchKernelGGL(k, dim3(B / 64), dim3(THREADS), LDS, 0, out, B / 64); }));        \
  }
  REC_RUN(false, 512, false, 80 * 1024)
  REC_RUN(false, 512, true, 80 * 1024)
  REC_RUN(true, 512, true, 80 * 1024)
  REC_RUN(false, 512, true, 56 * 1024)
  REC_RUN(false, 256, true, 40 * 1024)
  REC_RUN(false, 1024, true, 80 * 1024)
  // row stride: B + pad floats (the world's default is 1 KiB past a multiple of 4 KiB = pad 256 floats)
  for (long padb : {0L, 256L, 1024L, 4096L + 1024, 16384L + 1024, 65536L + 1024, 131072L + 1024, 262144L + 1024, 524288L + 1024}) {
    const long ldp = B + padb / 4;
    if ((size_t)ROWS * ldp > (size_t)(ROWS + 256) * LD + 0) {
      // buffer holds (ROWS + 256) * LD floats: fewer rows for the big pads
    }
    const long rows = ((ROWS + 256) * LD) / ldp < ROWS ? ((ROWS + 256) * LD) / ldp : ROWS;
    auto k = tile_rows<false, 256, 512>; SETLDS(k, 160 * 1024);
    float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(B * 4 / 256), dim3(512), 80 * 1024, 0, out, ldp, rows); });
    printf("row segments of 256 B, 512 thr, LDS 80 KiB, row stride 1 MiB + %7ld B, %5ld rows: %8.3f ms  %7.1f GB/s\n", padb, rows, ms,
           (double)rows * B * 4 / ms / 1e6);
    fflush(stdout);
  }
  TILE_RUN(false, 256, 256, 0)
  TILE_RUN(true, 256, 256, 0)
  TILE_RUN(false, 256, 512, 0)
  TILE_RUN(false, 256, 512, 40 * 1024)
  TILE_RUN(false, 256, 512, 80 * 1024)
  TILE_RUN(true, 256, 512, 80 * 1024)
  TILE_RUN(false, 512, 512, 0)
  TILE_RUN(false, 512, 512, 80 * 1024)
  TILE_RUN(false, 1024, 256, 0)
  TILE_RUN(true, 1024, 256, 0)
  TILE_RUN(false, 1024, 512, 0)
  TILE_RUN(false, 1024, 1024, 0)
  TILE_RUN(false, 1024, 1024, 160 * 1024)
  return 0;
}
