// Access-pattern probe #2: the simple_tag row shape (40 read rows, 86 write rows, 1Mi envs) with a
// dependent-VALU "compute" phase between the loads and the stores, to study how occupancy, a
// persistent grid and software prefetch of the next tile interact with the memory system.
//   hipcc -O3 --offload-arch=gfx950 tools/pipeline_probe.hip -o tools/variants/pipeline_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NR = 40, NW = 86;

template <int DELAY>
__device__ __forceinline__ float burn(float acc) {
#pragma unroll 16
  for (int i = 0; i < DELAY; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
  return acc;
}

// one tile (256 envs) per workgroup, like the shipped kernel
template <int DELAY, int CAP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, CAP))) void plain(const float *__restrict__ in, float *__restrict__ out, long B) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float r[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) r[k] = k < 20 ? __builtin_nontemporal_load(in + (long)k * B + i) : in[(long)k * B + i];
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < NR; ++k) acc += r[k];
  acc = burn<DELAY>(acc);
#pragma unroll
  for (int k = 0; k < NW; ++k) __builtin_nontemporal_store(acc + r[k % NR] * (float)k, out + (long)k * B + i);
}

// persistent grid: each workgroup strides over tiles; the next tile's loads are issued before the
// current tile's compute + stores
template <int DELAY, int CAP, bool PREFETCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, CAP))) void persistent(const float *__restrict__ in, float *__restrict__ out, long B) {
  const long ntiles = B / 256;
  long tile = blockIdx.x;
  float r[NR], nx[NR];
  if (tile < ntiles) {
    const long i = tile * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < NR; ++k) r[k] = k < 20 ? __builtin_nontemporal_load(in + (long)k * B + i) : in[(long)k * B + i];
  }
  for (; tile < ntiles; tile += gridDim.x) {
    const long i = tile * 256 + threadIdx.x;
    const long nt = tile + gridDim.x;
    if (PREFETCH && nt < ntiles) {
      const long j = nt * 256 + threadIdx.x;
#pragma unroll
      for (int k = 0; k < NR; ++k) nx[k] = k < 20 ? __builtin_nontemporal_load(in + (long)k * B + j) : in[(long)k * B + j];
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NR; ++k) acc += r[k];
    acc = burn<DELAY>(acc);
#pragma unroll
    for (int k = 0; k < NW; ++k) __builtin_nontemporal_store(acc + r[k % NR] * (float)k, out + (long)k * B + i);
    if (PREFETCH) {
#pragma unroll
      for (int k = 0; k < NR; ++k) r[k] = nx[k];
    } else if (nt < ntiles) {
      const long j = nt * 256 + threadIdx.x;
#pragma unroll
      for (int k = 0; k < NR; ++k) r[k] = k < 20 ? __builtin_nontemporal_load(in + (long)k * B + j) : in[(long)k * B + j];
    }
  }
}

template <typename F>
static float time_it(F f, int iters = 30) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}
static void report(const char *name, float ms, long B) {
  printf("%-44s : %7.1f us  %6.0f GB/s\n", name, ms * 1e3, (double)(NR + NW) * B * 4 / ms / 1e6);
}

int main() {
  const long B = 1 << 20;
  float *in, *out;
  CK(hipMalloc(&in, (size_t)NR * B * 4)); CK(hipMalloc(&out, (size_t)NW * B * 4));
  CK(hipMemset(in, 0, (size_t)NR * B * 4)); CK(hipMemset(out, 0, (size_t)NW * B * 4));
  const dim3 g((unsigned)(B / 256)), b(256);
#define PLAIN(D, C) report("plain delay=" #D " cap=" #C, time_it([&] { hipLaunchKernelGGL((plain<D, C>), g, b, 0, 0, in, out, B); }), B)
  PLAIN(0, 8); PLAIN(0, 3); PLAIN(300, 8); PLAIN(300, 3); PLAIN(700, 8); PLAIN(700, 5); PLAIN(700, 3); PLAIN(700, 2); PLAIN(1500, 8); PLAIN(1500, 3);
#define PERS(D, C, P, G) report("persistent delay=" #D " cap=" #C " prefetch=" #P " grid=" #G, time_it([&] { hipLaunchKernelGGL((persistent<D, C, P>), dim3(G), b, 0, 0, in, out, B); }), B)
  PERS(700, 3, false, 768); PERS(700, 3, true, 768); PERS(700, 2, true, 512); PERS(700, 3, true, 1536); PERS(700, 4, true, 1024); PERS(700, 2, true, 1024);
  PERS(0, 3, true, 768); PERS(1500, 3, true, 768); PERS(1500, 2, true, 512);
  return 0;
}
