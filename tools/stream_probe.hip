// Access-pattern probe for the small-world env-step kernel: NR read rows + NW write rows of an
// SoA [row][B] layout, one (or V) env(s) per lane, trivial arithmetic.  Tells what HBM rate the
// access pattern itself can reach on MI355X, independent of the physics.
//   hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o gpurun_out/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int V> struct Vec;
template <> struct Vec<1> { using type = float; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <> struct Vec<2> { using type = f32x2; };
template <> struct Vec<4> { using type = f32x4; };

template <int NR, int NW, int V, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void soa_kernel(const float *__restrict__ in, float *__restrict__ out, long B) {
  using VT = typename Vec<V>::type;
  const long i = ((long)blockIdx.x * THREADS + threadIdx.x) * V;
  if (i >= B) return;
  float acc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) acc[v] = 0.f;
  float r[NR][V];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    VT x = *reinterpret_cast<const VT *>(in + (long)k * B + i);
#pragma unroll
    for (int v = 0; v < V; ++v) r[k][v] = reinterpret_cast<float *>(&x)[v];
  }
#pragma unroll
  for (int k = 0; k < NR; ++k)
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] += r[k][v];
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    VT y;
#pragma unroll
    for (int v = 0; v < V; ++v) reinterpret_cast<float *>(&y)[v] = acc[v] + r[k % NR][v] * (float)k;
    VT *p = reinterpret_cast<VT *>(out + (long)k * B + i);
    if (NT) __builtin_nontemporal_store(y, p); else *p = y;
  }
}

// Same rows through a raw buffer resource (one VGPR lane offset + one SGPR row offset per access)
template <int NR, int NW, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void soa_buf_kernel(const float *__restrict__ in, float *__restrict__ out, long B) {
  const long blk0 = (long)blockIdx.x * THREADS;
  const bool active = blk0 + threadIdx.x < B;
  const int lo = active ? (int)(threadIdx.x * 4) : -1;
  __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in + blk0), 0, -1, 0x00020000);
  __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + blk0, 0, -1, 0x00020000);
  const unsigned rb = (unsigned)B * 4u;
  float acc = 0.f;
  float r[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) r[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, lo, (int)(k * rb), 0));
#pragma unroll
  for (int k = 0; k < NR; ++k) acc += r[k];
#pragma unroll
  for (int k = 0; k < NW; ++k)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc + r[k % NR] * (float)k), ro, lo, (int)(k * rb), NT ? 2 : 0);
}

__global__ void copy4(const float4 *__restrict__ in, float4 *__restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

template <bool NT>
__global__ void copy4v(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    f32x4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
    if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
  }
}
__global__ void fill4(f32x4 *__restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (; i < n; i += stride) __builtin_nontemporal_store(v, out + i);
}
__global__ void read4(const f32x4 *__restrict__ in, float *__restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i < n; i += stride) acc += __builtin_nontemporal_load(in + i);
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

template <typename F>
static float time_it(F f, int iters = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

template <int NR, int NW, bool NT, int THREADS>
static void run_buf(const char *name, const float *in, float *out, long B) {
  dim3 grid((unsigned)((B + THREADS - 1) / THREADS));
  float ms = time_it([&] { hipLaunchKernelGGL((soa_buf_kernel<NR, NW, NT, THREADS>), grid, dim3(THREADS), 0, 0, in, out, B); });
  double bytes = (double)(NR + NW) * B * 4;
  printf("%-34s NR=%3d NW=%3d V=1 NT=%d T=%4d : %8.1f us  %7.1f GB/s\n", name, NR, NW, (int)NT, THREADS, ms * 1e3, bytes / ms / 1e6);
}

template <int NR, int NW, int V, bool NT, int THREADS>
static void run(const char *name, const float *in, float *out, long B) {
  const long nthreads = B / V;
  dim3 grid((unsigned)((nthreads + THREADS - 1) / THREADS));
  float ms = time_it([&] { hipLaunchKernelGGL((soa_kernel<NR, NW, V, NT, THREADS>), grid, dim3(THREADS), 0, 0, in, out, B); });
  double bytes = (double)(NR + NW) * B * 4;
  printf("%-34s NR=%3d NW=%3d V=%d NT=%d T=%4d : %8.1f us  %7.1f GB/s\n", name, NR, NW, V, (int)NT, THREADS, ms * 1e3, bytes / ms / 1e6);
}

int main() {
  const long B = 1 << 20;
  float *in, *out;
  CK(hipMalloc(&in, (size_t)64 * B * 4));
  CK(hipMalloc(&out, (size_t)128 * B * 4));
  CK(hipMemset(in, 0, (size_t)64 * B * 4));
  CK(hipMemset(out, 0, (size_t)128 * B * 4));
  {
    const long n = (long)40 * B / 4;  // 160 MB read + 160 MB write
    float ms = time_it([&] { hipLaunchKernelGGL(copy4, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)in, (float4 *)out, n); });
    printf("%-34s : %8.1f us  %7.1f GB/s\n", "copy float4 160MB+160MB", ms * 1e3, 2.0 * n * 16 / ms / 1e6);
  }
  {
    const long n = (long)256 * 1024 * 1024 / 16;  // 256 MiB each way (beyond the Infinity Cache together)
    float *big_in, *big_out;
    CK(hipMalloc(&big_in, n * 16)); CK(hipMalloc(&big_out, n * 16));
    CK(hipMemset(big_in, 0, n * 16));
    for (int g : {1024, 2048, 4096, 8192, 16384, 65536}) {
      float ms = time_it([&] { hipLaunchKernelGGL(copy4v<false>, dim3(g), dim3(256), 0, 0, (const f32x4 *)big_in, (f32x4 *)big_out, n); });
      float ms2 = time_it([&] { hipLaunchKernelGGL(copy4v<true>, dim3(g), dim3(256), 0, 0, (const f32x4 *)big_in, (f32x4 *)big_out, n); });
      printf("copy 256MiB+256MiB grid=%6d : %7.1f GB/s   nt: %7.1f GB/s\n", g, 2.0 * n * 16 / ms / 1e6, 2.0 * n * 16 / ms2 / 1e6);
    }
    for (int g : {2048, 8192, 65536}) {
      float ms = time_it([&] { hipLaunchKernelGGL(fill4, dim3(g), dim3(256), 0, 0, (f32x4 *)big_out, n); });
      float ms2 = time_it([&] { hipLaunchKernelGGL(read4, dim3(g), dim3(256), 0, 0, (const f32x4 *)big_in, out, n); });
      printf("write-only 256MiB grid=%6d : %7.1f GB/s   read-only: %7.1f GB/s\n", g, 1.0 * n * 16 / ms / 1e6, 1.0 * n * 16 / ms2 / 1e6);
    }
    hipFree(big_in); hipFree(big_out);
  }
  run<40, 86, 1, false, 256>("soa tag-shaped", in, out, B);
  run<40, 86, 1, true, 256>("soa tag-shaped nt", in, out, B);
  run_buf<40, 86, false, 256>("soa tag-shaped BUFFER", in, out, B);
  run_buf<40, 86, true, 256>("soa tag-shaped BUFFER nt", in, out, B);
  run_buf<40, 86, true, 1024>("soa tag-shaped BUFFER nt", in, out, B);
  run<40, 86, 1, true, 1024>("soa tag-shaped nt T1024", in, out, B);
  run<40, 86, 2, false, 256>("soa tag-shaped x2", in, out, B);
  run<40, 86, 2, true, 256>("soa tag-shaped x2 nt", in, out, B);
  run<40, 86, 4, false, 256>("soa tag-shaped x4", in, out, B);
  run<40, 86, 4, true, 256>("soa tag-shaped x4 nt", in, out, B);
  run<40, 86, 1, false, 64>("soa tag-shaped T64", in, out, B);
  run<40, 86, 1, false, 128>("soa tag-shaped T128", in, out, B);
  run<40, 86, 1, false, 512>("soa tag-shaped T512", in, out, B);
  run<40, 86, 1, false, 1024>("soa tag-shaped T1024", in, out, B);
  run<40, 40, 1, false, 256>("soa 40r/40w", in, out, B);
  run<40, 40, 4, false, 256>("soa 40r/40w x4", in, out, B);
  run<8, 8, 1, false, 256>("soa 8r/8w", in, out, B);
  run<8, 8, 4, false, 256>("soa 8r/8w x4", in, out, B);
  run<2, 2, 4, false, 256>("soa 2r/2w x4", in, out, B);
  return 0;
}
