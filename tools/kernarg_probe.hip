// Does a by-value kernel argument larger than 4 KiB launch correctly on this stack?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { double v[N]; };
template <int N> __global__ void k(Big<N> b, double *out) { out[threadIdx.x] = b.v[threadIdx.x % N] + b.v[N - 1]; }
template <int N> int run() {
  Big<N> b; for (int i = 0; i < N; ++i) b.v[i] = i;
  double *d; hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
  hipError_t e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
  double h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("bytes=%d launch=%s sync=%s out[5]=%g (want %g)\n", N * 8, hipGetErrorString(e), hipGetErrorString(e2), h[5], 5.0 + N - 1);
  return 0;
}
int main() { run<500>(); run<512>(); run<600>(); run<1000>(); run<2000>(); return 0; }
