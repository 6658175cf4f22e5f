// Accuracy of the gfx950 f64 reciprocal / reciprocal-square-root seeds (v_rcp_f64, v_rsq_f64): max relative error over
// 2^24 arguments in [0.5, 4).  Build: hipcc -O3 --offload-arch=gfx950 -o tools/rcp_probe.bin tools/rcp_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void probe(double *out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const double x = 0.5 + 3.5 * ((double)i + 0.37) / 16777216.0;
  const double r = __builtin_amdgcn_rcp(x), q = __builtin_amdgcn_rsq(x);
  const double er = fabs(fma(r, x, -1.0));          // |r x - 1|
  const double eq = fabs(fma(q * q, x, -1.0)) * 0.5;  // |q^2 x - 1| / 2 ~ relative error of q
  double m0 = er, m1 = eq;
  for (int off = 32; off >= 1; off >>= 1) {
    m0 = fmax(m0, __shfl_xor(m0, off, 64));
    m1 = fmax(m1, __shfl_xor(m1, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    out[2 * (i >> 6)] = m0;
    out[2 * (i >> 6) + 1] = m1;
  }
}
int main() {
  const int n = 1 << 24, waves = n / 64;
  double *d, *h = (double *)malloc(sizeof(double) * 2 * waves);
  if (hipMalloc(&d, sizeof(double) * 2 * waves) != hipSuccess) return 1;
  probe<<<n / 256, 256>>>(d);
  if (hipMemcpy(h, d, sizeof(double) * 2 * waves, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  double m0 = 0, m1 = 0;
  for (int i = 0; i < waves; ++i) { m0 = fmax(m0, h[2 * i]); m1 = fmax(m1, h[2 * i + 1]); }
  printf("v_rcp_f64 max rel err %.3e = 2^%.1f   v_rsq_f64 max rel err %.3e = 2^%.1f\n", m0, log2(m0), m1, log2(m1));
  hipFree(d);
  return 0;
}
