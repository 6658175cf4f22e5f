// How does a scalar load treat a base address that is not a multiple of 4?  (The compiler forms such a base when a byte
// array and a 4-byte array of a kernel-argument struct are indexed by the same scalar i: base = kernarg + i for the byte,
// then s_load_dword s, s[base], s[3 i] offset:... for the word.)  Table t[k] = 100 + k; for i = 0..3 load with
// base = t + i bytes, register offset 3 i, immediate 0: the sum is t + 4 i, entry i.
//   hipcc --offload-arch=gfx950 -O2 -o smem_align_probe tools/smem_align_probe.hip && ./smem_align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(const uint32_t *t, uint32_t *out) {
  for (int i = 0; i < 4; ++i) {
    const uint64_t base = reinterpret_cast<uint64_t>(t) + (uint64_t)i;
    const uint32_t off = 3u * (uint32_t)i;
    uint32_t both, imm_only;
    asm volatile("s_load_dword %0, %1, %2 offset:0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(both) : "s"(base), "s"(off) : "memory");
    asm volatile("s_load_dword %0, %1, 0x8\n\ts_waitcnt lgkmcnt(0)" : "=s"(imm_only) : "s"(base) : "memory");
    if (threadIdx.x == 0) {
      out[2 * i] = both;          // wanted: entry i
      out[2 * i + 1] = imm_only;  // base + i + 8: entry 2 if the low bits of the SUM are dropped
    }
  }
}

int main() {
  uint32_t h[16], *d, *o, r[8];
  for (int k = 0; k < 16; ++k) h[k] = 100 + k;
  if (hipMalloc(&d, sizeof h) != hipSuccess || hipMalloc(&o, sizeof r) != hipSuccess) return 2;
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipMemset(o, 0, sizeof r);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
  if (hipDeviceSynchronize() != hipSuccess) return 3;
  hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i)
    printf("base = t + %d B: [sgpr offset %d + imm 0] -> %u (entry %d wanted)   [imm 8] -> %u\n", i, 3 * i, r[2 * i], i, r[2 * i + 1]);
  return 0;
}
