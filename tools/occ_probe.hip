// How many 512-thread workgroups of a given dynamic-LDS size does one MI355X CU really hold?  (API answer + a census:
// every workgroup records its CU-slot id (XCC, SE, CU) and start/end times; resident = max overlap per CU.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void census(unsigned long long *out, int spin) {
  extern __shared__ float lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  float a = lds[(threadIdx.x + 1) & 511];
  for (int i = 0; i < spin; ++i) a = __builtin_fmaf(a, 1.0001f, 0.5f);
  if (a == 123.456f) out[0] = 1;
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11));     // HW_REG_XCC_ID [3:0]
    unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((32 - 1) << 11));    // HW_REG_HW_ID
    out[blockIdx.x * 3 + 0] = ((unsigned long long)xcc << 32) | hwid;
    out[blockIdx.x * 3 + 1] = t0;
    out[blockIdx.x * 3 + 2] = t1;
  }
}

int main() {
  const int nb = 4096;
  unsigned long long *d, *h = (unsigned long long *)malloc(nb * 3 * 8);
  CK(hipMalloc(&d, nb * 3 * 8));
  for (int lds : {81920, 81408, 80896, 79872, 78848, 73728, 65536, 53248, 40960, 32768}) {
    CK(hipFuncSetAttribute((const void *)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int api = -1;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, census, 512, lds));
    hipLaunchKernelGGL(census, dim3(nb), dim3(512), lds, 0, d, 20000);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, nb * 3 * 8, hipMemcpyDeviceToHost));
    // per CU id (xcc, se, cu from HW_ID bits: cu_id [11:8], sh_id [12], se_id [15:13]) -> max concurrently resident blocks
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
    for (int b = 0; b < nb; ++b) {
      const unsigned long long id = h[b * 3], hw = id & 0xffffffffu;
      const unsigned long long cu = (id >> 32) << 16 | ((hw >> 8) & 0xff);   // xcc | cu_id, sh_id, se_id
      ev[cu].push_back({h[b * 3 + 1], +1});
      ev[cu].push_back({h[b * 3 + 2], -1});
    }
    int worst = 0, best = 1 << 30;
    for (auto &kv : ev) {
      std::sort(kv.second.begin(), kv.second.end());
      int cur = 0, mx = 0;
      for (auto &e : kv.second) { cur += e.second; mx = std::max(mx, cur); }
      worst = std::max(worst, mx); best = std::min(best, mx);
    }
    unsigned long long lo = ~0ull, hi = 0;
    for (int b = 0; b < nb; ++b) { lo = std::min(lo, h[b * 3 + 1]); hi = std::max(hi, h[b * 3 + 2]); }
    printf("dynamic LDS %6d B: occupancy API %d blocks/CU; census: %zu CU ids, max resident per CU %d..%d; span %.3f ms\n", lds, api,
           ev.size(), best, worst, (double)(hi - lo) / 1e5);
  }
  return 0;
}
