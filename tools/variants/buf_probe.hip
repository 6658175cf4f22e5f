#include <hip/hip_runtime.h>
__global__ void k(float* out, const float* in, long ld, int rows) {
  __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + blockIdx.x * 256, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)(in + blockIdx.x * 256), 0, 0x7fffffff, 0x00020000);
  int vo = threadIdx.x * 4;
  float v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, vo, (int)(r * ld * 4), 2));
#pragma unroll
  for (int r = 0; r < 8; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r] * 2.f), ro, vo, (int)(r * ld * 4), 2);
}
