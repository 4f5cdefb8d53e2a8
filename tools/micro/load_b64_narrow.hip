#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, _Float16* out) {
  extern __shared__ _Float16 s[];
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 0x1000, 0x00020000);
  u32x2 ibuf[2];
  for (int i = 0; i < 2; ++i) ibuf[i] = __builtin_amdgcn_raw_buffer_load_b64(srd, threadIdx.x * 8 + i * 1024, 0, 0);
  __syncthreads();
  for (int i = 0; i < 2; ++i) *reinterpret_cast<f16x4*>(s + threadIdx.x * 4 + i * 1024) = __builtin_bit_cast(f16x4, ibuf[i]);
  __syncthreads();
  out[threadIdx.x] = s[threadIdx.x * 3];
}
