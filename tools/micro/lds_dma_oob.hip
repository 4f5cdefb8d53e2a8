// Micro-test: does a buffer_load ... lds (LDS-DMA) with an out-of-range voffset write ZEROS to LDS or leave it alone?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, float* out, int n_valid) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4];
  const int lane = threadIdx.x;
  for (int j = 0; j < 4; ++j) lds[lane * 4 + j] = -5.0f;   // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 0x80000000, 0x00020000);
  unsigned vo = lane < n_valid ? lane * 16u : 0xFFFFFFF0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)vo, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = lds[lane * 4 + j];
}
int main() {
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
  float *din, *dout;
  hipMalloc(&din, 1024); hipMalloc(&dout, 1024);
  hipMemcpy(din, h.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, 40);
  std::vector<float> o(256);
  hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost);
  printf("lane 0: %g %g | lane 39: %g %g | lane 40 (OOB): %g %g | lane 63 (OOB): %g %g\n", o[0], o[3], o[156], o[159], o[160], o[163], o[252], o[255]);
  return 0;
}
