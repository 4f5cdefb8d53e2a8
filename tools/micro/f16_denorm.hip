// Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs?  A = 2^-20 (subnormal), B = 2^10 -> 2^-10 per product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float av, float bv) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)av; b[j] = (_Float16)bv; }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  out[threadIdx.x] = acc[0];
  if (threadIdx.x == 0) out[64] = (float)a[0];
}
int main() {
  float* d; hipMalloc(&d, 65 * 4);
  float h[65];
  k<<<1, 64>>>(d, 9.5367431640625e-07f, 1024.f);   // 2^-20
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("cvt(2^-20) = %g (expect 9.53674e-07), sum of 32 products = %g (expect %g; 0 means inputs flushed)\n", h[64], h[0], 32 * 0.0009765625);
  k<<<1, 64>>>(d, 3.0e-6f, 1.0f);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("cvt(3e-6) = %g, sum = %g (expect ~%g)\n", h[64], h[0], 32 * 3.0e-6);
  return 0;
}
