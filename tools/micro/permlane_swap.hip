// v_permlane16_swap_b32 (gfx950): x + (x of the lane 16 further, xor 16) without the LDS-queue ds_bpermute of __shfl_xor.
// After the swap of two copies, a = [row0, row0, row2, row2], b = [row1, row1, row3, row3] (rows of 16 lanes).
// build: hipcc -O3 --offload-arch=gfx950 permlane_swap.hip -o permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, float* o1, float* o2) {
  const float x = in[threadIdx.x];
  o1[threadIdx.x] = x + __shfl_xor(x, 16, 64);
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  o2[threadIdx.x] = a + b;
}
int main() {
  float h[64], r1[64], r2[64], *d, *a, *b;
  for (int i = 0; i < 64; ++i) h[i] = 1.0f + i * 0.37f;
  (void)hipMalloc(&d, 256); (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256);
  (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, a, b);
  (void)hipMemcpy(r1, a, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(r2, b, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) bad += r1[i] != r2[i];
  printf("mismatches %d  (lane 0: %g vs %g, lane 40: %g vs %g)\n", bad, r1[0], r2[0], r1[40], r2[40]);
  return bad != 0;
}
