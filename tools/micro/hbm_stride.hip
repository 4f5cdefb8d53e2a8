// Do alternating 128-byte segments (the x-blocked concat buffer: 4 voxels x 8 channels of the "up" half, then 128 B of
// the skip half, written by two different kernels) cost HBM write / read bandwidth against a planar layout?
// Same bytes moved: 1.34 GB written (or read) either as every second 128-B segment of a 2.68 GB buffer, or contiguously.
// build: hipcc -O3 --offload-arch=gfx950 hbm_stride.hip -o hbm_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// i-th 16-byte piece of the moved data -> piece index in the buffer; seg = pieces per contiguous segment (8 = 128 B)
__device__ __forceinline__ size_t where(size_t i, int seg, int strided) { return strided ? (i / seg) * (2 * seg) + i % seg : i; }
__global__ __launch_bounds__(256) void fill(f32x4* p, size_t n, int seg, int strided) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[where(i, seg, strided)] = (f32x4){1.f, 2.f, 3.f, 4.f};
}
__global__ __launch_bounds__(256) void rd(const f32x4* a, float* out, size_t n, int seg, int strided) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[where(i, seg, strided)];
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}
int main() {
  const size_t bytes = 1342177280ull, n = bytes / 16;
  f32x4* a; float* o;
  CK(hipMalloc(&a, 2 * bytes)); CK(hipMalloc(&o, 4));
  CK(hipMemset(a, 0, 2 * bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int seg : {8, 16, 64}) {
    for (int strided = 0; strided < 2; ++strided) {
      for (int mode = 0; mode < 2; ++mode) {
        auto launch = [&] {
          if (mode == 0) fill<<<16384, 256>>>(a, n, seg, strided);
          else rd<<<16384, 256>>>(a, o, n, seg, strided);
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("segment %4d B %-10s %-5s %.3f ms  %.2f TB/s\n", seg * 16, strided ? "alternate" : "contiguous", mode ? "read" : "write", ms, bytes / (ms * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
