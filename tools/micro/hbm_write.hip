// HBM write roof: plain 16-byte streaming stores (and a read+write copy) over 1.34 GB, the size of one full-resolution
// 8-channel activation tensor at 160 samples.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void fill(f32x4* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (f32x4){v, v, v, v};
}
__global__ __launch_bounds__(256) void fill_nt(f32x4* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store((f32x4){v, v, v, v}, p + i);
}
__global__ __launch_bounds__(256) void copy(const f32x4* a, f32x4* b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void rd(const f32x4* a, float* out, size_t n) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}
int main() {
  const size_t bytes = 1342177280ull, n = bytes / 16;
  f32x4 *a, *b; float* o;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {2048, 8192, 32768}) {
    for (int mode = 0; mode < 4; ++mode) {
      auto launch = [&] {
        if (mode == 0) fill<<<grid, 256>>>(a, n, 1.f);
        else if (mode == 1) fill_nt<<<grid, 256>>>(a, n, 1.f);
        else if (mode == 2) copy<<<grid, 256>>>(a, b, n);
        else rd<<<grid, 256>>>(a, o, n);
      };
      for (int i = 0; i < 3; ++i) launch();
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; ++i) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
      const double moved = (mode == 2 ? 2.0 : 1.0) * bytes;
      printf("grid %5d %-8s %.3f ms  %.2f TB/s\n", grid, mode == 0 ? "fill" : mode == 1 ? "fill_nt" : mode == 2 ? "copy" : "read", ms, moved / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
