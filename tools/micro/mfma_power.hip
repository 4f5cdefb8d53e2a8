// Power probe: run ONE matrix-core instruction mix for a few seconds so that rocm-smi can sample socket power and
// the shader clock (tools/power_probe.sh).   mfma_power <mode> [seconds]
//   0: v_mfma_f32_16x16x4_f32, registers only        1: same, B operand re-read from LDS (1 ds_read_b128 / 8 MFMAs)
//   2: v_mfma_f32_4x4x1_16b_f32, registers only      3: same, B from LDS (1 ds_read_b128 / 8 MFMAs), A broadcast (cbsz)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  extern __shared__ float smem[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) {
    unsigned h = (i + blockIdx.x * 977) * 2654435761u; h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    smem[i] = (float)(int)(h & 0xFFFF) * 3.0e-5f - 1.0f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const f32x4* sx = reinterpret_cast<const f32x4*>(smem) + lane;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  f32x4 w = sx[64 * 40], xr = sx[64 * 41];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      f32x4 x = xr;
      if (MODE & 1) x = sx[((it + t) & 31) * 64];
      if (MODE < 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], x[j], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3 - j], x[j], a1, 0, 0, 0);
        }
      } else {
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0], x[0], a0, 4, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0], x[0], a1, 4, 8, 0);
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0], x[1], a0, 4, 1, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0], x[1], a1, 4, 9, 0);
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0], x[2], a0, 4, 2, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0], x[2], a1, 4, 10, 0);
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0], x[3], a0, 4, 3, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0], x[3], a1, 4, 11, 0);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3];
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 9.0;
  float* out; CK(hipMalloc(&out, 1024 * 512 * 4));
  const int iters = 20000;
  auto launch = [&] {
    switch (mode) {
      case 0: k<0><<<512, 512, 65536>>>(out, iters); break;
      case 1: k<1><<<512, 512, 65536>>>(out, iters); break;
      case 2: k<2><<<512, 512, 65536>>>(out, iters); break;
      default: k<3><<<512, 512, 65536>>>(out, iters); break;
    }
  };
  const double flops_per_launch = 512.0 * 8 * iters * 16 * 8 * (mode < 2 ? 2048.0 : 512.0);
  auto t0 = std::chrono::steady_clock::now();
  int n = 0;
  double el = 0;
  while (el < secs) {
    for (int i = 0; i < 4; ++i) launch();
    CK(hipDeviceSynchronize());
    n += 4;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  printf("mode %d: %d launches in %.2f s -> %.1f TFLOP/s\n", mode, n, el, flops_per_launch * n / el / 1e12);
  return 0;
}
