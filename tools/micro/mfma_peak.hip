// Micro-benchmark: what does v_mfma_f32_16x16x4_f32 sustain on this part, and how does the conv inner-loop shape compare?
//   mode 0: pure MFMA chains (NACC independent accumulators per wave)
//   mode 1: the conv inner-loop shape: per "tap" 3 ds_read_b128 feeding 8 MFMAs on 2 accumulators
// Prints TFLOP/s from wall time and the per-wave MFMA issue interval from s_memtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_pure(float* out, int iters, unsigned long long* clk, int rnd) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float av[16], bv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    unsigned h = (threadIdx.x * 16 + k + blockIdx.x * 977) * 2654435761u; h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    av[k] = rnd ? (float)(int)(h & 0xFFFF) * 3.0e-5f - 1.0f : threadIdx.x * 1e-3f;
    bv[k] = rnd ? (float)(int)(h >> 16) * 3.0e-5f - 1.0f : 1.0f;
  }
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], bv[k], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; }
}

__global__ __launch_bounds__(512) void mfma_lds(float* out, int iters, unsigned long long* clk, int rnd) {
  extern __shared__ float smem[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) { unsigned h = (i + blockIdx.x * 977) * 2654435761u; h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    smem[i] = rnd ? (float)(int)(h & 0xFFFF) * 3.0e-5f - 1.0f : i * 1e-4f; }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const f32x4* sw = reinterpret_cast<const f32x4*>(smem) + lane;
  const f32x4* sx = reinterpret_cast<const f32x4*>(smem + 8192) + lane;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      f32x4 w = sw[tap * 64], x0 = sx[tap * 64], x1 = sx[tap * 64 + 17];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], x0[j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], x1[j], acc1, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static void run(const char* name, F launch, double flops) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  printf("%-34s %8.3f ms  %7.2f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  float* out; unsigned long long* clk;
  CK(hipMalloc(&out, 4096 * 1024 * 4)); CK(hipMalloc(&clk, 16));
  for (int rnd = 0; rnd < 2; ++rnd) {
    for (int nw : {4, 8}) {
      char nm[64];
      double fl = 256.0 * nw * iters * 16 * 4 * 2048.0;
      snprintf(nm, 64, "pure acc4 waves%d %s", nw, rnd ? "random" : "smooth");
      run(nm, [&] { mfma_pure<4><<<256, nw * 64>>>(out, iters, clk, rnd); }, fl);
      fl = 256.0 * nw * iters * 16 * 2 * 2048.0;
      snprintf(nm, 64, "pure acc2 waves%d %s", nw, rnd ? "random" : "smooth");
      run(nm, [&] { mfma_pure<2><<<256, nw * 64>>>(out, iters, clk, rnd); }, fl);
      fl = 256.0 * nw * iters * 64 * 2048.0;
      snprintf(nm, 64, "lds-fed waves%d %s", nw, rnd ? "random" : "smooth");
      run(nm, [&] { mfma_lds<<<256, nw * 64, 65536>>>(out, iters, clk, rnd); }, fl);
    }
  }
  return 0;
}
