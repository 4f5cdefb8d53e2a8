// Micro-test: layout and rate of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction).
//   layout: D_b[i][j] += A_b[i] * B_b[j];  which lane holds A_b[i], B_b[j], and where does D_b[i][j] land?
//   rate:   independent accumulator chains, 1 and 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void layout(float* out, int cbsz_mode) {
  const int l = threadIdx.x;
  // A value encodes its lane as 1000 + l, B value as (l + 1): the product identifies both source lanes
  f32x4 acc = {0, 0, 0, 0};
  if (cbsz_mode == 0) acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1000 + l), (float)(l + 1), acc, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1000 + l), (float)(l + 1), acc, 4, 5, 0);  // broadcast block 5's A
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(512) void rate(float* out, int iters) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float av[8], bv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { av[k] = (threadIdx.x * 8 + k) * 1e-4f - 0.2f; bv[k] = 0.5f - k * 0.1f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[k], bv[(k + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// conv-shaped loop: 27 taps x 4 channel quads; per quad one ds_read_b128 of the lane's voxel feeds 4 x 2 MFMAs whose A
// operand is a register holding 16 channels x 4 couts, broadcast with cbsz = 4 / abid = channel
template <int J> struct IC { static constexpr int v = J; };
template <int Q>
__device__ __forceinline__ void quad(f32x4& a0, f32x4& a1, float w0, float w1, f32x4 x) {
  a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w0, x[0], a0, 4, 4 * Q + 0, 0);
  a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w1, x[0], a1, 4, 4 * Q + 0, 0);
  a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w0, x[1], a0, 4, 4 * Q + 1, 0);
  a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w1, x[1], a1, 4, 4 * Q + 1, 0);
  a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w0, x[2], a0, 4, 4 * Q + 2, 0);
  a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w1, x[2], a1, 4, 4 * Q + 2, 0);
  a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w0, x[3], a0, 4, 4 * Q + 3, 0);
  a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w1, x[3], a1, 4, 4 * Q + 3, 0);
}
__global__ __launch_bounds__(512) void convlike(float* out, const float* win, int iters) {
  extern __shared__ float smem[];
  constexpr int PLANE = 1236;   // positions per channel-quad plane
  for (int i = threadIdx.x; i < 4 * PLANE * 4; i += blockDim.x) smem[i] = (i % 977) * 1e-3f - 0.4f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w[27][2];
#pragma unroll
  for (int t = 0; t < 27; ++t) { w[t][0] = win[(t * 2) * 64 + lane]; w[t][1] = win[(t * 2 + 1) * 64 + lane]; }
  const int vb = ((wave >> 1) * 6 * 34 + (wave & 1) * 2 * 34 + (lane >> 5) * 34 + (lane & 31)) * 4;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const float* sb = smem + vb + (it & 3) * 4;   // varies per iteration so the reads stay in the loop
    f32x4 xr[4];
    auto ld = [&](int idx) -> f32x4 {   // idx = tap * 4 + q
      const int t = idx >> 2, q = idx & 3;
      const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
      return *reinterpret_cast<const f32x4*>(sb + (q * PLANE + (kz * 6 + ky) * 34 + kx) * 4);
    };
    xr[0] = ld(0); xr[1] = ld(1); xr[2] = ld(2);
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      if (t * 4 + 3 < 108) xr[3] = ld(t * 4 + 3);
      quad<0>(a0, a1, w[t][0], w[t][1], xr[0]);
      if (t * 4 + 4 < 108) xr[0] = ld(t * 4 + 4);
      quad<1>(a0, a1, w[t][0], w[t][1], xr[1]);
      if (t * 4 + 5 < 108) xr[1] = ld(t * 4 + 5);
      quad<2>(a0, a1, w[t][0], w[t][1], xr[2]);
      if (t * 4 + 6 < 108) xr[2] = ld(t * 4 + 6);
      quad<3>(a0, a1, w[t][0], w[t][1], xr[3]);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3];
}

template <typename F>
static void run(const char* name, F launch, double flops) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  printf("%-34s %8.3f ms  %7.2f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; CK(hipMalloc(&out, 4096 * 1024 * 4));
  float h[256];
  for (int mode = 0; mode < 2; ++mode) {
    layout<<<1, 64>>>(out, mode);
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("mode %d (cbsz %d): D in lane l, reg r = A_lane * B_lane\n", mode, mode ? 4 : 0);
    for (int l : {0, 1, 4, 5, 21, 63})
      for (int r = 0; r < 4; ++r) {
        // factor: find (a in 1000..1063, b in 1..64) with a*b == h
        int fa = -1, fb = -1;
        for (int a = 1000; a < 1064 && fa < 0; ++a)
          for (int b = 1; b <= 64; ++b) if ((float)a * (float)b == h[l * 4 + r]) { fa = a - 1000; fb = b - 1; break; }
        printf("  lane %2d reg %d: A from lane %2d, B from lane %2d\n", l, r, fa, fb);
      }
  }
  const int iters = 8000;
  for (int nw : {4, 8, 16 / 2 * 1}) {
    char nm[64];
    snprintf(nm, 64, "4x4x1 acc8 waves%d", nw);
    run(nm, [&] { rate<8><<<256, nw * 64>>>(out, iters); }, 256.0 * nw * iters * 8 * 8 * 512.0);
    snprintf(nm, 64, "4x4x1 acc4 waves%d", nw);
    run(nm, [&] { rate<4><<<256, nw * 64>>>(out, iters); }, 256.0 * nw * iters * 8 * 4 * 512.0);
    snprintf(nm, 64, "4x4x1 acc2 waves%d", nw);
    run(nm, [&] { rate<2><<<256, nw * 64>>>(out, iters); }, 256.0 * nw * iters * 8 * 2 * 512.0);
    snprintf(nm, 64, "4x4x1 acc1 waves%d", nw);
    run(nm, [&] { rate<1><<<256, nw * 64>>>(out, iters); }, 256.0 * nw * iters * 8 * 1 * 512.0);
  }
  float* win; CK(hipMalloc(&win, 27 * 2 * 64 * 4)); CK(hipMemset(win, 0, 27 * 2 * 64 * 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(convlike), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  for (int per_cu : {1, 2}) {
    char nm[64];
    snprintf(nm, 64, "conv-like 4x4x1 8 waves x %d WG/CU", per_cu);
    const int it2 = 400;
    run(nm, [&] { convlike<<<256 * per_cu, 512, 4 * 1236 * 16>>>(out, win, it2); }, 256.0 * per_cu * 8 * it2 * 864 * 512.0);
  }
  return 0;
}
