// Micro-test for the round-2 direction: fp32 products on the fp16 matrix cores by operand splitting.
//   x = hi + lo * 2^-11,  hi = fp16(x),  lo = fp16((x - hi) * 2^11)          (exact residual, 22+ mantissa bits)
//   a * b ~= hi_a hi_b + 2^-11 (hi_a lo_b + lo_a hi_b)                        (dropped lo*lo term <= 2^-24 relative)
// three v_mfma_f32_16x16x32_f16 (fp32 accumulate; two accumulators: main and cross) per fp32-equivalent K = 32 step,
// against eight v_mfma_f32_16x16x4_f32.  Reports (1) max / rms error of both against a float64 dot product on
// conv-like data (K = 432 = 27 taps x 16 channels, activations ~ N(0,1) with half of them zeroed, weights
// ~ U(+-1/sqrt(K))), (2) TFLOP/s of fp32-equivalent work of both loops with operands in registers.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// D (16x16) = A (16xK) * B (Kx16); one wave; A row-major [16][K], B [K][16] (column n contiguous per k)
__global__ void gemm_f32(const float* A, const float* B, int K, float* D) {
  const int l = threadIdx.x, m = l & 15, g = l >> 4;
  f32x4 acc = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * K + k0 + g], B[(k0 + g) * 16 + m], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + m] = acc[r];
}
__global__ void gemm_split(const float* A, const float* B, int K, float* D) {
  const int l = threadIdx.x, m = l & 15, g = l >> 4;
  f32x4 acc = {0, 0, 0, 0}, accx = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 32) {
    f16x8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + g * 8 + j;
      const float a = k < K ? A[m * K + k] : 0.f, b = k < K ? B[k * 16 + m] : 0.f;
      const _Float16 h1 = (_Float16)a, h2 = (_Float16)b;
      ah[j] = h1; al[j] = (_Float16)((a - (float)h1) * 2048.f);
      bh[j] = h2; bl[j] = (_Float16)((b - (float)h2) * 2048.f);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, accx, 0, 0, 0);
    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, accx, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + m] = acc[r] + accx[r] * (1.0f / 2048.f);
}

template <int MODE>
__global__ __launch_bounds__(512) void rate(float* out, int iters) {
  const int l = threadIdx.x;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
  f16x8 h[4];
  float f[8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) h[i][j] = (_Float16)(((l * 8 + j + i * 77) % 251) * 0.007f - 0.8f);
  for (int j = 0; j < 8; ++j) f[j] = ((l * 8 + j) % 251) * 0.007f - 0.8f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {   // fp32-equivalent K = 32 for two independent tiles: 2 x 8 f32 MFMAs
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j], f[7 - j], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[7 - j], f[j], a1, 0, 0, 0);
      }
    } else {            // the same work: 2 x 3 f16 MFMAs
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[0], h[1], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[0], h[3], a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[2], h[1], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[1], h[0], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[1], h[2], a3, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[3], h[0], a3, 0, 0, 0);
    }
  }
  out[blockIdx.x * blockDim.x + l] = a0[0] + a1[1] + a2[2] + a3[3];
}

int main(int argc, char** argv) {
  const int K = 432, NT = 64;
  std::vector<float> A(16 * K), B(K * 16);
  double max32 = 0, max16 = 0, s32 = 0, s16 = 0, sref = 0;
  float *dA, *dB, *dD; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 256 * 4));
  unsigned st = 12345;
  auto rnd = [&] { st = st * 1664525u + 1013904223u; return (st >> 8) * (1.0f / 16777216.0f); };
  auto gauss = [&] { float u1 = rnd() + 1e-7f, u2 = rnd(); return sqrtf(-2 * logf(u1)) * cosf(6.2831853f * u2); };
  for (int t = 0; t < NT; ++t) {
    for (auto& v : A) v = (2 * rnd() - 1) / sqrtf((float)K);
    for (auto& v : B) v = rnd() < 0.5f ? 0.f : 2.f * fmaxf(gauss(), 0.01f * gauss());   // dropout-like activations
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    float D32[256], D16[256];
    gemm_f32<<<1, 64>>>(dA, dB, K, dD); CK(hipMemcpy(D32, dD, sizeof(D32), hipMemcpyDeviceToHost));
    gemm_split<<<1, 64>>>(dA, dB, K, dD); CK(hipMemcpy(D16, dD, sizeof(D16), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) {
      double r = 0; for (int k = 0; k < K; ++k) r += (double)A[i * K + k] * (double)B[k * 16 + n];
      const double e32 = fabs(D32[i * 16 + n] - r), e16 = fabs(D16[i * 16 + n] - r);
      max32 = fmax(max32, e32); max16 = fmax(max16, e16); s32 += e32 * e32; s16 += e16 * e16; sref += r * r;
    }
  }
  const double n = NT * 256.0;
  printf("K=%d dot products, rms |ref| %.3f:  fp32 MFMA  max err %.3e rms %.3e   |   split-fp16 x3  max err %.3e rms %.3e\n", K,
         sqrt(sref / n), max32, sqrt(s32 / n), max16, sqrt(s16 / n));
  float* out; CK(hipMalloc(&out, 512 * 512 * 4));
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&] { if (mode == 0) rate<0><<<512, 512>>>(out, iters / 8); else rate<1><<<512, 512>>>(out, iters); };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 2;
    const double steps = (double)(mode == 0 ? iters / 8 : iters) * 2 * 512 * 8;   // fp32-equivalent 16x16x32 steps
    printf("%s: %.3f ms, %.1f TFLOP/s fp32-equivalent\n", mode == 0 ? "fp32 16x16x4  " : "split f16 x3  ", ms,
           steps * 16384.0 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
