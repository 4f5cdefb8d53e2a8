// K14: sliding-window accumulation (DataCarrier3D.concat_data, uncertainty_modeling/data_carrier_3D.py:137-179):
//   softmax_pred[pred_idx, :, crop] += softmax(patch logits);  num_predictions[crop] += 1 when pred_idx == 0
// for a batch of B patches x T predictions in one launch.  The class softmax (test_3D.py:435,448,472) is fused, so
// patch probabilities never exist in memory.  HBM-bound read-modify-write.  Overlapping patches of one launch hit
// the same voxels, so sums go through float atomics when `overlap` is set (patch_overlap < 1); with the shipped
// patch_overlap = 1 every voxel is written once and plain stores are used (bit-reproducible).
#include "common.h"

template <int C>
__global__ __launch_bounds__(256) void softmax_accumulate_kernel(const float* __restrict__ logits, int B, int T, int P0,
                                                                 int P1, int P2, const int32_t* __restrict__ crop,
                                                                 float* __restrict__ sum, float* __restrict__ count,
                                                                 int X, int Y, int Z, int overlap) {
  const int64_t pv = (int64_t)P0 * P1 * P2;
  const int64_t total = (int64_t)B * T * pv;
  const int64_t img = (int64_t)X * Y * Z;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i % pv;
    const int t = (int)((i / pv) % T);
    const int b = (int)(i / (pv * T));
    const int k = (int)(v % P2), j = (int)((v / P2) % P1), ii = (int)(v / ((int64_t)P1 * P2));
    const int x = crop[b * 3 + 0] + ii, y = crop[b * 3 + 1] + j, z = crop[b * 3 + 2] + k;
    if (x >= X || y >= Y || z >= Z) continue;
    const float* lg = logits + ((size_t)(b * T + t) * C) * pv + v;
    float e[C];
    float m = lg[0];
#pragma unroll
    for (int c = 1; c < C; ++c) m = fmaxf(m, lg[(size_t)c * pv]);
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { e[c] = expf(lg[(size_t)c * pv] - m); den += e[c]; }
    const float inv = 1.f / den;
    const int64_t o = ((int64_t)x * Y + y) * Z + z;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float* dst = sum + ((size_t)t * C + c) * img + o;
      if (overlap) atomicAdd(dst, e[c] * inv);
      else *dst += e[c] * inv;
    }
    if (t == 0) {
      if (overlap) atomicAdd(count + o, 1.f);
      else count[o] += 1.f;
    }
  }
}

extern "C" int vx_softmax_accumulate(const float* logits, int B, int T, int C, int P0, int P1, int P2, const int32_t* crop,
                                     float* sum, float* count, int X, int Y, int Z, int overlap, vx_stream_t stream) {
  if (!logits || !crop || !sum || !count) VX_FAIL(VX_E_NULL, "vx_softmax_accumulate: null pointer");
  if (B <= 0 || T <= 0 || P0 <= 0 || P1 <= 0 || P2 <= 0 || X <= 0 || Y <= 0 || Z <= 0)
    VX_FAIL(VX_E_SHAPE, "vx_softmax_accumulate: empty shape");
  const int64_t total = (int64_t)B * T * P0 * P1 * P2;
  int bx = (int)((total + 255) / 256);
  if (bx > 16384) bx = 16384;
  hipStream_t s = (hipStream_t)stream;
#define VX_ACC(CC)                                                                                                      \
  case CC:                                                                                                              \
    hipLaunchKernelGGL(softmax_accumulate_kernel<CC>, dim3(bx), dim3(256), 0, s, logits, B, T, P0, P1, P2, crop, sum,    \
                       count, X, Y, Z, overlap);                                                                        \
    break;
  switch (C) {
    VX_ACC(2) VX_ACC(3) VX_ACC(4) VX_ACC(5) VX_ACC(6) VX_ACC(7) VX_ACC(8)
    default: VX_FAIL(VX_E_SHAPE, "vx_softmax_accumulate: 2 <= C <= 8 (got %d)", C);
  }
#undef VX_ACC
  VX_CHECK_LAUNCH("vx_softmax_accumulate");
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Aleatoric-head sampling (predict_cases, test_3D.py:458-469): one forward gives (mu, s) = split(final_aleatoric);
// sigma = exp(s / 2); sample t: logits_t = mu + sigma * eps_t, eps ~ N(0, 1).  eps is either injected (parity with
// the reference's torch.randn stream is impossible otherwise) or generated here: Box-Muller on two avalanche
// hashes of (seed, sample, element).
__device__ __forceinline__ float vx_gauss(uint32_t seed, uint32_t a, uint32_t b) {
  const uint32_t h1 = vx_mix32(a * 0x9E3779B1u ^ vx_mix32(seed ^ (b * 0x85EBCA6Bu + 0x165667B1u)));
  const uint32_t h2 = vx_mix32(h1 ^ 0x27D4EB2Fu ^ (a * 0xC2B2AE35u));
  const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
  const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);           // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}

__global__ __launch_bounds__(256) void aleatoric_sample_kernel(const float* __restrict__ mu_s, const float* __restrict__ eps,
                                                               uint32_t seed, int N, int T, int C, int64_t nvox,
                                                               float* __restrict__ out, float* __restrict__ sigma) {
  const int64_t per = (int64_t)C * nvox;
  const int64_t total = (int64_t)N * per;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / per);
    const int64_t r = i - (int64_t)n * per;  // c * nvox + v
    const float mu = mu_s[(size_t)n * 2 * per + r];
    const float sg = expf(0.5f * mu_s[(size_t)n * 2 * per + per + r]);
    if (sigma) sigma[i] = sg;
    for (int t = 0; t < T; ++t) {
      const size_t o = ((size_t)n * T + t) * per + r;
      const float e = eps ? eps[o] : vx_gauss(seed, (uint32_t)r, (uint32_t)(n * T + t));
      out[o] = fmaf(sg, e, mu);
    }
  }
}

extern "C" int vx_aleatoric_sample(const float* mu_s, const float* eps, uint32_t seed, int N, int T, int C, int64_t nvox,
                                   float* out, float* sigma, vx_stream_t stream) {
  if (N <= 0 || T <= 0 || C <= 0 || nvox < 0) VX_FAIL(VX_E_SHAPE, "vx_aleatoric_sample: bad shape");
  if (nvox == 0) return VX_OK;
  if (!mu_s || !out) VX_FAIL(VX_E_NULL, "vx_aleatoric_sample: null pointer");
  if ((int64_t)C * nvox >= (1ll << 32)) VX_FAIL(VX_E_SHAPE, "vx_aleatoric_sample: sample too large");
  const int64_t total = (int64_t)N * C * nvox;
  int bx = (int)((total + 255) / 256);
  if (bx > 16384) bx = 16384;
  hipLaunchKernelGGL(aleatoric_sample_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, mu_s, eps, seed, N, T, C, nvox,
                     out, sigma);
  VX_CHECK_LAUNCH("vx_aleatoric_sample");
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Stochastic segmentation networks (SsnUNet3D.forward, ssn_unet3D_module.py:39-70; predict_cases_ssn,
// test_3D.py:361-396): the three 1x1x1 heads on the decoder features are ONE 1x1x1 conv with (2 + R) * C output
// channels -- [mean (C) | log_cov_diag (C) | cov_factor (R * C, channel r * C + c)] -- whose planar output `head`
// this kernel turns into samples of LowRankMultivariateNormal(loc, cov_factor, cov_diag).rsample:
//     y_s[c][v] = mean[c][v] + sum_r factor[r][c][v] * eps_w[s][r] + sqrt(exp(logd[c][v]) + epsilon) * eps_d[s][c][v]
// eps_w is ONE rank-R vector per sample (the low-rank term couples all voxels), eps_d one normal per element; both
// are injected (parity) or generated from `seed`.  HBM-bound: (2 + R) * C floats read, S * C written per voxel.
__global__ __launch_bounds__(256) void ssn_sample_kernel(const float* __restrict__ head, const float* __restrict__ eps_w,
                                                         const float* __restrict__ eps_d, uint32_t seed, int N, int S,
                                                         int C, int R, int64_t nvox, float epsilon,
                                                         float* __restrict__ out) {
  const int64_t per = (int64_t)C * nvox;
  const int64_t total = (int64_t)N * per;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / per);
    const int64_t r = i - (int64_t)n * per;   // c * nvox + v
    const float* h = head + (size_t)n * (2 + R) * per;
    const float mean = h[r];
    const float sd = sqrtf(expf(h[per + r]) + epsilon);
    for (int s = 0; s < S; ++s) {
      float low = 0.f;
      for (int k = 0; k < R; ++k) {
        const float ew = eps_w ? eps_w[((size_t)s * N + n) * R + k]
                               : vx_gauss(seed ^ 0x5bd1e995u, (uint32_t)k, (uint32_t)(n * S + s));
        low = fmaf(h[(size_t)(2 + k) * per + r], ew, low);
      }
      const size_t o = ((size_t)n * S + s) * per + r;
      const float ed = eps_d ? eps_d[((size_t)s * N + n) * per + r] : vx_gauss(seed, (uint32_t)r, (uint32_t)(n * S + s));
      out[o] = (mean + low) + sd * ed;   // loc + W eps_W + D^(1/2) eps_D, the order torch adds them
    }
  }
}

extern "C" int vx_ssn_sample(const float* head, const float* eps_w, const float* eps_d, uint32_t seed, int N, int S, int C,
                             int R, int64_t nvox, float epsilon, float* out, vx_stream_t stream) {
  if (N <= 0 || S <= 0 || C <= 0 || R < 0 || nvox < 0) VX_FAIL(VX_E_SHAPE, "vx_ssn_sample: bad shape");
  if (nvox == 0) return VX_OK;
  if (!head || !out) VX_FAIL(VX_E_NULL, "vx_ssn_sample: null pointer");
  if ((int64_t)C * nvox >= (1ll << 32)) VX_FAIL(VX_E_SHAPE, "vx_ssn_sample: sample too large");
  const int64_t total = (int64_t)N * C * nvox;
  int bx = (int)((total + 255) / 256);
  if (bx > 16384) bx = 16384;
  hipLaunchKernelGGL(ssn_sample_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, head, eps_w, eps_d, seed, N, S, C, R,
                     nvox, epsilon, out);
  VX_CHECK_LAUNCH("vx_ssn_sample");
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// 2D stochastic segmentation head (HighResolutionNet.hrnet_ssn, hrnet_module.py:559-595).  The reference
// interpolates mean, exp(.)+eps and the rank-R factor to the input size and samples the full-resolution low-rank
// normal.  Bilinear interpolation is linear, so  interp(mean) + sum_r interp(factor_r) eps_r  =
// interp(mean + sum_r factor_r eps_r): the rank-R combination is formed at the head's 1/4 resolution (16 x fewer
// elements, and the (B, C*H*W, R) factor tensor -- 117 MB per 256x478 image at R = 10 -- never exists), then ONE
// upsample per sample; the diagonal term needs interp(exp(mean)) at full resolution (second kernel).
//   ssn2d_lowres_kernel : channels-last low-res mean [P][pm], factor [P][pf] (channel r * C + c), eps_w [S][B][R]
//                         -> comb [S][P][C] = mean + sum_r factor_r * eps_w[s][b][r],  expm [P][C] = exp(mean)
//   ssn2d_add_diag_kernel: out [S][B][C][HW] += sqrt(diag [B][C][HW] + epsilon) * eps_d (injected or generated)
__global__ __launch_bounds__(256) void ssn2d_lowres_kernel(const float* __restrict__ mean, int pm, const float* __restrict__ fac,
                                                           int pf, const float* __restrict__ eps_w, uint32_t seed, int B,
                                                           int64_t pix_per_image, int S, int C, int R,
                                                           float* __restrict__ comb, float* __restrict__ expm) {
  const int64_t P = (int64_t)B * pix_per_image;
  const int64_t total = P * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / C;
    const int c = (int)(i - pix * C);
    const int b = (int)(pix / pix_per_image);
    const float m = mean[pix * pm + c];
    if (expm) expm[i] = expf(m);
    for (int s = 0; s < S; ++s) {
      float low = 0.f;
      for (int k = 0; k < R; ++k) {
        const float ew = eps_w ? eps_w[((size_t)s * B + b) * R + k]
                               : vx_gauss(seed ^ 0x5bd1e995u, (uint32_t)k, (uint32_t)(b * S + s));
        low = fmaf(fac[pix * pf + k * C + c], ew, low);
      }
      comb[((size_t)s * P + pix) * C + c] = m + low;
    }
  }
}

extern "C" int vx_ssn2d_lowres(const float* mean, int mean_pitch, const float* factor, int factor_pitch, const float* eps_w,
                               uint32_t seed, int B, int64_t pix_per_image, int S, int C, int R, float* comb, float* expm,
                               vx_stream_t stream) {
  if (B <= 0 || pix_per_image <= 0 || S <= 0 || C <= 0 || R < 0 || mean_pitch < C || (R > 0 && factor_pitch < R * C))
    VX_FAIL(VX_E_SHAPE, "vx_ssn2d_lowres: bad shape");
  if (!mean || !comb || (R > 0 && !factor)) VX_FAIL(VX_E_NULL, "vx_ssn2d_lowres: null pointer");
  const int64_t total = (int64_t)B * pix_per_image * C;
  int bx = (int)((total + 255) / 256);
  if (bx > 8192) bx = 8192;
  hipLaunchKernelGGL(ssn2d_lowres_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, mean, mean_pitch, factor, factor_pitch,
                     eps_w, seed, B, pix_per_image, S, C, R, comb, expm);
  VX_CHECK_LAUNCH("vx_ssn2d_lowres");
  return VX_OK;
}

__global__ __launch_bounds__(256) void ssn2d_add_diag_kernel(float* __restrict__ out, const float* __restrict__ diag,
                                                             const float* __restrict__ eps_d, uint32_t seed, int S, int B,
                                                             int64_t per, float epsilon) {
  // out [B][S][per] (slot b * S + s), diag [B][per], eps_d [S][B][per]
  const int64_t total = (int64_t)B * per;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per);
    const int64_t r = i - (int64_t)b * per;
    const float sd = sqrtf(diag[i] + epsilon);
    for (int s = 0; s < S; ++s) {
      const size_t o = ((size_t)b * S + s) * per + r;
      const float e = eps_d ? eps_d[((size_t)s * B + b) * per + r] : vx_gauss(seed, (uint32_t)r, (uint32_t)(b * S + s));
      out[o] = out[o] + sd * e;
    }
  }
}

extern "C" int vx_ssn2d_add_diag(float* out, const float* diag, const float* eps_d, uint32_t seed, int S, int B, int64_t per,
                                 float epsilon, vx_stream_t stream) {
  if (S <= 0 || B <= 0 || per <= 0 || per >= (1ll << 32)) VX_FAIL(VX_E_SHAPE, "vx_ssn2d_add_diag: bad shape");
  if (!out || !diag) VX_FAIL(VX_E_NULL, "vx_ssn2d_add_diag: null pointer");
  const int64_t total = (int64_t)B * per;
  int bx = (int)((total + 255) / 256);
  if (bx > 16384) bx = 16384;
  hipLaunchKernelGGL(ssn2d_add_diag_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, out, diag, eps_d, seed, S, B, per,
                     epsilon);
  VX_CHECK_LAUNCH("vx_ssn2d_add_diag");
  return VX_OK;
}
