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
