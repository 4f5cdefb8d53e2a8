// K8 (+K13): final 1x1x1 convolution F -> C (unet3D_module.py:199, 365), reading the channels-last
// decoder output and writing logits in the reference's NCDHW layout: sample n goes to slot dst[n]
// (the pred_idx of predict_cases, test_3D.py:423-482) and is un-flipped by flip[n] on the way out
// (torch.flip(model.forward(torch.flip(x, dims)), dims), test_3D.py:445-447).  HBM-bound streaming.
#include "common.h"

template <int F>
__global__ __launch_bounds__(256) void conv1x1_ncdhw_kernel(const float* __restrict__ in, int in_pitch,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ out, int N, int D, int H, int W, int C,
                                                            const int32_t* __restrict__ dst,
                                                            const int32_t* __restrict__ flip) {
  const int64_t nvox = (int64_t)D * H * W;
  const int64_t total = (int64_t)N * nvox;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / nvox);
    int64_t v = i - (int64_t)n * nvox;
    float xin[F];
#pragma unroll
    for (int k = 0; k < F; k += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(in + (size_t)i * in_pitch + k);
      xin[k] = t[0]; xin[k + 1] = t[1]; xin[k + 2] = t[2]; xin[k + 3] = t[3];
    }
    const int f = flip ? flip[n] : 0;
    if (f) {
      int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((int64_t)W * H));
      if (f & 1) z = D - 1 - z;
      if (f & 2) y = H - 1 - y;
      if (f & 4) x = W - 1 - x;
      v = ((int64_t)z * H + y) * W + x;
    }
    const int slot = dst ? dst[n] : n;
    float* o = out + (size_t)slot * C * nvox + v;
    for (int c = 0; c < C; ++c) {
      float acc = bias[c];
#pragma unroll
      for (int k = 0; k < F; ++k) acc = fmaf(w[c * F + k], xin[k], acc);
      o[(size_t)c * nvox] = acc;
    }
  }
}

extern "C" int vx_conv1x1_ncdhw(const float* in, int in_pitch, const float* w, const float* bias, float* out, int N,
                                int D, int H, int W, int F, int C, const int32_t* dst, const int32_t* flip,
                                vx_stream_t stream) {
  if (!in || !w || !bias || !out) VX_FAIL(VX_E_NULL, "vx_conv1x1_ncdhw: null pointer");
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) VX_FAIL(VX_E_SHAPE, "vx_conv1x1_ncdhw: empty tensor");
  if (in_pitch % 4 || in_pitch < F || !vx_aligned16(in)) VX_FAIL(VX_E_ALIGN, "vx_conv1x1_ncdhw: input pitch/alignment");
  const int64_t total = (int64_t)N * D * H * W;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipStream_t s = (hipStream_t)stream;
#define VX_1X1(FF)                                      \
  vx_note_kernel("conv1x1_ncdhw_kernel<" #FF ">");        \
  hipLaunchKernelGGL(conv1x1_ncdhw_kernel<FF>, dim3(blocks), dim3(256), 0, s, in, in_pitch, w, bias, out, N, D, H, W, C, dst, flip)
  switch (F) {
    case 8: VX_1X1(8); break;
    case 16: VX_1X1(16); break;
    case 32: VX_1X1(32); break;
    default: VX_FAIL(VX_E_SHAPE, "vx_conv1x1_ncdhw: F=%d unsupported (8, 16, 32)", F);
  }
#undef VX_1X1
  VX_CHECK_LAUNCH("vx_conv1x1_ncdhw");
  return VX_OK;
}
