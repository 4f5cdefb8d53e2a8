// The TTA branch of Cityscapes_dataset.__getitem__ (uncertainty_modeling/data/cityscapes_dataset.py:76-99) on the device:
// one launch turns a batch of images into the G normalised views the 2D driver forwards (test_2D.py:299-311), already in
// the layout the stem convolution stages from -- channels-last float32 at pitch 4 (channel 3 = 0) -- with the flips as index
// arithmetic.  The host path (values_amd.data.tta_views_2d + torch.flip / permute / zero-fill per view) moved 6 tensors
// per view through ATen kernels; this is one streaming pass: 3 (+12) bytes read, 16 bytes written per output pixel.
//
//   u8 source (the reference's dataset path): img [B][H][W][3] uint8; a noisy view adds a float field to the uint8 image,
//     clips to 0..255 and truncates back to uint8 (albumentations.GaussNoise on uint8 input) BEFORE
//     Normalize(mean, std, max_pixel_value) = (v - mean * max) * (1 / (std * max)), evaluated as values_amd.data.tta_views_2d
//     does (two float32 roundings): bit-exact with it.  The noise field of a flipped view is indexed in the VIEW's
//     coordinates (the dataset applies GaussNoise after HorizontalFlip: an independent draw per view) unless the view code
//     says otherwise (bit 3: the field travels with the image -- "flip of the noisy image", the 8-view set of config C4).
//   f32 source (already normalised tensors, what values_amd.predict2d.tta_views_8 takes): clean / noisy [B][3][H][W].
#include "common.h"

namespace {
struct TtaArgs {
  const void* src;          // u8 [B][H][W][3]  |  f32 [B][3][H][W]
  const float* noise[2];    // u8: additive fields [B][H][W][3] (slot 0 / 1, nullable);  f32: noise[0] = the noisy tensor [B][3][H][W]
  float mean255[3], inv[3];
  float* out;               // [G][B][H][W][4]
  int B, H, W, G;
  int code[16];             // per view: bit 0 hflip, bit 1 vflip, bit 2 noisy, bit 3 field indexed at the SOURCE pixel, bits 4.. noise slot
};

template <bool U8>
__global__ __launch_bounds__(256) void tta_views_2d_kernel(TtaArgs a) {
  const int64_t per = (int64_t)a.H * a.W;
  const int64_t total = per * a.B * a.G;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % a.W);
    int64_t r = i / a.W;
    const int y = (int)(r % a.H); r /= a.H;
    const int b = (int)(r % a.B);
    const int g = (int)(r / a.B);
    const int code = a.code[g];
    const int xs = (code & 1) ? a.W - 1 - x : x;
    const int ys = (code & 2) ? a.H - 1 - y : y;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if constexpr (U8) {
      const uint8_t* p = reinterpret_cast<const uint8_t*>(a.src) + ((int64_t)b * per + (int64_t)ys * a.W + xs) * 3;
      const float* nf = (code & 4) ? a.noise[(code >> 4) & 1] : nullptr;
      const int64_t ni = (code & 8) ? ((int64_t)b * per + (int64_t)ys * a.W + xs) * 3 : ((int64_t)b * per + (int64_t)y * a.W + x) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = (float)p[c];
        if (nf) {
          v = fminf(fmaxf(v + nf[ni + c], 0.f), 255.f);     // np.clip(img.astype(float32) + n, 0, 255)
          v = (float)(uint8_t)v;                            // .astype(uint8): truncation
        }
        o[c] = (v - a.mean255[c]) * a.inv[c];
      }
    } else {
      const float* base = (code & 4) ? a.noise[0] : reinterpret_cast<const float*>(a.src);
      const float* p = base + (int64_t)b * 3 * per + (int64_t)ys * a.W + xs;
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = p[(int64_t)c * per];
    }
    *reinterpret_cast<f32x4*>(a.out + i * 4) = o;
  }
}
}  // namespace

extern "C" int vx_tta_views_2d(const void* src, int src_u8, const float* noise0, const float* noise1, const float* mean,
                               const float* std, float max_pixel_value, int B, int H, int W, int G, const int32_t* view_code,
                               float* out, vx_stream_t stream) {
  if (!src || !out || !view_code) VX_FAIL(VX_E_NULL, "vx_tta_views_2d: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || G <= 0 || G > 16) VX_FAIL(VX_E_SHAPE, "vx_tta_views_2d: B=%d H=%d W=%d G=%d (1..16 views)", B, H, W, G);
  if (!vx_aligned16(out)) VX_FAIL(VX_E_ALIGN, "vx_tta_views_2d: out must be 16-byte aligned");
  TtaArgs a = {};
  a.src = src; a.noise[0] = noise0; a.noise[1] = noise1; a.out = out;
  a.B = B; a.H = H; a.W = W; a.G = G;
  for (int g = 0; g < G; ++g) {
    const int c = view_code[g];      // HOST array: the view list is part of the call, like the transform names of the dataset
    if (c < 0 || c > 31) VX_FAIL(VX_E_DTYPE, "vx_tta_views_2d: view code %d", c);
    if ((c & 4) && !a.noise[src_u8 ? (c >> 4) & 1 : 0]) VX_FAIL(VX_E_NULL, "vx_tta_views_2d: view %d is noisy but its noise tensor is null", g);
    a.code[g] = c;
  }
  if (src_u8) {
    if (!mean || !std || !(max_pixel_value > 0.f)) VX_FAIL(VX_E_NULL, "vx_tta_views_2d: uint8 source needs mean / std / max_pixel_value");
    for (int c = 0; c < 3; ++c) {
      a.mean255[c] = mean[c] * max_pixel_value;            // float32 products, as numpy forms them
      a.inv[c] = 1.0f / (std[c] * max_pixel_value);
    }
  }
  const int64_t total = (int64_t)G * B * H * W;
  int bx = (int)((total + 255) / 256);
  if (bx > 16384) bx = 16384;
  if (src_u8) hipLaunchKernelGGL(tta_views_2d_kernel<true>, dim3((unsigned)bx), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(tta_views_2d_kernel<false>, dim3((unsigned)bx), dim3(256), 0, (hipStream_t)stream, a);
  VX_CHECK_LAUNCH("vx_tta_views_2d");
  return VX_OK;
}
