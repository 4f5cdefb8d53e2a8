// K16/K17/K18 of the 2D (HRNet) path, all HBM-bound streaming kernels on channels-last fp32 [N][H][W][pitch]:
//   vx_bn_finalize      TRAINING-mode BatchNorm2d statistics (hrnet_module.py:30, 50, 55 ...): reduce the
//                       conv epilogue's per-tile (sum, sumsq) partials over the whole batch in float64 ->
//                       scale[c] = gamma * rstd, shift[c] = beta - mean * scale (biased variance, eps 1e-5).
//   vx_affine_gather    out = act( [add] + scale[c] * G(drop(x)) + shift[c] ), where G is the identity or the
//                       bilinear resize of F.interpolate(mode="bilinear", align_corners=False)
//                       (hrnet_module.py:324-329, 650-658).  One kernel covers: BN+ReLU, BN + residual + ReLU (block
//                       ends, :72-75), the SUM fusion of HighResolutionModule (:316-333, accumulating term by term in
//                       the reference's order), F.dropout on the stage-4 outputs (:642-646) fused with their upsample
//                       into the 720-channel concat buffer (torch.cat never runs).
//   vx_bilinear_nchw    final upsample of the class logits to the input size, written in the reference's NCHW layout
//                       into the pred_idx slot (hrnet_module.py:667-669; test_2D.py:302-316).
#include "common.h"

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int ntiles, int C,
                                                          double inv_count, float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ scale,
                                                          float* __restrict__ shift, int cpitch) {
  __shared__ double s_s[4], s_q[4];
  const int c = blockIdx.x;
  // blockIdx.y = statistics group (vx_bn_finalize_groups): its tiles are rows [g * ntiles, (g + 1) * ntiles) of the
  // partials, its scale / shift row g of [G][cpitch]
  partial += (size_t)blockIdx.y * ntiles * C * 2;
  scale += (size_t)blockIdx.y * cpitch;
  shift += (size_t)blockIdx.y * cpitch;
  double s = 0.0, q = 0.0;
  for (int t = threadIdx.x; t < ntiles; t += 256) {
    s += (double)partial[((size_t)t * C + c) * 2 + 0];
    q += (double)partial[((size_t)t * C + c) * 2 + 1];
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
  if ((threadIdx.x & 63) == 0) { s_s[threadIdx.x >> 6] = s; s_q[threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = s_s[0] + s_s[1] + s_s[2] + s_s[3];
    q = s_q[0] + s_q[1] + s_q[2] + s_q[3];
    const double mu = s * inv_count;
    double var = q * inv_count - mu * mu;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
    scale[c] = (float)(g * rstd);
    shift[c] = (float)(b - mu * g * rstd);
  }
}

extern "C" int vx_bn_finalize(const float* stats_partial, int ntiles, int C, int64_t count, float eps, const float* gamma,
                              const float* beta, float* scale, float* shift, vx_stream_t stream) {
  if (!stats_partial || !scale || !shift) VX_FAIL(VX_E_NULL, "vx_bn_finalize: null pointer");
  if (ntiles <= 0 || C <= 0 || count <= 0) VX_FAIL(VX_E_SHAPE, "vx_bn_finalize: empty");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, stats_partial, ntiles, C,
                     1.0 / (double)count, eps, gamma, beta, scale, shift, C);
  VX_CHECK_LAUNCH("vx_bn_finalize");
  return VX_OK;
}

// The same for G statistics groups in one launch: the batch holds G independent BatchNorm batches (the TTA views of
// test_2D.py:299-311 are separate forwards, each normalised with ITS batch statistics; here they travel as one batch of
// G x B images and every group of B consecutive images keeps its own statistics -- bit for bit the separate forwards).
extern "C" int vx_bn_finalize_groups(const float* stats_partial, int ntiles_per_group, int G, int C, int cpitch,
                                     int64_t count_per_group, float eps, const float* gamma, const float* beta, float* scale,
                                     float* shift, vx_stream_t stream) {
  if (!stats_partial || !scale || !shift) VX_FAIL(VX_E_NULL, "vx_bn_finalize_groups: null pointer");
  if (ntiles_per_group <= 0 || G <= 0 || C <= 0 || cpitch < C || count_per_group <= 0) VX_FAIL(VX_E_SHAPE, "vx_bn_finalize_groups: empty");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)C, (unsigned)G), dim3(256), 0, (hipStream_t)stream, stats_partial,
                     ntiles_per_group, C, 1.0 / (double)count_per_group, eps, gamma, beta, scale, shift, cpitch);
  VX_CHECK_LAUNCH("vx_bn_finalize_groups");
  return VX_OK;
}

// bilinear source coordinates of F.interpolate(align_corners=False): src = (dst + 0.5) * in/out - 0.5, clamped at 0
__device__ __forceinline__ void bil_coord(int d, int in, float ratio, int& i0, int& i1, float& l1) {
  float s = ((float)d + 0.5f) * ratio - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

// Index scheme of the two streaming kernels below (round 5): grid = (pieces of an output row / 256, OH, N) -- image and row are
// block indices, the piece within the row splits into (pixel, channel quad) with ONE 32-bit multiply-high.  The first form
// decoded a flat 64-bit piece index with three 64-bit divisions per thread (~600 vector instructions around 3 loads and a
// store): the passes ran at 0.15-0.2 of the HBM rate and were the top line of the C4 profile (18 % of the step).
template <bool RESIZE>
__global__ __launch_bounds__(256) void affine_gather_kernel(vx_affine_args a, unsigned mC4, int rowp) {
  const int C4 = a.C / 4;
  const float ry = (float)a.H / (float)a.OH, rx = (float)a.W / (float)a.OW;
  {
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    if ((int)p >= rowp) return;
    const int ox = C4 == 1 ? (int)p : (int)__umulhi(p, mC4);
    const int c = ((int)p - ox * C4) * 4;
    const int oy = (int)blockIdx.y;
    const int n = (int)blockIdx.z;
    const vx_dkey dkey = vx_drop_key(a.drop_seed, a.drop_layer, (uint32_t)n);
    auto fetch = [&](int y, int x) {
      const size_t pix = ((size_t)n * a.H + y) * a.W + x;
      f32x4 v = *reinterpret_cast<const f32x4*>(a.x + pix * a.x_pitch + c);
      if (a.drop_mode == VX_DROP_HASH) {
        const uint32_t bits = vx_drop_bits4(dkey, (uint32_t)((y * a.W + x) * a.C + c));
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);
      } else if (a.drop_mode == VX_DROP_MASK) {
        const uint32_t mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + pix * a.C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * v[j] : 0.f;
      }
      return v;
    };
    f32x4 v;
    if (RESIZE) {
      int y0, y1, x0, x1;
      float ly, lx;
      bil_coord(oy, a.H, ry, y0, y1, ly);
      bil_coord(ox, a.W, rx, x0, x1, lx);
      const f32x4 v00 = fetch(y0, x0), v01 = fetch(y0, x1), v10 = fetch(y1, x0), v11 = fetch(y1, x1);
      // ATen upsample_bilinear2d: h0lambda * (w0lambda * v00 + w1lambda * v01) + h1lambda * (w0lambda * v10 + w1lambda * v11)
      const float hy = 1.f - ly, hx = 1.f - lx;
      v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    } else {
      v = fetch(oy, ox);
    }
    if (a.scale) {
      // group_images > 0: image n takes row n / group_images of scale / shift [G][C] (vx_bn_finalize_groups)
      const size_t row = a.group_images > 0 ? (size_t)(n / a.group_images) * a.C : 0;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + row + c), sh = *reinterpret_cast<const f32x4*>(a.shift + row + c);
      v = v * sc + sh;
    }
    const size_t opix = ((size_t)n * a.OH + oy) * a.OW + ox;
    if (a.add) v = *reinterpret_cast<const f32x4*>(a.add + opix * a.add_pitch + c) + v;
    if (a.act == VX_ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    *reinterpret_cast<f32x4*>(a.out + opix * a.out_pitch + a.out_coff + c) = v;
  }
}

extern "C" int vx_affine_gather(const vx_affine_args* ap, vx_stream_t stream) {
  if (!ap) VX_FAIL(VX_E_NULL, "vx_affine_gather: null args");
  const vx_affine_args& a = *ap;
  if (!a.x || !a.out) VX_FAIL(VX_E_NULL, "vx_affine_gather: null tensor");
  if ((a.scale == nullptr) != (a.shift == nullptr)) VX_FAIL(VX_E_NULL, "vx_affine_gather: scale/shift must come together");
  if (a.N <= 0 || a.H <= 0 || a.W <= 0 || a.OH <= 0 || a.OW <= 0 || a.C <= 0 || a.C % 4)
    VX_FAIL(VX_E_SHAPE, "vx_affine_gather: bad shape (C must be a multiple of 4)");
  if (a.x_pitch % 4 || a.x_pitch < a.C || a.out_pitch % 4 || a.out_coff % 4 || a.out_pitch < a.out_coff + a.C ||
      (a.add && (a.add_pitch % 4 || a.add_pitch < a.C)))
    VX_FAIL(VX_E_ALIGN, "vx_affine_gather: pitches/offsets must be multiples of 4 floats");
  if (a.drop_mode == VX_DROP_MASK && !a.drop_mask) VX_FAIL(VX_E_NULL, "vx_affine_gather: mask mode without mask");
  if (a.act != VX_ACT_NONE && a.act != VX_ACT_RELU) VX_FAIL(VX_E_DTYPE, "vx_affine_gather: act must be none or relu");
  const int C4 = a.C / 4;
  const int64_t rowp64 = (int64_t)a.OW * C4;
  if (a.OH > 65535 || a.N > 65535 || rowp64 * C4 >= (1ll << 31))
    VX_FAIL(VX_E_SHAPE, "vx_affine_gather: at most 65535 rows / images and 2^31 / (C / 4) pieces per row (got %d x %d x %d x %d)", a.N, a.OH, a.OW, a.C);
  const int rowp = (int)rowp64;
  const unsigned mC4 = (unsigned)((1ull << 32) / (unsigned)C4) + 1u;      // exact for p < 2^32 / C4
  const dim3 grid((unsigned)((rowp + 255) / 256), (unsigned)a.OH, (unsigned)a.N);
  hipStream_t s = (hipStream_t)stream;
  if (a.OH != a.H || a.OW != a.W)
    hipLaunchKernelGGL(affine_gather_kernel<true>, grid, dim3(256), 0, s, a, mC4, rowp);
  else
    hipLaunchKernelGGL(affine_gather_kernel<false>, grid, dim3(256), 0, s, a, mC4, rowp);
  VX_CHECK_LAUNCH("vx_affine_gather");
  return VX_OK;
}

// vx_fuse_sum: the terms of a SUM fusion in one pass (see the header); every term is evaluated with affine_gather_kernel's
// expressions, the sum runs in term order -- the bits of the chained passes
template <int NT>
__global__ __launch_bounds__(256) void fuse_sum_kernel(vx_fuse_args a, unsigned mC4, int rowp) {
  const int C4 = a.C / 4;
  {
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    if ((int)p >= rowp) return;
    const int ox = C4 == 1 ? (int)p : (int)__umulhi(p, mC4);
    const int c = ((int)p - ox * C4) * 4;
    const int oy = (int)blockIdx.y;
    const int n = (int)blockIdx.z;
    const size_t row = a.group_images > 0 ? (size_t)(n / a.group_images) * a.C : 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const vx_fuse_term& tm = a.term[t];
      f32x4 v;
      if (tm.H != a.OH || tm.W != a.OW) {
        const float ry = (float)tm.H / (float)a.OH, rx = (float)tm.W / (float)a.OW;
        int y0, y1, x0, x1;
        float ly, lx;
        bil_coord(oy, tm.H, ry, y0, y1, ly);
        bil_coord(ox, tm.W, rx, x0, x1, lx);
        const float* base = tm.x + (size_t)n * tm.H * tm.W * tm.x_pitch + c;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * tm.W + x0) * tm.x_pitch);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * tm.W + x1) * tm.x_pitch);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * tm.W + x0) * tm.x_pitch);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * tm.W + x1) * tm.x_pitch);
        const float hy = 1.f - ly, hx = 1.f - lx;
        v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
      } else {
        v = *reinterpret_cast<const f32x4*>(tm.x + (((size_t)n * a.OH + oy) * a.OW + ox) * tm.x_pitch + c);
      }
      if (tm.scale) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(tm.scale + row + c), sh = *reinterpret_cast<const f32x4*>(tm.shift + row + c);
        v = v * sc + sh;
      }
      acc = t == 0 ? v : acc + v;
    }
    if (a.act == VX_ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.f);
    }
    *reinterpret_cast<f32x4*>(a.out + (((size_t)n * a.OH + oy) * a.OW + ox) * a.out_pitch + c) = acc;
  }
}

extern "C" int vx_fuse_sum(const vx_fuse_args* ap, vx_stream_t stream) {
  if (!ap) VX_FAIL(VX_E_NULL, "vx_fuse_sum: null args");
  const vx_fuse_args& a = *ap;
  if (a.nterms < 1 || a.nterms > 4) VX_FAIL(VX_E_SHAPE, "vx_fuse_sum: 1 .. 4 terms, got %d", a.nterms);
  if (!a.out || a.N <= 0 || a.OH <= 0 || a.OW <= 0 || a.C <= 0 || a.C % 4 || a.out_pitch % 4 || a.out_pitch < a.C)
    VX_FAIL(VX_E_SHAPE, "vx_fuse_sum: bad output shape (C and the pitch are multiples of 4 floats)");
  if (a.act != VX_ACT_NONE && a.act != VX_ACT_RELU) VX_FAIL(VX_E_DTYPE, "vx_fuse_sum: act must be none or relu");
  for (int t = 0; t < a.nterms; ++t) {
    const vx_fuse_term& tm = a.term[t];
    if (!tm.x || tm.H <= 0 || tm.W <= 0 || tm.x_pitch % 4 || tm.x_pitch < a.C) VX_FAIL(VX_E_SHAPE, "vx_fuse_sum: term %d", t);
    if ((tm.scale == nullptr) != (tm.shift == nullptr)) VX_FAIL(VX_E_NULL, "vx_fuse_sum: scale / shift of term %d must come together", t);
  }
  const int C4 = a.C / 4;
  const int64_t rowp64 = (int64_t)a.OW * C4;
  if (a.OH > 65535 || a.N > 65535 || rowp64 * C4 >= (1ll << 31))
    VX_FAIL(VX_E_SHAPE, "vx_fuse_sum: at most 65535 rows / images and 2^31 / (C / 4) pieces per row (got %d x %d x %d x %d)", a.N, a.OH, a.OW, a.C);
  const int rowp = (int)rowp64;
  const unsigned mC4 = (unsigned)((1ull << 32) / (unsigned)C4) + 1u;
  const dim3 grid((unsigned)((rowp + 255) / 256), (unsigned)a.OH, (unsigned)a.N);
  hipStream_t s = (hipStream_t)stream;
  switch (a.nterms) {
    case 1: hipLaunchKernelGGL(fuse_sum_kernel<1>, grid, dim3(256), 0, s, a, mC4, rowp); break;
    case 2: hipLaunchKernelGGL(fuse_sum_kernel<2>, grid, dim3(256), 0, s, a, mC4, rowp); break;
    case 3: hipLaunchKernelGGL(fuse_sum_kernel<3>, grid, dim3(256), 0, s, a, mC4, rowp); break;
    default: hipLaunchKernelGGL(fuse_sum_kernel<4>, grid, dim3(256), 0, s, a, mC4, rowp); break;
  }
  VX_CHECK_LAUNCH("vx_fuse_sum");
  return VX_OK;
}

__global__ __launch_bounds__(256) void bilinear_nchw_kernel(const float* __restrict__ x, int x_pitch, int N, int H, int W,
                                                            int C, int OH, int OW, float* __restrict__ out,
                                                            const int32_t* __restrict__ dst, const int32_t* __restrict__ flip) {
  const float ry = (float)H / (float)OH, rx = (float)W / (float)OW;
  const int64_t total = (int64_t)N * OH * OW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), n = (int)(i / ((int64_t)OW * OH));
    int y0, y1, x0, x1;
    float ly, lx;
    bil_coord(oy, H, ry, y0, y1, ly);
    bil_coord(ox, W, rx, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p00 = x + (((size_t)n * H + y0) * W + x0) * x_pitch;
    const float* p01 = x + (((size_t)n * H + y0) * W + x1) * x_pitch;
    const float* p10 = x + (((size_t)n * H + y1) * W + x0) * x_pitch;
    const float* p11 = x + (((size_t)n * H + y1) * W + x1) * x_pitch;
    const int slot = dst ? dst[n] : n;
    // un-flip of a TTA view: bit 0 HorizontalFlip (test_2D.py:304-309: torch.flip(output, [-1])), bit 1 VerticalFlip
    // (the 8-view extension of BASELINE config 4; the reference ships 4 views)
    const int fl = flip ? flip[n] : 0;
    const int wx = (fl & 1) ? OW - 1 - ox : ox;
    const int wy = (fl & 2) ? OH - 1 - oy : oy;
    float* o = out + (size_t)slot * C * OH * OW + (size_t)wy * OW + wx;
    for (int c = 0; c < C; ++c)
      o[(size_t)c * OH * OW] = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
  }
}

// The same upsample with F.softmax(dim=1) of the upsampled logits applied before the store (test_2D.py:300-303 takes the
// softmax of every view's output): the full-resolution LOGITS are never written -- at 1024 x 512 x 19 classes x 32 views
// that is a 1.27 GB write and read, and the separate softmax pass another read.  Every value is computed exactly as
// bilinear_nchw_kernel followed by softmax_planar_kernel compute it (same expressions, same order): the same bits.
// (three sweeps over the classes, each re-evaluating the interpolation from the four low-resolution pixels -- they sit
// in L1 / L2 -- instead of a register array of C values: with the array fully unrolled for 32 classes hipcc hoisted all
// 128 loads and spilled 6 000 registers' worth; the sweeps need 40 registers and any C)
__global__ __launch_bounds__(256) void bilinear_softmax_nchw_kernel(const float* __restrict__ x, int x_pitch, int N, int H, int W,
                                                                    int C, int OH, int OW, float* __restrict__ out,
                                                                    const int32_t* __restrict__ dst, const int32_t* __restrict__ flip) {
  const float ry = (float)H / (float)OH, rx = (float)W / (float)OW;
  const int64_t total = (int64_t)N * OH * OW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), n = (int)(i / ((int64_t)OW * OH));
    int y0, y1, x0, x1;
    float ly, lx;
    bil_coord(oy, H, ry, y0, y1, ly);
    bil_coord(ox, W, rx, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p00 = x + (((size_t)n * H + y0) * W + x0) * x_pitch;
    const float* p01 = x + (((size_t)n * H + y0) * W + x1) * x_pitch;
    const float* p10 = x + (((size_t)n * H + y1) * W + x0) * x_pitch;
    const float* p11 = x + (((size_t)n * H + y1) * W + x1) * x_pitch;
    const int slot = dst ? dst[n] : n;
    const int fl = flip ? flip[n] : 0;
    const int wx = (fl & 1) ? OW - 1 - ox : ox;
    const int wy = (fl & 2) ? OH - 1 - oy : oy;
    float* o = out + (size_t)slot * C * OH * OW + (size_t)wy * OW + wx;
    auto val = [&](int c) { return hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]); };
    float m = val(0);
    for (int c = 1; c < C; ++c) m = fmaxf(m, val(c));
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(val(c) - m);
    const float inv = 1.f / den;
    for (int c = 0; c < C; ++c) o[(size_t)c * OH * OW] = expf(val(c) - m) * inv;
  }
}

// Up to 32 classes with the low-resolution pitch a multiple of 4: the four corner pixels arrive as 16-byte pieces and the
// C interpolated logits stay in registers (one exp per class).  The value of every logit is the same expression as above.
template <int C4>
__global__ __launch_bounds__(256) void bilinear_softmax_nchw_vec_kernel(const float* __restrict__ x, int x_pitch, int N, int H, int W,
                                                                        int C, int OH, int OW, float* __restrict__ out,
                                                                        const int32_t* __restrict__ dst, const int32_t* __restrict__ flip) {
  const float ry = (float)H / (float)OH, rx = (float)W / (float)OW;
  const int64_t total = (int64_t)N * OH * OW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), n = (int)(i / ((int64_t)OW * OH));
    int y0, y1, x0, x1;
    float ly, lx;
    bil_coord(oy, H, ry, y0, y1, ly);
    bil_coord(ox, W, rx, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const f32x4* p00 = reinterpret_cast<const f32x4*>(x + (((size_t)n * H + y0) * W + x0) * x_pitch);
    const f32x4* p01 = reinterpret_cast<const f32x4*>(x + (((size_t)n * H + y0) * W + x1) * x_pitch);
    const f32x4* p10 = reinterpret_cast<const f32x4*>(x + (((size_t)n * H + y1) * W + x0) * x_pitch);
    const f32x4* p11 = reinterpret_cast<const f32x4*>(x + (((size_t)n * H + y1) * W + x1) * x_pitch);
    const int slot = dst ? dst[n] : n;
    const int fl = flip ? flip[n] : 0;
    const int wx = (fl & 1) ? OW - 1 - ox : ox;
    const int wy = (fl & 2) ? OH - 1 - oy : oy;
    float* o = out + (size_t)slot * C * OH * OW + (size_t)wy * OW + wx;
    float v[4 * C4];
#pragma unroll
    for (int q = 0; q < C4; ++q) {
      const f32x4 a = p00[q], b = p01[q], c_ = p10[q], d = p11[q];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * q + j] = hy * (hx * a[j] + lx * b[j]) + ly * (hx * c_[j] + lx * d[j]);
    }
    float m = v[0];
#pragma unroll
    for (int c = 1; c < 4 * C4; ++c)
      if (c < C) m = fmaxf(m, v[c]);
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < 4 * C4; ++c)
      if (c < C) { v[c] = expf(v[c] - m); den += v[c]; }
    const float inv = 1.f / den;
#pragma unroll
    for (int c = 0; c < 4 * C4; ++c)
      if (c < C) o[(size_t)c * OH * OW] = v[c] * inv;
  }
}

extern "C" int vx_bilinear_softmax_nchw(const float* x, int x_pitch, int N, int H, int W, int C, int OH, int OW, float* out,
                                        const int32_t* dst, const int32_t* flip, vx_stream_t stream) {
  if (!x || !out) VX_FAIL(VX_E_NULL, "vx_bilinear_softmax_nchw: null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || x_pitch < C) VX_FAIL(VX_E_SHAPE, "vx_bilinear_softmax_nchw: bad shape");
  const int64_t total = (int64_t)N * OH * OW;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  const int c4 = (C + 3) / 4;
  const bool vec = x_pitch % 4 == 0 && x_pitch >= 4 * c4 && vx_aligned16(x) && c4 <= 8;
#define VX_BS_LAUNCH(K) hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_pitch, N, H, W, C, OH, OW, out, dst, flip)
  if (vec && c4 <= 1) VX_BS_LAUNCH(bilinear_softmax_nchw_vec_kernel<1>);
  else if (vec && c4 <= 2) VX_BS_LAUNCH(bilinear_softmax_nchw_vec_kernel<2>);
  else if (vec && c4 <= 5) VX_BS_LAUNCH(bilinear_softmax_nchw_vec_kernel<5>);
  else if (vec) VX_BS_LAUNCH(bilinear_softmax_nchw_vec_kernel<8>);
  else VX_BS_LAUNCH(bilinear_softmax_nchw_kernel);
#undef VX_BS_LAUNCH
  VX_CHECK_LAUNCH("vx_bilinear_softmax_nchw");
  return VX_OK;
}

extern "C" int vx_bilinear_nchw(const float* x, int x_pitch, int N, int H, int W, int C, int OH, int OW, float* out,
                                const int32_t* dst, const int32_t* flip, vx_stream_t stream) {
  if (!x || !out) VX_FAIL(VX_E_NULL, "vx_bilinear_nchw: null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || x_pitch < C) VX_FAIL(VX_E_SHAPE, "vx_bilinear_nchw: bad shape");
  const int64_t total = (int64_t)N * OH * OW;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(bilinear_nchw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_pitch, N, H, W, C, OH, OW,
                     out, dst, flip);
  VX_CHECK_LAUNCH("vx_bilinear_nchw");
  return VX_OK;
}
