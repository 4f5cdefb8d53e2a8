// K1 (first layer): 3x3x3 convolution with ONE input channel (contr_1_1, unet3D_module.py:36-42).
// 27*Cout MACs per voxel against 4 B read + 4*Cout B written: HBM-bound, so plain VALU FMAs with
// wave-uniform weights (scalar loads) and an LDS halo tile; no matrix cores.  The kernel also
// resolves the pass scheduling of predict_cases (test_3D.py:417-482): sample n reads volume src[n]
// (MC-dropout: n / repeat, all T samples of a volume read the same input) through the flip code of
// its TTA view.
#include "s16_common.h"

// Round 3: a thread computes a column of FOUR voxels along z and walks the nine (ky, kx) offsets: per offset the three
// kz weight vectors come out of LDS once (broadcast 16-byte reads) and serve the column's six input planes, so a voxel
// costs 27 LDS reads instead of 243 (the round-2 kernel read one weight per FMA: 0.16 of the HBM roof); the InstanceNorm
// sums are accumulated over the column before the cross-lane reduction (a quarter of the shuffles per voxel).
template <int COUT>
__global__ __launch_bounds__(256) void conv3d_k3_c1_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ out, int out_pitch, int N, int D, int H, int W,
                                                           int repeat, const int32_t* __restrict__ src,
                                                           const int32_t* __restrict__ flip, float* __restrict__ stats_partial,
                                                           int tiles_x, int tiles_y, int tiles_z) {
  constexpr int TX = 64, TY = 4, TZ = 4;  // 1024 voxels, a z-column of four per thread
  constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2;
  __shared__ float s_in[HZ * HY * HX];
  __shared__ __attribute__((aligned(16))) float s_w[27 * COUT];
  __shared__ float s_red[16][COUT][2];

  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y; t /= tiles_y;
  const int tz = t % tiles_z; t /= tiles_z;
  const int n = t;
  const int v = src ? src[n] : n / repeat;
  const int f = flip ? flip[n] : 0;
  const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
  const float* in_v = in + (size_t)v * D * H * W;

  for (int idx = tid; idx < HZ * HY * HX; idx += 256) {
    const int hx = idx % HX, hy = (idx / HX) % HY, hz = idx / (HX * HY);
    int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
    float val = 0.f;
    if ((unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D) {
      if (f & 1) gz = D - 1 - gz;
      if (f & 2) gy = H - 1 - gy;
      if (f & 4) gx = W - 1 - gx;
      val = in_v[((size_t)gz * H + gy) * W + gx];
    }
    s_in[idx] = val;
  }
  // weights: torch (COUT,1,3,3,3) -> s_w[tap][cout]
  for (int idx = tid; idx < 27 * COUT; idx += 256) {
    const int co = idx % COUT, tap = idx / COUT;
    s_w[idx] = w[co * 27 + tap];
  }
  __syncthreads();

  const int lx = tid % TX, ly = tid / TX;
  float acc[TZ][COUT];
#pragma unroll
  for (int z = 0; z < TZ; ++z)
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[z][c] = bias[c];
  // (a run-time loop on purpose: fully unrolled, hipcc hoists all 27 x Cout weights into registers -- 360 VGPRs)
#pragma unroll 1
  for (int kk = 0; kk < 9; ++kk) {
    const int ky = kk / 3, kx = kk - 3 * ky;
    float wk[3][COUT];                 // the three kz taps of this (ky, kx), all output channels
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int c = 0; c < COUT; c += 4) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(&s_w[(kz * 9 + kk) * COUT + c]);
        wk[kz][c] = w4[0]; wk[kz][c + 1] = w4[1]; wk[kz][c + 2] = w4[2]; wk[kz][c + 3] = w4[3];
      }
    const float* sp = s_in + (ly + ky) * HX + lx + kx;
    // input plane p (0 .. TZ + 1) feeds output plane z through tap kz = p - z
#pragma unroll
    for (int p = 0; p < HZ; ++p) {
      const float xv = sp[p * HY * HX];
#pragma unroll
      for (int z = 0; z < TZ; ++z) {
        const int kz = p - z;
        if (kz < 0 || kz > 2) continue;
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[z][c] = fmaf(wk[kz][c], xv, acc[z][c]);
      }
    }
  }

  const int gx = x0 + lx, gy = y0 + ly;
  float ssum[COUT], ssq[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; }
#pragma unroll
  for (int z = 0; z < TZ; ++z) {
    const int gz = z0 + z;
    const bool valid = gx < W && gy < H && gz < D;
    if (valid) {
      float* o = out + (((size_t)(n * D + gz) * H + gy) * W + gx) * out_pitch;
#pragma unroll
      for (int c = 0; c < COUT; c += 4)
        *reinterpret_cast<f32x4*>(o + c) = (f32x4){acc[z][c], acc[z][c + 1], acc[z][c + 2], acc[z][c + 3]};
#pragma unroll
      for (int c = 0; c < COUT; ++c) { ssum[c] += acc[z][c]; ssq[c] = fmaf(acc[z][c], acc[z][c], ssq[c]); }
    }
  }
  if (stats_partial) {
    // sums over the workgroup: DPP row rotations inside each 16-lane row (vector ALU, no trip through the LDS queue as
    // __shfl_xor takes), then the 16 row sums of the four waves through LDS
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      float s = ssum[c], q = ssq[c];
#pragma unroll
      for (int rot = 8; rot >= 1; rot >>= 1) { s += vx_row_ror(s, rot); q += vx_row_ror(q, rot); }
      if ((lane & 15) == 0) { s_red[wave * 4 + (lane >> 4)][c][0] = s; s_red[wave * 4 + (lane >> 4)][c][1] = q; }
    }
    __syncthreads();
    if (tid < COUT) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int wv = 0; wv < 16; ++wv) { s += s_red[wv][tid][0]; q += s_red[wv][tid][1]; }
      const int ntiles = tiles_x * tiles_y * tiles_z;
      const int tile = blockIdx.x % ntiles;
      float* dst = stats_partial + (((size_t)n * ntiles + tile) * COUT + tid) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

extern "C" int vx_conv3d_k3_c1_tiles(int D, int H, int W) { return ((W + 63) / 64) * ((H + 3) / 4) * ((D + 3) / 4); }

extern "C" int vx_conv3d_k3_c1(const float* in, const float* w_torch, const float* bias, float* out, int out_pitch,
                               int N, int D, int H, int W, int Cout, int repeat, const int32_t* src,
                               const int32_t* flip, float* stats_partial, vx_stream_t stream) {
  if (!in || !w_torch || !bias || !out) VX_FAIL(VX_E_NULL, "vx_conv3d_k3_c1: null pointer");
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3_c1: empty tensor");
  if (repeat <= 0) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3_c1: repeat must be >= 1");
  if (out_pitch < Cout || out_pitch % 4 || !vx_aligned16(out)) VX_FAIL(VX_E_ALIGN, "vx_conv3d_k3_c1: out pitch/alignment");
  const int tiles_x = (W + 63) / 64, tiles_y = (H + 3) / 4, tiles_z = (D + 3) / 4;
  dim3 grid((unsigned)(tiles_x * tiles_y * tiles_z * N));
  hipStream_t s = (hipStream_t)stream;
#define VX_C1(CO)                                                                                                  \
  vx_note_kernel("conv3d_k3_c1_kernel<" #CO ">");                                                                  \
  hipLaunchKernelGGL(conv3d_k3_c1_kernel<CO>, grid, dim3(256), 0, s, in, w_torch, bias, out, out_pitch, N, D, H, W, \
                     repeat, src, flip, stats_partial, tiles_x, tiles_y, tiles_z)
  switch (Cout) {
    case 8: VX_C1(8); break;
    case 16: VX_C1(16); break;
    case 32: VX_C1(32); break;
    default: VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3_c1: Cout=%d unsupported (8, 16, 32)", Cout);
  }
#undef VX_C1
  VX_CHECK_LAUNCH("vx_conv3d_k3_c1");
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// in_channels > 1 (unet3D_module.py:8-35): the first conv then runs on the general 3x3x3 kernels, which take
// channels-last tensors with Cin % 8 == 0.  This kernel lays the reference's (V, Cin, D, H, W) input out as
// [N][D][H][W][8] (channels Cin .. 7 zero; the weights are zero-padded to match at pack time), resolving the
// per-sample source volume and TTA flip exactly as conv3d_k3_c1_kernel does.
__global__ __launch_bounds__(256) void pack_input_cl8_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int Cin,
                                                             int D, int H, int W, int repeat, const int32_t* __restrict__ src,
                                                             const int32_t* __restrict__ flip) {
  const int64_t nvox = (int64_t)D * H * W, total = (int64_t)N * nvox;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / nvox);
    const int64_t v = i - (int64_t)n * nvox;
    int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((int64_t)W * H));
    const int vol = src ? src[n] : n / repeat;
    const int fl = flip ? flip[n] : 0;
    if (fl & 1) z = D - 1 - z;
    if (fl & 2) y = H - 1 - y;
    if (fl & 4) x = W - 1 - x;
    const float* p = in + ((size_t)vol * Cin) * nvox + ((size_t)z * H + y) * W + x;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < Cin) lo[c] = p[(size_t)c * nvox];
      if (c + 4 < Cin) hi[c] = p[(size_t)(c + 4) * nvox];
    }
    reinterpret_cast<f32x4*>(out + (size_t)i * 8)[0] = lo;
    reinterpret_cast<f32x4*>(out + (size_t)i * 8)[1] = hi;
  }
}

extern "C" int vx_pack_input_cl8(const float* in, float* out, int N, int Cin, int D, int H, int W, int repeat,
                                 const int32_t* src, const int32_t* flip, vx_stream_t stream) {
  if (!in || !out) VX_FAIL(VX_E_NULL, "vx_pack_input_cl8: null pointer");
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || repeat <= 0) VX_FAIL(VX_E_SHAPE, "vx_pack_input_cl8: empty tensor");
  if (Cin < 1 || Cin > 8) VX_FAIL(VX_E_SHAPE, "vx_pack_input_cl8: in_channels %d (1 .. 8)", Cin);
  if (!vx_aligned16(out)) VX_FAIL(VX_E_ALIGN, "vx_pack_input_cl8: output alignment");
  const int64_t total = (int64_t)N * D * H * W;
  int bx = (int)((total + 255) / 256);
  if (bx > 16384) bx = 16384;
  vx_note_kernel("pack_input_cl8_kernel");
  hipLaunchKernelGGL(pack_input_cl8_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, in, out, N, Cin, D, H, W, repeat, src, flip);
  VX_CHECK_LAUNCH("vx_pack_input_cl8");
  return VX_OK;
}
