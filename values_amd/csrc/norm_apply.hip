// K2 finalize + K2/K3/K4/K5/K7 apply.
//   vx_instnorm_finalize : per-(sample, channel) reduction of the conv epilogue's per-tile
//                          (sum, sumsq) partials in float64 -> mean, rstd (biased var, eps).
//   vx_norm_act_drop_pool: y = Dropout(LeakyReLU((x - mean) * rstd)); writes y with an arbitrary
//                          channel pitch/offset (the skip half of the decoder concat buffer) and the
//                          2x2x2 max-pool of y.  Pure streaming: 16-byte vectors, HBM-bound.
#include "common.h"

__global__ __launch_bounds__(64) void instnorm_finalize_kernel(const float* __restrict__ partial, int ntiles, int C,
                                                               double inv_count, float eps, float* __restrict__ mean,
                                                               float* __restrict__ rstd) {
  const int n = blockIdx.x / C, c = blockIdx.x % C;
  const float* p = partial + ((size_t)n * ntiles * C + c) * 2;
  double s = 0.0, q = 0.0;
  for (int t = threadIdx.x; t < ntiles; t += 64) {
    s += (double)p[(size_t)t * C * 2 + 0];
    q += (double)p[(size_t)t * C * 2 + 1];
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  if (threadIdx.x == 0) {
    const double mu = s * inv_count;
    double var = q * inv_count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[blockIdx.x] = (float)mu;
    rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

extern "C" int vx_instnorm_finalize(const float* stats_partial, int N, int ntiles, int C, int64_t nvox, float eps,
                                    float* mean, float* rstd, vx_stream_t stream) {
  if (!stats_partial || !mean || !rstd) VX_FAIL(VX_E_NULL, "vx_instnorm_finalize: null pointer");
  if (N <= 0 || ntiles <= 0 || C <= 0 || nvox <= 0) VX_FAIL(VX_E_SHAPE, "vx_instnorm_finalize: empty");
  hipLaunchKernelGGL(instnorm_finalize_kernel, dim3((unsigned)(N * C)), dim3(64), 0, (hipStream_t)stream,
                     stats_partial, ntiles, C, 1.0 / (double)nvox, eps, mean, rstd);
  VX_CHECK_LAUNCH("vx_instnorm_finalize");
  return VX_OK;
}

// One thread = one 2x2x2 voxel block x 4 channels (8 x 16-byte loads, 8 stores, 1 pooled store) when
// POOL; one voxel x 4 channels otherwise.
template <bool POOL>
__global__ __launch_bounds__(256) void norm_act_drop_pool_kernel(vx_norm_args a, int64_t total) {
  const int C4 = a.C / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int q = r % C4; r /= C4;
    const int c = q * 4;
    int bx, by, bz, n;
    if (POOL) {
      bx = r % (a.W / 2); r /= (a.W / 2);
      by = r % (a.H / 2); r /= (a.H / 2);
      bz = r % (a.D / 2); r /= (a.D / 2);
    } else {
      bx = r % a.W; r /= a.W;
      by = r % a.H; r /= a.H;
      bz = r % a.D; r /= a.D;
    }
    n = (int)r;
    f32x4 mu = (f32x4){0.f, 0.f, 0.f, 0.f}, rs = (f32x4){1.f, 1.f, 1.f, 1.f};
    if (a.mean) {
      mu = *reinterpret_cast<const f32x4*>(a.mean + (size_t)n * a.C + c);
      rs = *reinterpret_cast<const f32x4*>(a.rstd + (size_t)n * a.C + c);
    }
    const uint32_t dkey = vx_drop_key(a.drop_seed, a.drop_layer, (uint32_t)n);
    f32x4 mx = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    constexpr int NV = POOL ? 8 : 1;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int z = POOL ? bz * 2 + (k >> 2) : bz;
      const int y = POOL ? by * 2 + ((k >> 1) & 1) : by;
      const int x = POOL ? bx * 2 + (k & 1) : bx;
      const size_t vox = ((size_t)(n * a.D + z) * a.H + y) * a.W + x;
      f32x4 v = *reinterpret_cast<const f32x4*>(a.x + vox * a.x_pitch + c);
      v = (v - mu) * rs;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = vx_act(v[j], a.act);
      if (a.drop_mode == VX_DROP_HASH) {
        const uint32_t e = (uint32_t)(((z * a.H + y) * a.W + x) * a.C + c);
        const uint32_t bits = vx_drop_bits4(dkey, e);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((bits >> j) & 1u) ? 2.f * v[j] : 0.f;
      } else if (a.drop_mode == VX_DROP_MASK) {
        const uint32_t mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + vox * a.C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * v[j] : 0.f;
      }
      *reinterpret_cast<f32x4*>(a.out + vox * a.out_pitch + a.out_coff + c) = v;
      if (POOL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mx[j] = fmaxf(mx[j], v[j]);
      }
    }
    if (POOL) {
      const size_t pv = ((size_t)(n * (a.D / 2) + bz) * (a.H / 2) + by) * (a.W / 2) + bx;
      *reinterpret_cast<f32x4*>(a.pool_out + pv * a.pool_pitch + c) = mx;
    }
  }
}

extern "C" int vx_norm_act_drop_pool(const vx_norm_args* ap, vx_stream_t stream) {
  if (!ap) VX_FAIL(VX_E_NULL, "vx_norm_act_drop_pool: null args");
  const vx_norm_args& a = *ap;
  if (!a.x || !a.out) VX_FAIL(VX_E_NULL, "vx_norm_act_drop_pool: null tensor");
  if ((a.mean == nullptr) != (a.rstd == nullptr)) VX_FAIL(VX_E_NULL, "vx_norm_act_drop_pool: mean/rstd must come together");
  if (a.N <= 0 || a.D <= 0 || a.H <= 0 || a.W <= 0 || a.C <= 0 || a.C % 4)
    VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: bad shape (C must be a multiple of 4)");
  if (a.x_pitch % 4 || a.out_pitch % 4 || a.out_coff % 4 || a.x_pitch < a.C || a.out_pitch < a.out_coff + a.C)
    VX_FAIL(VX_E_ALIGN, "vx_norm_act_drop_pool: pitches/offsets must be multiples of 4 floats");
  if (a.drop_mode == VX_DROP_MASK && !a.drop_mask) VX_FAIL(VX_E_NULL, "vx_norm_act_drop_pool: mask mode without mask");
  if ((int64_t)a.D * a.H * a.W * a.C >= (1ll << 32)) VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: sample too large");
  hipStream_t s = (hipStream_t)stream;
  if (a.pool_out) {
    if (a.D % 2 || a.H % 2 || a.W % 2) VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: pooling needs even dims");
    if (a.pool_pitch % 4 || a.pool_pitch < a.C) VX_FAIL(VX_E_ALIGN, "vx_norm_act_drop_pool: pool pitch");
    const int64_t total = (int64_t)a.N * (a.D / 2) * (a.H / 2) * (a.W / 2) * (a.C / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(norm_act_drop_pool_kernel<true>, dim3(blocks), dim3(256), 0, s, a, total);
  } else {
    const int64_t total = (int64_t)a.N * a.D * a.H * a.W * (a.C / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(norm_act_drop_pool_kernel<false>, dim3(blocks), dim3(256), 0, s, a, total);
  }
  VX_CHECK_LAUNCH("vx_norm_act_drop_pool");
  return VX_OK;
}
