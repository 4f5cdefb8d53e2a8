// K2 finalize + K2/K3/K4/K5/K7 apply.
//   vx_instnorm_finalize : per-(sample, channel) reduction of the conv epilogue's per-tile
//                          (sum, sumsq) partials in float64 -> mean, rstd (biased var, eps).
//   vx_norm_act_drop_pool: y = Dropout(LeakyReLU((x - mean) * rstd)); writes y with an arbitrary
//                          channel pitch/offset (the skip half of the decoder concat buffer) and the
//                          2x2x2 max-pool of y.  Pure streaming: 16-byte vectors, HBM-bound.
#include "s16_common.h"

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
  vx_note_kernel("instnorm_finalize_kernel");
  hipLaunchKernelGGL(instnorm_finalize_kernel, dim3((unsigned)(N * C)), dim3(64), 0, (hipStream_t)stream,
                     stats_partial, ntiles, C, 1.0 / (double)nvox, eps, mean, rstd);
  VX_CHECK_LAUNCH("vx_instnorm_finalize");
  return VX_OK;
}

// Round 5: the statistics of ONE sample reduced by the consuming streaming pass itself (vx_stat_src): every workgroup of
// sample n sums the sample's partials (tiles x C x 2 floats, L2-resident) in float64 -- L = 256 / C threads per channel over
// strided tiles, then one thread per channel over the L partial sums -- and keeps mean / rstd in LDS; the workgroups with
// blockIdx.x == 0 also write them out for later readers.  Same formula as instnorm_finalize_kernel (biased variance, eps inside
// the root); the float64 summation ORDER differs, i.e. the float results agree to the last bit except on a rounding boundary.
// The four instnorm_finalize launches in front of these passes (first block's pre-split, vx_pool_finish_z, the two normalise +
// pool passes of the deep levels) are gone: 33 -> 29 launches per forward.
struct StatSrc { const float* partial; int tiles; double inv_count; float eps; float* mean_out; float* rstd_out; };
constexpr int VX_STAT_MAXC = 512;
__device__ __forceinline__ void vx_block_instnorm(const StatSrc& st, int n, int C, bool write, float* s_mu, float* s_rs, double* s_part) {
  const float* p = st.partial + (size_t)n * st.tiles * C * 2;
  const int tid = threadIdx.x;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int cw = C - c0 < 256 ? C - c0 : 256;
    int L = 1;
    while (2 * L * cw <= 256) L *= 2;
    const int c = tid % cw, sub = tid / cw;
    double s = 0.0, q = 0.0;
    if (sub < L) {
      // (four tiles' loads in flight per trip: the trips are dependent L2 round trips)
      const size_t tstride = (size_t)L * C * 2;
      const float* e = p + ((size_t)sub * C + c0 + c) * 2;
      int t = sub;
      for (; t + 3 * L < st.tiles; t += 4 * L, e += 4 * tstride) {
        const f32x2 v0 = *reinterpret_cast<const f32x2*>(e), v1 = *reinterpret_cast<const f32x2*>(e + tstride);
        const f32x2 v2 = *reinterpret_cast<const f32x2*>(e + 2 * tstride), v3 = *reinterpret_cast<const f32x2*>(e + 3 * tstride);
        s += (double)v0[0]; q += (double)v0[1];
        s += (double)v1[0]; q += (double)v1[1];
        s += (double)v2[0]; q += (double)v2[1];
        s += (double)v3[0]; q += (double)v3[1];
      }
      for (; t < st.tiles; t += L, e += tstride) {
        const f32x2 v0 = *reinterpret_cast<const f32x2*>(e);
        s += (double)v0[0];
        q += (double)v0[1];
      }
    }
    s_part[tid] = s;
    s_part[256 + tid] = q;
    __syncthreads();
    if (tid < cw) {
      double S = 0.0, Q = 0.0;
      for (int k = 0; k < L; ++k) { S += s_part[k * cw + tid]; Q += s_part[256 + k * cw + tid]; }
      const double mu = S * st.inv_count;
      double var = Q * st.inv_count - mu * mu;
      if (var < 0.0) var = 0.0;
      const float m = (float)mu, r = (float)(1.0 / sqrt(var + (double)st.eps));
      s_mu[c0 + tid] = m;
      s_rs[c0 + tid] = r;
      if (write && st.mean_out) st.mean_out[(size_t)n * C + c0 + tid] = m;
      if (write && st.rstd_out) st.rstd_out[(size_t)n * C + c0 + tid] = r;
    }
    __syncthreads();
  }
}
static int vx_stat_src_check(const vx_stat_src* st, int C, const char* who, StatSrc* o) {
  if (!st || !st->stats_partial) VX_FAIL(VX_E_NULL, "%s: null statistics source", who);
  if (st->tiles <= 0 || st->count <= 0 || C > VX_STAT_MAXC) VX_FAIL(VX_E_SHAPE, "%s: %d tiles, %lld values per channel, C = %d (<= %d)", who, st->tiles, (long long)st->count, C, VX_STAT_MAXC);
  o->partial = st->stats_partial; o->tiles = st->tiles; o->inv_count = 1.0 / (double)st->count; o->eps = st->eps;
  o->mean_out = st->mean_out; o->rstd_out = st->rstd_out;
  return VX_OK;
}

// One thread = one 16-byte piece (4 channels of one voxel) of a full-resolution ROW; consecutive lanes hold
// consecutive pieces, so every load/store instruction of a wave is one contiguous 1 KiB segment of the row.
// POOL: the thread handles the same piece of the 4 rows (2z+dz, 2y+dy) of a 2x2 row bundle, takes their max,
// exchanges with the lane that holds the x-neighbour voxel (C/4 lanes away: one cross-lane shuffle) and the
// even-x lanes store the pooled piece.  Input sample = n / x_repeat: the T MC-dropout samples of a volume share
// the first layer's conv output and statistics (same input, dropout comes after), so contr_1_1 is computed once
// per volume and only this kernel fans it out into T differently-dropped copies.
// Index decode: blockIdx.y = sample, the piece index within the sample is 32-bit and is split with exact
// multiply-high divisions (magic = 2^32 / d + 1, exact while index * d < 2^32 -- checked by the launcher); the
// original int64 % and / chain cost ~300 VALU instructions per piece and made the write-only fan-out kernel
// instruction-bound at 2.7 TB/s.
struct NormDecode { unsigned per_sample, mPW, mC4, mH; };   // magic 0 = divisor 1
// fp16 range word (vx_conv3d_args.range_flag): one atomic per wave, and only for magnitudes within a factor two of the limit
__device__ __forceinline__ void vx_range_report(uint32_t* flag, float rmax) {
  if (!flag) return;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, off, 64));
  if ((threadIdx.x & 63) == 0 && !(rmax < 32768.f)) atomicMax(flag, __float_as_uint(rmax));
}
__device__ __forceinline__ unsigned vx_magic_div(unsigned n, unsigned m) { return m ? __umulhi(n, m) : n; }
// WIDE (pooling with more than 128 channels): the x-neighbour's piece would sit in another wave, so a thread takes
// BOTH voxels of an x-pair (a row then has W/2 * C/4 work items) and no shuffle is needed.
template <bool POOL, bool WIDE = false, bool FOLD = false>
__global__ __launch_bounds__(256) void norm_act_drop_pool_kernel(vx_norm_args a, int x_repeat, NormDecode dc, StatSrc st) {
  const int C4 = a.C / 4;
  const int PW = (WIDE ? a.W / 2 : a.W) * C4;  // work items per row
  const int n = blockIdx.y;
  const int RH = POOL ? a.H / 2 : a.H;   // rows (row bundles) per z
  alignas(16) __shared__ float s_mu[FOLD ? VX_STAT_MAXC : 4], s_rs[FOLD ? VX_STAT_MAXC : 4];   // (read as 16-byte vectors)
  __shared__ double s_part[FOLD ? 512 : 1];
  if constexpr (FOLD) vx_block_instnorm(st, n / x_repeat, a.C, blockIdx.x == 0, s_mu, s_rs, s_part);
  float rmax = 0.f;   // largest |value| stored: range guard of the split-fp16 consumers when nothing normalises (a.range_flag)
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < dc.per_sample; i += gridDim.x * 256u) {
    const unsigned row = vx_magic_div(i, dc.mPW);       // i / PW
    const int p = (int)(i - row * (unsigned)PW);
    const int xq = (int)vx_magic_div((unsigned)p, dc.mC4), c = (p - xq * C4) * 4;
    const int bz = (int)vx_magic_div(row, dc.mH);       // row / RH
    const int by = (int)(row - (unsigned)bz * (unsigned)RH);
    const int ns = n / x_repeat;
    f32x4 mu = (f32x4){0.f, 0.f, 0.f, 0.f}, rs = (f32x4){1.f, 1.f, 1.f, 1.f};
    if constexpr (FOLD) {
      mu = *reinterpret_cast<const f32x4*>(s_mu + c);
      rs = *reinterpret_cast<const f32x4*>(s_rs + c);
    } else if (a.mean) {
      mu = *reinterpret_cast<const f32x4*>(a.mean + (size_t)ns * a.C + c);
      rs = *reinterpret_cast<const f32x4*>(a.rstd + (size_t)ns * a.C + c);
    }
    const vx_dkey dkey = vx_drop_key(vx_seed_of(a, a.drop_seed), a.drop_layer, (uint32_t)n);
    f32x4 mx = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    constexpr int NR = POOL ? (WIDE ? 8 : 4) : 1;
    f32x4 v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int z = POOL ? bz * 2 + ((k >> 1) & 1) : bz;
      const int y = POOL ? by * 2 + (k & 1) : by;
      const int x = WIDE ? 2 * xq + (k >> 2) : xq;
      if (a.x_xblk) {   // the raw tensor sits in a concat buffer's half (dense blocks of xb voxels)
        const size_t srow = (size_t)(ns * a.D + z) * a.H + y;
        const int xb = a.x_xblk;
        v[k] = *reinterpret_cast<const f32x4*>(a.x + srow * (2 * (size_t)a.W * a.C) + ((x / xb) * 2 + a.x_half) * xb * a.C +
                                               (x % xb) * a.C + c);
      } else {
        const size_t svox = ((size_t)(ns * a.D + z) * a.H + y) * a.W + x;
        v[k] = *reinterpret_cast<const f32x4*>(a.x + svox * a.x_pitch + c);
      }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int z = POOL ? bz * 2 + ((k >> 1) & 1) : bz;
      const int y = POOL ? by * 2 + (k & 1) : by;
      const int x = WIDE ? 2 * xq + (k >> 2) : xq;
      const size_t vox = ((size_t)(n * a.D + z) * a.H + y) * a.W + x;
      f32x4 t = (v[k] - mu) * rs;
      if (a.act == VX_ACT_LRELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = fmaxf(t[j], 0.01f * t[j]);
      } else if (a.act == VX_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = fmaxf(t[j], 0.f);
      }
      if (a.drop_mode == VX_DROP_HASH) {
        const uint32_t e = (uint32_t)(((z * a.H + y) * a.W + x) * a.C + c);
        const uint32_t bits = vx_drop_bits4(dkey, e);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);  // keep ? 2.0f : 0.0f
      } else if (a.drop_mode == VX_DROP_MASK) {
        const uint32_t mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + vox * a.C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * t[j] : 0.f;
      }
      if (a.range_flag) rmax = fmaxf(fmaxf(rmax, fmaxf(fabsf(t[0]), fabsf(t[1]))), fmaxf(fabsf(t[2]), fabsf(t[3])));
      if (!a.out) {
        // pooled tensor only
      } else if (a.out_xblk) {
        // concat buffer [N][D][H][W/xb][2][xb][C]: this kernel's half as dense blocks of xb voxels
        const size_t row = (size_t)(n * a.D + z) * a.H + y;
        const int xb = a.out_xblk;
        *reinterpret_cast<f32x4*>(a.out + row * (2 * (size_t)a.W * a.C) + ((x / xb) * 2 + a.out_half) * xb * a.C +
                                  (x % xb) * a.C + c) = t;
      } else {
        *reinterpret_cast<f32x4*>(a.out + vox * a.out_pitch + a.out_coff + c) = t;
      }
      if (POOL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mx[j] = fmaxf(mx[j], t[j]);
      }
    }
    if (POOL && WIDE) {
      const size_t pv = ((size_t)(n * (a.D / 2) + bz) * (a.H / 2) + by) * (a.W / 2) + xq;
      *reinterpret_cast<f32x4*>(a.pool_out + pv * a.pool_pitch + c) = mx;
    } else if (POOL) {
      const int x = xq;
      // x-neighbour voxel's piece with the same channels sits C4 pieces (= lanes) away
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(mx[j], C4, 64);
      if ((x & 1) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mx[j] = fmaxf(mx[j], o[j]);
        const size_t pv = ((size_t)(n * (a.D / 2) + bz) * (a.H / 2) + by) * (a.W / 2) + (x >> 1);
        *reinterpret_cast<f32x4*>(a.pool_out + pv * a.pool_pitch + c) = mx;
      }
    }
  }
  vx_range_report(a.range_flag, rmax);
}

// Fan-out variant (MC-dropout first layer: x_repeat = T samples share one source volume, no pooling): one thread
// per SOURCE piece -- one load and one normalise + activation, then T dropout patterns and T fire-and-forget
// stores.  The per-sample kernel above re-read and re-normalised the source T times with one 16-byte load in
// flight per thread: latency-bound at 3.2 TB/s written; this one streams at the write roof.
__global__ __launch_bounds__(256) void norm_act_drop_fanout_kernel(vx_norm_args a, int x_repeat, NormDecode dc) {
  const int C4 = a.C / 4;
  const int PW = a.W * C4;
  const int ns = blockIdx.y;
  const size_t out_sample = (size_t)a.D * a.H * a.W * (a.out_xblk ? 2 * a.C : a.out_pitch);
  const int xs = a.out_xblk ? __builtin_ctz((unsigned)a.out_xblk) : 0;
  float rmax = 0.f;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < dc.per_sample; i += gridDim.x * 256u) {
    const unsigned row = vx_magic_div(i, dc.mPW);
    const int p = (int)(i - row * (unsigned)PW);
    const int x = (int)vx_magic_div((unsigned)p, dc.mC4), c = (p - x * C4) * 4;
    const int z = (int)vx_magic_div(row, dc.mH);
    const int y = (int)(row - (unsigned)z * (unsigned)a.H);
    f32x4 mu = (f32x4){0.f, 0.f, 0.f, 0.f}, rs = (f32x4){1.f, 1.f, 1.f, 1.f};
    if (a.mean) {
      mu = *reinterpret_cast<const f32x4*>(a.mean + (size_t)ns * a.C + c);
      rs = *reinterpret_cast<const f32x4*>(a.rstd + (size_t)ns * a.C + c);
    }
    const size_t svox = ((size_t)(ns * a.D + z) * a.H + y) * a.W + x;
    f32x4 t0 = (*reinterpret_cast<const f32x4*>(a.x + svox * a.x_pitch + c) - mu) * rs;
    if (a.act == VX_ACT_LRELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) t0[j] = fmaxf(t0[j], 0.01f * t0[j]);
    } else if (a.act == VX_ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) t0[j] = fmaxf(t0[j], 0.f);
    }
    // (the dropout's factor 2 included: what the T samples store is at most twice this)
    if (a.range_flag) rmax = fmaxf(fmaxf(rmax, 2.f * fmaxf(fabsf(t0[0]), fabsf(t0[1]))), 2.f * fmaxf(fabsf(t0[2]), fabsf(t0[3])));
    const uint32_t e = (uint32_t)(((z * a.H + y) * a.W + x) * a.C + c);
    size_t off;   // float offset of the piece within a sample's output
    if (a.out_xblk)
      off = ((size_t)z * a.H + y) * (2 * (size_t)a.W * a.C) + ((((x >> xs) * 2 + a.out_half) << xs) + (x & (a.out_xblk - 1))) * a.C + c;
    else
      off = (((size_t)z * a.H + y) * a.W + x) * a.out_pitch + a.out_coff + c;
    for (int k = 0; k < x_repeat; ++k) {
      const int n = ns * x_repeat + k;
      f32x4 t = t0;
      if (a.drop_mode == VX_DROP_HASH) {
        const uint32_t bits = vx_drop_bits4(vx_drop_key(vx_seed_of(a, a.drop_seed), a.drop_layer, (uint32_t)n), e);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);
      } else if (a.drop_mode == VX_DROP_MASK) {
        const uint32_t mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + (size_t)n * a.D * a.H * a.W * a.C + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * t[j] : 0.f;
      }
      *reinterpret_cast<f32x4*>(a.out + (size_t)n * out_sample + off) = t;
    }
  }
  vx_range_report(a.range_flag, rmax);
}

static int norm_act_drop_pool_impl(const vx_norm_args* ap, int x_repeat, const vx_stat_src* stsrc, vx_stream_t stream);
extern "C" int vx_norm_act_drop_pool(const vx_norm_args* ap, vx_stream_t stream) {
  return norm_act_drop_pool_impl(ap, 1, nullptr, stream);
}
extern "C" int vx_norm_act_drop_pool_bcast(const vx_norm_args* ap, int x_repeat, vx_stream_t stream) {
  return norm_act_drop_pool_impl(ap, x_repeat, nullptr, stream);
}
extern "C" int vx_norm_act_drop_pool_stats(const vx_norm_args* ap, const vx_stat_src* st, vx_stream_t stream) {
  if (!st) VX_FAIL(VX_E_NULL, "vx_norm_act_drop_pool_stats: null statistics source");
  return norm_act_drop_pool_impl(ap, 1, st, stream);
}

static int norm_act_drop_pool_impl(const vx_norm_args* ap, int x_repeat, const vx_stat_src* stsrc, vx_stream_t stream) {
  if (!ap) VX_FAIL(VX_E_NULL, "vx_norm_act_drop_pool: null args");
  StatSrc st = {};
  if (stsrc) {
    const int rc = vx_stat_src_check(stsrc, ap->C, "vx_norm_act_drop_pool_stats", &st);
    if (rc != VX_OK) return rc;
    if (ap->mean || ap->rstd) VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool_stats: the statistics come from the source, mean / rstd must be NULL");
  }
  if (x_repeat < 1) VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: x_repeat must be >= 1");
  const vx_norm_args& a = *ap;
  if (!a.x || (!a.out && !a.pool_out)) VX_FAIL(VX_E_NULL, "vx_norm_act_drop_pool: null tensor");
  if (a.x_xblk && ((a.x_xblk != 1 && a.x_xblk != 2 && a.x_xblk != 4) || a.W % a.x_xblk || (a.x_half != 0 && a.x_half != 1)))
    VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: bad concat input (xblk=%d, half=%d, W=%d)", a.x_xblk, a.x_half, a.W);
  if (a.x_xblk && (x_repeat > 1 || !a.pool_out)) VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: a concat input goes with pooling, x_repeat 1");
  if ((a.mean == nullptr) != (a.rstd == nullptr)) VX_FAIL(VX_E_NULL, "vx_norm_act_drop_pool: mean/rstd must come together");
  if (a.N <= 0 || a.D <= 0 || a.H <= 0 || a.W <= 0 || a.C <= 0 || a.C % 4)
    VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: bad shape (C must be a multiple of 4)");
  if (!a.x_xblk && (a.x_pitch % 4 || a.x_pitch < a.C)) VX_FAIL(VX_E_ALIGN, "vx_norm_act_drop_pool: input pitch");
  if (!a.out) {
    // pooled tensor only
  } else if (a.out_xblk) {
    if ((a.out_xblk != 1 && a.out_xblk != 2 && a.out_xblk != 4) || a.W % a.out_xblk || (a.out_half != 0 && a.out_half != 1))
      VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: bad concat layout (xblk=%d, half=%d, W=%d)", a.out_xblk, a.out_half, a.W);
  } else if (a.out_pitch % 4 || a.out_coff % 4 || a.out_pitch < a.out_coff + a.C) {
    VX_FAIL(VX_E_ALIGN, "vx_norm_act_drop_pool: pitches/offsets must be multiples of 4 floats");
  }
  if (a.drop_mode == VX_DROP_MASK && !a.drop_mask) VX_FAIL(VX_E_NULL, "vx_norm_act_drop_pool: mask mode without mask");
  if ((int64_t)a.D * a.H * a.W * a.C >= (1ll << 32)) VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: sample too large");
  hipStream_t s = (hipStream_t)stream;
  if (a.pool_out) {
    if (a.D % 2 || a.H % 2 || a.W % 2) VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: pooling needs even dims");
    if (a.pool_pitch % 4 || a.pool_pitch < a.C) VX_FAIL(VX_E_ALIGN, "vx_norm_act_drop_pool: pool pitch");
    // the x-pair exchange is a shuffle over C/4 lanes: both voxels of a pair must sit in one wave, i.e. the
    // pieces of a row must not straddle a 64-lane boundary mid-pair: (W*C/4) % (2*C/4) == 0 always holds (W even)
    // and a wave starts at a multiple of 64 pieces, which is a multiple of 2*C/4 when C/4 divides 32
    if (a.C / 4 <= 32 && (32 % (a.C / 4)) != 0)
      VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: pooling needs C/4 to divide 32, or C > 128 (C = %d)", a.C);
  }
  {
    const int pool = a.pool_out ? 1 : 0;
    const int wide = pool && a.C / 4 > 32;
    const int C4 = a.C / 4, PW = (wide ? a.W / 2 : a.W) * C4, RH = pool ? a.H / 2 : a.H, RD = pool ? a.D / 2 : a.D;
    const int64_t per = (int64_t)RD * RH * PW;
    const int64_t dmax = PW > RH ? PW : RH;
    if (per * dmax >= (1ll << 32) || a.N > 65535)
      VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: sample too large for the 32-bit index decode (%lld pieces) or N > 65535", (long long)per);
    NormDecode dc;
    dc.per_sample = (unsigned)per;
    auto magic = [](int d) { return d == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d) + 1u; };
    dc.mPW = magic(PW); dc.mC4 = magic(C4); dc.mH = magic(RH);
    int bx = (int)((per + 255) / 256);
    const int cap = (16384 + a.N - 1) / a.N;
    if (!pool && bx > cap) bx = cap > 0 ? cap : 1;
    if (!pool && x_repeat > 1) {
      if (a.N % x_repeat) VX_FAIL(VX_E_SHAPE, "vx_norm_act_drop_pool: N=%d is not a multiple of x_repeat=%d", a.N, x_repeat);
      const int ns = a.N / x_repeat;
      int fx = (int)((per + 255) / 256);
      const int fcap = (32768 + ns - 1) / ns;
      if (fx > fcap) fx = fcap;
      vx_note_kernel("norm_act_drop_fanout_kernel");
      hipLaunchKernelGGL(norm_act_drop_fanout_kernel, dim3(fx, ns), dim3(256), 0, s, a, x_repeat, dc);
    } else if (stsrc) {
      // (four pieces per thread: the workgroup's reduction of the partials is paid once per 1 024 pieces instead of 256)
      bx = (bx + 3) / 4;
      if (wide) {
        vx_note_kernel("norm_act_drop_pool_kernel<true,true,true>");
        hipLaunchKernelGGL((norm_act_drop_pool_kernel<true, true, true>), dim3(bx, a.N), dim3(256), 0, s, a, x_repeat, dc, st);
      } else if (pool) {
        vx_note_kernel("norm_act_drop_pool_kernel<true,false,true>");
        hipLaunchKernelGGL((norm_act_drop_pool_kernel<true, false, true>), dim3(bx, a.N), dim3(256), 0, s, a, x_repeat, dc, st);
      } else {
        vx_note_kernel("norm_act_drop_pool_kernel<false,false,true>");
        hipLaunchKernelGGL((norm_act_drop_pool_kernel<false, false, true>), dim3(bx, a.N), dim3(256), 0, s, a, x_repeat, dc, st);
      }
    } else if (wide) {
      vx_note_kernel("norm_act_drop_pool_kernel<true,true>");
      hipLaunchKernelGGL((norm_act_drop_pool_kernel<true, true>), dim3(bx, a.N), dim3(256), 0, s, a, x_repeat, dc, st);
    } else if (pool) {
      vx_note_kernel("norm_act_drop_pool_kernel<true,false>");
      hipLaunchKernelGGL(norm_act_drop_pool_kernel<true>, dim3(bx, a.N), dim3(256), 0, s, a, x_repeat, dc, st);
    } else {
      vx_note_kernel("norm_act_drop_pool_kernel<false,false>");
      hipLaunchKernelGGL(norm_act_drop_pool_kernel<false>, dim3(bx, a.N), dim3(256), 0, s, a, x_repeat, dc, st);
    }
  }
  VX_CHECK_LAUNCH("vx_norm_act_drop_pool");
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The keep-bits VX_DROP_HASH uses, written out as a VX_DROP_MASK-style mask: mask[n][e] = bit of element e of sample n
// of dropout layer `layer` under `seed` (e = channels-last linear index, voxel * C + c -- what every kernel hashes).
// Lets a caller replay a hash-dropout run with explicit masks (parity tests feed them to the float64 oracle).
__global__ __launch_bounds__(256) void drop_hash_mask_kernel(uint32_t seed, uint32_t layer, int N, int64_t per,
                                                             uint8_t* __restrict__ out) {
  const int64_t groups = per / 4;
  const int64_t total = (int64_t)N * groups;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / groups, g = i - n * groups;
    const uint32_t bits = vx_drop_bits4(vx_drop_key(seed, layer, (uint32_t)n), (uint32_t)(4 * g));
    const uint32_t v = (bits & 1u) | ((bits >> 1 & 1u) << 8) | ((bits >> 2 & 1u) << 16) | ((bits >> 3 & 1u) << 24);
    *reinterpret_cast<uint32_t*>(out + n * per + 4 * g) = v;
  }
}

extern "C" int vx_drop_hash_mask(uint32_t seed, uint32_t layer, int N, int64_t elems_per_sample, uint8_t* mask,
                                 vx_stream_t stream) {
  if (N <= 0 || elems_per_sample <= 0 || elems_per_sample % 4 || elems_per_sample >= (1ll << 32))
    VX_FAIL(VX_E_SHAPE, "vx_drop_hash_mask: N=%d, %lld elements per sample (a positive multiple of 4 below 2^32)", N,
            (long long)elems_per_sample);
  if (!mask || (((uintptr_t)mask) & 3u)) VX_FAIL(VX_E_ALIGN, "vx_drop_hash_mask: mask must be a 4-byte aligned device pointer");
  const int64_t total = (int64_t)N * (elems_per_sample / 4);
  int bx = (int)((total + 255) / 256);
  if (bx > 16384) bx = 16384;
  hipLaunchKernelGGL(drop_hash_mask_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, seed, layer, N, elems_per_sample, mask);
  VX_CHECK_LAUNCH("vx_drop_hash_mask");
  return VX_OK;
}

// The second half of a pooled contract block whose conv left window maxima and any-dropped bits (vx_conv3d_args.pool_out):
// one thread = one 16-byte piece (4 channels of one pooled voxel, C = 8).
__global__ __launch_bounds__(256) void pool_finish_kernel(const float* __restrict__ raw, const uint32_t* __restrict__ flags,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          float* __restrict__ out, int out_pitch, unsigned pieces, float s) {
  const int n = blockIdx.y;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < pieces; i += gridDim.x * 256u) {
    const unsigned vox = i >> 1, q = i & 1u;
    const size_t pv = (size_t)n * (pieces >> 1) + vox;
    const f32x4 m = *reinterpret_cast<const f32x4*>(raw + pv * 8 + q * 4);
    const uint32_t fl = flags[pv * 2 + q];
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (size_t)n * 8 + q * 4);
    const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + (size_t)n * 8 + q * 4);
    f32x4 t = (m - mu) * rs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = fmaxf(t[j], 0.01f * t[j]) * s;       // as vx_norm_act_drop_pool: normalise, LeakyReLU, then the dropout's 2
      if ((fl >> j) & 1u) v = fmaxf(v, 0.f);          // a dropped element contributes 0 to the window
      t[j] = v;
    }
    *reinterpret_cast<f32x4*>(out + pv * out_pitch + q * 4) = t;
  }
}

// vx_pool_finish_z: the 16-channel z-column kernel leaves the (y, x) half of every window per z-plane (values_amd.h); one
// thread = one 16-byte piece of a pooled voxel: maximum / OR over the z pair, then pool_finish_kernel's arithmetic
template <bool FOLD>
__global__ __launch_bounds__(256) void pool_finish_z_kernel(const float* __restrict__ raw, const uint32_t* __restrict__ flags,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ out, int out_pitch, unsigned Dp, unsigned pv_plane, float s,
                                                            StatSrc st) {
  const int n = blockIdx.y;
  alignas(16) __shared__ float s_mu[FOLD ? 16 : 4], s_rs[FOLD ? 16 : 4];      // (read as 16-byte vectors)
  __shared__ double s_part[FOLD ? 512 : 1];
  if constexpr (FOLD) vx_block_instnorm(st, n, 16, blockIdx.x == 0, s_mu, s_rs, s_part);
  // this sample's 16 means / reciprocal deviations: the workgroup's own reduction (LDS) or row n of the caller's tables
  const float* mu_n = FOLD ? s_mu : mean + (size_t)n * 16;
  const float* rs_n = FOLD ? s_rs : rstd + (size_t)n * 16;
  const unsigned pieces = Dp * pv_plane * 4u;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < pieces; i += gridDim.x * 256u) {
    const unsigned vox = i >> 2, q = i & 3u;
    const unsigned zp = vox / pv_plane, r = vox - zp * pv_plane;
    const size_t v0 = ((size_t)n * 2 * Dp + 2 * zp) * pv_plane + r, v1 = v0 + pv_plane;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(raw + v0 * 16 + q * 4);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(raw + v1 * 16 + q * 4);
    const uint32_t fl = flags[v0 * 4 + q] | flags[v1 * 4 + q];
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mu_n + q * 4);
    const f32x4 rs = *reinterpret_cast<const f32x4*>(rs_n + q * 4);
    f32x4 t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float tt = (fmaxf(a0[j], a1[j]) - mu[j]) * rs[j];
      float v = fmaxf(tt, 0.01f * tt) * s;
      if ((fl >> j) & 1u) v = fmaxf(v, 0.f);
      t[j] = v;
    }
    *reinterpret_cast<f32x4*>(out + ((size_t)n * Dp * pv_plane + vox) * out_pitch + q * 4) = t;
  }
}

// vx_prenorm_split: one thread = one 16-byte piece, in place (values_amd.h)
template <bool FOLD>
__global__ __launch_bounds__(256) void prenorm_split_kernel(float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, unsigned pieces, float scale, StatSrc st) {
  const int n = blockIdx.y;
  alignas(16) __shared__ float s_mu[FOLD ? 8 : 4], s_rs[FOLD ? 8 : 4];
  __shared__ double s_part[FOLD ? 512 : 1];
  if constexpr (FOLD) vx_block_instnorm(st, n, 8, blockIdx.x == 0, s_mu, s_rs, s_part);
  const float* mu_n = FOLD ? s_mu : mean + (size_t)n * 8;
  const float* rs_n = FOLD ? s_rs : rstd + (size_t)n * 8;
  f32x4* xs = reinterpret_cast<f32x4*>(x) + (size_t)n * pieces;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < pieces; i += gridDim.x * 256u) {
    const int c = (i & 1u) * 4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mu_n + c);
    f32x4 sc = *reinterpret_cast<const f32x4*>(rs_n + c);
    f32x4 v = xs[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = (v[j] - mu[j]) * (sc[j] * scale);     // the z-column kernel's prologue arithmetic, bit for bit
      v[j] = fmaxf(t, 0.01f * t);
    }
    f16x4 hi, lo;
    vx_split4(v, hi, lo);
    f32x4 o;
    o[0] = __builtin_bit_cast(float, (f16x2){hi[0], hi[1]});
    o[1] = __builtin_bit_cast(float, (f16x2){hi[2], hi[3]});
    o[2] = __builtin_bit_cast(float, (f16x2){lo[0], lo[1]});
    o[3] = __builtin_bit_cast(float, (f16x2){lo[2], lo[3]});
    xs[i] = o;
  }
}

extern "C" int vx_prenorm_split(float* x, const float* mean, const float* rstd, int N, int64_t nvox, float scale,
                                vx_stream_t stream) {
  if (!x || !mean || !rstd) VX_FAIL(VX_E_NULL, "vx_prenorm_split: null pointer");
  if (N <= 0 || N >= 65536 || nvox <= 0 || nvox >= (1ll << 30)) VX_FAIL(VX_E_SHAPE, "vx_prenorm_split: empty / too large");
  if (!vx_aligned16(x) || !vx_aligned16(mean) || !vx_aligned16(rstd)) VX_FAIL(VX_E_ALIGN, "vx_prenorm_split: alignment");
  const unsigned pieces = (unsigned)(nvox * 2);
  unsigned bx = (pieces + 255u) / 256u;
  if (bx > 512u) bx = 512u;
  vx_note_kernel("prenorm_split_kernel<false>");
  hipLaunchKernelGGL(prenorm_split_kernel<false>, dim3(bx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, pieces, scale, StatSrc{});
  VX_CHECK_LAUNCH("vx_prenorm_split");
  return VX_OK;
}

extern "C" int vx_prenorm_split_stats(float* x, const vx_stat_src* stsrc, int N, int64_t nvox, float scale, vx_stream_t stream) {
  if (!x) VX_FAIL(VX_E_NULL, "vx_prenorm_split_stats: null pointer");
  if (N <= 0 || N >= 65536 || nvox <= 0 || nvox >= (1ll << 30)) VX_FAIL(VX_E_SHAPE, "vx_prenorm_split_stats: empty / too large");
  if (!vx_aligned16(x)) VX_FAIL(VX_E_ALIGN, "vx_prenorm_split_stats: alignment");
  StatSrc st;
  const int rc = vx_stat_src_check(stsrc, 8, "vx_prenorm_split_stats", &st);
  if (rc != VX_OK) return rc;
  const unsigned pieces = (unsigned)(nvox * 2);
  unsigned bx = (pieces + 255u) / 256u;
  if (bx > 128u) bx = 128u;       // (longer-lived workgroups than the plain pass: each reduces the partials once)
  vx_note_kernel("prenorm_split_kernel<true>");
  hipLaunchKernelGGL(prenorm_split_kernel<true>, dim3(bx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, x, nullptr, nullptr, pieces, scale, st);
  VX_CHECK_LAUNCH("vx_prenorm_split_stats");
  return VX_OK;
}

extern "C" int vx_pool_finish(const float* pool_raw, const uint32_t* pool_flags, const float* mean, const float* rstd,
                              float* out, int out_pitch, int N, int64_t voxels_per_sample, int drop_scale2,
                              vx_stream_t stream) {
  if (!pool_raw || !pool_flags || !mean || !rstd || !out) VX_FAIL(VX_E_NULL, "vx_pool_finish: null pointer");
  if (N <= 0 || voxels_per_sample <= 0 || voxels_per_sample >= (1ll << 30)) VX_FAIL(VX_E_SHAPE, "vx_pool_finish: empty / too large");
  if (out_pitch < 8 || out_pitch % 4 || !vx_aligned16(pool_raw) || !vx_aligned16(out))
    VX_FAIL(VX_E_ALIGN, "vx_pool_finish: pitch %d / alignment", out_pitch);
  const unsigned pieces = (unsigned)(voxels_per_sample * 2);
  unsigned bx = (pieces + 255u) / 256u;
  if (bx > 64u) bx = 64u;
  if (N >= 65536) VX_FAIL(VX_E_SHAPE, "vx_pool_finish: N");
  vx_note_kernel("pool_finish_kernel");
  hipLaunchKernelGGL(pool_finish_kernel, dim3(bx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, pool_raw, pool_flags, mean,
                     rstd, out, out_pitch, pieces, drop_scale2 ? 2.f : 1.f);
  VX_CHECK_LAUNCH("vx_pool_finish");
  return VX_OK;
}

extern "C" int vx_pool_finish_z(const float* pool_raw, const uint32_t* pool_flags, const float* mean, const float* rstd,
                                float* out, int out_pitch, int N, int Dp, int64_t plane_voxels, int drop_scale2,
                                vx_stream_t stream) {
  if (!pool_raw || !pool_flags || !mean || !rstd || !out) VX_FAIL(VX_E_NULL, "vx_pool_finish_z: null pointer");
  if (N <= 0 || Dp <= 0 || plane_voxels <= 0 || (int64_t)Dp * plane_voxels >= (1ll << 28)) VX_FAIL(VX_E_SHAPE, "vx_pool_finish_z: empty / too large");
  if (out_pitch < 16 || out_pitch % 4 || !vx_aligned16(pool_raw) || !vx_aligned16(out))
    VX_FAIL(VX_E_ALIGN, "vx_pool_finish_z: pitch %d / alignment", out_pitch);
  if (N >= 65536) VX_FAIL(VX_E_SHAPE, "vx_pool_finish_z: N");
  const unsigned pieces = (unsigned)(Dp * plane_voxels * 4);
  unsigned bx = (pieces + 255u) / 256u;
  if (bx > 64u) bx = 64u;
  vx_note_kernel("pool_finish_z_kernel<false>");
  hipLaunchKernelGGL(pool_finish_z_kernel<false>, dim3(bx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, pool_raw, pool_flags, mean,
                     rstd, out, out_pitch, (unsigned)Dp, (unsigned)plane_voxels, drop_scale2 ? 2.f : 1.f, StatSrc{});
  VX_CHECK_LAUNCH("vx_pool_finish_z");
  return VX_OK;
}

extern "C" int vx_pool_finish_z_stats(const float* pool_raw, const uint32_t* pool_flags, const vx_stat_src* stsrc, float* out,
                                      int out_pitch, int N, int Dp, int64_t plane_voxels, int drop_scale2, vx_stream_t stream) {
  if (!pool_raw || !pool_flags || !out) VX_FAIL(VX_E_NULL, "vx_pool_finish_z_stats: null pointer");
  if (N <= 0 || Dp <= 0 || plane_voxels <= 0 || (int64_t)Dp * plane_voxels >= (1ll << 28)) VX_FAIL(VX_E_SHAPE, "vx_pool_finish_z_stats: empty / too large");
  if (out_pitch < 16 || out_pitch % 4 || !vx_aligned16(pool_raw) || !vx_aligned16(out))
    VX_FAIL(VX_E_ALIGN, "vx_pool_finish_z_stats: pitch %d / alignment", out_pitch);
  if (N >= 65536) VX_FAIL(VX_E_SHAPE, "vx_pool_finish_z_stats: N");
  StatSrc st;
  const int rc = vx_stat_src_check(stsrc, 16, "vx_pool_finish_z_stats", &st);
  if (rc != VX_OK) return rc;
  const unsigned pieces = (unsigned)(Dp * plane_voxels * 4);
  unsigned bx = (pieces + 255u) / 256u;
  if (bx > 16u) bx = 16u;
  vx_note_kernel("pool_finish_z_kernel<true>");
  hipLaunchKernelGGL(pool_finish_z_kernel<true>, dim3(bx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, pool_raw, pool_flags,
                     (const float*)nullptr, (const float*)nullptr, out, out_pitch, (unsigned)Dp, (unsigned)plane_voxels, drop_scale2 ? 2.f : 1.f, st);
  VX_CHECK_LAUNCH("vx_pool_finish_z_stats");
  return VX_OK;
}


extern "C" int vx_zero(void* p, int64_t bytes, vx_stream_t stream) {
  if (bytes < 0 || (!p && bytes > 0)) VX_FAIL(VX_E_NULL, "vx_zero: null pointer / negative size");
  if (bytes == 0) return VX_OK;
  const hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
  if (e != hipSuccess) VX_FAIL((int)e, "vx_zero: hipMemsetAsync(%lld B): %s", (long long)bytes, hipGetErrorString(e));
  return VX_OK;
}
