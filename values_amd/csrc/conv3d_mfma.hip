// K1: 3x3x3 convolution (padding 1), channels-last fp32, as an implicit GEMM on the gfx950
// fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain, same numerics as VALU).
//
//   GEMM view:  D[cout][voxel] = sum_k  Wt[cout][k] * X[k][voxel],  k = (tap, cin)
//   A operand = weights  (lane l holds A[row = l&15][k = l>>4])
//   B operand = inputs   (lane l holds B[k = l>>4][col = l&15])
//   D: lane l holds rows (l>>4)*4 + {0..3} of column l&15  -> 4 consecutive couts of one voxel,
//      so the epilogue is ONE 16-byte store per lane and a wave writes 16 voxels x 64 B contiguously.
//
// Work split: a 256-thread workgroup (4 waves) owns an output tile of TX*TY*TZ voxels (a multiple
// of 64) for 16*NT output channels.  Each wave owns R = TX*TY*TZ/64 "voxel tiles" of 16 voxels and
// all NT channel tiles: R*NT independent accumulators (hides the 40-cycle dependent MFMA latency).
// Cin is consumed in chunks of CB (8 or 16) channels: per chunk the halo'd input tile
// [(TZ+2)(TY+2)(TX+2)][CB] and the chunk's weights [27][NT][64 lanes][CB/4] are staged in LDS;
// a lane's CB/4 consecutive channels come from one ds_read_b128 (b64 for CB=8) and feed CB/4
// MFMAs, so per tap a wave issues NT + R wide LDS reads for 4*R*NT (CB=16) MFMAs.
//
// Epilogue (fused): bias, then either (a) raw store + per-workgroup (sum, sumsq) partials for the
// InstanceNorm that follows (contract blocks), or (b) LeakyReLU/ReLU + dropout (expand / center).
#include "common.h"

struct ConvKArgs {
  vx_conv3d_args a;
  int tiles_x, tiles_y, tiles_z, nchunks;
};

template <int CB, int NT, int TX, int TY, int TZ>
__global__ __launch_bounds__(256) void conv3d_k3_mfma_kernel(ConvKArgs ka) {
  constexpr int CPL = CB / 4;                      // channels per lane per tap
  constexpr int NVT = TX * TY * TZ / 16;           // voxel tiles per workgroup
  constexpr int R = NVT / 4;                       // voxel tiles per wave
  constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2;
  constexpr int NHALO = HX * HY * HZ;
  constexpr int IN_FLOATS = NHALO * CB;
  constexpr int W_FLOATS = 27 * NT * 64 * CPL;
  static_assert(NVT % 4 == 0, "tile must give each wave a whole number of voxel tiles");
  typedef float vecc __attribute__((ext_vector_type(CPL)));

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + IN_FLOATS;

  const vx_conv3d_args& a = ka.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int m = lane & 15;   // voxel within voxel tile (B column) / cout within tile (A row)
  const int g = lane >> 4;   // k index within the MFMA's K=4

  int t = blockIdx.x;
  const int tx = t % ka.tiles_x; t /= ka.tiles_x;
  const int ty = t % ka.tiles_y; t /= ka.tiles_y;
  const int tz = t % ka.tiles_z; t /= ka.tiles_z;
  const int n = t;
  const int cg = blockIdx.y;  // cout group of 16*NT
  const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;

  // per-lane LDS voxel base (halo coordinates, tap (0,0,0)) of each of the wave's voxel tiles
  int vbase[R];
  int vx_[R], vy_[R], vz_[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int v = (wave * R + r) * 16 + m;
    const int lx = v % TX, ly = (v / TX) % TY, lz = v / (TX * TY);
    vx_[r] = lx; vy_[r] = ly; vz_[r] = lz;
    vbase[r] = ((lz * HY + ly) * HX + lx) * CB + g * CPL;
  }

  f32x4 acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float* in_n = a.in + (size_t)n * a.D * a.H * a.W * a.in_pitch;
  const float* w_cg = a.w_packed + (size_t)cg * ka.nchunks * W_FLOATS;

  for (int chunk = 0; chunk < ka.nchunks; ++chunk) {
    if (chunk > 0) __syncthreads();
    // ---- stage the halo'd input tile (zero padded) ----
    {
      constexpr int Q = CB / 4;  // 16-byte pieces per voxel
      const int c0 = chunk * CB;
      for (int idx = tid; idx < NHALO * Q; idx += 256) {
        const int vox = idx / Q, q = idx % Q;
        const int hx = vox % HX, hy = (vox / HX) % HY, hz = vox / (HX * HY);
        const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if ((unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H && (unsigned)gz < (unsigned)a.D)
          v = *reinterpret_cast<const f32x4*>(in_n + ((size_t)(gz * a.H + gy) * a.W + gx) * a.in_pitch + c0 + q * 4);
        *reinterpret_cast<f32x4*>(s_in + vox * CB + q * 4) = v;
      }
    }
    // ---- stage this chunk's weights (already in fragment order) ----
    {
      const f32x4* src = reinterpret_cast<const f32x4*>(w_cg + (size_t)chunk * W_FLOATS);
      f32x4* dst = reinterpret_cast<f32x4*>(s_w);
      for (int idx = tid; idx < W_FLOATS / 4; idx += 256) dst[idx] = src[idx];
    }
    __syncthreads();

    // ---- 27 taps x CPL MFMAs x R x NT ----
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int tap = (kz * 3 + ky) * 3 + kx;
          const int toff = ((kz * HY + ky) * HX + kx) * CB;
          vecc wf[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            wf[nt] = *reinterpret_cast<const vecc*>(s_w + ((tap * NT + nt) * 64 + lane) * CPL);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const vecc xf = *reinterpret_cast<const vecc*>(s_in + vbase[r] + toff);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int j = 0; j < CPL; ++j)
                acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][j], xf[j], acc[r][nt], 0, 0, 0);
          }
        }
  }

  // ---- epilogue ----
  float ssum[NT][4], ssq[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[nt][j] = 0.f; ssq[nt][j] = 0.f; }

  const uint32_t dkey = vx_drop_key(a.drop_seed, a.drop_layer, (uint32_t)n);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = (cg * NT + nt) * 16 + g * 4;  // this lane's 4 couts
    const bool cvalid = co < a.Cout;
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (cvalid) b4 = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int gx = x0 + vx_[r], gy = y0 + vy_[r], gz = z0 + vz_[r];
      const bool valid = cvalid && gx < a.W && gy < a.H && gz < a.D;
      if (!valid) continue;
      f32x4 v = acc[r][nt] + b4;
      const size_t vox = ((size_t)(n * a.D + gz) * a.H + gy) * a.W + gx;
      if (a.stats_partial) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[nt][j] += v[j]; ssq[nt][j] += v[j] * v[j]; }
      }
      if (a.act != VX_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = vx_act(v[j], a.act);
      }
      if (a.drop_mode == VX_DROP_HASH) {
        const uint32_t e = (uint32_t)(((gz * a.H + gy) * a.W + gx) * a.Cout + co);
        const uint32_t bits = vx_drop_bits4(dkey, e);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((bits >> j) & 1u) ? 2.f * v[j] : 0.f;
      } else if (a.drop_mode == VX_DROP_MASK) {
        const uint32_t mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + vox * a.Cout + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * v[j] : 0.f;
      }
      *reinterpret_cast<f32x4*>(a.out + vox * a.out_pitch + a.out_coff + co) = v;
    }
  }

  if (a.stats_partial) {
    // reduce over the 16 lanes that share (g) -> per-wave sums per cout, then over waves via LDS
    __syncthreads();  // everyone is done with s_w / s_in
    float* s_red = smem;  // [4 waves][NT][16 couts][2]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = ssum[nt][j], q = ssq[nt][j];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          s += __shfl_xor(s, off, 64);
          q += __shfl_xor(q, off, 64);
        }
        if (m == 0) {
          s_red[((wave * NT + nt) * 16 + g * 4 + j) * 2 + 0] = s;
          s_red[((wave * NT + nt) * 16 + g * 4 + j) * 2 + 1] = q;
        }
      }
    __syncthreads();
    if (tid < NT * 16) {
      const int nt = tid / 16, c = tid % 16;
      const int co = (cg * NT + nt) * 16 + c;
      if (co < a.Cout) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s += s_red[((w * NT + nt) * 16 + c) * 2 + 0];
          q += s_red[((w * NT + nt) * 16 + c) * 2 + 1];
        }
        const int ntiles = ka.tiles_x * ka.tiles_y * ka.tiles_z;
        const int tile = blockIdx.x - n * ntiles;
        float* dst = a.stats_partial + (((size_t)n * ntiles + tile) * a.Cout + co) * 2;
        dst[0] = s;
        dst[1] = q;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight packing: torch (Cout, Cin, 3,3,3) -> [cgrp16][chunk][tap][lane 64][CPL]
// (NT consecutive cgrp16 blocks of one chunk are NOT contiguous in this order, so the packed
//  order is [cgrpNT][chunk][tap][nt][lane][CPL] with NT fixed per (Cin, Cout) by conv_config().)
struct ConvCfg { int CB, NT; };
static inline ConvCfg conv_config(int Cin, int Cout) {
  ConvCfg c;
  c.CB = (Cin % 16 == 0) ? 16 : 8;
  c.NT = (Cout % 32 == 0) ? 2 : 1;
  return c;
}
static inline int conv_cout_padded(int Cout, int NT) { return ((Cout + 16 * NT - 1) / (16 * NT)) * (16 * NT); }

__global__ void pack_conv3d_k3_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int CB,
                                      int NT, int64_t total) {
  const int CPL = CB / 4;
  const int nchunks = Cin / CB;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int j = r % CPL; r /= CPL;
    const int lane = r % 64; r /= 64;
    const int nt = r % NT; r /= NT;
    const int tap = r % 27; r /= 27;
    const int chunk = r % nchunks; r /= nchunks;
    const int cgrp = (int)r;
    const int co = (cgrp * NT + nt) * 16 + (lane & 15);
    const int ci = chunk * CB + (lane >> 4) * CPL + j;
    float v = 0.f;
    if (co < Cout) v = w[((size_t)co * Cin + ci) * 27 + tap];
    out[i] = v;
  }
}

extern "C" int64_t vx_conv3d_k3_packed_floats(int Cin, int Cout) {
  if (Cin % 8 != 0 || Cout % 8 != 0 || Cin <= 0 || Cout <= 0) return -1;
  ConvCfg c = conv_config(Cin, Cout);
  return (int64_t)conv_cout_padded(Cout, c.NT) * Cin * 27;
}

extern "C" int vx_pack_conv3d_k3(const float* w_torch, float* w_packed, int Cin, int Cout, vx_stream_t stream) {
  if (!w_torch || !w_packed) VX_FAIL(VX_E_NULL, "vx_pack_conv3d_k3: null pointer");
  int64_t total = vx_conv3d_k3_packed_floats(Cin, Cout);
  if (total < 0) VX_FAIL(VX_E_SHAPE, "vx_pack_conv3d_k3: Cin=%d Cout=%d must be positive multiples of 8", Cin, Cout);
  ConvCfg c = conv_config(Cin, Cout);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_conv3d_k3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_torch, w_packed, Cin,
                     Cout, c.CB, c.NT, total);
  VX_CHECK_LAUNCH("vx_pack_conv3d_k3");
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------
struct TileCfg { int TX, TY, TZ; };
static inline TileCfg tile_config(int W) {
  if (W >= 16) return {16, 4, 4};
  if (W >= 8) return {8, 8, 4};
  return {4, 4, 4};
}

extern "C" int vx_conv3d_k3_tiles(int D, int H, int W) {
  TileCfg t = tile_config(W);
  return ((W + t.TX - 1) / t.TX) * ((H + t.TY - 1) / t.TY) * ((D + t.TZ - 1) / t.TZ);
}

template <int CB, int NT, int TX, int TY, int TZ>
static int launch_conv(const ConvKArgs& ka, hipStream_t s) {
  constexpr int IN_FLOATS = (TX + 2) * (TY + 2) * (TZ + 2) * CB;
  constexpr int W_FLOATS = 27 * NT * 64 * (CB / 4);
  constexpr size_t lds = (size_t)(IN_FLOATS + W_FLOATS) * sizeof(float);
  static bool attr_set = false;
  auto kern = conv3d_k3_mfma_kernel<CB, NT, TX, TY, TZ>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv3d_k3: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr_set = true;
  }
  const vx_conv3d_args& a = ka.a;
  const int NTc = NT;
  dim3 grid((unsigned)(ka.tiles_x * ka.tiles_y * ka.tiles_z * a.N), (unsigned)((a.Cout + 16 * NTc - 1) / (16 * NTc)));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv3d_k3");
  return VX_OK;
}

template <int CB, int NT>
static int dispatch_tile(const ConvKArgs& ka, const TileCfg& t, hipStream_t s) {
  if (t.TX == 16) return launch_conv<CB, NT, 16, 4, 4>(ka, s);
  if (t.TX == 8) return launch_conv<CB, NT, 8, 8, 4>(ka, s);
  return launch_conv<CB, NT, 4, 4, 4>(ka, s);
}

extern "C" int vx_conv3d_k3(const vx_conv3d_args* ap, vx_stream_t stream) {
  if (!ap) VX_FAIL(VX_E_NULL, "vx_conv3d_k3: null args");
  const vx_conv3d_args& a = *ap;
  if (!a.in || !a.w_packed || !a.bias || !a.out) VX_FAIL(VX_E_NULL, "vx_conv3d_k3: null tensor pointer");
  if (a.Cin <= 0 || a.Cout <= 0 || a.Cin % 8 || a.Cout % 8)
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: Cin=%d Cout=%d must be positive multiples of 8", a.Cin, a.Cout);
  if (a.N <= 0 || a.D <= 0 || a.H <= 0 || a.W <= 0) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: empty tensor");
  if (a.in_pitch < a.Cin || a.in_pitch % 4 || a.out_pitch < a.out_coff + a.Cout || a.out_pitch % 4 || a.out_coff % 4)
    VX_FAIL(VX_E_ALIGN, "vx_conv3d_k3: pitches/offset must be multiples of 4 floats and cover the channels");
  if (!vx_aligned16(a.in) || !vx_aligned16(a.out) || !vx_aligned16(a.w_packed) || !vx_aligned16(a.bias))
    VX_FAIL(VX_E_ALIGN, "vx_conv3d_k3: pointers must be 16-byte aligned");
  if (a.act < 0 || a.act > 2 || a.drop_mode < 0 || a.drop_mode > 2) VX_FAIL(VX_E_DTYPE, "vx_conv3d_k3: bad act/drop enum");
  if (a.drop_mode == VX_DROP_MASK && !a.drop_mask) VX_FAIL(VX_E_NULL, "vx_conv3d_k3: VX_DROP_MASK without mask");
  if ((int64_t)a.D * a.H * a.W * a.Cout >= (1ll << 32)) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: sample too large");

  ConvCfg c = conv_config(a.Cin, a.Cout);
  TileCfg t = tile_config(a.W);
  ConvKArgs ka;
  ka.a = a;
  ka.tiles_x = (a.W + t.TX - 1) / t.TX;
  ka.tiles_y = (a.H + t.TY - 1) / t.TY;
  ka.tiles_z = (a.D + t.TZ - 1) / t.TZ;
  ka.nchunks = a.Cin / c.CB;
  hipStream_t s = (hipStream_t)stream;
  if (c.CB == 16 && c.NT == 1) return dispatch_tile<16, 1>(ka, t, s);
  if (c.CB == 16 && c.NT == 2) return dispatch_tile<16, 2>(ka, t, s);
  if (c.CB == 8 && c.NT == 1) return dispatch_tile<8, 1>(ka, t, s);
  return dispatch_tile<8, 2>(ka, t, s);
}
