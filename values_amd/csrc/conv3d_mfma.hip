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
#include <stdlib.h>

struct ConvKArgs {
  vx_conv3d_args a;
  int tiles_x, tiles_y, tiles_z, nchunks, prio_mode;
  unsigned mx, my, mz;  // floor(2^32 / tiles_*) + 1: exact t / tiles_* = umulhi(t, m) for t * tiles_* < 2^32
  unsigned long long* dbg;  // VX_CONV_STAMPS diagnostic builds only
};

#ifdef VX_CONV_STAMPS
#define VX_STAMP(i)                                                                      \
  do {                                                                                   \
    unsigned long long t_;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    st_sum[i] += t_ - st_last;                                                           \
    st_last = t_;                                                                        \
  } while (0)
#else
#define VX_STAMP(i) do {} while (0)
#endif


template <int CB, int NT, int TX, int TY, int TZ, int NW>
__global__ __launch_bounds__(64 * NW) void conv3d_k3_mfma_kernel(ConvKArgs ka) {
  constexpr int NTH = 64 * NW;                     // threads per workgroup
  constexpr int CPL = CB / 4;                      // channels per lane per tap
  constexpr int NVT = TX * TY * TZ / 16;           // voxel tiles per workgroup
  constexpr int R = NVT / NW;                      // voxel tiles per wave
  constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2;
  constexpr int NHALO = HX * HY * HZ;
  constexpr int IN_FLOATS = NHALO * CB;
  constexpr int W_FLOATS = 27 * NT * 64 * CPL;
  constexpr int Q = CB / 4;                        // 16-byte pieces per voxel
  constexpr int IN_IT = (NHALO * Q + NTH - 1) / NTH;   // staging iterations per thread (input tile)
  constexpr int W_IT = (W_FLOATS / 4 + NTH - 1) / NTH; // staging iterations per thread (weights)
  static_assert(NVT % NW == 0, "tile must give each wave a whole number of voxel tiles");
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
  const int cg = blockIdx.y;  // cout group of 16*NT
  const int ntiles = ka.tiles_x * ka.tiles_y * ka.tiles_z;
  const int total = ntiles * a.N;

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
  // per-thread staging pattern, identical for every tile: halo coordinates (packed) of each piece
  int hc[IN_IT];
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int idx = tid + it * NTH;
    const int vox = idx / Q;
    const int hx = vox % HX, hy = (vox / HX) % HY, hz = vox / (HX * HY);
    hc[it] = (idx < NHALO * Q) ? (hx | (hy << 8) | (hz << 16) | ((idx % Q) << 24)) : -1;
  }

  const float* w_cg = a.w_packed + (size_t)cg * ka.nchunks * W_FLOATS;
  f32x4 ibuf[IN_IT];
  f32x4 wbuf[W_IT];

  // issue the global loads of one (tile, chunk) work item into registers (no wait)
  const bool w_resident = ka.nchunks == 1;  // single-chunk layers: the weights never change, stage them once
  bool w_fresh = true;
  auto decode = [&](int tile_lin, int& n, int& tx, int& ty, int& tz) {
    unsigned t = (unsigned)tile_lin, q;
    q = ka.tiles_x == 1 ? t : __umulhi(t, ka.mx); tx = (int)(t - q * ka.tiles_x); t = q;
    q = ka.tiles_y == 1 ? t : __umulhi(t, ka.my); ty = (int)(t - q * ka.tiles_y); t = q;
    q = ka.tiles_z == 1 ? t : __umulhi(t, ka.mz); tz = (int)(t - q * ka.tiles_z); n = (int)q;
  };
  // per-piece offset (in floats) relative to the tile's halo origin: loop-invariant
  int goff[IN_IT];
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int c = hc[it];
    goff[it] = (((c >> 16) & 0xff) * a.H + ((c >> 8) & 0xff)) * a.W * a.in_pitch + (c & 0xff) * a.in_pitch + ((c >> 24) & 0xff) * 4;
  }
  auto prefetch = [&](int tile_lin, int chunk, bool have, bool with_w) {
    int n, tx, ty, tz;
    decode(tile_lin, n, tx, ty, tz);
    const int x0 = tx * TX - 1, y0 = ty * TY - 1, z0 = tz * TZ - 1;
    // pointer to the (possibly out-of-volume) halo origin of this tile; only in-bounds pieces are dereferenced
    const float* org = a.in + ((size_t)n * a.D * a.H * a.W) * a.in_pitch + chunk * CB +
                       ((long long)(z0 * a.H + y0) * a.W + x0) * a.in_pitch;
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const int c = hc[it];
      const int gx = x0 + (c & 0xff), gy = y0 + ((c >> 8) & 0xff), gz = z0 + ((c >> 16) & 0xff);
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (have && c >= 0 && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H && (unsigned)gz < (unsigned)a.D)
        v = *reinterpret_cast<const f32x4*>(org + goff[it]);
      ibuf[it] = v;
    }
    const f32x4* src = reinterpret_cast<const f32x4*>(w_cg + (size_t)chunk * W_FLOATS);
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int idx = tid + it * NTH;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (have && with_w && idx < W_FLOATS / 4) v = src[idx];
      wbuf[it] = v;
    }
  };
  auto commit = [&](bool with_w) {
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const int idx = tid + it * NTH;
      if (idx < NHALO * Q) *reinterpret_cast<f32x4*>(s_in + idx * 4) = ibuf[it];
    }
    if (with_w) {
#pragma unroll
      for (int it = 0; it < W_IT; ++it) {
        const int idx = tid + it * NTH;
        if (idx < W_FLOATS / 4) reinterpret_cast<f32x4*>(s_w)[idx] = wbuf[it];
      }
    }
  };

  // One flat loop over (tile, chunk) work items: commit the prefetched registers to LDS, issue the NEXT item's
  // global loads (straight-line, predicated -- no loop-carried register copies that would force a vmcnt(0)),
  // compute this item, and run the epilogue after a tile's last chunk.
  if (ka.prio_mode) {
    // co-resident workgroups (dispatch order: b and b+256 share a CU at 2 WGs/CU) get different static
    // priorities so they de-phase: one computes while the other stages/stores
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    if ((lin >> 8) & 1) __builtin_amdgcn_s_setprio(1);
  }
  // bias once per workgroup: a load inside the loop would sit behind the prefetch loads in the in-order vmcnt
  // queue and make every epilogue wait for the NEXT tile's data
  f32x4 bias4[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = (cg * NT + nt) * 16 + g * 4;
    bias4[nt] = co < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + co) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  int tile_lin = blockIdx.x, chunk = 0;
  bool have = tile_lin < total;
  prefetch(tile_lin, 0, have, true);
  f32x4 acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifdef VX_CONV_STAMPS
  unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_last, st_iters = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
  while (have) {
    __syncthreads();   // everyone finished reading the previous item from LDS
    VX_STAMP(0);
    commit(w_fresh);   // (waits for the prefetched loads)
    VX_STAMP(1);
    __syncthreads();
    VX_STAMP(2);
    w_fresh = !w_resident;
    int ntile = tile_lin, nchunk = chunk + 1;
    if (nchunk == ka.nchunks) { nchunk = 0; ntile = tile_lin + (int)gridDim.x; }
    const bool nhave = ntile < total;
    prefetch(ntile, nchunk, nhave, !w_resident);
    VX_STAMP(3);

    {
      // ---- 27 taps x CPL MFMAs x R x NT, fragments double-buffered one tap ahead ----
      vecc wf[2][NT], xf[2][R];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wf[0][nt] = *reinterpret_cast<const vecc*>(s_w + (nt * 64 + lane) * CPL);
#pragma unroll
      for (int r = 0; r < R; ++r) xf[0][r] = *reinterpret_cast<const vecc*>(s_in + vbase[r]);
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        const int cur = tap & 1, nxt = cur ^ 1;
        if (tap + 1 < 27) {
          const int t1 = tap + 1;
          const int kz = t1 / 9, ky = (t1 / 3) % 3, kx = t1 % 3;
          const int toff = ((kz * HY + ky) * HX + kx) * CB;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            wf[nxt][nt] = *reinterpret_cast<const vecc*>(s_w + ((t1 * NT + nt) * 64 + lane) * CPL);
#pragma unroll
          for (int r = 0; r < R; ++r) xf[nxt][r] = *reinterpret_cast<const vecc*>(s_in + vbase[r] + toff);
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j)
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cur][nt][j], xf[cur][r][j], acc[r][nt], 0, 0, 0);
      }
    }

    VX_STAMP(4);
    if (chunk == ka.nchunks - 1) {
    // ---- epilogue ----
    int n, tx, ty, tz;
    decode(tile_lin, n, tx, ty, tz);
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;

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
      const f32x4 b4 = bias4[nt];
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
      float* s_red = smem + IN_FLOATS + W_FLOATS;  // [NW waves][NT][16 couts][2], outside the staged tiles
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
          for (int w = 0; w < NW; ++w) {
            s += s_red[((w * NT + nt) * 16 + c) * 2 + 0];
            q += s_red[((w * NT + nt) * 16 + c) * 2 + 1];
          }
          const int tile = tile_lin - n * ntiles;
          float* dst = a.stats_partial + (((size_t)n * ntiles + tile) * a.Cout + co) * 2;
          dst[0] = s;
          dst[1] = q;
        }
      }
      // s_red is re-written only after the next item's two barriers, so no extra barrier is needed here
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }  // last chunk of the tile
    VX_STAMP(5);
#ifdef VX_CONV_STAMPS
    ++st_iters;
#endif
    tile_lin = ntile; chunk = nchunk; have = nhave;
  }
#ifdef VX_CONV_STAMPS
  if (ka.dbg && blockIdx.y == 0 && (tid & 63) == 0) {
    unsigned long long* d = ka.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
    for (int i = 0; i < 6; ++i) d[i] = st_sum[i];
    d[6] = st_iters;
  }
#endif
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

template <int CB, int NT, int TX, int TY, int TZ, int NW>
static int launch_conv(const ConvKArgs& ka, hipStream_t s) {
  constexpr int IN_FLOATS = (TX + 2) * (TY + 2) * (TZ + 2) * CB;
  constexpr int W_FLOATS = 27 * NT * 64 * (CB / 4);
  constexpr size_t lds = (size_t)(IN_FLOATS + W_FLOATS + NW * NT * 16 * 2) * sizeof(float);
  static bool attr_set = false;
  auto kern = conv3d_k3_mfma_kernel<CB, NT, TX, TY, TZ, NW>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv3d_k3: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr_set = true;
  }
  const vx_conv3d_args& a = ka.a;
  const int NTc = NT;
  // persistent workgroups: each loops over tiles with stride gridDim.x (prefetching the next tile's loads
  // while it computes); enough workgroups to fill every CU at the LDS-limited occupancy
  const int total_tiles = ka.tiles_x * ka.tiles_y * ka.tiles_z * a.N;
  const int ygroups = (a.Cout + 16 * NTc - 1) / (16 * NTc);
  int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
  if (const char* e = getenv("VX_CONV_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;  // tuning knob
  int gx = (256 * per_cu + ygroups - 1) / ygroups;
  if (gx > total_tiles) gx = total_tiles;
  dim3 grid((unsigned)gx, (unsigned)ygroups);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv3d_k3");
  return VX_OK;
}

template <int CB, int NT>
static int dispatch_tile(const ConvKArgs& ka, const TileCfg& t, hipStream_t s) {
  static const int nw8 = getenv("VX_CONV_NW4") ? 0 : 1;  // tuning knob: 8 waves (default) vs 4 waves per workgroup
  if (t.TX == 16) return nw8 ? launch_conv<CB, NT, 16, 4, 4, 8>(ka, s) : launch_conv<CB, NT, 16, 4, 4, 4>(ka, s);
  if (t.TX == 8) return nw8 ? launch_conv<CB, NT, 8, 8, 4, 8>(ka, s) : launch_conv<CB, NT, 8, 8, 4, 4>(ka, s);
  return launch_conv<CB, NT, 4, 4, 4, 4>(ka, s);
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
  ka.mx = (unsigned)((1ull << 32) / (unsigned)ka.tiles_x) + 1u;
  ka.my = (unsigned)((1ull << 32) / (unsigned)ka.tiles_y) + 1u;
  ka.mz = (unsigned)((1ull << 32) / (unsigned)ka.tiles_z) + 1u;
  ka.prio_mode = getenv("VX_CONV_PRIO") ? 1 : 0;
  ka.dbg = nullptr;
#ifdef VX_CONV_STAMPS
  if (const char* e = getenv("VX_CONV_DBG_PTR")) ka.dbg = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  hipStream_t s = (hipStream_t)stream;
  if (c.CB == 16 && c.NT == 1) return dispatch_tile<16, 1>(ka, t, s);
  if (c.CB == 16 && c.NT == 2) return dispatch_tile<16, 2>(ka, t, s);
  if (c.CB == 8 && c.NT == 1) return dispatch_tile<8, 1>(ka, t, s);
  return dispatch_tile<8, 2>(ka, t, s);
}
