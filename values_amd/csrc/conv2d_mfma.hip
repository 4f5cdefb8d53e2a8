// K15: 2D convolutions of HRNet (uncertainty_modeling/models/hrnet_module.py:37-41, 85-93, 349-358, 411-428):
// 3x3 stride 1, 3x3 stride 2, 1x1 -- channels-last fp32, implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32),
// same skeleton as conv3d_mfma.hip: persistent 512-thread workgroups over a flat list of (tile, cin-chunk) items,
// next item's loads in flight in registers (buffer loads, out-of-image pieces steered out of range -> zeros),
// group-major conflict-free LDS image, weights pre-packed in fragment order.
//
//   D[cout][x] = sum_k W[cout][k] X[k][x],  k = (ky, kx, cin);  rows = 16*NT couts, columns = 16 consecutive output x
//   Workgroup tile = 16 (x) x TY (y) outputs; wave w owns R = TY/8 rows.
//   Stride 2: output x reads input 2x + kx - 1, so even/odd input columns (and rows) live in separate parity planes
//   of the LDS image and a tile's 16 columns stay consecutive positions (same trick as the 3D x-pair packing).
//   1x1: no halo, up to NSUB = 4 sub-blocks of 16 input channels per item (a GEMM with K = 64 per barrier pair).
//
// Epilogue: optional bias, raw store, and per-workgroup (sum, sum of squares) partials per channel for the
// TRAINING-mode BatchNorm that follows every conv (batch statistics over N,H,W) -- deterministic, no
// atomics; vx_bn_finalize reduces them.
#include "common.h"
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// conv2d_s16.hip: the split-fp16 schedule (default; VX_CONV_FP32=1 selects the native-fp32 kernels of this file)
int64_t vx_conv2d_s16_packed_floats(int Cin, int Cout, int KS);
int vx_pack_conv2d_s16(const float* w_torch, float* w_packed, int Cin, int Cout, int KS, hipStream_t s);
int vx_conv2d_s16(const vx_conv2d_args& a, hipStream_t s);
int vx_conv2d_s16_row_tiles(int KS, int Cout);
int vx_conv2d_s16_octets(int Cin, int KS);
static inline bool c2_split16() { return vx_cfg().conv_fp32 == 0; }

struct Conv2dKArgs {
  vx_conv2d_args a;
  int OH, OW, tiles_x, tiles_y, nchunks;   // nchunks = ceil(Cin/16 / NSUB)
  unsigned mx, my;
};

namespace {
constexpr unsigned K_OOB = 0xFFFFFFF0u;
constexpr unsigned K_NUMREC = 0x80000000u;
}

template <int KS, int S, int NT, int NSUB, int TY>
__global__ __launch_bounds__(512) void conv2d_mfma_kernel(Conv2dKArgs ka) {
  constexpr int NW = 8, NTH = 512, TX = 16;
  constexpr int R = TY / NW;
  constexpr int HX = (TX - 1) * S + KS, HY = (TY - 1) * S + KS;   // input halo tile
  constexpr int NPAR = S * S;                                     // parity planes (stride 2: 4)
  constexpr int PXW = (HX + S - 1) / S, PYH = (HY + S - 1) / S;   // positions per parity plane
  constexpr int NPP = PXW * PYH;
  constexpr int PLANE = ((NPAR * NPP + 15) / 16) * 16;            // positions per (sub, g) plane, 16-aligned
  constexpr int IN_FLOATS = NSUB * 4 * PLANE * 4;
  constexpr int NTAP = KS * KS;
  constexpr int W_SUB = NTAP * NT * 64 * 4;                       // weight floats per sub-block
  constexpr int W_FLOATS = NSUB * W_SUB;
  constexpr int NPIECE = HX * HY * 4 * NSUB;                      // 16-byte pieces of the input tile
  constexpr int IN_IT = (NPIECE + NTH - 1) / NTH;
  constexpr int W_IT = (W_FLOATS / 4 + NTH - 1) / NTH;
  static_assert(TY % NW == 0 && IN_IT <= 16, "tile config");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + IN_FLOATS;

  const vx_conv2d_args& a = ka.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int cg = blockIdx.y;
  const int ntiles = ka.tiles_x * ka.tiles_y;
  const int total = ntiles * a.N;
  const int lastx = (ka.tiles_x - 1) * TX, lasty = (ka.tiles_y - 1) * TY;  // output coords of the last tile
  const int nsub_all = a.Cin / 16;

  // ---- per-lane constants ----
  int vbase[R];
  unsigned ovoff[R];
  unsigned obad_xhi = 0, obad_yhi = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ly = wave * R + r;
    vbase[r] = (g * PLANE + ly * PXW + m) * 4;
    ovoff[r] = (unsigned)(((ly * ka.OW + m) * a.out_pitch + a.out_coff + g * 4) * 4);
    if (m >= ka.OW - lastx) obad_xhi |= 1u << r;
    if (ly >= ka.OH - lasty) obad_yhi |= 1u << r;
  }
  // staging pattern: piece idx -> (sub, hy, hx, q)
  const int rowf = a.W * a.in_pitch;
  const int biasf = KS == 1 ? 0 : rowf + a.in_pitch;   // one row + one pixel of padding offset
  unsigned voff[IN_IT];
  int ldst[IN_IT];
  unsigned ibad_always = 0, ibad_xlo = 0, ibad_xhi = 0, ibad_ylo = 0, ibad_yhi = 0;
  unsigned isub[IN_IT];
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int idx = tid + it * NTH;
    const int q = idx % 4;
    const int pix = (idx / 4) % (HX * HY);
    const int sub = idx / (4 * HX * HY);
    const int hx = pix % HX, hy = pix / HX;
    const int dxr = hx - KS / 2, dyr = hy - KS / 2;       // input pixel relative to the tile's input origin
    voff[it] = (unsigned)((dyr * rowf + dxr * a.in_pitch + sub * 16 + q * 4 + biasf) * 4);
    const int par = (hy % S) * S + (hx % S);
    ldst[it] = ((sub * 4 + q) * PLANE + par * NPP + (hy / S) * PXW + hx / S) * 4;
    isub[it] = (unsigned)sub;
    if (idx >= NPIECE) ibad_always |= 1u << it;
    if (dxr < 0) ibad_xlo |= 1u << it;
    if (dxr >= a.W - lastx * S) ibad_xhi |= 1u << it;
    if (dyr < 0) ibad_ylo |= 1u << it;
    if (dyr >= a.H - lasty * S) ibad_yhi |= 1u << it;
  }
  const size_t in_sample = (size_t)a.H * rowf;
  const size_t out_sample = (size_t)ka.OH * ka.OW * a.out_pitch;

  auto decode = [&](int t_, int& n, int& tx, int& ty) {
    unsigned t = (unsigned)t_, q;
    q = ka.tiles_x == 1 ? t : __umulhi(t, ka.mx); tx = (int)(t - q * ka.tiles_x); t = q;
    q = ka.tiles_y == 1 ? t : __umulhi(t, ka.my); ty = (int)(t - q * ka.tiles_y); n = (int)q;
  };

  const float* w_cg = a.w_packed + (size_t)cg * nsub_all * W_SUB;
  f32x4 ibuf[IN_IT], wbuf[W_IT];
  const bool w_resident = ka.nchunks == 1;
  bool w_fresh = true;

  auto prefetch = [&](int tile_lin, int chunk, bool have, bool with_w) {
    int n, tx, ty;
    decode(tile_lin, n, tx, ty);
    const int nsub = min(NSUB, nsub_all - chunk * NSUB);   // sub-blocks in this chunk
    unsigned bad = ibad_always;
    if (tx == 0) bad |= ibad_xlo;
    if (tx == ka.tiles_x - 1) bad |= ibad_xhi;
    if (ty == 0) bad |= ibad_ylo;
    if (ty == ka.tiles_y - 1) bad |= ibad_yhi;
    if (!have) bad = 0xFFFFFFFFu;
    const unsigned soff = (unsigned)(((ty * TY * S) * rowf + (tx * TX * S) * a.in_pitch + chunk * NSUB * 16) * 4);
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.in + (size_t)(have ? n : 0) * in_sample - biasf), 0, K_NUMREC, 0x00020000);
    // channels at and beyond min(Cin, in_pitch) read as zero: a tensor of C real channels may arrive with pitch round4(C)
    // (HRNet-W18's 18-channel branch: 20 floats per pixel instead of 32)
    const int clim = min(a.Cin, a.in_pitch) - chunk * NSUB * 16;
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const bool b = ((bad >> it) & 1u) || (int)isub[it] * 16 + (tid & 3) * 4 >= clim;   // NTH % 4 == 0: q = tid % 4
      const unsigned vo = b ? K_OOB : voff[it];
      ibuf[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)vo, (int)soff, 0));
    }
    const f32x4* src = reinterpret_cast<const f32x4*>(w_cg + (size_t)chunk * NSUB * W_SUB);
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int idx = tid + it * NTH;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (have && with_w && idx < nsub * (W_SUB / 4)) v = src[idx];
      wbuf[it] = v;
    }
  };
  auto commit = [&](bool with_w) {
#pragma unroll
    for (int it = 0; it < IN_IT; ++it)
      if (tid + it * NTH < NPIECE) *reinterpret_cast<f32x4*>(s_in + ldst[it]) = ibuf[it];
    if (with_w) {
#pragma unroll
      for (int it = 0; it < W_IT; ++it) {
        const int idx = tid + it * NTH;
        if (idx < W_FLOATS / 4) reinterpret_cast<f32x4*>(s_w)[idx] = wbuf[it];
      }
    }
  };

  f32x4 bias4[NT];
  bool cvalid[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = (cg * NT + nt) * 16 + g * 4;
    cvalid[nt] = co < a.Cout;
    bias4[nt] = (a.bias && cvalid[nt]) ? *reinterpret_cast<const f32x4*>(a.bias + co) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  int tile_lin = blockIdx.x, chunk = 0;
  bool have = tile_lin < total;
  prefetch(tile_lin, 0, have, true);
  f32x4 acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  while (have) {
    __syncthreads();
    commit(w_fresh);
    __syncthreads();
    w_fresh = !w_resident;
    const int nsub = min(NSUB, nsub_all - chunk * NSUB);
    int ntile = tile_lin, nchunk = chunk + 1;
    if (nchunk == ka.nchunks) { nchunk = 0; ntile = tile_lin + (int)gridDim.x; }
    const bool nhave = ntile < total;
    prefetch(ntile, nchunk, nhave, !w_resident);

    for (int sub = 0; sub < nsub; ++sub) {   // NSUB == 1: a single pass
      const float* sw = s_w + sub * W_SUB;
      const float* si = s_in + sub * 4 * PLANE * 4;
      f32x4 wf[2][NT], xf[2][R];
      auto load_tap = [&](int t1, int slot) {
        const int ky = t1 / KS, kx = t1 % KS;
        const int toff = (((ky % S) * S + (kx % S)) * NPP + (ky / S) * PXW + kx / S) * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[slot][nt] = *reinterpret_cast<const f32x4*>(sw + ((t1 * NT + nt) * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < R; ++r) xf[slot][r] = *reinterpret_cast<const f32x4*>(si + vbase[r] + toff);
      };
      load_tap(0, 0);
#pragma unroll
      for (int tap = 0; tap < NTAP; ++tap) {
        if (tap + 1 < NTAP) load_tap(tap + 1, (tap + 1) & 1);
        const int cur = tap & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cur][nt][j], xf[cur][r][j], acc[r][nt], 0, 0, 0);
      }
    }

    if (chunk == ka.nchunks - 1) {
      int n, tx, ty;
      decode(tile_lin, n, tx, ty);
      unsigned obad = 0;
      if (tx == ka.tiles_x - 1) obad |= obad_xhi;
      if (ty == ka.tiles_y - 1) obad |= obad_yhi;
      const unsigned osoff = (unsigned)((ty * TY * ka.OW + tx * TX) * a.out_pitch) * 4u;
      const __amdgpu_buffer_rsrc_t osrd =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * out_sample), 0, K_NUMREC, 0x00020000);
      float ssum[NT][4], ssq[NT][4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[nt][j] = 0.f; ssq[nt][j] = 0.f; }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned cshift = (unsigned)((cg * NT + nt) * 16);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const bool bad = ((obad >> r) & 1u) || !cvalid[nt];
          const f32x4 v = acc[r][nt] + bias4[nt];
          if (a.stats_partial && !bad) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { ssum[nt][j] += v[j]; ssq[nt][j] += v[j] * v[j]; }
          }
          const unsigned vo = bad ? K_OOB : ovoff[r] + cshift * 4u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), osrd, (int)vo, (int)osoff, 0);
          // gfx950 store-data hazard with an SGPR soffset (see conv3d_mfma.hip)
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 3" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (a.stats_partial) {
        float* s_red = smem + IN_FLOATS + W_FLOATS;  // [NW][NT][16][2]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float s = ssum[nt][j], q = ssq[nt][j];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
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
            float* dst = a.stats_partial + ((size_t)tile_lin * a.Cout + co) * 2;
            dst[0] = s;
            dst[1] = q;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    tile_lin = ntile; chunk = nchunk; have = nhave;
  }
}

// ---------------------------------------------------------------------------------------------------------------
struct C2Cfg { int NT, NSUB, TY; };
static inline C2Cfg c2_config(int KS, int S, int Cout) {
  C2Cfg c;
  c.NT = (Cout % 48 == 0) ? 3 : ((Cout % 32 == 0) ? 2 : 1);
  c.NSUB = KS == 1 ? 4 : 1;
  c.TY = 16;
  return c;
}

__global__ void pack_conv2d_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cin_pad, int Cout,
                                   int KS, int NT, int64_t total) {
  const int ntap = KS * KS, nsub = Cin_pad / 16;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int j = r % 4; r /= 4;
    const int lane = r % 64; r /= 64;
    const int nt = r % NT; r /= NT;
    const int tap = r % ntap; r /= ntap;
    const int sub = r % nsub; r /= nsub;
    const int rgrp = (int)r;
    const int row = (rgrp * NT + nt) * 16 + (lane & 15);
    const int ci = sub * 16 + (lane >> 4) * 4 + j;
    float v = 0.f;
    if (row < Cout && ci < Cin) v = w[((size_t)row * Cin + ci) * ntap + tap];
    out[i] = v;
  }
}

static inline int c2_rows_padded(int Cout, int NT) { return ((Cout + 16 * NT - 1) / (16 * NT)) * (16 * NT); }

// Kernel family = packed layout under the current configuration: 1 split-fp16, 2 split-fp16 with five row tiles
// (wide 1x1 layers), 3 native fp32
extern "C" int vx_conv2d_family(int Cin, int Cout, int KS) {
  if (Cin <= 0 || Cout <= 0 || (KS != 1 && KS != 3)) return 0;
  if (!c2_split16()) return 3;
  // the packed layout is [row group][...][row tile]: bound to the tile count; + 100 x the octets of the octet-granular K
  // schedule when the layer's REAL input channel count is packed for it (vx_conv2d_s16_octets)
  return 10 + vx_conv2d_s16_row_tiles(KS, Cout) + 100 * vx_conv2d_s16_octets(Cin, KS);
}

extern "C" int64_t vx_conv2d_packed_floats(int Cin, int Cout, int KS) {
  if (Cin <= 0 || Cout <= 0 || (KS != 1 && KS != 3)) return -1;
  if (c2_split16()) return vx_conv2d_s16_packed_floats(Cin, Cout, KS);
  const int cin_pad = (Cin + 15) / 16 * 16;
  C2Cfg c = c2_config(KS, 1, Cout);
  return (int64_t)c2_rows_padded(Cout, c.NT) * cin_pad * KS * KS;
}

extern "C" int vx_pack_conv2d(const float* w_torch, float* w_packed, int Cin, int Cout, int KS, vx_stream_t stream) {
  if (!w_torch || !w_packed) VX_FAIL(VX_E_NULL, "vx_pack_conv2d: null pointer");
  const int64_t total = vx_conv2d_packed_floats(Cin, Cout, KS);
  if (total < 0) VX_FAIL(VX_E_SHAPE, "vx_pack_conv2d: Cin=%d Cout=%d KS=%d", Cin, Cout, KS);
  if (c2_split16()) return vx_pack_conv2d_s16(w_torch, w_packed, Cin, Cout, KS, (hipStream_t)stream);
  C2Cfg c = c2_config(KS, 1, Cout);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_conv2d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_torch, w_packed, Cin,
                     (Cin + 15) / 16 * 16, Cout, KS, c.NT, total);
  VX_CHECK_LAUNCH("vx_pack_conv2d");
  return VX_OK;
}

extern "C" int vx_conv2d_tiles(int H, int W, int KS, int S) {
  const int OH = (H + 2 * (KS / 2) - KS) / S + 1, OW = (W + 2 * (KS / 2) - KS) / S + 1;
  const int TY = 16;
  return ((OW + 15) / 16) * ((OH + TY - 1) / TY);
}

template <int KS, int S, int NT, int NSUB, int TY>
static int launch_c2(const Conv2dKArgs& ka, hipStream_t s) {
  constexpr int HX = 15 * S + KS, HY = (TY - 1) * S + KS;
  constexpr int NPP = ((HX + S - 1) / S) * ((HY + S - 1) / S);
  constexpr int PLANE = ((S * S * NPP + 15) / 16) * 16;
  constexpr int IN_FLOATS = NSUB * 4 * PLANE * 4;
  constexpr int W_FLOATS = NSUB * KS * KS * NT * 64 * 4;
  constexpr size_t lds = (size_t)(IN_FLOATS + W_FLOATS + 8 * NT * 16 * 2) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  auto kern = conv2d_mfma_kernel<KS, S, NT, NSUB, TY>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv2d: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr_set = true;
  }
  const vx_conv2d_args& a = ka.a;
  const int total_tiles = ka.tiles_x * ka.tiles_y * a.N;
  const int ygroups = (a.Cout + 16 * NT - 1) / (16 * NT);
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  int gx = (256 * per_cu + ygroups - 1) / ygroups;
  if (gx > total_tiles) gx = total_tiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ygroups), dim3(512), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv2d");
  return VX_OK;
}

template <int KS, int S, int NSUB, int TY>
static int dispatch_c2(const Conv2dKArgs& ka, int NT, hipStream_t s) {
  if (NT == 3) return launch_c2<KS, S, 3, NSUB, TY>(ka, s);
  if (NT == 2) return launch_c2<KS, S, 2, NSUB, TY>(ka, s);
  return launch_c2<KS, S, 1, NSUB, TY>(ka, s);
}

extern "C" int vx_conv2d(const vx_conv2d_args* ap, vx_stream_t stream) {
  if (!ap) VX_FAIL(VX_E_NULL, "vx_conv2d: null args");
  const vx_conv2d_args& a = *ap;
  if (!a.in || !a.w_packed || !a.out) VX_FAIL(VX_E_NULL, "vx_conv2d: null tensor pointer");
  if (a.N <= 0 || a.H <= 0 || a.W <= 0) VX_FAIL(VX_E_SHAPE, "vx_conv2d: empty tensor");
  if (a.Cin <= 0 || a.Cin % 16 || a.Cout <= 0)
    VX_FAIL(VX_E_SHAPE, "vx_conv2d: Cin=%d must be a positive multiple of 16 (pad the input), Cout=%d positive", a.Cin, a.Cout);
  if ((a.KS != 1 && a.KS != 3) || (a.S != 1 && a.S != 2) || (a.KS == 1 && a.S != 1))
    VX_FAIL(VX_E_SHAPE, "vx_conv2d: kernel %d stride %d unsupported (3x3 s1/s2, 1x1 s1)", a.KS, a.S);
  {
    // a.Cin is the PADDED channel count; weights packed for the octet-granular schedule carry their octet count in the
    // family (the real count was 8 oct - 7 .. 8 oct): accepted when it pads to this Cin
    const int fam = vx_conv2d_family(a.Cin, a.Cout, a.KS), oct = a.w_family / 100;
    const bool octet_ok = oct > 0 && c2_split16() && a.KS == 3 && (oct * 8 + 15) / 16 * 16 == a.Cin &&
                          vx_conv2d_s16_octets(oct * 8, a.KS) == oct && a.w_family % 100 == fam % 100;
    if (a.w_family != fam && !octet_ok)
      VX_FAIL(VX_E_DTYPE, "vx_conv2d: weights packed for kernel family %d, the library is configured for family %d: re-pack them",
              a.w_family, fam);
  }
  // outputs leave as 16-byte groups of 4 channels: a Cout that is not a multiple of 4 writes its last group up to
  // round4(Cout) (zeros beyond Cout), which the pitch must cover
  if (a.in_pitch <= a.Cin - 16 || a.in_pitch % 4 || a.out_pitch < a.out_coff + (a.Cout + 3) / 4 * 4 || a.out_pitch % 4 || a.out_coff % 4)
    VX_FAIL(VX_E_ALIGN, "vx_conv2d: pitches/offsets must be multiples of 4 floats and cover the channels");
  if (!vx_aligned16(a.in) || !vx_aligned16(a.out) || !vx_aligned16(a.w_packed) || (a.bias && !vx_aligned16(a.bias)))
    VX_FAIL(VX_E_ALIGN, "vx_conv2d: pointers must be 16-byte aligned");
  Conv2dKArgs ka;
  ka.a = a;
  ka.OH = (a.H + 2 * (a.KS / 2) - a.KS) / a.S + 1;
  ka.OW = (a.W + 2 * (a.KS / 2) - a.KS) / a.S + 1;
  if ((int64_t)(a.H + 2) * a.W * a.in_pitch * 4 >= (1ll << 31) || (int64_t)ka.OH * ka.OW * a.out_pitch * 4 >= (1ll << 31))
    VX_FAIL(VX_E_SHAPE, "vx_conv2d: one image must stay below 2 GiB");
  if ((a.in_scale == nullptr) != (a.in_shift == nullptr)) VX_FAIL(VX_E_NULL, "vx_conv2d: in_scale / in_shift must come together");
  if (a.in_scale && (a.in_cpitch < a.Cin - 15 || a.in_group_images < 0 || !c2_split16()))
    VX_FAIL(VX_E_SHAPE, "vx_conv2d: the input prologue needs rows of in_cpitch >= the real channel count and the split-fp16 "
            "kernels (vx_config.conv_fp32 = 0)");
  if (c2_split16()) return vx_conv2d_s16(a, (hipStream_t)stream);
  C2Cfg c = c2_config(a.KS, a.S, a.Cout);
  ka.tiles_x = (ka.OW + 15) / 16;
  ka.tiles_y = (ka.OH + c.TY - 1) / c.TY;
  ka.nchunks = (a.Cin / 16 + c.NSUB - 1) / c.NSUB;
  ka.mx = (unsigned)((1ull << 32) / (unsigned)ka.tiles_x) + 1u;
  ka.my = (unsigned)((1ull << 32) / (unsigned)ka.tiles_y) + 1u;
  hipStream_t s = (hipStream_t)stream;
  if (a.KS == 3 && a.S == 1) return dispatch_c2<3, 1, 1, 16>(ka, c.NT, s);
  if (a.KS == 3 && a.S == 2) return dispatch_c2<3, 2, 1, 16>(ka, c.NT, s);
  return dispatch_c2<1, 1, 4, 16>(ka, c.NT, s);
}
