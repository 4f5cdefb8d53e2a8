// K15, split-precision schedule: the 2D convolutions of conv2d_mfma.hip (HRNet: 3x3 s1, 3x3 s2, 1x1) with every
// fp32 product evaluated on the fp16 matrix cores by operand splitting, exactly as conv3d_s16.hip does for the 3D
// network (x = hi + lo * 2^-11; three v_mfma_f32_16x16x32_f16 per K = 32 step; fp32 accumulation in a main and a
// cross accumulator; measured closer to float64 than the native fp32 matrix instruction).
//
//   K = 32 step:  3x3: two taps x one sub-block of 16 channels (9 taps + 1 zero-weight pad = 5 steps per sub-block);
//                 1x1: two sub-blocks of 16 channels (NSUB = 4 -> 2 steps per item)
//                 lane k-group kg = 2 * (tap or sub-block parity) + channel octet
//   LDS image:    hi plane + lo plane, each [sub-block][octet][parity plane | position][8 halves]
//   weights:      [row group][chunk][step][row tile][hi | lo][lane][8 halves]
// Everything else (work items, register prefetch, stride-2 parity planes, epilogue with BatchNorm batch-statistics
// partials) is conv2d_mfma.hip's.
#include "common.h"
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct Conv2dSArgs {
  vx_conv2d_args a;
  int OH, OW, tiles_x, tiles_y, nchunks;   // nchunks = ceil(Cin/16 / NSUB)
  int w_all;                               // every chunk's weights stay in LDS for the kernel's life (launch_c2s: they fit)
  unsigned mx, my;
};

namespace {
constexpr unsigned K_OOB = 0xFFFFFFF0u;
constexpr unsigned K_NUMREC = 0x80000000u;

__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo) {   // conv3d_s16.hip: vx_split4
#pragma unroll
  for (int j = 0; j < 4; j += 2) {
    const f32x2 x = {v[j], v[j + 1]};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    // lo = fp16(2048 x - 2048 hi): one mixed-precision fma per element (same bits as the cvt / sub / mul / cvt form)
    const f32x2 xs = x * 2048.f;
    const float m2048 = -2048.f;
    const uint32_t hv = __builtin_bit_cast(uint32_t, h);
    uint32_t lv = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lv) : "v"(hv), "v"(m2048), "v"(xs[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lv) : "v"(hv), "v"(m2048), "v"(xs[1]));
    const f16x2 l = __builtin_bit_cast(f16x2, lv);
    hi[j] = h[0]; hi[j + 1] = h[1];
    lo[j] = l[0]; lo[j + 1] = l[1];
  }
}
}

// lane i of every 16-lane row <- lane (i + rot) % 16 of the same row (DPP; conv3d_s16.hip: vx_row_ror)
__device__ __forceinline__ float row_ror(float x, int rot) {
  const int v = __builtin_bit_cast(int, x);
  int r;
  switch (rot) {
    case 8: r = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true); break;
    case 4: r = __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, true); break;
    default: r = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, true); break;
  }
  return __builtin_bit_cast(float, r);
}

// OCT > 0 (3x3 only, round 3): OCTET-granular K.  The legacy schedule spends a K = 32 step on two taps x one 16-channel
// sub-block, so an 18-channel layer (two sub-blocks) takes 10 steps and the 3-channel image 5; here a step holds four
// (tap, octet) units of 8 channels, u = 4 step + k-group -> tap = u / OCT, octet = u % OCT: 18 channels (OCT = 3) take 7
// steps, up to 8 channels (OCT = 1) 3.  One work item per tile, weights resident; the packed layout is its own
// (pack_conv2d_s16_kernel, vx_conv2d_family = ... + 100 OCT).
template <int KS, int S, int NT, int NSUB, int TY, int OCT = 0>
__global__ __launch_bounds__(512) void conv2d_s16_kernel(Conv2dSArgs ka) {
  constexpr int NW = 8, NTH = 512, TX = 16;
  constexpr int R = TY / NW;
  constexpr int HX = (TX - 1) * S + KS, HY = (TY - 1) * S + KS;   // input halo tile
  constexpr int NPAR = S * S;                                     // parity planes (stride 2: 4)
  constexpr int PXW = (HX + S - 1) / S, PYH = (HY + S - 1) / S;   // positions per parity plane
  constexpr int NPP = PXW * PYH;
  constexpr int PLANE = ((NPAR * NPP + 15) / 16) * 16;            // positions per (sub, octet) plane, 16-aligned
  static_assert(OCT == 0 || (KS == 3 && NSUB == 1), "octet-granular K: 3x3 layers, one item per tile");
  constexpr int NOCTP = OCT ? OCT : 2 * NSUB;                     // octet planes of the LDS image
  constexpr int QPP = 2 * NOCTP;                                  // 16-byte quads staged per pixel
  constexpr int IMG_H = NOCTP * PLANE * 8;                        // halves per precision plane
  constexpr int IN_FLOATS = IMG_H;                                // (hi + lo planes = 2 * IMG_H halves = IMG_H floats)
  constexpr int NSTEP = KS == 3 ? (OCT ? (9 * OCT + 3) / 4 : 5 * NSUB) : NSUB / 2;   // K = 32 steps per item
  constexpr int TABC = OCT ? ((OCT * 8 + 15) / 16) * 16 : NSUB * 16;                 // channels of the prologue table
  constexpr int W_STEP = NT * 2 * 64 * 8;                         // weight halves per step
  constexpr int W_FLOATS = NSTEP * W_STEP / 2;                    // per item (chunk), in floats
  static_assert(KS == 3 || NSUB % 2 == 0, "1x1: sub-blocks are consumed in pairs");
  constexpr int NPIECE = HX * HY * QPP;                           // 16-byte pieces of the input tile
  constexpr int IN_IT = (NPIECE + NTH - 1) / NTH;
  constexpr int W_IT = (W_FLOATS / 4 + NTH - 1) / NTH;
  static_assert(TY % NW == 0 && IN_IT <= 16, "tile config");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* s_hi = reinterpret_cast<_Float16*>(smem);
  _Float16* s_lo = s_hi + IMG_H;
  _Float16* s_w = s_hi + 2 * IMG_H;

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
    vbase[r] = ly * PXW + m;    // position within a (sub, octet) plane
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
  // (the piece's channel quad rides in bits 24.. of ldst: one register per piece less across the multiply phase)
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int idx = tid + it * NTH;
    // qq = the piece's channel quad (channels 4 qq .. 4 qq + 3).  Legacy order [sub-block][pixel][quad of the block] (a
    // thread's quad within its block is tid % 4 in every iteration); octet mode [pixel][quad]
    const int pix = OCT ? (idx / QPP) % (HX * HY) : (idx / 4) % (HX * HY);
    const int qq = OCT ? idx % QPP : (idx / (4 * HX * HY)) * 4 + idx % 4;
    const int hx = pix % HX, hy = pix / HX;
    const int dxr = hx - KS / 2, dyr = hy - KS / 2;       // input pixel relative to the tile's input origin
    voff[it] = (unsigned)((dyr * rowf + dxr * a.in_pitch + qq * 4 + biasf) * 4);
    const int par = (hy % S) * S + (hx % S);
    ldst[it] = ((qq >> 1) * PLANE + par * NPP + (hy / S) * PXW + hx / S) * 8 + (qq & 1) * 4;   // halves
    ldst[it] |= qq << 24;
    if (idx >= NPIECE) ibad_always |= 1u << it;
    if (dxr < 0) ibad_xlo |= 1u << it;
    if (dxr >= a.W - lastx * S) ibad_xhi |= 1u << it;
    if (dyr < 0) ibad_ylo |= 1u << it;
    if (dyr >= a.H - lasty * S) ibad_yhi |= 1u << it;
  }
  // this lane's (tap | sub-block, octet) of every step, as a position offset into a precision plane.  3x3: the five steps
  // of a sub-block repeat for the next one 2 PLANE positions further -- five registers, the rest an immediate offset
  constexpr int NTOFF = OCT ? NSTEP : (KS == 3 ? 5 : NSTEP);
  int toff[NTOFF];
#pragma unroll
  for (int s = 0; s < NTOFF; ++s) {
    const int oct = g & 1;
    if (OCT) {
      int u = 4 * s + g;                          // this lane's (tap, octet) unit of the step
      if (u > 9 * OCT - 1) u = 9 * OCT - 1;       // zero-weight padding units re-read a valid position
      const int tap = u / OCT, o = u % OCT;
      const int ky = tap / 3, kx = tap % 3;
      toff[s] = o * PLANE + ((ky % S) * S + (kx % S)) * NPP + (ky / S) * PXW + kx / S;
    } else if (KS == 3) {
      int tap = 2 * s + (g >> 1);
      if (tap > 8) tap = 8;                       // zero-weight padding tap: re-read a valid position
      const int ky = tap / 3, kx = tap % 3;
      toff[s] = oct * PLANE + ((ky % S) * S + (kx % S)) * NPP + (ky / S) * PXW + kx / S;
    } else {
      toff[s] = ((2 * s + (g >> 1)) * 2 + oct) * PLANE;
    }
  }
  const size_t in_sample = (size_t)a.H * rowf;
  const size_t out_sample = (size_t)ka.OH * ka.OW * a.out_pitch;

  auto decode = [&](int t_, int& n, int& tx, int& ty) {
    unsigned t = (unsigned)t_, q;
    q = ka.tiles_x == 1 ? t : __umulhi(t, ka.mx); tx = (int)(t - q * ka.tiles_x); t = q;
    q = ka.tiles_y == 1 ? t : __umulhi(t, ka.my); ty = (int)(t - q * ka.tiles_y); n = (int)q;
  };

  const float* w_cg = a.w_packed + (size_t)cg * ka.nchunks * W_FLOATS;
  // whole-Cin items (3x3, NSUB > 1: vx_conv2d_s16 runs them with ONE chunk): the weights are copied once, no weight
  // prefetch registers live across the multiply phase
  constexpr bool WRES = (KS == 3 && NSUB > 1) || OCT > 0;
  f32x4 ibuf[IN_IT], wbuf[WRES ? 1 : W_IT];
  // prologue (vx_conv2d_args.in_scale): the folded BatchNorm of the PRODUCING conv (+ ReLU) applied on the way into LDS.
  // The scale / shift rows of the item's image group are staged into a small LDS table by prefetch() (after the second
  // barrier of the previous item: commit() of this item reads them behind the next barrier).
  const bool pre = a.in_scale != nullptr;
  const int w_res_floats = (ka.w_all ? ka.nchunks : 1) * W_FLOATS;   // LDS floats the weights take
  float* s_ss = smem + IN_FLOATS + w_res_floats + NW * NT * 16 * 2;  // [scale | shift][TABC] of the item's chunk
  unsigned p_bad = 0;
  const bool w_resident = ka.nchunks == 1 || ka.w_all;
  bool w_fresh = true;

  auto prefetch = [&](int tile_lin, int chunk, bool have, bool with_w) {
    int n, tx, ty;
    decode(tile_lin, n, tx, ty);
    unsigned bad = ibad_always;
    if (tx == 0) bad |= ibad_xlo;
    if (tx == ka.tiles_x - 1) bad |= ibad_xhi;
    if (ty == 0) bad |= ibad_ylo;
    if (ty == ka.tiles_y - 1) bad |= ibad_yhi;
    if (!have) bad = 0xFFFFFFFFu;
    const unsigned soff = (unsigned)(((ty * TY * S) * rowf + (tx * TX * S) * a.in_pitch + chunk * NSUB * 16) * 4);
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.in + (size_t)(have ? n : 0) * in_sample - biasf), 0, K_NUMREC, 0x00020000);
    // channels at and beyond min(Cin, in_pitch) read as zero (conv2d_mfma.hip)
    const int clim = min(a.Cin, a.in_pitch) - chunk * NSUB * 16;
    unsigned cbad = 0;
#pragma unroll
    for (int it = 0; it < IN_IT; ++it)
      if ((ldst[it] >> 24) * 4 >= clim) cbad |= 1u << it;
    bad |= cbad;
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const bool b = (bad >> it) & 1u;
      const unsigned vo = b ? K_OOB : voff[it];
      ibuf[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)vo, (int)soff, 0));
    }
    if (pre) {
      p_bad = bad;
      // this chunk's channels of the image's statistics group (channels beyond Cin: scale = shift = 0 -> zeros)
      if (tid < 2 * TABC) {
        const int which = tid / TABC, c = tid % TABC, ch = chunk * NSUB * 16 + c;
        const int grp = a.in_group_images > 0 ? (have ? n : 0) / a.in_group_images : 0;
        const float* row = (which ? a.in_shift : a.in_scale) + (size_t)grp * a.in_cpitch;
        s_ss[tid] = ch < min(a.Cin, a.in_cpitch) ? row[ch] : 0.f;
      }
    }
    if constexpr (!WRES) {
      const f32x4* src = reinterpret_cast<const f32x4*>(w_cg + (size_t)chunk * W_FLOATS);
#pragma unroll
      for (int it = 0; it < W_IT; ++it) {
        const int idx = tid + it * NTH;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (have && with_w && idx < W_FLOATS / 4) v = src[idx];
        wbuf[it] = v;
      }
    }
  };
  auto commit = [&](bool with_w) {
#pragma unroll
    for (int it = 0; it < IN_IT; ++it)
      if (tid + it * NTH < NPIECE) {
        f16x4 hi, lo;
        if (pre) {
          const int c = (ldst[it] >> 24) * 4;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(s_ss + c);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(s_ss + TABC + c);
          f32x4 v = ibuf[it];
          const bool zero = (p_bad >> it) & 1u;                        // zero padding belongs to the activated tensor
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = v[j] * sc[j] + sh[j];                            // vx_affine_gather's arithmetic (two roundings: -ffp-contract=off)
            if (a.in_relu) t = fmaxf(t, 0.f);
            v[j] = zero ? 0.f : t;
          }
          ibuf[it] = v;
        }
        split4(ibuf[it], hi, lo);
        *reinterpret_cast<f16x4*>(s_hi + (ldst[it] & 0xFFFFFF)) = hi;
        *reinterpret_cast<f16x4*>(s_lo + (ldst[it] & 0xFFFFFF)) = lo;
      }
    if constexpr (!WRES) {
      if (with_w) {
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
          const int idx = tid + it * NTH;
          if (idx < W_FLOATS / 4) reinterpret_cast<f32x4*>(s_w)[idx] = wbuf[it];
        }
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
  if (WRES || (ka.w_all && ka.nchunks > 1)) {   // all chunks' weights resident: one cooperative copy for the kernel's life (conv3d_s16.hip).
    // Per item the 3x3 layers of 18 / 36 channels otherwise re-stage 20 / 30 KB of weights next to a 20-KB image
    const f32x4* src = reinterpret_cast<const f32x4*>(w_cg);
    for (int i = tid; i < ka.nchunks * (W_FLOATS / 4); i += NTH) reinterpret_cast<f32x4*>(s_w)[i] = src[i];
    w_fresh = false;
  }
  prefetch(tile_lin, 0, have, !(WRES || (ka.w_all && ka.nchunks > 1)));
  f32x4 acc[R][NT], accx[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; accx[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  // statistics of the tile finished last: its per-wave sums sit in s_red and are combined after the NEXT barrier
  // of the item loop (no barrier of their own in the epilogue; conv3d_s16.hip)
  float* s_red = smem + IN_FLOATS + w_res_floats;  // [NW][NT][16][2] (floats: image = IMG_H, weights = W_FLOATS per chunk)
  int pend_tile = -1;
  auto flush_stats = [&]() {
    if (pend_tile >= 0 && tid < NT * 16) {
      const int nt = tid / 16, c = tid % 16;
      const int co = (cg * NT + nt) * 16 + c;
      if (co < a.Cout) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          s += s_red[((w * NT + nt) * 16 + c) * 2 + 0];
          q += s_red[((w * NT + nt) * 16 + c) * 2 + 1];
        }
        float* dst = a.stats_partial + ((size_t)pend_tile * a.Cout + co) * 2;
        dst[0] = s;
        dst[1] = q;
      }
    }
    pend_tile = -1;
  };

  while (have) {
    __syncthreads();
    flush_stats();
    commit(w_fresh);
    __syncthreads();
    w_fresh = !w_resident;
    const int nsub = min(NSUB, nsub_all - chunk * NSUB);
    int ntile = tile_lin, nchunk = chunk + 1;
    if (nchunk == ka.nchunks) { nchunk = 0; ntile = tile_lin + (int)gridDim.x; }
    const bool nhave = ntile < total;
    prefetch(ntile, nchunk, nhave, !w_resident);

    {
      // ---- NSTEP steps x 3 x R x NT MFMAs; fragments of step s + 1 are read before the MFMAs of step s.  (1x1:
      // sub-blocks beyond the layer's Cin were steered out of range by prefetch -> zeros; their weights are zero too)
      (void)nsub;
      f16x8 ah[2][NT], al[2][NT], bh[2][R], bl[2][R];
      const _Float16* s_wc = s_w + (ka.w_all ? chunk * (2 * W_FLOATS) : 0);   // this item's weights
      auto load_step = [&](int s, int slot) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const _Float16* wp = s_wc + s * W_STEP + ((nt * 2) * 64 + lane) * 8;
          ah[slot][nt] = *reinterpret_cast<const f16x8*>(wp);
          al[slot][nt] = *reinterpret_cast<const f16x8*>(wp + 64 * 8);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          // output row r of the wave = PXW positions further: one address register per step, the rest is the
          // instruction's immediate offset (conv3d_s16.hip)
          const int p = (vbase[0] + toff[(KS == 3 && !OCT) ? s % 5 : s]) * 8 + ((KS == 3 && !OCT) ? (s / 5) * 2 * PLANE * 8 : 0) + r * PXW * 8;
          bh[slot][r] = *reinterpret_cast<const f16x8*>(s_hi + p);
          bl[slot][r] = *reinterpret_cast<const f16x8*>(s_lo + p);
        }
      };
      load_step(0, 0);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        if (s + 1 < NSTEP) load_step(s + 1, (s + 1) & 1);
        const int cur = s & 1;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][nt], bh[cur][r], acc[r][nt], 0, 0, 0);
            accx[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][nt], bl[cur][r], accx[r][nt], 0, 0, 0);
            accx[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur][nt], bh[cur][r], accx[r][nt], 0, 0, 0);
          }
#ifndef VX_S16_NO_PIN
        // the schedule, pinned (conv3d_s16.hip): one fragment read of step s + 1 behind each of this step's first matrix
        // instructions -- left alone hipcc sinks every read to just before its consumer and the LDS latency is exposed
        // (not the five-row-tile 1x1 instance: it spills as it is, and ten times more with two sets of fragments live)
        // (nor the stride-2, three-row-tile octet instance: 13 staged pieces per thread leave no room for the second set)
        // (round 4: every stride-2 three-row-tile instance -- the sub-block one sat at 256 VGPRs with 3 spilled into scratch)
        if constexpr (!(KS == 1 && NT >= 5) && !(S == 2 && NT == 3)) {
          constexpr int NRD = 2 * NT + 2 * R, NMF = 3 * R * NT;
          constexpr int PAIRS = NRD < NMF ? NRD : NMF;
          if (s == 0) __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
          if (s + 1 < NSTEP) {
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (NRD > PAIRS) __builtin_amdgcn_sched_group_barrier(0x100, NRD - PAIRS, 0);
            if (NMF > PAIRS) __builtin_amdgcn_sched_group_barrier(0x008, NMF - PAIRS, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
          }
        }
#endif
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
          f32x4 v;     // main + cross * 2^-11: one fma per element (exact scaling: the bits of multiply-then-add)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaf(accx[r][nt][j], 1.0f / 2048.f, acc[r][nt][j]) + bias4[nt][j];
          if (a.stats_partial && !bad) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { ssum[nt][j] += v[j]; ssq[nt][j] = fmaf(v[j], v[j], ssq[nt][j]); }
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
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float s = ssum[nt][j], q = ssq[nt][j];
#pragma unroll
            for (int rot = 8; rot >= 1; rot >>= 1) { s += row_ror(s, rot); q += row_ror(q, rot); }
            if (m == 0) {
              s_red[((wave * NT + nt) * 16 + g * 4 + j) * 2 + 0] = s;
              s_red[((wave * NT + nt) * 16 + g * 4 + j) * 2 + 1] = q;
            }
          }
        pend_tile = tile_lin;
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; accx[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
    tile_lin = ntile; chunk = nchunk; have = nhave;
  }
  __syncthreads();
  flush_stats();
}

// ---------------------------------------------------------------------------------------------------------------
struct C2SCfg { int NT, NSUB, TY; };
static inline C2SCfg c2s_config(int KS, int S, int Cout) {
  C2SCfg c;
  // row tiles (16 output channels each) per workgroup: every workgroup of a row group stages the SAME input tile and
  // reads the same image fragments, so fewer groups are better as long as the padding they need (rows beyond Cout hold
  // zero weights and are not stored) costs less: minimise groups x (staging + tiles) with the staging of a 1x1 layer
  // weighing ~3 tiles of products and that of a 3x3 layer ~0.7.  HRNet-W18: 18 -> 2 tiles, 36 -> 3, 72 -> 5, 144 -> 3 x 3,
  // the 270 -> 270 head conv 4 groups of 5 (it ran as 17 groups of one tile: 4.8 ms of a 43 ms step); W48: 48 / 96 /
  // 192 / 384 -> 3 tiles, the 720 -> 720 head conv 9 x 5.  Five tiles only on 1x1 layers: the 3x3 instance needs 256
  // VGPRs plus scratch.
  const int r16 = (Cout + 15) / 16;
  const float cs = KS == 1 ? 3.0f : 0.7f;
  float bestc = 1e30f;
  c.NT = 1;
  for (int nt : {1, 2, 3, 5}) {
    if (nt == 5 && KS == 3) continue;   // 3x3 with five tiles: 256 VGPRs and scratch
    const float cost = (float)((r16 + nt - 1) / nt) * (cs + (float)nt);
    if (cost < bestc) { bestc = cost; c.NT = nt; }   // ties: the smaller tile count (less padding)
  }
  c.NSUB = KS == 1 ? 4 : 1;
  c.TY = 16;
  return c;
}
int vx_conv2d_s16_row_tiles(int KS, int Cout) { return c2s_config(KS, 1, Cout).NT; }
// octets of the octet-granular K schedule (conv2d_s16_kernel's OCT) a 3x3 layer of Cin REAL input channels is packed for;
// 0 = the sub-block schedule.  1: up to 8 channels (the image: 3 instead of 5 K steps); 3: 17..24 channels (HRNet-W18's
// full-resolution branch: 7 instead of 10).  Two octets are one sub-block (the same 5 steps), 4 and 6 save one step of 10 /
// 15, and five octets (36 channels) would save 3 of 15 but do not fit beside the stride-2 parity planes: not built.
int vx_conv2d_s16_octets(int Cin, int KS) {
  if (KS != 3 || vx_cfg().c2s_no_oct) return 0;
  if (Cin <= 8) return 1;
  if (Cin > 16 && Cin <= 24) return 3;
  return 0;
}
static inline int c2s_rows_padded(int Cout, int NT) { return ((Cout + 16 * NT - 1) / (16 * NT)) * (16 * NT); }

// torch (Cout, Cin, KS, KS) fp32 -> [row group][chunk][step][nt][hi | lo][lane][8 halves]; OCT > 0: one chunk, k-group kg of
// step s = unit u = 4 s + kg -> (tap u / OCT, octet u % OCT), units beyond 9 OCT are zeros
__global__ void pack_conv2d_s16_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int Cin, int Cout, int KS,
                                       int NT, int NSUB, int nchunks, int64_t total, int OCT) {
  const int nstep = OCT ? (9 * OCT + 3) / 4 : (KS == 3 ? 5 * NSUB : NSUB / 2), ntap = KS * KS;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int j = r % 8; r /= 8;
    const int lane = r % 64; r /= 64;
    const int hl = r % 2; r /= 2;
    const int nt = r % NT; r /= NT;
    const int step = r % nstep; r /= nstep;
    const int chunk = r % nchunks; r /= nchunks;
    const int rgrp = (int)r;
    const int row = (rgrp * NT + nt) * 16 + (lane & 15);
    const int kg = lane >> 4;
    int tap, sub, ci;
    if (OCT) {
      const int u = 4 * step + kg;
      tap = u < 9 * OCT ? u / OCT : ntap;
      ci = (u % OCT) * 8 + j;
    } else {
      if (KS == 3) { sub = chunk * NSUB + step / 5; tap = 2 * (step % 5) + (kg >> 1); }
      else { sub = chunk * NSUB + 2 * step + (kg >> 1); tap = 0; }
      ci = sub * 16 + 8 * (kg & 1) + j;
    }
    float v = 0.f;
    if (row < Cout && ci < Cin && tap < ntap) v = w[((size_t)row * Cin + ci) * ntap + tap];
    const _Float16 h = (_Float16)v;
    out[i] = hl == 0 ? h : (_Float16)((v - (float)h) * 2048.f);
  }
}

int64_t vx_conv2d_s16_packed_floats(int Cin, int Cout, int KS) {
  const C2SCfg c = c2s_config(KS, 1, Cout);
  const int oct = vx_conv2d_s16_octets(Cin, KS);
  const int nsub_all = (Cin + 15) / 16, nchunks = oct ? 1 : (nsub_all + c.NSUB - 1) / c.NSUB;
  const int nstep = oct ? (9 * oct + 3) / 4 : (KS == 3 ? 5 * c.NSUB : c.NSUB / 2);
  return (int64_t)(c2s_rows_padded(Cout, c.NT) / 16 / c.NT) * nchunks * nstep * c.NT * 2 * 64 * 8 / 2;
}

int vx_pack_conv2d_s16(const float* w_torch, float* w_packed, int Cin, int Cout, int KS, hipStream_t s) {
  const C2SCfg c = c2s_config(KS, 1, Cout);
  const int oct = vx_conv2d_s16_octets(Cin, KS);
  const int nsub_all = (Cin + 15) / 16, nchunks = oct ? 1 : (nsub_all + c.NSUB - 1) / c.NSUB;
  const int64_t total = vx_conv2d_s16_packed_floats(Cin, Cout, KS) * 2;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_conv2d_s16_kernel, dim3(blocks), dim3(256), 0, s, w_torch, reinterpret_cast<_Float16*>(w_packed), Cin,
                     Cout, KS, c.NT, c.NSUB, nchunks, total, oct);
  VX_CHECK_LAUNCH("vx_pack_conv2d(s16)");
  return VX_OK;
}

template <int KS, int S, int NT, int NSUB, int TY, int OCT = 0>
static int launch_c2s(const Conv2dSArgs& ka_in, hipStream_t s) {
  constexpr int HX = 15 * S + KS, HY = (TY - 1) * S + KS;
  constexpr int NPP = ((HX + S - 1) / S) * ((HY + S - 1) / S);
  constexpr int PLANE = ((S * S * NPP + 15) / 16) * 16;
  constexpr int IMG_H = (OCT ? OCT : 2 * NSUB) * PLANE * 8;
  constexpr int NSTEP = KS == 3 ? (OCT ? (9 * OCT + 3) / 4 : 5 * NSUB) : NSUB / 2;
  constexpr int TABC = OCT ? ((OCT * 8 + 15) / 16) * 16 : NSUB * 16;
  constexpr size_t wch = (size_t)NSTEP * NT * 2 * 64 * 8 * 2;
  constexpr size_t rest = (size_t)IMG_H * 4 + (size_t)8 * NT * 16 * 2 * 4 + (size_t)2 * TABC * 4;   // image, statistics, prologue table
  static_assert(rest + wch <= 160 * 1024, "LDS budget");
  Conv2dSArgs ka = ka_in;
  ka.w_all = (ka.nchunks > 1 && rest + ka.nchunks * wch <= 160 * 1024) ? 1 : 0;
  const size_t lds = rest + (ka.w_all ? ka.nchunks : 1) * wch;
  static size_t attr_lds = 0;
  auto kern = conv2d_s16_kernel<KS, S, NT, NSUB, TY, OCT>;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv2d(s16): hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr_lds = lds;
  }
  const vx_conv2d_args& a = ka.a;
  const int total_tiles = ka.tiles_x * ka.tiles_y * a.N;
  const int ygroups = (a.Cout + 16 * NT - 1) / (16 * NT);
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 2) per_cu = 2;
  int gx = (256 * per_cu + ygroups - 1) / ygroups;
  if (gx > total_tiles) gx = total_tiles;
  // every template argument, defaulted ones included: the list rocprofv3 prints for the instance
  static const char* kname = vx_kname("conv2d_s16_kernel<%d,%d,%d,%d,%d,%d>", KS, S, NT, NSUB, TY, OCT);
  vx_note_kernel(kname);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ygroups), dim3(512), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv2d(s16)");
  return VX_OK;
}

template <int KS, int S, int NSUB, int TY, int OCT = 0>
static int dispatch_c2s(const Conv2dSArgs& ka, int NT, hipStream_t s) {
  if constexpr (KS == 1) {
    if (NT == 5) return launch_c2s<KS, S, 5, NSUB, TY, OCT>(ka, s);
  }
  if (NT == 3) return launch_c2s<KS, S, 3, NSUB, TY, OCT>(ka, s);
  if (NT == 2) return launch_c2s<KS, S, 2, NSUB, TY, OCT>(ka, s);
  return launch_c2s<KS, S, 1, NSUB, TY, OCT>(ka, s);
}

// arguments validated by vx_conv2d (conv2d_mfma.hip)
int vx_conv2d_s16(const vx_conv2d_args& a, hipStream_t s) {
  Conv2dSArgs ka;
  ka.a = a;
  ka.OH = (a.H + 2 * (a.KS / 2) - a.KS) / a.S + 1;
  ka.OW = (a.W + 2 * (a.KS / 2) - a.KS) / a.S + 1;
  const C2SCfg c = c2s_config(a.KS, a.S, a.Cout);
  ka.tiles_x = (ka.OW + 15) / 16;
  ka.tiles_y = (ka.OH + c.TY - 1) / c.TY;
  ka.nchunks = (a.Cin / 16 + c.NSUB - 1) / c.NSUB;
  ka.mx = (unsigned)((1ull << 32) / (unsigned)ka.tiles_x) + 1u;
  ka.my = (unsigned)((1ull << 32) / (unsigned)ka.tiles_y) + 1u;
  // 3x3 layers of 2 / 3 input sub-blocks (HRNet: 18 / 36 / 48 channels): ONE item per tile with all sub-blocks staged
  // together -- a third / half of the barriers, prefetches and commits, and the weights resident.  The packed layout does
  // not change ([chunk][step] with five steps per sub-block is the same sequence either way).  Stride 2 with three
  // sub-blocks does not fit (120 KB of parity planes + 92 KB of weights).
  const int oct = a.w_family / 100;     // octet-granular K: the weights were packed for it (vx_conv2d checked the family)
  if (oct) {
    ka.nchunks = 1;
    if (oct == 1) return a.S == 1 ? dispatch_c2s<3, 1, 1, 16, 1>(ka, c.NT, s) : dispatch_c2s<3, 2, 1, 16, 1>(ka, c.NT, s);
    return a.S == 1 ? dispatch_c2s<3, 1, 1, 16, 3>(ka, c.NT, s) : dispatch_c2s<3, 2, 1, 16, 3>(ka, c.NT, s);
  }
  const int nsub = a.Cin / 16;
  if (a.KS == 3 && a.S == 1 && !vx_cfg().c2s_no_wide && (nsub == 2 || nsub == 3)) {
    ka.nchunks = 1;
    return nsub == 2 ? dispatch_c2s<3, 1, 2, 16>(ka, c.NT, s) : dispatch_c2s<3, 1, 3, 16>(ka, c.NT, s);
  }
  if (a.KS == 3 && a.S == 1) return dispatch_c2s<3, 1, 1, 16>(ka, c.NT, s);
  if (a.KS == 3 && a.S == 2) return dispatch_c2s<3, 2, 1, 16>(ka, c.NT, s);
  return dispatch_c2s<1, 1, 4, 16>(ka, c.NT, s);
}
