// K1, second schedule: the same implicit-GEMM 3x3x3 convolution as conv3d_mfma.hip (same packings, same epilogue),
// but the input tile reaches LDS by LDS-DMA (`buffer_load_dwordx4 ... offen lds`) into a DOUBLE-BUFFERED image and
// ALL of the layer's weights stay resident in LDS:
//
//   item i :  issue the DMA of item i+1 into buffer (i+1)&1        (no VGPRs, no ds_write, no commit phase)
//             MFMAs of item i from buffer i&1                      (fragments read one/three taps ahead)
//             epilogue after a tile's last chunk
//             s_waitcnt vmcnt(0) ; ONE s_barrier
//
// Out-of-volume halo pieces are steered out of the buffer descriptor's range: measured on gfx950, an out-of-range
// LDS-DMA lane WRITES ZEROS to its LDS slot (tools/micro/lds_dma_oob.hip), which is exactly the zero padding.
// The DMA is issued through inline asm so that hipcc does not serialise it against the LDS reads of the other
// buffer (it would wait vmcnt(0) before the first ds_read after a builtin LDS-DMA); the single wait is placed by
// hand before the barrier.
//
// Covers the layers where weights-for-all-chunks + 2 input images fit in 160 KiB: the x-pair (Cout = 8) layers as
// CB = 8 chunks and the plain NT = 1 layers (Cout = 16) -- 64 % of the network's FLOPs at 64^3.
//
// STATUS: correct (same tests as the register-staged kernel) but not the default -- see vx_conv3d_k3_try_dma.  Kept
// as the measured alternative; enable with VX_CONV_DMA=1.
#include "common.h"
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ConvDArgs {
  vx_conv3d_args a;
  int tiles_x, tiles_y, tiles_z, nchunks;
  unsigned mx, my, mz;
  int dbg;  // tuning experiments only: 1 = no DMA in the loop, 2 = also no epilogue, 3 = also no barrier
};

namespace {
constexpr unsigned D_OOB = 0xFFFFFFF0u;
constexpr unsigned D_NUMREC = 0x80000000u;
}

template <int CB, int TX, int TY, int TZ, int NW, int XP, int MAXCH>
__global__ __launch_bounds__(64 * NW) void conv3d_k3_dma_kernel(ConvDArgs ka) {
  constexpr int NT = 1;
  constexpr int NTH = 64 * NW;
  constexpr int CPL = CB / 4;
  constexpr int NQ = CB / 4;                       // 16-byte pieces (q-planes) per voxel
  constexpr int TXV = XP ? 2 * TX : TX;
  constexpr int NVT = TX * TY * TZ / 16;
  constexpr int R = NVT / NW;
  constexpr int HX = TXV + 2, HY = TY + 2, HZ = TZ + 2;
  constexpr int NPAR = XP ? 2 : 1;
  constexpr int HXP = XP ? HX / 2 : HX;
  constexpr int NPP = HXP * HY * HZ;
  constexpr int PLANE = ((NPAR * NPP + 15) / 16) * 16;          // slots (16 B) per q-plane
  constexpr int NSLOT = ((NQ * PLANE + 63) / 64) * 64;          // slots per image, whole DMA instructions
  constexpr int IN_FLOATS = NSLOT * 4;
  constexpr int NDMA = NSLOT / 64;                              // DMA instructions per image
  constexpr int DMA_IT = (NDMA + NW - 1) / NW;                  // per wave
  constexpr int NTAP = XP ? 36 : 27;
  constexpr int W_FLOATS = NTAP * NT * 64 * CPL;                // per chunk
  constexpr int W_ALL = MAXCH * W_FLOATS;
  static_assert(NVT % NW == 0 && DMA_IT <= 16, "tile config");
  typedef float vecc __attribute__((ext_vector_type(CPL)));

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w = smem + 2 * IN_FLOATS;

  const vx_conv3d_args& a = ka.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int ntiles = ka.tiles_x * ka.tiles_y * ka.tiles_z;
  const int total = ntiles * a.N;
  const int lastx = (ka.tiles_x - 1) * TXV, lasty = (ka.tiles_y - 1) * TY, lastz = (ka.tiles_z - 1) * TZ;

  // ---- per-lane constants of compute and epilogue ----
  int vbase[R];
  unsigned ovoff[R], eoff[R];
  unsigned obad_xhi = 0, obad_yhi = 0, obad_zhi = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int v = (wave * R + r) * 16 + m;
    const int cx = v % TX, ly = (v / TX) % TY, lz = v / (TX * TY);
    const int lx = XP ? 2 * cx : cx;
    const int lpos = (lz * HY + ly) * HXP + cx;
    vbase[r] = CB == 16 ? (g * PLANE + lpos) * 4 : ((g >> 1) * PLANE + lpos) * 4 + (g & 1) * 2;
    const int ox = XP ? lx + (g >> 1) : lx;
    const int oc = XP ? (g & 1) * 4 : g * 4;
    const int ovox = (lz * a.H + ly) * a.W + ox;
    ovoff[r] = (unsigned)((ovox * a.out_pitch + a.out_coff + oc) * 4);
    eoff[r] = (unsigned)(ovox * a.Cout + oc);
    if (ox >= a.W - lastx) obad_xhi |= 1u << r;
    if (ly >= a.H - lasty) obad_yhi |= 1u << r;
    if (lz >= a.D - lastz) obad_zhi |= 1u << r;
  }

  // ---- per-lane DMA pattern: instruction j = it*NW + wave covers slots [64j, 64j+64) ----
  const int xb = a.in_xblk;
  const int Csrc = xb ? a.Cin / 2 : a.Cin;
  const int voxf = xb ? 2 * Csrc : a.in_pitch;
  const int rowf = a.W * voxf;
  const int biasf = (a.H + 1) * rowf + 4 * voxf;
  unsigned voff[DMA_IT];
  unsigned ibad_always = 0, ibad_xlo = 0, ibad_xhi = 0, ibad_ylo = 0, ibad_yhi = 0, ibad_zlo = 0, ibad_zhi = 0;
#pragma unroll
  for (int it = 0; it < DMA_IT; ++it) {
    const int j = it * NW + wave;
    const int S = j * 64 + lane;
    const int q = S / PLANE, pp = S % PLANE;
    const int par = pp / NPP, pos = pp % NPP;
    const int px = pos % HXP, hy = (pos / HXP) % HY, hz = pos / (HXP * HY);
    const int hx = XP ? 2 * px + par : px;
    const int dxr = hx - 1, dyr = hy - 1, dzr = hz - 1;
    int xf;
    if (xb) {
      const int blk = dxr >= 0 ? dxr / xb : -((-dxr + xb - 1) / xb);
      const int rem = dxr - blk * xb;
      const int sl = (Csrc < CB) ? (4 * q) / Csrc : 0;
      const int cs = (Csrc < CB) ? (4 * q) % Csrc : 4 * q;
      xf = (blk * 2 + sl) * xb * Csrc + rem * Csrc + cs;
    } else {
      xf = dxr * a.in_pitch + 4 * q;
    }
    const int rel = (dzr * a.H + dyr) * rowf + xf;
    voff[it] = (unsigned)((rel + biasf) * 4);
    if (j >= NDMA || q >= NQ || par >= NPAR) ibad_always |= 1u << it;
    if (dxr < 0) ibad_xlo |= 1u << it;
    if (dxr >= a.W - lastx) ibad_xhi |= 1u << it;
    if (dyr < 0) ibad_ylo |= 1u << it;
    if (dyr >= a.H - lasty) ibad_yhi |= 1u << it;
    if (dzr < 0) ibad_zlo |= 1u << it;
    if (dzr >= a.D - lastz) ibad_zhi |= 1u << it;
  }
  const size_t in_sample = (size_t)a.D * a.H * rowf;
  const size_t out_sample = (size_t)a.D * a.H * a.W * a.out_pitch;
  const int cper = xb && Csrc >= CB ? Csrc / CB : 0;

  auto decode = [&](int tile_lin, int& n, int& tx, int& ty, int& tz) {
    unsigned t = (unsigned)tile_lin, q;
    q = ka.tiles_x == 1 ? t : __umulhi(t, ka.mx); tx = (int)(t - q * ka.tiles_x); t = q;
    q = ka.tiles_y == 1 ? t : __umulhi(t, ka.my); ty = (int)(t - q * ka.tiles_y); t = q;
    q = ka.tiles_z == 1 ? t : __umulhi(t, ka.mz); tz = (int)(t - q * ka.tiles_z); n = (int)q;
  };

  // issue the LDS-DMA of one (tile, chunk) item into image `buf`
  auto dma = [&](int tile_lin, int chunk, bool have, int buf) {
    int n, tx, ty, tz;
    decode(tile_lin, n, tx, ty, tz);
    unsigned bad = ibad_always;
    if (tx == 0) bad |= ibad_xlo;
    if (tx == ka.tiles_x - 1) bad |= ibad_xhi;
    if (ty == 0) bad |= ibad_ylo;
    if (ty == ka.tiles_y - 1) bad |= ibad_yhi;
    if (tz == 0) bad |= ibad_zlo;
    if (tz == ka.tiles_z - 1) bad |= ibad_zhi;
    if (!have) bad = 0xFFFFFFFFu;
    int coff;
    if (!xb) coff = chunk * CB;
    else if (cper) coff = (chunk / cper) * xb * Csrc + (chunk % cper) * CB;
    else coff = 0;
    const unsigned soff = (unsigned)((((tz * TZ) * a.H + ty * TY) * rowf + tx * TXV * voxf + coff) * 4);
    const unsigned long long base = (unsigned long long)(a.in + (size_t)(have ? n : 0) * in_sample - biasf);
    i32x4 srd;
    srd[0] = (int)(unsigned)(base & 0xFFFFFFFFull);
    srd[1] = (int)(unsigned)((base >> 32) & 0xFFFFull);   // stride 0
    srd[2] = (int)D_NUMREC;
    srd[3] = 0x00020000;
    const unsigned lds0 = (unsigned)(unsigned long long)smem + (unsigned)(buf * IN_FLOATS * 4);  // LDS byte address
#pragma unroll
    for (int it = 0; it < DMA_IT; ++it) {
      const int j = it * NW + wave;
      if (j < NDMA) {   // wave-uniform
        const unsigned vo = ((bad >> it) & 1u) ? D_OOB : voff[it];
        const unsigned m0v = lds0 + (unsigned)j * 1024u;
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(m0v), "v"(vo), "s"(srd), "s"(soff) : "memory");
      }
    }
  };

  f32x4 bias4;
  {
    const int co = XP ? (g & 1) * 4 : blockIdx.y * 16 + g * 4;
    bias4 = co < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + co) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool cvalid = XP ? true : (blockIdx.y * 16 + g * 4) < a.Cout;

  // ---- prologue: all weights of this row group -> LDS; first item's DMA ----
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.w_packed + (size_t)blockIdx.y * ka.nchunks * W_FLOATS);
    f32x4* dst = reinterpret_cast<f32x4*>(s_w);
    for (int idx = tid; idx < ka.nchunks * (W_FLOATS / 4); idx += NTH) dst[idx] = src[idx];
  }
  int tile_lin = blockIdx.x, chunk = 0, buf = 0;
  bool have = tile_lin < total;
  dma(tile_lin, 0, have, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // make the compiler consume the bias registers HERE: otherwise it places their (first-use) vmcnt(0) inside the
  // loop, in front of every epilogue, where the hardware counter also holds the next item's DMA
  asm volatile("" :: "v"(bias4) : "memory");
  __syncthreads();

  f32x4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};

  while (have) {
    int ntile = tile_lin, nchunk = chunk + 1;
    if (nchunk == ka.nchunks) { nchunk = 0; ntile = tile_lin + (int)gridDim.x; }
    const bool nhave = ntile < total;
    if (ka.dbg < 1) dma(ntile, nchunk, nhave, buf ^ 1);

    {
      const float* si = smem + buf * IN_FLOATS;
      const float* sw = s_w + chunk * W_FLOATS;
      constexpr int PD = CB == 8 ? 3 : 1;
      constexpr int NB = PD + 1;
      vecc wf[NB], xf[NB][R];
      auto load_tap = [&](int t1, int slot) {
        const int kz = XP ? t1 / 12 : t1 / 9, ky = XP ? (t1 / 4) % 3 : (t1 / 3) % 3, kx = XP ? t1 % 4 : t1 % 3;
        const int toff = (XP ? (kx & 1) * NPP + (kz * HY + ky) * HXP + (kx >> 1) : (kz * HY + ky) * HXP + kx) * 4;
        wf[slot] = *reinterpret_cast<const vecc*>(sw + (t1 * 64 + lane) * CPL);
#pragma unroll
        for (int r = 0; r < R; ++r) xf[slot][r] = *reinterpret_cast<const vecc*>(si + vbase[r] + toff);
      };
#pragma unroll
      for (int t = 0; t < PD; ++t) load_tap(t, t);
#pragma unroll
      for (int tap = 0; tap < NTAP; ++tap) {
        if (tap + PD < NTAP) load_tap(tap + PD, (tap + PD) % NB);
        const int cur = tap % NB;
#pragma unroll
        for (int j = 0; j < CPL; ++j)
#pragma unroll
          for (int r = 0; r < R; ++r)
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cur][j], xf[cur][r][j], acc[r], 0, 0, 0);
      }
    }

    if (ka.dbg >= 2) {
#pragma unroll
      for (int r = 0; r < R; ++r) asm volatile("" :: "v"(acc[r]));
    } else if (chunk == ka.nchunks - 1) {
      int n, tx, ty, tz;
      decode(tile_lin, n, tx, ty, tz);
      unsigned obad = 0;
      if (tx == ka.tiles_x - 1) obad |= obad_xhi;
      if (ty == ka.tiles_y - 1) obad |= obad_yhi;
      if (tz == ka.tiles_z - 1) obad |= obad_zhi;
      const unsigned vox0 = (unsigned)(((tz * TZ) * a.H + ty * TY) * a.W + tx * TXV);
      const unsigned osoff = vox0 * (unsigned)a.out_pitch * 4u;
      const unsigned e0 = vox0 * (unsigned)a.Cout;
      const __amdgpu_buffer_rsrc_t osrd =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * out_sample), 0, D_NUMREC, 0x00020000);
      const uint32_t dkey = vx_drop_key(vx_seed_of(a, a.drop_seed), a.drop_layer, (uint32_t)n);
      const unsigned cshift = XP ? 0u : (unsigned)(blockIdx.y * 16);
      float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool bad = ((obad >> r) & 1u) || !cvalid;
        f32x4 v = acc[r] + bias4;
        if (a.stats_partial && !bad) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { ssum[j] += v[j]; ssq[j] += v[j] * v[j]; }
        }
        if (a.act == VX_ACT_LRELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.01f * v[j]);
        } else if (a.act == VX_ACT_RELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        const unsigned e = e0 + eoff[r] + cshift;
        if (a.drop_mode == VX_DROP_HASH) {
          const uint32_t bits = vx_drop_bits4(dkey, e);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);
        } else if (a.drop_mode == VX_DROP_MASK) {
          uint32_t mk = 0;
          if (!bad) mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + (size_t)n * a.D * a.H * a.W * a.Cout + e);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * v[j] : 0.f;
        }
        const unsigned vo = bad ? D_OOB : ovoff[r] + cshift * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), osrd, (int)vo, (int)osoff, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 3" ::: "memory");   // gfx950 store-data hazard with an SGPR soffset (conv3d_mfma.hip)
        __builtin_amdgcn_sched_barrier(0);
        acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (a.stats_partial) {
        float* s_red = s_w + W_ALL;  // [NW][16][2]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s = ssum[j], q = ssq[j];
#pragma unroll
          for (int off = 1; off < 16; off <<= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
          if (m == 0) {
            s_red[(wave * 16 + g * 4 + j) * 2 + 0] = s;
            s_red[(wave * 16 + g * 4 + j) * 2 + 1] = q;
          }
        }
        __syncthreads();
        if (tid < 16) {
          const int c = tid;
          const int co = XP ? c : blockIdx.y * 16 + c;
          if (co < a.Cout && (!XP || c < 8)) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
              s += s_red[(w * 16 + c) * 2 + 0];
              q += s_red[(w * 16 + c) * 2 + 1];
              if (XP) { s += s_red[(w * 16 + c + 8) * 2 + 0]; q += s_red[(w * 16 + c + 8) * 2 + 1]; }
            }
            const int tile = tile_lin - n * ntiles;
            float* dst = a.stats_partial + (((size_t)n * ntiles + tile) * a.Cout + co) * 2;
            dst[0] = s;
            dst[1] = q;
          }
        }
      }
    }
    // the next image has landed (this wave's share) and everybody is done reading the current one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ka.dbg < 3) __syncthreads();
    tile_lin = ntile; chunk = nchunk; have = nhave; buf ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
template <int CB, int TX, int TY, int TZ, int NW, int XP, int MAXCH>
static int launch_dma(const ConvDArgs& ka, hipStream_t s) {
  constexpr int TXV = XP ? 2 * TX : TX;
  constexpr int NPOS = (TXV + 2) * (TY + 2) * (TZ + 2);
  constexpr int PLANE = ((NPOS + 15) / 16) * 16;
  constexpr int NSLOT = (((CB / 4) * PLANE + 63) / 64) * 64;
  constexpr int IN_FLOATS = NSLOT * 4;
  constexpr int W_ALL = MAXCH * (XP ? 36 : 27) * 64 * (CB / 4);
  constexpr size_t lds = (size_t)(2 * IN_FLOATS + W_ALL + NW * 16 * 2) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  auto kern = conv3d_k3_dma_kernel<CB, TX, TY, TZ, NW, XP, MAXCH>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv3d_k3(dma): hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr_set = true;
  }
  const vx_conv3d_args& a = ka.a;
  const int total_tiles = ka.tiles_x * ka.tiles_y * ka.tiles_z * a.N;
  const int ygroups = XP ? 1 : (a.Cout + 15) / 16;
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu < 1) per_cu = 1;
  if (per_cu * NW > 32) per_cu = 32 / NW;
  int gx = (256 * per_cu + ygroups - 1) / ygroups;
  if (gx > total_tiles) gx = total_tiles;
  vx_note_kernel("conv3d_k3_dma_kernel");
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ygroups), dim3(64 * NW), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv3d_k3(dma)");
  return VX_OK;
}

// Returns VX_OK if this schedule handled the launch, 1 if the caller should use the register-staged kernel.
int vx_conv3d_k3_try_dma(const vx_conv3d_args& a, hipStream_t s) {
  // Opt-in: measured at steady clocks (1000 launches) this schedule is 7-10 % SLOWER than the register-staged
  // kernel (104 vs 112 TFLOP/s on 16->16 @32^3, 78 vs 86 on 16->8 @64^3): its DMA reads are 16 B at voxel-pitch
  // stride (the LDS image is channel-group-major), where the register-staged loads are fully contiguous.
  if (vx_cfg().conv_dma != 1) return 1;
  const bool xp = a.Cout == 8;
  ConvDArgs ka;
  ka.a = a;
  ka.dbg = vx_cfg().dma_dbg;
  if (xp && a.W >= 32 && (a.Cin == 8 || a.Cin == 16)) {
    ka.tiles_x = (a.W + 31) / 32; ka.tiles_y = (a.H + 3) / 4; ka.tiles_z = (a.D + 3) / 4;
  } else if (!xp && a.Cout == 16 && a.W >= 16 && (a.Cin == 16 || a.Cin == 32 || a.Cin == 8)) {
    ka.tiles_x = (a.W + 15) / 16; ka.tiles_y = (a.H + 3) / 4; ka.tiles_z = (a.D + 3) / 4;
  } else {
    return 1;
  }
  ka.mx = (unsigned)((1ull << 32) / (unsigned)ka.tiles_x) + 1u;
  ka.my = (unsigned)((1ull << 32) / (unsigned)ka.tiles_y) + 1u;
  ka.mz = (unsigned)((1ull << 32) / (unsigned)ka.tiles_z) + 1u;
  if (xp) {
    // weights were packed for CB = 16 when Cin == 16 (conv_config): this schedule needs the CB = 8 packing
    ka.nchunks = a.Cin / 8;
    const int nw16 = vx_cfg().dma_nw16 ? 1 : 0;
    if (nw16) return launch_dma<8, 16, 4, 4, 16, 1, 2>(ka, s);
    return launch_dma<8, 16, 4, 4, 8, 1, 2>(ka, s);
  }
  if (a.Cin == 8) { ka.nchunks = 1; return launch_dma<8, 16, 4, 4, 8, 0, 1>(ka, s); }
  ka.nchunks = a.Cin / 16;
  const int nw16b = vx_cfg().dma_nw16 ? 1 : 0;
  if (nw16b) return launch_dma<16, 16, 4, 4, 16, 0, 2>(ka, s);
  return launch_dma<16, 16, 4, 4, 8, 0, 2>(ka, s);
}
