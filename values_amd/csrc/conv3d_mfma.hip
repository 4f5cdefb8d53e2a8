// K1: 3x3x3 convolution (padding 1), channels-last fp32, as an implicit GEMM on the gfx950
// fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain, same numerics as VALU).
//
//   GEMM view:  D[row][col] = sum_k  A[row][k] * B[k][col]
//   A operand = weights  (lane l holds A[row = l&15][k = l>>4])
//   B operand = inputs   (lane l holds B[k = l>>4][col = l&15])
//   D: lane l holds rows (l>>4)*4 + {0..3} of column l&15 -> 4 consecutive output channels of one voxel,
//      so the epilogue is ONE 16-byte store per lane and a wave writes 1 KiB contiguously.
//
// Two packings of the 16x16 tile:
//   plain (XP = 0): rows = 16 output channels, cols = 16 voxels of an x-row, k = (kz,ky,kx, cin): 27*Cin.
//   x-pair (XP = 1, Cout == 8): a 16-wide tile would be half padding, so two x-adjacent outputs share it:
//      rows = (dx in {0,1}, cout 0..7), cols = 16 voxel PAIRS (x = 2p, 2p+1), k = (kz,ky, ix in 0..3, cin) where
//      input x = 2p + ix - 1 and the weight is W[kx = ix - dx] (zero when ix - dx is outside 0..2).
//      K grows by 4/3 while the columns cover twice the voxels: 1.5x fewer MFMAs (75 % dense instead of 50 %).
//
// Work split: a workgroup of NW waves owns an output tile of TXV x TY x TZ voxels (TXV = 16, or 32 for x-pair)
// for 16*NT rows; each wave owns R column tiles and all NT row tiles: R*NT independent accumulators.
// Cin is consumed in chunks of CB (8 or 16) channels: per chunk the halo'd input tile [HZ][HY][HX][CB] and the
// chunk's weights [taps][NT][64 lanes][CB/4] sit in LDS; a lane's CB/4 consecutive channels come from one
// ds_read_b128 (b64 for CB = 8) and feed CB/4 MFMAs.
//
// Schedule: persistent workgroups walk a flat list of (tile, chunk) work items.  While item i computes, the global
// loads of item i+1 are already in flight into registers (buffer loads: out-of-volume halo pieces are steered
// out of the descriptor's range and come back as zeros -- no branches, ~3 VALU per 16-byte piece); they are
// committed to LDS after the barrier that ends item i.  Single-chunk layers keep their weights resident in LDS.
// Non-MFMA instructions are the enemy here (they compete with the matrix pipe for the SIMD's issue port), so
// everything per-lane that does not change between tiles is computed once per workgroup.
//
// Input layouts: plain [N][D][H][W][pitch], or the decoder's x-blocked concat buffer
// [N][D][H][W/xb][2][xb][C] (up half, skip half as alternating dense blocks; see unet3d_forward.hip).
//
// Epilogue (fused): bias, then either (a) raw store + per-workgroup (sum, sumsq) partials for the
// InstanceNorm that follows (contract blocks), or (b) LeakyReLU/ReLU + dropout (expand / center).
#include "common.h"
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// conv3d_c8.hip: the Cout = 8, Cin in {8, 16} layers on the 4x4x1 matrix instruction
bool vx_conv3d_c8_applies(int Cin, int Cout);
int vx_pack_conv3d_k3_c8(const float* w_torch, float* w_packed, int Cin, hipStream_t s);
int vx_conv3d_k3_c8(const vx_conv3d_args& a, int txv, int ty, int tz, hipStream_t s);
// conv3d_s16.hip: split-fp16 schedule for the Cout >= 16 layers
int64_t vx_conv3d_s16_packed_floats(int Cin, int Cout);
int vx_pack_conv3d_k3_s16(const float* w_torch, float* w_packed, int Cin, int Cout, hipStream_t s);
int vx_conv3d_k3_s16(const vx_conv3d_args& a, hipStream_t s);
bool vx_conv3d_s16_head_fusable(int Cin, int Cout);
bool vx_conv3d_xp8_applies(int D, int H, int W, int Cin, int Cout);
int vx_conv3d_k3_xp8(const vx_conv3d_args& a, int stat_tiles, hipStream_t s);   // 1 = not taken
// conv3d_zc16.hip: the role-split z-column kernel of the Cout = 16 layers
bool vx_conv3d_zc16_packs(int Cin, int Cout);
bool vx_conv3d_zc16_applies(int D, int H, int W, int Cin, int Cout);
int64_t vx_conv3d_zc16_packed_floats(int Cin, int Cout);
int vx_pack_conv3d_zc16(const float* w_torch, float* w_packed, int Cin, int Cout, hipStream_t s);
int vx_conv3d_k3_zc16(const vx_conv3d_args& a, const float* w_block, int stat_tiles, hipStream_t s);   // 1 = not taken
// conv3d_deep.hip: the role-split tile kernel of the deep layers (Cout % 32 == 0, volumes of 32^3 and below)
bool vx_conv3d_deep_packs(int Cin, int Cout);
bool vx_conv3d_deep_applies(int N, int D, int H, int W, int Cin, int Cout);
int64_t vx_conv3d_deep_packed_floats(int Cin, int Cout);
int vx_pack_conv3d_deep(const float* w_torch, float* w_packed, int Cin, int Cout, hipStream_t s);
int vx_conv3d_k3_deep(const vx_conv3d_args& a, const float* w_block, int stat_tiles, hipStream_t s);   // 1 = not taken
void vx_conv3d_s16_tile(int H, int W, int Cout, int* txv, int* ty, int* tz);

struct ConvKArgs {
  vx_conv3d_args a;
  int tiles_x, tiles_y, tiles_z, nchunks;
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

constexpr unsigned VX_OOB = 0xFFFFFFF0u;      // voffset beyond every descriptor's num_records
constexpr unsigned VX_NUMREC = 0x80000000u;

template <int CB, int NT, int TX, int TY, int TZ, int NW, int XP>
__global__ __launch_bounds__(64 * NW) void conv3d_k3_mfma_kernel(ConvKArgs ka) {
  constexpr int NTH = 64 * NW;                     // threads per workgroup
  constexpr int CPL = CB / 4;                      // channels per lane per tap
  constexpr int TXV = XP ? 2 * TX : TX;            // voxels per tile along x (TX = columns per column tile = 16)
  constexpr int NVT = TX * TY * TZ / 16;           // column tiles per workgroup
  constexpr int R = NVT / NW;                      // column tiles per wave
  constexpr int HX = TXV + 2, HY = TY + 2, HZ = TZ + 2;
  constexpr int NHALO = HX * HY * HZ;
  // LDS image of the input tile: [g = channel group of CPL][parity plane (x-pair only)][position][CPL floats].
  // A lane (column m, k-group g) reads CPL floats at position p0 + m of plane g: the 16 columns of a tile are
  // CONSECUTIVE positions, so a wave's read is conflict-free whatever the tap offset (plane size: see PLANE),
  // and every tap is an immediate offset.  x-pair columns are 2 voxels apart, so even and odd x live in
  // separate parity planes and stay consecutive.
  constexpr int NPAR = XP ? 2 : 1;
  constexpr int HXP = XP ? HX / 2 : HX;            // positions per x-row (per parity)
  constexpr int NPP = HXP * HY * HZ;               // positions per parity plane
  // positions per g-plane, padded so that plane stride keeps the 4 k-groups on disjoint banks:
  // 16-byte reads (CB=16): multiple of 16 positions; 8-byte reads (CB=8): 16 mod 32 positions
  constexpr int PLANE = CB == 16 ? ((NPAR * NPP + 15) / 16) * 16 : ((NPAR * NPP + 15) / 32) * 32 + 16;
  constexpr int IN_FLOATS = 4 * PLANE * (CB / 4);
  constexpr int NTAP = XP ? 36 : 27;
  constexpr int W_FLOATS = NTAP * NT * 64 * CPL;
  constexpr int Q = CB / 4;                        // 16-byte pieces per voxel
  constexpr int IN_IT = (NHALO * Q + NTH - 1) / NTH;   // staging iterations per thread (input tile)
  constexpr int W_IT = (W_FLOATS / 4 + NTH - 1) / NTH; // staging iterations per thread (weights)
  static_assert(NVT % NW == 0, "tile must give each wave a whole number of column tiles");
  static_assert(IN_IT <= 16, "invalid-piece masks are 16 bits per class");
  static_assert(!XP || NT == 1, "x-pair packing is for Cout == 8");
  typedef float vecc __attribute__((ext_vector_type(CPL)));

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + IN_FLOATS;

  const vx_conv3d_args& a = ka.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int m = lane & 15;   // column within column tile (B) / row within row tile (A)
  const int g = lane >> 4;   // k index within the MFMA's K=4; D rows 4g..4g+3
  const int cg = blockIdx.y;  // row group of 16*NT (plain) -- always 0 for x-pair
  const int ntiles = ka.tiles_x * ka.tiles_y * ka.tiles_z;
  const int total = ntiles * a.N;
  const int lastx = (ka.tiles_x - 1) * TXV, lasty = (ka.tiles_y - 1) * TY, lastz = (ka.tiles_z - 1) * TZ;

  // ---- per-lane constants of the compute phase and of the epilogue (identical for every tile) ----
  int vbase[R];      // LDS float index of the lane's B fragment at tap (0,0,0)
  unsigned ovoff[R]; // byte offset of the lane's output piece relative to the tile origin (rows of row tile 0)
  unsigned eoff[R];  // element index (dropout / mask) of the same, relative to the tile origin
  unsigned obad_xhi = 0, obad_yhi = 0, obad_zhi = 0;  // bit r: piece outside the volume in the last tile of that axis
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int v = (wave * R + r) * 16 + m;                 // column index within the workgroup tile
    const int cx = v % TX, ly = (v / TX) % TY, lz = v / (TX * TY);
    const int lx = XP ? 2 * cx : cx;                       // first voxel of the column
    vbase[r] = (g * PLANE + (lz * HY + ly) * HXP + cx) * CPL;
    const int ox = XP ? lx + (g >> 1) : lx;                // voxel this lane stores
    const int oc = XP ? (g & 1) * 4 : g * 4;               // its first channel (within row tile 0)
    const int ovox = (lz * a.H + ly) * a.W + ox;
    ovoff[r] = (unsigned)((ovox * a.out_pitch + a.out_coff + oc) * 4);
    eoff[r] = (unsigned)(ovox * a.Cout + oc);
    if (ox >= a.W - lastx) obad_xhi |= 1u << r;
    if (ly >= a.H - lasty) obad_yhi |= 1u << r;
    if (lz >= a.D - lastz) obad_zhi |= 1u << r;
  }

  // ---- per-thread staging pattern (identical for every tile) ----
  // input rows: plain: voxel stride in_pitch; x-blocked concat: [W/xb][2][xb][C], C = Cin/2
  const int xb = a.in_xblk;
  const int Csrc = xb ? a.Cin / 2 : a.Cin;
  const int voxf = xb ? 2 * Csrc : a.in_pitch;           // floats per voxel step along x (row average)
  const int rowf = a.W * voxf;                           // floats per x-row
  const int biasf = (a.H + 1) * rowf + 4 * voxf;          // keeps every voffset non-negative
  unsigned voff[IN_IT];
  int ldst[IN_IT];   // LDS float index of the piece's first channel group (a 16-byte piece spans 4/CPL groups)
  unsigned ibad_always = 0, ibad_xlo = 0, ibad_xhi = 0, ibad_ylo = 0, ibad_yhi = 0, ibad_zlo = 0, ibad_zhi = 0;
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int idx = tid + it * NTH;
    const int vox = idx / Q, q = idx % Q;
    const int hx = vox % HX, hy = (vox / HX) % HY, hz = vox / (HX * HY);
    const int dxr = hx - 1, dyr = hy - 1, dzr = hz - 1;   // voxel relative to the tile origin
    int xf;                                               // float offset of (dxr, channel 4q) within its row
    if (xb) {
      const int blk = dxr >= 0 ? dxr / xb : -((-dxr + xb - 1) / xb);
      const int rem = dxr - blk * xb;
      const int sl = (Csrc < CB) ? (4 * q) / Csrc : 0;    // C < CB: the chunk spans both halves
      const int cs = (Csrc < CB) ? (4 * q) % Csrc : 4 * q;
      xf = (blk * 2 + sl) * xb * Csrc + rem * Csrc + cs;
    } else {
      xf = dxr * a.in_pitch + 4 * q;
    }
    const int rel = (dzr * a.H + dyr) * rowf + xf;
    voff[it] = (unsigned)((rel + biasf) * 4);
    {
      const int par = XP ? (hx & 1) : 0, px = XP ? (hx >> 1) : hx;
      const int pos = par * NPP + (hz * HY + hy) * HXP + px;
      ldst[it] = ((q * (4 / CPL)) * PLANE + pos) * CPL;
    }
    if (idx >= NHALO * Q) ibad_always |= 1u << it;
    if (dxr < 0) ibad_xlo |= 1u << it;
    if (dxr >= a.W - lastx) ibad_xhi |= 1u << it;
    if (dyr < 0) ibad_ylo |= 1u << it;
    if (dyr >= a.H - lasty) ibad_yhi |= 1u << it;
    if (dzr < 0) ibad_zlo |= 1u << it;
    if (dzr >= a.D - lastz) ibad_zhi |= 1u << it;
  }
  const size_t in_sample = (size_t)a.D * a.H * rowf;
  const size_t out_sample = (size_t)a.D * a.H * a.W * a.out_pitch;
  const int cper = xb && Csrc >= CB ? Csrc / CB : 0;      // chunks per concat half (0: not chunk-uniform)

  auto decode = [&](int tile_lin, int& n, int& tx, int& ty, int& tz) {
    unsigned t = (unsigned)tile_lin, q;
    q = ka.tiles_x == 1 ? t : __umulhi(t, ka.mx); tx = (int)(t - q * ka.tiles_x); t = q;
    q = ka.tiles_y == 1 ? t : __umulhi(t, ka.my); ty = (int)(t - q * ka.tiles_y); t = q;
    q = ka.tiles_z == 1 ? t : __umulhi(t, ka.mz); tz = (int)(t - q * ka.tiles_z); n = (int)q;
  };

  const float* w_cg = a.w_packed + (size_t)cg * ka.nchunks * W_FLOATS;
  f32x4 ibuf[IN_IT];
  f32x4 wbuf[W_IT];
  const bool w_resident = ka.nchunks == 1;  // single-chunk layers: the weights never change, stage them once
  bool w_fresh = true;

  // issue the global loads of one (tile, chunk) work item into registers (no wait)
  auto prefetch = [&](int tile_lin, int chunk, bool have, bool with_w) {
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
    int coff;  // float offset of this chunk's first channel
    if (!xb) coff = chunk * CB;
    else if (cper) coff = (chunk / cper) * xb * Csrc + (chunk % cper) * CB;
    else coff = 0;
    const unsigned soff = (unsigned)((((tz * TZ) * a.H + ty * TY) * rowf + tx * TXV * voxf + coff) * 4);
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.in + (size_t)(have ? n : 0) * in_sample - biasf), 0, VX_NUMREC, 0x00020000);
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const unsigned vo = ((bad >> it) & 1u) ? VX_OOB : voff[it];
      ibuf[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)vo, (int)soff, 0));
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
      if (idx < NHALO * Q) {
        if (CPL == 4) {
          *reinterpret_cast<f32x4*>(s_in + ldst[it]) = ibuf[it];
        } else {  // CB = 8: the piece's two channel pairs belong to two g-planes
          *reinterpret_cast<f32x2*>(s_in + ldst[it]) = (f32x2){ibuf[it][0], ibuf[it][1]};
          *reinterpret_cast<f32x2*>(s_in + ldst[it] + PLANE * CPL) = (f32x2){ibuf[it][2], ibuf[it][3]};
        }
      }
    }
    if (with_w) {
#pragma unroll
      for (int it = 0; it < W_IT; ++it) {
        const int idx = tid + it * NTH;
        if (idx < W_FLOATS / 4) reinterpret_cast<f32x4*>(s_w)[idx] = wbuf[it];
      }
    }
  };

  // bias once per workgroup: a load inside the loop would sit behind the prefetch loads in the in-order vmcnt
  // queue and make every epilogue wait for the NEXT tile's data
  f32x4 bias4[NT];
  bool cvalid[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = XP ? (g & 1) * 4 : (cg * NT + nt) * 16 + g * 4;
    cvalid[nt] = co < a.Cout;
    bias4[nt] = cvalid[nt] ? *reinterpret_cast<const f32x4*>(a.bias + co) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // One flat loop over (tile, chunk) work items: commit the prefetched registers to LDS, issue the NEXT item's
  // global loads (straight-line, predicated -- no loop-carried register copies that would force a vmcnt(0)),
  // compute this item, and run the epilogue after a tile's last chunk.
  // XCD-aware tile order (see conv3d_c8.hip): each XCD takes a contiguous run of tiles per round
  int tile_lin = blockIdx.x, chunk = 0;
  if ((gridDim.x & 7) == 0) tile_lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
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
      // ---- NTAP taps x CPL MFMAs x R x NT; fragments are read PD taps ahead of their MFMAs ----
      constexpr int PD = CB == 8 ? 3 : 1;  // CB = 8 has only 2*R*NT MFMAs per tap to cover an LDS read
      constexpr int NB = PD + 1;
      vecc wf[NB][NT], xf[NB][R];
      auto load_tap = [&](int t1, int slot) {
        // plain: tap = (kz*3 + ky)*3 + kx ; x-pair: tap = (kz*3 + ky)*4 + ix
        const int kz = XP ? t1 / 12 : t1 / 9, ky = XP ? (t1 / 4) % 3 : (t1 / 3) % 3, kx = XP ? t1 % 4 : t1 % 3;
        // x-pair: tap ix reads voxel 2p + ix -> parity ix & 1, position p + (ix >> 1)
        const int toff = (XP ? (kx & 1) * NPP + (kz * HY + ky) * HXP + (kx >> 1) : (kz * HY + ky) * HXP + kx) * CPL;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          wf[slot][nt] = *reinterpret_cast<const vecc*>(s_w + ((t1 * NT + nt) * 64 + lane) * CPL);
#pragma unroll
        for (int r = 0; r < R; ++r) xf[slot][r] = *reinterpret_cast<const vecc*>(s_in + vbase[r] + toff);
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
      unsigned obad = 0;
      if (tx == ka.tiles_x - 1) obad |= obad_xhi;
      if (ty == ka.tiles_y - 1) obad |= obad_yhi;
      if (tz == ka.tiles_z - 1) obad |= obad_zhi;
      const unsigned vox0 = (unsigned)(((tz * TZ) * a.H + ty * TY) * a.W + tx * TXV);  // tile origin voxel (in sample)
      const unsigned osoff = vox0 * (unsigned)a.out_pitch * 4u;
      const unsigned e0 = vox0 * (unsigned)a.Cout;
      const __amdgpu_buffer_rsrc_t osrd =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * out_sample), 0, VX_NUMREC, 0x00020000);
      const vx_dkey dkey = vx_drop_key(vx_seed_of(a, a.drop_seed), a.drop_layer, (uint32_t)n);

      float ssum[NT][4], ssq[NT][4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[nt][j] = 0.f; ssq[nt][j] = 0.f; }

#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned cshift = XP ? 0u : (unsigned)((cg * NT + nt) * 16);  // first channel of this row tile
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const bool bad = ((obad >> r) & 1u) || !cvalid[nt];
          f32x4 v = acc[r][nt] + bias4[nt];
          if (a.stats_partial && !bad) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { ssum[nt][j] += v[j]; ssq[nt][j] += v[j] * v[j]; }
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
            for (int j = 0; j < 4; ++j) v[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);  // keep ? 2 : 0
          } else if (a.drop_mode == VX_DROP_MASK) {
            uint32_t mk = 0;
            if (!bad) mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + (size_t)n * a.D * a.H * a.W * a.Cout + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * v[j] : 0.f;
          }
          const unsigned vo = bad ? VX_OOB : ovoff[r] + cshift * 4u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), osrd, (int)vo, (int)osoff, 0);
          // gfx950: a 16-byte buffer store with an SGPR soffset still reads its data registers for a few cycles
          // after issue; hipcc (ROCm 7.2) pads this hazard only for an immediate soffset, so the next VALU write
          // of those registers corrupted the last beat (dword 3, lanes 12-15 of each row).  Pad it by hand.
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 3" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      if (a.stats_partial) {
        // reduce over the 16 lanes that share g -> per-wave sums per row, then over waves via LDS
        float* s_red = smem + IN_FLOATS + W_FLOATS;  // [NW waves][NT][16 rows][2], outside the staged tiles
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
          // x-pair: rows c and c+8 are the two x-parities of channel c
          const int co = XP ? c : (cg * NT + nt) * 16 + c;
          if (co < a.Cout && (!XP || c < 8)) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
              s += s_red[((w * NT + nt) * 16 + c) * 2 + 0];
              q += s_red[((w * NT + nt) * 16 + c) * 2 + 1];
              if (XP) {
                s += s_red[((w * NT + nt) * 16 + c + 8) * 2 + 0];
                q += s_red[((w * NT + nt) * 16 + c + 8) * 2 + 1];
              }
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
// weight packing: torch (Cout, Cin, 3,3,3) -> [rowgroup][chunk][tap][nt][lane 64][CPL]
//   plain : row = cout, tap = (kz,ky,kx)                       (NT fixed per (Cin, Cout) by conv_config())
//   x-pair: row = dx*8 + cout, tap = (kz,ky,ix), value W[kx = ix - dx] or 0      (Cout == 8)
struct ConvCfg { int CB, NT, XP, C8, S16; };
static inline ConvCfg conv_config(int Cin, int Cout) {
  ConvCfg c;
  // Default: the split-fp16 schedule (conv3d_s16.hip) for every layer -- faster AND closer to float64 than the
  // native fp32 matrix instruction.  VX_CONV_FP32=1 selects the native-fp32 kernels of this
  // file / conv3d_c8.hip (an exact fmaf chain; the A/B baseline), VX_CONV_FP32=2 keeps fp32 only for Cout = 8.
  // (vx_config: read once; a launch checks its weights' w_family against what this returns NOW)
  const int fp32 = vx_cfg().conv_fp32;
  c.S16 = (fp32 == 0 || (fp32 == 2 && Cout != 8)) ? 1 : 0;
  c.C8 = (!c.S16 && vx_conv3d_c8_applies(Cin, Cout)) ? 1 : 0;
  c.NT = (Cout % 32 == 0) ? 2 : 1;
  c.XP = (Cout == 8) ? 1 : 0;
  // x-pair layers always go in chunks of 8 channels (same speed as one chunk of 16 here, and the packing the
  // opt-in LDS-DMA schedule in conv3d_dma.hip needs: all weights + two input images in LDS)
  c.CB = c.XP ? 8 : ((Cin % 16 == 0) ? 16 : 8);
  return c;
}
static inline int conv_rows_padded(int Cout, int NT) { return ((Cout + 16 * NT - 1) / (16 * NT)) * (16 * NT); }

__global__ void pack_conv3d_k3_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int CB,
                                      int NT, int XP, int64_t total) {
  const int CPL = CB / 4;
  const int nchunks = Cin / CB;
  const int ntap = XP ? 36 : 27;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int j = r % CPL; r /= CPL;
    const int lane = r % 64; r /= 64;
    const int nt = r % NT; r /= NT;
    const int tap = r % ntap; r /= ntap;
    const int chunk = r % nchunks; r /= nchunks;
    const int rgrp = (int)r;
    const int row = (rgrp * NT + nt) * 16 + (lane & 15);
    const int ci = chunk * CB + (lane >> 4) * CPL + j;
    float v = 0.f;
    if (XP) {
      const int dx = row >> 3, co = row & 7;
      const int kzy = tap / 4, ix = tap % 4, kx = ix - dx;
      if (kx >= 0 && kx <= 2) v = w[((size_t)co * Cin + ci) * 27 + kzy * 3 + kx];
    } else if (row < Cout) {
      v = w[((size_t)row * Cin + ci) * 27 + tap];
    }
    out[i] = v;
  }
}

// Kernel family = packed layout of a layer under the current configuration (values_amd.h, vx_config):
//   1 split-fp16 fragments, 2 split-fp16 x-pair blocks, 3 fp32 fragments, 4 fp32 x-pair, 5 fp32 4x4x1 (Cout = 8)
extern "C" int vx_conv3d_k3_family(int Cin, int Cout) {
  if (Cin % 8 != 0 || Cout % 8 != 0 || Cin <= 0 || Cout <= 0) return 0;
  const ConvCfg c = conv_config(Cin, Cout);
  // 6: the tile kernel's fragments FOLLOWED BY the z-column kernel's (conv3d_zc16.hip: which of the two runs depends on the
  // volume's shape, known only at launch)
  // 7: the tile kernel's fragments FOLLOWED BY the deep-layer kernel's (conv3d_deep.hip; Cout % 32 == 0, Cin >= 16)
  if (c.S16 && !vx_conv3d_s16_head_fusable(Cin, Cout) && vx_conv3d_deep_packs(Cin, Cout)) return 7;
  if (c.S16) return vx_conv3d_s16_head_fusable(Cin, Cout) ? 2 : (vx_conv3d_zc16_packs(Cin, Cout) ? 6 : 1);   // head-fusable == x-pair packing
  if (c.C8) return 5;
  return c.XP ? 4 : 3;
}

extern "C" int64_t vx_conv3d_k3_packed_floats(int Cin, int Cout) {
  if (Cin % 8 != 0 || Cout % 8 != 0 || Cin <= 0 || Cout <= 0) return -1;
  ConvCfg c = conv_config(Cin, Cout);
  if (c.C8) return (int64_t)27 * Cin * 8;
  if (c.S16) return vx_conv3d_s16_packed_floats(Cin, Cout) + vx_conv3d_zc16_packed_floats(Cin, Cout) + vx_conv3d_deep_packed_floats(Cin, Cout);
  if (c.XP) return (int64_t)16 * Cin * 36;
  return (int64_t)conv_rows_padded(Cout, c.NT) * Cin * 27;
}

extern "C" int vx_pack_conv3d_k3(const float* w_torch, float* w_packed, int Cin, int Cout, vx_stream_t stream) {
  if (!w_torch || !w_packed) VX_FAIL(VX_E_NULL, "vx_pack_conv3d_k3: null pointer");
  int64_t total = vx_conv3d_k3_packed_floats(Cin, Cout);
  if (total < 0) VX_FAIL(VX_E_SHAPE, "vx_pack_conv3d_k3: Cin=%d Cout=%d must be positive multiples of 8", Cin, Cout);
  ConvCfg c = conv_config(Cin, Cout);
  if (c.C8) return vx_pack_conv3d_k3_c8(w_torch, w_packed, Cin, (hipStream_t)stream);
  if (c.S16) {
    const int rc = vx_pack_conv3d_k3_s16(w_torch, w_packed, Cin, Cout, (hipStream_t)stream);
    if (rc != VX_OK) return rc;
    if (vx_conv3d_deep_packs(Cin, Cout))      // (never both: the z-column kernel packs Cout = 16)
      return vx_pack_conv3d_deep(w_torch, w_packed + vx_conv3d_s16_packed_floats(Cin, Cout), Cin, Cout, (hipStream_t)stream);
    if (!vx_conv3d_zc16_packs(Cin, Cout)) return rc;
    return vx_pack_conv3d_zc16(w_torch, w_packed + vx_conv3d_s16_packed_floats(Cin, Cout), Cin, Cout, (hipStream_t)stream);
  }
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_conv3d_k3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_torch, w_packed, Cin,
                     Cout, c.CB, c.NT, c.XP, total);
  VX_CHECK_LAUNCH("vx_pack_conv3d_k3");
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------
// tile shapes: TX = columns per column tile row (x extent in voxels is 2*TX for x-pair)
struct TileCfg { int TX, TY, TZ, TXV; };
static inline TileCfg tile_config(int W, int XP) {
  if (XP) {
    if (W >= 32) return {16, 4, 4, 32};
    if (W >= 16) return {8, 8, 4, 16};
    return {4, 4, 4, 8};
  }
  if (W >= 16) return {16, 4, 4, 16};
  if (W >= 8) return {8, 8, 4, 8};
  return {4, 4, 4, 4};
}

static int conv_tiles(int D, int H, int W, int Cout) {
  if (conv_config(8, Cout).S16) {   // split-fp16 schedule: its own tile choice (larger tiles for large layers)
    int txv, ty, tz;
    vx_conv3d_s16_tile(H, W, Cout, &txv, &ty, &tz);
    return ((W + txv - 1) / txv) * ((H + ty - 1) / ty) * ((D + tz - 1) / tz);
  }
  TileCfg t = tile_config(W, Cout == 8);
  return ((W + t.TXV - 1) / t.TXV) * ((H + t.TY - 1) / t.TY) * ((D + t.TZ - 1) / t.TZ);
}
extern "C" int vx_conv3d_k3_tiles(int D, int H, int W) {
  // upper bound over every packing, Cout and mode (stats_partial sizing).  Neither tiling dominates: for 8 <= W < 16
  // the x-pair tile (4 pairs x 4 x 4) makes twice the tiles of the plain 8 x 8 x 4 one, for W >= 16 it is the reverse
  int best = 0;
  for (int xp = 0; xp < 2; ++xp) {
    const TileCfg t = tile_config(W, xp);
    const int n = ((W + t.TXV - 1) / t.TXV) * ((H + t.TY - 1) / t.TY) * ((D + t.TZ - 1) / t.TZ);
    if (n > best) best = n;
  }
  const int couts[3] = {8, 16, 32};   // split-fp16: x-pair / one row tile (large tile) / two row tiles
  for (int i = 0; i < 3; ++i) {
    int txv, ty, tz;
    vx_conv3d_s16_tile(H, W, couts[i], &txv, &ty, &tz);
    const int n = ((W + txv - 1) / txv) * ((H + ty - 1) / ty) * ((D + tz - 1) / tz);
    if (n > best) best = n;
  }
  return best;
}
extern "C" int vx_conv3d_k3_tiles_for(int D, int H, int W, int Cout) { return conv_tiles(D, H, W, Cout); }

template <int CB, int NT, int TX, int TY, int TZ, int NW, int XP>
static int launch_conv(const ConvKArgs& ka, hipStream_t s) {
  constexpr int TXV = XP ? 2 * TX : TX;
  constexpr int NPOS = (TXV + 2) * (TY + 2) * (TZ + 2);  // = NPAR * NPP (HX is even for x-pair)
  constexpr int PLANE = CB == 16 ? ((NPOS + 15) / 16) * 16 : ((NPOS + 15) / 32) * 32 + 16;
  constexpr int IN_FLOATS = 4 * PLANE * (CB / 4);
  constexpr int W_FLOATS = (XP ? 36 : 27) * NT * 64 * (CB / 4);
  constexpr size_t lds = (size_t)(IN_FLOATS + W_FLOATS + NW * NT * 16 * 2) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  auto kern = conv3d_k3_mfma_kernel<CB, NT, TX, TY, TZ, NW, XP>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv3d_k3: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr_set = true;
  }
  const vx_conv3d_args& a = ka.a;
  // persistent workgroups: each loops over tiles with stride gridDim.x (prefetching the next tile's loads
  // while it computes); enough workgroups to fill every CU at the LDS-limited occupancy
  const int total_tiles = ka.tiles_x * ka.tiles_y * ka.tiles_z * a.N;
  const int ygroups = XP ? 1 : (a.Cout + 16 * NT - 1) / (16 * NT);
  int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
  if (per_cu * NW > 32) per_cu = 32 / NW;
  int gx = (256 * per_cu + ygroups - 1) / ygroups;
  if (gx > total_tiles) gx = total_tiles;
  dim3 grid((unsigned)gx, (unsigned)ygroups);
  static const char* kname = vx_kname("conv3d_k3_mfma_kernel<%d,%d,%d,%d,%d,%d,%d>", CB, NT, TX, TY, TZ, NW, XP);
  vx_note_kernel(kname);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv3d_k3");
  return VX_OK;
}

template <int CB, int NT>
static int dispatch_tile(const ConvKArgs& ka, const TileCfg& t, hipStream_t s) {
  if (t.TX == 16) return launch_conv<CB, NT, 16, 4, 4, 8, 0>(ka, s);
  if (t.TX == 8) return launch_conv<CB, NT, 8, 8, 4, 8, 0>(ka, s);
  return launch_conv<CB, NT, 4, 4, 4, 4, 0>(ka, s);
}
template <int CB>
static int dispatch_tile_xp(const ConvKArgs& ka, const TileCfg& t, hipStream_t s) {
  if (t.TX == 16) return launch_conv<CB, 1, 16, 4, 4, 8, 1>(ka, s);
  if (t.TX == 8) return launch_conv<CB, 1, 8, 8, 4, 8, 1>(ka, s);
  return launch_conv<CB, 1, 4, 4, 4, 4, 1>(ka, s);
}

bool vx_conv3d_s16_prologue_ok(int Cin, int Cout);
extern "C" int vx_conv3d_k3_prologue_ok(int D, int H, int W, int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8) return 0;
  if (vx_conv3d_xp8_applies(D, H, W, Cin, Cout)) return 1;
  // the split-fp16 tile kernel's plain layers (dense input, see vx_conv3d_k3)
  return conv_config(Cin, Cout).S16 && vx_conv3d_s16_prologue_ok(Cin, Cout) ? 1 : 0;
}

extern "C" int vx_conv3d_k3_acc_ok(int D, int H, int W, int Cin, int Cout) {
  return Cin == 16 && Cout == 16 && conv_config(Cin, Cout).S16 && vx_conv3d_zc16_applies(D, H, W, Cin, Cout) ? 1 : 0;
}

// in_planar / out_planar: the 16 -> 16 layers of the z-column kernel
extern "C" int vx_conv3d_k3_planar_ok(int D, int H, int W, int Cin, int Cout) {
  return Cin == 16 && Cout == 16 && conv_config(Cin, Cout).S16 && vx_conv3d_zc16_applies(D, H, W, Cin, Cout) ? 1 : 0;
}

// in_mean on the SKIP half of an x-blocked concat input (the decoder's first conv reading a contract block's raw output)
extern "C" int vx_conv3d_k3_skip_prologue_ok(int D, int H, int W, int Cin, int Cout, int xblk) {
  if (Cin <= 0 || Cout <= 0 || Cin % 16 || Cout % 8 || (xblk != 1 && xblk != 2 && xblk != 4)) return 0;
  if (Cin == 16 && vx_conv3d_xp8_applies(D, H, W, Cin, Cout)) return 1;      // (16 -> 8 at full resolution: the z-column kernel's prologue)
  if (Cin % 32) return 0;
  const ConvCfg c = conv_config(Cin, Cout);
  const int csrc = Cin / 2;
  return c.S16 && c.CB == 16 && vx_conv3d_s16_prologue_ok(Cin, Cout) && ((xblk * csrc) & (xblk * csrc - 1)) == 0 ? 1 : 0;
}

extern "C" int vx_conv3d_k3_upfuse_ok(int D, int H, int W, int Cin, int Cout) {
  if (vx_cfg().s16_no_upfuse) return 0;
  if (Cin == 16 && vx_conv3d_xp8_applies(D, H, W, Cin, Cout)) return 1;
  // 2 (round 5): the 16-channel z-column kernel evaluates ConvTranspose3d(32 -> 16) for ALL 16 of its input channels (`in` is not
  // read; up_w = vx_pack_convT_zc16, up_pitch >= 32) -- the up half of a decoder conv that runs over its halves (acc_in)
  if (Cin == 16 && Cout == 16 && conv_config(Cin, Cout).S16 && vx_conv3d_zc16_applies(D, H, W, Cin, Cout)) return 2;
  return 0;
}

extern "C" int vx_conv3d_k3_pool_layout(int D, int H, int W, int Cin, int Cout) {
  if (vx_cfg().s16_no_poolfuse) return 0;
  if (Cin == 8 && vx_conv3d_xp8_applies(D, H, W, Cin, Cout)) return 1;
  if (Cin == 16 && Cout == 16 && conv_config(Cin, Cout).S16 && vx_conv3d_zc16_applies(D, H, W, Cin, Cout)) return 2;
  return 0;
}
extern "C" int vx_conv3d_k3_poolfuse_ok(int D, int H, int W, int Cin, int Cout) {
  return vx_conv3d_k3_pool_layout(D, H, W, Cin, Cout) != 0 ? 1 : 0;
}

// in_split (vx_prenorm_split's fp16 pairs) is read by the z-column kernel's staging waves only: the tile kernel's prologue
// takes the raw tensor (in_repeat) -- a caller that pre-splits where this says 0 has overwritten its tensor for nothing
extern "C" int vx_conv3d_k3_presplit_ok(int D, int H, int W, int Cin, int Cout) {
  return Cin == 8 && vx_conv3d_xp8_applies(D, H, W, Cin, Cout) ? 1 : 0;
}

// in_pool_flags: the tile kernel's prologue on a dense 8-channel input (contr_2_1 of the F = 8 networks)
// (the code lives only in the instances an 8 -> 16 layer of a contract block runs on: 8-channel chunks, plain rows, ONE row
// tile -- Cout % 32 == 0 takes two row tiles per wave and Cout == 8 the x-pair kernels, neither carries it)
extern "C" int vx_conv3d_k3_poolfin_ok(int Cin, int Cout) {
  if (Cin != 8 || Cout <= 0 || Cout % 8 || Cout % 32 == 0) return 0;
  return conv_config(Cin, Cout).S16 && vx_conv3d_s16_prologue_ok(Cin, Cout) ? 1 : 0;
}

extern "C" int vx_conv3d_k3_head_fusable(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8) return 0;
  const ConvCfg c = conv_config(Cin, Cout);
  if (c.S16) return vx_conv3d_s16_head_fusable(Cin, Cout) ? 1 : 0;   // x-pair split-fp16 kernel: halves in lanes l, l ^ 16
  return c.C8;   // the 4x4x1 kernel keeps all channels of a voxel in one lane
}

extern "C" int vx_conv3d_k3(const vx_conv3d_args* ap, vx_stream_t stream) {
  if (!ap) VX_FAIL(VX_E_NULL, "vx_conv3d_k3: null args");
  const vx_conv3d_args& a = *ap;
  if (!a.in || !a.w_packed || !a.bias || (!a.out && !a.head_out)) VX_FAIL(VX_E_NULL, "vx_conv3d_k3: null tensor pointer");
  if (a.head_out) {
    if (!a.head_w || !a.head_b) VX_FAIL(VX_E_NULL, "vx_conv3d_k3: fused head without weights");
    if (a.head_C < 1 || a.head_C > 8) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: fused head takes 1..8 classes, got %d", a.head_C);
    if (conv_config(a.Cin, a.Cout).S16 && a.head_C > 4)
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: the split-fp16 kernel fuses heads of up to 4 classes, got %d", a.head_C);
    if (!vx_conv3d_k3_head_fusable(a.Cin, a.Cout))
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: no fused head for Cin=%d Cout=%d (see vx_conv3d_k3_head_fusable)", a.Cin, a.Cout);
    if (a.stats_partial) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: fused head on a layer with InstanceNorm statistics");
  }
  if (a.Cin <= 0 || a.Cout <= 0 || a.Cin % 8 || a.Cout % 8)
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: Cin=%d Cout=%d must be positive multiples of 8", a.Cin, a.Cout);
  if (a.N <= 0 || a.D <= 0 || a.H <= 0 || a.W <= 0) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: empty tensor");
  if (a.w_family != vx_conv3d_k3_family(a.Cin, a.Cout))
    VX_FAIL(VX_E_DTYPE, "vx_conv3d_k3: weights packed for kernel family %d, the library is configured for family %d "
            "(vx_conv3d_k3_family(%d, %d)): re-pack them", a.w_family, vx_conv3d_k3_family(a.Cin, a.Cout), a.Cin, a.Cout);
  if (a.out_xblk && ((a.out_xblk != 1 && a.out_xblk != 2 && a.out_xblk != 4) || a.W % a.out_xblk || (a.out_half != 0 && a.out_half != 1)))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: bad concat output (xblk=%d, half=%d, W=%d)", a.out_xblk, a.out_half, a.W);
  if ((a.in_mean == nullptr) != (a.in_rstd == nullptr)) VX_FAIL(VX_E_NULL, "vx_conv3d_k3: in_mean / in_rstd must come together");
  // products: 0 / 3 = the split scheme's three products (default); 1 = the opt-in one-product mode of the full-resolution z-column kernel
  if (a.products != 0 && a.products != 1 && a.products != 3)
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: products = %d (0 / 3: the default split scheme, 1: one fp16 product)", a.products);
  if (a.products == 1 && !(conv_config(a.Cin, a.Cout).S16 && vx_conv3d_xp8_applies(a.D, a.H, a.W, a.Cin, a.Cout) &&
                           a.drop_mode != VX_DROP_MASK && a.in_drop_mode != VX_DROP_MASK))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: products = 1 is taken by the full-resolution z-column kernel only (got %dx%dx%d, %d -> %d)",
            a.D, a.H, a.W, a.Cin, a.Cout);
  if (a.up_in) {
    if (!a.up_w || !a.up_b) VX_FAIL(VX_E_NULL, "vx_conv3d_k3: fused up-convolution without weights");
    if (!vx_conv3d_k3_upfuse_ok(a.D, a.H, a.W, a.Cin, a.Cout) || a.drop_mode == VX_DROP_MASK || a.stats_partial || a.head_out)
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: no fused up-convolution for this layer (see vx_conv3d_k3_upfuse_ok; hash or no "
              "dropout, no statistics, no head): %dx%dx%d, %d -> %d", a.D, a.H, a.W, a.Cin, a.Cout);
    if (a.up_pitch < 16 || a.up_pitch % 4 || !vx_aligned16(a.up_in) || !vx_aligned16(a.up_b))
      VX_FAIL(VX_E_ALIGN, "vx_conv3d_k3: up_in pitch %d (>= 16, multiple of 4 floats) / alignment", a.up_pitch);
    if ((int64_t)(a.D / 2 + 2) * (a.H / 2) * (a.W / 2) * a.up_pitch * 4 >= (1ll << 31))
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: one coarse sample must stay below 2 GiB");
    if (a.up_fused && !vx_aligned16(a.up_fused)) VX_FAIL(VX_E_ALIGN, "vx_conv3d_k3: up_fused alignment");
  } else if (a.up_fused) {
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: up_fused (composed up-convolution weights) without up_in");
  }
  if (a.pool_out) {
    if (!a.pool_flags || !a.stats_partial) VX_FAIL(VX_E_NULL, "vx_conv3d_k3: pool_out goes with pool_flags and stats_partial");
    if (!vx_conv3d_k3_poolfuse_ok(a.D, a.H, a.W, a.Cin, a.Cout) || a.drop_mode == VX_DROP_MASK || a.act != VX_ACT_NONE || a.head_out)
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: no pooled output for this layer (see vx_conv3d_k3_poolfuse_ok; hash or no dropout, "
              "no activation): %dx%dx%d, %d -> %d", a.D, a.H, a.W, a.Cin, a.Cout);
    if (!vx_aligned16(a.pool_out)) VX_FAIL(VX_E_ALIGN, "vx_conv3d_k3: pool_out alignment");
  }
  if (a.in_drop_mode != VX_DROP_NONE && a.in_drop_mode != VX_DROP_HASH) VX_FAIL(VX_E_DTYPE, "vx_conv3d_k3: in_drop_mode %d", a.in_drop_mode);
  if (a.in_pool_flags) {
    if (!a.in_mean) VX_FAIL(VX_E_NULL, "vx_conv3d_k3: in_pool_flags goes with in_mean / in_rstd");
    if (!vx_conv3d_k3_poolfin_ok(a.Cin, a.Cout) || a.in_xblk || a.in_pitch != 8 || a.in_split || a.up_in || a.in_repeat > 1)
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: pool-finish on load takes a dense 8-channel tensor of window maxima (see "
              "vx_conv3d_k3_poolfin_ok): %d -> %d, pitch %d", a.Cin, a.Cout, a.in_pitch);
    // the first conv of a contract block: bias + statistics (an InstanceNorm follows) -- the only epilogue whose instances carry
    // the pool-finish code; with an activation / dropout / head the launch would fall onto an instance that ignores the flag words
    if (a.act != VX_ACT_NONE || a.drop_mode != VX_DROP_NONE || a.head_out || !a.out)
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: pool-finish on load goes with the plain epilogue (no activation, no dropout, no head): "
              "act=%d drop_mode=%d", a.act, a.drop_mode);
  }
  if (a.out && !a.out_xblk && (a.out_pitch < a.out_coff + a.Cout || a.out_pitch % 4 || a.out_coff % 4))
    VX_FAIL(VX_E_ALIGN, "vx_conv3d_k3: output pitch/offset must be multiples of 4 floats and cover the channels");
  if (a.in_xblk) {
    if (a.in_xblk != 1 && a.in_xblk != 2 && a.in_xblk != 4) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: in_xblk must be 0, 1, 2 or 4");
    if (a.W % a.in_xblk) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: W=%d not a multiple of in_xblk=%d", a.W, a.in_xblk);
    if (a.Cin % 16) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: concat input needs Cin %% 16 == 0 (two halves of Cin/2)");
  } else if (a.in_pitch < (a.up_in ? a.Cin / 2 : a.Cin) || a.in_pitch % 4) {
    VX_FAIL(VX_E_ALIGN, "vx_conv3d_k3: input pitch must be a multiple of 4 floats and cover the channels");
  }
  if (!vx_aligned16(a.in) || !vx_aligned16(a.out) || !vx_aligned16(a.w_packed) || !vx_aligned16(a.bias))
    VX_FAIL(VX_E_ALIGN, "vx_conv3d_k3: pointers must be 16-byte aligned");
  if (a.act < 0 || a.act > 2 || a.drop_mode < 0 || a.drop_mode > 2) VX_FAIL(VX_E_DTYPE, "vx_conv3d_k3: bad act/drop enum");
  if (a.drop_mode == VX_DROP_MASK && !a.drop_mask) VX_FAIL(VX_E_NULL, "vx_conv3d_k3: VX_DROP_MASK without mask");
  const int64_t inrow = a.in_xblk ? (int64_t)a.Cin : (int64_t)a.in_pitch;
  if ((int64_t)a.D * a.H * a.W * a.out_pitch * 4 >= (1ll << 31) || ((int64_t)a.D + 2) * a.H * a.W * inrow * 4 >= (1ll << 31))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: one sample must stay below 2 GiB (32-bit buffer offsets)");

  ConvCfg c = conv_config(a.Cin, a.Cout);
  TileCfg t = tile_config(a.W, c.XP);
  ConvKArgs ka;
  ka.a = a;
  ka.tiles_x = (a.W + t.TXV - 1) / t.TXV;
  ka.tiles_y = (a.H + t.TY - 1) / t.TY;
  ka.tiles_z = (a.D + t.TZ - 1) / t.TZ;
  ka.nchunks = a.Cin / c.CB;
  ka.mx = (unsigned)((1ull << 32) / (unsigned)ka.tiles_x) + 1u;
  ka.my = (unsigned)((1ull << 32) / (unsigned)ka.tiles_y) + 1u;
  ka.mz = (unsigned)((1ull << 32) / (unsigned)ka.tiles_z) + 1u;
  ka.dbg = nullptr;
#ifdef VX_CONV_STAMPS
  if (const char* e = getenv("VX_CONV_DBG_PTR")) ka.dbg = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  hipStream_t s = (hipStream_t)stream;
  if (c.C8 && !a.in_mean && !a.out_xblk) return vx_conv3d_k3_c8(a, t.TXV, t.TY, t.TZ, s);
  if (c.S16 && vx_conv3d_xp8_applies(a.D, a.H, a.W, a.Cin, a.Cout) && a.drop_mode != VX_DROP_MASK &&
      a.in_drop_mode != VX_DROP_MASK) {
    // the full-resolution layers: z-column walk with a rolling LDS window (conv3d_xp8w.hip)
    const int rc = vx_conv3d_k3_xp8(a, conv_tiles(a.D, a.H, a.W, a.Cout), s);
    if (rc != 1) return rc;
    if (a.products == 1) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: products = 1, but the z-column kernel does not take this launch");
  }
  if (c.S16 && vx_conv3d_zc16_applies(a.D, a.H, a.W, a.Cin, a.Cout)) {
    // the Cout = 16 layers below full resolution: role-split z-column kernel (conv3d_zc16.hip); its weights follow the tile
    // kernel's in the packed block (family 6)
    const int rc = vx_conv3d_k3_zc16(a, a.w_packed + vx_conv3d_s16_packed_floats(a.Cin, a.Cout), conv_tiles(a.D, a.H, a.W, a.Cout), s);
    if (rc != 1) return rc;
  }
  if (a.in_planar || a.out_planar)
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: the planar pre-split hand-over (in_planar / out_planar) is taken where vx_conv3d_k3_planar_ok "
            "(got %dx%dx%d, %d -> %d)", a.D, a.H, a.W, a.Cin, a.Cout);
  if (c.S16 && vx_conv3d_deep_applies(a.N, a.D, a.H, a.W, a.Cin, a.Cout)) {
    // the deep layers (Cout % 32 == 0 on small volumes): role-split tile kernel (conv3d_deep.hip); its weights follow the tile
    // kernel's in the packed block (family 7)
    const int rc = vx_conv3d_k3_deep(a, a.w_packed + vx_conv3d_s16_packed_floats(a.Cin, a.Cout), conv_tiles(a.D, a.H, a.W, a.Cout), s);
    if (rc != 1) return rc;
  }
  // the tile kernel's prologue: a dense input, or (round 5) the skip half of an x-blocked concat input whose halves are whole
  // 16-channel chunks with xb * Cin / 2 a power of two (the element index of the dropout bits follows from the load offset)
  const int csrc = a.Cin / 2;
  const bool xblk_pre = a.in_xblk && c.CB == 16 && csrc % 16 == 0 && ((a.in_xblk * csrc) & (a.in_xblk * csrc - 1)) == 0 &&
                        !a.in_pool_flags && !(a.in_repeat > 1);
  const bool tile_pre = a.in_mean && c.S16 && vx_conv3d_s16_prologue_ok(a.Cin, a.Cout) &&
                        ((!a.in_xblk && a.in_pitch == a.Cin) || xblk_pre) && a.in_drop_mode != VX_DROP_MASK;
  if ((a.in_mean && !tile_pre) || a.out_xblk || a.up_in || a.pool_out || a.in_split || a.in_f16 || a.out_f16)
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: the input prologue / concat output / fused up-convolution are only available where "
            "vx_conv3d_k3_prologue_ok(D, H, W, Cin, Cout), with hash or no dropout (got %dx%dx%d, %d -> %d)", a.D, a.H, a.W, a.Cin, a.Cout);
  if (a.up_split) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: up_split goes with the fused up-convolution (up_in)");
  if (a.acc_in) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: partial sums (acc_in) are taken where vx_conv3d_k3_acc_ok (got %dx%dx%d, %d -> %d)",
                        a.D, a.H, a.W, a.Cin, a.Cout);
  if (c.S16) return vx_conv3d_k3_s16(a, s);
  if (a.out_split) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3: out_split is an epilogue of the split-fp16 tile kernel");
  if (c.XP) return dispatch_tile_xp<8>(ka, t, s);
  if (c.CB == 16 && c.NT == 1) return dispatch_tile<16, 1>(ka, t, s);
  if (c.CB == 16 && c.NT == 2) return dispatch_tile<16, 2>(ka, t, s);
  if (c.CB == 8 && c.NT == 1) return dispatch_tile<8, 1>(ka, t, s);
  return dispatch_tile<8, 2>(ka, t, s);
}
