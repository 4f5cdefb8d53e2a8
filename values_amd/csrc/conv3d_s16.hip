// K1, split-precision schedule: the 3x3x3 convolution of conv3d_mfma.hip with every fp32 product evaluated on the
// fp16 matrix cores by operand splitting (Ootomo & Yokota's scheme for SGEMM on tensor cores, here on CDNA4):
//
//     x = hi + lo * 2^-11,   hi = fp16(x),  lo = fp16((x - hi) * 2^11)         (the residual is exact in fp32)
//     a * b = hi_a hi_b + 2^-11 (hi_a lo_b + lo_a hi_b) + 2^-22 lo_a lo_b      (last term <= 2^-24 |a b|: dropped)
//
// Three v_mfma_f32_16x16x32_f16 (products exact, fp32 accumulation; a main and a cross accumulator) replace the
// eight v_mfma_f32_16x16x4_f32 of one K = 32 step.  Measured (tools/micro/split_f16.hip, K = 432 dot products of
// conv-like data against float64): max error 5.9e-7 / rms 9.1e-8 for this scheme, 1.1e-6 / 1.6e-7 for the native
// fp32 matrix instruction (which rounds after every K = 4), at 650 vs 155 TFLOP/s of fp32-equivalent work with
// operands in registers.  Activations and weights stay float32 in HBM; the split happens on the way into LDS
// (activations) and at pack time (weights).  Range: |x| < 65504 (fp16).
//
// Everything around the MFMA loop -- work items, register prefetch with out-of-range steering, tile shapes,
// x-blocked concat input, XCD-aware tile order, the fused bias / activation / dropout / statistics epilogue and its
// store layout (the D layout of 16x16x32 is that of 16x16x4) -- is conv3d_mfma.hip's.
//
//   K = 32 step:  CB = 16: two taps x 16 channels,  lane k-group kg = 2 * (tap & 1) + (channel >> 3)
//                 CB =  8: four taps x 8 channels,   kg = tap & 3                         (27 taps padded with zero weights)
//   LDS image:    hi plane and lo plane, each [channel octet][halo position][8 halves]; a lane's B fragment for a
//                 step is one ds_read_b128 per plane at (position of its column + its k-group's tap offset).
//   weights:      [chunk][step][row tile][hi | lo][lane][8 halves], one ds_read_b128 per (step, row tile, plane).
// The loop reads (2 NT + 2 R) KiB of LDS per 3 R NT MFMAs.  Around it (x-pair layers on the large tile): two LDS images,
// one barrier per item, staggered SIMD partners (DB), compile-time epilogues (EPI) -- see the kernel's header below and
// DESIGN.md section 5, round-1g, for the counters behind each step.
#include "common.h"
#include <stdlib.h>

#include "s16_common.h"

struct ConvSArgs {
  vx_conv3d_args a;
  int tiles_x, tiles_y, tiles_z, nchunks;
  unsigned mx, my, mz;
  int ty8;   // 16 x 8 x 4 tiles (vx_conv3d_s16_tile)
  int w_all; // every chunk's weights fit in LDS next to the image: staged once, never re-staged per item
  int dbg;   // DIAGNOSTIC BUILD ONLY (-DVX_CONV_STAMPS, env VX_S16_DBG): 1 no epilogue, 2 also no staging, 3 also no
             // barriers -- phase ablation, the results are wrong by design; the product library compiles S16_DBG to 0
  unsigned long long* stamps;   // VX_CONV_STAMPS diagnostic builds only: per wave, cycles spent per phase
};

// Phase stamps of the item loop (diagnostic build -DVX_CONV_STAMPS, tools/stamp_s16.py): s_memtime deltas summed per
// wave into st_sum[phase]; the values go to a buffer of their own, never into an output.
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
#define VX_STAMP_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define S16_DBG ka.dbg
#else
#define VX_STAMP(i) do {} while (0)
#define VX_STAMP_WAIT_LOADS() do {} while (0)
#define S16_DBG 0
#endif

// XP (Cout == 8, CB == 8): x-pair packing as in conv3d_mfma.hip -- rows = (dx, cout), columns = voxel pairs
// (x = 2p, 2p + 1), a K = 32 step = one (kz, ky) row with the four x-offsets ix = 0..3 as the four k-groups
// (weight W[kx = ix - dx], zero outside 0..2): 9 steps for two voxels per column instead of 2 x 7, all 16 rows and
// all four lane groups of the epilogue useful.  Even / odd x live in two parity planes so that columns stay
// consecutive positions.
// DB (x-pair, single-chunk layers with the 16x8x4 tile): TWO LDS images and ONE barrier per item.  Item k + 1 is
// converted and written into the other image while item k is being multiplied.  DB = 1: every wave stages right after
// the barrier (measured: no gain over two barriers, the waves still move through the phases together).  DB = 2 adds
// a stagger between the two waves that share a SIMD (w and w + 4): waves 0..3 run stage, multiply, epilogue; waves
// 4..7 run the PREVIOUS item's epilogue, multiply, stage -- one wave's stores / conversions under the other's MFMAs
// (MI355X_MICROARCH.md, two waves per SIMD, item 9).  +2.9 % end to end on the 64^3 network; results bit-identical.
// EPI: the epilogue's features fixed at compile time for the large-tile instances (fewer scalar registers to spill --
// 76 -> 8 for the plain one -- and a shorter epilogue: contr_1_2 1.72 -> 1.57 ms).  0 = plain (bias, statistics,
// store: the layers an InstanceNorm follows); 1 = LeakyReLU + hash dropout (decoder layers); 2 = 1 + the fused
// 1x1x1 head; 3 = everything chosen at run time from vx_conv3d_args (all other instances).
template <int CB, int NT, int TX, int TY, int TZ, int NW, int XP, int DB = 0, int EPI = 3>
__global__ __launch_bounds__(64 * NW) void conv3d_k3_s16_kernel(ConvSArgs ka) {
  constexpr int NTH = 64 * NW;
  constexpr bool SINGLE = DB == 1 || DB == 2;   // double-buffered, every item a whole tile (one chunk)
  constexpr bool DEFER = DB == 0 && XP == 0;     // two-barrier schedule, plain layers: next item's loads issued inside the multiply loop
  constexpr bool STAG = DB == 2 || DB == 3;        // staggered waves (DB = 3: also for several chunks per tile)
  constexpr int TXV = XP ? 2 * TX : TX;            // voxels per tile along x
  constexpr int NVT = TX * TY * TZ / 16;
  constexpr int R = NVT / NW;
  constexpr int HX = TXV + 2, HY = TY + 2, HZ = TZ + 2;
  constexpr int NHALO = HX * HY * HZ;
  constexpr int HXP = XP ? HX / 2 : HX;            // positions per x-row (per parity plane)
  constexpr int OCT = CB / 8;                      // channel octets per chunk
  constexpr int TPS = 32 / CB;                     // taps per K = 32 step
  constexpr int NSTEP = XP ? 9 : (27 + TPS - 1) / TPS;
  // positions per plane (octet plane, or parity plane for x-pair): a multiple of 16, so that the two planes a
  // 16-lane read group mixes fall on complementary slots
  constexpr int PLANE = (((XP ? HXP * HY * HZ : NHALO) + 15) / 16) * 16;
  constexpr int IMG_H = (XP ? 2 : OCT) * PLANE * 8;  // halves per precision plane
  static_assert(!XP || (CB == 8 && NT == 1), "x-pair packing is for Cout == 8 in chunks of 8 channels");
  constexpr int IN_BYTES = 2 * IMG_H * 2;
  // halves of one chunk's weights.  x-pair: the two x-rows (dx = 0, 1) of an output channel hold the SAME three taps,
  // shifted by one k-group, and one k-group of each is zero -- so a (step, precision) block is [kx 4][co 8] pieces
  // (kx = 3: zeros) = 512 B instead of 64 lanes x 16 B, and lane (m, g) reads piece (kx = (g - dx) & 3, co) with
  // dx = m >> 3, co = m & 7.  Two lanes per piece (an LDS broadcast); pieces ordered [co >> 2][kx][co & 3] so that
  // each of ds_read_b128's four lane groups {0-3,12-15,20-27}, ... touches 64 distinct banks
  constexpr int WL = XP ? 32 : 64;                 // 16-byte pieces per (step, nt, precision) block
  constexpr int W_H = NSTEP * NT * 2 * WL * 8;
  constexpr int Q = CB / 4;                        // 16-byte fp32 pieces per voxel
  constexpr int IN_IT = (NHALO * Q + NTH - 1) / NTH;
  constexpr int W_IT = (W_H / 8 + NTH - 1) / NTH;  // 16-byte pieces per thread
  static_assert(NVT % NW == 0 && IN_IT <= 16, "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* s_hi = reinterpret_cast<_Float16*>(smem_raw);
  _Float16* s_lo = s_hi + IMG_H;
  constexpr int BUF_H = 2 * IMG_H;                 // halves of one image (hi + lo planes)
  _Float16* s_w = s_hi + (DB ? 2 : 1) * BUF_H;
  float* s_red = reinterpret_cast<float*>(smem_raw + (DB ? 2 : 1) * IN_BYTES + (size_t)(ka.w_all ? ka.nchunks : 1) * W_H * 2);

  const vx_conv3d_args& a = ka.a;
  const bool f_lrelu = EPI == 3 ? a.act == VX_ACT_LRELU : (EPI == 1 || EPI == 2);
  const bool f_relu = EPI == 3 && a.act == VX_ACT_RELU;
  const bool f_dhash = EPI == 3 ? a.drop_mode == VX_DROP_HASH : (EPI == 1 || EPI == 2);
  const bool f_dmask = EPI == 3 && a.drop_mode == VX_DROP_MASK;
  const bool f_head = XP && (EPI == 3 ? a.head_out != nullptr : EPI == 2);
  const bool f_store = (EPI == 0 || EPI == 1) ? true : a.out != nullptr;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int m = lane & 15;
  const int g = lane >> 4;      // k-group of the operands; D rows 4g..4g+3
  // the lane's piece of a weight block; x-pair: [co >> 2][kx][co & 3], conflict-free within ds_read_b128's lane groups
  const int wslot = XP ? ((((m & 7) >> 2) * 4 + ((g - (m >> 3)) & 3)) * 4 + (m & 3)) : lane;
  const int cg = blockIdx.y;
  const int ntiles = ka.tiles_x * ka.tiles_y * ka.tiles_z;
  const int total = ntiles * a.N;
  const int lastx = (ka.tiles_x - 1) * TXV, lasty = (ka.tiles_y - 1) * TY, lastz = (ka.tiles_z - 1) * TZ;

  // ---- per-lane constants of the compute phase and the epilogue ----
  int vbase[R];      // halo position of the lane's column at tap (0,0,0)
  unsigned ovoff[R], eoff[R];
  unsigned obad_xhi = 0, obad_yhi = 0, obad_zhi = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int v = (wave * R + r) * 16 + m;
    const int cx = v % TX, ly = (v / TX) % TY, lz = v / (TX * TY);
    vbase[r] = (lz * HY + ly) * HXP + cx;
    const int lx = XP ? 2 * cx + (g >> 1) : cx;            // voxel this lane stores
    const int oc = XP ? (g & 1) * 4 : g * 4;               // its first channel (within row tile 0)
    const int ovox = (lz * a.H + ly) * a.W + lx;
    ovoff[r] = (unsigned)((ovox * a.out_pitch + a.out_coff + oc) * 4);
    eoff[r] = (unsigned)(ovox * a.Cout + oc);
    if (lx >= a.W - lastx) obad_xhi |= 1u << r;
    if (ly >= a.H - lasty) obad_yhi |= 1u << r;
    if (lz >= a.D - lastz) obad_zhi |= 1u << r;
  }
  // this lane's (tap, octet) of every step, as a position offset into a precision plane (padding taps re-read
  // tap 26: their weights are zero, but the data must be finite)
  int toff[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    if (XP) {   // step = (kz, ky) row, k-group = x-offset ix: voxel 2p + ix -> parity ix & 1, position p + (ix >> 1)
      toff[s] = (g & 1) * PLANE + ((s / 3) * HY + s % 3) * HXP + (g >> 1);
    } else {
      int tap = CB == 16 ? 2 * s + (g >> 1) : 4 * s + g;
      if (tap > 26) tap = 26;
      const int oct = CB == 16 ? (g & 1) : 0;
      toff[s] = oct * PLANE + ((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3;
    }
  }

  // ---- per-thread staging pattern ----
  const int xb = a.in_xblk;
  const int Csrc = xb ? a.Cin / 2 : a.Cin;
  const int voxf = xb ? 2 * Csrc : a.in_pitch;
  const int rowf = a.W * voxf;
  const int biasf = (a.H + 1) * rowf + 4 * voxf;
  unsigned voff[IN_IT];
  int ldst[IN_IT];   // halves index of the piece within a precision plane
  unsigned ibad_always = 0, ibad_xlo = 0, ibad_xhi = 0, ibad_ylo = 0, ibad_yhi = 0, ibad_zlo = 0, ibad_zhi = 0;
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int idx = tid + it * NTH;
    const int vox = idx / Q, q = idx % Q;
    const int hx = vox % HX, hy = (vox / HX) % HY, hz = vox / (HX * HY);
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
    voff[it] = (unsigned)(((dzr * a.H + dyr) * rowf + xf + biasf) * 4);
    ldst[it] = XP ? ((hx & 1) * PLANE + (hz * HY + hy) * HXP + (hx >> 1)) * 8 + (q & 1) * 4
                  : ((q >> 1) * PLANE + (hz * HY + hy) * HX + hx) * 8 + (q & 1) * 4;
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
  const int cper = xb && Csrc >= CB ? Csrc / CB : 0;

  auto decode = [&](int tile_lin, int& n, int& tx, int& ty, int& tz) {
    unsigned t = (unsigned)tile_lin, q;
    q = ka.tiles_x == 1 ? t : __umulhi(t, ka.mx); tx = (int)(t - q * ka.tiles_x); t = q;
    q = ka.tiles_y == 1 ? t : __umulhi(t, ka.my); ty = (int)(t - q * ka.tiles_y); t = q;
    q = ka.tiles_z == 1 ? t : __umulhi(t, ka.mz); tz = (int)(t - q * ka.tiles_z); n = (int)q;
  };

  const f32x4* w_cg = reinterpret_cast<const f32x4*>(a.w_packed) + (size_t)cg * ka.nchunks * (W_H / 8);
  f32x4 ibuf[IN_IT];
  f32x4 wbuf[W_IT];
  // normalise-on-load prologue (plain layers, dense input): the input is the RAW output of the previous conv of a
  // contract block; InstanceNorm with the given statistics, LeakyReLU and that conv's dropout are applied on the way into
  // LDS (unet3D_module.py:231-237), as conv3d_xp8w.hip does for the full-resolution layers.  Run-time flag: the element
  // index of a piece is (soff + voff) / 4 - biasf because the tensor is dense (in_pitch == Cin), so it costs no table.
  const bool pre = !XP && a.in_mean != nullptr;
  // Round 5: the prologue on the SKIP half of an x-blocked concat input (expand_2_1 reading contr_2_2's raw output from the
  // concat buffer): only the chunks of half 1 are normalised, the statistics have Csrc channels, and the element index of a piece
  // (in the producing layer's dense [voxel][Csrc] space) follows from its float offset F in the buffer [..][W/xb][2][xb][Csrc]:
  // E = (F >> (xs + 1) << xs) | (F & (2^xs - 1)) with 2^xs = xb * Csrc (the dispatch admits powers of two only)
  const int pre_xs = (pre && xb) ? 31 - __builtin_clz((unsigned)(xb * Csrc)) : 0;
  bool p_on = false;          // the staged item's chunk takes the prologue
  // pool-finish on load (round 4, vx_conv3d_args.in_pool_flags): the input is the previous block's window maxima of RAW values;
  // statistics, LeakyReLU, the dropout's 2 and the zero of a dropped element are applied here with vx_pool_finish's expressions
  // (the separate pass over the pooled tensor and the tensor itself disappear).  One flag word per 16-byte piece: the flags
  // tensor is the input tensor's image at a quarter of every byte offset (dense 8-channel input).
  // Only the instances the 8 -> 16 layer of a contract block runs on carry the code (one more buffer load per staged piece).
  constexpr bool POOLFIN = CB == 8 && XP == 0 && NT == 1 && (EPI == 0 || EPI == 3);
  const bool prepool = POOLFIN && pre && a.in_pool_flags != nullptr;
  uint32_t fbuf[POOLFIN ? IN_IT : 1];
  const int in_rep = a.in_repeat > 1 ? a.in_repeat : 1;
  f32x4 p_mean = {0.f, 0.f, 0.f, 0.f}, p_rstd = {1.f, 1.f, 1.f, 1.f};
  unsigned p_bad = 0, p_e0 = 0;
  vx_dkey p_key = {0u, 0u};
  const bool w_resident = ka.nchunks == 1 || ka.w_all;
  bool w_fresh = true;

  // the prologue's dropout seed: read ONCE (a device word under hipGraph replay: a load inside the item loop would make
  // every item wait for all the loads in flight -- s_waitcnt vmcnt(0) -- before it can use the word)
  const uint32_t seed_in = pre ? vx_seed_of(a, a.in_drop_seed) : 0u;
  const uint32_t seed_out = vx_seed_of(a, a.drop_seed);
  // the loads of the next item: addressing set up by prefetch(); issued either right there or, for the two-barrier
  // schedule of the plain layers, one per K-step INSIDE the multiply loop (pf_issue(s) behind step s): issuing all of
  // them at once after the barrier kept every wave of the workgroup in the load queue for 2 000 - 3 400 cycles per item
  // (72 KB through the CU's one address path) with the matrix pipe idle -- 15-25 % of the item (tools/stamp_s16.py)
  __amdgpu_buffer_rsrc_t pf_srd = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t pf_fsrd = pf_srd;
  unsigned pf_bad = 0xFFFFFFFFu, pf_soff = 0;
  auto pf_issue = [&](int it) {
    const unsigned vo = ((pf_bad >> it) & 1u) ? VX_OOB : voff[it];
    ibuf[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pf_srd, (int)vo, (int)pf_soff, 0));
    if constexpr (POOLFIN)   // (no run-time branch around the load: without flags it is steered out of range)
      fbuf[it] = __builtin_amdgcn_raw_buffer_load_b32(pf_fsrd, (int)((prepool && !((pf_bad >> it) & 1u)) ? (voff[it] >> 2) : VX_OOB),
                                                      (int)(pf_soff >> 2), 0);
  };
  auto prefetch = [&](int tile_lin, int chunk, bool have, bool with_w, bool deferred = false) {
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
    const int n_in = have ? n / in_rep : 0;
    pf_srd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)n_in * in_sample - biasf), 0, VX_NUMREC, 0x00020000);
    if (prepool)   // (in_sample and biasf are multiples of 8 floats here: the quarter offsets are exact)
      pf_fsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in_pool_flags + ((ptrdiff_t)n_in * (ptrdiff_t)in_sample - biasf) / 4), 0, VX_NUMREC,
                                                  0x00020000);
    pf_bad = bad;
    pf_soff = soff;
    if (!deferred) {
#pragma unroll
      for (int it = 0; it < IN_IT; ++it) pf_issue(it);
    }
    if (pre) {
      p_bad = bad;
      p_e0 = (soff >> 2) - (unsigned)biasf;
      p_key = vx_drop_key(seed_in, a.in_drop_layer, (uint32_t)n);
      p_on = !xb || (cper && chunk / cper == 1);
      // (n_in = 0 when the workgroup has run out of tiles: a valid address -- no branch around the loads, a join behind
      // one makes the compiler wait for every load in flight)
      const size_t mo = xb ? (size_t)n_in * Csrc + (cper ? chunk % cper : 0) * CB + (tid % Q) * 4
                           : (size_t)n_in * a.Cin + chunk * CB + (tid % Q) * 4;
      p_mean = *reinterpret_cast<const f32x4*>(a.in_mean + mo);
      p_rstd = *reinterpret_cast<const f32x4*>(a.in_rstd + mo);
    }
    const f32x4* src = w_cg + (size_t)chunk * (W_H / 8);
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int idx = tid + it * NTH;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (have && with_w && idx < W_H / 8) v = src[idx];
      wbuf[it] = v;
    }
  };
  auto commit = [&](bool with_w, int bofs = 0) {
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      if (tid + it * NTH < NHALO * Q) {
        f16x4 hi, lo;
        if (prepool) {
          f32x4 v = ibuf[it];
          const uint32_t fl = fbuf[POOLFIN ? it : 0];
          const float s2 = a.in_drop_mode == VX_DROP_HASH ? 2.f : 1.f;
          const bool outside = (p_bad >> it) & 1u;             // zero padding belongs to the pooled, normalised tensor
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float t = (v[j] - p_mean[j]) * p_rstd[j];
            float w = fmaxf(t, 0.01f * t) * s2;                // vx_pool_finish: normalise, LeakyReLU, then the dropout's 2
            if ((fl >> j) & 1u) w = fmaxf(w, 0.f);              // a dropped element contributes 0 to the window
            v[j] = outside ? 0.f : w;
          }
          ibuf[it] = v;
        } else if (pre && p_on) {
          f32x4 v = ibuf[it];
          // dropout's factor 2 rides in the scale: 2 lrelu(t) = lrelu(2 t)
          const f32x4 sc = p_rstd * (a.in_drop_mode == VX_DROP_HASH ? 2.f : 1.f);
          uint32_t bits = 0xFu;
          if (a.in_drop_mode == VX_DROP_HASH) {
            unsigned e = p_e0 + (voff[it] >> 2);
            if (xb) e = ((e >> (pre_xs + 1)) << pre_xs) | (e & ((1u << pre_xs) - 1u));
            bits = vx_drop_bits4(p_key, e);
          }
          if ((p_bad >> it) & 1u) bits = 0u;             // zero padding belongs to the normalised tensor
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = (v[j] - p_mean[j]) * sc[j];
            t = fmaxf(t, 0.01f * t);
            const int keep = __builtin_amdgcn_sbfe(bits, j, 1);
            v[j] = __int_as_float(__float_as_int(t) & keep);
          }
          ibuf[it] = v;
        }
        vx_split4(ibuf[it], hi, lo);
        *reinterpret_cast<f16x4*>(s_hi + bofs + ldst[it]) = hi;
        *reinterpret_cast<f16x4*>(s_lo + bofs + ldst[it]) = lo;
      }
    }
    if (with_w) {
#pragma unroll
      for (int it = 0; it < W_IT; ++it) {
        const int idx = tid + it * NTH;
        if (idx < W_H / 8) reinterpret_cast<f32x4*>(s_w)[idx] = wbuf[it];
      }
    }
  };

  f32x4 bias4[NT];
  bool cvalid[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = XP ? (g & 1) * 4 : (cg * NT + nt) * 16 + g * 4;
    cvalid[nt] = co < a.Cout;
    bias4[nt] = cvalid[nt] ? *reinterpret_cast<const f32x4*>(a.bias + co) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // fused head (x-pair only): this lane's 4 of the 8 weights of up to HC classes; the bias rides in the g-even lane
  constexpr int HC = XP ? 4 : 1;
  float hw4[HC][4], hb[HC];
#pragma unroll
  for (int c = 0; c < HC; ++c) {
    hb[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) hw4[c][k] = 0.f;
    if (f_head && c < a.head_C) {
      if (!(g & 1)) hb[c] = a.head_b[c];
#pragma unroll
      for (int k = 0; k < 4; ++k) hw4[c][k] = a.head_w[c * 8 + (g & 1) * 4 + k];
    }
  }

  if (ka.w_all && ka.nchunks > 1) {   // all chunks' weights resident: one cooperative copy for the kernel's life
    for (int i = tid; i < ka.nchunks * (W_H / 8); i += NTH) reinterpret_cast<f32x4*>(s_w)[i] = w_cg[i];
  }
  int tile_lin = blockIdx.x, chunk = 0;
  if ((gridDim.x & 7) == 0) tile_lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  bool have = tile_lin < total;
  prefetch(tile_lin, 0, have, !(ka.w_all && ka.nchunks > 1));
  if (ka.w_all && ka.nchunks > 1) w_fresh = false;
  f32x4 acc[R][NT], accx[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; accx[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  // tiles whose per-wave statistics sit in s_red, not yet combined.  B = the item of the latest epilogue; A (DB == 2
  // only) = the one before, whose waves 4..7 wrote their part one barrier later.  s_red holds 1 / 2 / 3 items' slots
  // for DB = 0 / 1 / 2.
  constexpr int RED_F = NW * NT * 32;   // floats of one item's slots
  int pendA_n = -1, pendA_tile = 0, pendA_red = 0, pendB_n = -1, pendB_tile = 0, pendB_red = 0;
  auto flush_one = [&](int pn, int ptile, int pred) {
    if (pn >= 0 && tid < NT * 16) {
      const int nt = tid / 16, c = tid % 16;
      const int co = XP ? c : (cg * NT + nt) * 16 + c;   // x-pair: rows c and c + 8 are the two x of channel c
      if (co < a.Cout && (!XP || c < 8)) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          s += s_red[pred + ((w * NT + nt) * 16 + c) * 2 + 0];
          q += s_red[pred + ((w * NT + nt) * 16 + c) * 2 + 1];
          if (XP) {
            s += s_red[pred + ((w * NT + nt) * 16 + c + 8) * 2 + 0];
            q += s_red[pred + ((w * NT + nt) * 16 + c + 8) * 2 + 1];
          }
        }
        float* dst = a.stats_partial + (((size_t)pn * ntiles + ptile) * a.Cout + co) * 2;
        dst[0] = s;
        dst[1] = q;
      }
    }
  };
  // after a barrier of the item loop: combine what is complete
  auto flush_stats = [&]() {
    if constexpr (STAG) {
      flush_one(pendA_n, pendA_tile, pendA_red);
      pendA_n = pendB_n; pendA_tile = pendB_tile; pendA_red = pendB_red;
    } else {
      flush_one(pendB_n, pendB_tile, pendB_red);
    }
    pendB_n = -1;
  };

  float rmax = 0.f;   // largest |value| this lane stored (range guard of the split-fp16 consumers, vx_conv3d_args.range_flag)
  auto range_out = [&]() {
    if (a.range_flag) {
      float mx = rmax;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      if (lane == 0 && !(mx < 32768.f)) atomicMax(a.range_flag, __float_as_uint(mx));   // see conv3d_xp8w.hip
    }
  };
  // ---- epilogue of item tl (conv3d_mfma.hip); its statistics go to the s_red slots at redo ----
  auto epilogue = [&](int tl, int redo) {
    // ---- epilogue (conv3d_mfma.hip) ----
    int n, tx, ty, tz;
    decode(tl, n, tx, ty, tz);
    unsigned obad = 0;
    if (tx == ka.tiles_x - 1) obad |= obad_xhi;
    if (ty == ka.tiles_y - 1) obad |= obad_yhi;
    if (tz == ka.tiles_z - 1) obad |= obad_zhi;
    const unsigned vox0 = (unsigned)(((tz * TZ) * a.H + ty * TY) * a.W + tx * TXV);
    const unsigned osoff = vox0 * (unsigned)a.out_pitch * 4u;
    const unsigned e0 = vox0 * (unsigned)a.Cout;
    const __amdgpu_buffer_rsrc_t osrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * out_sample), 0, VX_NUMREC, 0x00020000);
    const vx_dkey dkey = vx_drop_key(seed_out, a.drop_layer, (uint32_t)n);
    const size_t hnvox = (size_t)a.D * a.H * a.W;
    int hflip = 0;
    float* hbase = nullptr;
    // row-reuse layout (a wave's R column tiles = R consecutive y-rows of one z): the head's output address of
    // column tile r is that of tile 0 plus r rows (minus, under a y-flip) -- computed once per item
    constexpr bool HROWS = XP == 1 && TX == 16 && TY % R == 0;
    float* ho0 = nullptr;
    int hystep = 0;
    if (f_head) {
      hflip = a.head_flip ? a.head_flip[n] : 0;
      hbase = a.head_out + (size_t)(a.head_dst ? a.head_dst[n] : n) * a.head_C * hnvox;
      if constexpr (HROWS) {
        int gx = tx * TXV + 2 * m + (g >> 1), gy = ty * TY + (wave * R) % TY, gz = tz * TZ + (wave * R) / TY;
        if (hflip & 1) gz = a.D - 1 - gz;
        if (hflip & 2) gy = a.H - 1 - gy;
        if (hflip & 4) gx = a.W - 1 - gx;
        ho0 = hbase + ((size_t)gz * a.H + gy) * a.W + gx;
        hystep = (hflip & 2) ? -a.W : a.W;
      }
    }

    float ssum[NT][4], ssq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) { ssum[nt][j] = 0.f; ssq[nt][j] = 0.f; }
    const bool f_range = a.range_flag != nullptr;

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const unsigned cshift = XP ? 0u : (unsigned)((cg * NT + nt) * 16);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool bad = ((obad >> r) & 1u) || !cvalid[nt];
        // main + cross * 2^-11 as ONE fma per element (the scaling is exact, so the same bits as multiply-then-add, which
        // -ffp-contract=off would otherwise keep as two instructions -- packed ones, at twice the issue cost here)
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(accx[r][nt][j], 1.0f / 2048.f, acc[r][nt][j]) + bias4[nt][j];
        if (a.stats_partial && !bad) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { ssum[nt][j] += v[j]; ssq[nt][j] = fmaf(v[j], v[j], ssq[nt][j]); }
        }
        if (f_lrelu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.01f * v[j]);
        } else if (f_relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        const unsigned e = e0 + eoff[r] + cshift;
        if (f_dhash) {
          const uint32_t bits = vx_drop_bits4(dkey, e);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);
        } else if (f_dmask) {
          uint32_t mk = 0;
          if (!bad) mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + (size_t)n * a.D * a.H * a.W * a.Cout + e);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * v[j] : 0.f;
        }
        if (f_head) {
          // fused 1x1x1 head (conv1x1.hip): this lane holds channels 4 (g & 1) .. + 3 of voxel 2p + (g >> 1), the
          // lane 16 further (g ^ 1) the other four -- one cross-lane add per class, then the g-even lane stores
          // (a wave's stores of one class cover 32 consecutive voxels).  The lane's 4 weights per class and the
          // bias were selected once per workgroup (hw4 / hb).  Two partial chains + one add: not bit-equal to
          // conv1x1.hip's single chain.
          float* o;
          if constexpr (HROWS) {
            o = ho0 + (ptrdiff_t)r * hystep;
          } else {
            const int vv = (wave * R + r) * 16 + m;
            int gx = tx * TXV + 2 * (vv % TX) + (g >> 1), gy = ty * TY + (vv / TX) % TY, gz = tz * TZ + vv / (TX * TY);
            if (hflip & 1) gz = a.D - 1 - gz;
            if (hflip & 2) gy = a.H - 1 - gy;
            if (hflip & 4) gx = a.W - 1 - gx;
            o = hbase + ((size_t)gz * a.H + gy) * a.W + gx;
          }
#pragma unroll
          for (int c = 0; c < HC; ++c) {
            if (c < a.head_C) {
              float part = hb[c];
#pragma unroll
              for (int k = 0; k < 4; ++k) part = fmaf(hw4[c][k], v[k], part);
              part = vx_add_xor16(part);
              if (!bad && !(g & 1)) o[(size_t)c * hnvox] = part;
            }
          }
        }
        if (f_store) {
          if (f_range && !bad) rmax = fmaxf(fmaxf(rmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
          u32x4 sv = __builtin_bit_cast(u32x4, v);
          if (a.out_split) {   // the consumer is the fused up-convolution: hand the piece over as the fp16 pairs it multiplies
            // (integer lanes all the way: hipcc 7.2 mishandles bit patterns that travel through float lanes of a wider
            // vector -- tools/micro/load_b64_narrow.hip)
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            f16x4 hi, lo;
            vx_split4(v, hi, lo);
            const u32x2 h2 = __builtin_bit_cast(u32x2, hi), l2 = __builtin_bit_cast(u32x2, lo);
            sv = (u32x4){h2[0], h2[1], l2[0], l2[1]};
          }
          const unsigned vo = bad ? VX_OOB : ovoff[r] + cshift * 4u;
          __builtin_amdgcn_raw_buffer_store_b128(sv, osrd, (int)vo, (int)osoff, 0);
          // gfx950 store-data hazard with an SGPR soffset (conv3d_mfma.hip)
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 3" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    if (a.stats_partial) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s = ssum[nt][j], q = ssq[nt][j];
#pragma unroll
          for (int rot = 8; rot >= 1; rot >>= 1) {   // sum over the row group's 16 columns: DPP row rotations, no LDS-queue shuffles
            s += vx_row_ror(s, rot);
            q += vx_row_ror(q, rot);
          }
          if (m == 0) {
            s_red[redo + ((wave * NT + nt) * 16 + g * 4 + j) * 2 + 0] = s;
            s_red[redo + ((wave * NT + nt) * 16 + g * 4 + j) * 2 + 1] = q;
          }
        }
      // the cross-wave sum and the global write wait for the NEXT barrier of the item loop (flush_stats): no
      // extra barrier in the epilogue
      pendB_n = n;
      pendB_red = redo;
      pendB_tile = tl - n * ntiles;
    }
    if constexpr (!SINGLE) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; accx[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
  };

  // ---- the multiply phase of one item: image at halves offset cofs, weights of chunk ck ----
  auto multiply = [&](int cofs, int ck, bool issue_loads = false) {
    const _Float16* s_wc = s_w + (ka.w_all ? ck * W_H : 0);   // this item's weights

    if constexpr (XP == 1 && TX == 16 && TY % R == 0) {
      // ---- x-pair, a wave's R column tiles are R consecutive y-rows of one z: the input row (z + kz, y) is the
      // (ky = 0) operand of output row y, the (ky = 1) operand of y - 1 and the (ky = 2) operand of y - 2, so per kz
      // the wave reads R + 2 row fragments once instead of 3 R -- the B-operand LDS traffic, which bounds this loop
      // (one ds_read_b128 pair per 3 MFMAs against 128 B/clk per CU), drops by 3R / (R + 2).
      const int p0 = vbase[0] + (g & 1) * PLANE + (g >> 1);   // row ly0, kz = ky = 0; x-offset ix = g
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        f16x8 bh[R + 2], bl[R + 2];
#pragma unroll
        for (int j = 0; j < R + 2; ++j) {
          const int p = (p0 + (kz * HY + j) * HXP) * 8;
          bh[j] = *reinterpret_cast<const f16x8*>(s_hi + cofs + p);
          bl[j] = *reinterpret_cast<const f16x8*>(s_lo + cofs + p);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const _Float16* wp = s_wc + ((((kz * 3 + ky) * NT) * 2) * WL + wslot) * 8;
          const f16x8 ah = *reinterpret_cast<const f16x8*>(wp);
          const f16x8 al = *reinterpret_cast<const f16x8*>(wp + WL * 8);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            // DB: one item = one tile, so the first step starts from zero (no accumulator clearing in the epilogue)
            const bool fresh = SINGLE && kz == 0 && ky == 0;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[r + ky], fresh ? zero : acc[r][0], 0, 0, 0);
            accx[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[r + ky], fresh ? zero : accx[r][0], 0, 0, 0);
            accx[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[r + ky], accx[r][0], 0, 0, 0);
          }
        }
      }
    } else {
      // ---- NSTEP steps x 3 x R x NT MFMAs; fragments of step s + 1 are read before the MFMAs of step s ----
      f16x8 ah[2][NT], al[2][NT], bh[2][R], bl[2][R];
      auto load_step = [&](int s, int slot) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const _Float16* wp = s_wc + (((s * NT + nt) * 2) * WL + wslot) * 8;
          ah[slot][nt] = *reinterpret_cast<const f16x8*>(wp);
          al[slot][nt] = *reinterpret_cast<const f16x8*>(wp + WL * 8);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          // column tile r of a wave sits a lane-independent distance from its tile 0 (16 r voxels further in the tile's
          // x-fastest order, never across the wave's z-plane): ONE address register per step, the rest is the
          // instruction's immediate offset
          constexpr int ROWS16 = 16 / TX;                                  // y-rows one column tile covers
          static_assert(TX == 16 || TX == 8 || TX == 4, "tile width");
          static_assert((R * ROWS16) % TY == 0 || TY % (R * ROWS16) == 0, "a wave's tiles do not straddle z-planes unevenly");
          const int dr = (((r * ROWS16) / TY) * HY + (r * ROWS16) % TY) * HXP;
          const int p = (vbase[0] + toff[s]) * 8 + dr * 8;
          bh[slot][r] = *reinterpret_cast<const f16x8*>(s_hi + cofs + p);
          bl[slot][r] = *reinterpret_cast<const f16x8*>(s_lo + cofs + p);
        }
      };
      if constexpr (SINGLE) {   // one item = one tile: the epilogue does not clear the accumulators
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) { acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; accx[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      }
      load_step(0, 0);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        if (s + 1 < NSTEP) load_step(s + 1, (s + 1) & 1);
        const int cur = s & 1;
        if (issue_loads) {
          // the next item's loads, spread over the steps (IN_IT <= NSTEP for every tile this path takes; else the rest
          // goes out with the last step)
          if (s < IN_IT) pf_issue(s);
          if (s == NSTEP - 1) {
#pragma unroll
            for (int it = NSTEP; it < IN_IT; ++it) pf_issue(it);
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][nt], bh[cur][r], acc[r][nt], 0, 0, 0);
            accx[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][nt], bl[cur][r], accx[r][nt], 0, 0, 0);
            accx[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur][nt], bh[cur][r], accx[r][nt], 0, 0, 0);
          }
#ifndef VX_S16_NO_PIN
        // The schedule, pinned: left alone hipcc sinks every fragment read to just before the matrix instruction that
        // consumes it (ds_read_b128, s_waitcnt lgkmcnt(0), v_mfma ... -- the LDS latency exposed a dozen times per step).
        //   mode 1: one read of step s + 1 behind each of this step's first matrix instructions (a whole step of cover;
        //           two sets of fragments live);
        //   mode 2: the reads of column tile r of step s + 1 behind the matrix instructions of tile r of step s, whose
        //           fragment registers they can take over (three quarters of a step of cover, no extra registers).
        // Same-process A/B against the unpinned loop: mode 1 -7..-9 % on every 16-channel-chunk instance (-3 % on the 8 -> 16
        // layer), mode 2 -5.5..-7 % on the four-tile instances; mode 1 everywhere (no instance spills since the fragment
        // addresses share one register per step).
        {
#ifdef VX_S16_PINMODE
          constexpr int PINMODE = VX_S16_PINMODE;
#else
          constexpr int PINMODE = 1;
#endif
          constexpr int NRD = 2 * NT + 2 * R, NMF = 3 * R * NT;
          constexpr int PAIRS = NRD < NMF ? NRD : NMF;
          if (s == 0) __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);    // step 0's own fragments
          if (s + 1 < NSTEP) {
            if constexpr (PINMODE == 1) {
#pragma unroll
              for (int i = 0; i < PAIRS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
              if (NRD > PAIRS) __builtin_amdgcn_sched_group_barrier(0x100, NRD - PAIRS, 0);
              if (NMF > PAIRS) __builtin_amdgcn_sched_group_barrier(0x008, NMF - PAIRS, 0);
            } else {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT, 0);             // the next step's weights
              __builtin_amdgcn_sched_group_barrier(0x008, 3 * NT - 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
              for (int r = 1; r < R; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              }
            }
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
          }
          if (issue_loads && s < IN_IT) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#else
        if (issue_loads && s < IN_IT) {
          // pin the load behind this step's matrix instructions (left alone the scheduler sinks all of them to the end
          // of the loop, where they queue up exactly as before)
          __builtin_amdgcn_sched_group_barrier(0x008, R * NT * 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#endif
      }
    }

  };

  // DB == 2: DB plus a stagger between the two waves of a SIMD (waves w and w + 4 of a 512-thread workgroup share
  // one): waves 4..7 run the epilogue of item k after the barrier of item k + 1, so their stores and statistics
  // overlap the MFMA loop of waves 0..3 and vice versa instead of every wave hitting the same phase together.
  const bool late = STAG && __builtin_amdgcn_readfirstlane(wave) >= NW / 2;
  int red_cur = 0, red_prev = 0, prev_tile = 0;
  bool have_prev = false;
  // DB: item 0 goes into image 0 before the loop, item 1's loads are in flight
  int db_cur = 0, db_ntile = 0, db_nchunk = 0;   // the item whose loads are in flight
  bool db_nhave = false;
  if constexpr (DB != 0) {
    commit(w_fresh, 0);
    w_fresh = false;
    db_ntile = tile_lin; db_nchunk = 1;
    if (db_nchunk == ka.nchunks) { db_nchunk = 0; db_ntile = tile_lin + (int)gridDim.x; }
    db_nhave = db_ntile < total;
    prefetch(db_ntile, db_nchunk, db_nhave, false);
  }

  // (A ping-pong schedule -- the two waves of a SIMD alternating between the matrix phase and everything else, one
  // barrier per phase -- was measured at +20 % time in round 2 and is no longer built; tools/rejected/README.md.)
#ifdef VX_CONV_STAMPS
  // phases: 0 barrier wait, 1 multiply, 2 epilogue, 3 wait for the prefetched loads, 4 convert + LDS write, 5 issue loads
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last, st_iters = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
  auto st_flush = [&]() {
    if (ka.stamps && lane == 0) {
      unsigned long long* d = ka.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
      for (int i = 0; i < 6; ++i) d[i] = st_sum[i];
      d[6] = st_iters;
    }
  };
#endif
  while (have) {
    int ntile, nchunk;
    bool nhave;
    int cofs = 0;    // halves offset of the image this item reads
    if constexpr (DB != 0) {
      __syncthreads();       // image db_cur is complete; everyone is done reading image db_cur ^ 1
      VX_STAMP(0);
      flush_stats();
      if constexpr (STAG) {
        if (late && have_prev) { epilogue(prev_tile, red_prev); have_prev = false; }
        VX_STAMP(2);
      }
      ntile = db_ntile; nchunk = db_nchunk; nhave = db_nhave;   // the next item: its loads are in flight
      db_nchunk = nchunk + 1; db_ntile = ntile;                  // and the one after it
      if (db_nchunk == ka.nchunks) { db_nchunk = 0; db_ntile = ntile + (int)gridDim.x; }
      db_nhave = db_ntile < total;
      cofs = db_cur * BUF_H;
      db_cur ^= 1;
      if (!late) {
        VX_STAMP_WAIT_LOADS();
        VX_STAMP(3);
        if (nhave) commit(false, db_cur * BUF_H);                // next item -> the other image (waits for its loads)
        VX_STAMP(4);
        prefetch(db_ntile, db_nchunk, db_nhave, false);          // the item after next -> registers
        VX_STAMP(5);
      }
    } else {
      if (S16_DBG < 3) __syncthreads();
      VX_STAMP(0);
      flush_stats();
      VX_STAMP_WAIT_LOADS();
      VX_STAMP(3);
      if (S16_DBG < 2) commit(w_fresh);
      VX_STAMP(4);
      if (S16_DBG < 3) __syncthreads();
      VX_STAMP(0);
      w_fresh = !w_resident;
      ntile = tile_lin; nchunk = chunk + 1;
      if (nchunk == ka.nchunks) { nchunk = 0; ntile = tile_lin + (int)gridDim.x; }
      nhave = ntile < total;
      if (S16_DBG < 2) prefetch(ntile, nchunk, nhave, !w_resident, DEFER);
      VX_STAMP(5);
    }
    multiply(cofs, chunk, DB == 0 && DEFER && S16_DBG < 2);
    VX_STAMP(1);
    if constexpr (STAG) {   // waves 4..7 stage the next item after their MFMA loop: E M C against C M E of waves 0..3
      if (late) {
        VX_STAMP_WAIT_LOADS();
        VX_STAMP(3);
        if (nhave) commit(false, db_cur * BUF_H);
        VX_STAMP(4);
        prefetch(db_ntile, db_nchunk, db_nhave, false);
        VX_STAMP(5);
      }
    }
    if (S16_DBG >= 1) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" :: "v"(acc[r][nt]), "v"(accx[r][nt]));
    } else if (chunk == ka.nchunks - 1) {
      if constexpr (STAG) {
        if (!late) epilogue(tile_lin, red_cur);
        prev_tile = tile_lin; have_prev = true;
        red_prev = red_cur;
        red_cur = red_cur == 2 * RED_F ? 0 : red_cur + RED_F;
      } else {
        epilogue(tile_lin, red_cur);
        if constexpr (DB == 1) red_cur ^= RED_F;
      }
    }
    VX_STAMP(2);
#ifdef VX_CONV_STAMPS
    ++st_iters;
#endif
    tile_lin = ntile; chunk = nchunk; have = nhave;
  }
  if constexpr (STAG) {
    if (late && have_prev) epilogue(prev_tile, red_prev);
  }
  __syncthreads();
  flush_stats();
  if constexpr (STAG) flush_stats();
  range_out();
#ifdef VX_CONV_STAMPS
  st_flush();
#endif
}

// The instances of this file compile in THREE translation units (the Makefile builds conv3d_s16_cb8.o (a two-line source
// defining S16_PART = 1) and conv3d_s16_nt2.o (S16_PART = 2)): the one-row-tile 16-channel-chunk instances + everything
// host-side here, the 8-channel-chunk instances (x-pair included) and the two-row-tile ones there -- as one unit the file was
// the longest compile of the library by a factor of three.
#ifndef S16_PART
#define S16_PART 0
#endif
#if S16_PART == 0
// ---------------------------------------------------------------------------------------------
// weight packing: torch (Cout, Cin, 3,3,3) fp32 -> [rowgroup][chunk][step][nt][hi | lo][lane 64][8 halves]
__global__ void pack_conv3d_k3_s16_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int Cin, int Cout, int CB,
                                          int NT, int XP, int64_t total) {
  const int TPS = 32 / CB, NSTEP = XP ? 9 : (27 + TPS - 1) / TPS, nchunks = Cin / CB;
  const int WL = XP ? 32 : 64;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int j = r % 8; r /= 8;
    const int lane = r % WL; r /= WL;
    const int hl = r % 2; r /= 2;
    const int nt = r % NT; r /= NT;
    const int step = r % NSTEP; r /= NSTEP;
    const int chunk = r % nchunks; r /= nchunks;
    const int rgrp = (int)r;
    float v = 0.f;
    if (XP) {   // piece [co >> 2][kx][co & 3] of step (kz, ky); kx = 3 is the zero k-group
      const int kx = (lane >> 2) & 3, co = (lane >> 4) * 4 + (lane & 3), ci = chunk * CB + j;
      if (kx <= 2) v = w[((size_t)co * Cin + ci) * 27 + step * 3 + kx];
    } else {
      const int row = (rgrp * NT + nt) * 16 + (lane & 15);
      const int kg = lane >> 4;
      const int tap = CB == 16 ? 2 * step + (kg >> 1) : 4 * step + kg;
      const int ci = chunk * CB + (CB == 16 ? 8 * (kg & 1) + j : j);
      if (row < Cout && tap < 27) v = w[((size_t)row * Cin + ci) * 27 + tap];
    }
    const float c = fminf(fmaxf(v, -65504.f), 65504.f);
    const _Float16 h = (_Float16)c;
    out[i] = hl == 0 ? h : (_Float16)((v - (float)h) * 2048.f);
  }
}

struct S16Cfg { int CB, NT, XP; };
static inline S16Cfg s16_config(int Cin, int Cout) {
  S16Cfg c;
  c.NT = (Cout % 32 == 0) ? 2 : 1;
  c.XP = Cout == 8 ? 1 : 0;
  c.CB = c.XP ? 8 : ((Cin % 16 == 0) ? 16 : 8);
  return c;
}
static inline int s16_rows_padded(int Cout, int NT) { return ((Cout + 16 * NT - 1) / (16 * NT)) * (16 * NT); }

int64_t vx_conv3d_s16_packed_floats(int Cin, int Cout) {
  const S16Cfg c = s16_config(Cin, Cout);
  const int TPS = 32 / c.CB, NSTEP = c.XP ? 9 : (27 + TPS - 1) / TPS;
  const int64_t halves = (int64_t)(s16_rows_padded(Cout, c.NT) / 16) * (Cin / c.CB) * NSTEP * 2 * (c.XP ? 32 : 64) * 8;
  return halves / 2;
}

int vx_pack_conv3d_k3_s16(const float* w_torch, float* w_packed, int Cin, int Cout, hipStream_t s) {
  const S16Cfg c = s16_config(Cin, Cout);
  const int64_t total = vx_conv3d_s16_packed_floats(Cin, Cout) * 2;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_conv3d_k3_s16_kernel, dim3(blocks), dim3(256), 0, s, w_torch, reinterpret_cast<_Float16*>(w_packed),
                     Cin, Cout, c.CB, c.NT, c.XP, total);
  VX_CHECK_LAUNCH("vx_pack_conv3d_k3(s16)");
  return VX_OK;
}

#endif   // S16_PART == 0

template <int CB, int NT, int TX, int TY, int TZ, int NW, int XP, int DB = 0, int EPI = 3>
static int launch_s16(const ConvSArgs& ka_in, hipStream_t s) {
  constexpr int TXV = XP ? 2 * TX : TX;
  constexpr int NHALO = (TXV + 2) * (TY + 2) * (TZ + 2);
  constexpr int PLANE = (((XP ? NHALO / 2 : NHALO) + 15) / 16) * 16;
  constexpr int TPS = 32 / CB, NSTEP = XP ? 9 : (27 + TPS - 1) / TPS;
  constexpr size_t img = (size_t)2 * (XP ? 2 : CB / 8) * PLANE * 8 * 2, wch = (size_t)NSTEP * NT * 2 * (XP ? 32 : 64) * 8 * 2;
  constexpr size_t red = (size_t)(DB >= 2 ? 3 : DB + 1) * NW * NT * 16 * 2 * 4;
  static_assert((DB ? 2 : 1) * img + (DB == 3 ? 2 : 1) * wch + red <= 160 * 1024, "LDS budget");
  ConvSArgs ka = ka_in;
  ka.w_all = DB ? (ka.nchunks > 1 ? 1 : 0)   // double-buffered variants: the dispatch made sure everything fits
                : ((ka.nchunks > 1 && img + ka.nchunks * wch + red <= 160 * 1024) ? 1 : 0);
  const size_t lds = (DB ? 2 : 1) * img + (ka.w_all ? ka.nchunks : 1) * wch + red;
  auto kern = conv3d_k3_s16_kernel<CB, NT, TX, TY, TZ, NW, XP, DB, EPI>;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv3d_k3(s16): hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr_lds = lds;
  }
  const vx_conv3d_args& a = ka.a;
  const int total_tiles = ka.tiles_x * ka.tiles_y * ka.tiles_z * a.N;
  const int ygroups = XP ? 1 : (a.Cout + 16 * NT - 1) / (16 * NT);
  int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
  if (per_cu * NW > 16) per_cu = 16 / NW > 0 ? 16 / NW : 1;
  int gx = (256 * per_cu + ygroups - 1) / ygroups;
  if (gx > total_tiles) gx = total_tiles;
  static const char* kname = vx_kname("conv3d_k3_s16_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%d>", CB, NT, TX, TY, TZ, NW, XP, DB, EPI);
  vx_note_kernel(kname);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ygroups), dim3(64 * NW), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv3d_k3(s16)");
  return VX_OK;
}

static inline bool s16_dbplain() { return !vx_cfg().s16_no_dbplain && !vx_cfg().s16_generic; }

template <int CB, int NT, int XP>
static int dispatch_s16(const ConvSArgs& ka, int tx, hipStream_t s) {
  // epilogue specialisation of the large-tile instances (EPI in the kernel's header)
  const vx_conv3d_args& a = ka.a;
  int epi = 3;
  const bool generic = vx_cfg().s16_generic != 0;   // parity reference: run-time epilogue, one LDS image, two barriers per item
  if (generic) epi = 3;
  else if (a.act == VX_ACT_NONE && a.drop_mode == VX_DROP_NONE && !a.head_out && a.out) epi = 0;
  else if (a.act == VX_ACT_LRELU && a.drop_mode == VX_DROP_HASH && !a.head_out && a.out) epi = 1;
  else if (a.act == VX_ACT_LRELU && a.drop_mode == VX_DROP_HASH && a.head_out && XP) epi = 2;
  if constexpr (XP == 1) {   // single-chunk x-pair layers: double-buffered LDS image (one barrier per item)
    if (tx == 16 && ka.ty8 && ka.nchunks == 1 && !generic) {
      if (epi == 0) return launch_s16<CB, NT, 16, 8, 4, 8, XP, 2, 0>(ka, s);
      if (epi == 1) return launch_s16<CB, NT, 16, 8, 4, 8, XP, 2, 1>(ka, s);
      if (epi == 2) return launch_s16<CB, NT, 16, 8, 4, 8, XP, 2, 2>(ka, s);
      return launch_s16<CB, NT, 16, 8, 4, 8, XP, 2, 3>(ka, s);
    }
    if (tx == 16 && ka.ty8 && ka.nchunks == 2 && !generic) {   // 16 -> 8 channels
      if (epi == 1) return launch_s16<CB, NT, 16, 8, 4, 8, XP, 3, 1>(ka, s);
      return launch_s16<CB, NT, 16, 8, 4, 8, XP, 3, 3>(ka, s);
    }
  }
  if constexpr (XP == 0) {
    // Round 3: single-chunk plain layers on two LDS images with staggered SIMD partners (DB = 2) WHERE THE LAYER'S TILE FITS
    // TWICE: -11 % on 16 -> 32 at 16^3.  Shrinking the tile to make room loses more than the stagger wins: the 16-channel
    // chunks at 32^3 on 16 x 4 x 4 instead of 16 x 8 x 4 were 14-15 % SLOWER double-buffered, and 8 -> 16 at 32^3 (whose large
    // tile does fit twice) measured neutral (same-process A/B)
    if (tx == 16 && ka.nchunks == 1 && s16_dbplain()) {
      if (!ka.ty8) {                             // 16 x 4 x 4 tile (two row tiles, or H < 32)
        if (epi == 0) return launch_s16<CB, NT, 16, 4, 4, 8, XP, 2, 0>(ka, s);
        if (epi == 1) return launch_s16<CB, NT, 16, 4, 4, 8, XP, 2, 1>(ka, s);
        return launch_s16<CB, NT, 16, 4, 4, 8, XP, 2, 3>(ka, s);
      }
    }
  }
  if constexpr (NT == 1) {   // large layers: 4 column tiles per wave (vx_conv3d_s16_tile: never with two row tiles)
    if (tx == 16 && ka.ty8) {
      if (epi == 0) return launch_s16<CB, NT, 16, 8, 4, 8, XP, 0, 0>(ka, s);
      if (epi == 1) return launch_s16<CB, NT, 16, 8, 4, 8, XP, 0, 1>(ka, s);
      return launch_s16<CB, NT, 16, 8, 4, 8, XP>(ka, s);
    }
  }
  // (the decoder's LeakyReLU + hash-dropout epilogue as a compile-time instance on the small tiles too: its run-time form
  // re-tests five flags per column tile)
  if (tx == 16) {
    if (epi == 1) return launch_s16<CB, NT, 16, 4, 4, 8, XP, 0, 1>(ka, s);
    if (epi == 0) return launch_s16<CB, NT, 16, 4, 4, 8, XP, 0, 0>(ka, s);
    return launch_s16<CB, NT, 16, 4, 4, 8, XP>(ka, s);
  }
  if (tx == 8) {
    if (epi == 1) return launch_s16<CB, NT, 8, 8, 4, 8, XP, 0, 1>(ka, s);
    if (epi == 0) return launch_s16<CB, NT, 8, 8, 4, 8, XP, 0, 0>(ka, s);
    return launch_s16<CB, NT, 8, 8, 4, 8, XP>(ka, s);
  }
  return launch_s16<CB, NT, 4, 4, 4, 4, XP>(ka, s);
}

#if S16_PART == 2
// the 16-channel-chunk instances with two row tiles per wave (this translation unit: conv3d_s16_nt2.o)
int vx_conv3d_k3_s16_nt2(const ConvSArgs& ka, int tx, hipStream_t s) { return dispatch_s16<16, 2, 0>(ka, tx, s); }
#elif S16_PART == 1
// the 8-channel-chunk half of the dispatch (this translation unit: conv3d_s16_cb8.o)
int vx_conv3d_k3_s16_cb8(const ConvSArgs& ka, int nt, int xp, int tx, hipStream_t s) {
  if (xp) return dispatch_s16<8, 1, 1>(ka, tx, s);
  if (nt == 1) return dispatch_s16<8, 1, 0>(ka, tx, s);
  return dispatch_s16<8, 2, 0>(ka, tx, s);
}
#else
int vx_conv3d_k3_s16_cb8(const ConvSArgs& ka, int nt, int xp, int tx, hipStream_t s);
int vx_conv3d_k3_s16_nt2(const ConvSArgs& ka, int tx, hipStream_t s);
// tiles = those of conv3d_mfma.hip's tile_config (tx columns per row: 16 / 8 / 4 by W, or by W / 2 for x-pair)
// Tile of a layer: tx columns per row (16 / 8 / 4 by W, or by W / 2 for x-pair) as in conv3d_mfma.hip; large layers
// (H >= 32) take 16 x 8 x 4 tiles = 4 column tiles per wave: the halo read per output voxel drops from 2.5x to 2.1x
// and the per-item costs (barriers, decode, masks) halve -- +14..20 % on the 64^3 layers; small layers keep
// 16 x 4 x 4 so that the 256 CUs still get enough work items.
void vx_conv3d_s16_tile(int H, int W, int Cout, int* txv, int* ty, int* tz) {
  const int xp = s16_config(8, Cout).XP;
  const int wcols = xp ? W / 2 : W;
  const int tx = wcols >= 16 ? 16 : (wcols >= 8 ? 8 : 4);
  // two row tiles per wave (Cout % 32 == 0) on the large tile would need > 256 registers (75 spilled): small tile there
  // (measured 32->32 @64^3: 321 TFLOP/s on 16x4x4 tiles against 204 on the spilling 16x8x4 instance)
  const bool nt2 = !xp && Cout % 32 == 0;
  const bool ty8 = tx == 16 && H >= 32 && !nt2;
  *txv = xp ? 2 * tx : tx;
  *ty = (tx == 8 || ty8) ? 8 : 4;
  *tz = 4;
}

// +2.6 % end to end over the separate conv1x1 kernel (2088 vs 2036 volumes/s) once the lane's weights are hoisted out of
// the item loop; a first version with per-piece weight loads measured no gain.  VX_NO_HEAD_FUSION=1 (forward) disables it.
bool vx_conv3d_s16_head_fusable(int Cin, int Cout) { return s16_config(Cin, Cout).XP != 0; }
// the tile kernel takes the normalise-on-load prologue on its plain (not x-pair) layers
bool vx_conv3d_s16_prologue_ok(int Cin, int Cout) { return s16_config(Cin, Cout).XP == 0; }
int vx_conv3d_s16_head_max_classes() { return 4; }

int vx_conv3d_k3_s16(const vx_conv3d_args& a, hipStream_t s) {
  const S16Cfg c = s16_config(a.Cin, a.Cout);
  int txv, ty, tz;
  vx_conv3d_s16_tile(a.H, a.W, a.Cout, &txv, &ty, &tz);
  const int tx = c.XP ? txv / 2 : txv;
  const int ty8 = (tx == 16 && ty == 8) ? 1 : 0;
  ConvSArgs ka;
  ka.a = a;
  ka.tiles_x = (a.W + txv - 1) / txv; ka.tiles_y = (a.H + ty - 1) / ty; ka.tiles_z = (a.D + tz - 1) / tz;
  ka.nchunks = a.Cin / c.CB;
  ka.mx = (unsigned)((1ull << 32) / (unsigned)ka.tiles_x) + 1u;
  ka.my = (unsigned)((1ull << 32) / (unsigned)ka.tiles_y) + 1u;
  ka.mz = (unsigned)((1ull << 32) / (unsigned)ka.tiles_z) + 1u;
  ka.dbg = 0;
  ka.stamps = nullptr;
#ifdef VX_CONV_STAMPS
  if (const char* e = getenv("VX_S16_DBG")) ka.dbg = atoi(e);
  if (const char* e = getenv("VX_CONV_DBG_PTR")) ka.stamps = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  ka.ty8 = ty8;
  if (c.CB == 16 && c.NT == 1) return dispatch_s16<16, 1, 0>(ka, tx, s);
  if (c.CB == 16 && c.NT == 2) return vx_conv3d_k3_s16_nt2(ka, tx, s);
  return vx_conv3d_k3_s16_cb8(ka, c.NT, c.XP, tx, s);
}
#endif   // S16_PART
