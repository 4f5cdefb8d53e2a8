// K1 for the FULL-RESOLUTION layers (Cout = 8, Cin = 8 or 16: contr_1_2, expand_1_1, expand_1_2 -- 42 % of the forward):
// the x-pair split-fp16 schedule of conv3d_s16.hip (same arithmetic, same packed weights, same results) on a work
// decomposition built for these layers, which are bound by the bytes and instructions AROUND the matrix loop:
//
//   * a workgroup walks a z-COLUMN of one sample (32 x 8 voxels in x / y, TZ = 4 z-planes per item, 2 for the
//     two-chunk layer) and keeps a ROLLING window of z-planes in LDS: an item stages only its TZ new planes
//     (halo re-read 2.0x -> 1.33x: a third fewer loads, conversions and LDS writes per voxel);
//   * LDS holds three groups of TZ plane slots; a step writes one group while the item in flight reads the other two,
//     so there is ONE barrier per item and no address wrap inside a step;
//   * InstanceNorm statistics are summed in registers over the whole column (one sample!) and leave once per column,
//     not once per item (64 DPP + 16 LDS instructions per item gone); the bias enters as the first MFMA's C operand;
//   * every per-lane offset (staging source, LDS slot, store address, dropout element, head pointer) is computed once
//     per workgroup or per column; the item loop adds scalars;
//   * optional PROLOGUE: the input is the raw output of the previous contract block's conv; (x - mean) * rstd, LeakyReLU
//     and that block's dropout are applied on the way into LDS, so the normalise / fan-out passes over the full-resolution
//     tensors disappear (unet3D_module.py:231-237: conv -> InstanceNorm -> LeakyReLU -> Dropout).  For an MC-dropout
//     batch the T samples of a volume read the SAME raw tensor (in_repeat = T) with T different dropout patterns.
//   * the two waves of a SIMD are staggered as in conv3d_s16.hip (DB = 2): waves 0..3 stage, multiply, store; waves 4..7
//     store the previous item, multiply, stage.
//   * optional FUSED UP-CONVOLUTION (UP = 1, two-chunk layer): the up half of the decoder's concat input is never read --
//     and never written: the transposed 2x2x2 / stride-2 convolution that produces it (unet3D_module.py:332-356,
//     upscale -> cat -> expand) is evaluated while the step is staged.  A fine voxel depends on ONE coarse voxel, so the
//     up half of a step is 24 small GEMMs [(dx, co) 16 rows] x [16 coarse voxels] x [16 ci] per workgroup, one class
//     (z-plane, y-parity) per wave pair: the B operands come straight from global memory (a wave instruction reads 16
//     coarse voxels x 64 B, contiguous), three split-fp16 v_mfma_f32_16x16x16_f16 per tile, and the result (bias as the
//     C operand, zero outside the volume) is split and written into the LDS image where the staged loads would have put
//     it.  5.4 GB of HBM traffic per 320 samples (the write and the read of `up`) and one launch disappear.
// Restrictions (the dispatch falls back to conv3d_s16.hip otherwise): W % 32 == 0, H % 8 == 0, D % 4 == 0, Cout == 8,
// Cin in {8, 16}, dropout by hash or none (injected masks take the general kernels).
#include "s16_common.h"

struct Xp8Args {
  vx_conv3d_args a;
  int tiles_x, tiles_y, kz;   // columns per sample = tiles_x * tiles_y; kz = items per column
  int ncols;                  // columns in the launch (N * tiles_y * tiles_x)
  unsigned mcps, mtx;         // multiply-high magics: / (tiles_x * tiles_y), / tiles_x
  int stat_epc;               // statistics entries per column in stats_partial (entry 0 real, the rest zero)
  int no_xcd;
  unsigned long long* stamps;
};

#ifdef VX_CONV_STAMPS
#define XP_STAMP(i)                                                                      \
  do {                                                                                   \
    unsigned long long t_;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    st_sum[i] += t_ - st_last;                                                           \
    st_last = t_;                                                                        \
  } while (0)
#define XP_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define XP_STAMP(i) do {} while (0)
#define XP_WAIT_LOADS() do {} while (0)
#endif

// NCH: chunks of 8 input channels (1, 2).  EPI: 0 bias + statistics + store (an InstanceNorm follows), 1 LeakyReLU +
// hash dropout + store, 2 = 1 with the fused 1x1x1 head instead of the store, 3 LeakyReLU / ReLU / none without dropout
// (run-time act) + store.  PRE: 1 = the prologue above on the LAST chunk of the input.
template <int NCH, int EPI, int PRE, int UP>
__global__ __launch_bounds__(512) void conv3d_xp8_kernel(Xp8Args ka) {
  static_assert(UP == 0 || NCH == 2, "the fused up-convolution produces chunk 0 of a two-chunk layer");
  constexpr int NW = 8, NTH = 512;
  constexpr int TZ = 4 / NCH;
  constexpr int R = TZ;                       // column tiles (y-rows of one z-plane) per wave
  constexpr int WPZ = 8 / R;                  // waves per z-plane
  constexpr int HX = 34, HXP = 17, HY = 10;
  constexpr int ZP = HY * HXP;                // positions per z-plane and x-parity
  constexpr int NZ = 3 * TZ;
  constexpr int PP = ((NZ * ZP + 15) / 16) * 16;
  constexpr int PREC_H = 2 * PP * 8;          // halves of one precision plane (both parities)
  constexpr int CHUNK_H = 2 * PREC_H;
  constexpr int W_H = 9 * 2 * 32 * 8;         // halves of one chunk's weights ([step 9][hi|lo][32 pieces][8])
  constexpr int PPS = TZ * HX * HY * 2;       // 16-byte pieces per step and chunk
  constexpr int IT_C = (PPS + NTH - 1) / NTH;
  constexpr int IN_IT = NCH * IT_C;
  constexpr int GRP_H = TZ * ZP * 8;          // halves between two slot groups
  static_assert(IN_IT <= 8, "staging iterations");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* s_img = reinterpret_cast<_Float16*>(smem_raw);
  _Float16* s_w = s_img + NCH * CHUNK_H;
  float* s_red = reinterpret_cast<float*>(s_w + NCH * W_H);

  const vx_conv3d_args& a = ka.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int lz = wave / WPZ, ly0 = (wave % WPZ) * R;
  const bool late = wave >= NW / 2;
  const int cps = ka.tiles_x * ka.tiles_y;
  const int KZ = ka.kz;

  // ---- staging pattern of this thread (fixed for the kernel's life) ----
  const int xb = a.in_xblk;
  const int voxf = xb ? 16 : a.in_pitch;             // floats per voxel step along x (concat: two halves of 8)
  const int rowf = a.W * voxf;
  const int biasf = ((TZ - 1) * a.H + 1) * rowf + 4 * voxf;
  unsigned voff[IN_IT], erel[PRE ? IN_IT : 1];
  int ldst[IN_IT];
  unsigned ib_always = 0, ib_xlo = 0, ib_xhi = 0, ib_ylo = 0, ib_yhi = 0, ib_zfirst = 0, ib_zlast = 0;
  int qq_thread = 0;
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int chunk = NCH == 2 ? (it & 1) : 0;
    const int pidx = tid + (NCH == 2 ? (it >> 1) : it) * NTH;
    const int vox = pidx >> 1, qq = pidx & 1;
    qq_thread = qq;                                   // NTH is even: the same channel quad in every iteration
    const int pz = vox / (HX * HY), rem = vox % (HX * HY);
    const int hy = rem / HX, hx = rem % HX;
    const int dx = hx - 1, dy = hy - 1, dz = pz - (TZ - 1);
    int xf;
    if (xb) {
      const int blk = dx >= 0 ? dx / xb : -((-dx + xb - 1) / xb);
      xf = (blk * 2 + chunk) * xb * 8 + (dx - blk * xb) * 8 + qq * 4;
    } else {
      xf = dx * a.in_pitch + (UP ? 0 : chunk * 8) + qq * 4;   // UP: `in` is the skip tensor alone
    }
    voff[it] = (unsigned)(((dz * a.H + dy) * rowf + xf + biasf) * 4);
    ldst[it] = chunk * CHUNK_H + (hx & 1) * PP * 8 + ((pz * HY + hy) * HXP + (hx >> 1)) * 8 + qq * 4;
    if constexpr (PRE != 0) erel[it] = (unsigned)(((dz * a.H + dy) * a.W + dx) * 8 + qq * 4);
    if (pidx >= PPS) ib_always |= 1u << it;
    if (dx < 0) ib_xlo |= 1u << it;
    if (dx >= 32) ib_xhi |= 1u << it;
    if (dy < 0) ib_ylo |= 1u << it;
    if (dy >= 8) ib_yhi |= 1u << it;
    if (pz < TZ - 1) ib_zfirst |= 1u << it;           // step 0 of a column: planes -(TZ-1) .. -1 do not exist
    if (pz >= TZ - 1) ib_zlast |= 1u << it;           // step KZ: plane D does not exist
  }
  const size_t in_sample = (size_t)a.D * a.H * rowf;
  const int in_rep = a.in_repeat > 1 ? a.in_repeat : 1;

  // ---- fused up-convolution: this wave's class and its three column tiles (fixed for the kernel's life) ----
  // wave -> (plane u_pz of the step, y-parity u_ay, half u_th of the class's 6 column tiles).  Column c of a class is the
  // coarse voxel (Yi, Xi) = (c / 18, c % 18) of the 5 x 18 coarse positions whose fine row of parity u_ay lies in the
  // staged 10 x 34 window; rows of the product are (dx, co), so lane (m, g) ends with channels 4 (g & 1) .. + 3 of the
  // fine voxel x = 2 X + (g >> 1) -- the piece layout of the staged loads.
  constexpr int UT = 3;
  const int u_pz = wave >> 2, u_ay = (wave >> 1) & 1, u_th = wave & 1;
  const int Hc = a.H >> 1, Wc = a.W >> 1;
  const int urow = Wc * a.up_pitch;
  const int ubiasf = (Hc + 1) * urow + a.up_pitch;
  unsigned u_voff[UP ? UT : 1];
  int u_ldst[UP ? UT : 1];
  unsigned ub_always = 0, ub_xlo = 0, ub_xhi = 0, ub_ylo = 0, ub_yhi = 0, u_nowrite = 0;
  f16x4 u_ah = {0, 0, 0, 0}, u_al = {0, 0, 0, 0};
  if constexpr (UP != 0) {
#pragma unroll
    for (int i = 0; i < UT; ++i) {
      const int c = 16 * (UT * u_th + i) + m;
      const int Yi = c / 18, Xi = c % 18;
      const int Yrel = Yi + 1 - u_ay;                   // row of the 6-row coarse window (row 0 = coarse y of fine y = -1)
      const int zrel = u_pz ? 0 : -1;                   // plane 2 s - 1 comes from coarse plane s - 1, plane 2 s from s
      u_voff[i] = (unsigned)(((zrel * Hc + Yrel - 1) * urow + (Xi - 1) * a.up_pitch + g * 4 + ubiasf) * 4);
      const int hy = 2 * Yrel + u_ay - 1, hx = 2 * Xi + (g >> 1) - 1;
      u_ldst[i] = (hx & 1) * PP * 8 + ((u_pz * HY + hy) * HXP + (hx >> 1)) * 8 + (g & 1) * 4;
      if (c >= 90) { ub_always |= 1u << i; u_nowrite |= 1u << i; }
      if (hx < 0 || hx >= HX) u_nowrite |= 1u << i;
      if (Xi == 0) ub_xlo |= 1u << i;
      if (Xi == 17) ub_xhi |= 1u << i;
      if (Yrel == 0) ub_ylo |= 1u << i;
      if (Yrel == 5) ub_yhi |= 1u << i;
    }
    // A operand: row m = (dx, co), k = ci = 4 g .. 4 g + 3, tap (dz, dy) = (1 - u_pz, u_ay); vx_pack_convT_k2s2 layout
    // [dz][dy][ci][dx][co]
    f32x4 wv;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      wv[jj] = a.up_w[(((((1 - u_pz) * 2 + u_ay) * 16 + 4 * g + jj) * 2) + (m >> 3)) * 8 + (m & 7)];
    vx_split4(wv, u_ah, u_al);
  }
  const size_t up_sample = (size_t)(a.D >> 1) * Hc * urow;

  // ---- compute-phase constants ----
  // B fragment of (kz, row j): s_img[chunk][prec][parity g & 1][(slot * HY + ly0 + j) * HXP + m + (g >> 1)]
  const int bfrag0 = (g & 1) * PP * 8 + ((ly0 * HXP) + m + (g >> 1)) * 8;
  const int wslot = (((m & 7) >> 2) * 4 + ((g - (m >> 3)) & 3)) * 4 + (m & 3);   // conv3d_s16.hip: [co >> 2][kx][co & 3]

  // ---- epilogue constants: this lane stores voxel x = 2 m + (g >> 1) of row ly0 + r, channels 4 (g & 1) .. + 3 ----
  const int lx = 2 * m + (g >> 1), oc = (g & 1) * 4;
  unsigned ovoff[R], eoff[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ovox = (lz * a.H + ly0 + r) * a.W + lx;
    if (a.out_xblk) {
      const int oxb = a.out_xblk;
      ovoff[r] = (unsigned)((((lz * a.H + ly0 + r) * (2 * a.W * 8)) + ((lx / oxb) * 2 + a.out_half) * oxb * 8 + (lx % oxb) * 8 + oc) * 4);
    } else {
      ovoff[r] = (unsigned)((ovox * a.out_pitch + a.out_coff + oc) * 4);
    }
    eoff[r] = (unsigned)(ovox * 8 + oc);
  }
  const int out_voxf = a.out_xblk ? 16 : a.out_pitch;
  const size_t out_sample = (size_t)a.D * a.H * a.W * out_voxf;
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.bias + oc);
  const bool f_lrelu = EPI == 3 ? a.act == VX_ACT_LRELU : EPI != 0;
  const bool f_relu = EPI == 3 && a.act == VX_ACT_RELU;
  // fused head: this lane's 4 of the 8 weights of up to 4 classes; the bias rides in the g-even lane (conv3d_s16.hip)
  constexpr int HC = EPI == 2 ? 4 : 1;
  float hw4[HC][4], hb[HC];
#pragma unroll
  for (int c = 0; c < HC; ++c) {
    hb[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) hw4[c][k] = 0.f;
    if (EPI == 2 && c < a.head_C) {
      if (!(g & 1)) hb[c] = a.head_b[c];
#pragma unroll
      for (int k = 0; k < 4; ++k) hw4[c][k] = a.head_w[c * 8 + oc + k];
    }
  }
  const size_t hnvox = (size_t)a.D * a.H * a.W;

  // ---- weights: resident for the kernel's life ----
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.w_packed);
    for (int i = tid; i < NCH * W_H / 8; i += NTH) reinterpret_cast<f32x4*>(s_w)[i] = src[i];
  }

  // ---- the columns of this workgroup ----
  int vb = blockIdx.x;
  const int G = (int)gridDim.x;
  if ((G & 7) == 0 && !ka.no_xcd) vb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);   // one XCD: neighbouring columns
  const int ncol_wg = vb < ka.ncols ? (ka.ncols - vb + G - 1) / G : 0;

  struct Cur { int ci, s; };      // column number of this workgroup, step within the column (0 .. KZ)
  auto advance = [&](Cur& c) { if (++c.s > KZ) { c.s = 0; ++c.ci; } };
  auto col_of = [&](int ci, int& n, int& ty, int& tx) {
    const unsigned col = (unsigned)(vb + ci * G);
    const unsigned q = cps == 1 ? col : __umulhi(col, ka.mcps);
    n = (int)q;
    const unsigned rem = col - q * (unsigned)cps;
    const unsigned q2 = ka.tiles_x == 1 ? rem : __umulhi(rem, ka.mtx);
    ty = (int)q2;
    tx = (int)(rem - q2 * (unsigned)ka.tiles_x);
  };

  // ---- register staging: the loads of one step (and what its commit needs to know) ----
  f32x4 ibuf[IN_IT];
  f32x4 ubuf[UP ? UT : 1];
  unsigned p_ubad = 0;
  f32x4 p_mean = {0.f, 0.f, 0.f, 0.f}, p_rstd = {1.f, 1.f, 1.f, 1.f};
  unsigned p_bad = 0, p_e0 = 0, p_key = 0;

  auto prefetch = [&](const Cur& c) {
    const bool have = c.ci < ncol_wg;
    int n = 0, ty = 0, tx = 0;
    if (have) col_of(c.ci, n, ty, tx);
    unsigned bad = ib_always;
    if (tx == 0) bad |= ib_xlo;
    if (tx == ka.tiles_x - 1) bad |= ib_xhi;
    if (ty == 0) bad |= ib_ylo;
    if (ty == ka.tiles_y - 1) bad |= ib_yhi;
    if (c.s == 0) bad |= ib_zfirst;
    if (c.s == KZ) bad |= ib_zlast;
    if (!have) bad = 0xFFFFFFFFu;
    const int nin = n / in_rep;
    const unsigned soff = (unsigned)((((TZ * c.s) * a.H + ty * 8) * rowf + tx * 32 * voxf) * 4);
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.in + (size_t)nin * in_sample - biasf), 0, VX_NUMREC, 0x00020000);
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      if (UP != 0 && !(it & 1)) continue;               // chunk 0 is computed, not loaded
      const unsigned vo = ((bad >> it) & 1u) ? VX_OOB : voff[it];
      ibuf[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)vo, (int)soff, 0));
    }
    p_bad = bad;
    if constexpr (UP != 0) {
      unsigned ub = ub_always;
      if (tx == 0) ub |= ub_xlo;
      if (tx == ka.tiles_x - 1) ub |= ub_xhi;
      if (ty == 0) ub |= ub_ylo;
      if (ty == ka.tiles_y - 1) ub |= ub_yhi;
      if (!have || (c.s == 0 && u_pz == 0) || (c.s == KZ && u_pz == 1)) ub = 0xFFFFFFFFu;
      const unsigned usoff = (unsigned)(((c.s * Hc + ty * 4) * urow + tx * 16 * a.up_pitch) * 4);
      const __amdgpu_buffer_rsrc_t usrd = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.up_in + (size_t)n * up_sample - ubiasf), 0, VX_NUMREC, 0x00020000);
#pragma unroll
      for (int i = 0; i < UT; ++i) {
        const unsigned vo = ((ub >> i) & 1u) ? VX_OOB : u_voff[i];
        ubuf[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(usrd, (int)vo, (int)usoff, 0));
      }
      p_ubad = ub;
    }
    if constexpr (PRE != 0) {
      if (have) {
        p_mean = *reinterpret_cast<const f32x4*>(a.in_mean + (size_t)nin * 8 + qq_thread * 4);
        p_rstd = *reinterpret_cast<const f32x4*>(a.in_rstd + (size_t)nin * 8 + qq_thread * 4);
      }
      p_e0 = (unsigned)(((TZ * c.s) * a.H + ty * 8) * a.W + tx * 32) * 8u;
      p_key = vx_drop_key(vx_seed_of(a, a.in_drop_seed), a.in_drop_layer, (uint32_t)n);
    }
  };

  float rmax = 0.f;   // largest |value| this wave stored or produced (range guard of the split-fp16 consumers)
  f32x4 ubias4 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (UP != 0) ubias4 = *reinterpret_cast<const f32x4*>(a.up_b + (g & 1) * 4);

  auto commit = [&](int grp) {
    const int gofs = grp * GRP_H;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f};
    if constexpr (PRE != 0) {
      // dropout's factor 2 rides in the scale: 2 lrelu(t) = lrelu(2 t)
      sc = p_rstd * (a.in_drop_mode == VX_DROP_HASH ? 2.f : 1.f);
    }
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      if (UP != 0 && !(it & 1)) continue;
      if (tid + (NCH == 2 ? (it >> 1) : it) * NTH < PPS) {
        f32x4 v = ibuf[it];
        if constexpr (PRE != 0) if (NCH == 1 || (it & 1)) {
          uint32_t bits = 0xFu;
          if (a.in_drop_mode == VX_DROP_HASH) bits = vx_drop_bits4(p_key, p_e0 + erel[it]);
          if ((p_bad >> it) & 1u) bits = 0u;           // zero padding belongs to the normalised tensor
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = (v[j] - p_mean[j]) * sc[j];      // (x - mean) first: no cancellation against a rounded mean * rstd
            t = fmaxf(t, 0.01f * t);
            // keep ? t : 0  -- an all-ones / all-zeros word from one signed bit-field extract
            const int keep = __builtin_amdgcn_sbfe(bits, j, 1);
            v[j] = __int_as_float(__float_as_int(t) & keep);
          }
        }
        f16x4 hi, lo;
        vx_split4(v, hi, lo);
        *reinterpret_cast<f16x4*>(s_img + gofs + ldst[it]) = hi;
        *reinterpret_cast<f16x4*>(s_img + gofs + ldst[it] + PREC_H) = lo;
      }
    }
    if constexpr (UP != 0) {
      // the up half of the step: ConvTranspose3d(k = 2, s = 2) of the coarse voxels just loaded, three split products
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      f16x4 ubh[UT], ubl[UT];
#pragma unroll
      for (int i = 0; i < UT; ++i) vx_split4(ubuf[i], ubh[i], ubl[i]);
      // vx_split4 writes the lo halves from inline assembly: the compiler does not know a VALU result is about to be a
      // matrix operand and inserts no wait states for it (measured: stale lo operands without this)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < UT; ++i) {
        const f16x4 bh = ubh[i], bl = ubl[i];
        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x16f16(u_ah, bh, ubias4, 0, 0, 0);
        f32x4 dx = __builtin_amdgcn_mfma_f32_16x16x16f16(u_ah, bl, zero, 0, 0, 0);
        dx = __builtin_amdgcn_mfma_f32_16x16x16f16(u_al, bh, dx, 0, 0, 0);
        f32x4 v = d + dx * (1.0f / 2048.f);
        if ((p_ubad >> i) & 1u) v = zero;               // outside the volume: the conv's zero padding, not the bias
        rmax = fmaxf(fmaxf(rmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        f16x4 hi, lo;
        vx_split4(v, hi, lo);
        if (!((u_nowrite >> i) & 1u)) {
          *reinterpret_cast<f16x4*>(s_img + gofs + u_ldst[i]) = hi;
          *reinterpret_cast<f16x4*>(s_img + gofs + u_ldst[i] + PREC_H) = lo;
        }
      }
    }
  };

  // ---- accumulators ----
  f32x4 acc[R], accx[R];
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};

  // the multiply phase of the item whose first plane sits in slot rb (slots rb .. rb + TZ + 1, modulo NZ)
  auto multiply = [&](int rb) {
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk) {
      const _Float16* img = s_img + chunk * CHUNK_H + bfrag0;
      const _Float16* wch = s_w + chunk * W_H + wslot * 8;
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        int slot = rb + lz + kz;
        if (slot >= NZ) slot -= NZ;
        const _Float16* row0 = img + slot * (ZP * 8);
        f16x8 bh[R + 2], bl[R + 2];
#pragma unroll
        for (int j = 0; j < R + 2; ++j) {
          bh[j] = *reinterpret_cast<const f16x8*>(row0 + j * HXP * 8);
          bl[j] = *reinterpret_cast<const f16x8*>(row0 + j * HXP * 8 + PREC_H);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const _Float16* wp = wch + (kz * 3 + ky) * (2 * 32 * 8);
          const f16x8 ah = *reinterpret_cast<const f16x8*>(wp);
          const f16x8 al = *reinterpret_cast<const f16x8*>(wp + 32 * 8);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const bool fresh = chunk == 0 && kz == 0 && ky == 0;      // the bias is the first product's C operand
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[r + ky], fresh ? bias4 : acc[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[r + ky], fresh ? zero : accx[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[r + ky], accx[r], 0, 0, 0);
          }
        }
      }
    }
  };

  // ---- epilogue state of THIS wave (column it is storing) ----
  int e_ci = -1, e_n = 0, e_ty = 0, e_tx = 0, e_hflip = 0;
  uint32_t e_key = 0;
  float* e_ho = nullptr;     // head: this lane's output pointer for z0 = 0, r = 0 (un-flipped position)
  ptrdiff_t e_hz = 0, e_hy = 0;

  auto epilogue = [&](int ci, int k) {
    if (ci != e_ci) {   // a new column: sample, tile row / column, dropout key, head pointers
      e_ci = ci;
      col_of(ci, e_n, e_ty, e_tx);
      if (EPI == 1 || EPI == 2) e_key = vx_drop_key(vx_seed_of(a, a.drop_seed), a.drop_layer, (uint32_t)e_n);
      if (EPI == 2) {
        e_hflip = a.head_flip ? a.head_flip[e_n] : 0;
        const int slot = a.head_dst ? a.head_dst[e_n] : e_n;
        int gx = e_tx * 32 + lx, gy = e_ty * 8 + ly0, gz = lz;
        if (e_hflip & 1) gz = a.D - 1 - gz;
        if (e_hflip & 2) gy = a.H - 1 - gy;
        if (e_hflip & 4) gx = a.W - 1 - gx;
        e_ho = a.head_out + (size_t)slot * a.head_C * hnvox + ((size_t)gz * a.H + gy) * a.W + gx;
        e_hz = (ptrdiff_t)((e_hflip & 1) ? -1 : 1) * TZ * a.H * a.W;
        e_hy = (e_hflip & 2) ? -a.W : a.W;
      }
    }
    const unsigned vox0 = (unsigned)(((k * TZ) * a.H + e_ty * 8) * a.W + e_tx * 32);
    const unsigned osoff = a.out_xblk ? (unsigned)((((k * TZ) * a.H + e_ty * 8) * (2 * a.W * 8) + e_tx * 32 * 16) * 4)
                                      : vox0 * (unsigned)a.out_pitch * 4u;
    const unsigned e0 = vox0 * 8u;
    const __amdgpu_buffer_rsrc_t osrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)e_n * out_sample), 0, VX_NUMREC, 0x00020000);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      f32x4 v = acc[r] + accx[r] * (1.0f / 2048.f);
      if (EPI == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[j] += v[j]; ssq[j] = fmaf(v[j], v[j], ssq[j]); }
      }
      if (f_lrelu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.01f * v[j]);
      } else if (f_relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (EPI == 1 || EPI == 2) {
        const uint32_t bits = vx_drop_bits4(e_key, e0 + eoff[r]);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);
      }
      if (EPI == 2) {
        float* o = e_ho + (ptrdiff_t)k * e_hz + (ptrdiff_t)r * e_hy;
#pragma unroll
        for (int c = 0; c < HC; ++c) {
          if (c < a.head_C) {
            float part = hb[c];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) part = fmaf(hw4[c][kk], v[kk], part);
            part = vx_add_xor16(part);
            if (!(g & 1)) o[(size_t)c * hnvox] = part;
          }
        }
      } else {
        rmax = fmaxf(fmaxf(rmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), osrd, (int)ovoff[r], (int)osoff, 0);
        // gfx950 store-data hazard with an SGPR soffset (conv3d_mfma.hip)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 3" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (EPI == 0 && k == KZ - 1) {
      // the column is complete for this wave: sum over its 16 pair columns and leave the 4 x 2 values of row group g
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = ssum[j], q = ssq[j];
#pragma unroll
        for (int rot = 8; rot >= 1; rot >>= 1) { s += vx_row_ror(s, rot); q += vx_row_ror(q, rot); }
        if (m == 0) {
          s_red[(wave * 16 + g * 4 + j) * 2 + 0] = s;
          s_red[(wave * 16 + g * 4 + j) * 2 + 1] = q;
        }
        ssum[j] = 0.f; ssq[j] = 0.f;
      }
    }
  };

  // statistics of a complete column: rows c and c + 8 are the two x of channel c; entry 0 of the column's block is
  // real, the other stat_epc - 1 are zero (vx_instnorm_finalize sums vx_conv3d_k3_tiles_for entries per sample)
  auto flush_col = [&](int ci) {
    int n, ty, tx;
    col_of(ci, n, ty, tx);
    const int ntile = cps * ka.stat_epc;
    float* dst = a.stats_partial + (((size_t)n * ntile + (size_t)(ty * ka.tiles_x + tx) * ka.stat_epc) * 8) * 2;
    if (tid < 8) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        s += s_red[(w * 16 + tid) * 2 + 0] + s_red[(w * 16 + tid + 8) * 2 + 0];
        q += s_red[(w * 16 + tid) * 2 + 1] + s_red[(w * 16 + tid + 8) * 2 + 1];
      }
      dst[tid * 2 + 0] = s;
      dst[tid * 2 + 1] = q;
    }
    for (int i = 16 + tid; i < ka.stat_epc * 16; i += 256) {
      if (tid < 256) dst[i] = 0.f;
    }
  };

#ifdef VX_CONV_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last, st_iters = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif

  // ---- pipeline: step S_j is visible after barrier j; iteration j commits S_{j+1}, loads S_{j+2}, computes the item
  // that S_j completes (step s >= 1 of a column completes item s - 1) ----
  Cur cx = {0, 0}, cc = {0, 0}, cp = {0, 0};   // visible / to commit / to prefetch
  prefetch(cp); advance(cp);
  commit(0);     advance(cc);                  // S_0 -> slot group 0
  prefetch(cp); advance(cp);
  int j = 0;                                   // S_j = cx;  its slot group is j % 3
  int grp_x = 0;
  int prev_ci = -1, prev_k = 0;                // waves 4..7: the item still to store
  int fl_ci = -1, fl_at = 0;                   // column whose statistics are complete after barrier fl_at
  while (cx.ci < ncol_wg) {
    __syncthreads();
    XP_STAMP(0);
    if (EPI == 0 && fl_ci >= 0 && j >= fl_at && !late) { flush_col(fl_ci); fl_ci = -1; }
    const bool comp = cx.s >= 1;
    const int item_k = cx.s - 1;
    int grp_c = grp_x + 1; if (grp_c == 3) grp_c = 0;               // group S_{j+1} goes into
    int rb = (grp_x == 0 ? 2 : grp_x - 1) * TZ + (TZ - 2);           // first plane of the item: group of S_{j-1}, plane TZ - 2
    // waves 0..3:  stage S_{j+1}, load S_{j+2}, multiply(j), store(j)
    // waves 4..7:  store(j-1), multiply(j), stage S_{j+1}, load S_{j+2}
    // (measured alternative, tools/stamp_s16.py: multiply FIRST in waves 0..3 and LAST in waves 4..7, so that the two
    // matrix phases never share the pipe -- 16 % slower: the loads, conversions and stores of an item are then issued by
    // four waves at a time instead of eight, and it is the memory side, not the matrix pipe, that paces these layers)
    if (late) {
      if (prev_ci >= 0) { epilogue(prev_ci, prev_k); prev_ci = -1; }
      XP_STAMP(2);
      if (comp) { multiply(rb); prev_ci = cx.ci; prev_k = item_k; }
      XP_STAMP(1);
      XP_WAIT_LOADS();
      XP_STAMP(3);
      if (cc.ci < ncol_wg) commit(grp_c);
      XP_STAMP(4);
      prefetch(cp);
      XP_STAMP(5);
    } else {
      XP_WAIT_LOADS();
      XP_STAMP(3);
      if (cc.ci < ncol_wg) commit(grp_c);
      XP_STAMP(4);
      prefetch(cp);
      XP_STAMP(5);
      if (comp) multiply(rb);
      XP_STAMP(1);
      if (comp) epilogue(cx.ci, item_k);
      XP_STAMP(2);
    }
    if (EPI == 0 && comp && item_k == KZ - 1) { fl_ci = cx.ci; fl_at = j + 2; }
    XP_STAMP(2);
#ifdef VX_CONV_STAMPS
    ++st_iters;
#endif
    advance(cx); advance(cc); advance(cp);
    grp_x = grp_c;
    ++j;
  }
  if (late && prev_ci >= 0) epilogue(prev_ci, prev_k);
  if (EPI == 0) {
    __syncthreads();
    if (fl_ci >= 0 && !late) flush_col(fl_ci);
  }
  if (EPI != 2 && a.range_flag) {
    float mx = rmax;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    // only magnitudes within a factor two of the fp16 limit are reported: thousands of atomics on one word per launch
    // cost tens of microseconds (they serialise at the memory side), and nobody needs the maximum of an ordinary tensor
    if (lane == 0 && !(mx < 32768.f)) atomicMax(a.range_flag, __float_as_uint(mx));
  }
#ifdef VX_CONV_STAMPS
  if (ka.stamps && lane == 0) {
    unsigned long long* d = ka.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
    for (int i = 0; i < 6; ++i) d[i] = st_sum[i];
    d[6] = st_iters;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
bool vx_conv3d_xp8_applies(int D, int H, int W, int Cin, int Cout) {
  if (vx_cfg().conv_fp32 != 0 || vx_cfg().s16_no_xp || vx_cfg().s16_no_xp8) return false;
  // D >= 8: a column has at least two items (the statistics hand-off between the wave halves needs the spacing)
  return Cout == 8 && (Cin == 8 || Cin == 16) && W % 32 == 0 && H % 8 == 0 && D % 4 == 0 && W >= 32 && H >= 8 && D >= 8;
}

template <int NCH, int EPI, int PRE, int UP>
static int launch_xp8(const Xp8Args& ka, hipStream_t s) {
  constexpr int TZ = 4 / NCH, NZ = 3 * TZ, ZP = 170;
  constexpr int PP = ((NZ * ZP + 15) / 16) * 16;
  constexpr size_t lds = (size_t)NCH * 2 * 2 * PP * 8 * 2 + (size_t)NCH * (9 * 2 * 32 * 8) * 2 + 8 * 16 * 2 * 4;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = conv3d_xp8_kernel<NCH, EPI, PRE, UP>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv3d_k3(xp8): hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr = true;
  }
  int gx = 256;
  if (vx_cfg().s16_per_cu > 0) gx = 256 * vx_cfg().s16_per_cu;
  if (gx > ka.ncols) gx = ka.ncols;
  static const char* kname = vx_kname("conv3d_xp8_kernel<%d,%d,%d,%d>", NCH, EPI, PRE, UP);
  vx_note_kernel(kname);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv3d_k3(xp8)");
  return VX_OK;
}

int vx_conv3d_k3_xp8w(const vx_conv3d_args& a, int stat_tiles, hipStream_t s);   // conv3d_xp8w.hip; 1 = not taken

int vx_conv3d_k3_xp8(const vx_conv3d_args& a, int stat_tiles, hipStream_t s) {
  Xp8Args ka;
  ka.a = a;
  const int nch = a.Cin / 8, tz = 4 / nch;
  ka.tiles_x = a.W / 32; ka.tiles_y = a.H / 8; ka.kz = a.D / tz;
  const int cps = ka.tiles_x * ka.tiles_y;
  ka.ncols = a.N * cps;
  ka.mcps = (unsigned)((1ull << 32) / (unsigned)cps) + 1u;
  ka.mtx = (unsigned)((1ull << 32) / (unsigned)ka.tiles_x) + 1u;
  ka.stat_epc = stat_tiles / cps;
  ka.no_xcd = vx_cfg().conv_no_xcd ? 1 : 0;
  ka.stamps = nullptr;
#ifdef VX_CONV_STAMPS
  if (const char* e = getenv("VX_CONV_DBG_PTR")) ka.stamps = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  if ((int64_t)a.N * cps >= (1ll << 31)) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8): too many columns");
  if (a.stats_partial && (stat_tiles % cps || a.act != VX_ACT_NONE || (a.drop_mode != VX_DROP_NONE && !a.pool_out) || a.head_out))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8): statistics go with a plain epilogue");
  if (!vx_cfg().s16_no_wspec) {   // producer / consumer waves
    const int rc = vx_conv3d_k3_xp8w(a, stat_tiles, s);
    if (rc != 1) return rc;
  }
  if (a.pool_out) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8): the pooled output needs the producer / consumer kernel (vx_config.s16_no_wspec = 0)");
  const int pre = a.in_mean ? 1 : 0;
  int epi;
  if (a.stats_partial) epi = 0;
  else if (a.head_out) epi = 2;
  else if (a.drop_mode == VX_DROP_HASH) epi = 1;
  else epi = 3;
  if (epi == 2 && !(a.act == VX_ACT_LRELU && a.drop_mode == VX_DROP_HASH)) return 1;   // head without dropout: general kernel
  if (epi == 1 && a.act != VX_ACT_LRELU) return 1;
  const int up = a.up_in ? 1 : 0;
#define XP8_CASE(N_, E_, P_, U_) if (nch == N_ && epi == E_ && pre == P_ && up == U_) return launch_xp8<N_, E_, P_, U_>(ka, s)
  XP8_CASE(1, 0, 0, 0); XP8_CASE(1, 0, 1, 0); XP8_CASE(1, 1, 0, 0); XP8_CASE(1, 2, 0, 0); XP8_CASE(1, 3, 0, 0); XP8_CASE(1, 3, 1, 0);
  XP8_CASE(2, 1, 0, 0); XP8_CASE(2, 1, 1, 0); XP8_CASE(2, 3, 0, 0); XP8_CASE(2, 3, 1, 0);
  XP8_CASE(2, 1, 0, 1); XP8_CASE(2, 1, 1, 1); XP8_CASE(2, 3, 0, 1); XP8_CASE(2, 3, 1, 1);
#undef XP8_CASE
  if (up) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8): no fused up-convolution for this epilogue (act=%d, drop_mode=%d, statistics=%d)",
                  a.act, a.drop_mode, a.stats_partial ? 1 : 0);
  return 1;   // not taken: the caller uses the general kernel
}
