// K1 for the FULL-RESOLUTION layers, wave-specialised form of round 2's tools/rejected/conv3d_xp8.hip (same arithmetic, same LDS image, same
// results bit for bit): the z-column walk with the rolling LDS window, but the two kinds of work of a step run in
// DIFFERENT waves instead of in sequence inside every wave:
//
//   * waves 0..7 (CONSUMERS) only multiply and store: the matrix loop of item j, then its epilogue (waves 0..3) or the
//     epilogue of item j - 1 first (waves 4..7), so that one wave of a SIMD stores while the other multiplies;
//   * waves 8..8+NPW-1 (PRODUCERS, one or two per SIMD) only stage: commit step S_{j+1} from registers into the LDS
//     image (split into fp16 hi / lo, the optional normalise-on-load prologue, the optional fused up-convolution), then
//     issue the loads of step S_{j+2}.
//
// In that first kernel a wave spends 40-55 % of an item in its matrix phase and the rest converting, issuing loads and
// storing -- with two waves per SIMD the matrix pipe idles whenever both are outside their matrix phase (item = 7300 ..
// 9400 cycles against 3456 matrix cycles per SIMD).  Here the staging instructions come from a third (and fourth)
// wave of the SIMD and issue in the shadow of the consumers' MFMAs; one barrier per item as before.  The roles live in
// two separate loops (same barrier count) so that the register allocation is the maximum, not the sum, of the two.
#include "s16_common.h"

struct Xp8wArgs {
  vx_conv3d_args a;
  int tiles_x, tiles_y, kz;   // columns per sample = tiles_x * tiles_y; kz = items per column
  int ncols;                  // columns in the launch (N * tiles_y * tiles_x)
  unsigned mcps, mtx;         // multiply-high magics: / (tiles_x * tiles_y), / tiles_x
  int rep;                    // > 1: column order (source volume, column, sample of the volume) -- see col_of
  unsigned mrep;              // multiply-high magic: / rep
  int stat_epc;               // statistics entries per column in stats_partial (entry 0 real, the rest zero)
  unsigned long long* stamps;
  int abl;                    // diagnostic build only: phase ablation bits (1 no multiply, 2 no epilogue, 4 no commit, 8 no loads)
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
#define XP_ABL ka.abl
#else
#define XP_ABL 0
#define XP_STAMP(i) do {} while (0)
#define XP_WAIT_LOADS() do {} while (0)
#endif

// NCH: 8-channel input chunks (1 or 2).  EPI: epilogue (0 bias + statistics + store, 1 LeakyReLU + hash dropout, 2 that + the
// fused 1x1x1 head, 3 run-time activation, 4 / 5 below).  PRE: normalise-on-load prologue (0 none, 1 InstanceNorm + LeakyReLU +
// dropout of the producing block, 2 pre-split input: mask only).  UP: fused up-convolution.  NPW: producer waves (4 or 8).
// EPI = 4 (one-chunk layers): EPI 0 (bias + statistics + store) PLUS the 2 x 2 x 2 max-pool of the block that follows the
// InstanceNorm (conv -> norm -> LeakyReLU -> Dropout -> MaxPool, unet3D_module.py:231-237, 303-310) -- the statistics are
// not known yet, but (x - mean) * rstd and LeakyReLU are monotone, so max over the window of drop(f(x)) =
// max(2 f(max over the KEPT x), 0 if any element was dropped): the epilogue writes the maximum of the kept RAW values and
// an any-dropped bit per channel; vx_pool_finish applies the statistics to 1/8 of the voxels later.  For this a consumer
// wave owns 2 rows x 2 planes (whole windows in one lane pair) instead of 4 rows of one plane.
// IN16 (opt-in reduced-storage mode, vx_config.storage16): the input tensor is stored as fp16 -- every value IS its own hi
// part, the lo plane is zero: the staging waves copy 8 bytes per piece without a split, the multiplying waves skip the
// lo-activation product (2 matrix instructions per product instead of 3).  Not the default: a tensor rounded to fp16
// (2^-11 relative) cannot meet the 1e-4 parity of the maps.
// IN16 >= 2 (round 6, vx_config.storage16 = 2, opt-in like IN16 = 1): ONE fp16 product per fp32 product -- the multiplying waves run
// only hi x hi (activations and weights rounded to fp16, fp32 accumulation), a third of the matrix instructions; IN16 = 3 combines it
// with the fp16 input.  What BASELINE config 2 calls "bf16": a throughput mode outside the 1e-4 parity bar, never the default.
template <int NCH, int EPI, int PRE, int UP, int NPW, int IN16 = 0>
__global__ __launch_bounds__((8 + NPW) * 64) void conv3d_xp8w_kernel(Xp8wArgs ka) {
  constexpr int IN16S = IN16 & 1;          // the input tensor is stored as fp16
  constexpr bool ONEP = IN16 >= 2;         // one product per fp32 product
  static_assert(IN16S == 0 || (NCH == 1 && PRE == 0 && UP == 0), "fp16 input: one-chunk layers without a prologue");
  static_assert(UP == 0 || NCH == 2, "the fused up-convolution produces chunk 0 of a two-chunk layer");
  // UP = 2 (round 4): the up-convolution COMPOSED into this conv's weights (vx_conv3d_args.up_fused).  The up half of the
  // input never exists, not even in LDS: the image holds the skip chunk plus a rolling window of COARSE planes (18 x 6
  // positions x 16 channels, 4 plane slots), and a consumer's K loop is 9 (kz, ky) steps over the skip chunk + 6 steps over
  // the 2 x 2 x 3 coarse taps of its output parity class (z & 1, y & 1) -- 45 instead of 54 matrix instructions per row,
  // and the staging waves copy one coarse plane per step instead of evaluating 24 small GEMMs.
  constexpr int NIMG = UP == 2 ? 1 : NCH;                 // image chunks of fine planes in LDS
  constexpr int CW = 18, CH = 6, CSL = 4;                 // coarse window: x positions, rows, plane slots
  constexpr int CPL = CSL * CH * CW;                      // positions per (octet, precision) plane of the coarse image
  constexpr int CSLOT_H = CH * CW * 8;                    // halves between two plane slots
  constexpr int CIMG_H = UP == 2 ? 2 * 2 * CPL * 8 : 0;   // halves of the coarse image ([octet][precision][slot][row][x][8])
  constexpr int WC_H = UP == 2 ? 4 * 6 * 2 * 64 * 8 : 0;  // halves of the composed weights ([class][step][hi | lo][lane][8])
  constexpr int NWCH = UP == 2 ? 1 : NCH;                 // conv weight chunks held in LDS (UP = 2: the skip chunk's)
  constexpr int NW = 8, NTH = (NW + NPW) * 64;
  constexpr int TZ = 4 / NCH;
  constexpr int R = TZ;                       // column tiles (y-rows of one z-plane) per consumer wave
  constexpr int WPZ = 8 / R;                  // consumer waves per z-plane
  constexpr int HX = 34, HXP = 17, HY = 10;
  constexpr int ZP = HY * HXP;                // positions per z-plane and x-parity
  constexpr int NZ = 3 * TZ;
  constexpr int PP = ((NZ * ZP + 15) / 16) * 16;
  constexpr int PREC_H = 2 * PP * 8;          // halves of one precision plane (both parities)
  constexpr int CHUNK_H = 2 * PREC_H;
  constexpr int W_H = 9 * 2 * 32 * 8;         // halves of one chunk's weights ([step 9][hi|lo][32 pieces][8])
  constexpr int GRP_H = TZ * ZP * 8;          // halves between two slot groups
  constexpr bool STATS = EPI == 0 || EPI == 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* s_img = reinterpret_cast<_Float16*>(smem_raw);
  _Float16* s_cimg = s_img + NIMG * CHUNK_H;
  _Float16* s_w = s_cimg + CIMG_H;
  _Float16* s_wc = s_w + NWCH * W_H;
  float* s_btab = reinterpret_cast<float*>(s_wc + WC_H);            // UP = 2: bias of the 27 border classes, [27][8]
  float* s_red = s_btab + (UP == 2 ? 27 * 8 : 0);

  const vx_conv3d_args& a = ka.a;
  // Fields a wave needs once per item or per column (the pooled output's pointers, the statistics buffer) are re-read from
  // the kernarg segment WHERE THEY ARE USED: left to the compiler every kernel argument is loaded at the top and stays in
  // SGPRs for the kernel's life -- the pooling instances held 106 and spilled 8-10 of them into VGPR lanes (v_readlane in
  // the item loop, scratch reserved for the lane register).  The empty asm hides the pointer's origin, so the loads cannot
  // be hoisted back.
  auto kernarg = [&]() {
    typedef const Xp8wArgs __attribute__((address_space(4))) * kp_t;      // constant address space: scalar loads
    kp_t p = (kp_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
  };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int cps = ka.tiles_x * ka.tiles_y;
  const int KZ = ka.kz;

  // ---- weights: resident for the kernel's life ----
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.w_packed) + (UP == 2 ? W_H / 8 : 0);   // UP = 2: chunk 1 = the skip half
    for (int i = tid; i < NWCH * W_H / 8; i += NTH) reinterpret_cast<f32x4*>(s_w)[i] = src[i];
    if constexpr (UP == 2) {
      const f32x4* csrc = reinterpret_cast<const f32x4*>(a.up_fused);
      for (int i = tid; i < WC_H / 8; i += NTH) reinterpret_cast<f32x4*>(s_wc)[i] = csrc[i];
      for (int i = tid; i < 27 * 8; i += NTH) s_btab[i] = a.up_fused[WC_H / 2 + i];
      // the coarse window starts as zeros: the first item of the workgroup reads plane -1 from the slot "before" step 0
      // (later columns find the previous column's last step there: coarse plane D / 2, outside the volume = zeros)
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      for (int i = tid; i < CIMG_H / 8; i += NTH) reinterpret_cast<f32x4*>(s_cimg)[i] = z4;
    }
  }
  if constexpr (UP == 2) __syncthreads();      // (every wave, both roles) the zero fill before the first coarse plane lands

  // ---- the columns of this workgroup ----
  int vb = blockIdx.x;
  const int G = (int)gridDim.x;
  if ((G & 7) == 0) vb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);   // one XCD: neighbouring columns
  const int ncol_wg = vb < ka.ncols ? (ka.ncols - vb + G - 1) / G : 0;

  struct Cur { int ci, s; };      // column number of this workgroup, step within the column (0 .. KZ)
  auto advance = [&](Cur& c) { if (++c.s > KZ) { c.s = 0; ++c.ci; } };
  // Column order.  Plain: (sample, column).  With in_repeat = T samples reading ONE raw tensor (MC-dropout: contr_1_2 over the
  // once-per-volume first-layer tensor) the T samples of a source volume's column are NEIGHBOURS in the order -- (volume, column,
  // sample) -- so that they run at the same time on workgroups of one XCD and the column is fetched from HBM once: in the plain
  // order the ten samples of a volume sat on five XCDs and the shared 0.27 GB tensor was fetched 1.31 GB worth (PMC, rounds 3-5).
  auto col_of = [&](int ci, int& n, int& ty, int& tx) {
    unsigned col = (unsigned)(vb + ci * G);
    unsigned t_rep = 0;
    if (ka.rep > 1) {
      const unsigned qr = __umulhi(col, ka.mrep);
      t_rep = col - qr * (unsigned)ka.rep;
      col = qr;
    }
    const unsigned q = cps == 1 ? col : __umulhi(col, ka.mcps);
    n = ka.rep > 1 ? (int)(q * (unsigned)ka.rep + t_rep) : (int)q;
    const unsigned rem = col - q * (unsigned)cps;
    const unsigned q2 = ka.tiles_x == 1 ? rem : __umulhi(rem, ka.mtx);
    ty = (int)q2;
    tx = (int)(rem - q2 * (unsigned)ka.tiles_x);
  };

  // dropout seeds: the by-value seed + the optional device word (hipGraph replay), read ONCE -- a load inside the item
  // loops would wait for every load / store in flight (s_waitcnt vmcnt(0)) before the word can be used
  const uint32_t seed_in = PRE != 0 ? vx_seed_of(a, a.in_drop_seed) : 0u;
  const uint32_t seed_out = (EPI == 1 || EPI == 2 || EPI == 4) ? vx_seed_of(a, a.drop_seed) : 0u;
  float rmax = 0.f;   // largest |value| this wave stored or produced (range guard of the split-fp16 consumers)
#ifdef VX_CONV_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last, st_iters = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif

  // pipeline (both roles): step S_j is visible after barrier j; in iteration j the producers commit S_{j+1} and load
  // S_{j+2}, the consumers compute the item that S_j completes (step s >= 1 of a column completes item s - 1)
  if (wave >= NW) {
    // =============================================== PRODUCER ===============================================
    // Round 3: the staging works on whole ROWS.  The stamps of the round-2 kernel (tools/stamp_s16.py) showed the
    // producers, not the consumers, on the critical path of the two largest layers (7 000 of 10 000 cycles per item in
    // `convert`, the consumers parked 40 % at the barrier): ~65 vector instructions per 16-byte piece -- a hash per lane,
    // LDS / element offsets recomputed from (plane, row, x) in every iteration, three instructions per dropout select.
    // A wave-iteration now is one row of the step: its 32 interior voxels x 2 channel quads are exactly 64 pieces, so
    //   * the global row offset, the LDS row offset and the row's validity (outside the volume in y / z) are SCALARS
    //     (soffset / an immediate / a branch), the per-lane part of every address is a constant of the lane;
    //   * ONE hash round per step serves eight rows: lane (i, j) computes keep-word j of row i, the row's lanes fetch
    //     theirs with ds_bpermute (1 + 1 instructions per piece instead of 13);
    //   * dropout = the keep bit ANDed into the piece's SCALE (2 instructions per element);
    //   * rows outside the volume issue no load and commit zeros without touching the vector ALU.
    // The 2 halo voxels per row (x = -1, x = 32) are gathered into one or two extra iterations of the old per-lane form.
    const int pw = wave - NW;
    constexpr int NCS = UP ? 1 : NCH;             // chunks staged from memory (UP: chunk 0 is computed)
    constexpr int RU = TZ * HY;                   // rows of a step, per chunk
    constexpr int NU = NCS * RU;                  // row units of a step; unit u = pw + NPW * i belongs to this wave
    constexpr int RPW = (NU + NPW - 1) / NPW;
    constexpr int HR = (RPW + 7) / 8;             // hash rounds per step (eight rows each)
    constexpr int NH = NU * 4;                    // halo pieces of a step: per row 2 sides x 2 channel quads
    constexpr int HIT = (NH + 63) / 64;           // halo iterations, one each for waves NPW - 1, NPW - 2, ...
    static_assert(HIT <= NPW && RPW <= 16, "producer row units");
    const int xb = a.in_xblk;
    const int voxf = xb ? 16 : a.in_pitch;              // floats per voxel step along x (concat: two halves of 8)
    const int rowf = a.W * voxf;
    const int biasf = ((TZ - 1) * a.H + 1) * rowf + 4 * voxf;
    const int qq = lane & 1;                            // channel quad of this lane (interior and halo pieces alike)
    auto xpart = [&](int dx, int lch) {                 // float offset of (voxel dx, quad qq) of chunk lch within its row
      if (xb) {
        const int blk = dx >= 0 ? dx / xb : -((-dx + xb - 1) / xb);
        return (blk * 2 + lch) * xb * 8 + (dx - blk * xb) * 8 + qq * 4;
      }
      return dx * a.in_pitch + (UP ? 0 : lch * 8) + qq * 4;   // UP: `in` is the skip tensor alone
    };
    // ---- interior pieces: lane = (x parity half, voxel pair, quad); within a row the even hx come first (lanes 0..31),
    // then the odd ones: 16 consecutive lanes write 128 contiguous bytes of ONE parity plane (natural order: 2-way conflicts)
    const int l_hx = (lane >> 5) ? 1 + (lane & 30) : 2 + (lane & 30);
    const int l_dx = l_hx - 1;
    constexpr int ISZ = IN16S ? 2 : 4;                                      // bytes per input element
    const unsigned l_voff = (unsigned)((xpart(l_dx, 0) + biasf) * ISZ);    // chunk term rides in the row's scalar offset
    const int l_lds = (l_hx & 1) * PP * 8 + (l_hx >> 1) * 8 + qq * 4;      // halves
    const int l_bp = 4 * (l_dx >> 2);                                      // ds_bpermute address of this lane's keep-word within its row
    const unsigned l_sh = (unsigned)((l_dx & 3) * 8 + qq * 4);             // first of its four bits in that word
    // Row units of this wave.  LIN (all but the fused up-convolution at 8 producer waves): the wave owns RPW CONSECUTIVE
    // rows of ONE z-plane of one chunk (RPW divides the plane's 10 rows), so row i is (first row) + i: a running scalar
    // add on the global side, an immediate on the LDS side, no per-row tables in scalar registers.
    constexpr bool LIN = NU % NPW == 0 && HY % RPW == 0;
    auto unit_of = [&](int i) { return LIN ? pw * RPW + i : pw + NPW * i; };
    auto unit_geo = [&](int u, int& lch, int& r, int& dzy, int& dy, int& pz) {
      const int scn = u / RU;
      r = u - scn * RU;
      pz = r / HY;
      const int hy = r - pz * HY;
      dy = hy - 1;
      dzy = (pz - (TZ - 1)) * a.H + dy;                // row offset in rows of the tensor
      lch = UP ? 1 : scn;
    };
    int u_soff[LIN ? 1 : RPW], u_lds[LIN ? 1 : RPW];   // LIN: entry 0 = row 0 of the wave
    unsigned um_valid = 0, um_ylo = 0, um_yhi = 0, um_zf = 0, um_zl = 0, um_pre = 0;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      int lch, r, dzy, dy, pz;
      unit_geo(unit_of(i), lch, r, dzy, dy, pz);
      if (!LIN || i == 0) {
        u_soff[LIN ? 0 : i] = (dzy * rowf + (xpart(0, lch) - xpart(0, 0))) * ISZ;
        u_lds[LIN ? 0 : i] = (UP == 2 ? 0 : lch) * CHUNK_H + r * HXP * 8;
      }
      if (unit_of(i) < NU) um_valid |= 1u << i;
      if (dy < 0) um_ylo |= 1u << i;
      if (dy >= 8) um_yhi |= 1u << i;
      if (pz < TZ - 1) um_zf |= 1u << i;               // step 0 of a column: planes -(TZ-1) .. -1 do not exist
      else um_zl |= 1u << i;                           // step KZ: plane D does not exist
      if (PRE != 0 && (NCH == 1 || lch == 1)) um_pre |= 1u << i;
    }
    const int row_soff_step = rowf * ISZ;
    // keep-word this lane computes in hash round rd: word (lane & 7) of row slot 8 rd + (lane >> 3)
    int l_hw[PRE ? HR : 1];
    if constexpr (PRE != 0) {
#pragma unroll
      for (int rd = 0; rd < HR; ++rd) {
        int lch, r, dzy, dy, pz;
        unit_geo(unit_of(8 * rd + (lane >> 3)), lch, r, dzy, dy, pz);
        l_hw[rd] = dzy * (a.W / 4) + (lane & 7);       // (row's first element) / 32 + word
      }
    }
    // ---- halo pieces: iteration hk of the step (this wave's only one, if any): lane -> (chunk, row, side, quad)
    const int hk = NPW - 1 - pw;
    const bool has_halo = hk < HIT;
    unsigned h_voff = 0, h_erel = 0, h_flags = 1u;     // flags: 1 never a piece, 2 / 4 side x = -1 / 32, 8 / 16 y = -1 / 8,
    int h_lds = 0;                                     //        32 / 64 plane before the last / the last of the step, 128 prologue
    if (has_halo) {
      const int hp = lane + 64 * hk;
      const int rem = hp % (RU * 4), side = (rem >> 1) & 1;
      int lch, r, dzy, dy, pz;
      unit_geo(hp >> 2, lch, r, dzy, dy, pz);
      const int dx = side ? 32 : -1, hx = dx + 1;
      h_voff = (unsigned)((dzy * rowf + xpart(dx, lch) + biasf) * ISZ);
      h_lds = (UP == 2 ? 0 : lch) * CHUNK_H + (hx & 1) * PP * 8 + (r * HXP + (hx >> 1)) * 8 + qq * 4;
      h_erel = (unsigned)((dzy * a.W + dx) * 8 + qq * 4);
      h_flags = (hp >= NH ? 1u : 0u) | (side ? 4u : 2u) | (dy < 0 ? 8u : 0u) | (dy >= 8 ? 16u : 0u) | (pz < TZ - 1 ? 32u : 64u) |
                ((PRE != 0 && (NCH == 1 || lch == 1)) ? 128u : 0u);
    }
    const size_t in_sample = (size_t)a.D * a.H * rowf;
    const int in_rep = a.in_repeat > 1 ? a.in_repeat : 1;

    // ---- fused up-convolution: this wave's class and its column tiles (fixed for the kernel's life) ----
    // A class is (plane u_pz of the step, y-parity u_ay): 6 column tiles of 16 coarse voxels (c -> (Yi, Xi) = (c / 18,
    // c % 18) of the 5 x 18 coarse positions whose fine row of parity u_ay lies in the staged 10 x 34 window).  Rows of
    // the product are (dx, co): lane (m, g) ends with channels 4 (g & 1) .. + 3 of the fine voxel x = 2 X + (g >> 1) --
    // the piece layout of the staged loads.
    constexpr int UT = UP == 1 ? 24 / NPW : 1;          // column tiles per producer wave and step
    const int u_cls = (pw * UT) / 6, u_t0 = (pw * UT) % 6;
    const int u_pz = u_cls >> 1, u_ay = u_cls & 1;
    const int Hc = a.H >> 1, Wc = a.W >> 1;
    const int urow = Wc * a.up_pitch;
    const int ubiasf = (Hc + 1) * urow + a.up_pitch;
    unsigned u_voff[UP == 1 ? UT : 1];
    int u_ldst[UP == 1 ? UT : 1];
    unsigned ub_always = 0, ub_xlo = 0, ub_xhi = 0, ub_ylo = 0, ub_yhi = 0, u_nowrite = 0;
    f16x4 u_ah = {0, 0, 0, 0}, u_al = {0, 0, 0, 0};
    f32x4 ubias4 = {0.f, 0.f, 0.f, 0.f};
    // UP = 2: this lane's pieces of a coarse plane (6 rows x 18 x 4 channel quads = 432 pieces of 16 bytes per step)
    constexpr int CPT = UP == 2 ? (CH * CW * 4 + NPW * 64 - 1) / (NPW * 64) : 1;
    unsigned c_voff[CPT];
    int c_lds[CPT];
    unsigned cb_always = 0, cb_xlo = 0, cb_xhi = 0, cb_ylo = 0, cb_yhi = 0;
    if constexpr (UP == 2) {
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int pc = pw * 64 + lane + i * NPW * 64;
        const int q = pc & 3, xi = (pc >> 2) % CW, ry = (pc >> 2) / CW;
        c_voff[i] = (unsigned)((((ry - 1) * Wc + (xi - 1)) * a.up_pitch + q * 4 + ubiasf) * 4);
        c_lds[i] = (((q >> 1) * 2) * CPL + ry * CW + xi) * 8 + (q & 1) * 4;
        if (pc >= CH * CW * 4) cb_always |= 1u << i;
        if (xi == 0) cb_xlo |= 1u << i;
        if (xi == CW - 1) cb_xhi |= 1u << i;
        if (ry == 0) cb_ylo |= 1u << i;
        if (ry == CH - 1) cb_yhi |= 1u << i;
      }
    }
    if constexpr (UP == 1) {
#pragma unroll
      for (int i = 0; i < UT; ++i) {
        const int c = 16 * (u_t0 + i) + m;
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
      ubias4 = *reinterpret_cast<const f32x4*>(a.up_b + (g & 1) * 4);
    }
    const size_t up_sample = (size_t)(a.D >> 1) * Hc * urow;

    // ---- register staging: the loads of one step (and what its commit needs to know) ----
    f32x4 ibuf[IN16S ? 1 : RPW], hbuf = {0.f, 0.f, 0.f, 0.f};
    // fp16 input: 8 bytes per piece, kept as integer pairs (hipcc 7.2 narrows a 64-bit buffer load to ONE dword when its
    // halves travel through float lanes of a wider vector -- the second dword was garbage; tools/micro/load_b64_narrow.hip)
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 ibuf16[IN16S ? RPW : 1], hbuf16 = {0u, 0u};
    f32x4 ubuf[UP == 1 ? UT : 1];
    f32x4 cbuf[CPT];
    unsigned p_ubad = 0;
    f32x4 p_mean = {0.f, 0.f, 0.f, 0.f}, p_rstd = {1.f, 1.f, 1.f, 1.f};
    vx_dkey p_key = {0u, 0u};
    unsigned p_rowbad = 0xFFFFFFFFu, p_e0 = 0;   // bit i: row unit i of the staged step lies outside the volume
    bool p_hbad = true;                                      // this lane's halo piece of the staged step lies outside

    // per-COLUMN state of the staging (recomputed at step 0 of a column, KZ + 1 steps apart): sample, tile position, the
    // x / y parts of the validity masks and offsets, the buffer descriptors, the prologue's statistics and dropout key --
    // per step only the z terms are left (round 3: ~100 scalar instructions and two loads per step gone)
    bool cs_have = false;
    unsigned cs_bad = 0xFFFFFFFFu, cs_hb = 0x7Fu, cs_ub = 0xFFFFFFFFu, cs_e0 = 0;
    int cs_soff = 0;
    unsigned cs_usoff = 0;
    __amdgpu_buffer_rsrc_t cs_srd = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t cs_usrd = cs_srd;
    auto column_state = [&](int ci) {
      const bool have = ci < ncol_wg;
      int n = 0, ty = 0, tx = 0;
      if (have) col_of(ci, n, ty, tx);
      cs_have = have;
      unsigned bad = ~um_valid;
      if (ty == 0) bad |= um_ylo;
      if (ty == ka.tiles_y - 1) bad |= um_yhi;
      cs_bad = bad;
      unsigned hb = 1u;
      if (tx == 0) hb |= 2u;
      if (tx == ka.tiles_x - 1) hb |= 4u;
      if (ty == 0) hb |= 8u;
      if (ty == ka.tiles_y - 1) hb |= 16u;
      cs_hb = hb;
      const int nin = n / in_rep;
      cs_soff = ((ty * 8) * rowf + tx * 32 * voxf) * ISZ;
      cs_srd = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(reinterpret_cast<const char*>(a.in) + ((size_t)nin * in_sample - biasf) * ISZ), 0, VX_NUMREC, 0x00020000);
      if constexpr (UP != 0) {
        unsigned ub = UP == 2 ? cb_always : ub_always;
        if (tx == 0) ub |= UP == 2 ? cb_xlo : ub_xlo;
        if (tx == ka.tiles_x - 1) ub |= UP == 2 ? cb_xhi : ub_xhi;
        if (ty == 0) ub |= UP == 2 ? cb_ylo : ub_ylo;
        if (ty == ka.tiles_y - 1) ub |= UP == 2 ? cb_yhi : ub_yhi;
        cs_ub = ub;
        cs_usoff = (unsigned)(((ty * 4) * urow + tx * 16 * a.up_pitch) * 4);
        cs_usrd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.up_in + (size_t)n * up_sample - ubiasf), 0, VX_NUMREC, 0x00020000);
      }
      if constexpr (PRE == 1) {
        // (n = 0 when the workgroup has run out of columns: a valid address, no branch around the loads)
        p_mean = *reinterpret_cast<const f32x4*>(a.in_mean + (size_t)nin * 8 + qq * 4);
        p_rstd = *reinterpret_cast<const f32x4*>(a.in_rstd + (size_t)nin * 8 + qq * 4);
      }
      if constexpr (PRE != 0) {
        cs_e0 = (unsigned)((ty * 8) * a.W + tx * 32) * 8u;
        p_key = vx_drop_key(seed_in, a.in_drop_layer, (uint32_t)n);
      }
    };

    auto prefetch = [&](const Cur& c) {
      if (c.s == 0) column_state(c.ci);            // (a wave-uniform branch BEFORE the loads, nothing in flight at the join)
      const bool have = cs_have;
      unsigned bad = cs_bad;
      if (c.s == 0) bad |= um_zf;
      if (c.s == KZ) bad |= um_zl;
      if (!have) bad = 0xFFFFFFFFu;
      const int soff = cs_soff + ((TZ * c.s) * a.H) * rowf * ISZ;
      const __amdgpu_buffer_rsrc_t srd = cs_srd;
      // NO branch may enclose a load: behind a join the compiler's wait-count bookkeeping gives up and waits for EVERY load
      // in flight (s_waitcnt vmcnt(0) in the middle of this function: the whole memory latency, every step -- measured
      // 2 400 cycles).  A row outside the volume reads through an out-of-range offset instead (zeros, no memory access).
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int so = soff + (LIN ? u_soff[0] + i * row_soff_step : u_soff[LIN ? 0 : i]);
        const bool rb = (bad >> i) & 1u;
        if constexpr (IN16S != 0) {
          ibuf16[i] = __builtin_amdgcn_raw_buffer_load_b64(srd, (int)(rb ? VX_OOB : l_voff), rb ? 0 : so, 0);
        } else {
          ibuf[IN16S ? 0 : i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(rb ? VX_OOB : l_voff), rb ? 0 : so, 0));
        }
      }
      p_rowbad = bad;
      {
        unsigned hb = cs_hb;
        if (c.s == 0) hb |= 32u;
        if (c.s == KZ) hb |= 64u;
        if (!have) hb = 0x7Fu;
        const bool lbad = (h_flags & hb) != 0u;      // waves without a halo iteration: flag 1 in every lane
        if constexpr (IN16S != 0) {
          hbuf16 = __builtin_amdgcn_raw_buffer_load_b64(srd, (int)(lbad ? VX_OOB : h_voff), soff, 0);
        } else {
          hbuf = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(lbad ? VX_OOB : h_voff), soff, 0));
        }
        p_hbad = lbad;
      }
      if constexpr (UP == 1) {
        unsigned ub = cs_ub;
        if (!have || (c.s == 0 && u_pz == 0) || (c.s == KZ && u_pz == 1)) ub = 0xFFFFFFFFu;
        const unsigned usoff = cs_usoff + (unsigned)((c.s * Hc) * urow * 4);
#pragma unroll
        for (int i = 0; i < UT; ++i) {
          const unsigned vo = ((ub >> i) & 1u) ? VX_OOB : u_voff[i];
          ubuf[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cs_usrd, (int)vo, (int)usoff, 0));
        }
        p_ubad = ub;
      }
      if constexpr (UP == 2) {
        // step s of a column carries coarse plane s (item s - 1 = fine planes 2 s - 2, 2 s - 1 needs coarse s - 2 .. s);
        // plane D / 2 (step KZ) lies outside: zeros -- which the NEXT column's first item reads as its plane -1
        unsigned ub = cs_ub;
        if (!have || c.s == KZ) ub = 0xFFFFFFFFu;
        const unsigned usoff = cs_usoff + (unsigned)((c.s * Hc) * urow * 4);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
          const unsigned vo = ((ub >> i) & 1u) ? VX_OOB : c_voff[i];
          cbuf[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cs_usrd, (int)vo, (int)usoff, 0));
        }
      }
      if constexpr (PRE != 0) p_e0 = cs_e0 + (unsigned)(((TZ * c.s) * a.H) * a.W) * 8u;
    };

    // the producing block's InstanceNorm + LeakyReLU + Dropout on one piece: (x - mean) * scale with the keep bit ANDed into
    // the scale (dropout's factor 2 rides in it: 2 lrelu(t) = lrelu(2 t)); (x - mean) first: no cancellation against a
    // rounded mean * rstd
    auto pre_piece = [&](f32x4 v, const f32x4 sc, uint32_t bits) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int keep = __builtin_amdgcn_sbfe((int)bits, j, 1);          // all ones / all zeros
        const float scj = __int_as_float(__float_as_int(sc[j]) & keep);
        // plain single instructions (vx_*1): packed fp32 beside the consumers' MFMAs costs three times as much
        const float t = vx_mul1(vx_sub1(v[j], p_mean[j]), scj);
        v[j] = vx_max1(t, vx_mul1(t, 0.01f));
      }
      return v;
    };

    // plain-instruction split where the prologue runs (measured: -5 % on both prologue layers); the instances without a
    // prologue keep the packed form (expand_1_2 + head lost 20 % with the plain one: its consumers are the critical path)
    auto split4 = [&](const f32x4 v, f16x4& hi, f16x4& lo) {
      if constexpr (PRE != 0) vx_split4_s(v, hi, lo);
      else vx_split4(v, hi, lo);
    };
    // PRE == 2: the piece is [hi0 hi1 | hi2 hi3 | lo0 lo1 | lo2 lo3] already; keep bit j -> a 16-bit mask on element j
    auto masked_piece = [&](const f32x4 v, uint32_t bits, f16x4& hi, f16x4& lo) {
      const u32x4 d = __builtin_bit_cast(u32x4, v);
      const uint32_t k0 = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 0, 1), k1 = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 1, 1);
      const uint32_t k2 = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 2, 1), k3 = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 3, 1);
      const uint32_t m01 = __builtin_amdgcn_perm(k1, k0, 0x05040100u), m23 = __builtin_amdgcn_perm(k3, k2, 0x05040100u);
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      hi = __builtin_bit_cast(f16x4, (u32x2){d[0] & m01, d[1] & m23});
      lo = __builtin_bit_cast(f16x4, (u32x2){d[2] & m01, d[3] & m23});
    };

    auto commit = [&](int grp, int jstep) {      // jstep: running number of the step (all columns of the workgroup)
      const int gofs = grp * GRP_H;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f};
      const bool hashed = PRE != 0 && a.in_drop_mode == VX_DROP_HASH;
      uint32_t hw[PRE ? HR : 1];
      if constexpr (PRE == 1) {
        const float two = hashed ? 2.f : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[j] = vx_mul1(p_rstd[j], two);
      }
      if constexpr (PRE != 0) {
        if (hashed) {
          // keep-words of this wave's rows, one hash round per eight rows
#pragma unroll
          for (int rd = 0; rd < HR; ++rd) hw[rd] = vx_drop_word(p_key, (uint32_t)((int)(p_e0 >> 5) + l_hw[rd]));
        }
      }
      // rows: straight-line code (the compiler interleaves the rows' dependent chains; a row outside the volume arrives as
      // zeros and its keep bits are forced to zero, so it commits the zero padding of the NORMALISED tensor)
      const bool wave_pre = PRE != 0 && ((NCH == 1 || UP != 0) ? true : (um_pre & 1u) != 0u);   // LIN: one chunk per wave
      uint32_t wrow[PRE ? RPW : 1];
      if constexpr (PRE != 0) {
        if (hashed && wave_pre) {
#pragma unroll
          for (int i = 0; i < RPW; ++i)
            wrow[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(l_bp + 32 * (i & 7), (int)hw[i >> 3]);
        }
      }
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        if (!LIN && !((um_valid >> i) & 1u)) continue;       // (only the last unit of a wave can be missing)
        _Float16* dst = s_img + gofs + l_lds + (LIN ? u_lds[0] + i * (HXP * 8) : u_lds[LIN ? 0 : i]);
        if constexpr (IN16S != 0) {     // the stored halves ARE the hi plane; nobody reads a lo plane in this mode
          *reinterpret_cast<f16x4*>(dst) = __builtin_bit_cast(f16x4, ibuf16[i]);
          continue;
        }
        f32x4 v = ibuf[IN16S ? 0 : i];
        f16x4 hi, lo;
        if constexpr (PRE == 2) {
          // pre-split input (vx_prenorm_split): only this sample's keep bits are left to apply (a row outside the volume
          // arrives as zeros)
          masked_piece(v, hashed ? (wrow[i] >> l_sh) : 0xFu, hi, lo);
        } else {
          if constexpr (PRE == 1) {
            if (wave_pre && !(XP_ABL & 32)) {
              uint32_t bits = hashed ? (wrow[i] >> l_sh) : 0xFu;                 // pre_piece looks at bits 0..3 only
              bits &= ((p_rowbad >> i) & 1u) ? 0u : 0xFu;
              v = pre_piece(v, sc, bits);
            }
          }
          if (XP_ABL & 32) { hi = __builtin_bit_cast(f16x4, (f32x2){v[0], v[1]}); lo = __builtin_bit_cast(f16x4, (f32x2){v[2], v[3]}); }
          else split4(v, hi, lo);
        }
        if (!(XP_ABL & 16)) {
          *reinterpret_cast<f16x4*>(dst) = hi;
          *reinterpret_cast<f16x4*>(dst + PREC_H) = lo;
        } else {
          asm volatile("" :: "v"(hi), "v"(lo));
        }
      }
      if (has_halo && !(h_flags & 1u)) {
        f32x4 v = hbuf;                      // zeros where the piece lies outside the volume (out-of-range load)
        f16x4 hi, lo;
        if constexpr (IN16S != 0) {
          hi = __builtin_bit_cast(f16x4, hbuf16);
          lo = (f16x4){0, 0, 0, 0};
        } else if constexpr (PRE == 2) {
          masked_piece(v, hashed ? vx_drop_bits4(p_key, p_e0 + h_erel) : 0xFu, hi, lo);
        } else {
          if constexpr (PRE == 1) {
            if (h_flags & 128u) {
              uint32_t bits = 0xFu;
              if (hashed) bits = vx_drop_bits4(p_key, p_e0 + h_erel);
              if (p_hbad) bits = 0u;           // zero padding belongs to the normalised tensor
              v = pre_piece(v, sc, bits);
            }
          }
          split4(v, hi, lo);
        }
        *reinterpret_cast<f16x4*>(s_img + gofs + h_lds) = hi;
        *reinterpret_cast<f16x4*>(s_img + gofs + h_lds + PREC_H) = lo;
      }
      if constexpr (UP == 2) {
        // the step's coarse plane into slot jstep % 4 of the coarse window, as fp16 pairs (pre-split by the producing conv's
        // epilogue, vx_conv3d_args.up_split, or split here: one piece per lane and step)
        _Float16* cdst = s_cimg + (jstep & (CSL - 1)) * CSLOT_H;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
          f16x4 hi, lo;
          if (a.up_split) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x4 d = __builtin_bit_cast(u32x4, cbuf[i]);
            hi = __builtin_bit_cast(f16x4, (u32x2){d[0], d[1]});
            lo = __builtin_bit_cast(f16x4, (u32x2){d[2], d[3]});
          } else {
            split4(cbuf[i], hi, lo);
          }
          if (!((cb_always >> i) & 1u)) {
            *reinterpret_cast<f16x4*>(cdst + c_lds[i]) = hi;
            *reinterpret_cast<f16x4*>(cdst + c_lds[i] + CPL * 8) = lo;
          }
        }
      }
      if constexpr (UP == 1) {
        // the up half of the step: ConvTranspose3d(k = 2, s = 2) of the coarse voxels just loaded, three split products
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f16x4 ubh[UT], ubl[UT];
        if (a.up_split) {          // the producer of the coarse tensor stored the pairs (vx_conv3d_args.out_split)
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int i = 0; i < UT; ++i) {
            const u32x4 d = __builtin_bit_cast(u32x4, ubuf[i]);
            ubh[i] = __builtin_bit_cast(f16x4, (u32x2){d[0], d[1]});
            ubl[i] = __builtin_bit_cast(f16x4, (u32x2){d[2], d[3]});
          }
        } else {
#pragma unroll
          for (int i = 0; i < UT; ++i) split4(ubuf[i], ubh[i], ubl[i]);
        }
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
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaf(dx[j], 1.0f / 2048.f, d[j]);   // NOT inline asm: the first reader of a matrix
                                                                                 // result needs the compiler's wait states
          if ((p_ubad >> i) & 1u) v = zero;               // outside the volume: the conv's zero padding, not the bias
          rmax = vx_max3abs(vx_max3abs(rmax, v[0], v[1]), v[2], v[3]);
          f16x4 hi, lo;
          split4(v, hi, lo);
          if (!((u_nowrite >> i) & 1u)) {
            *reinterpret_cast<f16x4*>(s_img + gofs + u_ldst[i]) = hi;
            *reinterpret_cast<f16x4*>(s_img + gofs + u_ldst[i] + PREC_H) = lo;
          }
        }
      }
    };

    Cur cx = {0, 0}, cc = {0, 0}, cp = {0, 0};   // visible / to commit / to prefetch
    prefetch(cp); advance(cp);
    commit(0, 0);  advance(cc);                  // S_0 -> slot group 0
    prefetch(cp); advance(cp);
    int grp_x = 0;
    int jrun = 0;                                // S_jrun is the visible step
    while (cx.ci < ncol_wg) {
      __syncthreads();
      XP_STAMP(0);
      int grp_c = grp_x + 1; if (grp_c == 3) grp_c = 0;               // group S_{j+1} goes into
      XP_WAIT_LOADS();
      XP_STAMP(3);
      if (cc.ci < ncol_wg && !(XP_ABL & 4)) commit(grp_c, jrun + 1);
      XP_STAMP(4);
      if (!(XP_ABL & 8)) prefetch(cp);
      XP_STAMP(5);
#ifdef VX_CONV_STAMPS
      ++st_iters;
#endif
      advance(cx); advance(cc); advance(cp);
      grp_x = grp_c;
      ++jrun;
    }
    if (STATS) __syncthreads();
  } else {
    // =============================================== CONSUMER ===============================================
    constexpr bool POOLM = EPI == 4;            // tile r of this wave = plane lz + (r >> 1), row ly0 + (r & 1)
    static_assert(!POOLM || (NCH == 1 && R == 4), "pooling epilogue: one-chunk layers");
    // UP = 2 (round 6): a wave's two rows are ly0 and ly0 + 2 -- the SAME y-parity, i.e. one composed-weight class for both, so a
    // coarse K-step reads its class' weights once (2 + 4 reads instead of 4 + 4 for 6 matrix instructions; the skip planes then
    // need 5 input rows instead of 4: 75 reads per item instead of 81)
    constexpr int RS = UP == 2 ? 2 : 1;                                   // rows between the wave's column tiles
    const int wz_ = wave % WPZ;
    const int lz = POOLM ? (wave >> 2) * 2 : wave / WPZ, ly0 = POOLM ? (wave & 3) * 2 : (UP == 2 ? (wz_ >> 1) * 4 + (wz_ & 1) : wz_ * R);
    const bool late = wave >= NW / 2;
    // ---- compute-phase constants ----
    // B fragment of (kz, row j): s_img[chunk][prec][parity g & 1][(slot * HY + ly0 + j) * HXP + m + (g >> 1)]
    const int bfrag0 = (g & 1) * PP * 8 + ((ly0 * HXP) + m + (g >> 1)) * 8;
    const int wslot = (((m & 7) >> 2) * 4 + ((g - (m >> 3)) & 3)) * 4 + (m & 3);   // conv3d_s16.hip: [co >> 2][kx][co & 3]

    // ---- epilogue constants: this lane stores voxel x = 2 m + (g >> 1) of row ly0 + r, channels 4 (g & 1) .. + 3 ----
    const int lx = 2 * m + (g >> 1), oc = (g & 1) * 4;
    // ---- UP = 2: the composed coarse part.  K-step s, k-group g = coarse octet o = 4 s + g = ((czi 2 + cyi) 3 + cxi) 2 + oct:
    // coarse tap (plane czi, row cyi, x cxi) relative to the output pair's class origin, channels 8 oct .. + 7.  czi = s / 3
    // for every lane; the rest is this lane's offset into the coarse image (hi plane; row of the tile, plane slot on top)
    int cofs[UP == 2 ? 6 : 1];
    if constexpr (UP == 2) {
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const int o = 4 * s + g;
        const int oct = o & 1, cxi = (o >> 1) % 3, cyi = (o / 6) & 1;
        cofs[s] = ((oct * 2) * CPL + (((ly0 + 1) >> 1) + cyi) * CW + cxi + m) * 8;     // window row of fine row y: ((y + 1) >> 1) + cyi
      }
    }
    int m_ci = -1, m_ty = 0;           // UP = 2: column of the bias classes below
    int m_bofs = 0;                    //         byte offset of this lane's (x class, channel quad) within a table row
    unsigned ovoff[R], eoff[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int tz_ = lz + (POOLM ? (r >> 1) : 0), ty_ = ly0 + (POOLM ? (r & 1) : RS * r);
      const int ovox = (tz_ * a.H + ty_) * a.W + lx;
      if (a.out_xblk) {
        const int oxb = a.out_xblk;
        ovoff[r] = (unsigned)((((tz_ * a.H + ty_) * (2 * a.W * 8)) + ((lx / oxb) * 2 + a.out_half) * oxb * 8 + (lx % oxb) * 8 + oc) * 4);
      } else {
        ovoff[r] = (unsigned)((ovox * a.out_pitch + a.out_coff + oc) * (a.out_f16 ? 2 : 4));
      }
      eoff[r] = (unsigned)(ovox * 8 + oc);
    }
    // Round 4: ONE hash round per item serves the wave's R rows.  A row of 32 voxels x 8 channels is 8 keep-words (32 bits
    // each: 4 voxels); lane i computes word i & 7 of row i >> 3, a row's lanes fetch theirs with ds_bpermute (1 instruction
    // instead of a hash per row: the staging waves have shared theirs since round 3).  Word of (row r, lane) =
    // ((voxel of the row's first element) >> 2) + (m >> 1); bit offset within it (lx & 3) * 8 + oc.  Same bits as
    // vx_drop_bits4(key, element) per row; +0.3 % end to end (same box, twice).
    unsigned hword_l;
    {
      const int rr_ = (lane >> 3) < R ? (lane >> 3) : 0;
      const int tz_ = lz + (POOLM ? (rr_ >> 1) : 0), ty_ = ly0 + (POOLM ? (rr_ & 1) : RS * rr_);
      hword_l = (unsigned)((((tz_ * a.H + ty_) * a.W) >> 2) + (lane & 7));
    }
    const int hbp = 4 * (m >> 1);                                    // ds_bpermute byte address of this lane's word in row 0
    const unsigned hsh = (unsigned)((lx & 3) * 8 + oc);
    const int out_voxf = a.out_xblk ? 16 : a.out_pitch;
    const size_t out_sample = (size_t)a.D * a.H * a.W * out_voxf;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.bias + oc);
    const bool f_lrelu = EPI == 3 ? a.act == VX_ACT_LRELU : !STATS;
    const bool f_relu = EPI == 3 && a.act == VX_ACT_RELU;
    // fused head: this lane's 4 of the 8 weights of up to 4 classes; the bias rides in the g-even lane (conv3d_s16.hip)
    constexpr bool HEAD = EPI == 2 || EPI == 5;     // EPI 5: LeakyReLU + fused head WITHOUT dropout (deterministic ensemble members)
    constexpr int HC = HEAD ? 4 : 1;
    float hw4[HC][4], hb[HC];
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      hb[c] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) hw4[c][k] = 0.f;
      if (HEAD && c < a.head_C) {
        if (!(g & 1)) hb[c] = a.head_b[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) hw4[c][k] = a.head_w[c * 8 + oc + k];
      }
    }
    const size_t hnvox = (size_t)a.D * a.H * a.W;


    // UP = 2 (round 6): the hi fragments of the nine skip-chunk weight steps stay in registers for the kernel's life -- they do not
    // depend on the item, the instance has the registers (87 of the 128 its 16 waves allow) and its multiply loop sits at the LDS
    // array's limit (1.0 ds_read_b128 per matrix instruction): 9 of the 90 reads of an item gone
    f16x8 sAh[UP == 2 ? 9 : 1];
    if constexpr (UP == 2) {
#pragma unroll
      for (int q = 0; q < 9; ++q) sAh[q] = *reinterpret_cast<const f16x8*>(s_w + wslot * 8 + q * (2 * 32 * 8));
    }

    // ---- accumulators ----
    f32x4 acc[R], accx[R];
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    float pl_max[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // EPI 4: maximum of the kept raw values of the window
    uint32_t pl_any = 0;                                              //        bit j: an element of channel oc + j was dropped

    // the multiply phase of the item whose first plane sits in slot rb (slots rb .. rb + TZ + 1, modulo NZ)
    // (measured alternative: explicitly software-pipelined LDS reads -- weights one step ahead, the next group's image
    // rows requested as the products free their registers: 2.59 -> 2.52 ms on the two-chunk layer, 1-5 % SLOWER on the
    // one-chunk layers where the extra live rows spill; operands-in-registers rate of this instruction is 78 % of the
    // nominal peak and the plain loop already runs at 71 %)
    auto multiply = [&](int rb, int jstep, int ci, int k) {
      if constexpr (UP == 2) {
        // bias of the rows: b + sum over the 3x3x3 taps INSIDE the volume of W[up half] . up_b -- 27 border classes (z, y, x)
        if (ci != m_ci) {
          m_ci = ci;
          int n_, tx_;
          col_of(ci, n_, m_ty, tx_);
          const int xc = (tx_ == 0 && lx == 0) ? 0 : ((tx_ == ka.tiles_x - 1 && lx == 31) ? 2 : 1);
          m_bofs = (xc * 8 + oc) * 4;
        }
        const int z = k * TZ + lz;
        const int zc = z == 0 ? 0 : (z == a.D - 1 ? 2 : 1);
        const int pz = lz & 1;                                        // (items start at even z)
        // slots of this wave's two coarse planes: item completed by step jstep reads the planes of steps jstep - 2 .. jstep
        const _Float16* cb[2];
        cb[0] = s_cimg + ((jstep - 2 + pz) & (CSL - 1)) * CSLOT_H;
        cb[1] = s_cimg + ((jstep - 1 + pz) & (CSL - 1)) * CSLOT_H;
        // the three fine planes of the skip chunk this wave's rows read
        const _Float16* prow[3];
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
          int slot = rb + lz + kz;
          if (slot >= NZ) slot -= NZ;
          prow[kz] = s_img + bfrag0 + slot * (ZP * 8);
        }
        const _Float16* wsk = s_w + wslot * 8;
        f32x4 b4[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int y = m_ty * 8 + ly0 + RS * r;
          const int yc = y == 0 ? 0 : (y == a.H - 1 ? 2 : 1);
          b4[r] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s_btab) + ((zc * 3 + yc) * 3) * 32 + m_bofs);
        }
        // ONE software pipeline of 15 K-steps (6 coarse, 9 (kz, ky) of the skip chunk), 6 matrix instructions each: the
        // fragments of step t + 1 are requested behind the matrix instructions of step t and pinned there
        // (sched_group_barrier) -- left alone hipcc sinks every ds_read to just before its consumer (ds_read; s_waitcnt
        // lgkmcnt(0..2); v_mfma), one exposed LDS latency per matrix instruction pair.  On the per-step kernel the staging
        // waves were the critical path and a denser matrix stream only took issue slots from them (round 3: -1..-6 %);
        // with the up-convolution composed into the weights the MULTIPLYING waves are (stamps: 72 % of an item in this
        // phase, the staging waves parked 31 %).
        //   coarse step s: per row r its class' weights (hi, lo) + its coarse row (hi, lo): 8 reads
        //   skip step (kz, ky): the weights (2 reads) + the image rows ky, ky + 1 of plane kz, of which row ky came with the
        //   previous step: rows 0, 1 with (kz, 0), row ky + 1 later -- 14 reads per kz as in the plain loop
        f16x8 cA[2][2], cB[2][R][2];             // coarse fragments, two sets: the class' weights (one class per wave) and the rows' coarse rows
        f16x8 sA[2][2];                          // skip weights, two sets
        constexpr int NSR = RS * (R - 1) + 3;    // input rows of a plane the wave's rows read (5)
        f16x8 sR[2][NSR][2];                     // skip image rows of plane kz (set kz & 1): [row][hi | lo]
        const int cls = pz * 2 + (ly0 & 1);
        auto load_coarse = [&](int cs_, int set) {
          const _Float16* bp = cb[cs_ / 3] + cofs[cs_];
          const _Float16* wp = s_wc + ((((cls) * 6 + cs_) * 2) * 64 + lane) * 8;
          cA[set][0] = *reinterpret_cast<const f16x8*>(wp);
          if constexpr (!ONEP) cA[set][1] = *reinterpret_cast<const f16x8*>(wp + 64 * 8);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            cB[set][r][0] = *reinterpret_cast<const f16x8*>(bp + r * CW * 8);
            if constexpr (!ONEP) cB[set][r][1] = *reinterpret_cast<const f16x8*>(bp + r * CW * 8 + CPL * 8);
          }
        };
        auto load_skip = [&](int kz, int ky) {    // what step (kz, ky) needs beyond what step (kz, ky - 1) left
          const _Float16* wp = wsk + (kz * 3 + ky) * (2 * 32 * 8);
          if constexpr (!ONEP) sA[(kz * 3 + ky) & 1][1] = *reinterpret_cast<const f16x8*>(wp + 32 * 8);      // (hi: sAh, resident)
          // rows the step's column tiles read (RS r + ky) that no earlier step of the plane has loaded: (0, 2), (1, 3), (4)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int jr = RS * r + ky;
            if (ky >= RS && r < R - 1) continue;      // row RS r + ky = RS (r + 1) + (ky - RS): came with step ky - RS
            sR[kz & 1][jr][0] = *reinterpret_cast<const f16x8*>(prow[kz] + jr * HXP * 8);
            if constexpr (!ONEP) sR[kz & 1][jr][1] = *reinterpret_cast<const f16x8*>(prow[kz] + jr * HXP * 8 + PREC_H);
          }
        };
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        constexpr int RDP = ONEP ? 1 : 2;         // fragments per operand: hi (+ lo)
        constexpr int NCR = RDP * (1 + R);        // reads of a coarse step: the class' weights + R coarse rows
        load_coarse(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NCR, 0);
#pragma unroll
        for (int t = 0; t < 15; ++t) {
          int nrd = 0;                            // reads requested during this step
          if (t + 1 < 6) { load_coarse(t + 1, (t + 1) & 1); nrd = NCR; }
          else if (t + 1 < 15) { const int q = t + 1 - 6; load_skip(q / 3, q % 3); nrd = (RDP - 1) + RDP * (q % 3 < RS ? R : 1); }
          if (t < 6) {
            const int set = t & 1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
              acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cA[set][0], cB[set][r][0], t == 0 ? b4[r] : acc[r], 0, 0, 0);
              if constexpr (ONEP) {
                if (t == 0) accx[r] = zero4;
              } else {
                accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cA[set][0], cB[set][r][1], t == 0 ? zero4 : accx[r], 0, 0, 0);
                accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cA[set][1], cB[set][r][0], accx[r], 0, 0, 0);
              }
            }
          } else {
            const int q = t - 6, kz = q / 3, ky = q % 3;
#pragma unroll
            for (int r = 0; r < R; ++r) {
              acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sAh[q], sR[kz & 1][RS * r + ky][0], acc[r], 0, 0, 0);
              if constexpr (!ONEP) {
                accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sAh[q], sR[kz & 1][RS * r + ky][1], accx[r], 0, 0, 0);
                accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sA[q & 1][1], sR[kz & 1][RS * r + ky][0], accx[r], 0, 0, 0);
              }
            }
          }
          // pin: one read of the next step behind each of this step's matrix instructions
          constexpr int NMF = (ONEP ? 1 : 3) * R;
          const int pairs = nrd < NMF ? nrd : NMF;
#pragma unroll
          for (int i = 0; i < NMF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < pairs) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          // (a step that requests more reads than it multiplies: the coarse steps; with one product also the first skip step of a plane)
          constexpr int EXC = NCR > NMF ? NCR - NMF : 1;
          if (t + 1 < 6 && NCR > NMF) __builtin_amdgcn_sched_group_barrier(0x100, EXC, 0);
          constexpr int NS0 = (RDP - 1) + R * RDP;                  // reads a plane's first skip steps request (R new rows each)
          constexpr int EXS = NS0 > NMF ? NS0 - NMF : 1;
          if (t + 1 >= 6 && t + 1 < 15 && (t + 1 - 6) % 3 == 0 && NS0 > NMF) __builtin_amdgcn_sched_group_barrier(0x100, EXS, 0);
        }
        return;
      }
      if constexpr (POOLM) {
        // 2 rows x 2 planes per wave: per kz the three weight fragments stay in registers while the two planes' four rows
        // each pass through the same row registers (live: 24 + 32 VGPRs instead of 8 + 64 for both planes at once)
        const _Float16* img = s_img + bfrag0;
        const _Float16* wch = s_w + wslot * 8;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
          f16x8 ah[3], al[ONEP ? 1 : 3];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const _Float16* wp = wch + (kz * 3 + ky) * (2 * 32 * 8);
            ah[ky] = *reinterpret_cast<const f16x8*>(wp);
            if constexpr (!ONEP) al[ky] = *reinterpret_cast<const f16x8*>(wp + 32 * 8);
          }
#pragma unroll
          for (int pq = 0; pq < 2; ++pq) {
            int slot = rb + lz + pq + kz;
            if (slot >= NZ) slot -= NZ;
            const _Float16* row0 = img + slot * (ZP * 8);
            f16x8 bh[4], bl[ONEP ? 1 : 4];
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
              bh[jr] = *reinterpret_cast<const f16x8*>(row0 + jr * HXP * 8);
              if constexpr (!ONEP) bl[jr] = *reinterpret_cast<const f16x8*>(row0 + jr * HXP * 8 + PREC_H);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
              for (int rr = 0; rr < 2; ++rr) {
                const int r = 2 * pq + rr;
                const bool fresh = kz == 0 && ky == 0;      // the bias is the first product's C operand
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ky], bh[rr + ky], fresh ? bias4 : acc[r], 0, 0, 0);
                if constexpr (ONEP) {
                  if (fresh) accx[r] = zero;
                } else {
                  accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ky], bl[rr + ky], fresh ? zero : accx[r], 0, 0, 0);
                  accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ky], bh[rr + ky], accx[r], 0, 0, 0);
                }
              }
            }
          }
        }
        return;
      }
#pragma unroll
      for (int chunk = (UP == 2 ? 1 : 0); chunk < NCH; ++chunk) {
        const _Float16* img = s_img + (UP == 2 ? 0 : chunk) * CHUNK_H + bfrag0;
        const _Float16* wch = s_w + (UP == 2 ? 0 : chunk) * W_H + wslot * 8;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
          int slot = rb + lz + kz;
          if (slot >= NZ) slot -= NZ;
          const _Float16* row0 = img + slot * (ZP * 8);
          f16x8 bh[R + 2], bl[(IN16S || ONEP) ? 1 : R + 2];
#pragma unroll
          for (int jr = 0; jr < R + 2; ++jr) {
            bh[jr] = *reinterpret_cast<const f16x8*>(row0 + jr * HXP * 8);
            if constexpr (IN16 == 0) bl[jr] = *reinterpret_cast<const f16x8*>(row0 + jr * HXP * 8 + PREC_H);      // (neither fp16 input nor one product)
          }
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const _Float16* wp = wch + (kz * 3 + ky) * (2 * 32 * 8);
            const f16x8 ah = *reinterpret_cast<const f16x8*>(wp);
            f16x8 al = ah;
            if constexpr (!ONEP) al = *reinterpret_cast<const f16x8*>(wp + 32 * 8);
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const bool fresh = UP != 2 && chunk == 0 && kz == 0 && ky == 0;      // the bias is the first product's C operand
              const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
              acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[r + ky], fresh ? bias4 : acc[r], 0, 0, 0);
              if constexpr (ONEP) {
                if (fresh) accx[r] = zero;
              } else if constexpr (IN16 == 0) {
                accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[r + ky], fresh ? zero : accx[r], 0, 0, 0);
                accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[r + ky], accx[r], 0, 0, 0);
              } else {                                                   // fp16 input: its lo part is zero
                accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[r + ky], fresh ? zero : accx[r], 0, 0, 0);
              }
            }
          }
        }
      }
    };

    // ---- epilogue state of THIS wave (column it is storing) ----
    int e_ci = -1, e_n = 0, e_ty = 0, e_tx = 0, e_hflip = 0;
    vx_dkey e_key = {0u, 0u};
    float* e_ho = nullptr;     // head: this lane's output pointer for z0 = 0, r = 0 (un-flipped position)
    ptrdiff_t e_hz = 0, e_hy = 0;

    auto epilogue = [&](int ci, int k) {
      if (ci != e_ci) {   // a new column: sample, tile row / column, dropout key, head pointers
        e_ci = ci;
        col_of(ci, e_n, e_ty, e_tx);
        if (EPI == 1 || EPI == 2 || EPI == 4) e_key = vx_drop_key(seed_out, kernarg()->a.drop_layer, (uint32_t)e_n);
        // the key words live in VECTOR registers across the items of the column: as scalars they raised the pooling instance's
        // SGPR spills from 6 to 8 and contr_1_2 ran 5 % slower (same-box A/B of the one- and two-word keys, profiles/r06_dropout_generator.txt)
        asm volatile("" : "+v"(e_key.a), "+v"(e_key.b));
        if (HEAD) {
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
      const uint32_t hw_item = (EPI == 1 || EPI == 2 || EPI == 4) ? vx_drop_word(e_key, (vox0 >> 2) + hword_l) : 0u;
      const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(reinterpret_cast<char*>(kernarg()->a.out) + (size_t)e_n * out_sample * (a.out_f16 ? 2 : 4)), 0, VX_NUMREC, 0x00020000);
      // (measured and not kept, round 5: the R keep-words requested up front instead of one ds_bpermute + s_waitcnt lgkmcnt(0) per
      // row: +-0.5 % on all four layers that draw them, same-process A/B -- the exposed LDS round trips are not what the epilogue costs)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        f32x4 v;       // main + cross * 2^-11: one fma per element (exact scaling: the bits of multiply-then-add)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ONEP ? acc[r][j] : fmaf(accx[r][j], 1.0f / 2048.f, acc[r][j]);
        if (STATS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { ssum[j] += v[j]; ssq[j] = fmaf(v[j], v[j], ssq[j]); }
        }
        if constexpr (POOLM) {
          // the block's dropout (applied after the InstanceNorm that is not known yet): maximum over the KEPT raw values
          uint32_t bits = 0xFu;
          if (a.drop_mode == VX_DROP_HASH) bits = ((uint32_t)__builtin_amdgcn_ds_bpermute(hbp + 32 * r, (int)hw_item) >> hsh) & 0xFu;
          pl_any |= ~bits & 0xFu;
#pragma unroll
          for (int j = 0; j < 4; ++j) pl_max[j] = fmaxf(pl_max[j], ((bits >> j) & 1u) ? v[j] : -INFINITY);
        }
        if (f_lrelu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.01f * v[j]);
        } else if (f_relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (EPI == 1 || EPI == 2) {
          const uint32_t bits = ((uint32_t)__builtin_amdgcn_ds_bpermute(hbp + 32 * r, (int)hw_item) >> hsh) & 0xFu;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);
        }
        if (HEAD) {
          float* o = e_ho + (ptrdiff_t)k * e_hz + (ptrdiff_t)r * e_hy;
#pragma unroll
          for (int c = 0; c < HC; ++c) {
            if (c < a.head_C) {
              float part = hb[c];
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) part = fmaf(hw4[c][kk], v[kk], part);
              part = vx_add_xor16(part);
              // (measured alternative: the g-odd lane finishing and storing class c + 1 -- one exchange and one store per
              // class pair with all 64 lanes active -- 1.282 instead of 1.249 ms per 320 samples, same box)
              if (XP_ABL & 64) asm volatile("" :: "v"(part));      // diagnostic build: head stores off
              else if (!(g & 1)) o[(size_t)c * hnvox] = part;
            }
          }
        } else {
          rmax = fmaxf(fmaxf(rmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
          if (EPI == 1 && a.out_f16) {     // reduced-storage mode: the tensor leaves as fp16 (out_pitch in halves)
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const f16x4 h4 = __builtin_convertvector(v, f16x4);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h4), osrd, (int)ovoff[r], (int)(osoff >> 1), 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), osrd, (int)ovoff[r], (int)osoff, 0);
          }
          // gfx950 store-data hazard with an SGPR soffset (conv3d_mfma.hip)
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 3" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (POOLM) {
        // the x-neighbour voxel (same channels) sits 32 lanes away: lanes 0..31 finish the window and store it
        // (v_permlane32_swap: the upper half of the first operand <-> the lower half of the second; with both = x the
        // second ends as [x.hi, x.hi] -- lanes 0..31 read their partner's value without a trip through the LDS queue)
        f32x4 o;
        uint32_t oany;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float sa = pl_max[j], sb = pl_max[j];
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(sa), "+v"(sb));
          o[j] = sb;
        }
        {
          uint32_t sa = pl_any, sb = pl_any;
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(sa), "+v"(sb));
          oany = sb;
        }
        if (lane < 32) {
          f32x4 mx;
#pragma unroll
          for (int j = 0; j < 4; ++j) mx[j] = fmaxf(pl_max[j], o[j]);
          const int Dp = a.D >> 1, Hp = a.H >> 1, Wp = a.W >> 1;
          const size_t pv = (((size_t)e_n * Dp + ((k * TZ + lz) >> 1)) * Hp + ((e_ty * 8 + ly0) >> 1)) * Wp + e_tx * 16 + m;
          const auto kp = kernarg();
          *reinterpret_cast<f32x4*>(kp->a.pool_out + pv * 8 + oc) = mx;
          kp->a.pool_flags[pv * 2 + (g & 1)] = pl_any | oany;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) pl_max[j] = -INFINITY;
        pl_any = 0;
      }
      if (STATS && k == KZ - 1) {
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
      const auto kp = kernarg();
      const int epc = kp->stat_epc;
      const int ntile = cps * epc;
      float* dst = kp->a.stats_partial + (((size_t)n * ntile + (size_t)(ty * ka.tiles_x + tx) * epc) * 8) * 2;
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
      for (int i = 16 + tid; i < epc * 16; i += 256) {
        if (tid < 256) dst[i] = 0.f;
      }
    };

    Cur cx = {0, 0};
    int j = 0;                                   // S_j = cx;  its slot group is j % 3
    int grp_x = 0;
    int prev_ci = -1, prev_k = 0;                // waves 4..7: the item still to store
    int fl_ci = -1, fl_at = 0;                   // column whose statistics are complete after barrier fl_at
    while (cx.ci < ncol_wg) {
      __syncthreads();
      XP_STAMP(0);
      if (STATS && fl_ci >= 0 && j >= fl_at && !late) { flush_col(fl_ci); fl_ci = -1; }
      const bool comp = cx.s >= 1;
      const int item_k = cx.s - 1;
      int grp_c = grp_x + 1; if (grp_c == 3) grp_c = 0;
      int rb = (grp_x == 0 ? 2 : grp_x - 1) * TZ + (TZ - 2);           // first plane of the item: group of S_{j-1}, plane TZ - 2
      // waves 0..3: multiply(j), store(j);  waves 4..7: store(j-1), multiply(j)
      if (late) {
        if (prev_ci >= 0 && !(XP_ABL & 2)) { epilogue(prev_ci, prev_k); prev_ci = -1; }
        XP_STAMP(2);
        if (comp) { if (!(XP_ABL & 1)) multiply(rb, j, cx.ci, item_k); prev_ci = cx.ci; prev_k = item_k; }
        XP_STAMP(1);
      } else {
        if (comp && !(XP_ABL & 1)) multiply(rb, j, cx.ci, item_k);
        XP_STAMP(1);
        if (comp && !(XP_ABL & 2)) epilogue(cx.ci, item_k);
        XP_STAMP(2);
      }
      if (STATS && comp && item_k == KZ - 1) { fl_ci = cx.ci; fl_at = j + 2; }
#ifdef VX_CONV_STAMPS
      ++st_iters;
#endif
      advance(cx);
      grp_x = grp_c;
      ++j;
    }
    if (late && prev_ci >= 0) epilogue(prev_ci, prev_k);
    if (STATS) {
      __syncthreads();
      if (fl_ci >= 0 && !late) flush_col(fl_ci);
    }
  }
  if (EPI != 2 && a.range_flag) {
    float mx = rmax;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    // only magnitudes within a factor two of the fp16 limit are reported (2 x the margin the norm kernels use)
    if (lane == 0 && !(mx < 32768.f)) atomicMax(a.range_flag, __float_as_uint(mx));
  }
#ifdef VX_CONV_STAMPS
  if (ka.stamps && lane == 0) {
    unsigned long long* d = ka.stamps + ((size_t)blockIdx.x * 16 + wave) * 8;
    for (int i = 0; i < 6; ++i) d[i] = st_sum[i];
    d[6] = st_iters;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
bool vx_conv3d_xp8_applies(int D, int H, int W, int Cin, int Cout) {
  if (vx_cfg().conv_fp32 != 0 || vx_cfg().s16_no_xp8) return false;
  // D >= 8: a column has at least two items (the statistics hand-off between the wave halves needs the spacing)
  return Cout == 8 && (Cin == 8 || Cin == 16) && W % 32 == 0 && H % 8 == 0 && D % 4 == 0 && W >= 32 && H >= 8 && D >= 8;
}

template <int NCH, int EPI, int PRE, int UP, int NPW, int IN16 = 0>
static int launch_xp8w(const Xp8wArgs& ka, hipStream_t s) {
  constexpr int TZ = 4 / NCH, NZ = 3 * TZ, ZP = 170;
  constexpr int PP = ((NZ * ZP + 15) / 16) * 16;
  constexpr int NIMG = UP == 2 ? 1 : NCH;      // UP = 2: skip chunk + coarse window + composed weights + bias table
  constexpr size_t lds = (size_t)NIMG * 2 * 2 * PP * 8 * 2 + (size_t)NIMG * (9 * 2 * 32 * 8) * 2 + 8 * 16 * 2 * 4 +
                         (UP == 2 ? (size_t)(2 * 2 * 4 * 6 * 18 * 8) * 2 + (size_t)(4 * 6 * 2 * 64 * 8) * 2 + 27 * 8 * 4 : 0);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = conv3d_xp8w_kernel<NCH, EPI, PRE, UP, NPW, IN16>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv3d_k3(xp8w): hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr = true;
  }
  int gx = vx_cu_count();      // one persistent workgroup per CU (its LDS image fills the CU)
  if (gx > ka.ncols) gx = ka.ncols;
  static const char* kname = vx_kname("conv3d_xp8w_kernel<%d,%d,%d,%d,%d,%d>", NCH, EPI, PRE, UP, NPW, IN16);   // as rocprofv3 prints it
  vx_note_kernel(kname);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3((8 + NPW) * 64), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv3d_k3(xp8w)");
  return VX_OK;
}

// 1 = not taken (the caller uses the general tile kernel)
int vx_conv3d_k3_xp8(const vx_conv3d_args& a, int stat_tiles, hipStream_t s) {
  Xp8wArgs ka;
  ka.a = a;
  const int nch = a.Cin / 8, tz = 4 / nch;
  ka.tiles_x = a.W / 32; ka.tiles_y = a.H / 8; ka.kz = a.D / tz;
  const int cps = ka.tiles_x * ka.tiles_y;
  ka.ncols = a.N * cps;
  ka.mcps = (unsigned)((1ull << 32) / (unsigned)cps) + 1u;
  ka.rep = (a.in_mean && a.in_repeat > 1 && a.N % a.in_repeat == 0) ? a.in_repeat : 1;
  ka.mrep = (unsigned)((1ull << 32) / (unsigned)ka.rep) + 1u;
  ka.mtx = (unsigned)((1ull << 32) / (unsigned)ka.tiles_x) + 1u;
  ka.stat_epc = stat_tiles / cps;
  ka.stamps = nullptr;
  ka.abl = 0;
#ifdef VX_CONV_STAMPS
  if (const char* e = getenv("VX_CONV_DBG_PTR")) ka.stamps = (unsigned long long*)strtoull(e, nullptr, 0);
  if (const char* e = getenv("VX_XP_ABL")) ka.abl = atoi(e);
#endif
  if ((int64_t)a.N * cps >= (1ll << 31)) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8w): too many columns");
  if (a.stats_partial && (stat_tiles % cps || a.act != VX_ACT_NONE || (a.drop_mode != VX_DROP_NONE && !a.pool_out) || a.head_out))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8w): statistics go with a plain epilogue");
  const int pre = a.in_split ? 2 : (a.in_mean ? 1 : 0);
  if (a.in_split && (a.Cin != 8 || a.in_xblk || a.in_pitch != 8 || a.up_in || !a.stats_partial))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8w): a pre-split input (vx_prenorm_split) goes with a dense 8-channel tensor and the "
            "statistics epilogue (contr_1_2)");
  int epi;
  if (a.stats_partial) epi = a.pool_out ? 4 : 0;
  else if (a.head_out) epi = 2;
  else if (a.drop_mode == VX_DROP_HASH) epi = 1;
  else epi = 3;
  if (epi == 2 && a.act == VX_ACT_LRELU && a.drop_mode == VX_DROP_NONE) epi = 5;        // head without dropout (ensemble members)
  if (epi == 2 && !(a.act == VX_ACT_LRELU && a.drop_mode == VX_DROP_HASH)) return 1;   // other heads: general kernel
  if (epi == 1 && a.act != VX_ACT_LRELU) return 1;
  // 2: the up-convolution composed into the weights (vx_pack_conv3d_upfused), 1: evaluated per step by the staging waves
  const int up = a.up_in ? ((a.up_fused && !vx_cfg().s16_no_upcompose) ? 2 : 1) : 0;
  // producer waves: 8 (two per SIMD) for the two-chunk layers, whose staging is the heavy side; 4 for the one-chunk layers:
  // their consumers hold R = 4 image rows + accumulators (134-161 VGPRs) and at 16 waves per workgroup (128-VGPR cap) EVERY
  // one-chunk instance spilled 5-30 VGPRs into its item loop (round-3 verdict; tools/check_spills.sh now fails the build on any
  // such instance).  The round-3 knob vx_config.s16_pw and the 20 instances behind it are gone.
  if (a.pool_out && nch != 1) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8w): the pooled output goes with Cin = 8");
  if (a.out_split) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8w): out_split is an epilogue of the tile kernel");
  if (a.up_split && !up) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8w): up_split without up_in");
  if (a.out_f16 && (epi != 1 || a.out_xblk || a.out_coff))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8w): fp16 output goes with the LeakyReLU + dropout epilogue and a dense output tensor");
  if (a.in_f16) {
    if (!(nch == 1 && epi == 2 && pre == 0 && !up && !a.in_xblk))
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8w): fp16 input goes with the dense 8-channel layer that carries the fused head");
    if (a.products == 1) return launch_xp8w<1, 2, 0, 0, 4, 3>(ka, s);
    return launch_xp8w<1, 2, 0, 0, 4, 1>(ka, s);
  }
  if (a.products == 1) {
    // opt-in throughput mode (vx_config.storage16 = 2): one fp16 product per fp32 product on the three full-resolution launches
    if (nch == 1 && epi == 4 && pre == 2 && up == 0) return launch_xp8w<1, 4, 2, 0, 4, 2>(ka, s);
    if (nch == 2 && epi == 1 && pre == 1 && up == 2) return launch_xp8w<2, 1, 1, 2, 8, 2>(ka, s);
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8w): products = 1 exists for the MC-dropout forward's contr_1_2, upscale2 + expand_1_1 and "
            "expand_1_2 + head (nch %d, epilogue %d, prologue %d, up %d)", nch, epi, pre, up);
  }
#define XP8W_CASE(N_, E_, P_, U_)                                                         \
  if (nch == N_ && epi == E_ && pre == P_ && up == U_)                                    \
    return launch_xp8w<N_, E_, P_, U_, (N_ == 2 ? 8 : 4)>(ka, s)
  XP8W_CASE(1, 4, 0, 0); XP8W_CASE(1, 4, 1, 0); XP8W_CASE(1, 4, 2, 0); XP8W_CASE(1, 0, 2, 0);
  XP8W_CASE(1, 0, 0, 0); XP8W_CASE(1, 0, 1, 0); XP8W_CASE(1, 1, 0, 0); XP8W_CASE(1, 2, 0, 0); XP8W_CASE(1, 3, 0, 0); XP8W_CASE(1, 3, 1, 0);
  XP8W_CASE(1, 5, 0, 0);
  XP8W_CASE(2, 1, 0, 0); XP8W_CASE(2, 1, 1, 0); XP8W_CASE(2, 3, 0, 0); XP8W_CASE(2, 3, 1, 0);
  XP8W_CASE(2, 1, 0, 1); XP8W_CASE(2, 1, 1, 1); XP8W_CASE(2, 3, 0, 1); XP8W_CASE(2, 3, 1, 1);
  XP8W_CASE(2, 1, 0, 2); XP8W_CASE(2, 1, 1, 2); XP8W_CASE(2, 3, 0, 2); XP8W_CASE(2, 3, 1, 2);
#undef XP8W_CASE
  if (up) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8w): no fused up-convolution for this epilogue (act=%d, drop_mode=%d, statistics=%d)",
                  a.act, a.drop_mode, a.stats_partial ? 1 : 0);
  if (a.pool_out) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(xp8w): no pooled output for this epilogue");
  return 1;
}

// ---------------------------------------------------------------------------------------------------------------
// Composed up-convolution (vx_conv3d_args.up_fused): W_eff[class (pz, py)][coarse tap (czi, cyi, cxi)][ci][(dx, co)] =
// sum over the 3x3x3 taps (kz, ky, kx) that land on that coarse voxel for this output parity of
// sum_cu W1[co][cu][kz][ky][kx] U[ci][cu][sub-position of the fine voxel], in float64; split into fp16 (hi, lo 2^11) in the
// A-fragment order of v_mfma_f32_16x16x32_f16: [class][step][hi | lo][lane (g 16 + m)][8 halves = channels 8 oct + j of octet
// o = 4 step + g].  Behind it the bias table [27 border classes (zc, yc, xc)][8]: b1 + the up bias through the taps that lie
// INSIDE the volume (a tap outside sees the zero padding of the concatenated tensor).
namespace {
__device__ __forceinline__ void up_tap(int parity, int k, int& coarse, int& sub) {
  const int t = parity + k - 1;                 // fine offset from 2 Q: -1 .. 2
  coarse = t < 0 ? -1 : t >> 1;                 // floor(t / 2)
  sub = t - 2 * coarse;
}
__global__ void pack_upfused_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ uw,
                                    const float* __restrict__ ub, float* __restrict__ out) {
  constexpr int NWT = 4 * 6 * 64 * 8;
  _Float16* oh = reinterpret_cast<_Float16*>(out);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NWT + 27 * 8; i += gridDim.x * blockDim.x) {
    if (i < NWT) {
      const int j = i & 7, lane = (i >> 3) & 63, s = (i >> 9) % 6, cls = i / (512 * 6);
      const int pz = cls >> 1, py = cls & 1, m = lane & 15, g = lane >> 4, dx = m >> 3, co = m & 7;
      const int o = 4 * s + g, oct = o & 1, cxi = (o >> 1) % 3, cyi = (o / 6) & 1, czi = o / 12, ci = oct * 8 + j;
      double acc = 0.0;
      for (int kz = 0; kz < 3; ++kz) {
        int cz, sz; up_tap(pz, kz, cz, sz);
        if (cz + 1 - pz != czi) continue;
        for (int ky = 0; ky < 3; ++ky) {
          int cy, sy; up_tap(py, ky, cy, sy);
          if (cy + 1 - py != cyi) continue;
          for (int kx = 0; kx < 3; ++kx) {
            int cx, sx; up_tap(dx, kx, cx, sx);
            if (cx + 1 != cxi) continue;
            for (int cu = 0; cu < 8; ++cu)
              acc += (double)w1[(co * 16 + cu) * 27 + kz * 9 + ky * 3 + kx] * (double)uw[(ci * 8 + cu) * 8 + sz * 4 + sy * 2 + sx];
          }
        }
      }
      const double c = fmin(fmax(acc, -65504.0), 65504.0);
      const _Float16 h = (_Float16)(float)c;
      const _Float16 l = (_Float16)(float)((acc - (double)(float)h) * 2048.0);
      oh[(((cls * 6 + s) * 2 + 0) * 64 + lane) * 8 + j] = h;
      oh[(((cls * 6 + s) * 2 + 1) * 64 + lane) * 8 + j] = l;
    } else {
      const int t = i - NWT, co = t & 7, cls = t >> 3, xc = cls % 3, yc = (cls / 3) % 3, zc = cls / 9;
      double acc = (double)b1[co];
      for (int kz = 0; kz < 3; ++kz) {
        if ((zc == 0 && kz == 0) || (zc == 2 && kz == 2)) continue;
        for (int ky = 0; ky < 3; ++ky) {
          if ((yc == 0 && ky == 0) || (yc == 2 && ky == 2)) continue;
          for (int kx = 0; kx < 3; ++kx) {
            if ((xc == 0 && kx == 0) || (xc == 2 && kx == 2)) continue;
            for (int cu = 0; cu < 8; ++cu) acc += (double)w1[(co * 16 + cu) * 27 + kz * 9 + ky * 3 + kx] * (double)ub[cu];
          }
        }
      }
      out[NWT + t] = (float)acc;              // floats behind the 2 NWT halves (= NWT floats) of the weights
    }
  }
}
}  // namespace

extern "C" int64_t vx_conv3d_upfused_packed_floats(void) { return 4 * 6 * 64 * 8 + 27 * 8 + 8; }   // (+ 8: 16-byte multiple)

extern "C" int vx_pack_conv3d_upfused(const float* w1_torch, const float* b1, const float* up_w_torch, const float* up_b,
                                      float* packed, vx_stream_t stream) {
  if (!w1_torch || !b1 || !up_w_torch || !up_b || !packed) VX_FAIL(VX_E_NULL, "vx_pack_conv3d_upfused: null pointer");
  if (!vx_aligned16(packed)) VX_FAIL(VX_E_ALIGN, "vx_pack_conv3d_upfused: packed must be 16-byte aligned");
  hipLaunchKernelGGL(pack_upfused_kernel, dim3(48), dim3(256), 0, (hipStream_t)stream, w1_torch, b1, up_w_torch, up_b, packed);
  VX_CHECK_LAUNCH("vx_pack_conv3d_upfused");
  return VX_OK;
}
