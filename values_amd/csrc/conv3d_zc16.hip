// K1 for the Cout = 16 layers below full resolution (round 5): the z-column walk of conv3d_xp8w.hip -- a workgroup owns a
// column of 32 x 8 voxels of one sample and walks it along z with a ROLLING window of z-planes in LDS, staging in its own
// waves -- for the PLAIN matrix form (rows = the 16 output channels, columns = 16 voxels along x), which is what 16 output
// channels fill without padding rows (the x-pair form of the 8-channel kernels would run two row tiles at 75 % density).
//
//   * waves 0..7 (MULTIPLYING) own (z-plane of the item, x-half, 4 consecutive y-rows) = 4 column tiles each: waves 0..3
//     multiply(j), store(j); waves 4..7 store(j - 1), multiply(j) -- one wave of a SIMD stores while its partner multiplies;
//   * waves 8..11 (STAGING, one per SIMD) commit step S(j+1) from registers into the LDS image (fp16 hi / lo split, the
//     optional normalise-on-load or pool-finish prologue of the PRODUCING block) and issue the loads of S(j+2).
//   One barrier per item.  In the tile kernel (conv3d_s16.hip) every wave stages AND multiplies in lockstep, so each
//   prologue instruction sits on the critical path (contr_2_2: +0.136 ms for the pass it replaces); here the prologue, the
//   pooled epilogue's partner pass and the loads ride in waves of their own.
//
// K schedule with input-row reuse (Cin = 16; 14 steps of K = 32 for 27 taps x 16 channels = 13.5):
//   steps (kz, ky), 9 of them: k-groups g = (kx = g >> 1 in {0, 1}, channel octet g & 1).  The B fragment of input row
//     (plane z + kz, row y) does not depend on ky: a wave reads the 4 + 2 row fragments of a plane ONCE and uses them for
//     ky = 0, 1, 2 of its four output rows (the tile kernel reads a fragment per (step, column tile): 0.83 ds_read_b128
//     per matrix instruction, at the LDS array's limit; here 0.5);
//   steps 9..11 (ky): the kx = 2 taps of kz = 0 and kz = 1 as the two k-group pairs (g >> 1 = kz);
//   step 12: the kx = 2 taps of (kz = 2, ky = 0) and (kz = 2, ky = 1) as the two pairs (a fragment per output row: rows r, r + 1);
//   step 13: the kx = 2 tap of (kz = 2, ky = 2) (k-groups 2, 3: zero weights).
// Cin = 8: 9 steps (kz, ky), k-groups = kx 0..2 + a zero group -- the same row reuse, 75 % dense (the layer is HBM-bound).
//
// LDS: image [octet][hi | lo][6 plane slots x 10 x 34 positions][8 halves] = 128 KB (Cin = 16) + weights 28 KB + 1 KB of
// statistics slots.  TZ = 2 output planes per item; three groups of 2 plane slots (a step writes one group while the item
// in flight reads the other two -- conv3d_xp8w.hip's scheme).
#include "s16_common.h"

struct Zc16Args {
  vx_conv3d_args a;
  const float* w;             // this kernel's block of the packed weights (vx_conv3d_zc16_packed_floats)
  int tiles_x, tiles_y, kz;   // columns per sample = tiles_x * tiles_y; kz = items per column
  int ncols;                  // columns in the launch
  unsigned mcps, mtx;         // multiply-high magics: / (tiles_x * tiles_y), / tiles_x
  int stat_epc;               // statistics entries per column in stats_partial (entry 0 real, the rest zero)
  unsigned long long* stamps;
  int abl;                    // diagnostic build only: phase ablation bits (1 no multiply, 2 no epilogue, 4 no commit, 8 no loads, 16 no LDS writes, 32 no split)
};

#ifdef VX_CONV_STAMPS
#define ZC_STAMP(i)                                                                      \
  do {                                                                                   \
    unsigned long long t_;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    st_sum[i] += t_ - st_last;                                                           \
    st_last = t_;                                                                        \
  } while (0)
#define ZC_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define ZC_ABL ka.abl
#else
#define ZC_ABL 0
#define ZC_STAMP(i) do {} while (0)
#define ZC_WAIT_LOADS() do {} while (0)
#endif

// CIN: input channels (8 or 16).  EPI: 0 bias + statistics + store, 1 LeakyReLU + hash dropout (+ out_split), 3 run-time
// activation without dropout, 4 = 0 PLUS the (y, x) half of the block's 2 x 2 x 2 max-pool (maximum over the KEPT raw values +
// any-dropped bits of every 2 x 2 window of a z-plane; conv3d_xp8w.hip EPI 4 explains the monotonicity argument; the z
// pair is finished by vx_pool_finish_z, which reads a quarter of the tensor's voxels).  PRE: 0 none, 1 InstanceNorm +
// LeakyReLU + dropout of the producing block on load (CIN = 16), 3 pool-finish on load (CIN = 8: vx_conv3d_args.in_pool_flags),
// 4 (round 6, CIN = 16) the input is the PLANAR pre-split tensor of vx_conv3d_args.in_planar -- [N][D][H][octet][hi | lo][W][8 halves],
// the producer's epilogue (out_planar below) has done the fp16 split -- and the staging waves move it into the image by LDS-DMA
// (buffer_load_dwordx4 ... lds: 64 consecutive positions of one (octet, precision) plane per instruction, out-of-range lanes
// write the zero padding): no registers, no conversion, no ds_write stream in front of the multiplying waves' reads
// (profiles/r05_zc16_stamps.txt: that stream cost 23 % of the launch).
// ACC: partial sums (vx_conv3d_args.acc_in) are added before the activation instead of the bias; a multiplying wave requests its
// item's pieces when it starts the item's matrix loop and uses them in the item's epilogue (no branch around the loads, and
// nothing else of that wave is in flight but the previous item's stores).
// UP: the 16 input channels are ConvTranspose3d(32 -> 16, k = 2, s = 2)(up_in) + up_b, EVALUATED BY THE STAGING WAVES while they
// stage (conv3d_xp8w.hip UP = 1 at this kernel's geometry): a fine voxel depends on ONE coarse voxel, so the planes of a step are
// 8 sub-position classes (dz, dy, dx) x [16 co] x [85 coarse voxels] x [32 ci] small GEMMs -- one v_mfma_f32_16x16x32_f16 (x 3:
// split products) per 16 coarse voxels, 12 tiles per staging wave and step, B operands straight from global memory (pre-split
// by the producing conv's epilogue: vx_conv3d_args.up_split), results split and written where staged loads would have gone.
// The transposed conv's launch, its 0.67 GB write and the conv's read of it disappear (unet3D_module.py:157-190, 332-356).
// EPI 5 / 6 (round 6) = 1 / 3 with the PLANAR pre-split output of vx_conv3d_args.out_planar (a compile-time variant: as a run-time
// branch its address arithmetic and the split's temporaries pushed the partial-sum instances past 168 registers into scratch).
template <int CIN, int EPI_, int PRE, int ACC = 0, int UP = 0>
__global__ __launch_bounds__(768) void conv3d_zc16_kernel(Zc16Args ka) {
  constexpr bool PLN = EPI_ == 5 || EPI_ == 6;
  constexpr int EPI = EPI_ == 5 ? 1 : (EPI_ == 6 ? 3 : EPI_);
  static_assert(CIN == 8 || CIN == 16, "8 or 16 input channels");
  static_assert(!PLN || CIN == 16, "the planar output is a 16-channel tensor");
  static_assert(UP == 0 || (CIN == 16 && PRE == 0), "the fused up-convolution produces the 16 input channels itself");
  static_assert(ACC == 0 || EPI == 1 || EPI == 3, "partial sums go with the activation epilogues");
  static_assert(PRE == 0 || ((PRE == 1 || PRE == 4) && CIN == 16) || (PRE == 3 && CIN == 8), "prologues: normalise-on-load / planar DMA for 16, pool-finish for 8 channels");
  static_assert(PRE != 4 || (ACC == 0 && UP == 0), "the planar input is staged by LDS-DMA: no partial sums, no fused up-convolution");
  constexpr int NW = 8, NPW = 4, NTH = (NW + NPW) * 64;
  constexpr int TZ = 2, R = 4;
  constexpr int HX = 34, HY = 10, ZP = HX * HY;
  constexpr int NZ = 3 * TZ;
  constexpr int PP = ((NZ * ZP + 15) / 16) * 16;   // positions per (octet, precision) plane
  constexpr int PREC_B = PP * 16;                  // bytes
  constexpr int OCT_B = 2 * PREC_B;
  constexpr int OCT = CIN / 8;
  constexpr int IMG_B = OCT * OCT_B;
  constexpr int NSTEP = CIN == 16 ? 14 : 9;
  constexpr int W_B = NSTEP * 2 * 1024;            // [step][hi | lo][lane 64][8 halves]
  constexpr int GRP_B = TZ * ZP * 16;              // bytes between two slot groups
  constexpr int PLN_B = ZP * 16;                   // bytes between two plane slots
  constexpr int ROW_B = HX * 16;
  constexpr int Q = CIN / 4;                       // 16-byte fp32 pieces per voxel
  constexpr bool STATS = EPI == 0 || EPI == 4;
  constexpr bool POOL = EPI == 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* s_img = smem_raw;
  unsigned char* s_w = smem_raw + IMG_B;
  float* s_red = reinterpret_cast<float*>(smem_raw + IMG_B + W_B);     // [NW][16][2]
  float* s_bias = s_red + NW * 16 * 2;                                 // [16]: re-read per item (4 registers less across the epilogue)

  const vx_conv3d_args& a = ka.a;
  auto kernarg = [&]() {      // fields used once per item / column are re-read where they are used (conv3d_xp8w.hip)
    typedef const Zc16Args __attribute__((address_space(4))) * kp_t;
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
    const f32x4* src = reinterpret_cast<const f32x4*>(ka.w);
    for (int i = tid; i < W_B / 16; i += NTH) reinterpret_cast<f32x4*>(s_w)[i] = src[i];
    if (tid < 16) s_bias[tid] = a.bias[tid];
  }
  // the image starts as zeros: a zero-weight k-group (Cin = 16: step 13; Cin = 8: the fourth group) multiplies whatever
  // sits at the position it reads, which must be finite from the first item on
  {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < IMG_B / 16; i += NTH) reinterpret_cast<f32x4*>(s_img)[i] = z4;
  }
  __syncthreads();

  // ---- the columns of this workgroup ----
  int vb = blockIdx.x;
  const int G = (int)gridDim.x;
  if ((G & 7) == 0) vb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);   // one XCD: neighbouring columns
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

  const uint32_t seed_in = PRE == 1 ? vx_seed_of(a, a.in_drop_seed) : 0u;
  const uint32_t seed_out = (EPI == 1 || EPI == 4) ? vx_seed_of(a, a.drop_seed) : 0u;
  float rmax = 0.f;   // largest |value| this wave stored (range guard of the split-fp16 consumers)
#ifdef VX_CONV_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last, st_iters = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif

  // pipeline (both roles): step S_j is visible after barrier j; in iteration j the staging waves commit S_{j+1} and load
  // S_{j+2}, the multiplying waves compute the item that S_j completes (step s >= 1 of a column completes item s - 1).
  // Step s of a column carries planes TZ s - (TZ - 1) .. TZ s.
  if (wave >= NW) {
    // =============================================== STAGING ===============================================
    // A wave-iteration is one ROW UNIT of 64 pieces: Cin = 16 -- half a row (16 voxels x 4 channel quads = 1 KiB contiguous
    // in memory); Cin = 8 -- a whole row (32 voxels x 2 quads).  A wave owns RPW consecutive rows of ONE plane (and x-half),
    // so the global row offset is a running scalar add and the LDS row offset an immediate; the 2 halo voxels per row (x = -1,
    // x = 32) are gathered into HIT extra iterations of the per-lane form (one each for waves 3, 2, 1).
    const int pw = wave - NW;
    // staging waves with a prologue (or the fused up-convolution) run at raised priority: they are the critical path (stamps: 93 %
    // busy, 70 % of it in the conversion; the multiplying waves park 33-53 % at the barrier).  Per layer in the forward, same box:
    // pool-finish prologue -8 %, normalise-on-load -6.6 %, fused up-convolution -1.3 %; +2.4 % WITHOUT a prologue and +2.8 % on the
    // pooling instance (whose multiplying waves carry the long epilogue) -- not raised there.
    if constexpr ((PRE != 0 && !POOL) || UP != 0) __builtin_amdgcn_s_setprio(2);
    if constexpr (UP != 0) {
      // ---- fused up-convolution.  A wave owns a plane of the step (u_pz; fine plane 2 s - 1 + u_pz: odd -> dz = 1 of coarse
      // plane s - 1, even -> dz = 0 of coarse plane s) and HALF of that plane's coarse window (6 rows x 18 columns = 108 coarse
      // voxels = 7 column tiles of 16: tiles 0..3 / 4..6), and evaluates all FOUR (dy, dx) classes on each tile it loads: every
      // coarse voxel is fetched once per plane (a first version with one class per wave fetched it four times -- 96 KB per step
      // through the CU's one address path, and the staging waves became the critical path: 0.64 ms against 0.46 + 0.23).
      const int u_pz = pw >> 1, u_half = pw & 1, u_dz = 1 - u_pz;
      constexpr int UT = 4;
      const int Hc = a.H >> 1, Wc = a.W >> 1;
      const int up = a.up_pitch;
      const int urow = Wc * up;
      const int ubiasf = (Hc + 1) * urow + up;                      // keeps (Z = -1, Y = -1, X = -1) offsets non-negative
      // tile i of this wave: coarse voxel c = 16 (4 u_half + i) + m of the window -> (Yi, Xi) = (c / 18, c % 18): coarse
      // (Yc0 - 1 + Yi, Xc0 - 1 + Xi).  Class (dy, dx) of it is the fine position (hy, hx) = (2 Yi + dy - 1, 2 Xi + dx - 1) of the
      // staged 10 x 34 window; outside it (Yi = 0 / 5, Xi = 0 / 17 with the wrong parity) nothing is written.
      unsigned t_voff[UT];
      int t_lds[UT];                   // LDS byte offset of class (0, 0); class (dy, dx): + (dy HX + dx) 16
      unsigned tb_always = 0, tb_xlo = 0, tb_xhi = 0, tb_ylo = 0, tb_yhi = 0;     // bit i: tile i's voxel is ... (per lane)
      unsigned nw[4] = {0, 0, 0, 0};                                              // bit i: class cls of tile i has no position
#pragma unroll
      for (int i = 0; i < UT; ++i) {
        const int c = 16 * (4 * u_half + i) + m;
        const int Yi = c / 18, Xi = c % 18;
        t_voff[i] = (unsigned)((((((u_pz - 1) * Hc + Yi - 1) * Wc) + Xi - 1) * up + 8 * g + ubiasf) * 4);
        t_lds[i] = (g >> 1) * OCT_B + ((u_pz * HY + 2 * Yi - 1) * HX + 2 * Xi - 1) * 16 + (g & 1) * 8;
        if (c >= 108) tb_always |= 1u << i;
        if (Xi == 0) tb_xlo |= 1u << i;
        if (Xi == 17) tb_xhi |= 1u << i;
        if (Yi == 0) tb_ylo |= 1u << i;
        if (Yi == 5) tb_yhi |= 1u << i;
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
          const int dy = cls >> 1, dx = cls & 1;
          const int hy = 2 * Yi + dy - 1, hx = 2 * Xi + dx - 1;
          if (c >= 108 || hy < 0 || hy >= HY || hx < 0 || hx >= HX) nw[cls] |= 1u << i;
        }
      }
      // A operands: [class (dz, dy, dx)][hi | lo][lane][8 halves]: row m = co, k = ci 8 g .. 8 g + 7 (vx_pack_convT_zc16)
      f16x8 u_ah[4], u_al[4];
#pragma unroll
      for (int cls = 0; cls < 4; ++cls) {
        const f16x8* wp = reinterpret_cast<const f16x8*>(a.up_w) + (size_t)(((u_dz * 4 + cls) * 2) * 64 + lane);
        u_ah[cls] = wp[0];
        u_al[cls] = wp[64];
      }
      const f32x4 ubias4 = *reinterpret_cast<const f32x4*>(a.up_b + 4 * g);
      const size_t up_sample = (size_t)(a.D >> 1) * Hc * urow;
      f32x4 ubuf[UT][2];                   // [tile][piece]: ci 8 g .. + 3, 8 g + 4 .. + 7 of the tile's coarse voxel
      unsigned p_bad = 0xFFFFFFFFu;
      bool cs_have = false;
      unsigned cs_bad = 0xFFFFFFFFu, cs_usoff = 0;
      __amdgpu_buffer_rsrc_t cs_usrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.up_in, 0, 0, 0x00020000);
      auto column_state = [&](int ci) {
        const bool have = ci < ncol_wg;
        int n = 0, ty = 0, tx = 0;
        if (have) col_of(ci, n, ty, tx);
        cs_have = have;
        unsigned b = tb_always;
        if (tx == 0) b |= tb_xlo;
        if (tx == ka.tiles_x - 1) b |= tb_xhi;
        if (ty == 0) b |= tb_ylo;
        if (ty == ka.tiles_y - 1) b |= tb_yhi;
        cs_bad = b;
        cs_usoff = (unsigned)(((ty * 4) * urow + tx * 16 * up) * 4);
        cs_usrd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.up_in + (size_t)n * up_sample - ubiasf), 0, VX_NUMREC, 0x00020000);
      };
      auto prefetch = [&](const Cur& c) {
        if (c.s == 0) column_state(c.ci);
        unsigned b = cs_bad;
        // coarse plane s - 1 + u_pz: -1 at step 0 (u_pz = 0), D / 2 at step KZ (u_pz = 1)
        if (!cs_have || (c.s == 0 && u_pz == 0) || (c.s == KZ && u_pz == 1)) b = 0xFFFFFFFFu;
        const unsigned usoff = cs_usoff + (unsigned)((c.s * Hc) * urow * 4);
#pragma unroll
        for (int i = 0; i < UT; ++i) {
          const unsigned v0 = ((b >> i) & 1u) ? VX_OOB : t_voff[i];
          ubuf[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cs_usrd, (int)v0, (int)usoff, 0));
          ubuf[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cs_usrd, (int)(v0 + 16u), (int)usoff, 0));   // (the builtin's last argument is the cache policy, not an offset)
        }
        p_bad = b;
      };
      auto commit = [&](int grp) {
        const int gofs = grp * GRP_B;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < UT; ++i) {
          f16x8 bh, bl;
          if (a.up_split) {          // the producer of the coarse tensor stored fp16 pairs per piece: [hi0 hi1 hi2 hi3 | lo0 .. lo3]
            const u32x4 d0 = __builtin_bit_cast(u32x4, ubuf[i][0]), d1 = __builtin_bit_cast(u32x4, ubuf[i][1]);
            bh = __builtin_bit_cast(f16x8, (u32x4){d0[0], d0[1], d1[0], d1[1]});
            bl = __builtin_bit_cast(f16x8, (u32x4){d0[2], d0[3], d1[2], d1[3]});
          } else {
            f16x4 h0, l0, h1, l1;
            vx_split4(ubuf[i][0], h0, l0);
            vx_split4(ubuf[i][1], h1, l1);
            const u32x2 a0 = __builtin_bit_cast(u32x2, h0), a1 = __builtin_bit_cast(u32x2, h1);
            const u32x2 c0 = __builtin_bit_cast(u32x2, l0), c1 = __builtin_bit_cast(u32x2, l1);
            bh = __builtin_bit_cast(f16x8, (u32x4){a0[0], a0[1], a1[0], a1[1]});
            bl = __builtin_bit_cast(f16x8, (u32x4){c0[0], c0[1], c1[0], c1[1]});
            // vx_split4 writes the lo halves from inline assembly: no wait states before a matrix instruction that reads them
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
          const bool bad = (p_bad >> i) & 1u;
#pragma unroll
          for (int cls = 0; cls < 4; ++cls) {
            f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(u_ah[cls], bh, ubias4, 0, 0, 0);
            f32x4 dxs = __builtin_amdgcn_mfma_f32_16x16x32_f16(u_ah[cls], bl, zero, 0, 0, 0);
            dxs = __builtin_amdgcn_mfma_f32_16x16x32_f16(u_al[cls], bh, dxs, 0, 0, 0);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(dxs[j], 1.0f / 2048.f, d[j]);     // (compiler-visible: the first reader of a matrix result)
            if (bad) v = zero;                               // outside the volume: the conv's zero padding, not the bias
            rmax = vx_max3abs(vx_max3abs(rmax, v[0], v[1]), v[2], v[3]);
            f16x4 hi, lo;
            vx_split4(v, hi, lo);
            if (!((nw[cls] >> i) & 1u)) {
              unsigned char* dst = s_img + gofs + t_lds[i] + ((cls >> 1) * HX + (cls & 1)) * 16;
              *reinterpret_cast<f16x4*>(dst) = hi;
              *reinterpret_cast<f16x4*>(dst + PREC_B) = lo;
            }
          }
        }
      };
      Cur cx = {0, 0}, cc = {0, 0}, cp = {0, 0};
      prefetch(cp); advance(cp);
      commit(0);    advance(cc);
      prefetch(cp); advance(cp);
      int grp_x = 0;
      while (cx.ci < ncol_wg) {
        __syncthreads();
        ZC_STAMP(0);
        int grp_c = grp_x + 1; if (grp_c == 3) grp_c = 0;
        ZC_WAIT_LOADS();
        ZC_STAMP(3);
        if (cc.ci < ncol_wg && !(ZC_ABL & 4)) commit(grp_c);
        ZC_STAMP(4);
        if (!(ZC_ABL & 8)) prefetch(cp);
        ZC_STAMP(5);
#ifdef VX_CONV_STAMPS
        ++st_iters;
#endif
        advance(cx); advance(cc); advance(cp);
        grp_x = grp_c;
      }
    } else if constexpr (PRE == 4) {
      // ---- planar pre-split input, LDS-DMA.  Wave pw owns plane (octet pw >> 1, precision pw & 1) of the image: the slot group of
      // a step is NPOS = 680 consecutive positions of that plane, moved by NI = 11 instructions of 64 positions (the last one
      // re-covers the tail: every lane of every instruction has a position).  Position -> (plane of the step, row, x) of the 2 x 10 x 34
      // window; its tensor byte offset is a constant of the lane, the column / step part rides in soffset; a position outside the
      // volume is steered past the descriptor's range and lands as zeros (tools/micro/lds_dma_oob.hip).
      const int pw = wave - NW;
      constexpr int NPOS = TZ * ZP;
      constexpr int NI = (NPOS + 63) / 64;
      const int rowB = a.W * 64;                                // bytes of a tensor row: 4 planes x W x 16
      const int biasB = ((TZ - 1) * a.H + 1) * rowB + 16;       // keeps (plane -(TZ - 1), row -1, x -1) non-negative
      unsigned l_off[NI];
      unsigned b_xlo = 0, b_xhi = 0, b_ylo = 0, b_yhi = 0, b_p0 = 0, b_p1 = 0;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int pos = (i < NI - 1 ? 64 * i : NPOS - 64) + lane;
        const int pz = pos / ZP, rem = pos - pz * ZP;
        const int r = rem / HX, hx = rem - r * HX;
        l_off[i] = (unsigned)(((pz - (TZ - 1)) * a.H + (r - 1)) * rowB + (hx - 1) * 16 + pw * a.W * 16 + biasB);
        if (hx == 0) b_xlo |= 1u << i;
        if (hx == HX - 1) b_xhi |= 1u << i;
        if (r == 0) b_ylo |= 1u << i;
        if (r == HY - 1) b_yhi |= 1u << i;
        if (pz < TZ - 1) b_p0 |= 1u << i;                       // step 0 of a column: planes -(TZ - 1) .. -1 do not exist
        else b_p1 |= 1u << i;                                   // step KZ: plane D does not exist
      }
      typedef int i32x4_ __attribute__((ext_vector_type(4)));
      const unsigned lds_plane = (unsigned)(unsigned long long)s_img + (unsigned)((pw >> 1) * OCT_B + (pw & 1) * PREC_B);
      const size_t sampleB = (size_t)a.D * a.H * rowB;
      bool cs_have = false;
      unsigned cs_bad = 0xFFFFFFFFu, cs_soff = 0;
      i32x4_ cs_srd = {0, 0, 0, 0x00020000};
      auto column_state = [&](int ci) {
        const bool have = ci < ncol_wg;
        int n = 0, ty = 0, tx = 0;
        if (have) col_of(ci, n, ty, tx);
        cs_have = have;
        unsigned bad = 0;
        if (tx == 0) bad |= b_xlo;
        if (tx == ka.tiles_x - 1) bad |= b_xhi;
        if (ty == 0) bad |= b_ylo;
        if (ty == ka.tiles_y - 1) bad |= b_yhi;
        cs_bad = bad;
        cs_soff = (unsigned)((ty * 8) * rowB + tx * 32 * 16);
        const unsigned long long base = (unsigned long long)a.in + (unsigned long long)n * sampleB - (unsigned long long)biasB;
        cs_srd[0] = (int)(unsigned)(base & 0xFFFFFFFFull);
        cs_srd[1] = (int)(unsigned)((base >> 32) & 0xFFFFull);
        cs_srd[2] = (int)VX_NUMREC;
        cs_srd[3] = 0x00020000;
      };
      auto dma = [&](const Cur& c, int grp) {
        if (c.s == 0) column_state(c.ci);
        if (!cs_have) return;                                   // (wave-uniform)
        unsigned bad = cs_bad;
        if (c.s == 0) bad |= b_p0;
        if (c.s == KZ) bad |= b_p1;
        const unsigned soff = cs_soff + (unsigned)(((TZ * c.s) * a.H) * rowB);
        const unsigned m0g = lds_plane + (unsigned)(grp * GRP_B);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const unsigned vo = ((bad >> i) & 1u) ? VX_OOB : l_off[i];
          const unsigned m0v = m0g + (unsigned)((i < NI - 1 ? 64 * i : NPOS - 64) * 16);
          asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(vo), "s"(cs_srd), "s"(soff) : "memory");
        }
      };
      Cur cx = {0, 0}, cc = {0, 0};      // visible / to stage
      dma(cc, 0); advance(cc);           // S_0 -> slot group 0
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int grp_x = 0;
      while (cx.ci < ncol_wg) {
        __syncthreads();
        ZC_STAMP(0);
        int grp_c = grp_x + 1; if (grp_c == 3) grp_c = 0;               // group S_{j+1} goes into: nobody reads it in this iteration
        if (cc.ci < ncol_wg && !(ZC_ABL & 4)) dma(cc, grp_c);
        ZC_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // landed before the barrier that makes the step visible
        ZC_STAMP(5);
#ifdef VX_CONV_STAMPS
        ++st_iters;
#endif
        advance(cx); advance(cc);
        grp_x = grp_c;
      }
    } else {
    constexpr int UPR = CIN == 16 ? 2 : 1;          // units per row
    constexpr int NU = TZ * HY * UPR;               // row units of a step
    constexpr int RPW = NU / NPW;                   // 10 (Cin = 16) / 5 (Cin = 8)
    static_assert(NU % NPW == 0 && HY % RPW == 0, "a wave's rows lie in one plane");
    constexpr int HR = (RPW + 7) / 8;               // hash rounds per step (eight rows each)
    constexpr int NH = TZ * HY * 2 * Q;             // halo pieces of a step
    constexpr int HIT = (NH + 63) / 64;
    static_assert(HIT <= NPW, "halo iterations");
    // this wave's rows: plane u_pz of the step, x-half u_h, rows u_hy0 .. u_hy0 + RPW - 1 of the 10-row window
    const int u0 = pw * RPW;                        // units ordered (plane, half, row)
    const int u_pz = u0 / (HY * UPR), u_h = CIN == 16 ? (u0 / HY) % UPR : 0, u_hy0 = u0 % HY;
    const int pitch = a.in_pitch;
    const int rowf = a.W * pitch;
    const int biasf = ((TZ - 1) * a.H + 1) * rowf + 4 * pitch;
    // lane -> (voxel of the unit, channel quad).  Cin = 16: lanes 0..31 carry quads 0, 1 (octet 0) of the unit's 16 voxels,
    // lanes 32..63 quads 2, 3 (octet 1): a half-wave writes 256 contiguous bytes of ONE octet plane (conflict-free ds_write_b64)
    const int l_v = CIN == 16 ? (lane & 31) >> 1 : lane >> 1;
    const int l_q = CIN == 16 ? (lane & 1) + 2 * (lane >> 5) : (lane & 1);
    const int l_dx = 16 * u_h + l_v;
    const unsigned l_voff = (unsigned)((l_dx * pitch + l_q * 4 + biasf) * 4);
    const int l_lds = (l_q >> 1) * OCT_B + (1 + l_dx) * 16 + (l_q & 1) * 8;
    // dropout keep-words (PRE = 1, 16 channels: one 32-bit word = 2 voxels): a unit row is 8 words; lane (i, j) computes word j
    // of row slot i, the row's lanes fetch theirs with ds_bpermute
    const int l_bp = 4 * (l_v >> 1);
    const unsigned l_sh = (unsigned)((l_v & 1) * 16 + l_q * 4);
    const int row0_dzy = (u_pz - (TZ - 1)) * a.H + (u_hy0 - 1);      // row offset (in tensor rows) of the wave's first row
    const int u_soff0 = row0_dzy * rowf * 4;
    const int u_lds0 = (u_pz * HY + u_hy0) * ROW_B;
    unsigned um_ylo = 0, um_yhi = 0;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      if (u_hy0 + i == 0) um_ylo |= 1u << i;
      if (u_hy0 + i == HY - 1) um_yhi |= 1u << i;
    }
    const unsigned um_all = (1u << RPW) - 1u;
    const unsigned um_zf = u_pz < TZ - 1 ? um_all : 0u;       // step 0 of a column: planes -(TZ-1) .. -1 do not exist
    const unsigned um_zl = u_pz == TZ - 1 ? um_all : 0u;      // step KZ: plane D does not exist
    int l_hw[PRE == 1 ? HR : 1];
    if constexpr (PRE == 1) {
#pragma unroll
      for (int rd = 0; rd < HR; ++rd) {
        const int i = 8 * rd + (lane >> 3);
        l_hw[rd] = (row0_dzy + i) * (a.W / 2) + 8 * u_h + (lane & 7);      // (row's first element) / 32 + word
      }
    }
    // ---- halo pieces: iteration hk of the step: lane -> (row of the step, side, quad)
    const int hk = NPW - 1 - pw;
    const bool has_halo = hk < HIT;
    unsigned h_voff = 0, h_erel = 0, h_flags = 1u;     // flags: 1 never a piece, 2 / 4 side x = -1 / 32, 8 / 16 y = -1 / 8,
    int h_lds = 0, h_q = 0;                            //        32 / 64 plane before the last / the last of the step
    if (has_halo) {
      const int hp = lane + 64 * hk;
      const int q = hp % Q, side = (hp / Q) & 1, r = hp / (2 * Q);
      const int pz = r / HY, hy = r % HY;
      const int dzy = (pz - (TZ - 1)) * a.H + (hy - 1);
      const int dx = side ? 32 : -1, hx = dx + 1;
      h_q = q;
      h_voff = (unsigned)((dzy * rowf + dx * pitch + q * 4 + biasf) * 4);
      h_lds = (q >> 1) * OCT_B + ((pz * HY + hy) * HX + hx) * 16 + (q & 1) * 8;
      h_erel = (unsigned)((dzy * a.W + dx) * CIN + q * 4);
      h_flags = (hp >= NH ? 1u : 0u) | (side ? 4u : 2u) | (hy == 0 ? 8u : 0u) | (hy == HY - 1 ? 16u : 0u) | (pz < TZ - 1 ? 32u : 64u);
    }
    const size_t in_sample = (size_t)a.D * a.H * rowf;

    // ---- register staging: the loads of one step ----
    f32x4 ibuf[RPW], hbuf = {0.f, 0.f, 0.f, 0.f};
    uint32_t fbuf[PRE == 3 ? RPW : 1], hfbuf = 0;          // pool-finish: the any-dropped word of every piece
    f32x4 p_mean = {0.f, 0.f, 0.f, 0.f}, p_rstd = {1.f, 1.f, 1.f, 1.f};
    f32x4 h_mean = {0.f, 0.f, 0.f, 0.f}, h_rstd = {1.f, 1.f, 1.f, 1.f};
    vx_dkey p_key = {0u, 0u};
    unsigned p_rowbad = 0xFFFFFFFFu, p_e0 = 0;
    bool p_hbad = true;

    // per-COLUMN state (recomputed at step 0 of a column)
    bool cs_have = false;
    unsigned cs_bad = 0xFFFFFFFFu, cs_hb = 0x7Fu, cs_e0 = 0;
    int cs_soff = 0;
    __amdgpu_buffer_rsrc_t cs_srd = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t cs_fsrd = cs_srd;
    auto column_state = [&](int ci) {
      const bool have = ci < ncol_wg;
      int n = 0, ty = 0, tx = 0;
      if (have) col_of(ci, n, ty, tx);
      cs_have = have;
      unsigned bad = 0;
      if (ty == 0) bad |= um_ylo;
      if (ty == ka.tiles_y - 1) bad |= um_yhi;
      cs_bad = bad;
      unsigned hb = 1u;
      if (tx == 0) hb |= 2u;
      if (tx == ka.tiles_x - 1) hb |= 4u;
      if (ty == 0) hb |= 8u;
      if (ty == ka.tiles_y - 1) hb |= 16u;
      cs_hb = hb;
      cs_soff = ((ty * 8) * rowf + tx * 32 * pitch) * 4;
      cs_srd = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(reinterpret_cast<const char*>(a.in) + ((size_t)n * in_sample - biasf) * 4), 0, VX_NUMREC, 0x00020000);
      if constexpr (PRE == 3)   // the flag words [N][D][H][W][2]: the input tensor's image at a quarter of every byte offset
        cs_fsrd = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(reinterpret_cast<const char*>(a.in_pool_flags) + ((ptrdiff_t)n * (ptrdiff_t)in_sample - biasf)), 0, VX_NUMREC, 0x00020000);
      if constexpr (PRE != 0) {
        // (n = 0 when the workgroup has run out of columns: a valid address, no branch around the loads)
        p_mean = *reinterpret_cast<const f32x4*>(a.in_mean + (size_t)n * CIN + l_q * 4);
        p_rstd = *reinterpret_cast<const f32x4*>(a.in_rstd + (size_t)n * CIN + l_q * 4);
        h_mean = *reinterpret_cast<const f32x4*>(a.in_mean + (size_t)n * CIN + h_q * 4);
        h_rstd = *reinterpret_cast<const f32x4*>(a.in_rstd + (size_t)n * CIN + h_q * 4);
      }
      if constexpr (PRE == 1) {
        cs_e0 = (unsigned)((ty * 8) * a.W + tx * 32) * (unsigned)CIN;
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
      const int soff = cs_soff + ((TZ * c.s) * a.H) * rowf * 4;
      const __amdgpu_buffer_rsrc_t srd = cs_srd;
      // NO branch may enclose a load (conv3d_xp8w.hip): a row outside the volume reads through an out-of-range offset
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int so = soff + u_soff0 + i * (rowf * 4);
        const bool rb = (bad >> i) & 1u;
        ibuf[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(rb ? VX_OOB : l_voff), rb ? 0 : so, 0));
        if constexpr (PRE == 3)
          fbuf[i] = __builtin_amdgcn_raw_buffer_load_b32(cs_fsrd, (int)(rb ? VX_OOB : (l_voff >> 2)), rb ? 0 : (so >> 2), 0);
      }
      p_rowbad = bad;
      {
        unsigned hb = cs_hb;
        if (c.s == 0) hb |= 32u;
        if (c.s == KZ) hb |= 64u;
        if (!have) hb = 0x7Fu;
        const bool lbad = (h_flags & hb) != 0u;      // waves without a halo iteration: flag 1 in every lane
        hbuf = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(lbad ? VX_OOB : h_voff), soff, 0));
        if constexpr (PRE == 3)
          hfbuf = __builtin_amdgcn_raw_buffer_load_b32(cs_fsrd, (int)(lbad ? VX_OOB : (h_voff >> 2)), soff >> 2, 0);
        p_hbad = lbad;
      }
      if constexpr (PRE == 1) p_e0 = cs_e0 + (unsigned)(((TZ * c.s) * a.H) * a.W) * (unsigned)CIN;
    };

    // the producing block's InstanceNorm + LeakyReLU + Dropout on one piece (conv3d_xp8w.hip: the keep bit ANDed into the
    // scale, dropout's factor 2 folded into it; (x - mean) first)
    auto pre_piece = [&](f32x4 v, const f32x4 mu, const f32x4 sc, uint32_t bits) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int keep = __builtin_amdgcn_sbfe((int)bits, j, 1);          // all ones / all zeros
        const float scj = __int_as_float(__float_as_int(sc[j]) & keep);
        const float t = vx_mul1(vx_sub1(v[j], mu[j]), scj);
        v[j] = vx_max1(t, vx_mul1(t, 0.01f));
      }
      return v;
    };
    // vx_pool_finish's arithmetic on one piece (same expressions, same order): the pooled tensor is never written
    auto poolfin_piece = [&](f32x4 v, const f32x4 mu, const f32x4 rs, uint32_t fl, bool outside, float s2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = vx_mul1(vx_sub1(v[j], mu[j]), rs[j]);
        float w = vx_mul1(vx_max1(t, vx_mul1(0.01f, t)), s2);
        if ((fl >> j) & 1u) w = vx_max1(w, 0.f);
        v[j] = outside ? 0.f : w;
      }
      return v;
    };
    auto split4 = [&](const f32x4 v, f16x4& hi, f16x4& lo) {
      vx_split4_s(v, hi, lo);      // plain instructions: the staging waves are this kernel's critical path (stamps: 96 % busy)
    };

    auto commit = [&](int grp) {
      const int gofs = grp * GRP_B;
      const bool hashed = PRE == 1 && a.in_drop_mode == VX_DROP_HASH;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, hsc = {1.f, 1.f, 1.f, 1.f};
      const float two = (PRE == 3 ? a.in_drop_mode == VX_DROP_HASH : hashed) ? 2.f : 1.f;
      uint32_t hw[PRE == 1 ? HR : 1];
      uint32_t wrow[PRE == 1 ? RPW : 1];
      if constexpr (PRE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[j] = vx_mul1(p_rstd[j], two); hsc[j] = vx_mul1(h_rstd[j], two); }
        if (hashed) {
#pragma unroll
          for (int rd = 0; rd < HR; ++rd) hw[rd] = vx_drop_word(p_key, (uint32_t)((int)(p_e0 >> 5) + l_hw[rd]));
#pragma unroll
          for (int i = 0; i < RPW; ++i)
            wrow[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(l_bp + 32 * (i & 7), (int)hw[i >> 3]);
        }
      }
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        unsigned char* dst = s_img + gofs + u_lds0 + i * ROW_B + l_lds;
        f32x4 v = ibuf[i];
        if constexpr (PRE == 1) {
          uint32_t bits = hashed ? (wrow[i] >> l_sh) : 0xFu;
          bits &= ((p_rowbad >> i) & 1u) ? 0u : 0xFu;        // zero padding belongs to the NORMALISED tensor
          v = pre_piece(v, p_mean, sc, bits);
        }
        if constexpr (PRE == 3) v = poolfin_piece(v, p_mean, p_rstd, fbuf[i], (p_rowbad >> i) & 1u, two);
        f16x4 hi, lo;
        if (ZC_ABL & 32) { hi = __builtin_bit_cast(f16x4, (f32x2){v[0], v[1]}); lo = __builtin_bit_cast(f16x4, (f32x2){v[2], v[3]}); }   // (diagnostic build: no split arithmetic)
        else split4(v, hi, lo);
        if (!(ZC_ABL & 16)) {
          *reinterpret_cast<f16x4*>(dst) = hi;
          *reinterpret_cast<f16x4*>(dst + PREC_B) = lo;
        } else {
          asm volatile("" :: "v"(hi), "v"(lo));
        }
      }
      if (has_halo && !(h_flags & 1u)) {
        f32x4 v = hbuf;                      // zeros where the piece lies outside the volume (out-of-range load)
        if constexpr (PRE == 1) {
          uint32_t bits = 0xFu;
          if (hashed) bits = vx_drop_bits4(p_key, p_e0 + h_erel);
          if (p_hbad) bits = 0u;
          v = pre_piece(v, h_mean, hsc, bits);
        }
        if constexpr (PRE == 3) v = poolfin_piece(v, h_mean, h_rstd, hfbuf, p_hbad, two);
        f16x4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<f16x4*>(s_img + gofs + h_lds) = hi;
        *reinterpret_cast<f16x4*>(s_img + gofs + h_lds + PREC_B) = lo;
      }
    };

    Cur cx = {0, 0}, cc = {0, 0}, cp = {0, 0};   // visible / to commit / to prefetch
    prefetch(cp); advance(cp);
    commit(0);    advance(cc);                   // S_0 -> slot group 0
    prefetch(cp); advance(cp);
    int grp_x = 0;
    while (cx.ci < ncol_wg) {
      __syncthreads();
      ZC_STAMP(0);
      int grp_c = grp_x + 1; if (grp_c == 3) grp_c = 0;               // group S_{j+1} goes into
      ZC_WAIT_LOADS();
      ZC_STAMP(3);
      if (cc.ci < ncol_wg && !(ZC_ABL & 4)) commit(grp_c);
      ZC_STAMP(4);
      if (!(ZC_ABL & 8)) prefetch(cp);
      ZC_STAMP(5);
#ifdef VX_CONV_STAMPS
      ++st_iters;
#endif
      advance(cx); advance(cc); advance(cp);
      grp_x = grp_c;
    }
    }   // UP == 0
    if (STATS) __syncthreads();
  } else {
    // =============================================== MULTIPLYING ===============================================
    const int lz = wave >> 2, xh = (wave >> 1) & 1, ly0 = (wave & 1) * R, x0 = 16 * xh;
    const bool late = wave >= NW / 2;
    // ---- compute-phase constants (byte offsets into the LDS image) ----
    // type-1 fragment of input row (slot, ly0 + j): lane (m, g) -> position x0 + m + kx, octet g & 1 (Cin = 16: kx = g >> 1;
    // Cin = 8: kx = g, the zero group g = 3 re-reads kx = 2: finite data under a zero weight)
    const int kxP = CIN == 16 ? (g >> 1) : (g < 2 ? g : 2);
    const int bP = (CIN == 16 ? (g & 1) * OCT_B : 0) + (ly0 * HX + x0 + m + kxP) * 16;
    const int bQ = (g & 1) * OCT_B + (ly0 * HX + x0 + m + 2) * 16;      // Cin = 16: the kx = 2 taps
    const unsigned char* wlane = s_w + lane * 16;

    // ---- epilogue constants: this lane stores voxel x0 + m of row ly0 + r, channels 4 g .. 4 g + 3 ----
    const int lx = x0 + m, oc = 4 * g;
    // (row r of the wave sits r output rows further: a scalar added to the store's soffset, one offset register per lane)
    unsigned ovoff0;
    if (a.out_xblk) {
      const int oxb = a.out_xblk;
      ovoff0 = (unsigned)((((lz * a.H + ly0) * (2 * a.W * 16)) + ((lx / oxb) * 2 + a.out_half) * oxb * 16 + (lx % oxb) * 16 + oc) * 4);
    } else {
      ovoff0 = (unsigned)((((lz * a.H + ly0) * a.W + lx) * a.out_pitch + a.out_coff + oc) * 4);
    }
    const unsigned orow = (unsigned)(a.W * (a.out_xblk ? 32 : a.out_pitch) * 4);   // bytes between two output rows
    // ONE hash round per item serves the wave's R rows (16 channels: a keep-word = 2 voxels, a half row = 8 words): lane i
    // computes word i & 7 of row i >> 3, a row's lanes fetch theirs with ds_bpermute.  Same bits as vx_drop_bits4(key, e).
    unsigned hword_l;
    {
      const int rr_ = (lane >> 3) < R ? (lane >> 3) : 0;
      hword_l = (unsigned)((((lz * a.H + ly0 + rr_) * a.W + x0) >> 1) + (lane & 7));
    }
    const int out_voxf = a.out_xblk ? 32 : a.out_pitch;
    const size_t out_sample = (size_t)a.D * a.H * a.W * out_voxf;
    const bool f_lrelu = EPI == 3 ? a.act == VX_ACT_LRELU : EPI == 1;
    const bool f_relu = EPI == 3 && a.act == VX_ACT_RELU;

    f32x4 acc[R], accx[R];
    f32x4 accin[ACC ? R : 1];
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned avoff0 = ACC ? (unsigned)((((lz * a.H + ly0) * a.W + lx) * a.acc_pitch + oc) * 4) : 0u;
    const unsigned arow = ACC ? (unsigned)(a.W * a.acc_pitch * 4) : 0u;
    const size_t acc_sample = (size_t)a.D * a.H * a.W * (ACC ? a.acc_pitch : 0);

    // the multiply phase of the item whose first input plane (z0 - 1) sits in slot rb (item k of column ci)
    auto multiply = [&](int rb, int ci, int k) {
      if constexpr (ACC != 0) {
        int n_, ty_, tx_;
        col_of(ci, n_, ty_, tx_);
        const unsigned asoff = (unsigned)(((k * TZ) * a.H + ty_ * 8) * a.W + tx_ * 32) * (unsigned)a.acc_pitch * 4u;
        const __amdgpu_buffer_rsrc_t asrd = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(const_cast<float*>(kernarg()->a.acc_in) + (size_t)n_ * acc_sample), 0, VX_NUMREC, 0x00020000);
#pragma unroll
        for (int r = 0; r < R; ++r)
          accin[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrd, (int)avoff0, (int)(asoff + (unsigned)r * arow), 0));
      }
      const f32x4 bias4 = *reinterpret_cast<const f32x4*>(s_bias + oc);
      int sl[3];
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        int s_ = rb + lz + kz;
        if (s_ >= NZ) s_ -= NZ;
        sl[kz] = s_ * PLN_B;
      }
      // phases = groups of steps that share their row fragments.  Cin = 16: phases 0..2 (type 1, kz = phase) and 3 (the kx = 2 taps
      // of kz = 0 / 1) have three steps (ky) over 6 rows -- tile r of step ky reads row r + ky; phase 4 is the ONE step that holds
      // the kx = 2 taps of (kz = 2, ky = 0 / 1) as its two k-group pairs: its fragment of tile r mixes rows r and r + 1 by lane
      // (4 fragments, no reuse); phase 5 the half-empty step of (kz = 2, ky = 2): rows 2..5.  14 steps for 13.5 (round 5, second
      // version: the first ran the kz = 2 taps as three half-empty steps, 15).  Cin = 8: three phases of three steps.
      constexpr int NPH = CIN == 16 ? 6 : 3;
      auto ph_of = [](int t) { return t < 12 ? t / 3 : t - 8; };           // step -> phase
      auto ky_of = [](int t) { return t < 12 ? t % 3 : 0; };               // step -> row offset of tile r within the phase's rows
      auto nrows = [](int ph) { return ph < 4 ? R + 2 : R; };
      const unsigned char* rowp[NPH];
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        if (ph < 3) rowp[ph] = s_img + bP + sl[ph];
        else if (ph == 3) rowp[ph] = s_img + bQ + ((g >> 1) ? sl[1] : sl[0]);
        else if (ph == 4) rowp[ph] = s_img + bQ + sl[2] + (g >> 1) * ROW_B;
        else rowp[ph] = s_img + bQ + sl[2] + 2 * ROW_B;
      }
#ifndef ZC_NO_PIPE
      if constexpr (!POOL)      // (the pooling instance holds 8 statistics registers + its window state across the loop: 168 + scratch)
      // ONE software pipeline over the NSTEP K-steps, 12 matrix instructions each.  Left alone hipcc sinks every ds_read_b128 to
      // just before its consumer (ds_read; s_waitcnt lgkmcnt(0..1); v_mfma).  Here every step requests, behind its own matrix
      // instructions and pinned there (sched_group_barrier), fragments of LATER steps:
      //   step (phase p, ky = 0): the weights of (p, 1) + rows 4, 5 of phase p          (needed at ky = 1 / ky = 2)
      //   step (p, 1):            the weights of (p, 2) + rows 0, 1 of phase p + 1
      //   step (p, 2):            the weights of (p + 1, 0) + rows 2, 3 of phase p + 1
      //   step 12 (phase 4):      the weights of step 13 + the four rows of phase 5
      // -- 0.5 reads per matrix instruction over the item.  Measured -2 .. -3.5 % (same-process A/B against the plain loop).
      {
        f16x8 rh[NPH][R + 2], rl[NPH][R + 2], wh[NSTEP], wl[NSTEP];
        auto ld_row = [&](int ph, int jr) {
          rh[ph][jr] = *reinterpret_cast<const f16x8*>(rowp[ph] + jr * ROW_B);
          rl[ph][jr] = *reinterpret_cast<const f16x8*>(rowp[ph] + jr * ROW_B + PREC_B);
        };
        auto ld_w = [&](int t) {
          wh[t] = *reinterpret_cast<const f16x8*>(wlane + t * 2048);
          wl[t] = *reinterpret_cast<const f16x8*>(wlane + t * 2048 + 1024);
        };
        ld_w(0);
#pragma unroll
        for (int jr = 0; jr < R; ++jr) ld_row(0, jr);
        __builtin_amdgcn_sched_group_barrier(0x100, 3 + 2 * R, 0);      // (+ the bias vector)
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {
          const int ph = ph_of(t), ky = ky_of(t);
          int nrd = 0;
          if (t + 1 < NSTEP) { ld_w(t + 1); nrd += 2; }
          if (t < 12) {
            if (ky == 0) { ld_row(ph, R); ld_row(ph, R + 1); nrd += 4; }
            else if (ph + 1 < NPH) { ld_row(ph + 1, 2 * (ky - 1)); ld_row(ph + 1, 2 * (ky - 1) + 1); nrd += 4; }
          } else if (t == 12) {
#pragma unroll
            for (int jr = 0; jr < R; ++jr) ld_row(5, jr);
            nrd += 2 * R;
          }
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const bool fresh = t == 0;      // the bias is the first product's C operand
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], rh[ph][r + ky], fresh ? (ACC ? zero : bias4) : acc[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], rl[ph][r + ky], fresh ? zero : accx[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], rh[ph][r + ky], accx[r], 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 3 * R; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < nrd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
        return;
      }
#endif
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        f16x8 bh[R + 2], bl[R + 2];
#pragma unroll
        for (int jr = 0; jr < R + 2; ++jr) {
          if (jr < nrows(ph)) {
            bh[jr] = *reinterpret_cast<const f16x8*>(rowp[ph] + jr * ROW_B);
            bl[jr] = *reinterpret_cast<const f16x8*>(rowp[ph] + jr * ROW_B + PREC_B);
          }
        }
#pragma unroll
        for (int ky = 0; ky < (ph < 4 ? 3 : 1); ++ky) {
          const int step = ph < 4 ? ph * 3 + ky : 8 + ph;
          const f16x8 ah = *reinterpret_cast<const f16x8*>(wlane + step * 2048);
          const f16x8 al = *reinterpret_cast<const f16x8*>(wlane + step * 2048 + 1024);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const bool fresh = ph == 0 && ky == 0;      // the bias is the first product's C operand
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[r + ky], fresh ? (ACC ? zero : bias4) : acc[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[r + ky], fresh ? zero : accx[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[r + ky], accx[r], 0, 0, 0);
          }
        }
      }
    };

    // ---- epilogue state of THIS wave (column it is storing) ----
    int e_ci = -1, e_n = 0, e_ty = 0, e_tx = 0;
    vx_dkey e_key = {0u, 0u};

    auto epilogue = [&](int ci, int k) {
      if (ci != e_ci) {
        e_ci = ci;
        col_of(ci, e_n, e_ty, e_tx);
        if (EPI == 1 || EPI == 4) e_key = vx_drop_key(seed_out, kernarg()->a.drop_layer, (uint32_t)e_n);
      }
      const unsigned vox0 = (unsigned)(((k * TZ) * a.H + e_ty * 8) * a.W + e_tx * 32);
      const unsigned osoff = a.out_xblk ? (unsigned)((((k * TZ) * a.H + e_ty * 8) * (2 * a.W * 16) + e_tx * 32 * 32) * 4)
                                        : vox0 * (unsigned)a.out_pitch * 4u;
      const int hbp = 4 * (m >> 1);                               // (recomputed per item: two registers less across the matrix loop)
      const unsigned hsh = (unsigned)((m & 1) * 16 + oc);
      const bool e_hash = (EPI == 1) || (EPI == 4 && a.drop_mode == VX_DROP_HASH);
      const uint32_t hw_item = e_hash ? vx_drop_word(e_key, (vox0 >> 1) + hword_l) : 0u;
      const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(reinterpret_cast<char*>(kernarg()->a.out) + (size_t)e_n * out_sample * 4), 0, VX_NUMREC, 0x00020000);
      float pl_max[4];
      uint32_t pl_any = 0u;
      __amdgpu_buffer_rsrc_t psrd = osrd, fsrd = osrd;
      if constexpr (POOL) {
        const auto kp = kernarg();
        const size_t pvs = (size_t)a.D * (a.H >> 1) * (a.W >> 1);      // pooled (y, x) voxels per sample
        psrd = __builtin_amdgcn_make_buffer_rsrc((void*)(kp->a.pool_out + (size_t)e_n * pvs * 16), 0, VX_NUMREC, 0x00020000);
        fsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(kp->a.pool_flags + (size_t)e_n * pvs * 4), 0, VX_NUMREC, 0x00020000);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        f32x4 v;       // main + cross * 2^-11: one fma per element (exact scaling)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(accx[r][j], 1.0f / 2048.f, acc[r][j]);
        if constexpr (ACC != 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += accin[r][j];
        }
        if (STATS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { ssum[j] += v[j]; ssq[j] = fmaf(v[j], v[j], ssq[j]); }
        }
        if constexpr (POOL) {
          uint32_t bits = 0xFu;
          if (e_hash) bits = ((uint32_t)__builtin_amdgcn_ds_bpermute(hbp + 32 * r, (int)hw_item) >> hsh) & 0xFu;
          pl_any = (r & 1) ? (pl_any | (~bits & 0xFu)) : (~bits & 0xFu);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float kept = ((bits >> j) & 1u) ? v[j] : -INFINITY;
            pl_max[j] = (r & 1) ? fmaxf(pl_max[j], kept) : kept;
          }
        }
        if (f_lrelu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.01f * v[j]);
        } else if (f_relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (EPI == 1) {
          const uint32_t bits = ((uint32_t)__builtin_amdgcn_ds_bpermute(hbp + 32 * r, (int)hw_item) >> hsh) & 0xFu;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);
        }
        if (!STATS) rmax = fmaxf(fmaxf(rmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        u32x4 sv = __builtin_bit_cast(u32x4, v);
        if (EPI == 1 || EPI == 3) {
          if constexpr (PLN) {
            // the consumer stages by LDS-DMA: its image rows are [octet][hi | lo][x][8 halves] -- this lane's 4 channels are one half
            // of a 16-byte entry of the hi plane and of the lo plane of octet g >> 1 (the lane pair g, g ^ 1 fills the entry)
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            f16x4 hi, lo;
            vx_split4_s(v, hi, lo);
            const unsigned rowB = (unsigned)a.W * 64u, plB = (unsigned)a.W * 16u;
            int ln = lane;
            asm volatile("" : "+v"(ln));          // (keeps this address arithmetic inside the epilogue: one register less across the matrix loop)
            const int g_ = ln >> 4, m_ = ln & 15;
            const unsigned pvo = (unsigned)(lz * a.H + ly0) * rowB + (unsigned)((g_ >> 1) * 2) * plB + (unsigned)(x0 + m_) * 16u + (unsigned)(g_ & 1) * 8u;
            const unsigned pso = (unsigned)((k * TZ) * a.H + e_ty * 8 + r) * rowB + (unsigned)(e_tx * 32) * 16u;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), osrd, (int)pvo, (int)pso, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), osrd, (int)pvo, (int)(pso + plB), 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 3" ::: "memory");      // (the store-data hazard with an SGPR soffset, below)
            __builtin_amdgcn_sched_barrier(0);
            continue;
          }
          if (a.out_split) {   // the consumer is the fused up-convolution: hand the piece over as the fp16 pairs it multiplies
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            f16x4 hi, lo;
            vx_split4_s(v, hi, lo);      // (plain instructions: packed fp32 beside the partner wave's matrix stream costs three)
            const u32x2 h2 = __builtin_bit_cast(u32x2, hi), l2 = __builtin_bit_cast(u32x2, lo);
            sv = (u32x4){h2[0], h2[1], l2[0], l2[1]};
          }
        }
        __builtin_amdgcn_raw_buffer_store_b128(sv, osrd, (int)ovoff0, (int)(osoff + (unsigned)r * orow), 0);
        // gfx950 store-data hazard with an SGPR soffset (conv3d_mfma.hip)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 3" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (POOL) {
          if (r & 1) {
            // a row pair is complete.  The x-neighbour (same channels) is the adjacent lane: quad_perm [1, 0, 3, 2]; even lanes
            // store the window of their z-plane: pool_out [N][D][H/2][W/2][16], pool_flags [N][D][H/2][W/2][4] (the flag word of a
            // 16-byte piece sits at a quarter of the piece's byte offset); odd lanes are steered out of the descriptor's range
            const int Hp = a.H >> 1, Wp = a.W >> 1;
            u32x4 mx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(pl_max[j]), 0xB1, 0xF, 0xF, true);
              mx[j] = __float_as_uint(fmaxf(pl_max[j], __int_as_float(o)));
            }
            const uint32_t oany = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pl_any, 0xB1, 0xF, 0xF, true);
            // lane part: ((lz Hp + ly0 / 2) Wp + lx / 2) 64 + 16 g bytes; scalar part: item, column, row pair
            const unsigned pvo = (m & 1) ? VX_OOB : (unsigned)((((lz * Hp + (ly0 >> 1)) * Wp + (lx >> 1)) * 16 + oc) * 4);
            const unsigned pso = (unsigned)(((((k * TZ) * Hp + e_ty * 4 + (r >> 1)) * Wp) + e_tx * 16) * 64);
            __builtin_amdgcn_raw_buffer_store_b128(mx, psrd, (int)pvo, (int)pso, 0);
            __builtin_amdgcn_raw_buffer_store_b32(pl_any | oany, fsrd, (int)((m & 1) ? VX_OOB : (pvo >> 2)), (int)(pso >> 2), 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 3" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (STATS && k == KZ - 1) {
        // the column is complete for this wave: sum over its 16 voxel columns and leave the 4 x 2 values of row group g
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

    // statistics of a complete column: entry 0 of the column's block is real, the other stat_epc - 1 are zero
    // (vx_instnorm_finalize sums vx_conv3d_k3_tiles_for entries per sample)
    auto flush_col = [&](int ci) {
      int n, ty, tx;
      col_of(ci, n, ty, tx);
      const auto kp = kernarg();
      const int epc = kp->stat_epc;
      const int ntile = cps * epc;
      float* dst = kp->a.stats_partial + (((size_t)n * ntile + (size_t)(ty * ka.tiles_x + tx) * epc) * 16) * 2;
      if (tid < 16) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          s += s_red[(w * 16 + tid) * 2 + 0];
          q += s_red[(w * 16 + tid) * 2 + 1];
        }
        dst[tid * 2 + 0] = s;
        dst[tid * 2 + 1] = q;
      }
      for (int i = 32 + tid; i < epc * 32; i += 256) {
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
      ZC_STAMP(0);
      if (STATS && fl_ci >= 0 && j >= fl_at && !late) { flush_col(fl_ci); fl_ci = -1; }
      const bool comp = cx.s >= 1;
      const int item_k = cx.s - 1;
      int grp_c = grp_x + 1; if (grp_c == 3) grp_c = 0;
      const int rb = (grp_x == 0 ? 2 : grp_x - 1) * TZ + (TZ - 2);           // first plane of the item: group of S_{j-1}, plane TZ - 2
      if (late) {
        if (prev_ci >= 0 && !(ZC_ABL & 2)) { epilogue(prev_ci, prev_k); prev_ci = -1; }
        ZC_STAMP(2);
        if (comp) { if (!(ZC_ABL & 1)) multiply(rb, cx.ci, item_k); prev_ci = cx.ci; prev_k = item_k; }
        ZC_STAMP(1);
      } else {
        if (comp && !(ZC_ABL & 1)) multiply(rb, cx.ci, item_k);
        ZC_STAMP(1);
        if (comp && !(ZC_ABL & 2)) epilogue(cx.ci, item_k);
        ZC_STAMP(2);
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
  if (!STATS && a.range_flag) {
    float mx = rmax;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
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
bool vx_conv3d_zc16_packs(int Cin, int Cout) { return Cout == 16 && (Cin == 8 || Cin == 16); }

bool vx_conv3d_zc16_applies(int D, int H, int W, int Cin, int Cout) {
  if (vx_cfg().conv_fp32 != 0 || vx_cfg().s16_no_zc16) return false;
  // D >= 4: a column has at least two items (the statistics hand-off between the wave halves needs the spacing)
  return vx_conv3d_zc16_packs(Cin, Cout) && W % 32 == 0 && H % 8 == 0 && D % 2 == 0 && W >= 32 && H >= 8 && D >= 4;
}

int64_t vx_conv3d_zc16_packed_floats(int Cin, int Cout) {
  if (!vx_conv3d_zc16_packs(Cin, Cout)) return 0;
  return (int64_t)(Cin == 16 ? 14 : 9) * 2 * 64 * 8 / 2;
}

// torch (16, Cin, 3,3,3) fp32 -> [step][hi | lo][lane 64][8 halves] in the K schedule of the kernel's header
__global__ void pack_conv3d_zc16_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int Cin, int total) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63, hl = (i >> 9) & 1, step = i >> 10;
    const int row = lane & 15, g = lane >> 4;
    int kz, ky, kx, ci;
    bool zero = false;
    if (Cin == 16) {
      ci = 8 * (g & 1) + j;
      if (step < 9) { kz = step / 3; ky = step % 3; kx = g >> 1; }
      else if (step < 12) { ky = step - 9; kz = g >> 1; kx = 2; }
      else if (step == 12) { ky = g >> 1; kz = 2; kx = 2; }
      else { ky = 2; kz = 2; kx = 2; zero = (g >> 1) != 0; }
    } else {
      ci = j; kz = step / 3; ky = step % 3; kx = g; zero = g == 3;
    }
    const float v = zero ? 0.f : w[((size_t)row * Cin + ci) * 27 + kz * 9 + ky * 3 + kx];
    const float c = fminf(fmaxf(v, -65504.f), 65504.f);
    const _Float16 h = (_Float16)c;
    out[i] = hl == 0 ? h : (_Float16)((v - (float)h) * 2048.f);
  }
}

int vx_pack_conv3d_zc16(const float* w_torch, float* w_packed, int Cin, int Cout, hipStream_t s) {
  const int total = (int)(vx_conv3d_zc16_packed_floats(Cin, Cout) * 2);
  if (total <= 0) return VX_OK;
  hipLaunchKernelGGL(pack_conv3d_zc16_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w_torch, reinterpret_cast<_Float16*>(w_packed), Cin,
                     total);
  VX_CHECK_LAUNCH("vx_pack_conv3d_k3(zc16)");
  return VX_OK;
}

// ConvTranspose3d(32 -> 16, k = 2, s = 2) for the fused evaluation: torch (32, 16, 2, 2, 2) -> [class (dz, dy, dx)][hi | lo][lane][8 halves],
// lane (m, g): row m = co, k = ci 8 g .. 8 g + 7 (the A operand of v_mfma_f32_16x16x32_f16)
__global__ void pack_convT_zc16_kernel(const float* __restrict__ w, _Float16* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 8 * 2 * 64 * 8) return;
  const int j = i & 7, lane = (i >> 3) & 63, hl = (i >> 9) & 1, cls = i >> 10;
  const int co = lane & 15, ci = 8 * (lane >> 4) + j;
  const float v = w[(ci * 16 + co) * 8 + cls];
  const float c = fminf(fmaxf(v, -65504.f), 65504.f);
  const _Float16 h = (_Float16)c;
  out[i] = hl == 0 ? h : (_Float16)((v - (float)h) * 2048.f);
}
extern "C" int64_t vx_convT_zc16_packed_floats(void) { return 8 * 2 * 64 * 8 / 2; }
extern "C" int vx_pack_convT_zc16(const float* w_torch, float* packed, vx_stream_t stream) {
  if (!w_torch || !packed) VX_FAIL(VX_E_NULL, "vx_pack_convT_zc16: null pointer");
  if (!vx_aligned16(packed)) VX_FAIL(VX_E_ALIGN, "vx_pack_convT_zc16: packed must be 16-byte aligned");
  hipLaunchKernelGGL(pack_convT_zc16_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, w_torch, reinterpret_cast<_Float16*>(packed));
  VX_CHECK_LAUNCH("vx_pack_convT_zc16");
  return VX_OK;
}

template <int CIN, int EPI, int PRE, int ACC = 0, int UP = 0>
static int launch_zc16(const Zc16Args& ka, hipStream_t s) {
  constexpr int PP = ((6 * 340 + 15) / 16) * 16;
  constexpr size_t lds = (size_t)(CIN / 8) * 2 * PP * 16 + (size_t)(CIN == 16 ? 14 : 9) * 2048 + 8 * 16 * 2 * 4 + 64;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = conv3d_zc16_kernel<CIN, EPI, PRE, ACC, UP>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv3d_k3(zc16): hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr = true;
  }
  int gx = vx_cu_count();      // one persistent workgroup per CU
  if (gx > ka.ncols) gx = ka.ncols;
  static const char* kname = vx_kname("conv3d_zc16_kernel<%d,%d,%d,%d,%d>", CIN, EPI, PRE, ACC, UP);   // as rocprofv3 prints it
  vx_note_kernel(kname);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(768), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv3d_k3(zc16)");
  return VX_OK;
}

// 1 = not taken (the caller uses the general tile kernel)
int vx_conv3d_k3_zc16(const vx_conv3d_args& a, const float* w_block, int stat_tiles, hipStream_t s) {
  if (a.in_xblk || a.head_out || a.in_split || a.in_f16 || a.out_f16 || (a.in_repeat > 1)) return 1;
  if ((a.in_planar || a.out_planar) && (a.drop_mode == VX_DROP_MASK || !a.out || a.in_pitch != a.Cin || a.Cout != 16))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(zc16): the planar pre-split hand-over needs hash or no dropout and dense 16-channel tensors");
  if (a.up_in && (a.Cin != 16 || a.in_mean || a.stats_partial || a.up_pitch < 32 || a.up_pitch % 4 || a.up_fused))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(zc16): the fused up-convolution (32 coarse channels -> the conv's 16 input channels) goes with an "
            "activation epilogue, no prologue, up_pitch >= 32 (got %d), up_w packed by vx_pack_convT_zc16", a.up_pitch);
  if (a.drop_mode == VX_DROP_MASK || a.in_drop_mode == VX_DROP_MASK) return 1;
  if (!a.out) return 1;
  Zc16Args ka;
  ka.a = a;
  ka.w = w_block;
  ka.tiles_x = a.W / 32; ka.tiles_y = a.H / 8; ka.kz = a.D / 2;
  const int cps = ka.tiles_x * ka.tiles_y;
  ka.ncols = a.N * cps;
  ka.mcps = (unsigned)((1ull << 32) / (unsigned)cps) + 1u;
  ka.mtx = (unsigned)((1ull << 32) / (unsigned)ka.tiles_x) + 1u;
  ka.stat_epc = stat_tiles / cps;
  ka.stamps = nullptr;
  ka.abl = 0;
#ifdef VX_CONV_STAMPS
  if (const char* e = getenv("VX_CONV_DBG_PTR")) ka.stamps = (unsigned long long*)strtoull(e, nullptr, 0);
  if (const char* e = getenv("VX_XP_ABL")) ka.abl = atoi(e);
#endif
  if ((int64_t)a.N * cps >= (1ll << 31)) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(zc16): too many columns");
  if (a.in_pitch != a.Cin) return 1;
  if (a.stats_partial && (stat_tiles % cps || a.act != VX_ACT_NONE || (a.drop_mode != VX_DROP_NONE && !a.pool_out)))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(zc16): statistics go with a plain epilogue");
  if (a.pool_out && (!a.stats_partial || a.Cin != 16)) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(zc16): the pooled output goes with statistics, 16 -> 16");
  int pre = 0;
  if (a.in_pool_flags) pre = 3;
  else if (a.in_mean) pre = 1;
  if ((pre == 1 && a.Cin != 16) || (pre == 3 && a.Cin != 8)) return 1;
  if (a.in_planar) {
    if (a.Cin != 16 || pre != 0 || a.up_in || a.acc_in || a.stats_partial || !vx_aligned16(a.in) ||
        (int64_t)a.D * a.H * a.W * 64 >= (1ll << 31))
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(zc16): a planar pre-split input (in_planar) goes with 16 -> 16, an activation epilogue, no "
              "prologue / up-convolution / partial sums, a 16-byte aligned tensor, one sample below 2 GiB");
    pre = 4;
  }
  if (a.out_planar && (a.stats_partial || a.out_pitch != 16 || a.out_coff != 0 || a.out_xblk || a.out_split || a.out == a.acc_in ||
                       (int64_t)a.D * a.H * a.W * 64 >= (1ll << 31)))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(zc16): a planar pre-split output (out_planar) goes with an activation epilogue into a dense "
            "16-channel tensor that is not the partial sums' (another layout), one sample below 2 GiB");
  int epi;
  if (a.stats_partial) epi = a.pool_out ? 4 : 0;
  else if (a.drop_mode == VX_DROP_HASH) { if (a.act != VX_ACT_LRELU) return 1; epi = 1; }
  else epi = 3;
  if (a.out_split && a.stats_partial) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(zc16): out_split goes with the activation epilogues");
  if (a.out_planar) {
    // the instances that exist: the decoder's up-half launch (partial sums + fused up-convolution) and the plain 16 -> 16 layer
    if (a.Cin != 16 || pre != 0 || (epi != 1 && epi != 3) || ((a.acc_in != nullptr) != (a.up_in != nullptr)))
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(zc16): out_planar goes with 16 -> 16, an activation epilogue, no prologue, and either partial "
              "sums + the fused up-convolution together or neither");
    epi = epi == 1 ? 5 : 6;
  }
  if (a.acc_in) {
    if (a.Cin != 16 || pre != 0 || a.stats_partial || a.acc_pitch < 16 || a.acc_pitch % 4 || !vx_aligned16(a.acc_in) ||
        (int64_t)a.D * a.H * a.W * a.acc_pitch * 4 >= (1ll << 31))
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(zc16): partial sums (acc_in) go with 16 -> 16, no prologue, an activation epilogue, a "
              "16-byte aligned tensor of pitch >= 16 (pitch %d), one sample below 2 GiB", a.acc_pitch);
    if (a.up_in) {
      if (epi == 5) return launch_zc16<16, 5, 0, 1, 1>(ka, s);
      if (epi == 6) return launch_zc16<16, 6, 0, 1, 1>(ka, s);
      if (epi == 1) return launch_zc16<16, 1, 0, 1, 1>(ka, s);
      return launch_zc16<16, 3, 0, 1, 1>(ka, s);
    }
    if (epi == 1) return launch_zc16<16, 1, 0, 1>(ka, s);
    return launch_zc16<16, 3, 0, 1>(ka, s);
  }
  if (a.up_in) {
    if (epi == 1) return launch_zc16<16, 1, 0, 0, 1>(ka, s);
    if (epi == 3) return launch_zc16<16, 3, 0, 0, 1>(ka, s);
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(zc16): no fused up-convolution for this epilogue");
  }
#define ZC16_CASE(C_, E_, P_) if (a.Cin == C_ && epi == E_ && pre == P_) return launch_zc16<C_, E_, P_>(ka, s)
  ZC16_CASE(16, 0, 0); ZC16_CASE(16, 0, 1); ZC16_CASE(16, 4, 0); ZC16_CASE(16, 4, 1);
  ZC16_CASE(16, 1, 0); ZC16_CASE(16, 1, 1); ZC16_CASE(16, 3, 0); ZC16_CASE(16, 3, 1);
  ZC16_CASE(16, 1, 4); ZC16_CASE(16, 3, 4); ZC16_CASE(16, 5, 0); ZC16_CASE(16, 6, 0);
  ZC16_CASE(8, 0, 0); ZC16_CASE(8, 0, 3); ZC16_CASE(8, 3, 0); ZC16_CASE(8, 1, 0);
#undef ZC16_CASE
  return 1;
}
