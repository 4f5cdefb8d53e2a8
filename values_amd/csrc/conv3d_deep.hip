// K1 for the DEEP layers (round 5): Cout a multiple of 32 on volumes of 16^3 voxels and below -- contr_3/4, center.0/2,
// expand_4/3 of the reference's UNet3D (unet3D_module.py:81-120, 231-243, 263-267), 20 % of the 64^3 step on the tile kernel.
// There a workgroup owns 256 voxels x 32 output channels and every wave stages AND multiplies in lockstep: per 16-channel input
// chunk it re-stages 57 KB of weights for 1 344 matrix instructions, and the conversion of the staged tile sits on the critical
// path (tools/stamp_s16.py: 25-47 % of an item outside the matrix phase on these layers).  Here the roles are split as in
// conv3d_xp8w.hip / conv3d_zc16.hip, on a plain 3-D tile instead of a z-column (the volumes are too small for a column walk):
//
//   * waves 0..7 (MULTIPLYING) own R column tiles of 16 voxels x 2 row tiles of 16 output channels each: a workgroup tile is
//     128 R voxels (R = 4: 512 -- 16 x 8 x 4 of a 16^3 volume, a whole 8^3 sample; R = 2: 256 -- half an 8^3 sample, FOUR 4^3
//     samples) x 32 output channels; accumulators live across the Cin / 8 items of a tile;
//   * waves 8..11 (STAGING) commit item j + 1 -- the 8-channel chunk of the tile's halo window, split into fp16 hi / lo, with
//     the optional normalise-on-load prologue -- into the other LDS buffer while item j is multiplied, move that item's 28 KB
//     of weights there by LDS-DMA, then issue the loads of item j + 2.  One barrier per item.
//
// K = 32 step: four taps x 8 channels (k-group g = tap & 3; 27 taps in 7 steps, the 28th has zero weights), B fragment = one
// ds_read_b128 per precision at (the column's halo position + the tap's offset), A fragment = one per (row tile, precision):
// 12 reads per 24 matrix instructions.  Per item a multiplying wave issues 42 R matrix instructions and nothing else; the
// epilogue runs once per tile.
// LDS: two images [hi | lo][1080 positions][8 halves] (67.5 KB), two weight chunks [step 7][row tile 2][hi | lo][lane][8
// halves] (56 KB), statistics slots and the bias vector.
#include "s16_common.h"

struct DeepArgs {
  vx_conv3d_args a;
  const float* w;                 // this kernel's block of the packed weights (vx_conv3d_deep_packed_floats)
  int tx, ty, tz, ts;             // tile: voxels along x, y, z (powers of two); samples (> 1 only when a tile is whole samples)
  int ltx, lty, ltz;              // their base-2 logarithms
  int tiles_x, tiles_y, tiles_z;
  int npos;                       // halo positions of a tile: ts (tx + 2)(ty + 2)(tz + 2)
  int nchunks, ncg;               // 8-channel input chunks; 32-row output groups
  int npairs;                     // (tile, output group) pairs in the launch
  unsigned m_cg, m_tx, m_ty, m_tz;   // multiply-high magics of the pair decode
  int stat_epc;                   // statistics entries per tile in stats_partial (entry 0 real, the rest zero)
  unsigned long long* stamps;
  int abl;
};

#ifdef VX_CONV_STAMPS
#define DP_STAMP(i)                                                                      \
  do {                                                                                   \
    unsigned long long t_;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    st_sum[i] += t_ - st_last;                                                           \
    st_last = t_;                                                                        \
  } while (0)
#define DP_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define DP_ABL ka.abl
#else
#define DP_ABL 0
#define DP_STAMP(i) do {} while (0)
#define DP_WAIT_LOADS() do {} while (0)
#endif

constexpr int DEEP_NPOS = 1080;                    // positions of one image (18 x 10 x 6)
constexpr int DEEP_PREC_B = DEEP_NPOS * 16;
constexpr int DEEP_IMG_B = 2 * DEEP_PREC_B;
constexpr int DEEP_W_B = 7 * 2 * 2 * 1024;         // one chunk's weights
constexpr int DEEP_MAXC = 512;                     // output channels the bias vector in LDS holds

// R: column tiles per multiplying wave (4 / 2).  EPI: 0 bias + statistics + store, 1 LeakyReLU + hash dropout (+ out_split),
// 3 run-time activation without dropout.  PRE: 1 = InstanceNorm + LeakyReLU + dropout of the producing block on load (dense
// input, one sample per tile).
template <int R, int EPI, int PRE>
__global__ __launch_bounds__(768) void conv3d_deep_kernel(DeepArgs ka) {
  constexpr int NW = 8, NPW = 4, NTH = (NW + NPW) * 64, NST = NPW * 64;
  constexpr int NT = 2, NSTEP = 7;
  constexpr int IN_IT = R == 4 ? 9 : 7;            // 16-byte pieces per staging thread and item (2 npos / 256, rounded up)
  constexpr int W_IT = DEEP_W_B / 16 / NST;        // 7
  constexpr bool STATS = EPI == 0;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* s_img = smem_raw;
  unsigned char* s_w = smem_raw + 2 * DEEP_IMG_B;
  float* s_red = reinterpret_cast<float*>(s_w + 2 * DEEP_W_B);        // [tile parity 2][NW][32][2]
  float* s_bias = s_red + 2 * NW * 32 * 2;                            // [Cout]

  const vx_conv3d_args& a = ka.a;
  auto kernarg = [&]() {      // fields used once per tile are re-read where they are used (conv3d_xp8w.hip)
    typedef const DeepArgs __attribute__((address_space(4))) * kp_t;
    kp_t p = (kp_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
  };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int TX = ka.tx, TY = ka.ty, TZ = ka.tz;
  const int HX = TX + 2, HY = TY + 2, HZ = TZ + 2;
  const int NCH = ka.nchunks;

  for (int i = tid; i < a.Cout; i += NTH) s_bias[i] = a.bias[i];

  // ---- the (tile, output group) pairs of this workgroup ----
  int vb = blockIdx.x;
  const int G = (int)gridDim.x;
  if ((G & 7) == 0) vb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);   // one XCD: neighbouring pairs (the groups of one tile)
  const int npair_wg = vb < ka.npairs ? (ka.npairs - vb + G - 1) / G : 0;

  struct Cur { int ci, c; };      // pair number of this workgroup, chunk
  auto advance = [&](Cur& x) { if (++x.c == NCH) { x.c = 0; ++x.ci; } };
  auto pair_of = [&](int ci, int& n0, int& tzi, int& tyi, int& txi, int& cg) {
    unsigned t = (unsigned)(vb + ci * G), q;
    q = ka.ncg == 1 ? t : __umulhi(t, ka.m_cg); cg = (int)(t - q * (unsigned)ka.ncg); t = q;
    q = ka.tiles_x == 1 ? t : __umulhi(t, ka.m_tx); txi = (int)(t - q * (unsigned)ka.tiles_x); t = q;
    q = ka.tiles_y == 1 ? t : __umulhi(t, ka.m_ty); tyi = (int)(t - q * (unsigned)ka.tiles_y); t = q;
    q = ka.tiles_z == 1 ? t : __umulhi(t, ka.m_tz); tzi = (int)(t - q * (unsigned)ka.tiles_z);
    n0 = (int)q * ka.ts;
  };

  float rmax = 0.f;   // largest |value| this wave stored (range guard of the split-fp16 consumers)
#ifdef VX_CONV_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last, st_iters = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif

  // pipeline (both roles): item S_j is visible after barrier j (buffer j & 1); in iteration j the staging waves commit S_{j+1}
  // into the other buffer and load S_{j+2}, the multiplying waves compute S_j.
  if (wave >= NW) {
    // =============================================== STAGING ===============================================
    // raised priority where the staging waves are the critical path -- with the prologue, and on the small tile (half the matrix
    // work per item for the same weight chunk): same-process A/B at 320 samples -3 % / -1.5 .. -3 %; +4 % on the large plain tile
    if constexpr (PRE != 0 || R == 2) __builtin_amdgcn_s_setprio(2);
    const int t = tid - NW * 64;
    const int q = t & 1;                           // the thread's 4-channel quad of the chunk (256 threads: constant over its pieces)
    const int xb = a.in_xblk;
    const int Csrc = xb ? a.Cin / 2 : a.Cin;
    const int voxf = xb ? 2 * Csrc : a.in_pitch;
    const int rowf = a.W * voxf;
    const int biasf = (a.H + 1) * rowf + 4 * voxf;
    const size_t in_sample = (size_t)a.D * a.H * rowf;
    const int cper = xb ? Csrc / 8 : 0;
    const int lastx = (ka.tiles_x - 1) * TX, lasty = (ka.tiles_y - 1) * TY, lastz = (ka.tiles_z - 1) * TZ;
    unsigned voff[IN_IT];
    int ldst[IN_IT];
    unsigned ibad_always = 0, ibad_xlo = 0, ibad_xhi = 0, ibad_ylo = 0, ibad_yhi = 0, ibad_zlo = 0, ibad_zhi = 0;
    unsigned ibad_sge[3] = {0, 0, 0};              // pieces of tile samples >= 1, 2, 3: the last tile of a batch that is no multiple of ts
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const int idx = t + it * NST;
      int vox = idx >> 1;
      if (vox >= ka.npos) { ibad_always |= 1u << it; vox = 0; }
      const int hx = vox % HX, r1 = vox / HX;
      const int hy = r1 % HY, r2 = r1 / HY;
      const int hz = r2 % HZ, smp = r2 / HZ;
      const int dxr = hx - 1, dyr = hy - 1, dzr = hz - 1;
      int xf;
      if (xb) {
        const int blk = dxr >= 0 ? dxr / xb : -((-dxr + xb - 1) / xb);
        const int rem = dxr - blk * xb;
        xf = (blk * 2) * xb * Csrc + rem * Csrc + 4 * q;
      } else {
        xf = dxr * a.in_pitch + 4 * q;
      }
      voff[it] = (unsigned)(((size_t)smp * in_sample + (size_t)((dzr * a.H + dyr) * rowf + xf + biasf)) * 4);
      ldst[it] = vox * 16 + q * 8;
      if (dxr < 0) ibad_xlo |= 1u << it;
      if (dxr >= a.W - lastx) ibad_xhi |= 1u << it;
      if (dyr < 0) ibad_ylo |= 1u << it;
      if (dyr >= a.H - lasty) ibad_yhi |= 1u << it;
      if (dzr < 0) ibad_zlo |= 1u << it;
      if (dzr >= a.D - lastz) ibad_zhi |= 1u << it;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (smp > k) ibad_sge[k] |= 1u << it;
    }
    const uint32_t seed_in = PRE == 1 ? vx_seed_of(a, a.in_drop_seed) : 0u;

    f32x4 ibuf[IN_IT];
    f32x4 p_mean = {0.f, 0.f, 0.f, 0.f}, p_rstd = {1.f, 1.f, 1.f, 1.f};
    unsigned p_bad = 0xFFFFFFFFu, p_e0 = 0;
    vx_dkey p_key = {0u, 0u};
    // per-PAIR state (recomputed at chunk 0)
    unsigned cs_bad = 0xFFFFFFFFu, cs_soff = 0;
    int cs_n = 0;
    __amdgpu_buffer_rsrc_t cs_srd = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0, 0x00020000);
    auto pair_state = [&](int ci) {
      const bool have = ci < npair_wg;
      int n0 = 0, tzi = 0, tyi = 0, txi = 0, cg = 0;
      if (have) pair_of(ci, n0, tzi, tyi, txi, cg);
      unsigned bad = ibad_always;
      if (txi == 0) bad |= ibad_xlo;
      if (txi == ka.tiles_x - 1) bad |= ibad_xhi;
      if (tyi == 0) bad |= ibad_ylo;
      if (tyi == ka.tiles_y - 1) bad |= ibad_yhi;
      if (tzi == 0) bad |= ibad_zlo;
      if (tzi == ka.tiles_z - 1) bad |= ibad_zhi;
      // samples past the batch (ts <= 4 samples per tile, host-checked): their pieces read zeros, their voxels are not stored --
      // a sample's numbers do not depend on how many batch mates share its tile
      const int nv = a.N - n0;
      if (nv < ka.ts) bad |= ibad_sge[nv >= 1 ? nv - 1 : 0];
      if (!have) bad = 0xFFFFFFFFu;
      cs_bad = bad;
      cs_soff = (unsigned)((((tzi * TZ) * a.H + tyi * TY) * rowf + txi * TX * voxf) * 4);
      cs_n = n0;
      cs_srd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)n0 * in_sample - biasf), 0, VX_NUMREC, 0x00020000);
      if constexpr (PRE == 1) p_key = vx_drop_key(seed_in, a.in_drop_layer, (uint32_t)n0);
    };
    auto prefetch = [&](const Cur& x) {
      if (x.c == 0) pair_state(x.ci);              // (a wave-uniform branch BEFORE the loads, nothing in flight at the join)
      const unsigned bad = cs_bad;
      const int chunk = x.c;
      int coff;
      if (!xb) coff = chunk * 8;
      else coff = (chunk / cper) * xb * Csrc + (chunk % cper) * 8;
      const unsigned soff = cs_soff + (unsigned)(coff * 4);
      // NO branch may enclose a load (conv3d_xp8w.hip): a piece outside the volume reads through an out-of-range offset
#pragma unroll
      for (int it = 0; it < IN_IT; ++it) {
        const unsigned vo = ((bad >> it) & 1u) ? VX_OOB : voff[it];
        ibuf[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cs_srd, (int)vo, (int)soff, 0));
      }
      p_bad = bad;
      if constexpr (PRE == 1) {
        p_e0 = (soff >> 2) - (unsigned)biasf;
        // (cs_n = 0 when the workgroup has run out of pairs: a valid address)
        p_mean = *reinterpret_cast<const f32x4*>(a.in_mean + (size_t)cs_n * a.Cin + chunk * 8 + q * 4);
        p_rstd = *reinterpret_cast<const f32x4*>(a.in_rstd + (size_t)cs_n * a.Cin + chunk * 8 + q * 4);
      }
    };
    // The chunk's weights go to LDS by LDS-DMA (buffer_load_dwordx4 ... lds: 28 consecutive KiB of global memory -> 28 consecutive KiB
    // of LDS, no layout change, no registers, no ds_write): 7 instructions per staging wave, issued in the iteration that commits the
    // item, landed before its closing barrier.  Against seven global loads + seven ds_write_b128 per thread (13 LDS-path cycles each):
    // -2 .. -4 % on every instance (tools/ab_layers.py, same process).
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    const unsigned lds_w0 = (unsigned)(unsigned long long)s_w;
    int wd_cg = 0;
    auto dma_w = [&](const Cur& x, int buf) {
      if (x.ci >= npair_wg) return;
      if (x.c == 0) { int n0, tzi, tyi, txi; pair_of(x.ci, n0, tzi, tyi, txi, wd_cg); }
      const unsigned long long base = (unsigned long long)(ka.w) + (unsigned long long)(wd_cg * NCH + x.c) * DEEP_W_B;
      i32x4_ srd;
      srd[0] = (int)(unsigned)(base & 0xFFFFFFFFull);
      srd[1] = (int)(unsigned)((base >> 32) & 0xFFFFull);
      srd[2] = DEEP_W_B;
      srd[3] = 0x00020000;
      const int pw_ = wave - NW;
#pragma unroll
      for (int i = 0; i < W_IT; ++i) {
        const unsigned kb = (unsigned)((pw_ * W_IT + i) * 1024);
        const unsigned m0v = lds_w0 + (unsigned)(buf * DEEP_W_B) + kb;
        const unsigned vo = kb + (unsigned)lane * 16u;
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(m0v), "v"(vo), "s"(srd) : "memory");
      }
    };
    auto commit = [&](int buf) {
      unsigned char* img = s_img + buf * DEEP_IMG_B;
      const bool hashed = PRE == 1 && a.in_drop_mode == VX_DROP_HASH;
      f32x4 sc = p_rstd;
      if constexpr (PRE == 1) {
        if (hashed) {
#pragma unroll
          for (int j = 0; j < 4; ++j) sc[j] = vx_mul1(p_rstd[j], 2.f);      // dropout's factor 2 rides in the scale
        }
      }
#pragma unroll
      for (int it = 0; it < IN_IT; ++it) {
        if (!((ibad_always >> it) & 1u)) {
          f32x4 v = ibuf[it];
          if constexpr (PRE == 1) {
            uint32_t bits = 0xFu;
            if (hashed) bits = vx_drop_bits4(p_key, p_e0 + (voff[it] >> 2));
            if ((p_bad >> it) & 1u) bits = 0u;           // zero padding belongs to the NORMALISED tensor
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int keep = __builtin_amdgcn_sbfe((int)bits, j, 1);          // all ones / all zeros
              const float scj = __int_as_float(__float_as_int(sc[j]) & keep);
              const float tt = vx_mul1(vx_sub1(v[j], p_mean[j]), scj);
              v[j] = vx_max1(tt, vx_mul1(tt, 0.01f));
            }
          }
          f16x4 hi, lo;
          vx_split4_s(v, hi, lo);
          if (!(DP_ABL & 16)) {
            *reinterpret_cast<f16x4*>(img + ldst[it]) = hi;
            *reinterpret_cast<f16x4*>(img + ldst[it] + DEEP_PREC_B) = lo;
          } else {
            asm volatile("" :: "v"(hi), "v"(lo));
          }
        }
      }
    };

    Cur cx = {0, 0}, cc = {0, 0}, cp = {0, 0};   // visible / to commit / to prefetch
    prefetch(cp); advance(cp);
    dma_w(cc, 0);
    commit(0);    advance(cc);                   // S_0 -> buffer 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    prefetch(cp); advance(cp);
    int j = 0;
    while (cx.ci < npair_wg) {
      __syncthreads();
      DP_STAMP(0);
      DP_WAIT_LOADS();
      DP_STAMP(3);
      dma_w(cc, (j + 1) & 1);
      if (cc.ci < npair_wg && !(DP_ABL & 4)) commit((j + 1) & 1);
      DP_STAMP(4);
      if (!(DP_ABL & 8)) prefetch(cp);
      // the DMA is older than the IN_IT image loads just issued: wait until only those are outstanding
      if constexpr (IN_IT == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      DP_STAMP(5);
#ifdef VX_CONV_STAMPS
      ++st_iters;
#endif
      advance(cx); advance(cc); advance(cp);
      ++j;
    }
    if (STATS) __syncthreads();
  } else {
    // =============================================== MULTIPLYING ===============================================
    // column tile r of this wave: voxels (wave R + r) 16 + m of the tile, ordered (sample, z, y, x)
    int vbase[R];                  // byte offset of the voxel's halo position at tap (0, 0, 0)
    const int out_voxf = a.out_pitch;
    const size_t out_sample = (size_t)a.D * a.H * a.W * out_voxf;
    // (tile extents are powers of two: a voxel's coordinates are shifts and masks -- the epilogue recomputes them per tile
    // instead of holding three more registers per column tile across the matrix loop)
    auto vox_of = [&](int v, int& cx_, int& ly, int& lz, int& smp) {
      cx_ = v & (TX - 1); v >>= ka.ltx;
      ly = v & (TY - 1); v >>= ka.lty;
      lz = v & (TZ - 1); smp = v >> ka.ltz;
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int cx_, ly, lz, smp;
      vox_of((wave * R + r) * 16 + m, cx_, ly, lz, smp);
      vbase[r] = (((smp * HZ + lz) * HY + ly) * HX + cx_) * 16;
    }
    int toff[NSTEP];               // this lane's tap of every step, as a byte offset (the 28th tap re-reads the 27th: zero weights, finite data)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      int tap = 4 * s + g;
      if (DP_ABL & 64) tap = 4 * s;              // (diagnostic build: every k-group reads the same positions -- no bank conflicts)
      if (tap > 26) tap = 26;
      toff[s] = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * 16;
    }
    const unsigned char* wlane = s_w + lane * 16;
    const bool f_lrelu = EPI == 3 ? a.act == VX_ACT_LRELU : EPI == 1;
    const bool f_relu = EPI == 3 && a.act == VX_ACT_RELU;
    const uint32_t seed_out = EPI == 1 ? vx_seed_of(a, a.drop_seed) : 0u;

    f32x4 acc[R][NT], accx[R][NT];

    // ONE software pipeline over the 7 K-steps of an item, 6 R matrix instructions each, row tile by row tile.  Every fragment of
    // step s + 1 is requested at least 11 matrix instructions before its first use, into registers that are free by then, and
    // pinned there (sched_group_barrier) -- left alone hipcc sinks every ds_read_b128 to just before its consumer and the wave
    // waits out an LDS latency per tile:
    //   hi halves of the B fragments (used by both products of a row tile): a second register set, requested behind the first
    //     row tile's matrix instructions;
    //   lo halves (one product per row tile): the same registers, each behind its last use in the second row tile;
    //   A fragments of a row tile: behind that row tile's last matrix instruction.
    // Within a row tile the two products into the cross accumulator sit R matrix instructions apart (no dependent pair).
    auto multiply = [&](int buf) {
      const unsigned char* img = s_img + buf * DEEP_IMG_B;
      const unsigned char* wb = wlane + buf * DEEP_W_B;
      f16x8 wh[NT], wl[NT], bh[2][R], bl[R];
      int vb_[R];
#pragma unroll
      for (int r = 0; r < R; ++r) { vb_[r] = vbase[r]; asm volatile("" : "+v"(vb_[r])); }   // (the 7 R address sums stay inside the item: not 28 registers across the loop)
      auto ldA = [&](int s, int nt) {
        wh[nt] = *reinterpret_cast<const f16x8*>(wb + ((s * NT + nt) * 2) * 1024);
        wl[nt] = *reinterpret_cast<const f16x8*>(wb + ((s * NT + nt) * 2 + 1) * 1024);
      };
      auto ldBh = [&](int s, int r) { bh[s & 1][r] = *reinterpret_cast<const f16x8*>(img + vb_[r] + toff[s]); };
      auto ldBl = [&](int s, int r) { bl[r] = *reinterpret_cast<const f16x8*>(img + vb_[r] + toff[s] + DEEP_PREC_B); };
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) ldA(0, nt);
#pragma unroll
      for (int r = 0; r < R; ++r) { ldBh(0, r); ldBl(0, r); }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT + 2 * R, 0);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const bool more = s + 1 < NSTEP;
        const int set = s & 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], bh[set][r], acc[r][nt], 0, 0, 0);
            accx[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], bl[r], accx[r][nt], 0, 0, 0);
            if (more && nt == 0) ldBh(s + 1, r);
            if (more && nt == NT - 1) ldBl(s + 1, r);
          }
#pragma unroll
          for (int r = 0; r < R; ++r) accx[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], bh[set][r], accx[r][nt], 0, 0, 0);
          if (more) ldA(s + 1, nt);
        }
#ifndef DEEP_NO_PIPE
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (more) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, R, 0);
          if (more) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#endif
      }
    };

    auto epilogue = [&](int ci) {
      float* red = s_red + (ci & 1) * (NW * 32 * 2);
      int n0, tzi, tyi, txi, cg;
      pair_of(ci, n0, tzi, tyi, txi, cg);
      const auto kp = kernarg();
      const unsigned vox0 = (unsigned)(((tzi * TZ) * a.H + tyi * TY) * a.W + txi * TX);
      const unsigned osoff = (vox0 * (unsigned)out_voxf + (unsigned)(cg * 32)) * 4u;
      const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(reinterpret_cast<char*>(kp->a.out) + (size_t)n0 * out_sample * 4), 0, VX_NUMREC, 0x00020000);
      float ssum[NT][4], ssq[NT][4];
      if (STATS) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) { ssum[nt][j] = 0.f; ssq[nt][j] = 0.f; }
      }
      int m_ = m;
      asm volatile("" : "+v"(m_));      // (keeps the per-tile address arithmetic below inside the epilogue)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int cx_, ly, lz, smp;
        vox_of((wave * R + r) * 16 + m_, cx_, ly, lz, smp);
        const int ovox = (lz * a.H + ly) * a.W + cx_;
        // (32-bit offsets: the host checked that a tile's samples stay below 2 GiB)
        unsigned ovoff_r = ((unsigned)smp * (unsigned)out_sample + (unsigned)ovox * (unsigned)out_voxf + (unsigned)kp->a.out_coff + 4u * g) * 4u;
        if constexpr (R == 2) {                         // (only the small tile holds several samples: the large one never fits 4^3 x 8)
          if (n0 + smp >= a.N) ovoff_r = VX_OOB;        // a tile sample past the batch: the store is dropped
        }
        // ONE keep-word per voxel: the workgroup's 32 output channels are bits 0..31 of word voxel * (Cout / 32) + cg
        uint32_t hword = 0;
        if (EPI == 1) {
          const vx_dkey key = vx_drop_key(seed_out, kp->a.drop_layer, (uint32_t)(n0 + smp));
          hword = vx_drop_word(key, (vox0 + (unsigned)ovox) * (unsigned)ka.ncg + (unsigned)cg) >> (4 * g);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          f32x4 v;       // main + cross * 2^-11: one fma per element (exact scaling)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaf(accx[r][nt][j], 1.0f / 2048.f, acc[r][nt][j]);
          if (STATS) {
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
          if (EPI == 1) {
            const uint32_t bits = hword >> (16 * nt);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);
          }
          if (!STATS) rmax = vx_max3abs(vx_max3abs(rmax, v[0], v[1]), v[2], v[3]);
          u32x4 sv = __builtin_bit_cast(u32x4, v);
          if (EPI == 1 || EPI == 3) {
            if (a.out_split) {   // the consumer is a fused up-convolution: hand the piece over as the fp16 pairs it multiplies
              typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
              f16x4 hi, lo;
              vx_split4_s(v, hi, lo);
              const u32x2 h2 = __builtin_bit_cast(u32x2, hi), l2 = __builtin_bit_cast(u32x2, lo);
              sv = (u32x4){h2[0], h2[1], l2[0], l2[1]};
            }
          }
          __builtin_amdgcn_raw_buffer_store_b128(sv, osrd, (int)ovoff_r, (int)(osoff + (unsigned)(nt * 64)), 0);
          // gfx950 store-data hazard with an SGPR soffset (conv3d_mfma.hip)
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 3" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (STATS) {
        // sum over the wave's 16 voxel columns and leave the 4 x 2 values of (row tile, row group g)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float s = ssum[nt][j], qq = ssq[nt][j];
#pragma unroll
            for (int rot = 8; rot >= 1; rot >>= 1) { s += vx_row_ror(s, rot); qq += vx_row_ror(qq, rot); }
            if (m == 0) {
              red[(wave * 32 + nt * 16 + g * 4 + j) * 2 + 0] = s;
              red[(wave * 32 + nt * 16 + g * 4 + j) * 2 + 1] = qq;
            }
          }
      }
    };

    // statistics of a complete tile: entry 0 of the tile's block is real, the other stat_epc - 1 are zero
    // (vx_instnorm_finalize sums vx_conv3d_k3_tiles_for entries per sample)
    auto flush_tile = [&](int ci) {
      const float* red = s_red + (ci & 1) * (NW * 32 * 2);
      int n0, tzi, tyi, txi, cg;
      pair_of(ci, n0, tzi, tyi, txi, cg);
      const auto kp = kernarg();
      const int epc = kp->stat_epc;
      const int tps = ka.tiles_x * ka.tiles_y * ka.tiles_z;
      const int tile = (tzi * ka.tiles_y + tyi) * ka.tiles_x + txi;
      float* dst = kp->a.stats_partial + (((size_t)n0 * tps + tile) * epc * a.Cout + cg * 32) * 2;
      if (tid < 32) {
        float s = 0.f, qq = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          s += red[(w * 32 + tid) * 2 + 0];
          qq += red[(w * 32 + tid) * 2 + 1];
        }
        dst[tid * 2 + 0] = s;
        dst[tid * 2 + 1] = qq;
      }
      for (int e = 1; e < epc; ++e)
        if (tid < 64) dst[(size_t)e * a.Cout * 2 + tid] = 0.f;
    };

    // waves 0..3 multiply(j) then store a finished tile; waves 4..7 store the tile they finished in the previous iteration FIRST,
    // then multiply(j): the partner wave of a SIMD multiplies while the other runs the tile's epilogue
    const bool late = wave >= NW / 2;
    Cur cx = {0, 0};
    int j = 0;
    int prev_ci = -1;                             // waves 4..7: the tile still to store
    int fl_ci = -1, fl_at = 0;                    // tile whose statistics are complete in s_red after barrier fl_at
    while (cx.ci < npair_wg) {
      __syncthreads();
      DP_STAMP(0);
      if (STATS && fl_ci >= 0 && j >= fl_at && !late) { flush_tile(fl_ci); fl_ci = -1; }
      if (late && prev_ci >= 0) {
        if (!(DP_ABL & 2)) epilogue(prev_ci);      // (raised priority for it: measured nothing)
        prev_ci = -1;
      }
      DP_STAMP(2);
      if (cx.c == 0) {
        int n0, tzi, tyi, txi, cg;
        pair_of(cx.ci, n0, tzi, tyi, txi, cg);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_bias + cg * 32 + nt * 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < R; ++r) { acc[r][nt] = b4; accx[r][nt] = zero; }
        }
      }
      if (!(DP_ABL & 1)) multiply(j & 1);
      DP_STAMP(1);
      if (cx.c == NCH - 1) {
        if (late) prev_ci = cx.ci;
        else if (!(DP_ABL & 2)) epilogue(cx.ci);
        if (STATS) { fl_ci = cx.ci; fl_at = j + 2; }
      }
      DP_STAMP(2);
#ifdef VX_CONV_STAMPS
      ++st_iters;
#endif
      advance(cx);
      ++j;
    }
    if (late && prev_ci >= 0) epilogue(prev_ci);
    if (STATS) {
      __syncthreads();
      if (fl_ci >= 0 && !late) flush_tile(fl_ci);
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
bool vx_conv3d_deep_packs(int Cin, int Cout) { return Cout % 32 == 0 && Cin % 8 == 0 && Cin >= 16 && Cout <= DEEP_MAXC; }

int64_t vx_conv3d_deep_packed_floats(int Cin, int Cout) {
  if (!vx_conv3d_deep_packs(Cin, Cout)) return 0;
  return (int64_t)(Cout / 32) * (Cin / 8) * (DEEP_W_B / 4);
}

// torch (Cout, Cin, 3,3,3) fp32 -> [output group of 32][chunk of 8][step 7][row tile 2][hi | lo][lane 64][8 halves]:
// lane (m, g) of row tile nt holds row 32 cg + 16 nt + m, tap 4 step + g (zero beyond 26), channels 8 chunk .. + 7
__global__ void pack_conv3d_deep_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int Cin, int64_t total) {
  const int nch = Cin / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), hl = (int)((i >> 9) & 1), nt = (int)((i >> 10) & 1);
    const int64_t sb = i >> 11;
    const int step = (int)(sb % 7);
    const int64_t blk = sb / 7;
    const int chunk = (int)(blk % nch), cg = (int)(blk / nch);
    const int row = cg * 32 + nt * 16 + (lane & 15), tap = 4 * step + (lane >> 4);
    const float v = tap > 26 ? 0.f : w[((size_t)row * Cin + 8 * chunk + j) * 27 + tap];
    const float c = fminf(fmaxf(v, -65504.f), 65504.f);
    const _Float16 h = (_Float16)c;
    out[i] = hl == 0 ? h : (_Float16)((v - (float)h) * 2048.f);
  }
}

int vx_pack_conv3d_deep(const float* w_torch, float* w_packed, int Cin, int Cout, hipStream_t s) {
  const int64_t total = vx_conv3d_deep_packed_floats(Cin, Cout) * 2;
  if (total <= 0) return VX_OK;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_conv3d_deep_kernel, dim3(blocks), dim3(256), 0, s, w_torch, reinterpret_cast<_Float16*>(w_packed), Cin, total);
  VX_CHECK_LAUNCH("vx_pack_conv3d_k3(deep)");
  return VX_OK;
}

// Tile of a layer: 128 R voxels, x first (the whole row up to 16), then y (up to 8), then z; a tile that is a whole sample
// takes further samples (the last tile of a batch may hold fewer: the kernel masks them).  false = no tile of this size.
// The geometry is a function of the VOLUME's shape and the channel count only, never of the batch size: a sample's bits must not
// depend on its batch mates (round-5 advice; tests/test_gpu_kernels.py::test_conv3d_deep_is_batch_independent).
struct DeepGeo { int tx, ty, tz, ts, tiles_x, tiles_y, tiles_z, npos; };
static bool deep_geo(int D, int H, int W, int R, DeepGeo* o) {
  const int vox = 128 * R;
  const int tx = W < 16 ? W : 16;
  if (tx < 4 || (tx & (tx - 1)) || W % tx) return false;
  int ty = H < 8 ? H : 8;
  while (tx * ty > vox) ty >>= 1;
  if (ty < 1 || H % ty) return false;
  int tz = vox / (tx * ty);
  if (tz > D) tz = D;
  if (tz < 1 || D % tz) return false;
  int ts = vox / (tx * ty * tz);
  if (ts * tx * ty * tz != vox) return false;
  if ((ty & (ty - 1)) || (tz & (tz - 1))) return false;      // (the kernel decodes a tile's voxels with shifts)
  if (ts > 1 && (tx != W || ty != H || tz != D || ts > 4 || R != 2)) return false;   // (the kernel masks a partial last tile for R = 2 only)
  const int npos = ts * (tx + 2) * (ty + 2) * (tz + 2);
  if (npos > (R == 4 ? DEEP_NPOS : 896)) return false;      // the image's positions; 2 npos pieces in IN_IT x 256
  o->tx = tx; o->ty = ty; o->tz = tz; o->ts = ts;
  o->tiles_x = W / tx; o->tiles_y = H / ty; o->tiles_z = D / tz; o->npos = npos;
  return true;
}

// the layer's tile: R = 4 wherever it fits, else R = 2.  (Until round 5 the choice also looked at how the batch filled the 256
// persistent workgroups -- R = 2 for small batches at 8^3 -- which made the fp32 tile sums behind InstanceNorm, and so a sample's
// bits, a function of the batch size.  At 8^3 x 64 channels and 320 samples the large tile is 2 % faster anyway; at 16^3 the small
// tile costs 20 %.)
static int deep_pick(int D, int H, int W, int Cout, DeepGeo* o) {
  DeepGeo g4, g2;
  const bool ok4 = deep_geo(D, H, W, 4, &g4), ok2 = deep_geo(D, H, W, 2, &g2);
  if (!ok4 && !ok2) return 0;
  const int r = ok4 ? 4 : 2;
  *o = r == 4 ? g4 : g2;
  return r;
}

bool vx_conv3d_deep_applies(int N, int D, int H, int W, int Cin, int Cout) {
  if (vx_cfg().conv_fp32 != 0 || vx_cfg().s16_no_deep) return false;
  if (!vx_conv3d_deep_packs(Cin, Cout)) return false;
  if (W > 32 || H > 32 || D > 32) return false;        // (the larger layers keep the tile kernel's XCD-ordered small tiles)
  DeepGeo g;
  return deep_pick(D, H, W, Cout, &g) != 0;
}

template <int R, int EPI, int PRE>
static int launch_deep(const DeepArgs& ka, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)DEEP_IMG_B + 2 * (size_t)DEEP_W_B + 2 * 8 * 32 * 2 * 4 + DEEP_MAXC * 4;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = conv3d_deep_kernel<R, EPI, PRE>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv3d_k3(deep): hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr = true;
  }
  int gx = vx_cu_count();      // one persistent workgroup per CU
  if (gx > ka.npairs) gx = ka.npairs;
  static const char* kname = vx_kname("conv3d_deep_kernel<%d,%d,%d>", R, EPI, PRE);   // as rocprofv3 prints it
  vx_note_kernel(kname);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(768), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv3d_k3(deep)");
  return VX_OK;
}

// 1 = not taken (the caller uses the general tile kernel)
int vx_conv3d_k3_deep(const vx_conv3d_args& a, const float* w_block, int stat_tiles, hipStream_t s) {
  if (a.head_out || a.in_split || a.in_f16 || a.out_f16 || (a.in_repeat > 1) || a.out_xblk || a.up_in || a.pool_out ||
      a.in_pool_flags || a.acc_in)
    return 1;
  if (a.drop_mode == VX_DROP_MASK || a.in_drop_mode == VX_DROP_MASK) return 1;
  if (!a.out) return 1;
  DeepGeo g;
  const int R = deep_pick(a.D, a.H, a.W, a.Cout, &g);
  if (!R) return 1;
  if (a.in_xblk) {
    const int csrc = a.Cin / 2;
    if (a.in_mean || csrc % 8 || a.W % a.in_xblk) return 1;
  } else if (a.in_pitch % 4) {
    return 1;
  }
  if (a.in_mean && (a.in_pitch != a.Cin || g.ts != 1)) return 1;
  if (a.stats_partial && g.ts != 1) return 1;
  DeepArgs ka;
  ka.a = a;
  ka.w = w_block;
  ka.tx = g.tx; ka.ty = g.ty; ka.tz = g.tz; ka.ts = g.ts;
  ka.ltx = __builtin_ctz((unsigned)g.tx); ka.lty = __builtin_ctz((unsigned)g.ty); ka.ltz = __builtin_ctz((unsigned)g.tz);
  ka.tiles_x = g.tiles_x; ka.tiles_y = g.tiles_y; ka.tiles_z = g.tiles_z;
  ka.npos = g.npos;
  ka.nchunks = a.Cin / 8;
  ka.ncg = a.Cout / 32;
  const int tps = g.tiles_x * g.tiles_y * g.tiles_z;
  const int64_t npairs = (int64_t)((a.N + g.ts - 1) / g.ts) * tps * ka.ncg;
  if (npairs >= (1ll << 31)) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(deep): too many tiles");
  ka.npairs = (int)npairs;
  auto magic = [](int d) { return (unsigned)((1ull << 32) / (unsigned)d) + 1u; };
  ka.m_cg = magic(ka.ncg); ka.m_tx = magic(g.tiles_x); ka.m_ty = magic(g.tiles_y); ka.m_tz = magic(g.tiles_z);
  ka.stat_epc = 0;
  if (a.stats_partial) {
    // the caller's partials buffer holds stat_tiles entries per sample (the tile kernel's count): this kernel's tiles must divide it,
    // else the tile kernel takes the launch (round-5 advice: D % 4 == 2 volumes gave 3 tiles against 4 entries and failed here)
    if (stat_tiles % tps) return 1;
    if (a.act != VX_ACT_NONE || a.drop_mode != VX_DROP_NONE)
      VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(deep): statistics go with a plain epilogue");
    ka.stat_epc = stat_tiles / tps;
  }
  ka.stamps = nullptr;
  ka.abl = 0;
#ifdef VX_CONV_STAMPS
  if (const char* e = getenv("VX_CONV_DBG_PTR")) ka.stamps = (unsigned long long*)strtoull(e, nullptr, 0);
  if (const char* e = getenv("VX_XP_ABL")) ka.abl = atoi(e);
#endif
  if ((int64_t)g.ts * a.D * a.H * a.W * a.out_pitch * 4 >= (1ll << 31))
    VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(deep): one tile's samples must stay below 2 GiB");
  const int pre = a.in_mean ? 1 : 0;
  int epi;
  if (a.stats_partial) epi = 0;
  else if (a.drop_mode == VX_DROP_HASH) { if (a.act != VX_ACT_LRELU) return 1; epi = 1; }
  else epi = 3;
  if (a.out_split && a.stats_partial) VX_FAIL(VX_E_SHAPE, "vx_conv3d_k3(deep): out_split goes with the activation epilogues");
#define DEEP_CASE(R_, E_, P_) if (R == R_ && epi == E_ && pre == P_) return launch_deep<R_, E_, P_>(ka, s)
  DEEP_CASE(4, 0, 0); DEEP_CASE(4, 0, 1); DEEP_CASE(4, 1, 0); DEEP_CASE(4, 3, 0);
  DEEP_CASE(2, 0, 0); DEEP_CASE(2, 0, 1); DEEP_CASE(2, 1, 0); DEEP_CASE(2, 3, 0);
#undef DEEP_CASE
  return 1;
}
