// K1 for the Cout = 8 layers (contr_1_2, expand_1_1, expand_1_2: 45 % of the 64^3 network's time): the same 3x3x3
// convolution as conv3d_mfma.hip, on v_mfma_f32_4x4x1_16b_f32 -- 16 independent 4x4 outer products per
// instruction, the only fp32 matrix shape whose M is not wasted on 8 output channels (the 16-row tiles of
// conv3d_mfma.hip's x-pair packing run 25 % structural zeros through the pipe).  Measured layout (tools/micro/
// mfma_4x4.hip): block b = lane / 4;  A_b[i] comes from lane 4b + i, B_b[j] from lane 4b + j, and D_b[i][j] lands in
// lane 4b + j, register i; cbsz = 4 / abid = s broadcasts block s's A to all 16 blocks.
//
//   lane      = one output voxel (B operand: that voxel's input at (tap, channel); D: its 4 + 4 output channels in
//               two accumulators), so the epilogue stores whole 32-byte voxels and all 8 channels of a voxel are
//               lane-local (dropout bits from ONE hash, no cross-lane traffic);
//   weights   = one dword per lane per (chunk of 8 input channels, tap) holds 8 channels x 8 couts -- lane (b, i)
//               keeps W[cout 4 (b >> 3) + i][cin 8 chunk + (b & 7)][tap] -- and the MFMA for channel c and cout half
//               h selects it with abid = 8 h + c.  Cin = 8: 27 VGPRs for the whole kernel, no weight traffic at all;
//               Cin = 16: 54 dwords per lane in LDS, one conflict-free ds_read_b32 per tap;
//   input     = LDS image [channel quad q][halo position][4 floats]: the lanes of a wave are consecutive x, so the
//               ds_read_b128 of (tap, q) is conflict-free for every tap and the tap is an immediate offset; one
//               16-byte read feeds 8 MFMAs (64 cycles of the pipe), read three reads ahead.
//
// Work items, staging (global -> registers -> LDS, prefetched one item ahead), tile shapes, the epilogue's fused
// bias / LeakyReLU / dropout / statistics and the stats partial layout are those of conv3d_mfma.hip.
// Conv-shaped inner loop alone: 145 TFLOP/s of useful work (tools/micro/mfma_4x4.hip) against 111 for x-pair.
#include "common.h"
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define VX_NUMREC 0xFFFFF000u   // buffer descriptors cover "everything"; VX_OOB is beyond it
#define VX_OOB 0xFFFFF800u

struct ConvC8Args {
  vx_conv3d_args a;
  int tiles_x, tiles_y, tiles_z;
  unsigned mx, my, mz;   // ceil(2^32 / tiles_*) for the tile decode
  int dbg;               // DIAGNOSTIC BUILD ONLY (-DVX_CONV_STAMPS, env VX_C8_DBG): phase ablation, wrong results by design
  unsigned long long* stamps;  // VX_CONV_STAMPS diagnostic builds only
};

#ifdef VX_CONV_STAMPS
#define C8_DBG ka.dbg
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
#define C8_DBG 0
#define VX_STAMP(i) do {} while (0)
#endif

// 8 MFMAs of one 16-byte read: channels 4 Q .. 4 Q + 3 of the chunk, both cout halves
template <int Q>
__device__ __forceinline__ void c8_quad(f32x4& a0, f32x4& a1, float w, f32x4 x) {
  a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x[0], a0, 4, 4 * Q + 0, 0);
  a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x[0], a1, 4, 8 + 4 * Q + 0, 0);
  a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x[1], a0, 4, 4 * Q + 1, 0);
  a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x[1], a1, 4, 8 + 4 * Q + 1, 0);
  a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x[2], a0, 4, 4 * Q + 2, 0);
  a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x[2], a1, 4, 8 + 4 * Q + 2, 0);
  a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x[3], a0, 4, 4 * Q + 3, 0);
  a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x[3], a1, 4, 8 + 4 * Q + 3, 0);
}

template <int NCH, int TXV, int TY, int TZ>
__global__ __launch_bounds__(TXV * TY * TZ, (TXV * TY * TZ) / 128) void conv3d_k3_c8_kernel(ConvC8Args ka) {
  constexpr int NTH = TXV * TY * TZ;               // one thread per output voxel of the tile
  constexpr int NW = NTH / 64;
  constexpr int HX = TXV + 2, HY = TY + 2, HZ = TZ + 2;
  constexpr int NHALO = HX * HY * HZ;
  // positions per channel-quad plane: 8 mod 16, so the two 16-byte pieces of a voxel (q = 0, 1), written by
  // neighbouring lanes, fall on disjoint banks
  constexpr int PLANE = ((NHALO + 7) / 16) * 16 + 8;
  constexpr int IN_FLOATS = 2 * PLANE * 4;
  constexpr int IN_IT = (NHALO * 2 + NTH - 1) / NTH;   // staging iterations per thread (16-byte pieces)
  constexpr int W_FLOATS = NCH == 1 ? 0 : NCH * 27 * 64;
  static_assert(PLANE >= NHALO && IN_IT <= 8, "tile shape");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;

  const vx_conv3d_args& a = ka.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ntiles = ka.tiles_x * ka.tiles_y * ka.tiles_z;
  const int total = ntiles * a.N;
  const int lastx = (ka.tiles_x - 1) * TXV, lasty = (ka.tiles_y - 1) * TY, lastz = (ka.tiles_z - 1) * TZ;

  // ---- the lane's voxel ----
  const int lx = tid % TXV, ly = (tid / TXV) % TY, lz = tid / (TXV * TY);
  const int vbase = ((lz * HY + ly) * HX + lx) * 4;                       // LDS float index at tap (0,0,0), q = 0
  const int ovox = (lz * a.H + ly) * a.W + lx;                            // relative to the tile origin
  const unsigned ovoff = (unsigned)((ovox * a.out_pitch + a.out_coff) * 4);
  const unsigned eoff = (unsigned)(ovox * 8);
  const bool obad_xhi = lx >= a.W - lastx, obad_yhi = ly >= a.H - lasty, obad_zhi = lz >= a.D - lastz;

  // ---- weights: 27 registers for the whole kernel (one chunk); with two chunks 54 registers would push the kernel
  // past 128 VGPRs (4 waves per SIMD), so those layers keep them in LDS and read one dword per tap (16 MFMAs) ----
  constexpr int WREG = NCH == 1 ? 27 : 1;
  float wreg[WREG];
  float* s_w = smem + IN_FLOATS;                       // [NCH][27][64], only for NCH > 1
  if (NCH == 1) {
#pragma unroll
    for (int t = 0; t < WREG; ++t) wreg[t] = a.w_packed[t * 64 + lane];
  } else {
    for (int i = tid; i < NCH * 27 * 64; i += NTH) s_w[i] = a.w_packed[i];
  }

  // ---- per-thread staging pattern (identical for every tile; see conv3d_mfma.hip) ----
  const int xb = a.in_xblk;
  const int Csrc = xb ? a.Cin / 2 : a.Cin;
  const int voxf = xb ? 2 * Csrc : a.in_pitch;
  const int rowf = a.W * voxf;
  const int biasf = (a.H + 1) * rowf + 4 * voxf;
  unsigned voff[IN_IT];
  int ldst[IN_IT];
  // invalid-piece masks, 8 bits per class: m0 = always | x-low << 8 | x-high << 16 | y-low << 24, m1 = y-high | z-low << 8 | z-high << 16
  unsigned m0 = 0, m1 = 0;
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int idx = tid + it * NTH;
    const int vox = idx >> 1, q = idx & 1;
    const int hx = vox % HX, hy = (vox / HX) % HY, hz = vox / (HX * HY);
    const int dxr = hx - 1, dyr = hy - 1, dzr = hz - 1;
    int xf;
    if (xb) {
      const int blk = dxr >= 0 ? dxr / xb : -((-dxr + xb - 1) / xb);
      const int rem = dxr - blk * xb;
      const int sl = (Csrc < 8) ? (4 * q) / Csrc : 0;    // 4-channel halves: the chunk spans both
      const int cs = (Csrc < 8) ? (4 * q) % Csrc : 4 * q;
      xf = (blk * 2 + sl) * xb * Csrc + rem * Csrc + cs;
    } else {
      xf = dxr * a.in_pitch + 4 * q;
    }
    voff[it] = (unsigned)(((dzr * a.H + dyr) * rowf + xf + biasf) * 4);
    ldst[it] = (q * PLANE + (hz * HY + hy) * HX + hx) * 4;
    if (idx >= NHALO * 2) m0 |= 1u << it;
    if (dxr < 0) m0 |= 0x100u << it;
    if (dxr >= a.W - lastx) m0 |= 0x10000u << it;
    if (dyr < 0) m0 |= 0x1000000u << it;
    if (dyr >= a.H - lasty) m1 |= 1u << it;
    if (dzr < 0) m1 |= 0x100u << it;
    if (dzr >= a.D - lastz) m1 |= 0x10000u << it;
  }
  const size_t in_sample = (size_t)a.D * a.H * rowf;
  const size_t out_sample = (size_t)a.D * a.H * a.W * a.out_pitch;
  const int cper = xb && Csrc >= 8 ? Csrc / 8 : 0;

  auto decode = [&](int tile_lin, int& n, int& tx, int& ty, int& tz) {
    unsigned t = (unsigned)tile_lin, q;
    q = ka.tiles_x == 1 ? t : __umulhi(t, ka.mx); tx = (int)(t - q * ka.tiles_x); t = q;
    q = ka.tiles_y == 1 ? t : __umulhi(t, ka.my); ty = (int)(t - q * ka.tiles_y); t = q;
    q = ka.tiles_z == 1 ? t : __umulhi(t, ka.mz); tz = (int)(t - q * ka.tiles_z); n = (int)q;
  };

  f32x4 ibuf[IN_IT];
  auto prefetch = [&](int tile_lin, int chunk, bool have) {
    int n, tx, ty, tz;
    decode(tile_lin, n, tx, ty, tz);
    // wave-uniform class selectors (scalar), then one AND per register and a fold of the 8-bit fields
    const unsigned s0 = 0xFFu | (tx == 0 ? 0xFF00u : 0u) | (tx == ka.tiles_x - 1 ? 0xFF0000u : 0u) | (ty == 0 ? 0xFF000000u : 0u);
    const unsigned s1 = (ty == ka.tiles_y - 1 ? 0xFFu : 0u) | (tz == 0 ? 0xFF00u : 0u) | (tz == ka.tiles_z - 1 ? 0xFF0000u : 0u);
    unsigned bad = (m0 & s0) | (m1 & s1);
    bad |= bad >> 16;
    bad |= bad >> 8;
    if (!have) bad = 0xFFu;
    int coff;
    if (!xb) coff = chunk * 8;
    else if (cper) coff = (chunk / cper) * xb * Csrc + (chunk % cper) * 8;
    else coff = 0;
    const unsigned soff = (unsigned)((((tz * TZ) * a.H + ty * TY) * rowf + tx * TXV * voxf + coff) * 4);
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.in + (size_t)(have ? n : 0) * in_sample - biasf), 0, VX_NUMREC, 0x00020000);
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const unsigned vo = ((bad >> it) & 1u) ? VX_OOB : voff[it];
      ibuf[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)vo, (int)soff, 0));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < IN_IT; ++it)
      if (tid + it * NTH < NHALO * 2) *reinterpret_cast<f32x4*>(s_in + ldst[it]) = ibuf[it];
  };

  // bias: wave-uniform, lives in scalar registers
  float bias[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) bias[c] = a.bias[c];

  // Workgroups b, b + 8, b + 16, ... share an XCD (one L2): give each XCD a CONTIGUOUS run of gridDim.x / 8 tiles
  // per round instead of every 8th tile, so the halo a tile shares with its neighbours is fetched into one L2 once
  // rather than into all eight (measured fabric-side FETCH_SIZE: 1.7-2.3 x the input with the round-robin order).
  int tile_lin = blockIdx.x, chunk = 0;
  if ((gridDim.x & 7) == 0) tile_lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  bool have = tile_lin < total;
  prefetch(tile_lin, 0, have);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

#ifdef VX_CONV_STAMPS
  unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_last, st_iters = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
  while (have) {
    if (C8_DBG < 3) __syncthreads();   // everyone finished reading the previous item from LDS
    VX_STAMP(0);
    if (C8_DBG < 2) commit();          // (waits for the prefetched loads)
    VX_STAMP(1);
    if (C8_DBG < 3) __syncthreads();
    VX_STAMP(2);
    int ntile = tile_lin, nchunk = chunk + 1;
    if (nchunk == NCH) { nchunk = 0; ntile = tile_lin + (int)gridDim.x; }
    const bool nhave = ntile < total;
    if (C8_DBG < 2) prefetch(ntile, nchunk, nhave);
    VX_STAMP(3);
    __builtin_amdgcn_s_setprio(0);

    {
      // ---- 27 taps x 2 reads x 8 MFMAs; reads run PD ahead of their MFMAs ----
      const float* sb = s_in + vbase;
      constexpr int PD = 3, NB = 4, NRD = 54;
      f32x4 xr[NB];
      auto rd = [&](int i) -> f32x4 {   // read i = tap * 2 + q
        const int t = i >> 1, q = i & 1;
        const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
        return *reinterpret_cast<const f32x4*>(sb + (q * PLANE + (kz * HY + ky) * HX + kx) * 4);
      };
#pragma unroll
      for (int i = 0; i < PD; ++i) xr[i] = rd(i);
      const float* sw = s_w + chunk * (27 * 64) + lane;
      float wl[3];
      if (NCH > 1) { wl[0] = sw[0]; wl[1] = sw[64]; }
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        if (NCH > 1 && t + 2 < 27) wl[(t + 2) % 3] = sw[(t + 2) * 64];
        const float w = NCH == 1 ? wreg[t % WREG] : wl[t % 3];
        if (2 * t + PD < NRD) xr[(2 * t + PD) % NB] = rd(2 * t + PD);
        c8_quad<0>(acc0, acc1, w, xr[(2 * t) % NB]);
        if (2 * t + 1 + PD < NRD) xr[(2 * t + 1 + PD) % NB] = rd(2 * t + 1 + PD);
        c8_quad<1>(acc0, acc1, w, xr[(2 * t + 1) % NB]);
      }
    }

    __builtin_amdgcn_s_setprio(3);
    VX_STAMP(4);
    if (C8_DBG >= 1) {
      asm volatile("" :: "v"(acc0), "v"(acc1));
    } else if (chunk == NCH - 1) {
      // ---- epilogue: this lane's voxel, 8 channels ----
      int n, tx, ty, tz;
      decode(tile_lin, n, tx, ty, tz);
      const bool bad = (tx == ka.tiles_x - 1 && obad_xhi) || (ty == ka.tiles_y - 1 && obad_yhi) ||
                       (tz == ka.tiles_z - 1 && obad_zhi);
      const unsigned vox0 = (unsigned)(((tz * TZ) * a.H + ty * TY) * a.W + tx * TXV);
      const unsigned osoff = vox0 * (unsigned)a.out_pitch * 4u;
      const unsigned e = vox0 * 8u + eoff;
      const __amdgpu_buffer_rsrc_t osrd =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * out_sample), 0, VX_NUMREC, 0x00020000);
      float v[8];
#pragma unroll
      for (int c = 0; c < 4; ++c) { v[c] = acc0[c] + bias[c]; v[4 + c] = acc1[c] + bias[4 + c]; }
      acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};

      if (a.stats_partial) {
        // per-wave transpose through LDS: S[k][lane] (k = channel, 8 + channel for the squares), then lane L sums
        // 16 lanes of value L >> 2, the 4 lanes of a quad combine, and 16 lanes hold the wave's 16 sums
        float* S = smem + IN_FLOATS + W_FLOATS + wave * (16 * 64);
        float* s_fin = smem + IN_FLOATS + W_FLOATS + NW * (16 * 64);   // [NW][16]
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float x = bad ? 0.f : v[c];
          S[c * 64 + lane] = x;
          S[(8 + c) * 64 + lane] = x * x;
        }
        const f32x4* rp = reinterpret_cast<const f32x4*>(S + (lane >> 2) * 64 + (lane & 3) * 16);
        f32x4 t4 = (rp[0] + rp[1]) + (rp[2] + rp[3]);   // (same wave: ds ops complete in order)
        float s = (t4[0] + t4[1]) + (t4[2] + t4[3]);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if ((lane & 3) == 0) s_fin[wave * 16 + (lane >> 2)] = s;
        __syncthreads();
        if (tid < 16) {
          float tot = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) tot += s_fin[w * 16 + tid];
          const int tile = tile_lin - n * ntiles;
          a.stats_partial[(((size_t)n * ntiles + tile) * 8 + (tid & 7)) * 2 + (tid >> 3)] = tot;
        }
        // S / s_fin are re-written only after the next item's two barriers
      }

      if (a.act == VX_ACT_LRELU) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.01f * v[c]);
      } else if (a.act == VX_ACT_RELU) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
      }
      if (a.drop_mode == VX_DROP_HASH) {
        // elements e .. e + 7 share one 32-element hash word (e % 8 == 0)
        const vx_dkey dkey = vx_drop_key(vx_seed_of(a, a.drop_seed), a.drop_layer, (uint32_t)n);
        const uint32_t bits = vx_drop_word(dkey, e >> 5) >> (e & 31u);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] *= __uint_as_float((bits << (30 - c)) & 0x40000000u);   // keep ? 2 : 0
      } else if (a.drop_mode == VX_DROP_MASK) {
        uint2 mk = make_uint2(0u, 0u);
        if (!bad) mk = *reinterpret_cast<const uint2*>(a.drop_mask + (size_t)n * a.D * a.H * a.W * 8 + e);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = ((mk.x >> (8 * c)) & 0xFFu) ? 2.f * v[c] : 0.f;
          v[4 + c] = ((mk.y >> (8 * c)) & 0xFFu) ? 2.f * v[4 + c] : 0.f;
        }
      }
      if (a.head_out) {
        // fused 1x1x1 head (conv1x1.hip: same fmaf chain, slot and un-flip): all 8 channels are in this lane
        int gx = tx * TXV + lx, gy = ty * TY + ly, gz = tz * TZ + lz;
        const int f = a.head_flip ? a.head_flip[n] : 0;
        if (f & 1) gz = a.D - 1 - gz;
        if (f & 2) gy = a.H - 1 - gy;
        if (f & 4) gx = a.W - 1 - gx;
        const size_t nvox = (size_t)a.D * a.H * a.W;
        const int slot = a.head_dst ? a.head_dst[n] : n;
        float* o = a.head_out + (size_t)slot * a.head_C * nvox + ((size_t)gz * a.H + gy) * a.W + gx;
        for (int c = 0; c < a.head_C; ++c) {
          float hacc = a.head_b[c];
#pragma unroll
          for (int k = 0; k < 8; ++k) hacc = fmaf(a.head_w[c * 8 + k], v[k], hacc);
          if (!bad) o[(size_t)c * nvox] = hacc;
        }
      }
      if (a.out) {
        const unsigned vo = bad ? VX_OOB : ovoff;
        const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), osrd, (int)vo, (int)osoff, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), osrd, (int)(vo + 16u), (int)osoff, 0);
        // gfx950 store-data hazard with an SGPR soffset (conv3d_mfma.hip): keep the data registers alive
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 3" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    VX_STAMP(5);
#ifdef VX_CONV_STAMPS
    ++st_iters;
#endif
    tile_lin = ntile; chunk = nchunk; have = nhave;
  }
#ifdef VX_CONV_STAMPS
  if (ka.stamps && lane == 0) {
    unsigned long long* d = ka.stamps + ((size_t)blockIdx.x * 8 + (wave & 7)) * 8;
    for (int i = 0; i < 6; ++i) d[i] = st_sum[i];
    d[6] = st_iters;
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// weight packing: torch (8, Cin, 3,3,3) -> [chunk][tap][lane]: lane (b = lane >> 2, i = lane & 3) holds
// W[cout = 4 (b >> 3) + i][cin = 8 chunk + (b & 7)][tap]
__global__ void pack_conv3d_k3_c8_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int total) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, tap = (i >> 6) % 27, chunk = (i >> 6) / 27;
    const int b = lane >> 2, co = 4 * (b >> 3) + (lane & 3), ci = 8 * chunk + (b & 7);
    out[i] = w[((size_t)co * Cin + ci) * 27 + tap];
  }
}

bool vx_conv3d_c8_applies(int Cin, int Cout) { return Cout == 8 && (Cin == 8 || Cin == 16); }

int vx_pack_conv3d_k3_c8(const float* w_torch, float* w_packed, int Cin, hipStream_t s) {
  const int total = 27 * Cin * 8;
  hipLaunchKernelGGL(pack_conv3d_k3_c8_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w_torch, w_packed, Cin, total);
  VX_CHECK_LAUNCH("vx_pack_conv3d_k3(c8)");
  return VX_OK;
}

template <int NCH, int TXV, int TY, int TZ>
static int launch_c8(const ConvC8Args& ka, hipStream_t s) {
  constexpr int NTH = TXV * TY * TZ, NW = NTH / 64;
  constexpr int NHALO = (TXV + 2) * (TY + 2) * (TZ + 2);
  constexpr int PLANE = ((NHALO + 7) / 16) * 16 + 8;
  const vx_conv3d_args& a = ka.a;
  const size_t lds = (size_t)(2 * PLANE * 4 + (NCH == 1 ? 0 : NCH * 27 * 64) + (a.stats_partial ? NW * 16 * 64 + NW * 16 : 0)) * sizeof(float);
  auto kern = conv3d_k3_c8_kernel<NCH, TXV, TY, TZ>;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) VX_FAIL((int)e, "vx_conv3d_k3(c8): hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr_lds = lds;
  }
  const int total_tiles = ka.tiles_x * ka.tiles_y * ka.tiles_z * a.N;
  int per_cu = (int)((160 * 1024) / lds);
  constexpr int max_waves = NCH == 1 ? 16 : 24;   // 100 / 73 VGPRs per lane: 4 / 6 waves per SIMD
  if (per_cu * NW > max_waves) per_cu = max_waves / NW;
  if (per_cu < 1) per_cu = 1;
  int gx = 256 * per_cu;
  if (gx > total_tiles) gx = total_tiles;
  static const char* kname = vx_kname("conv3d_k3_c8_kernel<%d,%d,%d,%d>", NCH, TXV, TY, TZ);
  vx_note_kernel(kname);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(NTH), lds, s, ka);
  VX_CHECK_LAUNCH("vx_conv3d_k3(c8)");
  return VX_OK;
}

// tile = the x-pair kernel's tile (conv3d_mfma.hip tile_config), so vx_conv3d_k3_tiles_for is the same for both
int vx_conv3d_k3_c8(const vx_conv3d_args& a, int txv, int ty, int tz, hipStream_t s) {
  ConvC8Args ka;
  ka.a = a;
  ka.dbg = 0;   // phase ablation: diagnostic build only (below)
  ka.stamps = nullptr;
#ifdef VX_CONV_STAMPS
  if (const char* e = getenv("VX_CONV_DBG_PTR")) ka.stamps = (unsigned long long*)strtoull(e, nullptr, 0);
  if (const char* e = getenv("VX_C8_DBG")) ka.dbg = atoi(e);
#endif
  ka.tiles_x = (a.W + txv - 1) / txv; ka.tiles_y = (a.H + ty - 1) / ty; ka.tiles_z = (a.D + tz - 1) / tz;
  ka.mx = (unsigned)((1ull << 32) / (unsigned)ka.tiles_x) + 1u;
  ka.my = (unsigned)((1ull << 32) / (unsigned)ka.tiles_y) + 1u;
  ka.mz = (unsigned)((1ull << 32) / (unsigned)ka.tiles_z) + 1u;
  const int nch = a.Cin / 8;
  if (txv == 32) return nch == 1 ? launch_c8<1, 32, 4, 4>(ka, s) : launch_c8<2, 32, 4, 4>(ka, s);
  if (txv == 16) return nch == 1 ? launch_c8<1, 16, 8, 4>(ka, s) : launch_c8<2, 16, 8, 4>(ka, s);
  return nch == 1 ? launch_c8<1, 8, 4, 4>(ka, s) : launch_c8<2, 8, 4, 4>(ka, s);
}
