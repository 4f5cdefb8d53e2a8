// K6: ConvTranspose3d(kernel 2, stride 2) -- non-overlapping 2x upsampling: every input voxel
// scatters Cin x (8 * Cout) products to its own 2x2x2 output block (unet3D_module.py:113-118, 157-190).
// 1.7 % of the network's MACs and 8 x more bytes written than read: HBM/store-bound, so VALU FMAs with
// wave-uniform weights (scalar loads).  One thread = one input voxel x one (dz,dy) x 8 output channels,
// producing the dx = 0,1 pair => 64 contiguous output bytes per lane when Cout == 8.
// The output goes straight into channels [out_coff, out_coff + Cout) of the decoder's concat buffer,
// which removes torch.cat (K7).  Optional ReLU + dropout epilogue for center.4.
#include "common.h"
#include <stdlib.h>

constexpr int CT_CO = 8;  // output channels per thread

// address of channel c of output voxel (row, ox): plain pitched tensor or one half of a concat buffer
__device__ __forceinline__ float* convT_out_ptr(const vx_convT_args& a, size_t orow, int ox, int OW, int c) {
  if (a.out_xblk) {
    const int xb = a.out_xblk;
    return a.out + orow * (2 * (size_t)OW * a.Cout) + ((ox / xb) * 2 + a.out_half) * xb * a.Cout + (ox % xb) * a.Cout + c;
  }
  return a.out + (orow * OW + ox) * a.out_pitch + a.out_coff + c;
}

__global__ __launch_bounds__(256) void convT_k2s2_kernel(vx_convT_args a, int64_t nvox_in) {
  // blockIdx.y = ((dz*2 + dy) * ngroups + cgroup)
  const int ngroups = a.Cout / CT_CO;
  const int cgp = blockIdx.y % ngroups;
  const int dzy = blockIdx.y / ngroups;
  const int dz = dzy >> 1, dy = dzy & 1;
  const int co0 = cgp * CT_CO;
  // packed weights: [dz][dy][ci][dx][co]
  const float* __restrict__ wp = a.w_packed + (size_t)dzy * a.Cin * 2 * a.Cout;
  const int OD = a.D * 2, OH = a.H * 2, OW = a.W * 2;

  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvox_in; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int x = r % a.W; r /= a.W;
    const int y = r % a.H; r /= a.H;
    const int z = r % a.D; r /= a.D;
    const int n = (int)r;
    const float* __restrict__ xin = a.in + (size_t)i * a.in_pitch;
    float acc[2][CT_CO];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int c = 0; c < CT_CO; ++c) acc[d][c] = a.bias[co0 + c];
    for (int ci = 0; ci < a.Cin; ci += 4) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xin + ci);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* __restrict__ wr = wp + (size_t)(ci + k) * 2 * a.Cout + co0;
#pragma unroll
        for (int c = 0; c < CT_CO; ++c) {
          acc[0][c] = fmaf(wr[c], xv[k], acc[0][c]);
          acc[1][c] = fmaf(wr[a.Cout + c], xv[k], acc[1][c]);
        }
      }
    }
    const int oz = 2 * z + dz, oy = 2 * y + dy;
    const vx_dkey dkey = vx_drop_key(vx_seed_of(a, a.drop_seed), a.drop_layer, (uint32_t)n);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int ox = 2 * x + d;
      const size_t ovox = ((size_t)(n * OD + oz) * OH + oy) * OW + ox;
#pragma unroll
      for (int c4 = 0; c4 < CT_CO; c4 += 4) {
        f32x4 v = (f32x4){acc[d][c4], acc[d][c4 + 1], acc[d][c4 + 2], acc[d][c4 + 3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = vx_act(v[j], a.act);
        if (a.drop_mode == VX_DROP_HASH) {
          const uint32_t e = (uint32_t)(((oz * OH + oy) * OW + ox) * a.Cout + co0 + c4);
          const uint32_t bits = vx_drop_bits4(dkey, e);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ((bits >> j) & 1u) ? 2.f * v[j] : 0.f;
        } else if (a.drop_mode == VX_DROP_MASK) {
          const uint32_t mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + ovox * a.Cout + co0 + c4);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * v[j] : 0.f;
        }
        *reinterpret_cast<f32x4*>(convT_out_ptr(a, (size_t)(n * OD + oz) * OH + oy, ox, OW, co0 + c4)) = v;
      }
    }
  }
}

// Row-streaming variant for the two large up-convolutions (Cin <= 32): one thread = one 16-byte piece of an
// OUTPUT row, consecutive lanes = consecutive pieces, so every store instruction of a wave is one contiguous
// 1 KiB segment; the thread's (dx, channel quad) is fixed by its lane, so its Cin x 4 weights live in registers
// for the whole kernel and the inner loop is FMAs only.  blockIdx.y = (dz, dy).
template <int CIN>
__global__ __launch_bounds__(256) void convT_k2s2_rows_kernel(vx_convT_args a, int64_t total) {
  const int dzy = blockIdx.y, dz = dzy >> 1, dy = dzy & 1;
  const int C4 = a.Cout / 4;
  const int OW = a.W * 2, OH = a.H * 2, OD = a.D * 2;
  const int PW = OW * C4;  // pieces per output row
  const int tid = threadIdx.x;
  const int cq = tid % C4, dx = (tid / C4) & 1;  // fixed per thread: 256 and PW are multiples of 2*C4
  f32x4 wreg[CIN];
  {
    const float* wp = a.w_packed + (size_t)dzy * CIN * 2 * a.Cout + dx * a.Cout + cq * 4;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) wreg[ci] = *reinterpret_cast<const f32x4*>(wp + (size_t)ci * 2 * a.Cout);
  }
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + cq * 4);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + tid; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int p = (int)(r % PW); r /= PW;
    const int y = (int)(r % a.H); r /= a.H;
    const int z = (int)(r % a.D); r /= a.D;
    const int n = (int)r;
    const int ox = p / C4, x = ox >> 1;
    const float* __restrict__ xin = a.in + (((size_t)(n * a.D + z) * a.H + y) * a.W + x) * a.in_pitch;
    f32x4 xv[CIN / 4];
#pragma unroll
    for (int k = 0; k < CIN / 4; ++k) xv[k] = *reinterpret_cast<const f32x4*>(xin + 4 * k);
    f32x4 acc = b4;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float xs = xv[ci >> 2][ci & 3];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(wreg[ci][j], xs, acc[j]);
    }
    const int oz = 2 * z + dz, oy = 2 * y + dy;
    const size_t ovox = ((size_t)(n * OD + oz) * OH + oy) * OW + ox;
    if (a.act == VX_ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.f);
    } else if (a.act == VX_ACT_LRELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.01f * acc[j]);
    }
    if (a.drop_mode == VX_DROP_HASH) {
      const uint32_t e = (uint32_t)(((oz * OH + oy) * OW + ox) * a.Cout + cq * 4);
      const uint32_t bits = vx_drop_bits4(vx_drop_key(vx_seed_of(a, a.drop_seed), a.drop_layer, (uint32_t)n), e);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);
    } else if (a.drop_mode == VX_DROP_MASK) {
      const uint32_t mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + ovox * a.Cout + cq * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * acc[j] : 0.f;
    }
    *reinterpret_cast<f32x4*>(convT_out_ptr(a, (size_t)(n * OD + oz) * OH + oy, ox, OW, cq * 4)) = acc;
  }
}

// Matrix-core variant (Cin in {16, 32, 64, 128}): the up-convolution is the GEMM
//     out[(dz,dy,dx,co)][voxel] = sum_ci W[(dz,dy,dx,co)][ci] * in[ci][voxel]
// with 8 * Cout rows.  v_mfma_f32_16x16x4_f32 tiles: 16 rows x 16 consecutive input voxels; k index g of step
// (q, j) is channel 16 q + 4 g + j, so a lane's B operand is one 16-byte load per 16 channels straight from global
// memory (a wave reads 16 voxels x 64 B = 1 KiB contiguous) and no LDS is involved.  A lane ends with 4 consecutive
// output channels of one output voxel -> one 16-byte store; with the rows ordered (dz, dy, dx, co) the 64 lanes of
// a wave store whole 32/64/128-byte voxels of neighbouring output x.  A workgroup keeps the weights of RT row tiles
// (blockIdx.y picks which) in registers for its whole life and walks column tiles with a grid stride.  The VALU
// work left is the bias/activation/dropout epilogue, which leaves the kernel HBM-store-bound.
struct ConvTDecode { unsigned mW, mH, mD; };   // 2^32 / d + 1 (0: d == 1); exact while v * d < 2^32 (launcher checks)
__device__ __forceinline__ unsigned ct_div(unsigned n, unsigned m) { return m ? __umulhi(n, m) : n; }

// MASK: the injected-mask dropout of the parity tests as its own instance -- a (run-time) branch around its mask load made
// hipcc put an `s_waitcnt vmcnt(0)` at the join, on EVERY path: each column tile then waited for the stores of the tile
// before and for its own freshly issued prefetch
template <int CIN, int RT, bool MASK = false>
__global__ __launch_bounds__(256) void convT_k2s2_mfma_kernel(vx_convT_args a, int ncoltiles, int nvox_in, ConvTDecode dc) {
  constexpr int Q = CIN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int OW = a.W * 2, OH = a.H * 2, OD = a.D * 2;
  const int rt0 = blockIdx.y * RT;
  // A fragments: row 16 (rt0 + rt) + m, channels 16 q + 4 g + j; gathered once from the [dz][dy][ci][dx][co] packing
  f32x4 wreg[RT][Q];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = 16 * (rt0 + rt) + m;
    const int pos = row / a.Cout, co = row % a.Cout;
    const float* wp = a.w_packed + (size_t)(pos >> 1) * CIN * 2 * a.Cout + (pos & 1) * a.Cout + co;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) wreg[rt][q][j] = wp[(size_t)(16 * q + 4 * g + j) * 2 * a.Cout];
  }
  // D fragments: rows 16 (rt0 + rt) + 4 g + r: one output position (dz, dy, dx) and 4 consecutive channels
  int opos[RT], oco[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = 16 * (rt0 + rt) + 4 * g;
    opos[rt] = row / a.Cout;
    oco[rt] = row % a.Cout;
  }
  const int xs = a.out_xblk ? __builtin_ctz((unsigned)a.out_xblk) : 0;   // x-block size is 1, 2 or 4
  const int wstride = (int)gridDim.x * 4;
  float rmax = 0.f;   // range guard of the split-fp16 consumer (vx_convT_args.range_flag)
  // the next column tile's input is in flight while this one is multiplied and stored (round 3: at 2 waves per SIMD --
  // 184 registers with 8 row tiles -- the load latency of every iteration was exposed: 0.34 of the HBM roof)
  auto load_x = [&](int ct, f32x4* dst) {
    int v = ct * 16 + m;
    if (v >= nvox_in) v = nvox_in - 1;
    const float* __restrict__ xin = a.in + (size_t)v * a.in_pitch + 4 * g;
#pragma unroll
    for (int q = 0; q < Q; ++q) dst[q] = *reinterpret_cast<const f32x4*>(xin + 16 * q);
  };
  // (only where the registers allow it: with 64 / 128 input channels the extra fragments spill -- measured +40 %)
  constexpr bool PF = CIN <= 32;
  // the lane's bias vectors, once: loaded inside the row-tile loop each of them drew an `s_waitcnt vmcnt(0)` -- which also waits
  // for the STORES of the row tile before, so every row tile paid a store's latency (0.29 ms for a 0.84-GB launch)
  constexpr bool HB = PF || RT <= 4;      // (64 input channels x 8 row tiles has no registers left for them)
  f32x4 biasr[HB ? RT : 1];
  if constexpr (HB) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) biasr[rt] = *reinterpret_cast<const f32x4*>(a.bias + oco[rt]);
  }
  f32x4 xnext[Q];
  if constexpr (PF) {
    const int ct0 = blockIdx.x * 4 + wave;
    load_x(ct0 < ncoltiles ? ct0 : ncoltiles - 1, xnext);
  }
  const uint32_t seed0 = vx_seed_of(a, a.drop_seed);    // (the optional device seed word: read once, not per column tile)
  for (int ct = blockIdx.x * 4 + wave; ct < ncoltiles; ct += wstride) {
    const int v = ct * 16 + m;                 // flattened input voxel of this lane's column
    const bool ok = v < nvox_in;
    const int vc = ok ? v : nvox_in - 1;
    f32x4 xv[Q];
    if constexpr (PF) {
#pragma unroll
      for (int q = 0; q < Q; ++q) xv[q] = xnext[q];
      const int ctn = ct + wstride;
      load_x(ctn < ncoltiles ? ctn : ncoltiles - 1, xnext);
    } else {
      load_x(ct, xv);
    }
    unsigned r = (unsigned)vc, q;
    q = ct_div(r, dc.mW); const int x = (int)(r - q * (unsigned)a.W); r = q;
    q = ct_div(r, dc.mH); const int y = (int)(r - q * (unsigned)a.H); r = q;
    q = ct_div(r, dc.mD); const int z = (int)(r - q * (unsigned)a.D);
    const int n = (int)q;
    const vx_dkey dkey = vx_drop_key(seed0, a.drop_layer, (uint32_t)n);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 acc;
      if constexpr (HB) acc = biasr[rt];
      else acc = *reinterpret_cast<const f32x4*>(a.bias + oco[rt]);
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[rt][q][j], xv[q][j], acc, 0, 0, 0);
      const int pos = opos[rt];
      const int oz = 2 * z + (pos >> 2), oy = 2 * y + ((pos >> 1) & 1), ox = 2 * x + (pos & 1);
      const size_t orow = ((size_t)n * OD + oz) * OH + oy;
      if (a.act == VX_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.f);
      } else if (a.act == VX_ACT_LRELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.01f * acc[j]);
      }
      if (a.drop_mode == VX_DROP_HASH) {
        const uint32_t bits = vx_drop_bits4(dkey, (uint32_t)(((oz * OH + oy) * OW + ox) * a.Cout + oco[rt]));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);
      }
      if constexpr (MASK) {
        const uint32_t mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + (orow * OW + ox) * a.Cout + oco[rt]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * acc[j] : 0.f;
      }
      float* op;
      if (a.out_xblk)
        op = a.out + orow * (2 * (size_t)OW * a.Cout) +
             ((((ox >> xs) * 2 + a.out_half) << xs) + (ox & (a.out_xblk - 1))) * a.Cout + oco[rt];
      else
        op = a.out + (orow * OW + ox) * a.out_pitch + a.out_coff + oco[rt];
      // no `if (ok)`: the lanes beyond the last voxel were clamped onto it (vc) and store ITS values a second time -- a branch
      // around the store made every wait for the prefetched column tile a vmcnt(0) over this tile's eight stores
      *reinterpret_cast<f32x4*>(op) = acc;
      rmax = fmaxf(fmaxf(rmax, fmaxf(fabsf(acc[0]), fabsf(acc[1]))), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
    }
  }
  if (a.range_flag) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, off, 64));
    if (lane == 0 && !(rmax < 32768.f)) atomicMax(a.range_flag, __float_as_uint(rmax));   // see conv3d_xp8w.hip
  }
}

// Split-fp16 variant for the DEEP up-convolutions (Cin = 64 / 128: center.4, upscale4; round 5).  On the native-fp32 matrix
// instruction those two launches were bound by the instruction itself (8 * Cout rows x Cin / 4 instructions of 32 cycles per 16
// voxels: 42-46 TF, 0.12 + 0.06 ms for 0.21 + 0.03 GB) and had no registers left for a prefetch.  Here the same GEMM runs as three
// v_mfma_f32_16x16x32_f16 per 32 channels (operand splitting as conv3d_s16.hip: products exact, fp32 accumulation; the weights are
// gathered from the fp32 packing and split once per workgroup), 5 x fewer matrix cycles, and the next column tile's input is in
// flight while this one is multiplied and stored.  k index of a lane: channels 32 q + 8 g .. + 7 (two 16-byte loads per K step).
#include "s16_common.h"
template <int CIN, int RT, bool MASK = false>
__global__ __launch_bounds__(256) void convT_k2s2_s16_kernel(vx_convT_args a, int ncoltiles, int nvox_in, ConvTDecode dc) {
  constexpr int Q = CIN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int OW = a.W * 2, OH = a.H * 2, OD = a.D * 2;
  const int rt0 = blockIdx.y * RT;
  f16x8 wh[RT][Q], wl[RT][Q];
  float rmax = 0.f;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = 16 * (rt0 + rt) + m;
    const int pos = row / a.Cout, co = row % a.Cout;
    const float* wp = a.w_packed + (size_t)(pos >> 1) * CIN * 2 * a.Cout + (pos & 1) * a.Cout + co;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float w = wp[(size_t)(32 * q + 8 * g + j) * 2 * a.Cout];
        // a weight past the fp16 range is clamped (finite arithmetic below) and reported through the range word like an
        // activation past it: the caller re-runs on the native-fp32 kernels (vx_config.conv_fp32 = 1)
        const float c = fminf(fmaxf(w, -65504.f), 65504.f);
        if (!(fabsf(w) <= 65504.f)) rmax = __builtin_inff();
        const _Float16 h = (_Float16)c;
        wh[rt][q][j] = h;
        wl[rt][q][j] = (_Float16)((c - (float)h) * 2048.f);
      }
  }
  int opos[RT], oco[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = 16 * (rt0 + rt) + 4 * g;
    opos[rt] = row / a.Cout;
    oco[rt] = row % a.Cout;
  }
  const int xs = a.out_xblk ? __builtin_ctz((unsigned)a.out_xblk) : 0;
  const int wstride = (int)gridDim.x * 4;
  auto load_x = [&](int ct, f32x4* dst) {
    int v = ct * 16 + m;
    if (v >= nvox_in) v = nvox_in - 1;
    const float* __restrict__ xin = a.in + (size_t)v * a.in_pitch + 8 * g;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      dst[2 * q] = *reinterpret_cast<const f32x4*>(xin + 32 * q);
      dst[2 * q + 1] = *reinterpret_cast<const f32x4*>(xin + 32 * q + 4);
    }
  };
  f32x4 xnext[2 * Q];
  {
    const int ct0 = blockIdx.x * 4 + wave;
    load_x(ct0 < ncoltiles ? ct0 : ncoltiles - 1, xnext);
  }
  const uint32_t seed0 = vx_seed_of(a, a.drop_seed);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int ct = blockIdx.x * 4 + wave; ct < ncoltiles; ct += wstride) {
    const int v = ct * 16 + m;
    const int vc = v < nvox_in ? v : nvox_in - 1;
    f16x8 bh[Q], bl[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      f16x4 h0, l0, h1, l1;
      vx_split4(xnext[2 * q], h0, l0);
      vx_split4(xnext[2 * q + 1], h1, l1);
      const u32x2 a0 = __builtin_bit_cast(u32x2, h0), a1 = __builtin_bit_cast(u32x2, h1);
      const u32x2 c0 = __builtin_bit_cast(u32x2, l0), c1 = __builtin_bit_cast(u32x2, l1);
      bh[q] = __builtin_bit_cast(f16x8, (u32x4){a0[0], a0[1], a1[0], a1[1]});
      bl[q] = __builtin_bit_cast(f16x8, (u32x4){c0[0], c0[1], c1[0], c1[1]});
    }
    {
      const int ctn = ct + wstride;
      load_x(ctn < ncoltiles ? ctn : ncoltiles - 1, xnext);
    }
    // vx_split4 writes the lo halves from inline assembly: no wait states before a matrix instruction that reads them
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    unsigned r = (unsigned)vc, qq;
    qq = ct_div(r, dc.mW); const int x = (int)(r - qq * (unsigned)a.W); r = qq;
    qq = ct_div(r, dc.mH); const int y = (int)(r - qq * (unsigned)a.H); r = qq;
    qq = ct_div(r, dc.mD); const int z = (int)(r - qq * (unsigned)a.D);
    const int n = (int)qq;
    const vx_dkey dkey = vx_drop_key(seed0, a.drop_layer, (uint32_t)n);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 acc = *reinterpret_cast<const f32x4*>(a.bias + oco[rt]);
      f32x4 accx = zero;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[rt][q], bh[q], acc, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[rt][q], bl[q], accx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[rt][q], bh[q], accx, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(accx[j], 1.0f / 2048.f, acc[j]);
      const int pos = opos[rt];
      const int oz = 2 * z + (pos >> 2), oy = 2 * y + ((pos >> 1) & 1), ox = 2 * x + (pos & 1);
      const size_t orow = ((size_t)n * OD + oz) * OH + oy;
      if (a.act == VX_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.f);
      } else if (a.act == VX_ACT_LRELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.01f * acc[j]);
      }
      if (a.drop_mode == VX_DROP_HASH) {
        const uint32_t bits = vx_drop_bits4(dkey, (uint32_t)(((oz * OH + oy) * OW + ox) * a.Cout + oco[rt]));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] *= __uint_as_float((bits << (30 - j)) & 0x40000000u);
      }
      if constexpr (MASK) {
        const uint32_t mk = *reinterpret_cast<const uint32_t*>(a.drop_mask + (orow * OW + ox) * a.Cout + oco[rt]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = ((mk >> (8 * j)) & 0xFFu) ? 2.f * acc[j] : 0.f;
      }
      float* op;
      if (a.out_xblk)
        op = a.out + orow * (2 * (size_t)OW * a.Cout) +
             ((((ox >> xs) * 2 + a.out_half) << xs) + (ox & (a.out_xblk - 1))) * a.Cout + oco[rt];
      else
        op = a.out + (orow * OW + ox) * a.out_pitch + a.out_coff + oco[rt];
      *reinterpret_cast<f32x4*>(op) = acc;      // (lanes beyond the last voxel store its values a second time: no branch around the store)
      rmax = fmaxf(fmaxf(rmax, fmaxf(fabsf(acc[0]), fabsf(acc[1]))), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
    }
  }
  if (a.range_flag) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, off, 64));
    if (lane == 0 && !(rmax < 32768.f)) atomicMax(a.range_flag, __float_as_uint(rmax));
  }
}

template <int CIN, int RT>
static int launch_convT_s16(const vx_convT_args& a, hipStream_t s) {
  const int64_t nvox = (int64_t)a.N * a.D * a.H * a.W;
  const int ncoltiles = (int)((nvox + 15) / 16);
  const int groups = (a.Cout / 2) / RT;
  int bx = (ncoltiles + 3) / 4;
  // 512 workgroups (two per CU at these instances' registers), each wave walking its share of the column tiles: the weight gather
  // of the prologue is paid once per wave.  Measured at 320 samples (tools/ab_convT.py): upscale4 0.108 -> 0.056 ms, center.4
  // 0.052 -> 0.033; with one column tile per wave (the fp32 kernel's grid) the same kernel is 20-30 % SLOWER than the fp32 one,
  // and <64,8> / <128,4> (256 registers, one workgroup per CU) reach 0.076 / 0.034.
  const int cap = (512 + groups - 1) / groups;
  if (bx > cap) bx = cap;
  auto magic = [](int d) { return d == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d) + 1u; };
  ConvTDecode dc;
  dc.mW = magic(a.W); dc.mH = magic(a.H); dc.mD = magic(a.D);
  static const char* kname = vx_kname("convT_k2s2_s16_kernel<%d,%d,false>", CIN, RT);
  static const char* kname_m = vx_kname("convT_k2s2_s16_kernel<%d,%d,true>", CIN, RT);
  vx_note_kernel(a.drop_mode == VX_DROP_MASK ? kname_m : kname);
  if (a.drop_mode == VX_DROP_MASK)
    hipLaunchKernelGGL((convT_k2s2_s16_kernel<CIN, RT, true>), dim3((unsigned)bx, (unsigned)groups), dim3(256), 0, s, a, ncoltiles,
                       (int)nvox, dc);
  else
    hipLaunchKernelGGL((convT_k2s2_s16_kernel<CIN, RT, false>), dim3((unsigned)bx, (unsigned)groups), dim3(256), 0, s, a, ncoltiles,
                       (int)nvox, dc);
  VX_CHECK_LAUNCH("vx_convT_k2s2(s16)");
  return VX_OK;
}

template <int CIN, int RT>
static int launch_convT_mfma(const vx_convT_args& a, hipStream_t s) {
  const int64_t nvox = (int64_t)a.N * a.D * a.H * a.W;
  const int ncoltiles = (int)((nvox + 15) / 16);
  const int groups = (a.Cout / 2) / RT;          // 8 * Cout rows = Cout / 2 row tiles
  int bx = (ncoltiles + 3) / 4;
  int per = 32;   // workgroups per CU in the grid: more, shorter address streams write faster (fill: 5.1 TB/s at 2048 WGs, 6.5 at 32768)
  const int cap = (256 * per + groups - 1) / groups;
  if (bx > cap) bx = cap;
  auto magic = [](int d) { return d == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d) + 1u; };
  ConvTDecode dc;
  dc.mW = magic(a.W); dc.mH = magic(a.H); dc.mD = magic(a.D);
  static const char* kname = vx_kname("convT_k2s2_mfma_kernel<%d,%d,false>", CIN, RT);
  static const char* kname_m = vx_kname("convT_k2s2_mfma_kernel<%d,%d,true>", CIN, RT);
  vx_note_kernel(a.drop_mode == VX_DROP_MASK ? kname_m : kname);
  if (a.drop_mode == VX_DROP_MASK)
    hipLaunchKernelGGL((convT_k2s2_mfma_kernel<CIN, RT, true>), dim3((unsigned)bx, (unsigned)groups), dim3(256), 0, s, a,
                       ncoltiles, (int)nvox, dc);
  else
    hipLaunchKernelGGL((convT_k2s2_mfma_kernel<CIN, RT, false>), dim3((unsigned)bx, (unsigned)groups), dim3(256), 0, s, a,
                       ncoltiles, (int)nvox, dc);
  VX_CHECK_LAUNCH("vx_convT_k2s2(mfma)");
  return VX_OK;
}

__global__ void pack_convT_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int64_t total) {
  // torch (Cin, Cout, 2,2,2) -> [dz][dy][ci][dx][co]
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int co = r % Cout; r /= Cout;
    const int dx = r % 2; r /= 2;
    const int ci = r % Cin; r /= Cin;
    const int dy = r % 2; r /= 2;
    const int dz = (int)r;
    out[i] = w[(((size_t)ci * Cout + co) * 2 + dz) * 4 + dy * 2 + dx];
  }
}

extern "C" int64_t vx_convT_k2s2_packed_floats(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % 4 || Cout % 8) return -1;
  return (int64_t)Cin * Cout * 8;
}

extern "C" int vx_pack_convT_k2s2(const float* w_torch, float* w_packed, int Cin, int Cout, vx_stream_t stream) {
  if (!w_torch || !w_packed) VX_FAIL(VX_E_NULL, "vx_pack_convT_k2s2: null pointer");
  const int64_t total = vx_convT_k2s2_packed_floats(Cin, Cout);
  if (total < 0) VX_FAIL(VX_E_SHAPE, "vx_pack_convT_k2s2: Cin=%d (mult of 4) Cout=%d (mult of 8)", Cin, Cout);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_convT_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_torch, w_packed, Cin, Cout,
                     total);
  VX_CHECK_LAUNCH("vx_pack_convT_k2s2");
  return VX_OK;
}

extern "C" int vx_convT_k2s2(const vx_convT_args* ap, vx_stream_t stream) {
  if (!ap) VX_FAIL(VX_E_NULL, "vx_convT_k2s2: null args");
  const vx_convT_args& a = *ap;
  if (!a.in || !a.w_packed || !a.bias || !a.out) VX_FAIL(VX_E_NULL, "vx_convT_k2s2: null tensor");
  if (a.Cin <= 0 || a.Cout <= 0 || a.Cin % 4 || a.Cout % 8)
    VX_FAIL(VX_E_SHAPE, "vx_convT_k2s2: Cin=%d (mult of 4) Cout=%d (mult of 8)", a.Cin, a.Cout);
  if (a.N <= 0 || a.D <= 0 || a.H <= 0 || a.W <= 0) VX_FAIL(VX_E_SHAPE, "vx_convT_k2s2: empty tensor");
  if (a.in_pitch % 4 || a.in_pitch < a.Cin) VX_FAIL(VX_E_ALIGN, "vx_convT_k2s2: input pitch");
  if (a.out_xblk) {
    if ((a.out_xblk != 1 && a.out_xblk != 2 && a.out_xblk != 4) || (2 * a.W) % a.out_xblk || (a.out_half != 0 && a.out_half != 1))
      VX_FAIL(VX_E_SHAPE, "vx_convT_k2s2: bad concat layout (xblk=%d, half=%d)", a.out_xblk, a.out_half);
  } else if (a.out_pitch % 4 || a.out_coff % 4 || a.out_pitch < a.out_coff + a.Cout) {
    VX_FAIL(VX_E_ALIGN, "vx_convT_k2s2: pitches/offsets must be multiples of 4 floats");
  }
  if (a.drop_mode == VX_DROP_MASK && !a.drop_mask) VX_FAIL(VX_E_NULL, "vx_convT_k2s2: mask mode without mask");
  if ((int64_t)a.D * a.H * a.W * 8 * a.Cout >= (1ll << 32)) VX_FAIL(VX_E_SHAPE, "vx_convT_k2s2: sample too large");
  // matrix-core kernel for the channel counts the networks use (row tiles per workgroup: weights stay in <= 128 VGPRs)
  // (flat voxel index * largest dimension < 2^32: the multiply-high divisions of the index decode are then exact)
  const int64_t dmax = a.W > a.H ? (a.W > a.D ? a.W : a.D) : (a.H > a.D ? a.H : a.D);
  if ((int64_t)a.N * a.D * a.H * a.W < (1ll << 27) && (int64_t)a.N * a.D * a.H * a.W * dmax < (1ll << 32)) {
    hipStream_t s = (hipStream_t)stream;
    const int tiles = a.Cout / 2;
    if (a.Cin == 16 && tiles % 4 == 0) return tiles % 8 ? launch_convT_mfma<16, 4>(a, s) : launch_convT_mfma<16, 8>(a, s);
    if (a.Cin == 32 && tiles % 4 == 0) return tiles % 8 ? launch_convT_mfma<32, 4>(a, s) : launch_convT_mfma<32, 8>(a, s);
    // the deep up-convolutions on the split-fp16 products (vx_config.conv_fp32 == 0: the family the convolutions run in)
    if (vx_cfg().conv_fp32 == 0) {
      if (a.Cin == 64 && tiles % 4 == 0) return launch_convT_s16<64, 4>(a, s);
      if (a.Cin == 128 && tiles % 2 == 0) return launch_convT_s16<128, 2>(a, s);
    }
    if (a.Cin == 64 && tiles % 4 == 0) return tiles % 8 ? launch_convT_mfma<64, 4>(a, s) : launch_convT_mfma<64, 8>(a, s);
    if (a.Cin == 128 && tiles % 4 == 0) return launch_convT_mfma<128, 4>(a, s);
  }
  // large, shallow up-convolutions: row-streaming kernel (needs 256 % (2*Cout/4) == 0 and Cout <= 128)
  if ((a.Cin == 16 || a.Cin == 32) && a.Cout <= 128 && 256 % (2 * (a.Cout / 4)) == 0) {
    const int64_t total = (int64_t)a.N * a.D * a.H * (2 * a.W) * (a.Cout / 4);
    int bx = (int)((total + 255) / 256);
    if (bx > 4096) bx = 4096;
    dim3 grid((unsigned)bx, 4);
    vx_note_kernel("convT_k2s2_rows_kernel");
    if (a.Cin == 16)
      hipLaunchKernelGGL(convT_k2s2_rows_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, a, total);
    else
      hipLaunchKernelGGL(convT_k2s2_rows_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, a, total);
    VX_CHECK_LAUNCH("vx_convT_k2s2");
    return VX_OK;
  }
  const int64_t nvox = (int64_t)a.N * a.D * a.H * a.W;
  int bx = (int)((nvox + 255) / 256);
  if (bx > 8192) bx = 8192;
  dim3 grid((unsigned)bx, (unsigned)(4 * (a.Cout / CT_CO)));
  vx_note_kernel("convT_k2s2_kernel");
  hipLaunchKernelGGL(convT_k2s2_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, nvox);
  VX_CHECK_LAUNCH("vx_convT_k2s2");
  return VX_OK;
}
