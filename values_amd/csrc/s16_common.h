// Shared pieces of the split-fp16 convolution kernels (conv3d_s16.hip, conv3d_xp8.hip): vector types, the
// fp32 -> (hi, lo) fp16 split, and the cross-lane helpers of their epilogues.
#pragma once
#include "common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define VX_NUMREC 0xFFFFF000u
#define VX_OOB 0xFFFFF800u

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// x -> (hi, lo) with x = hi + lo * 2^-11 (see the header).  Two elements at a time: one v_cvt_pk_f16_f32, one
// v_pk_mul_f32 and a mixed-precision fma per element -- 8 VALU instructions per 16-byte piece.  No clamping: |x| >= 65520 turns into inf and the output into NaN -- loud, and
// out of reach for activations that went through InstanceNorm / a dropout-scaled LeakyReLU.
__device__ __forceinline__ void vx_split4(const f32x4 v, f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; j += 2) {
    const f32x2 x = {v[j], v[j + 1]};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    // lo = fp16(2048 x - 2048 hi) as one mixed-precision fma per element, reading hi as the fp16 it is and writing
    // the fp16 half directly: 8 instead of 14 VALU instructions per 16-byte piece, the same bits (2048 x, 2048 hi
    // and their difference are all exact in fp32; tools/micro/split_mix.hip compares the two forms)
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

// Single-instruction fp32 helpers.  hipcc packs neighbouring fp32 multiplies / adds into v_pk_mul_f32 / v_pk_add_f32
// (explicit f32x2 arithmetic and the SLP vectoriser alike); beside another wave's MFMA stream a packed fp32 instruction
// costs 11-13 cycles against ~4 for a plain one (MI355X_MICROARCH.md, cycle constants, "price of one filler beside
// MFMAs") -- measured here: the staging waves of conv3d_xp8w.hip spent ~12 cycles per vector instruction.  Inline
// assembly keeps these single instructions whatever the flags.
__device__ __forceinline__ float vx_sub1(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vx_mul1(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vx_add1(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vx_fma1(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vx_max1(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// r = max(r, |a|, |b|) as ONE instruction (fmaxf(fabsf()) chains compile to a canonicalising v_max per operand first)
__device__ __forceinline__ float vx_max3abs(float r, float a, float b) {
  asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(r) : "v"(a), "v"(b));
  return r;
}

// vx_split4 with the 2048 x products as plain multiplies (staging waves that run beside MFMA waves)
__device__ __forceinline__ void vx_split4_s(const f32x4 v, f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; j += 2) {
    const f32x2 x = {v[j], v[j + 1]};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const float xs0 = vx_mul1(v[j], 2048.f), xs1 = vx_mul1(v[j + 1], 2048.f);
    const float m2048 = -2048.f;
    const uint32_t hv = __builtin_bit_cast(uint32_t, h);
    uint32_t lv = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lv) : "v"(hv), "v"(m2048), "v"(xs0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lv) : "v"(hv), "v"(m2048), "v"(xs1));
    const f16x2 l = __builtin_bit_cast(f16x2, lv);
    hi[j] = h[0]; hi[j + 1] = h[1];
    lo[j] = l[0]; lo[j + 1] = l[1];
  }
}

// x + (x of lane ^ 16): two copies, v_permlane16_swap_b32 exchanges row 1 of the first with row 0 of the second (and
// row 3 with row 2), so a = [r0, r0, r2, r2], b = [r1, r1, r3, r3] -- no LDS-queue ds_bpermute as __shfl_xor(x, 16) takes
// (tools/micro/permlane_swap.hip); the same bits as x + __shfl_xor(x, 16)
__device__ __forceinline__ float vx_add_xor16(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}

// lane i of every 16-lane row <- lane (i + rot) % 16 of the same row
__device__ __forceinline__ float vx_row_ror(float x, int rot) {
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

