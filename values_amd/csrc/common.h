// Shared device/host helpers for libvalues_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/values_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

void vx_set_error(const char* fmt, ...);
const vx_config& vx_cfg();
// name of the kernel instance a launcher dispatched to, as rocprofv3 prints it (vx_unet3d_forward_profiled reports it
// next to each launch's time); vx_kname formats once and keeps the string for the life of the process
void vx_note_kernel(const char* name);
const char* vx_last_kernel();
const char* vx_kname(const char* fmt, ...);   // lib.cpp: the environment is read once, never per launch

#define VX_FAIL(code, ...)      \
  do {                          \
    vx_set_error(__VA_ARGS__);  \
    return (code);              \
  } while (0)

#define VX_CHECK_LAUNCH(name)                                      \
  do {                                                             \
    hipError_t e_ = hipGetLastError();                             \
    if (e_ != hipSuccess) {                                        \
      vx_set_error("%s: %s", (name), hipGetErrorString(e_));       \
      return (int)e_;                                              \
    }                                                              \
  } while (0)

static inline bool vx_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---------------------------------------------------------------------------------------------
// Dropout bit generator (VX_DROP_HASH).  p = 0.5 needs one bit per element: a 32-bit avalanche
// hash of (key, element_index >> 5) yields the keep-bits of 32 consecutive elements, so a lane
// holding 4 consecutive channels of a voxel gets its 4 bits from ONE hash (about a dozen VALU
// ops per 16-byte vector).  key = mix(seed, layer id, sample).
__device__ __forceinline__ uint32_t vx_mix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x7feb352du;
  h ^= h >> 15;
  h *= 0x846ca68bu;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t vx_drop_key(uint32_t seed, uint32_t layer, uint32_t sample) {
  return vx_mix32(seed * 0x9E3779B1u + layer * 0x85EBCA6Bu + sample * 0xC2B2AE35u + 0x27D4EB2Fu);
}
// seed of a launch: the by-value seed plus an optional DEVICE word -- a captured hipGraph replays with the kernel
// arguments it was captured with, so a caller that wants fresh dropout bits per replay updates that word instead
template <typename A>
__device__ __forceinline__ uint32_t vx_seed_of(const A& a, uint32_t by_value) {
  return by_value + (a.seed_dev ? *a.seed_dev : 0u);
}
// keep-bits for elements [e, e+4) of sample-local linear index e (e % 4 == 0)
__device__ __forceinline__ uint32_t vx_drop_bits4(uint32_t key, uint32_t e) {
  // (the word index enters by XOR: the key is already a full avalanche of (seed, layer, sample), and vx_mix32's two
  // multiply-xorshift rounds spread consecutive indices on their own -- one quarter-rate integer multiply less per piece)
  uint32_t w = vx_mix32((e >> 5) ^ key);
  return (w >> (e & 31u)) & 0xFu;
}

__device__ __forceinline__ float vx_act(float v, int act) {
  if (act == VX_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
  if (act == VX_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
