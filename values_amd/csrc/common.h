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

// compute units of the current device (256 on a whole MI355X; a compute partition has fewer): the persistent kernels launch one
// workgroup per CU.  One process drives one device, so the first answer is kept.
static inline int vx_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

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
// Round 6: the key of a (seed, layer, sample) stream is TWO words.  Until round 5 a keep-word was vx_mix32(word index ^ key) with one
// 32-bit key: two streams whose keys agreed in the bits above the stream's word count (2^16 words at 64^3 x 8 channels) were the same
// words in XOR-permuted order (probability 2^-16 per pair of streams, ~0.2 such pairs among the 170 streams of one T = 10 volume).  The
// second word enters by ADDITION between the two multiply-xorshift rounds, so streams that meet after the first round part again in
// the second: a permuted copy now needs both words to agree (2^-48 per pair).  One v_add_u32 per hash round (= per 32 elements).
struct vx_dkey { uint32_t a, b; };
__device__ __forceinline__ vx_dkey vx_drop_key(uint32_t seed, uint32_t layer, uint32_t sample) {
  vx_dkey k;
  k.a = vx_mix32(seed * 0x9E3779B1u + layer * 0x85EBCA6Bu + sample * 0xC2B2AE35u + 0x27D4EB2Fu);
  k.b = vx_mix32(seed * 0xC2B2AE3Du + layer * 0x27D4EB2Fu + sample * 0x165667B1u + 0x9E3779B9u);
  return k;
}
// keep-word (32 consecutive elements) number widx of the stream
__device__ __forceinline__ uint32_t vx_drop_word(vx_dkey k, uint32_t widx) {
  uint32_t h = widx ^ k.a;
  h ^= h >> 16;
  h *= 0x7feb352du;
  h += k.b;
  h ^= h >> 15;
  h *= 0x846ca68bu;
  h ^= h >> 16;
  return h;
}
// seed of a launch: the by-value seed plus an optional DEVICE word -- a captured hipGraph replays with the kernel
// arguments it was captured with, so a caller that wants fresh dropout bits per replay updates that word instead
template <typename A>
__device__ __forceinline__ uint32_t vx_seed_of(const A& a, uint32_t by_value) {
  return by_value + (a.seed_dev ? *a.seed_dev : 0u);
}
// keep-bits for elements [e, e+4) of sample-local linear index e (e % 4 == 0)
__device__ __forceinline__ uint32_t vx_drop_bits4(vx_dkey key, uint32_t e) {
  // (the word index enters by XOR: key.a is already a full avalanche of (seed, layer, sample), and the two
  // multiply-xorshift rounds spread consecutive indices on their own -- one quarter-rate integer multiply less per piece)
  const uint32_t w = vx_drop_word(key, e >> 5);
  return (w >> (e & 31u)) & 0xFu;
}

__device__ __forceinline__ float vx_act(float v, int act) {
  if (act == VX_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
  if (act == VX_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
