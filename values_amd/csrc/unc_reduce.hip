// K9-K12: fused N-sample softmax -> {mean prob, predictive entropy, expected entropy, mutual
// information, argmax} reduction.  One pass over the T*C values of each voxel, 16-byte loads,
// everything else in registers: a pure HBM stream (T*C*4 B read, (3+C)*4+1 B written per voxel).
//
// Semantics follow calculate_uncertainty (uncertainty_modeling/test_3D.py:486-518):
//   mean over T first, natural log, terms p*log(p) that are NaN (p == 0, or a negative/NaN input)
//   are skipped, float32 maps.  With from_logits the class softmax (test_3D.py:472) is fused and the
//   entropy of a sample is computed from log-softmax, which is the same value without the 0*log(0)
//   singularity.
#include "common.h"

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<double> { static constexpr int N = 2; };

template <typename TIn> __device__ __forceinline__ TIn vx_log(TIn v);
template <> __device__ __forceinline__ float vx_log<float>(float v) { return logf(v); }
template <> __device__ __forceinline__ double vx_log<double>(double v) { return log(v); }
template <typename TIn> __device__ __forceinline__ TIn vx_exp(TIn v);
template <> __device__ __forceinline__ float vx_exp<float>(float v) { return expf(v); }
template <> __device__ __forceinline__ double vx_exp<double>(double v) { return exp(v); }
// The softmax of the per-sample loop (round 5): the pass over the logits was bound by its ARITHMETIC, not by memory -- libm's
// expf / logf / IEEE division are ~15 / ~20 / ~10 vector instructions each, ~85 per (voxel, sample) with two classes, 0.24 ms per
// 32 x 10 x 2 x 64^3 logits where the bytes need 0.15.  The arguments are confined: z - max <= 0, 1 <= den <= C.  There the
// hardware forms are accurate in ABSOLUTE terms -- exp2(d log2 e) is off by <= |d| 2^-23 relative, and |d| e^d <= 1 / e: < 5e-8
// absolute on a probability; v_log_f32 on [1, C] and v_rcp_f32 are good to 1 ulp -- three orders below the 1e-4 the maps are held to
// (and below the 2e-6 of the fixture tests).  The class of the maximum contributes exp(0) = 1 exactly and is not evaluated.
template <typename TIn> __device__ __forceinline__ TIn vx_exp_np(TIn v);      // v <= 0
template <> __device__ __forceinline__ float vx_exp_np<float>(float v) { return __expf(v); }
template <> __device__ __forceinline__ double vx_exp_np<double>(double v) { return exp(v); }
template <typename TIn> __device__ __forceinline__ TIn vx_log_den(TIn v);     // 1 <= v <= C
template <> __device__ __forceinline__ float vx_log_den<float>(float v) { return __logf(v); }
template <> __device__ __forceinline__ double vx_log_den<double>(double v) { return log(v); }
template <typename TIn> __device__ __forceinline__ TIn vx_inv(TIn v);
template <> __device__ __forceinline__ float vx_inv<float>(float v) { return __builtin_amdgcn_rcpf(v); }
template <> __device__ __forceinline__ double vx_inv<double>(double v) { return 1.0 / v; }

template <typename TIn, int VEC>
__device__ __forceinline__ void load_vec(const TIn* p, TIn (&v)[VEC]) {
  if (VEC == 1) {
    v[0] = p[0];
  } else {
    typedef TIn vt __attribute__((ext_vector_type(VEC)));
    const vt t = *reinterpret_cast<const vt*>(p);
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = t[k];
  }
}

template <int VEC>
__device__ __forceinline__ void store_f32(float* p, const float (&v)[VEC]) {
  if (VEC == 4) {
    *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[VEC > 1 ? 1 : 0], v[VEC > 2 ? 2 : 0], v[VEC > 3 ? 3 : 0]};
  } else if (VEC == 2) {
    *reinterpret_cast<f32x2*>(p) = (f32x2){v[0], v[VEC > 1 ? 1 : 0]};
  } else {
    p[0] = v[0];
  }
}

template <int VEC>
__device__ __forceinline__ void store_u8(uint8_t* p, const int (&v)[VEC]) {
  if (VEC == 4) {
    *reinterpret_cast<uint32_t*>(p) = (uint32_t)v[0] | ((uint32_t)v[VEC > 1 ? 1 : 0] << 8) |
                                      ((uint32_t)v[VEC > 2 ? 2 : 0] << 16) | ((uint32_t)v[VEC > 3 ? 3 : 0] << 24);
  } else {
#pragma unroll
    for (int k = 0; k < VEC; ++k) p[k] = (uint8_t)v[k];
  }
}

// ---- probabilities in, any C: class-outer loop, only scalars live ----
// Optional extras of vx_unc_reduce_ex, all in the same pass:
//   variance  : mean over classes of the population variance over the T samples (north_star's fourth map; the reference
//               has none).  Accumulated as deviations from sample 0 (d = p_t - p_0): sum d and sum d^2 stay small
//               where the samples agree, so var = sum d^2 / T - (sum d / T)^2 does not cancel like sum p^2 / T - mean^2.
//   in_count  : inputs divided by max(count, 1) on load (the normalised sliding-window sums)
//   out_count : maps divided by max(count, 1) on store (DataCarrier3D.save_data, data_carrier_3D.py:323-337, applied to
//               maps of the UN-normalised sums: quirk D10); the variance by its square
struct UncExtra {
  float* variance;
  const float* in_count;
  const float* out_count;
};

template <int VEC>
__device__ __forceinline__ void load_inv_count(const float* cnt, int64_t i, float (&inv)[VEC]) {
  float c[VEC];
  load_vec<float, VEC>(cnt + i, c);
#pragma unroll
  for (int k = 0; k < VEC; ++k) inv[k] = fmaxf(c[k], 1.f);   // np.clip(count, 1, None); callers DIVIDE by it
}

template <typename TIn, int VEC, bool EX>
__global__ __launch_bounds__(256) void unc_reduce_prob_kernel(const TIn* __restrict__ x, int T, int C, int64_t nvox,
                                                              float* __restrict__ mean_prob,
                                                              float* __restrict__ pred_entropy,
                                                              float* __restrict__ exp_entropy,
                                                              float* __restrict__ mutual_info,
                                                              uint8_t* __restrict__ argmax, UncExtra ex) {
  const int b = blockIdx.y;
  const TIn* xb = x + (size_t)b * T * C * nvox;
  const int64_t ngroups = nvox / VEC;
  for (int64_t gi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; gi < ngroups; gi += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v0 = gi * VEC;
    float pe[VEC];   // running -sum_c mean*log(mean), rounded to f32 after every class like the reference
    TIn ee[VEC];     // sum over (t, c) of p*log(p)
    TIn best[VEC];
    int besti[VEC];
    TIn var[VEC];
    float iin[VEC], iout[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { pe[k] = 0.f; ee[k] = 0; best[k] = 0; besti[k] = 0; var[k] = 0; iin[k] = 1.f; iout[k] = 1.f; }
    if (EX && ex.in_count) load_inv_count<VEC>(ex.in_count, (int64_t)b * nvox + v0, iin);
    if (EX && ex.out_count) load_inv_count<VEC>(ex.out_count, (int64_t)b * nvox + v0, iout);
    for (int c = 0; c < C; ++c) {
      TIn s[VEC], p0[VEC], d1[VEC], d2[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) { s[k] = 0; p0[k] = 0; d1[k] = 0; d2[k] = 0; }
      for (int t = 0; t < T; ++t) {
        TIn p[VEC];
        load_vec<TIn, VEC>(xb + ((size_t)t * C + c) * nvox + v0, p);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          if (EX && ex.in_count) p[k] /= (TIn)iin[k];
          s[k] += p[k];
          const TIn term = p[k] * vx_log<TIn>(p[k]);
          if (term == term) ee[k] += term;  // NaN-skip (test_3D.py:503-504)
          if (EX) {
            if (t == 0) p0[k] = p[k];
            const TIn d = p[k] - p0[k];
            d1[k] += d;
            d2[k] += d * d;
          }
        }
      }
      float mo[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const TIn mean = s[k] / (TIn)T;
        mo[k] = (float)mean / iout[k];
        const TIn term = mean * vx_log<TIn>(mean);
        if (term == term) pe[k] = (float)((TIn)pe[k] + term);  // f32 accumulator (test_3D.py:490-494)
        if (c == 0 || mean > best[k]) { best[k] = mean; besti[k] = c; }
        if (EX) {
          const TIn md = d1[k] / (TIn)T;
          const TIn vv = d2[k] / (TIn)T - md * md;
          var[k] += vv > (TIn)0 ? vv : (TIn)0;
        }
      }
      if (mean_prob) store_f32<VEC>(mean_prob + ((size_t)b * C + c) * nvox + v0, mo);
    }
    float o_pe[VEC], o_ee[VEC], o_mi[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      o_pe[k] = -pe[k];
      o_ee[k] = (float)(-ee[k] / (TIn)T);
      o_mi[k] = o_pe[k] - o_ee[k];
      if (EX && ex.out_count) { o_pe[k] /= iout[k]; o_ee[k] /= iout[k]; o_mi[k] /= iout[k]; }
    }
    store_f32<VEC>(pred_entropy + (size_t)b * nvox + v0, o_pe);
    store_f32<VEC>(exp_entropy + (size_t)b * nvox + v0, o_ee);
    store_f32<VEC>(mutual_info + (size_t)b * nvox + v0, o_mi);
    if (argmax) store_u8<VEC>(argmax + (size_t)b * nvox + v0, besti);
    if (EX && ex.variance) {
      float o_v[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) o_v[k] = (float)(var[k] / (TIn)C) / (iout[k] * iout[k]);
      store_f32<VEC>(ex.variance + (size_t)b * nvox + v0, o_v);
    }
  }
}

// ---- logits in, C known at compile time (<= 8): sample-outer loop, softmax in registers ----
template <typename TIn, int C, int VEC, bool EX>
__global__ __launch_bounds__(256) void unc_reduce_logit_kernel(const TIn* __restrict__ x, int T, int64_t nvox,
                                                               float* __restrict__ mean_prob,
                                                               float* __restrict__ pred_entropy,
                                                               float* __restrict__ exp_entropy,
                                                               float* __restrict__ mutual_info,
                                                               uint8_t* __restrict__ argmax,
                                                               uint8_t* __restrict__ sample_argmax, UncExtra ex) {
  const int b = blockIdx.y;
  const TIn* xb = x + (size_t)b * T * C * nvox;
  const int64_t ngroups = nvox / VEC;
  for (int64_t gi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; gi < ngroups; gi += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v0 = gi * VEC;
    TIn sum[C][VEC];
    TIn ee[VEC];
    // variance (EX): deviations from sample 0 per class, see UncExtra
    TIn p0[EX ? C : 1][VEC], d1[EX ? C : 1][VEC], d2[EX ? C : 1][VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      ee[k] = 0;
#pragma unroll
      for (int c = 0; c < C; ++c) sum[c][k] = 0;
#pragma unroll
      for (int c = 0; c < (EX ? C : 1); ++c) { p0[c][k] = 0; d1[c][k] = 0; d2[c][k] = 0; }
    }
    for (int t = 0; t < T; ++t) {
      TIn z[C][VEC];
#pragma unroll
      for (int c = 0; c < C; ++c) load_vec<TIn, VEC>(xb + ((size_t)t * C + c) * nvox + v0, z[c]);
      int am[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        TIn m = z[0][k];
        int mi = 0;
#pragma unroll
        for (int c = 1; c < C; ++c)
          if (z[c][k] > m) { m = z[c][k]; mi = c; }
        am[k] = mi;
        TIn e[C];
        TIn den = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) { e[c] = c == mi ? (TIn)1 : vx_exp_np<TIn>(z[c][k] - m); den += e[c]; }
        const TIn inv = vx_inv<TIn>(den);
        const TIn lden = vx_log_den<TIn>(den);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const TIn p = e[c] * inv;
          sum[c][k] += p;
          ee[k] += p * ((z[c][k] - m) - lden);  // p*log(p) via log-softmax
          if (EX) {
            if (t == 0) p0[c][k] = p;
            const TIn d = p - p0[c][k];
            d1[c][k] += d;
            d2[c][k] += d * d;
          }
        }
      }
      if (sample_argmax) store_u8<VEC>(sample_argmax + ((size_t)b * T + t) * nvox + v0, am);
    }
    float pe[VEC];
    int besti[VEC];
    float iout[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) iout[k] = 1.f;
    const bool counted = EX && ex.out_count;      // (uniform: without counts nothing is divided by 1.f -- an IEEE division is ~11 instructions)
    if (counted) load_inv_count<VEC>(ex.out_count, (int64_t)b * nvox + v0, iout);
    // 1 / T once: the expected entropy and the variance moments are scaled by it (1 ulp from the division; the class means, which
    // decide the arg-max, keep the division)
    const TIn invT = (TIn)1 / (TIn)T;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      pe[k] = 0.f;
      TIn best = 0;
      besti[k] = 0;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const TIn mean = sum[c][k] / (TIn)T;
        const TIn term = mean * vx_log<TIn>(mean);
        if (term == term) pe[k] = (float)((TIn)pe[k] + term);
        if (c == 0 || mean > best) { best = mean; besti[k] = c; }
      }
    }
    if (mean_prob) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float mo[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) { mo[k] = (float)(sum[c][k] / (TIn)T); if (counted) mo[k] /= iout[k]; }
        store_f32<VEC>(mean_prob + ((size_t)b * C + c) * nvox + v0, mo);
      }
    }
    float o_pe[VEC], o_ee[VEC], o_mi[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      o_pe[k] = -pe[k];
      o_ee[k] = (float)(-ee[k] * invT);
      o_mi[k] = o_pe[k] - o_ee[k];
      if (counted) { o_pe[k] /= iout[k]; o_ee[k] /= iout[k]; o_mi[k] /= iout[k]; }
    }
    store_f32<VEC>(pred_entropy + (size_t)b * nvox + v0, o_pe);
    store_f32<VEC>(exp_entropy + (size_t)b * nvox + v0, o_ee);
    store_f32<VEC>(mutual_info + (size_t)b * nvox + v0, o_mi);
    if (argmax) store_u8<VEC>(argmax + (size_t)b * nvox + v0, besti);
    if (EX && ex.variance) {
      float o_v[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        TIn var = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const TIn md = d1[c][k] * invT;
          const TIn vv = d2[c][k] * invT - md * md;
          var += vv > (TIn)0 ? vv : (TIn)0;
        }
        o_v[k] = (float)(var * ((TIn)1 / (TIn)C));
        if (counted) o_v[k] /= iout[k] * iout[k];
      }
      store_f32<VEC>(ex.variance + (size_t)b * nvox + v0, o_v);
    }
  }
}

// ---- sufficient statistics for member-/sample-sharded reductions (multi-GPU ensembles) ----
// stats[b][c][v] += sum_t p_tc ;  stats[b][C][v] += sum_t sum_c p_tc log p_tc.   Ranks add their members' passes
// into their own buffer, one RCCL sum-reduce combines them, and the finalize kernel produces the maps:
// H[S1/T], -S2/T, their difference -- the same quantities as the single-pass kernel, summed in a different order.
template <int C, int VEC>
__global__ __launch_bounds__(256) void unc_stats_accumulate_kernel(const float* __restrict__ x, int T, int64_t nvox,
                                                                   float* __restrict__ stats) {
  const int b = blockIdx.y;
  const float* xb = x + (size_t)b * T * C * nvox;
  float* sb = stats + (size_t)b * (C + 1) * nvox;
  const int64_t ngroups = nvox / VEC;
  for (int64_t gi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; gi < ngroups; gi += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v0 = gi * VEC;
    float sum[C][VEC], ee[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      ee[k] = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) sum[c][k] = 0.f;
    }
    for (int t = 0; t < T; ++t) {
      float z[C][VEC];
#pragma unroll
      for (int c = 0; c < C; ++c) load_vec<float, VEC>(xb + ((size_t)t * C + c) * nvox + v0, z[c]);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        float m = z[0][k];
#pragma unroll
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c][k]);
        float e[C], den = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { e[c] = expf(z[c][k] - m); den += e[c]; }
        const float inv = 1.f / den, lden = logf(den);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float p = e[c] * inv;
          sum[c][k] += p;
          ee[k] += p * ((z[c][k] - m) - lden);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float o[VEC];
      load_vec<float, VEC>(sb + (size_t)c * nvox + v0, o);
#pragma unroll
      for (int k = 0; k < VEC; ++k) o[k] += sum[c][k];
      store_f32<VEC>(sb + (size_t)c * nvox + v0, o);
    }
    float o[VEC];
    load_vec<float, VEC>(sb + (size_t)C * nvox + v0, o);
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] += ee[k];
    store_f32<VEC>(sb + (size_t)C * nvox + v0, o);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void unc_stats_finalize_kernel(const float* __restrict__ stats, int C, float inv_T,
                                                                 int64_t nvox, float* __restrict__ mean_prob,
                                                                 float* __restrict__ pred_entropy,
                                                                 float* __restrict__ exp_entropy,
                                                                 float* __restrict__ mutual_info,
                                                                 uint8_t* __restrict__ argmax) {
  const int b = blockIdx.y;
  const float* sb = stats + (size_t)b * (C + 1) * nvox;
  const int64_t ngroups = nvox / VEC;
  for (int64_t gi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; gi < ngroups; gi += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v0 = gi * VEC;
    float pe[VEC], best[VEC];
    int besti[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { pe[k] = 0.f; best[k] = 0.f; besti[k] = 0; }
    for (int c = 0; c < C; ++c) {
      float s[VEC], mo[VEC];
      load_vec<float, VEC>(sb + (size_t)c * nvox + v0, s);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float mean = s[k] * inv_T;
        mo[k] = mean;
        const float term = mean * logf(mean);
        if (term == term) pe[k] += term;
        if (c == 0 || mean > best[k]) { best[k] = mean; besti[k] = c; }
      }
      if (mean_prob) store_f32<VEC>(mean_prob + ((size_t)b * C + c) * nvox + v0, mo);
    }
    float s2[VEC], o_pe[VEC], o_ee[VEC], o_mi[VEC];
    load_vec<float, VEC>(sb + (size_t)C * nvox + v0, s2);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      o_pe[k] = -pe[k];
      o_ee[k] = -s2[k] * inv_T;
      o_mi[k] = o_pe[k] - o_ee[k];
    }
    store_f32<VEC>(pred_entropy + (size_t)b * nvox + v0, o_pe);
    store_f32<VEC>(exp_entropy + (size_t)b * nvox + v0, o_ee);
    store_f32<VEC>(mutual_info + (size_t)b * nvox + v0, o_mi);
    if (argmax) store_u8<VEC>(argmax + (size_t)b * nvox + v0, besti);
  }
}

extern "C" int vx_unc_stats_accumulate(const float* logits, int B, int T, int C, int64_t nvox, float* stats,
                                       vx_stream_t stream) {
  if (B <= 0 || T <= 0 || nvox < 0) VX_FAIL(VX_E_SHAPE, "vx_unc_stats_accumulate: bad shape");
  if (nvox == 0) return VX_OK;
  if (!logits || !stats) VX_FAIL(VX_E_NULL, "vx_unc_stats_accumulate: null pointer");
  const bool vec = vx_aligned16(logits) && vx_aligned16(stats) && nvox % 4 == 0;
  const int64_t ngroups = vec ? nvox / 4 : nvox;
  int bx = (int)((ngroups + 255) / 256);
  if (bx > 8192) bx = 8192;
  dim3 grid((unsigned)bx, (unsigned)B);
  hipStream_t s = (hipStream_t)stream;
#define VX_SA(CC)                                                                                              \
  case CC:                                                                                                     \
    if (vec) hipLaunchKernelGGL((unc_stats_accumulate_kernel<CC, 4>), grid, dim3(256), 0, s, logits, T, nvox, stats); \
    else hipLaunchKernelGGL((unc_stats_accumulate_kernel<CC, 1>), grid, dim3(256), 0, s, logits, T, nvox, stats);     \
    break;
  switch (C) {
    VX_SA(2) VX_SA(3) VX_SA(4) VX_SA(5) VX_SA(6) VX_SA(7) VX_SA(8)
    default: VX_FAIL(VX_E_SHAPE, "vx_unc_stats_accumulate: 2 <= C <= 8 (got %d)", C);
  }
#undef VX_SA
  VX_CHECK_LAUNCH("vx_unc_stats_accumulate");
  return VX_OK;
}

extern "C" int vx_unc_stats_finalize(const float* stats, int B, int T_total, int C, int64_t nvox, float* mean_prob,
                                     float* pred_entropy, float* exp_entropy, float* mutual_info, uint8_t* argmax,
                                     vx_stream_t stream) {
  if (B <= 0 || T_total <= 0 || C <= 0 || nvox < 0) VX_FAIL(VX_E_SHAPE, "vx_unc_stats_finalize: bad shape");
  if (nvox == 0) return VX_OK;
  if (!stats || !pred_entropy || !exp_entropy || !mutual_info) VX_FAIL(VX_E_NULL, "vx_unc_stats_finalize: null pointer");
  const bool vec = vx_aligned16(stats) && vx_aligned16(pred_entropy) && vx_aligned16(exp_entropy) &&
                   vx_aligned16(mutual_info) && (!mean_prob || vx_aligned16(mean_prob)) &&
                   (!argmax || (((uintptr_t)argmax) & 3u) == 0) && nvox % 4 == 0;
  const int64_t ngroups = vec ? nvox / 4 : nvox;
  int bx = (int)((ngroups + 255) / 256);
  if (bx > 8192) bx = 8192;
  dim3 grid((unsigned)bx, (unsigned)B);
  hipStream_t s = (hipStream_t)stream;
  if (vec)
    hipLaunchKernelGGL(unc_stats_finalize_kernel<4>, grid, dim3(256), 0, s, stats, C, 1.f / (float)T_total, nvox, mean_prob,
                       pred_entropy, exp_entropy, mutual_info, argmax);
  else
    hipLaunchKernelGGL(unc_stats_finalize_kernel<1>, grid, dim3(256), 0, s, stats, C, 1.f / (float)T_total, nvox, mean_prob,
                       pred_entropy, exp_entropy, mutual_info, argmax);
  VX_CHECK_LAUNCH("vx_unc_stats_finalize");
  return VX_OK;
}

// class softmax of planar logits [R][C][nvox] (F.softmax(dim=1), test_2D.py:302,315) for class counts beyond the
// register-resident fused path (C > 8: the 2D data sets have 19-24 classes)
__global__ __launch_bounds__(256) void softmax_planar_kernel(const float* __restrict__ x, int C, int64_t nvox, int64_t total,
                                                             float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nvox, v = i - r * nvox;
    const float* p = x + (size_t)r * C * nvox + v;
    float* o = out + (size_t)r * C * nvox + v;
    float m = p[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, p[(size_t)c * nvox]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(p[(size_t)c * nvox] - m);
    const float inv = 1.f / den;
    for (int c = 0; c < C; ++c) o[(size_t)c * nvox] = expf(p[(size_t)c * nvox] - m) * inv;
  }
}

extern "C" int vx_softmax_planar(const float* logits, int64_t R, int C, int64_t nvox, float* out, vx_stream_t stream) {
  if (R < 0 || C <= 0 || nvox < 0) VX_FAIL(VX_E_SHAPE, "vx_softmax_planar: bad shape");
  if (R == 0 || nvox == 0) return VX_OK;
  if (!logits || !out) VX_FAIL(VX_E_NULL, "vx_softmax_planar: null pointer");
  const int64_t total = R * nvox;
  int bx = (int)((total + 255) / 256);
  if (bx > 16384) bx = 16384;
  hipLaunchKernelGGL(softmax_planar_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, logits, C, nvox, total, out);
  VX_CHECK_LAUNCH("vx_softmax_planar");
  return VX_OK;
}

// per-sample argmax over classes for the probability path (data_carrier_3D.py:281-283)
template <typename TIn>
__global__ __launch_bounds__(256) void sample_argmax_kernel(const TIn* __restrict__ x, int C, int64_t nvox, int64_t nbt,
                                                            uint8_t* __restrict__ out) {
  const int64_t total = nbt * nvox;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bt = i / nvox, v = i - bt * nvox;
    const TIn* p = x + (size_t)bt * C * nvox + v;
    TIn best = p[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
      const TIn q = p[(size_t)c * nvox];
      if (q > best) { best = q; bi = c; }
    }
    out[i] = (uint8_t)bi;
  }
}

template <typename TIn>
__global__ __launch_bounds__(256) void one_minus_msr_kernel(const TIn* __restrict__ x, int C, int64_t nvox,
                                                            TIn* __restrict__ out) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
    TIn best = x[v];
    for (int c = 1; c < C; ++c) {
      const TIn q = x[(size_t)c * nvox + v];
      best = q > best ? q : best;
    }
    out[v] = (TIn)1 - best;
  }
}

template <typename TIn, int VEC>
static int launch_unc(const TIn* x, int from_logits, int B, int T, int C, int64_t nvox, float* mean_prob,
                      float* pred_entropy, float* exp_entropy, float* mutual_info, uint8_t* argmax,
                      uint8_t* sample_argmax, const UncExtra& ex, hipStream_t s) {
  const int64_t ngroups = nvox / VEC;
  int bx = (int)((ngroups + 255) / 256);
  if (bx > 8192) bx = 8192;
  if (bx < 1) bx = 1;
  dim3 grid((unsigned)bx, (unsigned)B);
  const bool extra = ex.variance || ex.in_count || ex.out_count;
  if (!from_logits) {
    if (extra)
      hipLaunchKernelGGL((unc_reduce_prob_kernel<TIn, VEC, true>), grid, dim3(256), 0, s, x, T, C, nvox, mean_prob, pred_entropy,
                         exp_entropy, mutual_info, argmax, ex);
    else
      hipLaunchKernelGGL((unc_reduce_prob_kernel<TIn, VEC, false>), grid, dim3(256), 0, s, x, T, C, nvox, mean_prob, pred_entropy,
                         exp_entropy, mutual_info, argmax, ex);
    if (sample_argmax) {
      const int64_t total = (int64_t)B * T * nvox;
      int bb = (int)((total + 255) / 256);
      if (bb > 16384) bb = 16384;
      hipLaunchKernelGGL(sample_argmax_kernel<TIn>, dim3(bb), dim3(256), 0, s, x, C, nvox, (int64_t)B * T, sample_argmax);
    }
    return VX_OK;
  }
  if (ex.in_count) VX_FAIL(VX_E_DTYPE, "vx_unc_reduce_ex: in_count applies to probability sums, not to logits");
#define VX_LOGIT(CC)                                                                                                \
  case CC:                                                                                                          \
    if (extra)                                                                                                      \
      hipLaunchKernelGGL((unc_reduce_logit_kernel<TIn, CC, VEC, true>), grid, dim3(256), 0, s, x, T, nvox, mean_prob, \
                         pred_entropy, exp_entropy, mutual_info, argmax, sample_argmax, ex);                        \
    else                                                                                                            \
      hipLaunchKernelGGL((unc_reduce_logit_kernel<TIn, CC, VEC, false>), grid, dim3(256), 0, s, x, T, nvox, mean_prob, \
                         pred_entropy, exp_entropy, mutual_info, argmax, sample_argmax, ex);                        \
    break;
  switch (C) {
    VX_LOGIT(2) VX_LOGIT(3) VX_LOGIT(4) VX_LOGIT(5) VX_LOGIT(6) VX_LOGIT(7) VX_LOGIT(8)
    default: VX_FAIL(VX_E_SHAPE, "vx_unc_reduce: from_logits supports 2 <= C <= 8 (got %d)", C);
  }
#undef VX_LOGIT
  return VX_OK;
}

extern "C" int vx_unc_reduce_ex(const void* x, int dtype, int from_logits, int B, int T, int C, int64_t nvox,
                                const vx_unc_outputs* o, vx_stream_t stream) {
  if (!o) VX_FAIL(VX_E_NULL, "vx_unc_reduce: null outputs");
  if (B <= 0 || T <= 0 || C <= 0 || nvox < 0) VX_FAIL(VX_E_SHAPE, "vx_unc_reduce: B=%d T=%d C=%d nvox=%lld", B, T, C, (long long)nvox);
  if (nvox == 0) return VX_OK;  // empty volume: nothing to do (the reference returns empty maps)
  float *mean_prob = o->mean_prob, *pred_entropy = o->pred_entropy, *exp_entropy = o->exp_entropy, *mutual_info = o->mutual_info;
  uint8_t *argmax = o->argmax, *sample_argmax = o->sample_argmax;
  if (!x || !pred_entropy || !exp_entropy || !mutual_info) VX_FAIL(VX_E_NULL, "vx_unc_reduce: null pointer");
  if (C > 255 && (argmax || sample_argmax)) VX_FAIL(VX_E_SHAPE, "vx_unc_reduce: uint8 argmax needs C <= 255");
  if (dtype != VX_F32 && dtype != VX_F64) VX_FAIL(VX_E_DTYPE, "vx_unc_reduce: dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  UncExtra ex;
  ex.variance = o->variance; ex.in_count = o->in_count; ex.out_count = o->out_count;
  const bool al = vx_aligned16(x) && vx_aligned16(pred_entropy) && vx_aligned16(exp_entropy) && vx_aligned16(mutual_info) &&
                  (!mean_prob || vx_aligned16(mean_prob)) && (!argmax || (((uintptr_t)argmax) & 3u) == 0) &&
                  (!sample_argmax || (((uintptr_t)sample_argmax) & 3u) == 0) && (!ex.variance || vx_aligned16(ex.variance)) &&
                  (!ex.in_count || vx_aligned16(ex.in_count)) && (!ex.out_count || vx_aligned16(ex.out_count));
  int rc;
  if (dtype == VX_F32) {
    if (al && nvox % 4 == 0)
      rc = launch_unc<float, 4>((const float*)x, from_logits, B, T, C, nvox, mean_prob, pred_entropy, exp_entropy, mutual_info, argmax, sample_argmax, ex, s);
    else
      rc = launch_unc<float, 1>((const float*)x, from_logits, B, T, C, nvox, mean_prob, pred_entropy, exp_entropy, mutual_info, argmax, sample_argmax, ex, s);
  } else {
    if (al && nvox % 4 == 0)
      rc = launch_unc<double, 2>((const double*)x, from_logits, B, T, C, nvox, mean_prob, pred_entropy, exp_entropy, mutual_info, argmax, sample_argmax, ex, s);
    else
      rc = launch_unc<double, 1>((const double*)x, from_logits, B, T, C, nvox, mean_prob, pred_entropy, exp_entropy, mutual_info, argmax, sample_argmax, ex, s);
  }
  if (rc != VX_OK) return rc;
  VX_CHECK_LAUNCH("vx_unc_reduce");
  return VX_OK;
}

extern "C" int vx_unc_reduce(const void* x, int dtype, int from_logits, int B, int T, int C, int64_t nvox,
                             float* mean_prob, float* pred_entropy, float* exp_entropy, float* mutual_info,
                             uint8_t* argmax, uint8_t* sample_argmax, vx_stream_t stream) {
  vx_unc_outputs o;
  o.mean_prob = mean_prob; o.pred_entropy = pred_entropy; o.exp_entropy = exp_entropy; o.mutual_info = mutual_info;
  o.variance = nullptr; o.argmax = argmax; o.sample_argmax = sample_argmax; o.in_count = nullptr; o.out_count = nullptr;
  return vx_unc_reduce_ex(x, dtype, from_logits, B, T, C, nvox, &o, stream);
}

extern "C" int vx_one_minus_msr(const void* x, int dtype, int C, int64_t nvox, void* out, vx_stream_t stream) {
  if (!x || !out) VX_FAIL(VX_E_NULL, "vx_one_minus_msr: null pointer");
  if (C <= 0 || nvox < 0) VX_FAIL(VX_E_SHAPE, "vx_one_minus_msr: bad shape");
  if (nvox == 0) return VX_OK;
  int bx = (int)((nvox + 255) / 256);
  if (bx > 16384) bx = 16384;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == VX_F32)
    hipLaunchKernelGGL(one_minus_msr_kernel<float>, dim3(bx), dim3(256), 0, s, (const float*)x, C, nvox, (float*)out);
  else if (dtype == VX_F64)
    hipLaunchKernelGGL(one_minus_msr_kernel<double>, dim3(bx), dim3(256), 0, s, (const double*)x, C, nvox, (double*)out);
  else
    VX_FAIL(VX_E_DTYPE, "vx_one_minus_msr: dtype %d", dtype);
  VX_CHECK_LAUNCH("vx_one_minus_msr");
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Softmax variance (named in BASELINE.json's north_star; the reference computes none, SURVEY D3 -- so the definition
// is this build's: the population variance over the T samples of each class probability, averaged over classes):
//     var[v] = 1/C * sum_c ( 1/T sum_t p_tc^2 - (1/T sum_t p_tc)^2 )
// One pass over the logits (softmax fused) or probabilities, float32 in / out, fp32 accumulation per voxel.
__global__ __launch_bounds__(256) void softmax_variance_kernel(const float* __restrict__ x, int from_logits, int B, int T, int C,
                                                               int64_t nvox, float* __restrict__ out) {
  const int64_t total = (int64_t)B * nvox;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / nvox, v = i - b * nvox;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
      float s1 = 0.f, s2 = 0.f;
      for (int t = 0; t < T; ++t) {
        const float* px = x + ((size_t)(b * T + t) * C) * nvox + v;
        float p;
        if (from_logits) {
          float mx = px[0];
          for (int k = 1; k < C; ++k) mx = fmaxf(mx, px[(size_t)k * nvox]);
          float den = 0.f;
          for (int k = 0; k < C; ++k) den += expf(px[(size_t)k * nvox] - mx);
          p = expf(px[(size_t)c * nvox] - mx) / den;
        } else {
          p = px[(size_t)c * nvox];
        }
        s1 += p;
        s2 = fmaf(p, p, s2);
      }
      const float m = s1 / (float)T;
      acc += fmaxf(s2 / (float)T - m * m, 0.f);
    }
    out[i] = acc / (float)C;
  }
}

extern "C" int vx_softmax_variance(const float* x, int from_logits, int B, int T, int C, int64_t nvox, float* out,
                                   vx_stream_t stream) {
  if (B <= 0 || T <= 0 || C <= 0 || nvox < 0) VX_FAIL(VX_E_SHAPE, "vx_softmax_variance: bad shape");
  if (nvox == 0) return VX_OK;
  if (!x || !out) VX_FAIL(VX_E_NULL, "vx_softmax_variance: null pointer");
  const int64_t total = (int64_t)B * nvox;
  int bx = (int)((total + 255) / 256);
  if (bx > 16384) bx = 16384;
  hipLaunchKernelGGL(softmax_variance_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, x, from_logits, B, T, C, nvox, out);
  VX_CHECK_LAUNCH("vx_softmax_variance");
  return VX_OK;
}
