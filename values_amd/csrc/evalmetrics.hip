// Per-voxel reductions behind the evaluation stage's downstream scalars:
//   vx_ncc_sums     evaluation/metrics/ncc.py:9-25       normalised cross correlation of two uncertainty maps
//   vx_platt_sums   evaluation/metrics/ace.py:13-41      Platt scaling (sklearn.calibration._sigmoid_calibration): the
//                                                        O(voxels) part of every optimiser step -- loss, gradient and
//                                                        Hessian sums of the two-parameter sigmoid fit
//   vx_calib_bins   evaluation/metrics/ace.py:44-90      platt_scale_confid + the 20-bin statistics of calib_stats
// All sums are float64 and DETERMINISTIC: a fixed grid of workgroups leaves one partial row each, a single workgroup
// adds the rows in index order (no atomics), so a rerun gives the same bits.
// "correct" follows ace.py:27-29 / :112-114: the mean prediction compared with each of the R reference segmentations
// (rater_correct = reference_segs == pred_seg, the map repeated per rater), voxels whose reference equals
// ignore_value dropped (ignore_value < 0: none).
#include "common.h"

namespace {
constexpr int EM_BLOCKS = 512;
constexpr int EM_THREADS = 256;

template <int K>
__device__ __forceinline__ void em_block_reduce(double (&v)[K], double* __restrict__ partial) {
  __shared__ double s_red[EM_THREADS / 64][K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double x = v[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) s_red[wave][k] = x;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < EM_THREADS / 64; ++w) t += s_red[w][threadIdx.x];
    partial[(size_t)blockIdx.x * K + threadIdx.x] = t;
  }
}

__global__ __launch_bounds__(256) void em_final_kernel(const double* __restrict__ partial, int nblocks, int K,
                                                       double* __restrict__ out) {
  const int k = threadIdx.x;
  if (k >= K) return;
  double t = 0.0;
  for (int b = 0; b < nblocks; ++b) t += partial[(size_t)b * K + k];
  out[k] = t;
}

__device__ __forceinline__ double em_load(const void* p, int dtype, int64_t i) {
  return dtype == VX_F64 ? reinterpret_cast<const double*>(p)[i] : (double)reinterpret_cast<const float*>(p)[i];
}

// pass 0: sum g, sum p.   pass 1 (means given): sum (g - mg)^2, sum (p - mp)^2, sum (g - mg)(p - mp)
__global__ __launch_bounds__(EM_THREADS) void ncc_kernel(const void* __restrict__ g, int gd, const void* __restrict__ p, int pd,
                                                         int64_t n, int pass, double mg, double mp,
                                                         double* __restrict__ partial) {
  double v[3] = {0.0, 0.0, 0.0};
  for (int64_t i = blockIdx.x * (int64_t)EM_THREADS + threadIdx.x; i < n; i += (int64_t)EM_BLOCKS * EM_THREADS) {
    const double a = em_load(g, gd, i), b = em_load(p, pd, i);
    if (pass == 0) {
      v[0] += a;
      v[1] += b;
    } else {
      const double da = a - mg, db = b - mp;
      v[0] += da * da;
      v[1] += db * db;
      v[2] += da * db;
    }
  }
  em_block_reduce<3>(v, partial);
}

struct PlattArgs {
  const void* unc; int dtype;
  const int32_t* ref; const int32_t* pred;
  int R; int64_t nvox; int ignore_value;
  double A, B, t_pos, t_neg;
};

// out[0] valid voxels (over all raters), out[1] correct ones, out[2] loss, out[3] dA, out[4] dB,
// out[5] sum w F^2, out[6] sum w F, out[7] sum w      (w = P (1 - P); F = -unc as ace.py:32-34 passes it)
__global__ __launch_bounds__(EM_THREADS) void platt_kernel(PlattArgs a, double* __restrict__ partial) {
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t total = (int64_t)a.R * a.nvox;
  for (int64_t i = blockIdx.x * (int64_t)EM_THREADS + threadIdx.x; i < total; i += (int64_t)EM_BLOCKS * EM_THREADS) {
    const int64_t vx = i % a.nvox;
    const int ref = a.ref[i];
    if (a.ignore_value >= 0 && ref == a.ignore_value) continue;
    const bool correct = ref == a.pred[vx];
    const double F = -em_load(a.unc, a.dtype, vx);
    const double T = correct ? a.t_pos : a.t_neg;
    // P = expit(-(A F + B));  loss = -(T log P + (1 - T) log(1 - P)) in the overflow-free form of Platt's pseudo-code
    const double z = a.A * F + a.B;
    double P, loss;
    if (z >= 0) {
      const double e = exp(-z);
      P = e / (1.0 + e);
      loss = T * z + log1p(e);
    } else {
      const double e = exp(z);
      P = 1.0 / (1.0 + e);
      loss = (T - 1.0) * z + log1p(e);
    }
    const double d = T - P, w = P * (1.0 - P);
    v[0] += 1.0;
    v[1] += correct ? 1.0 : 0.0;
    v[2] += loss;
    v[3] += d * F;
    v[4] += d;
    v[5] += w * F * F;
    v[6] += w * F;
    v[7] += w;
  }
  em_block_reduce<8>(v, partial);
}

constexpr int NB = 21;   // len(bins) of calib_stats: np.linspace(0, 1 + 1e-8, 21); bincount(minlength = 21)

struct BinArgs {
  const void* unc; int dtype;
  const int32_t* ref; const int32_t* pred;
  int R; int64_t nvox; int ignore_value;
  double A, B;
  double edges[NB];
};

// per workgroup: bin_sums[21], bin_true[21], bin_total[21] -> partial row of 63 doubles
__global__ __launch_bounds__(EM_THREADS) void calib_bins_kernel(BinArgs a, double* __restrict__ partial) {
  __shared__ double s_h[EM_THREADS / 64][3 * NB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per-lane private histograms would need 63 registers of doubles; instead every lane walks its elements and the wave
  // combines bin by bin with a ballot-free masked reduction: 21 bins x 3 values, all lanes take part (deterministic order)
  double hs[NB], ht[NB], hc[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) { hs[k] = 0.0; ht[k] = 0.0; hc[k] = 0.0; }
  const int64_t total = (int64_t)a.R * a.nvox;
  for (int64_t i = blockIdx.x * (int64_t)EM_THREADS + threadIdx.x; i < total; i += (int64_t)EM_BLOCKS * EM_THREADS) {
    const int64_t vx = i % a.nvox;
    const int ref = a.ref[i];
    if (a.ignore_value >= 0 && ref == a.ignore_value) continue;
    const double conf = -em_load(a.unc, a.dtype, vx);                 // uncalib_confid = -unc  (ace.py:117-121)
    const double prob = 1.0 / (1.0 + exp(conf * a.A + a.B));          // platt_scale_confid (ace.py:44-48)
    int bin = -1;                                                     // np.digitize(prob, bins) - 1
#pragma unroll
    for (int k = 0; k < NB; ++k) bin += (a.edges[k] <= prob) ? 1 : 0;
    const double corr = ref == a.pred[vx] ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const bool hit = bin == k;
      hs[k] += hit ? prob : 0.0;
      ht[k] += hit ? corr : 0.0;
      hc[k] += hit ? 1.0 : 0.0;
    }
  }
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    double x = hs[k], y = ht[k], z = hc[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      x += __shfl_down(x, off, 64);
      y += __shfl_down(y, off, 64);
      z += __shfl_down(z, off, 64);
    }
    if (lane == 0) { s_h[wave][k] = x; s_h[wave][NB + k] = y; s_h[wave][2 * NB + k] = z; }
  }
  __syncthreads();
  if (threadIdx.x < 3 * NB) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < EM_THREADS / 64; ++w) t += s_h[w][threadIdx.x];
    partial[(size_t)blockIdx.x * 3 * NB + threadIdx.x] = t;
  }
}
}  // namespace

extern "C" int64_t vx_evalmetrics_workspace_bytes(void) { return (int64_t)EM_BLOCKS * 3 * NB * sizeof(double); }

extern "C" int vx_ncc_sums(const void* gt, int gt_dtype, const void* pred, int pred_dtype, int64_t n, int pass, double mean_gt,
                           double mean_pred, double* sums, void* workspace, vx_stream_t stream) {
  if (!gt || !pred || !sums || !workspace) VX_FAIL(VX_E_NULL, "vx_ncc_sums: null pointer");
  if (n <= 0) VX_FAIL(VX_E_SHAPE, "vx_ncc_sums: empty map");
  if ((gt_dtype != VX_F32 && gt_dtype != VX_F64) || (pred_dtype != VX_F32 && pred_dtype != VX_F64) || (pass != 0 && pass != 1))
    VX_FAIL(VX_E_DTYPE, "vx_ncc_sums: dtype / pass");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ncc_kernel, dim3(EM_BLOCKS), dim3(EM_THREADS), 0, s, gt, gt_dtype, pred, pred_dtype, n, pass, mean_gt,
                     mean_pred, (double*)workspace);
  hipLaunchKernelGGL(em_final_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, EM_BLOCKS, 3, sums);
  VX_CHECK_LAUNCH("vx_ncc_sums");
  return VX_OK;
}

extern "C" int vx_platt_sums(const void* unc, int dtype, const int32_t* ref, const int32_t* pred, int R, int64_t nvox,
                             int ignore_value, double A, double B, double t_pos, double t_neg, double* sums, void* workspace,
                             vx_stream_t stream) {
  if (!unc || !ref || !pred || !sums || !workspace) VX_FAIL(VX_E_NULL, "vx_platt_sums: null pointer");
  if (R <= 0 || nvox <= 0) VX_FAIL(VX_E_SHAPE, "vx_platt_sums: empty input");
  if (dtype != VX_F32 && dtype != VX_F64) VX_FAIL(VX_E_DTYPE, "vx_platt_sums: dtype %d", dtype);
  PlattArgs a;
  a.unc = unc; a.dtype = dtype; a.ref = ref; a.pred = pred; a.R = R; a.nvox = nvox; a.ignore_value = ignore_value;
  a.A = A; a.B = B; a.t_pos = t_pos; a.t_neg = t_neg;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(platt_kernel, dim3(EM_BLOCKS), dim3(EM_THREADS), 0, s, a, (double*)workspace);
  hipLaunchKernelGGL(em_final_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, EM_BLOCKS, 8, sums);
  VX_CHECK_LAUNCH("vx_platt_sums");
  return VX_OK;
}

extern "C" int vx_calib_bins(const void* unc, int dtype, const int32_t* ref, const int32_t* pred, int R, int64_t nvox,
                             int ignore_value, double A, double B, const double* edges21, double* bins63, void* workspace,
                             vx_stream_t stream) {
  if (!unc || !ref || !pred || !edges21 || !bins63 || !workspace) VX_FAIL(VX_E_NULL, "vx_calib_bins: null pointer");
  if (R <= 0 || nvox <= 0) VX_FAIL(VX_E_SHAPE, "vx_calib_bins: empty input");
  if (dtype != VX_F32 && dtype != VX_F64) VX_FAIL(VX_E_DTYPE, "vx_calib_bins: dtype %d", dtype);
  BinArgs a;
  a.unc = unc; a.dtype = dtype; a.ref = ref; a.pred = pred; a.R = R; a.nvox = nvox; a.ignore_value = ignore_value;
  a.A = A; a.B = B;
  for (int k = 0; k < NB; ++k) a.edges[k] = edges21[k];   // HOST array (np.linspace, bit for bit the reference's edges)
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(calib_bins_kernel, dim3(EM_BLOCKS), dim3(EM_THREADS), 0, s, a, (double*)workspace);
  hipLaunchKernelGGL(em_final_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, EM_BLOCKS, 3 * NB, bins63);
  VX_CHECK_LAUNCH("vx_calib_bins");
  return VX_OK;
}
