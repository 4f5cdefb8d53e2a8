// Segmentation-metric reductions that follow the maps in run_test (test_3D.py:250-358, 537-575):
//   * vx_mask_agreement : for a set of M label masks (the T per-sample argmax masks and the R rater masks of one
//     image), the class-wise agreement counts  I[i][j][c] = #{v : mask_i(v) == c and mask_j(v) == c}  of every pair.
//     Every hard Dice the reference asks torchmetrics for -- Dice(pred_t, gt_r), the pooled pred/gt, pred/pred and
//     gt/gt distances of the generalised energy distance, the per-rater and per-prediction maxima -- is a ratio of
//     sums of these integers (tp = I, fp = I[i][i] - I, fn = I[j][j] - I), so ONE pass over the masks replaces the
//     T*R + T*T + R*R + 2*T*R mask comparisons of calculate_ged.
//   * vx_soft_metric_sums : per rater and class  sum_v p_c [gt == c],  sum_v [gt == c],  sum_v p_c  and
//     sum_v log p_gt(v)  for SoftDiceLoss + NLLLoss of calculate_test_metrics (loss_modules.py:7-97).
// Both are HBM-bound scans of a few MB; integer counts are exact and order-independent, the float sums are
// accumulated in fp64 per workgroup and combined in a fixed order (deterministic).
#include "common.h"

constexpr int MA_MAXM = 32;   // masks per call
constexpr int MA_MAXC = 8;    // classes

__global__ __launch_bounds__(256) void mask_agreement_kernel(const uint8_t* __restrict__ masks, int M, int C, int64_t nvox,
                                                             unsigned long long* __restrict__ out) {
  __shared__ unsigned long long bm[4][MA_MAXC][MA_MAXM];   // per wave: lanes whose mask i has class c
  __shared__ unsigned cnt[MA_MAXM * MA_MAXM * MA_MAXC];     // workgroup counters [i][j][c]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int npair = M * M * C;
  for (int i = tid; i < npair; i += 256) cnt[i] = 0;
  __syncthreads();
  const int64_t nchunk = (nvox + 63) / 64;
  for (int64_t ch = (int64_t)blockIdx.x * 4 + wave; ch < nchunk; ch += (int64_t)gridDim.x * 4) {
    const int64_t v = ch * 64 + lane;
    for (int i = 0; i < M; ++i) {
      const int l = v < nvox ? (int)masks[(size_t)i * nvox + v] : -1;
      for (int c = 0; c < C; ++c) {
        const unsigned long long b = __ballot(l == c);
        if (lane == 0) bm[wave][c][i] = b;
      }
    }
    // (same wave: LDS operations complete in order, no barrier needed)
    for (int p = lane; p < M * M; p += 64) {
      const int i = p / M, j = p - i * M;
      for (int c = 0; c < C; ++c) {
        const unsigned n = (unsigned)__popcll(bm[wave][c][i] & bm[wave][c][j]);
        if (n) atomicAdd(&cnt[p * C + c], n);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < npair; i += 256)
    if (cnt[i]) atomicAdd(&out[i], (unsigned long long)cnt[i]);
}

extern "C" int vx_mask_agreement(const uint8_t* masks, int M, int C, int64_t nvox, uint64_t* counts, vx_stream_t stream) {
  if (M <= 0 || M > MA_MAXM || C <= 0 || C > MA_MAXC || nvox < 0)
    VX_FAIL(VX_E_SHAPE, "vx_mask_agreement: M=%d (1..%d) C=%d (1..%d)", M, MA_MAXM, C, MA_MAXC);
  if (!counts) VX_FAIL(VX_E_NULL, "vx_mask_agreement: null output");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)M * M * C * sizeof(uint64_t), s);
  if (e != hipSuccess) VX_FAIL((int)e, "vx_mask_agreement: memset: %s", hipGetErrorString(e));
  if (nvox == 0) return VX_OK;
  if (!masks) VX_FAIL(VX_E_NULL, "vx_mask_agreement: null masks");
  const int64_t nchunk = (nvox + 63) / 64;
  int bx = (int)((nchunk + 3) / 4);
  if (bx > 1024) bx = 1024;
  // (32-bit workgroup counters: a workgroup would need 2^26 chunks = 4 G voxels of its own to wrap one)
  hipLaunchKernelGGL(mask_agreement_kernel, dim3(bx), dim3(256), 0, s, masks, M, C, nvox,
                     reinterpret_cast<unsigned long long*>(counts));
  VX_CHECK_LAUNCH("vx_mask_agreement");
  return VX_OK;
}

// out[r][c][0] = sum p_c [gt_r == c], [1] = sum [gt_r == c], [2] = sum p_c ; out_nll[r] = sum log p_{gt_r(v)}(v)
__global__ __launch_bounds__(256) void soft_metric_partial_kernel(const float* __restrict__ p, const uint8_t* __restrict__ gt,
                                                                  int C, int R, int64_t nvox, double* __restrict__ part) {
  // part [gridDim.x][R][C*3 + 1]
  __shared__ double red[256];
  const int tid = threadIdx.x;
  const int per = C * 3 + 1;
  for (int r = 0; r < R; ++r) {
    for (int k = 0; k < per; ++k) {
      const int c = k / 3, which = k - c * 3;
      double acc = 0.0;
      for (int64_t v = (int64_t)blockIdx.x * 256 + tid; v < nvox; v += (int64_t)gridDim.x * 256) {
        const int g = (int)gt[(size_t)r * nvox + v];
        if (k == per - 1) {
          if (g < C) acc += (double)logf(p[(size_t)g * nvox + v]);   // torch.log of the float32 probability
        } else if (which == 0) {
          if (g == c) acc += (double)p[(size_t)c * nvox + v];
        } else if (which == 1) {
          if (g == c) acc += 1.0;
        } else {
          acc += (double)p[(size_t)c * nvox + v];
        }
      }
      red[tid] = acc;
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
      }
      if (tid == 0) part[((size_t)blockIdx.x * R + r) * per + k] = red[0];
      __syncthreads();
    }
  }
}

__global__ void soft_metric_final_kernel(const double* __restrict__ part, int nblocks, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * n + i];   // fixed order
  out[i] = s;
}

extern "C" int64_t vx_soft_metric_workspace_bytes(int C, int R) {
  if (C <= 0 || R <= 0) return 0;
  return (int64_t)256 * R * (C * 3 + 1) * (int64_t)sizeof(double);
}

extern "C" int vx_soft_metric_sums(const float* prob, const uint8_t* gt, int C, int R, int64_t nvox, double* sums,
                                   void* workspace, vx_stream_t stream) {
  if (C <= 0 || C > 255 || R <= 0 || nvox <= 0) VX_FAIL(VX_E_SHAPE, "vx_soft_metric_sums: C=%d R=%d nvox=%lld", C, R, (long long)nvox);
  if (!prob || !gt || !sums || !workspace) VX_FAIL(VX_E_NULL, "vx_soft_metric_sums: null pointer");
  hipStream_t s = (hipStream_t)stream;
  int bx = (int)((nvox + 255) / 256);
  if (bx > 256) bx = 256;
  const int n = R * (C * 3 + 1);
  hipLaunchKernelGGL(soft_metric_partial_kernel, dim3(bx), dim3(256), 0, s, prob, gt, C, R, nvox, (double*)workspace);
  hipLaunchKernelGGL(soft_metric_final_kernel, dim3((n + 63) / 64), dim3(64), 0, s, (const double*)workspace, bx, n, sums);
  VX_CHECK_LAUNCH("vx_soft_metric_sums");
  return VX_OK;
}
