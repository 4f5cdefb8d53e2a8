// K19/K20: map -> scalar aggregations of evaluation/uncertainty_aggregation/aggregate_uncertainties.py.
//   vx_box_max: patch_level_aggregation (:13-31).  The reference box-sums with scipy.signal.convolve
//     (float64 result); here three separable sliding sums in float64, then a deterministic single-block
//     max + "first index with np.isclose(value, max)" (rtol 1e-5, atol 1e-8, C order) search.
//   vx_sum_thr: image_level_aggregation (:34-37) and threshold_aggregation (:61-67) in one pass.
// Maps are small (64^3 .. 256x478), so these are latency-, not bandwidth-bound: few launches, no atomics.
#include "common.h"

// out[o][j'][i] = sum_{k<p} in[o][j'+k][i]; axis length n -> n-p+1, inner stride `inner`
template <typename TIn>
__global__ __launch_bounds__(256) void box_axis_kernel(const TIn* __restrict__ in, double* __restrict__ out, int64_t outer,
                                                       int n, int64_t inner, int p) {
  const int no = n - p + 1;
  const int64_t total = outer * no * inner;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t in_i = i % inner;
    const int64_t j = (i / inner) % no;
    const int64_t o = i / (inner * no);
    const TIn* src = in + (o * n + j) * inner + in_i;
    double s = 0.0;
    for (int k = 0; k < p; ++k) s += (double)src[(int64_t)k * inner];
    out[i] = s;
  }
}

__global__ __launch_bounds__(1024) void max_first_close_kernel(const double* __restrict__ v, int64_t n, double* result,
                                                               int64_t* first) {
  __shared__ double s_max[16];
  __shared__ long long s_idx[16];
  __shared__ double s_gmax;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double m = -INFINITY;
  for (int64_t i = tid; i < n; i += 1024) m = fmax(m, v[i]);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) m = fmax(m, __shfl_xor(m, off, 64));
  if (lane == 0) s_max[wave] = m;
  __syncthreads();
  if (tid == 0) {
    double g = s_max[0];
    for (int w = 1; w < 16; ++w) g = fmax(g, s_max[w]);
    s_gmax = g;
  }
  __syncthreads();
  const double g = s_gmax;
  const double tol = 1e-8 + 1e-5 * fabs(g);  // np.isclose(a, b): |a-b| <= atol + rtol*|b|, b = max
  long long fi = 0x7fffffffffffffffLL;
  for (int64_t i = tid; i < n; i += 1024)
    if (fabs(v[i] - g) <= tol) { fi = (long long)i; break; }  // ascending i per thread: first hit is its minimum
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const long long o = __shfl_xor(fi, off, 64);
    fi = o < fi ? o : fi;
  }
  if (lane == 0) s_idx[wave] = fi;
  __syncthreads();
  if (tid == 0) {
    long long b = s_idx[0];
    for (int w = 1; w < 16; ++w) b = s_idx[w] < b ? s_idx[w] : b;
    result[0] = g;
    first[0] = b;
  }
}

__global__ void unravel_kernel(const int64_t* first, int od, int oh, int ow, int32_t* idx) {
  const int64_t f = first[0];
  idx[0] = (int32_t)(f / ((int64_t)oh * ow));
  idx[1] = (int32_t)((f / ow) % oh);
  idx[2] = (int32_t)(f % ow);
}

extern "C" int vx_box_max(const float* map, int D, int H, int W, int pd, int ph, int pw, double* result, int32_t* idx,
                          void* workspace, size_t workspace_bytes, vx_stream_t stream) {
  if (!map || !result || !idx || !workspace) VX_FAIL(VX_E_NULL, "vx_box_max: null pointer");
  if (D <= 0 || H <= 0 || W <= 0 || pd <= 0 || ph <= 0 || pw <= 0 || pd > D || ph > H || pw > W)
    VX_FAIL(VX_E_SHAPE, "vx_box_max: patch (%d,%d,%d) must fit map (%d,%d,%d)", pd, ph, pw, D, H, W);
  const int64_t n = (int64_t)D * H * W;
  if (workspace_bytes < (size_t)(2 * n + 2) * sizeof(double)) VX_FAIL(VX_E_WORKSPACE, "vx_box_max: workspace needs %lld bytes", (long long)((2 * n + 2) * 8));
  double* a = (double*)workspace;
  double* b = a + n;
  int64_t* first = (int64_t*)(b + n);
  hipStream_t s = (hipStream_t)stream;
  const int ow = W - pw + 1, oh = H - ph + 1, od = D - pd + 1;
  auto nb = [](int64_t t) { int x = (int)((t + 255) / 256); return x > 4096 ? 4096 : (x < 1 ? 1 : x); };
  // along W (inner = 1)
  hipLaunchKernelGGL(box_axis_kernel<float>, dim3(nb((int64_t)D * H * ow)), dim3(256), 0, s, map, a, (int64_t)D * H, W, (int64_t)1, pw);
  // along H (inner = ow)
  hipLaunchKernelGGL(box_axis_kernel<double>, dim3(nb((int64_t)D * oh * ow)), dim3(256), 0, s, a, b, (int64_t)D, H, (int64_t)ow, ph);
  // along D (inner = oh*ow)
  hipLaunchKernelGGL(box_axis_kernel<double>, dim3(nb((int64_t)od * oh * ow)), dim3(256), 0, s, b, a, (int64_t)1, D, (int64_t)oh * ow, pd);
  hipLaunchKernelGGL(max_first_close_kernel, dim3(1), dim3(1024), 0, s, a, (int64_t)od * oh * ow, result, first);
  hipLaunchKernelGGL(unravel_kernel, dim3(1), dim3(1), 0, s, first, od, oh, ow, idx);
  VX_CHECK_LAUNCH("vx_box_max");
  return VX_OK;
}

// The comparison runs in float64 on both sides: the reference compares the stored map (float64 when it was read back
// from NIfTI) with a float64 threshold (aggregate_uncertainties.py:61-66); a float32 threshold could move voxels within
// one float32 ulp of it across the >= boundary.
template <typename T>
__global__ __launch_bounds__(1024) void sum_thr_kernel(const T* __restrict__ v, int64_t n, double thr, double* sums) {
  __shared__ double s_red[3][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0, st = 0.0, ct = 0.0;
  for (int64_t i = tid; i < n; i += 1024) {
    const double x = (double)v[i];
    s += x;
    if (x >= thr) { st += x; ct += 1.0; }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    s += __shfl_xor(s, off, 64);
    st += __shfl_xor(st, off, 64);
    ct += __shfl_xor(ct, off, 64);
  }
  if (lane == 0) { s_red[0][wave] = s; s_red[1][wave] = st; s_red[2][wave] = ct; }
  __syncthreads();
  if (tid < 3) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += s_red[tid][w];
    sums[tid] = t;
  }
}

extern "C" int vx_sum_thr(const void* map, int dtype, int64_t n, double thr, double* sums, vx_stream_t stream) {
  if (!map || !sums) VX_FAIL(VX_E_NULL, "vx_sum_thr: null pointer");
  if (n < 0) VX_FAIL(VX_E_SHAPE, "vx_sum_thr: negative size");
  if (dtype == VX_F32)
    hipLaunchKernelGGL(sum_thr_kernel<float>, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)map, n, thr, sums);
  else if (dtype == VX_F64)
    hipLaunchKernelGGL(sum_thr_kernel<double>, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const double*)map, n, thr, sums);
  else
    VX_FAIL(VX_E_DTYPE, "vx_sum_thr: dtype %d", dtype);
  VX_CHECK_LAUNCH("vx_sum_thr");
  return VX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Colour rendering of an arg-max mask (Tester.save_prediction, test_2D.py:124-134): label -> RGB through a 256-entry
// table, pixels of the ignore map first set to `unlabeled`.  Byte gather, HBM-bound.
__global__ __launch_bounds__(256) void colorize_u8_kernel(const uint8_t* __restrict__ labels, const uint8_t* __restrict__ ignore,
                                                          int64_t n, const uint8_t* __restrict__ lut, int unlabeled,
                                                          uint8_t* __restrict__ rgb) {
  __shared__ uint8_t s_lut[768];
  for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = lut[i];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int l = labels[i];
    if (ignore && ignore[i]) l = unlabeled;
    rgb[3 * i + 0] = s_lut[3 * l + 0];
    rgb[3 * i + 1] = s_lut[3 * l + 1];
    rgb[3 * i + 2] = s_lut[3 * l + 2];
  }
}

extern "C" int vx_colorize_u8(const uint8_t* labels, const uint8_t* ignore, int64_t n, const uint8_t* lut, int unlabeled,
                              uint8_t* rgb, vx_stream_t stream) {
  if (n < 0 || unlabeled < 0 || unlabeled > 255) VX_FAIL(VX_E_SHAPE, "vx_colorize_u8: n=%lld unlabeled=%d", (long long)n, unlabeled);
  if (n == 0) return VX_OK;
  if (!labels || !lut || !rgb) VX_FAIL(VX_E_NULL, "vx_colorize_u8: null pointer");
  int bx = (int)((n + 255) / 256);
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(colorize_u8_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, labels, ignore, n, lut, unlabeled, rgb);
  VX_CHECK_LAUNCH("vx_colorize_u8");
  return VX_OK;
}
