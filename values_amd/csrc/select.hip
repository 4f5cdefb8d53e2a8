// K21: exact order statistics of a large float32 array by MSB-first radix select -- the device half of
// np.quantile(all validation uncertainty maps, q) in evaluation/uncertainty_aggregation/find_threshold.py:61-93
// (tens of millions of voxels; numpy sorts them on one core).  Four passes of 8 bits: a histogram of the current
// byte over the elements that match the prefix found so far, then one workgroup walks the 256 bins to the one that
// holds rank k and narrows (prefix, k).  Keys are the usual order-preserving map of IEEE floats to unsigned
// (negative: all bits flipped, non-negative: sign bit set), so any finite input is handled; NaNs are rejected by
// the host wrapper.  Integer histograms with integer atomics: exact and order-independent.  HBM-bound: 4 reads of
// the array per order statistic.
#include "common.h"

struct SelectState {
  unsigned long long k;   // rank still to find within the prefix
  unsigned prefix;        // key bits fixed so far (high bytes)
  unsigned pad;
  unsigned long long hist[256];
};

__device__ __forceinline__ unsigned vx_float_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float vx_key_float(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__global__ void select_init_kernel(SelectState* st, unsigned long long k) {
  const int t = threadIdx.x;
  if (t == 0) { st->k = k; st->prefix = 0; st->pad = 0; }
  if (t < 256) st->hist[t] = 0;
}

__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ x, int64_t n, int pass, SelectState* st) {
  __shared__ unsigned h[256];
  const int tid = threadIdx.x;
  h[tid] = 0;
  __syncthreads();
  const int shift = 24 - 8 * pass;
  const unsigned prefix = st->prefix;
  const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
    const unsigned key = vx_float_key(x[i]);
    if ((key & himask) == prefix) atomicAdd(&h[(key >> shift) & 0xFFu], 1u);
  }
  __syncthreads();
  if (h[tid]) atomicAdd(&st->hist[tid], (unsigned long long)h[tid]);
}

__global__ void select_pick_kernel(SelectState* st, int pass, float* out) {
  if (threadIdx.x != 0) return;
  unsigned long long k = st->k, acc = 0;
  int b = 0;
  for (; b < 256; ++b) {
    const unsigned long long c = st->hist[b];
    if (k < acc + c) break;
    acc += c;
  }
  if (b == 256) b = 255;   // k beyond the population: cannot happen for k < n
  st->k = k - acc;
  st->prefix |= (unsigned)b << (24 - 8 * pass);
  for (int i = 0; i < 256; ++i) st->hist[i] = 0;
  if (pass == 3) *out = vx_key_float(st->prefix);
}

extern "C" int64_t vx_select_workspace_bytes(void) { return (int64_t)sizeof(SelectState); }

// out[0] = the k-th smallest element of x (0-based)
extern "C" int vx_select_kth(const float* x, int64_t n, int64_t k, float* out, void* workspace, vx_stream_t stream) {
  if (n <= 0 || k < 0 || k >= n) VX_FAIL(VX_E_SHAPE, "vx_select_kth: n=%lld k=%lld", (long long)n, (long long)k);
  if (!x || !out || !workspace) VX_FAIL(VX_E_NULL, "vx_select_kth: null pointer");
  hipStream_t s = (hipStream_t)stream;
  SelectState* st = (SelectState*)workspace;
  int bx = (int)((n + 255) / 256);
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, s, st, (unsigned long long)k);
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(select_hist_kernel, dim3(bx), dim3(256), 0, s, x, n, pass, st);
    hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(64), 0, s, st, pass, out);
  }
  VX_CHECK_LAUNCH("vx_select_kth");
  return VX_OK;
}

// count of non-zero bytes (calculate_foreground_quantile_image, find_threshold.py:11-13)
__global__ __launch_bounds__(256) void count_nonzero_u8_kernel(const uint8_t* __restrict__ x, int64_t n, unsigned long long* out) {
  unsigned c = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) c += x[i] != 0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

extern "C" int vx_count_nonzero_u8(const uint8_t* x, int64_t n, uint64_t* out, vx_stream_t stream) {
  if (n < 0) VX_FAIL(VX_E_SHAPE, "vx_count_nonzero_u8: n=%lld", (long long)n);
  if (!out) VX_FAIL(VX_E_NULL, "vx_count_nonzero_u8: null output");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t), s);
  if (e != hipSuccess) VX_FAIL((int)e, "vx_count_nonzero_u8: memset: %s", hipGetErrorString(e));
  if (n == 0) return VX_OK;
  if (!x) VX_FAIL(VX_E_NULL, "vx_count_nonzero_u8: null input");
  int bx = (int)((n + 255) / 256);
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(count_nonzero_u8_kernel, dim3(bx), dim3(256), 0, s, x, n, reinterpret_cast<unsigned long long*>(out));
  VX_CHECK_LAUNCH("vx_count_nonzero_u8");
  return VX_OK;
}
