// Micro-benchmark for round 6, item 1: gfx950's 2:4 structured-sparse matrix instruction v_smfmac_f32_16x16x64_f16 against the dense
// v_mfma_f32_16x16x32_f16 the split-fp16 convolutions run on.   smfmac_f16 [seconds per rate leg]
//
//   part 1  operand layout, found by probing (nothing about it is in the guides): one-hot compressed A x coded B -> which dense K slot a
//           compressed element lands on for each 2-bit index, where the index bits of a lane sit (ABID), the B and D layouts
//   part 2  sparse vs dense on random data: the sparse product against a float64 sum over the decompressed operand and, bit for bit,
//           against two dense instructions over the same decompressed operand
//   part 3  rates on RANDOM data (the board is power-limited under these kernels): operands in registers, and the multiply loops of the
//           full-resolution kernel in its present x-pair form and in the x-quad 2:4 form, B (and A) re-read from LDS as the kernels do;
//           each leg runs for seconds, reports instructions / s / SIMD, the shader clock it held (s_memtime against the event clock) and
//           the USEFUL (voxel, cout, tap, cin) products per second
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ------------------------------------------------------------------ part 1 / 2: single instructions
template <int ABID>
__global__ void one_sparse(const f16x8* a, const f16x16* b, const int* idx, f32x4* d) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a[l], b[l], c, idx[l], 0, ABID);
  d[l] = c;
}
// two dense instructions over K = 0..31 and 32..63 of the decompressed operand (a0 / a1, b0 / b1 in the dense layout)
__global__ void two_dense(const f16x8* a0, const f16x8* a1, const f16x8* b0, const f16x8* b1, f32x4* d, int order) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (order == 0) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[l], b0[l], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[l], b1[l], c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[l], b1[l], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[l], b0[l], c, 0, 0, 0);
  }
  d[l] = c;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd32() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }
static float rndf() { return (float)(int)(rnd32() & 0xFFFF) / 32768.f - 1.f; }

struct Dev {
  _Float16 *a, *b, *a0, *a1, *b0, *b1;
  int* idx;
  float* d;
};
static Dev dev;
static void run_sparse(const _Float16* A, const _Float16* B, const int* idx, int abid, float* D) {
  CK(hipMemcpy(dev.a, A, 64 * 8 * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dev.b, B, 64 * 16 * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dev.idx, idx, 64 * 4, hipMemcpyHostToDevice));
  if (abid == 0) one_sparse<0><<<1, 64>>>((f16x8*)dev.a, (f16x16*)dev.b, dev.idx, (f32x4*)dev.d);
  else if (abid == 1) one_sparse<1><<<1, 64>>>((f16x8*)dev.a, (f16x16*)dev.b, dev.idx, (f32x4*)dev.d);
  else if (abid == 2) one_sparse<2><<<1, 64>>>((f16x8*)dev.a, (f16x16*)dev.b, dev.idx, (f32x4*)dev.d);
  else one_sparse<3><<<1, 64>>>((f16x8*)dev.a, (f16x16*)dev.b, dev.idx, (f32x4*)dev.d);
  CK(hipMemcpy(D, dev.d, 64 * 4 * 4, hipMemcpyDeviceToHost));
}

// maps found by part 1
static int a_row[64];            // compressed A lane -> matrix row
static int a_slot[64][8][4];     // (lane, element, 2-bit index) -> dense K slot, numbered by the B operand: 16 * (B lane >> 4) + B element
static int b_col[64];            // B lane -> column
static int idx_field[4][8];      // (ABID, element) -> bit position of its 2-bit index in the 32-bit index register

static int part1() {
  std::vector<_Float16> A(64 * 8), B(64 * 16);
  std::vector<int> idx(64);
  std::vector<float> D(256);
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 16; ++e) B[l * 16 + e] = (_Float16)(float)(1 + l * 16 + e);   // <= 1024: exact in fp16
  // (a) which index bits belong to which compressed element: lane 0, every element, every 2-bit field of the 32-bit register, ABID 0..3
  printf("part 1a: index fields (lane 0): element e -> for ABID a the field (bit position) that moves its dense slot\n");
  int ok = 1;
  for (int abid = 0; abid < 4; ++abid) {
    for (int e = 0; e < 8; ++e) {
      int found = -1, nfound = 0, base_slot = -1;
      for (int f = -1; f < 16; ++f) {
        for (auto& v : A) v = (_Float16)0.f;
        A[0 * 8 + e] = (_Float16)1.f;
        for (auto& v : idx) v = 0;
        if (f >= 0) idx[0] = (int)(1u << (2 * f));
        run_sparse(A.data(), B.data(), idx.data(), abid, D.data());
        // row of lane 0's element: D(lane n, reg r) non-zero for the 16 lanes of one row
        int slot = -1;
        for (int q = 0; q < 256; ++q)
          if (D[q] != 0.f && (q >> 2) % 16 == 0) { const int code = (int)D[q] - 1; slot = 16 * ((code >> 4) >> 4) + (code & 15); }
        if (f < 0) base_slot = slot;
        else if (slot != base_slot) { found = f; ++nfound; }
      }
      if (abid < 2 || e == 0) printf("  abid %d element %d: base slot %2d, field at bit %2d (%d fields move it)\n", abid, e, base_slot, found * 2, nfound);
      idx_field[abid][e] = found * 2;
      if (nfound != 1 && abid < 2) ok = 0;
    }
  }
  // (b) full map with ABID 0: every lane, element, index
  for (int l = 0; l < 64; ++l) {
    for (int e = 0; e < 8; ++e) {
      for (int p = 0; p < 4; ++p) {
        for (auto& v : A) v = (_Float16)0.f;
        A[l * 8 + e] = (_Float16)1.f;
        for (auto& v : idx) v = 0;
        idx[l] = p << idx_field[0][e];
        run_sparse(A.data(), B.data(), idx.data(), 0, D.data());
        int nz = 0, row = -1, slot = -1, bad = 0;
        for (int q = 0; q < 256; ++q) {
          if (D[q] == 0.f) continue;
          ++nz;
          const int lane = q >> 2, reg = q & 3, n = lane & 15, i = 4 * (lane >> 4) + reg;   // the dense instruction's D layout
          const int code = (int)D[q] - 1, lb = code >> 4, eb = code & 15;
          if (row < 0) row = i;
          if (i != row) bad = 1;
          if ((lb & 15) != n) bad = 1;                      // B lane -> column n = lane & 15
          const int s = 16 * (lb >> 4) + eb;
          if (slot < 0) slot = s;
          if (s != slot) bad = 1;
        }
        if (nz != 16 || bad) { printf("  lane %d element %d index %d: %d non-zeros, inconsistent\n", l, e, p, nz); ok = 0; }
        a_slot[l][e][p] = slot;
        if (e == 0 && p == 0) a_row[l] = row;
      }
    }
    b_col[l] = l & 15;
  }
  printf("part 1b: compressed A lane l = (k-group l >> 4, row l & 15)?  rows of lanes 0, 1, 15, 16, 33, 63: %d %d %d %d %d %d\n", a_row[0], a_row[1],
         a_row[15], a_row[16], a_row[33], a_row[63]);
  for (int l : {0, 16, 32, 48, 5}) {
    printf("  lane %2d (row %2d): element e, index p -> dense slot (= 16 * (B lane >> 4) + B element):", l, a_row[l]);
    for (int e = 0; e < 8; ++e) printf("  e%d:[%d %d %d %d]", e, a_slot[l][e][0], a_slot[l][e][1], a_slot[l][e][2], a_slot[l][e][3]);
    printf("\n");
  }
  // the hypothesis the kernels would be written against
  int hyp = 1;
  for (int l = 0; l < 64; ++l) {
    if (a_row[l] != (l & 15)) hyp = 0;
    for (int e = 0; e < 8; ++e)
      for (int p = 0; p < 4; ++p)
      {
        const int k = 16 * (l >> 4) + 4 * (e >> 1) + p;                       // K as the compressed operand orders it
        const int gb = (k & 31) >> 3, eb = (k & 7) + (k >= 32 ? 8 : 0);       // B: TWO dense 16x16x32 operands side by side
        if (a_slot[l][e][p] != 16 * gb + eb) hyp = 0;
      }
  }
  for (int e = 0; e < 8; ++e)
    if (idx_field[0][e] != 2 * e || idx_field[1][e] != 16 + 2 * e) hyp = 0;
  printf("part 1: layout %s the hypothesis [A lane (g, row): elements 2t, 2t+1 = the two kept values of K = 16 g + 4 t + {index}; index of element e at bits 2e..2e+1 of the "
         "16-bit half ABID & 1 selects; B lane (g, col): elements 0..7 = K 8 g .. 8 g + 7, elements 8..15 = K 32 + 8 g .. (two dense 16x16x32 B operands side by side); D as the dense "
         "16x16 instruction]\n", hyp ? "MATCHES" : "DIFFERS FROM");
  return ok;
}

static void part2() {
  // random compressed operands with valid index pairs (first < second), all four ABID sets
  std::vector<_Float16> A(64 * 8), B(64 * 16), A0(64 * 8), A1(64 * 8), B0(64 * 8), B1(64 * 8);
  std::vector<int> idx(64);
  std::vector<float> D(256), Dd(256), Dd2(256);
  double worst = 0, worst_dense = 0;
  int bit_equal = 0, bit_equal2 = 0, trials = 200, differ_elems = 0;
  for (int t = 0; t < trials; ++t) {
    const int abid = t & 1;
    for (auto& v : A) v = (_Float16)rndf();
    for (auto& v : B) v = (_Float16)rndf();
    static const int pairs[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    std::vector<double> Ad(16 * 64, 0.0), Bd(64 * 16, 0.0);
    std::vector<_Float16> Adh(16 * 64, (_Float16)0.f), Bdh(64 * 16, (_Float16)0.f);
    for (int l = 0; l < 64; ++l) {
      uint32_t w = rnd32();                      // garbage everywhere else: only the selected fields may matter
      uint32_t set = 0, fmask = 0;
      for (int tq = 0; tq < 4; ++tq) {
        const int* pr = pairs[rnd32() % 6];
        set |= (uint32_t)pr[0] << idx_field[abid][2 * tq];
        set |= (uint32_t)pr[1] << idx_field[abid][2 * tq + 1];
        fmask |= (3u << idx_field[abid][2 * tq]) | (3u << idx_field[abid][2 * tq + 1]);
        Ad[a_row[l] * 64 + a_slot[l][2 * tq][pr[0]]] = (double)(float)A[l * 8 + 2 * tq];
        Ad[a_row[l] * 64 + a_slot[l][2 * tq + 1][pr[1]]] = (double)(float)A[l * 8 + 2 * tq + 1];
        Adh[a_row[l] * 64 + a_slot[l][2 * tq][pr[0]]] = A[l * 8 + 2 * tq];
        Adh[a_row[l] * 64 + a_slot[l][2 * tq + 1][pr[1]]] = A[l * 8 + 2 * tq + 1];
      }
      w = (w & ~fmask) | set;
      idx[l] = (int)w;
      for (int e = 0; e < 16; ++e) {
        Bd[(16 * (l >> 4) + e) * 16 + (l & 15)] = (double)(float)B[l * 16 + e];
        Bdh[(16 * (l >> 4) + e) * 16 + (l & 15)] = B[l * 16 + e];
      }
    }
    run_sparse(A.data(), B.data(), idx.data(), abid, D.data());
    // dense operands: lane (g, i) holds K = 8 g .. 8 g + 7 of its 32-wide half
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 8; ++e) {
        const int k = 8 * (l >> 4) + e;
        A0[l * 8 + e] = Adh[(l & 15) * 64 + k];
        A1[l * 8 + e] = Adh[(l & 15) * 64 + 32 + k];
        B0[l * 8 + e] = Bdh[k * 16 + (l & 15)];
        B1[l * 8 + e] = Bdh[(32 + k) * 16 + (l & 15)];
      }
    CK(hipMemcpy(dev.a0, A0.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(dev.a1, A1.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(dev.b0, B0.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(dev.b1, B1.data(), 1024, hipMemcpyHostToDevice));
    two_dense<<<1, 64>>>((f16x8*)dev.a0, (f16x8*)dev.a1, (f16x8*)dev.b0, (f16x8*)dev.b1, (f32x4*)dev.d, 0);
    CK(hipMemcpy(Dd.data(), dev.d, 1024, hipMemcpyDeviceToHost));
    two_dense<<<1, 64>>>((f16x8*)dev.a0, (f16x8*)dev.a1, (f16x8*)dev.b0, (f16x8*)dev.b1, (f32x4*)dev.d, 1);
    CK(hipMemcpy(Dd2.data(), dev.d, 1024, hipMemcpyDeviceToHost));
    int eq = 1, eq2 = 1;
    for (int q = 0; q < 256; ++q) {
      const int lane = q >> 2, reg = q & 3, n = lane & 15, i = 4 * (lane >> 4) + reg;
      double ref = 0;
      for (int k = 0; k < 64; ++k) ref += Ad[i * 64 + k] * Bd[k * 16 + n];
      worst = fmax(worst, fabs((double)D[q] - ref));
      worst_dense = fmax(worst_dense, fabs((double)Dd[q] - ref));
      if (memcmp(&D[q], &Dd[q], 4)) { eq = 0; ++differ_elems; }
      if (memcmp(&D[q], &Dd2[q], 4)) eq2 = 0;
    }
    bit_equal += eq;
    bit_equal2 += eq2;
  }
  printf("part 2: %d random trials (K = 64 dense = 32 kept, |values| < 1): sparse vs float64 max |d| = %.3e (two dense instructions: %.3e); bit-identical to the dense pair in "
         "%d trials (K-halves in the other order: %d); %d of %d elements differ\n", trials, worst, worst_dense, bit_equal, bit_equal2, differ_elems, trials * 256);
}

// ------------------------------------------------------------------ part 3: rates
struct Clk { unsigned long long cyc, real; };
__device__ inline unsigned long long memtime() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
__device__ inline unsigned long long realtime() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

__device__ inline uint32_t mixh(uint32_t h) { h *= 2654435761u; h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; return h; }
// a random fp16 in (-1, 1) with a full mantissa
__device__ inline _Float16 rh(uint32_t h) { return (_Float16)((float)(int)(mixh(h) & 0xFFFF) * (1.f / 32768.f) - 1.f); }

// MODE 0: dense, registers only (4 accumulators)      MODE 1: sparse, registers only
template <int MODE>
__global__ __launch_bounds__(512) void rate_regs(float* out, int iters, Clk* clk) {
  f16x8 a[2];
  f16x16 b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 8; ++e) a[i][e] = rh(threadIdx.x * 64 + i * 32 + e + blockIdx.x * 7919);
#pragma unroll
    for (int e = 0; e < 16; ++e) b[i][e] = rh(threadIdx.x * 64 + i * 32 + 8 + e + blockIdx.x * 104729);
  }
  const int idx = 0x4E4E4E4E ^ ((threadIdx.x & 1) ? 0x00A0 : 0);   // valid pairs (first < second)
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const unsigned long long t0 = memtime(), r0 = realtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MODE == 0) {
          f16x8 bb;
#pragma unroll
          for (int e = 0; e < 8; ++e) bb[e] = b[k & 1][e + 8 * (i & 1)];
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k + i) & 1], bb, acc[i], 0, 0, 0);
        } else {
          acc[i] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a[(k + i) & 1], b[k & 1], acc[i], idx, 0, 0);
        }
      }
  }
  const unsigned long long t1 = memtime(), r1 = realtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk->cyc = t1 - t0; clk->real = r1 - r0; }
}

// The multiply loop of a consumer wave of conv3d_xp8w.hip (one-chunk layer, R = 4 rows of 32 voxels of one z-plane: 128 voxels per item
// and wave, 27 taps x 8 channels -> 8 output channels, split-fp16: three products per fp32 product).
//   MODE 0  present form: x-pair rows (dx, co), step = (kz, ky), K = 4 x-offsets x 8 channels (a quarter of A is structural zeros);
//           per kz: 6 image rows (hi, lo) = 12 ds_read_b128, per (kz, ky): weights (hi, lo) = 2 reads + 12 matrix instructions
//           -> 108 matrix instructions, 54 reads per item
//   MODE 1  x-quad 2:4 form: rows (dx 0..3, co 0..3) x 2 row tiles, column tile = 2 image rows x 8 voxel quads, K block = 4 units
//           (kz, ky, x-offset pair) x 8 channels; blocks (kz in {0, 1}, ky) carry kz = 2's units in their fourth k-group, so the image fragment
//           of a block depends on (kz, input row) only; a seventh block holds (2, ky, pair 2).  Per kz in {0, 1}: 5 fragments x (hi, lo) x 2
//           reads, per block the weights 2 tiles x (hi, lo) = 4 reads + 2 column tiles x 6 instructions; block 7: 2 fragments + weights
//           -> 84 matrix instructions, 76 reads per item (weights from LDS)
//   MODE 2  as 1 with the compressed weights in registers (112 VGPRs): 48 reads per item
//   MODE 3  as 2, every fragment fetched by four ds_read2_b32 instead of two ds_read_b128 (a plain image needs no interleaving)
// Every address pattern is conflict-free (the point is instruction mix and rate, not a particular image layout).
template <int MODE>
__global__ __launch_bounds__(512) void rate_loop(float* out, int iters, Clk* clk, int zero_half) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* s = reinterpret_cast<_Float16*>(smem_raw);
  constexpr int NH = 48 * 1024;                       // halves: 96 KiB of image + weights
  // zero_half: every second element (by hash) is zero, as behind Dropout(p = 0.5) -- the inputs of all three full-resolution layers
  for (int i = threadIdx.x; i < NH; i += blockDim.x) s[i] = (zero_half && (mixh(i * 7 + 3) & 1)) ? (_Float16)0.f : rh(i + blockIdx.x * 7919);
  // x-pair weights: zero where the x-offset lies outside the row's window (lane (m, g): dx = m >> 3, ix = g)
  _Float16* sw = s + NH;                              // 9 steps x (hi, lo) x 64 lanes x 8 halves (dense) / 7 blocks x 2 tiles x (hi, lo) x 64 x 8 (sparse)
  constexpr int WH = MODE == 0 ? 9 * 2 * 64 * 8 : 7 * 2 * 2 * 64 * 8;
  for (int i = threadIdx.x; i < WH; i += blockDim.x) {
    _Float16 v = rh(i * 3 + 1 + blockIdx.x * 31);
    if (MODE == 0) {
      const int l = (i >> 3) & 63, m = l & 15, g = l >> 4, dx = m >> 3;
      if (g - dx < 0 || g - dx > 2) v = (_Float16)0.f;
    }
    sw[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int idx = (lane & 16) ? 0x4E4E4E4E : 0xE4E4E4E4;           // {0, 1} or {2, 3} of every group, by lane
  f32x4 acc[4], accx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { acc[i] = (f32x4){0, 0, 0, 0}; accx[i] = (f32x4){0, 0, 0, 0}; }
  f16x8 wreg[MODE >= 2 ? 28 : 1];
  if constexpr (MODE >= 2) {
#pragma unroll
    for (int i = 0; i < 28; ++i) wreg[i] = *reinterpret_cast<const f16x8*>(sw + (i * 64 + lane) * 8);
  }
  const unsigned long long t0 = memtime(), r0 = realtime();
  for (int it = 0; it < iters; ++it) {
    // the wave's window moves with the item: row fragments at 256-byte multiples
    const _Float16* img = s + (((it * 5 + wave * 3) & 15) * 64 + lane) * 8;      // + fragment f * 1024 halves (2 KiB)
    if constexpr (MODE == 0) {
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        f16x8 bh[6], bl[6];
#pragma unroll
        for (int jr = 0; jr < 6; ++jr) {
          bh[jr] = *reinterpret_cast<const f16x8*>(img + (kz * 12 + jr * 2) * 1024);
          bl[jr] = *reinterpret_cast<const f16x8*>(img + (kz * 12 + jr * 2 + 1) * 1024);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const f16x8 ah = *reinterpret_cast<const f16x8*>(sw + (((kz * 3 + ky) * 2) * 64 + lane) * 8);
          const f16x8 al = *reinterpret_cast<const f16x8*>(sw + (((kz * 3 + ky) * 2 + 1) * 64 + lane) * 8);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[r + ky], acc[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[r + ky], accx[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[r + ky], accx[r], 0, 0, 0);
          }
        }
      }
    } else {
      auto frag = [&](int f, int prec) {               // one image fragment: 16 halves per lane
        f16x16 v;
        if constexpr (MODE == 3) {
          // four ds_read2_b32: dwords (d, d + 4 entries) of a plain image
          const uint32_t* p = reinterpret_cast<const uint32_t*>(s + (f * 2 + prec) * 1024 + (((it * 5 + wave * 3) & 15) * 64) * 8) + lane;
          uint32_t w[8];
#pragma unroll
          for (int d = 0; d < 4; ++d) { w[2 * d] = p[d * 128]; w[2 * d + 1] = p[d * 128 + 64]; }
          __builtin_memcpy(&v, w, 32);
        } else {
          const f16x8 lo8 = *reinterpret_cast<const f16x8*>(img + (f * 2 + prec) * 1024);
          const f16x8 hi8 = *reinterpret_cast<const f16x8*>(img + (f * 2 + prec) * 1024 + 512);
#pragma unroll
          for (int e = 0; e < 8; ++e) { v[e] = lo8[e]; v[8 + e] = hi8[e]; }
        }
        return v;
      };
      auto wfrag = [&](int blk, int tile, int prec) {
        if constexpr (MODE >= 2) return wreg[(blk * 2 + tile) * 2 + prec];
        else return *reinterpret_cast<const f16x8*>(sw + ((((blk * 2 + tile) * 2 + prec) * 64) + lane) * 8);
      };
      auto block = [&](int blk, const f16x16& b0h, const f16x16& b0l, const f16x16& b1h, const f16x16& b1l) {
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
          const f16x8 ah = wfrag(blk, tile, 0), al = wfrag(blk, tile, 1);
          acc[tile] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(ah, b0h, acc[tile], idx, 0, 0);
          accx[tile] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(ah, b0l, accx[tile], idx, 0, 0);
          accx[tile] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(al, b0h, accx[tile], idx, 0, 0);
          acc[2 + tile] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(ah, b1h, acc[2 + tile], idx, 0, 1);
          accx[2 + tile] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(ah, b1l, accx[2 + tile], idx, 0, 1);
          accx[2 + tile] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(al, b1h, accx[2 + tile], idx, 0, 1);
        }
      };
#pragma unroll
      for (int kz = 0; kz < 2; ++kz) {
        f16x16 fh[5], fl[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) { fh[j] = frag(kz * 5 + j, 0); fl[j] = frag(kz * 5 + j, 1); }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) block(kz * 3 + ky, fh[ky], fl[ky], fh[ky + 2], fl[ky + 2]);
      }
      {
        const f16x16 f0h = frag(10, 0), f0l = frag(10, 1), f1h = frag(11, 0), f1l = frag(11, 1);
        block(6, f0h, f0l, f1h, f1l);
      }
    }
  }
  const unsigned long long t1 = memtime(), r1 = realtime();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + accx[i][0] + accx[i][1] + accx[i][2] + accx[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk->cyc = t1 - t0; clk->real = r1 - r0; }
}

template <typename F>
static void leg(const char* name, F launch, double secs, double instr_per_wave, int waves_per_simd, double useful_per_launch, Clk* dclk) {
  launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto t0 = std::chrono::steady_clock::now();
  double el = 0, ms_last = 0;
  int n = 0;
  while (el < secs) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms_last = ms;                                   // the last launch: the board has settled
    ++n;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  Clk c;
  CK(hipMemcpy(&c, dclk, sizeof(c), hipMemcpyDeviceToHost));
  const double t = ms_last * 1e-3;
  const double ips = instr_per_wave * waves_per_simd / t;                  // matrix instructions per second and SIMD
  printf("%-58s %8.3f ms/launch  %6.1f M instr/s/SIMD  s_memtime %.3f GHz (s_memrealtime %.1f MHz) -> %5.2f cycles/instr  useful %7.2f T products/s\n", name, ms_last,
         ips * 1e-6, (double)c.cyc / t * 1e-9, (double)c.real / t * 1e-6, ((double)c.cyc / t) / ips, useful_per_launch / t * 1e-12);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 2.0;
  const int only = argc > 2 ? atoi(argv[2]) : -1;      // one rate leg only (0..7), for tools/power_probe.sh; skips parts 1 / 2
  const int zero_half = argc > 3 ? atoi(argv[3]) : 0;  // the loops' image with half of its elements zero (dropout)
  CK(hipMalloc(&dev.a, 1024)); CK(hipMalloc(&dev.b, 2048)); CK(hipMalloc(&dev.idx, 256)); CK(hipMalloc(&dev.d, 1024));
  CK(hipMalloc(&dev.a0, 1024)); CK(hipMalloc(&dev.a1, 1024)); CK(hipMalloc(&dev.b0, 1024)); CK(hipMalloc(&dev.b1, 1024));
  if (only < 0) {
    if (!part1()) printf("part 1: the probe was NOT consistent -- part 2 uses what it found anyway\n");
    part2();
  }

  float* out;
  Clk* dclk;
  CK(hipMalloc(&out, 1024 * 512 * 4));
  CK(hipMalloc(&dclk, sizeof(Clk)));
  const int iters = 4000;
  const int G = 256;   // one workgroup per CU
  // registers only: 32 matrix instructions per iteration and wave; useful products = all of K (dense K = 32, sparse kept K = 32)
  int legno = 0;
  auto want = [&]() { return only < 0 || only == legno++; };
  if (only >= 0) legno = 0;
  for (int wps : {1, 2}) {
    char nm[96];
    snprintf(nm, 96, "dense 16x16x32 f16, registers, %d wave(s)/SIMD", wps);
    if (want()) leg(nm, [&] { rate_regs<0><<<G, wps * 256>>>(out, iters, dclk); }, secs, 32.0 * iters, wps, (double)G * wps * 4 * iters * 32.0 * 8192, dclk);
    snprintf(nm, 96, "sparse 16x16x64 f16, registers, %d wave(s)/SIMD", wps);
    if (want()) leg(nm, [&] { rate_regs<1><<<G, wps * 256>>>(out, iters, dclk); }, secs, 32.0 * iters, wps, (double)G * wps * 4 * iters * 32.0 * 8192, dclk);
  }
  // the multiply loops: 8 waves per workgroup (two per SIMD), 128 voxels x 8 cout x 27 taps x 8 cin useful products per item and wave and
  // split product (x 3 issued)
  const size_t lds = (48 * 1024 + 7 * 2 * 2 * 64 * 8) * 2;
  const double useful = (double)G * 8 * 1000 * 128.0 * 8 * 27 * 8;
  CK(hipFuncSetAttribute((const void*)rate_loop<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)rate_loop<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)rate_loop<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)rate_loop<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (zero_half) printf("(the loops' image: every second element zero)\n");
  if (want()) leg("x-pair dense loop (108 instr, 54 ds_read_b128 per item)", [&] { rate_loop<0><<<G, 512, lds>>>(out, 1000, dclk, zero_half); }, secs, 108.0 * 1000, 2, useful, dclk);
  if (want()) leg("x-quad 2:4 loop, weights from LDS (84 instr, 76 reads)", [&] { rate_loop<1><<<G, 512, lds>>>(out, 1000, dclk, zero_half); }, secs, 84.0 * 1000, 2, useful, dclk);
  if (want()) leg("x-quad 2:4 loop, weights in registers (84 instr, 48 reads)", [&] { rate_loop<2><<<G, 512, lds>>>(out, 1000, dclk, zero_half); }, secs, 84.0 * 1000, 2, useful, dclk);
  if (want()) leg("x-quad 2:4 loop, registers + ds_read2_b32 fragments (96 reads)", [&] { rate_loop<3><<<G, 512, lds>>>(out, 1000, dclk, zero_half); }, secs, 84.0 * 1000, 2, useful, dclk);
  return 0;
}
