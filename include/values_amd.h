/*
 * values_amd.h -- C ABI of libvalues_amd.so: the MI355X (gfx950) hot path of ValUES'
 * multi-pass segmentation-uncertainty inference.
 *
 * The reference (IML-DKFZ/values) is pure Python on PyTorch and has no FFI of its own;
 * every entry point below names the reference call site whose device work it replaces.
 * The Python mirror of the reference's interface (the values_amd Python package) binds these with
 * ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = VX_E_* argument error, >0 = hipError_t;
 *     vx_last_error_string() describes the last failure on the calling thread.
 *   - no allocation inside: the caller owns every buffer (device pointers) and passes
 *     an explicit workspace; nothing is freed by the library.
 *   - every launch goes to the caller's stream (hipStream_t passed as void*), nothing
 *     synchronises, so all entry points are capturable into a hipGraph.
 *   - activations inside the library are channels-last ("NDHWC") float32 with an explicit
 *     per-voxel channel pitch; tensors crossing the boundary in the reference's NCDHW
 *     layout are marked so.
 */
#ifndef VALUES_AMD_H
#define VALUES_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vx_stream_t; /* hipStream_t */

enum {
  VX_OK = 0,
  VX_E_NULL = -1,      /* required pointer is null */
  VX_E_SHAPE = -2,     /* unsupported / inconsistent shape */
  VX_E_DTYPE = -3,     /* unsupported dtype / layout / mode enum */
  VX_E_WORKSPACE = -4, /* workspace too small */
  VX_E_ALIGN = -5      /* pointer or pitch not 16-byte aligned */
};

enum { VX_F32 = 0, VX_F64 = 1 };
enum { VX_ACT_NONE = 0, VX_ACT_LRELU = 1 /* slope 0.01 */, VX_ACT_RELU = 2 };
/* dropout (torch.nn.Dropout in training mode, p = 0.5, scale 2):
 *   NONE : identity
 *   HASH : counter-based bit generator keyed by (seed, layer id, sample, element)
 *   MASK : caller-supplied keep-mask, uint8 0/1, channels-last, same geometry as the
 *          tensor it is applied to (parity tests inject the reference's masks) */
enum { VX_DROP_NONE = 0, VX_DROP_HASH = 1, VX_DROP_MASK = 2 };

int vx_version(void);
const char* vx_last_error_string(void);
/* diagnostic: the kernel instance (as rocprofv3 names it) the calling thread's last launch dispatched to, or "" */
const char* vx_last_kernel_name(void);

/* The keep-bits of VX_DROP_HASH as an explicit VX_DROP_MASK mask: mask[n][e] (uint8 0/1) for sample n, channels-last
 * element e = voxel * C + c of dropout layer `layer` (index in DROPOUT order: contr_1_1 .. contr_4_2, center,
 * expand_4_1 .. expand_1_2) under `seed`.  A hash-dropout run can be replayed with explicit masks -- the parity tests
 * hand them to the float64 restatement of unet3D_module.py, so the production bit generator's kernels are what is
 * compared. */
int vx_drop_hash_mask(uint32_t seed, uint32_t layer, int N, int64_t elems_per_sample, uint8_t* mask, vx_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Library configuration: which kernel family runs a layer (and with it the PACKED WEIGHT LAYOUT) plus tuning knobs.
 * Read ONCE from the environment on first use (variable VX_<FIELD NAME IN CAPITALS>, e.g. VX_CONV_FP32=1), never per
 * launch; vx_set_config replaces it (not while launches of another thread are being issued).  Weights are bound to
 * the family they were packed for: see `w_family` in the args structs -- a launch whose weights were packed under a
 * different configuration fails with VX_E_DTYPE instead of computing with the wrong layout. */
typedef struct vx_config {
  int32_t conv_fp32;      /* 0: split-fp16 products on the f16 matrix cores (default); 1: native fp32 matrix kernels (the exact
                             fmaf-chain fallback family: what the fp16 range guard re-runs on); 2: native fp32 only for Cout == 8 layers */
  /* data-flow A/B switches: each selects between kernels / launch sequences that compute the same result, and each is run
   * against the float64 oracle by tests/test_gpu_unet3d.py::test_level0_fusion_variants_vs_oracle_32 */
  int32_t s16_no_prenorm;  /* separate InstanceNorm / LeakyReLU / dropout passes instead of normalise-on-load in the consuming conv */
  int32_t s16_no_xp8;      /* the general tile kernels instead of the z-column kernels on the full-resolution Cout == 8 layers */
  int32_t s16_skip_raw;    /* 1 (default): contr_1_2's raw output goes into the skip half, expand_1_1 normalises it on load */
  int32_t no_head_fusion;  /* separate vx_conv1x1_ncdhw launch instead of the 1x1x1 head in expand_1_2's epilogue */
  int32_t s16_no_upfuse;   /* separate upscale2 launch + concat read instead of the up-convolution fused into expand_1_1 */
  int32_t s16_no_poolfuse; /* separate pooling pass over contr_1_2's output instead of the window maxima from its epilogue */
  int32_t storage16;       /* OPT-IN reduced-precision throughput modes (default 0).  1: expand_1_1's full-resolution output is stored
                              as fp16 and expand_1_2 consumes it unsplit (2 instead of 3 matrix products).  2 (round 6): as 1, and
                              the three full-resolution launches run ONE fp16 product per fp32 product (vx_conv3d_args.products):
                              what BASELINE config 2 calls "bf16".  Maps then differ from the float64 reference by ~1e-3:
                              bench.py --storage16 reports the measured differences of both */
  int32_t s16_no_dbplain;  /* plain (not x-pair) single-chunk tile layers whose tile is 16 x 4 x 4: ONE LDS image with two barriers
                              per item instead of two staggered images with one (same tile, same statistics layout, same bits) */
  int32_t s16_generic;     /* the tile kernel's GENERIC instance (run-time epilogue, one LDS image, two barriers per item) wherever a
                              specialised one (compile-time epilogue, double-buffered staggered schedule) would run: the reference
                              of tests/test_gpu_kernels.py::test_conv3d_k3_specialised_instances_equal_generic (bit for bit) */
  int32_t s16_no_upcompose; /* the fused up-convolution evaluated per step by the staging waves (round 2) instead of composed into
                               expand_1_1's weights (round 4, vx_conv3d_args.up_fused) */
  int32_t s16_no_upsplit;  /* expand_2_2 stores plain floats and the fused up-convolution splits them per step (round 2) */
  int32_t s16_no_presplit; /* MC-dropout batches: contr_1_2 normalises the shared first-layer tensor on load for every sample
                              instead of reading the once-per-volume output of vx_prenorm_split */
  int32_t s16_no_poolfin;  /* a separate vx_pool_finish pass over contr_1_2's window maxima instead of contr_2_1 finishing them while it
                              stages its tiles (round 4, vx_conv3d_args.in_pool_flags) */
  int32_t s16_no_zc16;     /* the general tile kernels instead of the role-split z-column kernel (round 5, conv3d_zc16.hip) on the
                              Cout == 16 layers with Cin in {8, 16} and W % 32 == 0; with it the forward also keeps the separate
                              normalise + pool pass of the second contract block */
  int32_t s16_no_halves;   /* expand_2_1 as ONE launch of the tile kernel over the x-blocked concat buffer instead of two launches of the
                              16-channel z-column kernel over its halves (round 5, vx_conv3d_args.acc_in, vx_unet3d_weights.split_w) */
  int32_t s16_no_deep;     /* the general tile kernels instead of the role-split kernel of the deep layers (round 5, conv3d_deep.hip:
                              Cout % 32 == 0, Cin >= 16, volumes of 32^3 voxels and below) */
  int32_t s16_no_l1dma;    /* expand_2_1 -> expand_2_2 as a plain float tensor, staged through registers, instead of the PLANAR pre-split
                              hand-over staged by LDS-DMA (round 6, vx_conv3d_args.out_planar / in_planar) */
  int32_t c2s_no_wide;     /* 2D 3x3 layers of <= 48 input channels: one work item per 16-channel sub-block (round 2) instead of
                              one per tile with all sub-blocks staged together; same bits */
  int32_t c2s_no_oct;      /* 2D 3x3 layers of <= 8 or 17..24 input channels: the sub-block K schedule (5 / 10 steps) instead of
                              the octet-granular one (3 / 7 steps).  Changes the packed layout (vx_conv2d_family) */
} vx_config;
int vx_get_config(vx_config* out);
int vx_set_config(const vx_config* cfg);
/* kernel family (= packed layout) a layer's weights must be packed for under the current configuration; > 0 */
int vx_conv3d_k3_family(int Cin, int Cout);
int vx_conv2d_family(int Cin, int Cout, int KS);

/* ---------------------------------------------------------------------------------
 * K10/K11/K12: fused softmax -> {mean prob, predictive entropy, expected entropy,
 * mutual information, argmax} reduction over T predictions, one pass over the input.
 * Replaces calculate_uncertainty (uncertainty_modeling/test_3D.py:486-518), the mean /
 * argmax of DataCarrier3D.save_data (data_carrier_3D.py:253-255, 281-283) and, with
 * from_logits, the F.softmax of test_3D.py:435,448,472.
 *   x        : [B][T][C][nvox] (planar, the reference's (T,C,*spatial) per volume), f32 or f64
 *   from_logits: 0 = x holds probabilities, 1 = x holds logits (softmax over C fused; C <= 8)
 *   outputs  : per volume b: mean_prob [B][C][nvox] (nullable), pred_entropy / exp_entropy /
 *              mutual_info [B][nvox] f32, argmax [B][nvox] u8 (nullable, first maximal class
 *              like np.argmax), sample_argmax [B][T][nvox] u8 (nullable)
 * NaN products (0 * log 0) are skipped exactly like test_3D.py:493-494, 503-504.
 */
int vx_unc_reduce(const void* x, int dtype, int from_logits, int B, int T, int C, int64_t nvox,
                  float* mean_prob, float* pred_entropy, float* exp_entropy, float* mutual_info,
                  uint8_t* argmax, uint8_t* sample_argmax, vx_stream_t stream);

/* The same pass with its optional extras (all nullable):
 *   variance [B][nvox]: mean over classes of the population variance over the T samples of each class probability
 *            (BASELINE.json north_star's "softmax-variance"; the reference computes none, SURVEY D3: definition is this build's)
 *   in_count [B][nvox]: probabilities are divided by max(count, 1) on load  (sliding-window sums, normalised mode)
 *   out_count[B][nvox]: entropies / MI / mean_prob divided by max(count, 1) on store -- DataCarrier3D.save_data's division
 *            of maps computed on UN-normalised sums (data_carrier_3D.py:323-337); variance by its square */
typedef struct vx_unc_outputs {
  float* mean_prob; float* pred_entropy; float* exp_entropy; float* mutual_info;
  float* variance;
  uint8_t* argmax; uint8_t* sample_argmax;
  const float* in_count; const float* out_count;
} vx_unc_outputs;
int vx_unc_reduce_ex(const void* x, int dtype, int from_logits, int B, int T, int C, int64_t nvox,
                     const vx_unc_outputs* outputs, vx_stream_t stream);

/* The same reduction split for member-/sample-sharded ensembles (BASELINE config C3): every rank
 * ADDS the sufficient statistics of its own passes into stats [B][C+1][nvox] (planes 0..C-1: sum_t p_tc, plane C:
 * sum_t sum_c p_tc log p_tc; zero-initialised by the caller), one sum-reduce over RCCL combines the ranks, and
 * finalize turns the sums of T_total passes into the maps of calculate_uncertainty (test_3D.py:486-518). */
int vx_unc_stats_accumulate(const float* logits, int B, int T, int C, int64_t nvox, float* stats, vx_stream_t stream);
int vx_unc_stats_finalize(const float* stats, int B, int T_total, int C, int64_t nvox, float* mean_prob,
                          float* pred_entropy, float* exp_entropy, float* mutual_info, uint8_t* argmax,
                          vx_stream_t stream);

/* F.softmax(dim=1) of planar logits [R][C][nvox] (test_2D.py:302, 315), for class counts above the fused path's 8. */
int vx_softmax_planar(const float* logits, int64_t R, int C, int64_t nvox, float* out, vx_stream_t stream);

/* calculate_one_minus_msr (test_3D.py:521-525) / ExperimentDataloader.get_max_softmax_pred
 * (evaluation/experiment_dataloader.py:38-49): out[v] = 1 - max_c x[c][v]; x [C][nvox]. */
int vx_one_minus_msr(const void* x, int dtype, int C, int64_t nvox, void* out, vx_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Weight packing (one-off, at checkpoint load: load_models_from_checkpoint, test_3D.py:222-247).
 *   conv3d  : torch (Cout, Cin, 3,3,3) f32 -> MFMA fragment order; returns floats needed via *_size
 *   convT   : torch (Cin, Cout, 2,2,2) f32 -> [dz][dy][ci][dx][co]
 */
int64_t vx_conv3d_k3_packed_floats(int Cin, int Cout);
int vx_pack_conv3d_k3(const float* w_torch, float* w_packed, int Cin, int Cout, vx_stream_t stream);
int64_t vx_convT_k2s2_packed_floats(int Cin, int Cout);
int vx_pack_convT_k2s2(const float* w_torch, float* w_packed, int Cin, int Cout, vx_stream_t stream);

/* ---------------------------------------------------------------------------------
 * K1 (+K3,K4 epilogue, K2 statistics): 3x3x3 convolution, padding 1, channels-last.
 * Replaces nn.Conv3d(k=3,p=1) of the contract / expand / center blocks
 * (models/unet3D_module.py:233, 264, 99-111) with its trailing LeakyReLU / ReLU / Dropout
 * fused when the block has no norm, and with the InstanceNorm statistics (per sample and
 * channel sum / sum of squares per workgroup tile, deterministic, no atomics) emitted
 * for vx_instnorm_finalize when it has.
 */
typedef struct vx_conv3d_args {
  const float* in;      /* in_xblk == 0: [N][D][H][W][in_pitch], channels [0, Cin) used; Cin % 8 == 0
                         * in_xblk  > 0: x-blocked concat buffer [N][D][H][W/xb][2][xb][Cin/2] (see vx_concat layout) */
  const float* w_packed;
  const float* bias;    /* [Cout] */
  float* out;           /* [N][D][H][W][out_pitch], written at channel offset out_coff */
  int32_t in_pitch, out_pitch, out_coff;
  int32_t N, D, H, W, Cin, Cout; /* Cout % 8 == 0 */
  int32_t act;          /* VX_ACT_* applied after bias */
  int32_t drop_mode;    /* VX_DROP_* applied after act */
  uint32_t drop_seed, drop_layer;
  const uint8_t* drop_mask; /* [N][D][H][W][Cout] when VX_DROP_MASK */
  float* stats_partial; /* nullable: [N][ntiles][Cout][2] (sum, sumsq of out before act) */
  int32_t in_xblk;      /* 0 = plain input; 1, 2 or 4 = x-block size of a concat input */
  int32_t w_family;     /* vx_conv3d_k3_family(Cin, Cout) at the time w_packed was packed */
  /* Optional fused head (only where vx_conv3d_k3_head_fusable(Cin, Cout)): the 1x1x1 conv of
   * vx_conv1x1_ncdhw applied to this layer's output (after act / dropout) in the epilogue, same
   * arguments and the same bits; `out` may then be NULL and the feature map is never stored. */
  float* head_out;      /* nullable: [slots][head_C][D][H][W] */
  const float* head_w;  /* torch (head_C, Cout, 1,1,1) */
  const float* head_b;  /* [head_C] */
  const int32_t* head_dst;  /* nullable: slot of sample n (default n) */
  const int32_t* head_flip; /* nullable: un-flip code of sample n (bit 0 z, 1 y, 2 x) */
  int32_t head_C;       /* 1 .. 8 */
  /* Optional PROLOGUE (only where vx_conv3d_k3_prologue_ok(D, H, W, Cin, Cout)): the input -- or, for a concat input,
   * its skip half -- is the RAW output of a contract block's conv (unet3D_module.py:231-237); InstanceNorm with the
   * given statistics, LeakyReLU and the block's dropout are applied while the tile is staged, so the normalised tensor
   * is never written.  in_repeat > 1: sample n reads input sample / statistics row n / in_repeat (the T MC-dropout
   * samples of a volume share the first block's conv output; the dropout bits are those of sample n). */
  const float* in_mean; const float* in_rstd;   /* nullable together: [N / in_repeat][8] */
  int32_t in_drop_mode; uint32_t in_drop_seed, in_drop_layer;   /* VX_DROP_NONE | VX_DROP_HASH of the producing block */
  int32_t in_repeat;    /* 0 or 1: sample n reads input sample n */
  int32_t out_xblk, out_half; /* out_xblk > 0: `out` is a concat buffer, this conv writes its half out_half (as vx_norm_args) */
  uint32_t* range_flag; /* nullable device word: atomic max of the bit patterns of the |stored values| that reach 32768 (fp16
                           range guard of the split-fp16 consumers: anything >= 65504 must not reach them; smaller
                           magnitudes are not reported -- the word stays 0 for an ordinary tensor) */
  const uint32_t* seed_dev; /* nullable device word ADDED to drop_seed / in_drop_seed at run time (a captured hipGraph
                               replays with the arguments it was captured with; fresh dropout bits per replay = update
                               this word).  Same field in vx_norm_args / vx_convT_args / vx_unet3d_run. */
  /* Optional fused UP-CONVOLUTION (only where vx_conv3d_k3_upfuse_ok(D, H, W, Cin, Cout)): the decoder's
   * upscale -> torch.cat([up, skip], 1) -> expand (unet3D_module.py:332-356) in one kernel.  Channels [0, 8) of the
   * conv's input are ConvTranspose3d(16 -> 8, k = 2, s = 2)(up_in) + up_b, computed while the tiles are staged and never
   * stored; `in` then holds only the skip half: the concat buffer as before (in_xblk > 0, its up half is not read) or a
   * plain [N][D][H][W][in_pitch] tensor whose channels [0, 8) are the skip (in_xblk == 0).  The prologue fields apply to
   * the skip half.  range_flag also covers the up values. */
  const float* up_in;   /* nullable: [N][D/2][H/2][W/2][up_pitch], channels [0, 16) used */
  const float* up_w;    /* vx_pack_convT_k2s2(Cin = 16, Cout = 8) */
  const float* up_b;    /* [8] */
  int32_t up_pitch;
  /* Optional POOLED OUTPUT of a contract block's second conv (only where vx_conv3d_k3_poolfuse_ok(D, H, W, Cin, Cout);
   * needs stats_partial): besides its raw output and statistics the conv leaves what the block's
   * InstanceNorm -> LeakyReLU -> Dropout -> MaxPool3d(2, 2) (unet3D_module.py:231-237, 303-310) needs of every 2 x 2 x 2
   * window: the maximum over the raw values the dropout KEEPS (drop_mode / drop_seed / drop_layer describe that dropout;
   * -inf when none is kept) and an any-dropped bit per channel.  vx_pool_finish turns them into the pooled tensor once the
   * statistics exist -- the full-resolution tensor is not read again. */
  float* pool_out;      /* nullable: [N][D/2][H/2][W/2][8] raw window maxima (layout 2 of vx_conv3d_k3_pool_layout: [N][D][H/2][W/2][16]) */
  uint32_t* pool_flags; /* [N][D/2][H/2][W/2][2]: bit j of word q = a dropped element of channel 4 q + j in the window (layout 2: [..][4]) */
  /* PRE-SPLIT input (only with in_repeat > 1 where vx_conv3d_k3_prologue_ok): `in` is the output of vx_prenorm_split --
   * the shared raw tensor already normalised, activated and split into fp16 (hi, lo) pairs, ONCE per volume; the prologue
   * then only applies sample n's dropout bits (in_drop_*) while the tile is staged.  in_mean / in_rstd are not read. */
  int32_t in_split;
  /* Reduced-storage mode (vx_config.storage16; opt-in, NOT the default: an activation rounded to fp16 cannot meet the 1e-4
   * parity of the maps): out_f16 -- the LeakyReLU + dropout epilogue stores fp16 (out_pitch counts halves); in_f16 -- the
   * dense 8-channel input is such a tensor: it is copied into the hi plane unsplit and the lo-activation product is skipped. */
  int32_t out_f16, in_f16;
  /* PRE-SPLIT hand-over of the coarse tensor of the fused up-convolution: out_split -- the (tile-kernel) epilogue stores
   * every 16-byte piece as [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] fp16 (the layout of vx_prenorm_split) instead of four floats;
   * up_split -- up_in is such a tensor: the staging waves take the matrix operands as they are (each coarse voxel was split
   * by four waves per step before).  Same values, same bits. */
  int32_t out_split, up_split;
  /* COMPOSED up-convolution (round 4; with up_in, optional): the output of vx_pack_conv3d_upfused for this layer's weights and
   * the transposed conv's.  ConvTranspose3d(k = 2, s = 2) has no activation behind it (unet3D_module.py:157-190, 332-356), so
   * conv(cat([up, skip])) = conv_skip(skip) + (W_up-half o U)(coarse) + bias terms: per output parity class (z & 1, y & 1)
   * the up half of the 3x3x3 conv composed with the transposed conv is a 2 x 2 x 3 (x-pair) tap convolution over the 16
   * COARSE channels -- K = 192 instead of 288 per x-pair row, and the staging waves copy the coarse tensor into LDS instead
   * of evaluating the transposed conv per step.  The up bias enters through a table of the 27 border classes (a 3x3x3 tap
   * outside the volume sees the zero padding of the concatenated tensor, not the bias).  Same function of the inputs;
   * weights composed in float64 at pack time, so results differ from the two-stage evaluation by float32 rounding only. */
  const float* up_fused;
  /* POOL-FINISH on load (round 4; only where vx_conv3d_k3_poolfin_ok(Cin, Cout)): `in` is the pool_out of the previous block's
   * second conv -- window maxima of RAW values [N][D][H][W][8] -- and in_pool_flags its any-dropped words [N][D][H][W][2];
   * in_mean / in_rstd are that block's statistics ([N][8]) and in_drop_mode says whether the dropout's factor 2 applies.  The conv
   * evaluates vx_pool_finish's arithmetic (same expressions, same order) while it stages its tiles: the pooled tensor is never
   * written.  in_repeat / in_split / in_drop_seed are not used.  Only with the plain epilogue of a contract block's first conv
   * (act NONE, no dropout, no head; Cout % 32 != 0): anything else is refused with VX_E_SHAPE. */
  const uint32_t* in_pool_flags;
  /* PARTIAL SUMS (round 5; the 16-channel z-column kernel with an activation epilogue, vx_conv3d_k3_acc_ok): acc_in
   * [N][D][H][W][acc_pitch] (channels [0, Cout)) is added to the conv's result BEFORE the activation / dropout, and `bias` is
   * NOT added (it is part of the partial sums).  A conv over a channel concatenation = the sum of the convs over its parts: the
   * decoder's first conv of level 1 runs as conv(skip half) + bias -> partial, then conv(up half) + partial -> activation
   * (unet3D_module.py:332-356 without the concatenated tensor).  acc_in may alias `out` (every lane reads the pieces it writes). */
  const float* acc_in;
  int32_t acc_pitch;
  /* PLANAR PRE-SPLIT hand-over between two 16-channel layers of the z-column kernel (round 6; only where
   * vx_conv3d_k3_planar_ok(D, H, W, Cin, Cout): the decoder's expand_2_1 -> expand_2_2, unet3D_module.py:263-267, with no
   * normalisation between them).  out_planar: the activation epilogue stores its 16 channels as fp16 (hi, lo) pairs -- the
   * split the consumer's matrix instructions take -- in the consumer's LDS row order:
   *     [N][D][H][octet 2][hi | lo][W][8 halves]          (64 W bytes per row, as the float tensor; out_pitch == 16, out_coff == 0)
   * in_planar: `in` is such a tensor: the staging waves move it into the LDS image by buffer_load ... lds (one instruction per
   * 64 positions, the zero padding as out-of-range lanes) -- no registers, no conversion, no ds_write.  Same values, same
   * products, same bits as the float hand-over.  `out` must not alias acc_in (another layout). */
  int32_t out_planar, in_planar;
  /* Opt-in throughput mode (vx_config.storage16 = 2; never the default, outside the 1e-4 parity bar): products = 1 -- ONE fp16
   * product per fp32 product (activations and weights rounded to fp16, fp32 accumulation) instead of the three of the split
   * scheme.  Exists for the three full-resolution launches of the MC-dropout forward (contr_1_2 on the pre-split tensor,
   * upscale2 composed into expand_1_1, expand_1_2 + head on the fp16 tensor); anything else is refused.  0 = the default. */
  int32_t products;
} vx_conv3d_args;
int64_t vx_conv3d_upfused_packed_floats(void);
/* w1_torch (8, 16, 3,3,3) + b1 (8): the decoder conv whose input channels [0, 8) are the up half; up_w_torch (16, 8, 2,2,2) +
 * up_b (8): the transposed conv (torch layouts, device pointers) -> packed [4 classes][6 K-steps][hi | lo][64 lanes][8 halves]
 * + bias table [27][8] (vx_conv3d_upfused_packed_floats() floats, 16-byte aligned).  The composition runs on the device in
 * float64; a composed weight beyond the fp16 range has its hi part clamped to +-65504 like vx_pack_conv3d_k3's, i.e. the
 * consuming conv then produces inf / NaN loudly.  The B operands of the composed products are the COARSE tensor's values:
 * its producer's range_flag is what guards them (the up values themselves are never formed). */
int vx_pack_conv3d_upfused(const float* w1_torch, const float* b1, const float* up_w_torch, const float* up_b, float* packed,
                           vx_stream_t stream);
int vx_conv3d_k3_prologue_ok(int D, int H, int W, int Cin, int Cout); /* 1 if vx_conv3d_k3 takes in_mean for this layer */
int vx_conv3d_k3_upfuse_ok(int D, int H, int W, int Cin, int Cout);   /* != 0 if vx_conv3d_k3 takes up_in for this layer: 1 the form
   above (16 -> 8 at full resolution); 2 (round 5, the 16-channel z-column kernel): ALL 16 input channels are
   ConvTranspose3d(32 -> 16, k = 2, s = 2)(up_in) + up_b, evaluated while the tiles are staged -- `in` is not read, up_in is
   [N][D/2][H/2][W/2][up_pitch >= 32], up_w the output of vx_pack_convT_zc16, up_split as above; combines with acc_in */
int64_t vx_convT_zc16_packed_floats(void);
int vx_pack_convT_zc16(const float* w_torch /* (32, 16, 2,2,2) */, float* packed, vx_stream_t stream);
/* 1 if vx_conv3d_k3 takes in_mean for the SKIP half of an x-blocked concat input (in_xblk = xblk) of this layer: the decoder's
 * first conv of a level normalising the contract block's raw output on load (round 5: also the tile kernel, Cin % 32 == 0) */
int vx_conv3d_k3_skip_prologue_ok(int D, int H, int W, int Cin, int Cout, int xblk);
int vx_conv3d_k3_acc_ok(int D, int H, int W, int Cin, int Cout);      /* 1 if vx_conv3d_k3 takes acc_in for this layer */
int vx_conv3d_k3_planar_ok(int D, int H, int W, int Cin, int Cout);   /* 1 if vx_conv3d_k3 takes in_planar / out_planar for this layer */
int vx_conv3d_k3_poolfuse_ok(int D, int H, int W, int Cin, int Cout); /* 1 if vx_conv3d_k3 takes pool_out for this layer */
int vx_conv3d_k3_presplit_ok(int D, int H, int W, int Cin, int Cout); /* 1 if vx_conv3d_k3 takes in_split (vx_prenorm_split's output) for this layer */
int vx_conv3d_k3_poolfin_ok(int Cin, int Cout);                       /* 1 if vx_conv3d_k3 takes in_pool_flags for this layer */
/* pooled[c] = any_dropped ? max(s f(m), 0) : s f(m) with f(m) = LeakyReLU((m - mean[n][c]) * rstd[n][c]), s = 2 with
 * dropout (drop_scale2 != 0) else 1: the MaxPool3d(2, 2) of Dropout(LeakyReLU(InstanceNorm(x))) from the window maxima and
 * flags vx_conv3d_k3 left in pool_out / pool_flags (bit-identical to pooling the normalised tensor: f is monotone). */
/* In place: x [N][nvox][8] fp32 -> per 16-byte piece (4 channels) [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] fp16 with
 * y = scale * LeakyReLU((x - mean[n][c]) * rstd[n][c]) = hi + lo * 2^-11 (the split the fp16 matrix kernels consume; scale = 2
 * folds a following p = 0.5 dropout's factor).  For the MC-dropout batch of test_3D.py:462-472: the T samples of a volume
 * share the first block's conv output (InstanceNorm + LeakyReLU of unet3D_module.py:231-237 are the same for all of them,
 * only the dropout bits differ), so this runs once per VOLUME and contr_1_2 (vx_conv3d_args.in_split) masks per sample. */
int vx_prenorm_split(float* x, const float* mean, const float* rstd, int N, int64_t nvox, float scale, vx_stream_t stream);
int vx_pool_finish(const float* pool_raw, const uint32_t* pool_flags, const float* mean, const float* rstd, float* out,
                   int out_pitch, int N, int64_t voxels_per_sample, int drop_scale2, vx_stream_t stream);
/* The same for the 16-channel z-column kernel (round 5; vx_conv3d_k3_pool_layout(...) == 2): its epilogue pools the (y, x) half of
 * every 2 x 2 x 2 window -- pool_raw [N][2 Dp][Hp Wp][16] window maxima of one z-plane each, pool_flags [N][2 Dp][Hp Wp][4] (bit j
 * of word q: a dropped element of channel 4 q + j) -- and this pass takes the maximum / the OR over the z pair before the same
 * arithmetic: out [N][Dp][Hp Wp][out_pitch >= 16].  Bit-identical to pooling the normalised tensor. */
int vx_pool_finish_z(const float* pool_raw, const uint32_t* pool_flags, const float* mean, const float* rstd, float* out,
                     int out_pitch, int N, int Dp, int64_t plane_voxels, int drop_scale2, vx_stream_t stream);
/* layout of pool_out / pool_flags for a layer that takes them (vx_conv3d_k3_poolfuse_ok): 1 = [N][D/2][H/2][W/2][8] + [..][2]
 * (the 8-channel z-column kernel: whole windows), 2 = [N][D][H/2][W/2][16] + [..][4] (the 16-channel one: the z pair is left to
 * vx_pool_finish_z), 0 = no pooled output */
int vx_conv3d_k3_pool_layout(int D, int H, int W, int Cin, int Cout);
/* The decoder's concat buffer (torch.cat([up, skip], 1), unet3D_module.py:332-356) is never materialised as an
 * interleaved tensor: CAT[N][D][H][W/xb][2][xb][C] keeps the up half (s = 0, written by vx_convT_k2s2) and the
 * skip half (s = 1, written by vx_norm_act_drop_pool) as alternating DENSE blocks of xb voxels, so both producers
 * write whole cache lines and the consumer conv reads it through in_xblk.  xb = largest of {4,2,1} dividing W. */
int vx_conv3d_k3_tiles(int D, int H, int W);  /* upper bound of ntiles per sample (stats_partial sizing) */
int vx_conv3d_k3_tiles_for(int D, int H, int W, int Cout); /* exact ntiles for a given Cout (finalize) */
int vx_conv3d_k3(const vx_conv3d_args* a, vx_stream_t stream);
int vx_conv3d_k3_head_fusable(int Cin, int Cout); /* 1 if the layer's kernel can take head_out */

/* First layer, Cin == 1 (contr_1_1): input is the reference's (V,1,D,H,W) volume batch.
 * Sample n reads volume src[n] (nullable: n / repeat) with flip code flip[n] (nullable: 0;
 * bit0 = flip D, bit1 = flip H, bit2 = flip W -- torch.flip dims 2,3,4 of test_3D.py:430,445). */
int vx_conv3d_k3_c1_tiles(int D, int H, int W); /* ntiles per sample of this kernel's stats_partial */
int vx_conv3d_k3_c1(const float* in, const float* w_torch /* (Cout,1,3,3,3) */, const float* bias, float* out,
                    int out_pitch, int N, int D, int H, int W, int Cout, int repeat, const int32_t* src,
                    const int32_t* flip, float* stats_partial, vx_stream_t stream);

/* in_channels > 1 (unet3D_module.py:8-35): (V, Cin, D, H, W) -> channels-last [N][D][H][W][8] (channels Cin..7 zero) for
 * the general 3x3x3 kernels, with the per-sample source volume / TTA flip of vx_conv3d_k3_c1.  1 <= Cin <= 8. */
int vx_pack_input_cl8(const float* in, float* out, int N, int Cin, int D, int H, int W, int repeat, const int32_t* src,
                      const int32_t* flip, vx_stream_t stream);

/* K2: reduce stats_partial -> mean[N][C], rstd[N][C] (biased variance, eps 1e-5:
 * nn.InstanceNorm3d defaults, unet3D_module.py:234). */
int vx_instnorm_finalize(const float* stats_partial, int N, int ntiles, int C, int64_t nvox, float eps,
                         float* mean, float* rstd, vx_stream_t stream);

/* K2+K3+K4(+K5,K7): y = Dropout(LeakyReLU((x - mean) * rstd)) written to `out` (any pitch /
 * channel offset: the skip half of the decoder's concat buffer, unet3D_module.py:332-356)
 * and, if pool_out != NULL, MaxPool3d(2,2) of y (unet3D_module.py:50, 314-323). */
typedef struct vx_norm_args {
  const float* x; int32_t x_pitch;
  const float* mean; const float* rstd; /* nullable both: no normalisation */
  float* out; int32_t out_pitch, out_coff;
  float* pool_out; int32_t pool_pitch;
  int32_t N, D, H, W, C;
  int32_t act, drop_mode; uint32_t drop_seed, drop_layer; const uint8_t* drop_mask;
  int32_t out_xblk, out_half; /* out_xblk > 0: `out` is a concat buffer (xb = out_xblk), this kernel writes half
                                 out_half (0 = up, 1 = skip); out_pitch / out_coff are then ignored */
  int32_t x_xblk, x_half;     /* x_xblk > 0: `x` is half x_half of a concat buffer (a conv that wrote its raw output
                                 straight into the skip half); x_pitch is then ignored.  With pool_out set, `out` may be
                                 NULL: only the pooled tensor is produced (the consumer conv normalises the skip half
                                 itself, vx_conv3d_args.in_mean) */
  const uint32_t* seed_dev;   /* as in vx_conv3d_args */
  uint32_t* range_flag;       /* nullable, as in vx_conv3d_args: set where this pass does NOT normalise (do_instancenorm=False:
                                 the first layer's activation / dropout pass feeds a split-fp16 conv un-normalised) */
} vx_norm_args;
int vx_norm_act_drop_pool(const vx_norm_args* a, vx_stream_t stream);
/* Round 5: a streaming pass that takes its InstanceNorm statistics from the producing conv's PARTIALS instead of from a
 * vx_instnorm_finalize launch in front of it: every workgroup reduces the partials of its sample itself (float64, the formula of
 * vx_instnorm_finalize; the summation order differs, so mean / rstd agree with it to the last bit except on a rounding
 * boundary) and mean_out / rstd_out [N][C] (nullable) receive them for later readers.  stats_partial [N][tiles][C][2] as
 * vx_conv3d_args.stats_partial, count = voxels per sample the sums run over, C <= 512. */
typedef struct vx_stat_src {
  const float* stats_partial; int32_t tiles; float eps; int64_t count;
  float* mean_out; float* rstd_out;
} vx_stat_src;
/* vx_norm_act_drop_pool with a->mean == a->rstd == NULL and the statistics from `st` (x_repeat 1) */
int vx_norm_act_drop_pool_stats(const vx_norm_args* a, const vx_stat_src* st, vx_stream_t stream);
/* vx_prenorm_split / vx_pool_finish_z (above) with the statistics from `st` (C = 8 / 16) */
int vx_prenorm_split_stats(float* x, const vx_stat_src* st, int N, int64_t nvox, float scale, vx_stream_t stream);
/* Zero `bytes` bytes at a device pointer, in stream order (hipMemsetAsync).  The host mirrors allocate with torch.empty and fill
 * through this entry: the persistent zero-padded activation tensors of the 2D walk (channels [C, round4(C)) of a padded layer,
 * hrnet_module.py:37-41 with widths that are no multiple of 4) and the padded BatchNorm scale / shift rows are written ONCE per
 * geometry -- no ATen fill kernel runs on the product's paths (round-5 verdict, weak #13). */
int vx_zero(void* p, int64_t bytes, vx_stream_t stream);
int vx_pool_finish_z_stats(const float* pool_raw, const uint32_t* pool_flags, const vx_stat_src* st, float* out, int out_pitch,
                           int N, int Dp, int64_t plane_voxels, int drop_scale2, vx_stream_t stream);
/* Same, but sample n of the OUTPUT reads sample n / x_repeat of x / mean / rstd: the T MC-dropout samples of a
 * volume share contr_1_1's conv output and statistics (test_3D.py:462-472 feeds the same input T times; dropout
 * is the first thing that differs), so that conv runs once per volume and this kernel fans it out. */
int vx_norm_act_drop_pool_bcast(const vx_norm_args* a, int x_repeat, vx_stream_t stream);

/* K6: ConvTranspose3d(k=2, s=2) (+ReLU+Dropout for center.4), unet3D_module.py:113-120, 157-190;
 * writes channels [out_coff, out_coff+Cout) of the concat buffer (K7: torch.cat disappears).
 * Cin in {64, 128} with Cout % 8 == 0 (% 4 for 128) under vx_config.conv_fp32 == 0 (round 5): the products run on the fp16 matrix
 * cores by operand splitting, like the 3x3x3 convolutions -- same accuracy, and the same contract on the INPUT: |x| < 65504
 * (the producers inside vx_unet3d_forward record it in their range_flag).  A WEIGHT past 65504 is reported through this launch's
 * range_flag (infinity).  conv_fp32 != 0 keeps the native-fp32 instruction. */
typedef struct vx_convT_args {
  const float* in; int32_t in_pitch;
  const float* w_packed; const float* bias;
  float* out; int32_t out_pitch, out_coff;
  int32_t N, D, H, W, Cin, Cout; /* input dims; output is 2D x 2H x 2W */
  int32_t act, drop_mode; uint32_t drop_seed, drop_layer; const uint8_t* drop_mask; /* mask [N][2D][2H][2W][Cout] */
  int32_t out_xblk, out_half; /* as in vx_norm_args: write half out_half of a concat buffer */
  uint32_t* range_flag;       /* nullable: as in vx_conv3d_args (matrix-core kernels only: Cin in {16,32,64,128}) */
  const uint32_t* seed_dev;   /* as in vx_conv3d_args */
} vx_convT_args;
int vx_convT_k2s2(const vx_convT_args* a, vx_stream_t stream);

/* K8 (+K13): final 1x1x1 conv (unet3D_module.py:199, 365) -> logits in the reference's NCDHW
 * layout, sample n written to slot dst[n] (nullable: n) of out [slots][C][D][H][W], un-flipped by
 * flip[n] (test_3D.py:445-447). w: torch (C, F, 1,1,1). */
int vx_conv1x1_ncdhw(const float* in, int in_pitch, const float* w, const float* bias, float* out, int N, int D,
                     int H, int W, int F, int C, const int32_t* dst, const int32_t* flip, vx_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Whole-network launch: UNet3D.forward (unet3D_module.py:296-373) for N samples.
 */
typedef struct vx_unet3d_weights {
  /* packed 3x3x3 convs in execution order:
   * contr_1_1(torch layout, Cin==1) contr_1_2 contr_2_1 contr_2_2 contr_3_1 contr_3_2 contr_4_1 contr_4_2
   * center.0 center.2 expand_4_1 expand_4_2 expand_3_1 expand_3_2 expand_2_1 expand_2_2 expand_1_1 expand_1_2 */
  const float* conv_w[18];
  const float* conv_b[18];
  /* packed transposed convs: center.4 upscale4 upscale3 upscale2 */
  const float* up_w[4];
  const float* up_b[4];
  const float* final_w; /* torch (C, F) */
  const float* final_b;
  int32_t F;            /* initial_filter_size */
  int32_t num_classes;
  int32_t conv_family[18]; /* vx_conv3d_k3_family of each packed conv at pack time (entry 0, the Cin == 1 layer: 0) */
  int32_t in_channels;     /* 0 or 1: conv_w[0] is the torch layout of a Cin == 1 layer (vx_conv3d_k3_c1); 2 .. 8: conv_w[0]
                              is PACKED for Cin = 8 (weights zero-padded), the input goes through vx_pack_input_cl8 */
  int32_t no_instancenorm; /* 1: do_instancenorm=False -- contract blocks are conv + LeakyReLU + Dropout (unet3D_module.py:238-243) */
  const float* up_fused;   /* nullable: vx_pack_conv3d_upfused(expand_1_1, upscale2) -- the level-0 up-convolution composed into
                              expand_1_1 (vx_conv3d_args.up_fused); NULL: evaluated per step by the staging waves (round 2) */
  /* (round 5) nullable: expand_2_1's weights as TWO packed 16 -> 16 convs, [0] = input channels [0, 16) (the up half), [1] =
   * channels [16, 32) (the skip half), each vx_pack_conv3d_k3(16, 16) of the contiguous slice: the layer then runs as two
   * launches of the 16-channel z-column kernel (vx_conv3d_args.acc_in) without a concatenated tensor.  F = 8 networks only. */
  const float* split_w[2];
  int32_t split_family;    /* vx_conv3d_k3_family(16, 16) at pack time */
  const float* up3_zc16;   /* nullable: vx_pack_convT_zc16(upscale3): the up half's launch evaluates the transposed conv itself */
} vx_unet3d_weights;

typedef struct vx_unet3d_run {
  const float* x;        /* (V,1,D,H,W) */
  int32_t N, D, H, W;    /* N samples; D,H,W multiples of 16 */
  int32_t repeat;        /* sample n reads volume n / repeat when src == NULL */
  const int32_t* src;    /* nullable [N] */
  const int32_t* flip;   /* nullable [N] */
  const int32_t* dst;    /* nullable [N]: logits slot */
  int32_t drop_mode;     /* VX_DROP_*; HASH uses seed; MASK uses masks[17] */
  uint32_t seed;
  const uint8_t* masks[17]; /* channels-last keep-masks in DROPOUT order (oracle/unet3d_oracle.py) */
  float* logits;         /* [slots][C][D][H][W] */
  void* workspace; size_t workspace_bytes;
  uint32_t* range_flag;  /* nullable device word (zero it before the launch): ends as the bit pattern of the largest
                            |activation| an un-normalised layer handed to a split-fp16 convolution; >= 65504.0f means
                            the fp16 split overflowed somewhere and the logits must not be used (re-run with
                            vx_config.conv_fp32 = 1, the native-fp32 kernels have no such limit) */
  const uint32_t* seed_dev; /* nullable device word added to `seed` by every kernel (graph replays with fresh dropout) */
} vx_unet3d_run;

size_t vx_unet3d_workspace_bytes(int N, int D, int H, int W, int F);
int vx_unet3d_forward(const vx_unet3d_weights* w, const vx_unet3d_run* r, vx_stream_t stream);
/* Diagnostic (bench.py roofline leg; the one entry point that synchronises): the same forward with a HIP event
 * pair around every launch on `stream`; returns per-launch milliseconds and static label strings. */
int vx_unet3d_forward_profiled(const vx_unet3d_weights* w, const vx_unet3d_run* r, vx_stream_t stream,
                               int max_launches, float* ms, const char** labels, int* n_launches);

/* ---------------------------------------------------------------------------------
 * 2D path (HRNet, uncertainty_modeling/models/hrnet_module.py).  Activations: channels-last [N][H][W][pitch] fp32.
 * K15: 3x3 stride 1 / 3x3 stride 2 / 1x1 convolution, padding k/2 (hrnet_module.py:37-41, 85-93, 349-358, 411-428;
 * bias only on last_layer), raw output + per-tile (sum, sumsq) partials [ntiles_total][Cout][2] for the
 * TRAINING-mode BatchNorm that follows (batch statistics over N,H,W; the reference never calls .eval()).
 * Cin must be a multiple of 16, weights packed by vx_pack_conv2d (real channel count: it pads).  The input of C real
 * channels needs NO padding to Cin: in_pitch may be any multiple of 4 in (Cin - 16, ...); channels at and beyond
 * min(Cin, in_pitch) read as zeros, channels [C, in_pitch) must hold zeros (an 18-channel tensor travels at 20 floats
 * per pixel, the 3-channel image at 4). */
typedef struct vx_conv2d_args {
  const float* in; int32_t in_pitch;
  const float* w_packed; const float* bias; /* bias nullable, [Cout] */
  float* out; int32_t out_pitch, out_coff;
  int32_t N, H, W, Cin, Cout, KS, S;        /* KS in {1,3}; S in {1,2} (1x1: S = 1) */
  float* stats_partial;                     /* nullable */
  int32_t w_family;                         /* vx_conv2d_family(Cin, Cout, KS) at the time w_packed was packed */
  /* Optional PROLOGUE (split-fp16 kernels only): `in` is the RAW output of the previous conv; y = x * in_scale[g][c] +
   * in_shift[g][c] (the training-mode BatchNorm folded by vx_bn_finalize[_groups]), then ReLU if in_relu, are applied
   * while the tile is staged -- conv2(relu(bn1(conv1(x)))) of BasicBlock / Bottleneck (hrnet_module.py:59-77, 99-119)
   * without the pass that would write the activated tensor.  g = n / in_group_images (0: one group), rows in_cpitch
   * floats apart.  Zero padding is that of the ACTIVATED tensor. */
  const float* in_scale; const float* in_shift;   /* nullable together */
  int32_t in_cpitch, in_group_images, in_relu;
} vx_conv2d_args;
int64_t vx_conv2d_packed_floats(int Cin, int Cout, int KS);
int vx_pack_conv2d(const float* w_torch /* (Cout,Cin,KS,KS) */, float* w_packed, int Cin, int Cout, int KS, vx_stream_t stream);
int vx_conv2d_tiles(int H, int W, int KS, int S); /* tiles per image (stats_partial has N * this entries) */
int vx_conv2d(const vx_conv2d_args* a, vx_stream_t stream);

/* K16: BatchNorm2d in training mode: partials -> scale = gamma * rstd, shift = beta - mean * scale
 * (biased variance over count = N*OH*OW, eps 1e-5; hrnet_module.py:30, BN_MOMENTUM side effect not reproduced). */
int vx_bn_finalize(const float* stats_partial, int ntiles, int C, int64_t count, float eps, const float* gamma,
                   const float* beta, float* scale, float* shift, vx_stream_t stream);
/* G independent BatchNorm batches in one tensor (the TTA views of test_2D.py:299-311 are separate forwards, each with
 * its own batch statistics; batched here as G groups of consecutive images): group g owns tiles
 * [g * ntiles_per_group, ...) of the partials and row g of scale / shift [G][cpitch]. */
int vx_bn_finalize_groups(const float* stats_partial, int ntiles_per_group, int G, int C, int cpitch, int64_t count_per_group,
                          float eps, const float* gamma, const float* beta, float* scale, float* shift, vx_stream_t stream);

/* K16/K17/K18: out[.., out_coff + c] = act( add + scale[c] * G(drop(x))[c] + shift[c] ); G = identity (OH,OW == H,W) or
 * F.interpolate(mode="bilinear", align_corners=False) from (H,W) to (OH,OW); add / scale / drop optional; `add` may
 * alias `out` (term-by-term SUM fusion, hrnet_module.py:316-333). */
typedef struct vx_affine_args {
  const float* x; int32_t x_pitch;
  const float* scale; const float* shift;   /* nullable together */
  const float* add; int32_t add_pitch;      /* nullable; [N][OH][OW][add_pitch] */
  float* out; int32_t out_pitch, out_coff;
  int32_t N, H, W, C, OH, OW;
  int32_t act;                              /* VX_ACT_NONE | VX_ACT_RELU */
  int32_t drop_mode; uint32_t drop_seed, drop_layer; const uint8_t* drop_mask; /* F.dropout(x, 0.5, training=True) on x */
  int32_t group_images;                     /* > 0: scale / shift are [G][C], image n uses row n / group_images */
} vx_affine_args;
int vx_affine_gather(const vx_affine_args* a, vx_stream_t stream);

/* The SUM fusion of a HighResolutionModule output (hrnet_module.py:316-333: y = f_i0(x_0); y = y + f_ij(x_j) for j = 1 ..;
 * relu) as ONE pass: out = act(T_0 + T_1 + ... ) added in that order, T_j = scale_j * G_j(x_j) + shift_j with G_j the
 * identity (H, W == OH, OW) or the bilinear upsampling of vx_affine_gather, scale_j / shift_j nullable together (the
 * identity term x_i).  Bit-identical to nterms chained vx_affine_gather passes (the same expressions in the same order); the
 * chain re-read and re-wrote the full-resolution accumulator once per term. */
typedef struct vx_fuse_term {
  const float* x; int32_t x_pitch, H, W;
  const float* scale; const float* shift;   /* nullable together; [G][C] rows when group_images > 0 */
} vx_fuse_term;
typedef struct vx_fuse_args {
  vx_fuse_term term[4]; int32_t nterms;     /* 1 .. 4 */
  float* out; int32_t out_pitch;
  int32_t N, OH, OW, C;
  int32_t act;                              /* VX_ACT_NONE | VX_ACT_RELU, applied to the sum */
  int32_t group_images;
} vx_fuse_args;
int vx_fuse_sum(const vx_fuse_args* a, vx_stream_t stream);

/* Final upsample of the class logits to the input size (hrnet_module.py:667-669) into the reference's NCHW layout:
 * image n -> slot dst[n] (nullable) of out [slots][C][OH][OW]; flip[n] bit 0 un-flips a HorizontalFlip TTA view
 * (test_2D.py:304-309), bit 1 a VerticalFlip one (8-view extension, BASELINE config 4). */
int vx_bilinear_nchw(const float* x, int x_pitch, int N, int H, int W, int C, int OH, int OW, float* out,
                     const int32_t* dst, const int32_t* flip, vx_stream_t stream);
/* The same upsample followed by F.softmax(dim=1) (test_2D.py:300-303: `output_softmax = F.softmax(output, dim=1)` of every
 * forward), in one pass: out holds PROBABILITIES, the full-resolution logits are never written.  Bit-identical to
 * vx_bilinear_nchw + vx_softmax_planar. */
int vx_bilinear_softmax_nchw(const float* x, int x_pitch, int N, int H, int W, int C, int OH, int OW, float* out,
                             const int32_t* dst, const int32_t* flip, vx_stream_t stream);

/* ---------------------------------------------------------------------------------
 * K14: sliding-window accumulation of a batch of patches (DataCarrier3D.concat_data,
 * uncertainty_modeling/data_carrier_3D.py:137-179, with the F.softmax of test_3D.py:472 fused):
 *   sum[t][c][crop_b] += softmax_c(logits[b][t]);  count[crop_b] += 1 (once per patch, the reference's pred_idx == 0)
 *   logits [B][T][C][P0][P1][P2] (the pred_idx slots of vx_unet3d_forward), crop [B][3] = (x0, y0, z0) int32 on the
 *   device, sum [T][C][X][Y][Z], count [X][Y][Z] float32, zero-initialised by the caller.
 *   overlap != 0: patches of the batch may overlap (patch_overlap < 1) -> float atomics. 2 <= C <= 8. */
int vx_softmax_accumulate(const float* logits, int B, int T, int C, int P0, int P1, int P2, const int32_t* crop,
                          float* sum, float* count, int X, int Y, int Z, int overlap, vx_stream_t stream);

/* Aleatoric-head sampling (predict_cases, test_3D.py:458-469): mu_s [N][2C][nvox] = final_aleatoric output
 * (mu = first C channels, s = last C, unet3D_module.py:367-369); out [N][T][C][nvox] = mu + exp(s/2) * eps_t with
 * eps [N][T][C][nvox] injected (nullable: generated from `seed`, N(0,1)); sigma [N][C][nvox] nullable output. */
int vx_aleatoric_sample(const float* mu_s, const float* eps, uint32_t seed, int N, int T, int C, int64_t nvox,
                        float* out, float* sigma, vx_stream_t stream);

/* Softmax variance (BASELINE.json north_star; no counterpart in the reference): x [B][T][C][nvox] float32 logits
 * (from_logits != 0) or probabilities -> out [B][nvox] = mean over classes of the population variance over T. */
int vx_softmax_variance(const float* x, int from_logits, int B, int T, int C, int64_t nvox, float* out, vx_stream_t stream);

/* SSN sampling (SsnUNet3D.forward + distribution.sample, ssn_unet3D_module.py:39-70; test_3D.py:361-396):
 * head [N][(2+R)*C][nvox] = the three 1x1x1 heads run as one conv (mean | log_cov_diag | cov_factor, factor channel
 * r*C + c); out [N][S][C][nvox] = mean + sum_r factor_r * eps_w[s][n][r] + sqrt(exp(log_cov_diag) + epsilon) *
 * eps_d[s][n][c][v]  (torch.distributions.LowRankMultivariateNormal.rsample).  eps_w [S][N][R], eps_d [S][N][C][nvox]
 * injected, or NULL: generated from `seed`, N(0,1). */
int vx_ssn_sample(const float* head, const float* eps_w, const float* eps_d, uint32_t seed, int N, int S, int C, int R,
                  int64_t nvox, float epsilon, float* out, vx_stream_t stream);

/* The TTA branch of Cityscapes_dataset.__getitem__ (uncertainty_modeling/data/cityscapes_dataset.py:76-99) as ONE launch:
 * a batch of images -> the G views the 2D driver forwards (test_2D.py:299-311), normalised, channels-last float32 at pitch
 * 4 (channel 3 = 0: the layout the stem convolution stages from), out [G][B][H][W][4], flips as index arithmetic.
 *   src_u8 = 1: src [B][H][W][3] uint8; view = Normalize(mean, std, max_pixel_value)(maybe GaussNoise(maybe Flip(img))) with
 *     the arithmetic of albumentations on uint8 input: noisy = uint8(clip(float(img) + field, 0, 255)), then
 *     (v - mean * max) * (1 / (std * max)) in float32 -- bit-exact with values_amd.data.tta_views_2d.  noise0 / noise1:
 *     additive fields [B][H][W][3] (the generator is third-party: albumentations 1.3.0 GaussNoise, so the fields are INPUTS).
 *   src_u8 = 0: src = clean, noise0 = noisy, both [B][3][H][W] float32 already normalised (values_amd.predict2d.tta_views_8).
 *   view_code [G] (HOST array): bit 0 HorizontalFlip, bit 1 VerticalFlip, bit 2 noisy, bit 3 the noise field is indexed at
 *     the SOURCE pixel (flip of the noisy image) instead of the view's (noise drawn after the flip, as the dataset does),
 *     bit 4 noise slot (u8 source).  The reference's four views: {0, 1, 4, 1 | 4 | 16}. */
int vx_tta_views_2d(const void* src, int src_u8, const float* noise0, const float* noise1, const float* mean, const float* std,
                    float max_pixel_value, int B, int H, int W, int G, const int32_t* view_code, float* out, vx_stream_t stream);

/* Colour rendering of an arg-max mask (Tester.save_prediction, test_2D.py:124-134): rgb[i] = lut[labels[i]] with
 * labels[i] := unlabeled where ignore[i] != 0 (ignore nullable); lut [256][3] uint8, rgb [n][3]. */
int vx_colorize_u8(const uint8_t* labels, const uint8_t* ignore, int64_t n, const uint8_t* lut, int unlabeled, uint8_t* rgb,
                   vx_stream_t stream);

/* K21: threshold search (evaluation/uncertainty_aggregation/find_threshold.py).
 * vx_select_kth: out[0] = k-th smallest (0-based) of the n finite float32 values of x, by radix select
 *   (np.quantile(x, q) = lerp of the two order statistics around q*(n-1), done by the host in float64);
 *   workspace of vx_select_workspace_bytes().  vx_count_nonzero_u8: np.count_nonzero of a label mask. */
int64_t vx_select_workspace_bytes(void);
int vx_select_kth(const float* x, int64_t n, int64_t k, float* out, void* workspace, vx_stream_t stream);
int vx_count_nonzero_u8(const uint8_t* x, int64_t n, uint64_t* out, vx_stream_t stream);

/* Metric reductions behind calculate_test_metrics / calculate_ged (test_3D.py:250-358): see metrics.hip.
 * vx_mask_agreement: masks [M][nvox] uint8 labels < C; counts [M][M][C] uint64 (zeroed here),
 *   counts[i][j][c] = #{v: mask_i(v) == c and mask_j(v) == c}.  M <= 32, C <= 8.
 * vx_soft_metric_sums: prob [C][nvox] float32 (mean softmax), gt [R][nvox] uint8; sums [R][3*C+1] float64:
 *   for each class (sum p_c [gt==c], sum [gt==c], sum p_c), then sum_v log p_{gt(v)}(v);
 *   workspace of vx_soft_metric_workspace_bytes(C, R). */
int vx_mask_agreement(const uint8_t* masks, int M, int C, int64_t nvox, uint64_t* counts, vx_stream_t stream);
int64_t vx_soft_metric_workspace_bytes(int C, int R);
int vx_soft_metric_sums(const float* prob, const uint8_t* gt, int C, int R, int64_t nvox, double* sums, void* workspace,
                        vx_stream_t stream);

/* 2D SSN head (HighResolutionNet.hrnet_ssn, hrnet_module.py:559-595), see accumulate.hip:
 * vx_ssn2d_lowres: channels-last head outputs at the head's resolution -- mean [B*pix][mean_pitch] (C used),
 *   factor [B*pix][factor_pitch] (channel r*C + c) -- and eps_w [S][B][R] (nullable: generated) ->
 *   comb [S][B*pix][C] = mean + sum_r factor_r * eps_w,  expm [B*pix][C] = exp(mean) (nullable).
 * vx_ssn2d_add_diag: out [B][S][per] += sqrt(diag [B][per] + epsilon) * eps_d [S][B][per] (nullable: generated). */
int vx_ssn2d_lowres(const float* mean, int mean_pitch, const float* factor, int factor_pitch, const float* eps_w,
                    uint32_t seed, int B, int64_t pix_per_image, int S, int C, int R, float* comb, float* expm,
                    vx_stream_t stream);
int vx_ssn2d_add_diag(float* out, const float* diag, const float* eps_d, uint32_t seed, int S, int B, int64_t per,
                      float epsilon, vx_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Downstream scalars of the evaluation stage (evaluation/metrics/{ncc,ace}.py): the per-voxel float64 reductions, all
 * deterministic (fixed grid of partial rows added in index order).  workspace: vx_evalmetrics_workspace_bytes().
 *   vx_ncc_sums  (ncc.py:9-25)  pass 0: sums[0..1] = sum gt, sum pred;  pass 1 (means given): sums[0..2] =
 *                sum (gt-mg)^2, sum (pred-mp)^2, sum (gt-mg)(pred-mp)  -- numpy's two-pass mean / std(ddof=1) / product.
 *   vx_platt_sums (ace.py:13-41, sklearn.calibration._sigmoid_calibration on (-unc, reference == prediction)):
 *                ref [R][nvox] int32 reference segmentations, pred [nvox] int32 mean prediction, unc [nvox] f32/f64;
 *                voxels with ref == ignore_value dropped (ignore_value < 0: none).  For the sigmoid parameters (A, B)
 *                and Platt's targets t_pos / t_neg: sums[0] = valid count, [1] = correct count, [2] = loss,
 *                [3..4] = gradient (dA, dB), [5..7] = Hessian (AA, AB, BB).  The host iterates (Newton).
 *   vx_calib_bins (ace.py:44-90) platt_scale_confid + calib_stats' 20-bin statistics: bins63 = bin_sums[21],
 *                bin_true[21], bin_total[21] for the bin edges edges21 (a HOST array: np.linspace(0, 1 + 1e-8, 21)). */
int64_t vx_evalmetrics_workspace_bytes(void);
int vx_ncc_sums(const void* gt, int gt_dtype, const void* pred, int pred_dtype, int64_t n, int pass, double mean_gt,
                double mean_pred, double* sums, void* workspace, vx_stream_t stream);
int vx_platt_sums(const void* unc, int dtype, const int32_t* ref, const int32_t* pred, int R, int64_t nvox,
                  int ignore_value, double A, double B, double t_pos, double t_neg, double* sums, void* workspace,
                  vx_stream_t stream);
int vx_calib_bins(const void* unc, int dtype, const int32_t* ref, const int32_t* pred, int R, int64_t nvox,
                  int ignore_value, double A, double B, const double* edges21, double* bins63, void* workspace,
                  vx_stream_t stream);

/* ---------------------------------------------------------------------------------
 * K19/K20: map -> scalar aggregations (evaluation/uncertainty_aggregation/aggregate_uncertainties.py).
 *   vx_box_max : patch_level_aggregation (:13-31): box-sum 'valid' (pd,ph,pw) in float64, max and
 *                first (C-order) index with isclose(value, max); result[0]=max, idx[0..2]
 *   vx_sum_thr : image_level_aggregation (:34-37) + threshold_aggregation (:61-67):
 *                sums[0]=sum(map), sums[1]=sum(map[map>=thr]), sums[2]=count(map>=thr); map VX_F32 or VX_F64 (a map
 *                read back from NIfTI is float64), the comparison (double)map >= thr runs in float64 like the reference's
 * vx_box_max: map is f32 [D][H][W] (2D maps: D = 1, pd = 1). workspace: 2 * D*H*W doubles.
 */
int vx_box_max(const float* map, int D, int H, int W, int pd, int ph, int pw, double* result, int32_t* idx,
               void* workspace, size_t workspace_bytes, vx_stream_t stream);
int vx_sum_thr(const void* map, int dtype, int64_t n, double thr, double* sums, vx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VALUES_AMD_H */
