// Whole-network launch: UNet3D.forward (uncertainty_modeling/models/unet3D_module.py:296-373) for a batch
// of N samples (MC-dropout samples x TTA views x volumes), as ~45 kernel launches on the caller's
// stream -- no allocation, no synchronisation, capturable into one hipGraph.
//
// Data flow (channels-last fp32, level l has spatial (D,H,W) >> l and C_l = F << l channels):
//   encoder l:  conv -> A_l (raw + stats) -> finalize -> IN/LReLU/drop in place
//               conv -> B_l (raw + stats) -> finalize -> IN/LReLU/drop -> skip half of CAT_l + pool -> P_{l+1}
//   center:     conv+ReLU, conv+ReLU, convT+ReLU+drop -> up half of CAT_3
//   decoder l:  conv(CAT_l)+LReLU+drop -> A_l, conv+LReLU+drop -> B_l, convT -> up half of CAT_{l-1}
//   final:      1x1x1 conv(B_0) -> logits NCDHW, slot dst[n], un-flipped
// torch.cat never happens (K7): CAT_l is the x-blocked buffer [N][D][H][W/xb][2][xb][C_l] (values_amd.h) whose two
// halves are written as dense blocks by their producers and read by the decoder conv through in_xblk.
// First layer in MC-dropout mode: computed once per volume (the T samples share input and statistics).
#include "common.h"
#include <stdlib.h>

extern "C" int vx_conv3d_k3_c1_tiles(int D, int H, int W);

namespace {
struct Level { int D, H, W, C; int64_t nvox; };

struct Plan {
  Level lv[5];
  float *A[4], *B[4], *CAT[4], *P[5], *C0, *C1;
  float *stats, *mean, *rstd;
  float *mean0, *rstd0;   // level-0 skip statistics: kept until the decoder normalises the skip half itself
  float *meanS[4], *rstdS[4];   // (round 5) the same for the levels whose second contract conv leaves its RAW output in the skip half
  size_t bytes;
};

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

static void make_plan(Plan& p, int N, int D, int H, int W, int F, char* base) {
  size_t off = 0;
  auto carve = [&](size_t floats) {
    float* q = base ? (float*)(base + off) : nullptr;
    off += align_up(floats * sizeof(float));
    return q;
  };
  for (int l = 0; l < 5; ++l) {
    p.lv[l].D = D >> l; p.lv[l].H = H >> l; p.lv[l].W = W >> l; p.lv[l].C = F << l;
    p.lv[l].nvox = (int64_t)p.lv[l].D * p.lv[l].H * p.lv[l].W;
  }
  for (int l = 0; l < 4; ++l) {
    const size_t e = (size_t)N * p.lv[l].nvox * p.lv[l].C;
    p.A[l] = carve(e);
    p.B[l] = carve(e);
    p.CAT[l] = carve(2 * e);
  }
  p.P[0] = nullptr;
  for (int l = 1; l < 5; ++l) p.P[l] = carve((size_t)N * p.lv[l].nvox * p.lv[l - 1].C);
  p.C0 = carve((size_t)N * p.lv[4].nvox * p.lv[4].C);
  p.C1 = carve((size_t)N * p.lv[4].nvox * p.lv[4].C);
  // statistics: the largest partial buffer is level 0
  size_t smax = 0;
  for (int l = 0; l < 4; ++l) {
    size_t t = (size_t)vx_conv3d_k3_tiles(p.lv[l].D, p.lv[l].H, p.lv[l].W);
    if (l == 0) {
      size_t t1 = (size_t)vx_conv3d_k3_c1_tiles(p.lv[0].D, p.lv[0].H, p.lv[0].W);
      if (t1 > t) t = t1;
    }
    const size_t e = (size_t)N * t * p.lv[l].C * 2;
    if (e > smax) smax = e;
  }
  p.stats = carve(smax);
  p.mean = carve((size_t)N * p.lv[3].C);
  p.rstd = carve((size_t)N * p.lv[3].C);
  p.mean0 = carve((size_t)N * p.lv[0].C);
  p.rstd0 = carve((size_t)N * p.lv[0].C);
  for (int l = 0; l < 4; ++l) {
    p.meanS[l] = l == 0 ? p.mean0 : carve((size_t)N * p.lv[l].C);
    p.rstdS[l] = l == 0 ? p.rstd0 : carve((size_t)N * p.lv[l].C);
  }
  p.bytes = off;
}
}  // namespace

extern "C" size_t vx_unet3d_workspace_bytes(int N, int D, int H, int W, int F) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || F <= 0) return 0;
  Plan p;
  make_plan(p, N, D, H, W, F, nullptr);
  return p.bytes;
}

// Optional per-launch timing (vx_unet3d_forward_profiled): a HIP event pair around every launch, on the
// caller's stream.  Off (g_prof == nullptr) in normal and captured runs.
namespace {
struct Prof {
  static constexpr int MAXL = 96;
  hipEvent_t ev[2 * MAXL];
  const char* name[MAXL];
  const char* kernel[MAXL];
  int n = 0;
  hipStream_t s;
};
thread_local Prof* g_prof = nullptr;
thread_local char g_prof_labels[Prof::MAXL][192];   // "label|kernel instance", valid until the thread's next profiled forward
}  // namespace

#define VX_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != VX_OK) return rc_; \
  } while (0)

#define VX_STEP(label, expr)                                                       \
  do {                                                                             \
    Prof* pf_ = g_prof;                                                            \
    if (pf_ && pf_->n < Prof::MAXL) { vx_note_kernel(nullptr); (void)hipEventRecord(pf_->ev[2 * pf_->n], pf_->s); }  \
    int rc_ = (expr);                                                              \
    if (rc_ != VX_OK) return rc_;                                                  \
    if (pf_ && pf_->n < Prof::MAXL) {                                              \
      (void)hipEventRecord(pf_->ev[2 * pf_->n + 1], pf_->s);                             \
      pf_->kernel[pf_->n] = vx_last_kernel();                                      \
      pf_->name[pf_->n++] = (label);                                               \
    }                                                                              \
  } while (0)

extern "C" int vx_unet3d_forward(const vx_unet3d_weights* w, const vx_unet3d_run* r, vx_stream_t stream) {
  if (!w || !r) VX_FAIL(VX_E_NULL, "vx_unet3d_forward: null argument");
  if (!r->x || !r->logits || !r->workspace) VX_FAIL(VX_E_NULL, "vx_unet3d_forward: null tensor/workspace");
  const int N = r->N, D = r->D, H = r->H, W = r->W, F = w->F, NC = w->num_classes;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0) VX_FAIL(VX_E_SHAPE, "vx_unet3d_forward: empty batch");
  if (D % 16 || H % 16 || W % 16)
    VX_FAIL(VX_E_SHAPE, "vx_unet3d_forward: spatial size (%d,%d,%d) must be divisible by 16 (4 poolings; InstanceNorm "
            "needs > 1 voxel at the 8x level)", D, H, W);
  if (F != 8 && F != 16 && F != 32) VX_FAIL(VX_E_SHAPE, "vx_unet3d_forward: initial_filter_size %d unsupported (8,16,32)", F);
  if (NC <= 0) VX_FAIL(VX_E_SHAPE, "vx_unet3d_forward: num_classes %d", NC);
  if (r->drop_mode < 0 || r->drop_mode > 2) VX_FAIL(VX_E_DTYPE, "vx_unet3d_forward: drop_mode %d", r->drop_mode);
  if ((((uintptr_t)r->workspace) & 255u) != 0) VX_FAIL(VX_E_ALIGN, "vx_unet3d_forward: workspace must be 256-byte aligned");
  for (int i = 0; i < 18; ++i)
    if (!w->conv_w[i] || !w->conv_b[i]) VX_FAIL(VX_E_NULL, "vx_unet3d_forward: conv weight %d missing", i);
  if (w->in_channels < 0 || w->in_channels > 8) VX_FAIL(VX_E_SHAPE, "vx_unet3d_forward: in_channels %d (1 .. 8)", w->in_channels);
  for (int i = 0; i < 4; ++i)
    if (!w->up_w[i] || !w->up_b[i]) VX_FAIL(VX_E_NULL, "vx_unet3d_forward: transposed-conv weight %d missing", i);
  if (!w->final_w || !w->final_b) VX_FAIL(VX_E_NULL, "vx_unet3d_forward: final weights missing");
  if (r->drop_mode == VX_DROP_MASK)
    for (int i = 0; i < 17; ++i)
      if (!r->masks[i]) VX_FAIL(VX_E_NULL, "vx_unet3d_forward: mask %d missing", i);

  Plan p;
  make_plan(p, N, D, H, W, F, (char*)r->workspace);
  if (p.bytes > r->workspace_bytes)
    VX_FAIL(VX_E_WORKSPACE, "vx_unet3d_forward: workspace %zu B < required %zu B", r->workspace_bytes, p.bytes);

  const int dm = r->drop_mode;
  auto mask = [&](int i) { return dm == VX_DROP_MASK ? r->masks[i] : (const uint8_t*)nullptr; };

  auto xblk_of = [](int W) { return W % 4 == 0 ? 4 : (W % 2 == 0 ? 2 : 1); };
  // head fusion: where the last 3x3x3 conv runs on the kernel that holds a voxel's channels in one lane
  const bool fuse_head = NC <= 4 && vx_conv3d_k3_head_fusable(F, F) && !vx_cfg().no_head_fusion;
  // pre: the input is a contract block's RAW conv output; its InstanceNorm (p.mean / p.rstd), LeakyReLU and dropout
  // layer pre_layer are applied by the conv while it stages its tiles (pre_rep samples share one raw tensor)
  bool pre_split_ = false;           // set around the contr_1_2 launch when its input went through vx_prenorm_split
  bool osplit_ = false, usplit_ = false;   // pre-split hand-over of B_1 (expand_2_2 -> upscale2 inside expand_1_1)
  bool in_planar_ = false;                 // set around expand_2_2's launch when expand_2_1 left its output planar (round 6)
  int products_ = 0;                       // 1 around the three full-resolution launches in the fp16-products mode (storage16 = 2)
  int st16_ = 0;                     // reduced-storage mode: 1 around expand_1_1's launch (fp16 output), 2 around expand_1_2's (fp16 input)
  float* pool_raw_ = nullptr;        // set around the contr_1_2 launch when its epilogue pools (fuse_pool below)
  uint32_t* pool_flags_ = nullptr;
  const uint32_t* pf_in_flags_ = nullptr;   // set around the contr_2_1 launch: vx_conv3d_args.in_pool_flags
  bool poolfin_on_load_ = false;     // contr_2_1 reads contr_1_2's window maxima + flags and finishes them on load
  const float* poolfin_raw_ = nullptr;
  const uint32_t* poolfin_flags_ = nullptr;
  auto conv = [&](const float* in, int in_pitch, int wi, float* out, int out_pitch, int out_coff, const Level& L, int Cin,
                  int Cout, int act, int drop_layer, float* stats, int in_xblk, int pre_layer = -1, int pre_rep = 1,
                  const float* pre_mean = nullptr, const float* pre_rstd = nullptr, int out_xblk = 0, int n_samples = 0,
                  const float* up_in = nullptr, int up_idx = 0, int up_pitch = 0) {
    vx_conv3d_args a = {};
    if (up_in) {
      a.up_in = up_in; a.up_w = w->up_w[up_idx]; a.up_b = w->up_b[up_idx]; a.up_pitch = up_pitch;
      a.up_fused = w->up_fused;      // (nullable) the up-convolution composed into expand_1_1's weights; vx_config.s16_no_upcompose
    }
    a.head_out = nullptr; a.head_w = nullptr; a.head_b = nullptr; a.head_dst = nullptr; a.head_flip = nullptr; a.head_C = 0;
    if (fuse_head && wi == 17) {   // expand_1_2: the final 1x1x1 conv rides in its epilogue, B_0 is never stored
      a.head_out = r->logits; a.head_w = w->final_w; a.head_b = w->final_b; a.head_C = NC;
      a.head_dst = r->dst; a.head_flip = r->flip;
      out = nullptr;
    }
    a.in_xblk = in_xblk;
    a.w_family = w->conv_family[wi];
    a.in = in; a.w_packed = w->conv_w[wi]; a.bias = w->conv_b[wi]; a.out = out;
    a.in_pitch = in_pitch; a.out_pitch = out_pitch; a.out_coff = out_coff;
    a.N = n_samples > 0 ? n_samples : N; a.D = L.D; a.H = L.H; a.W = L.W; a.Cin = Cin; a.Cout = Cout;
    a.act = act;
    a.drop_mode = drop_layer >= 0 ? dm : VX_DROP_NONE;
    a.drop_seed = r->seed; a.drop_layer = (uint32_t)(drop_layer >= 0 ? drop_layer : 0);
    a.drop_mask = drop_layer >= 0 ? mask(drop_layer) : nullptr;
    a.stats_partial = stats;
    a.out_xblk = out_xblk; a.out_half = 1;
    if (pre_layer >= 0) {
      a.in_mean = pre_mean ? pre_mean : p.mean; a.in_rstd = pre_rstd ? pre_rstd : p.rstd;
      a.in_drop_mode = dm; a.in_drop_seed = r->seed; a.in_drop_layer = (uint32_t)pre_layer;
      a.in_repeat = pre_rep;
      a.in_split = pre_split_ ? 1 : 0;
      a.in_pool_flags = pf_in_flags_;
    }
    a.seed_dev = r->seed_dev;
    a.out_f16 = st16_ == 1 ? 1 : 0;
    a.in_f16 = st16_ == 2 ? 1 : 0;
    a.out_split = osplit_ ? 1 : 0;       // expand_2_2 hands B_1 to the fused up-convolution as fp16 pairs
    a.in_planar = in_planar_ ? 1 : 0;
    a.products = products_;
    a.up_split = (up_in && usplit_) ? 1 : 0;
    a.range_flag = stats ? nullptr : r->range_flag;   // decoder / center outputs feed split-fp16 consumers un-normalised
    if (pool_raw_ && (wi & 1) && wi < 8) {   // contr_l_2 also leaves the window maxima of its block's MaxPool (dropout layer 2 l + 1 = wi)
      a.pool_out = pool_raw_; a.pool_flags = pool_flags_;
      a.drop_mode = dm; a.drop_seed = r->seed; a.drop_layer = (uint32_t)wi;
    }
    return vx_conv3d_k3(&a, stream);
  };
  bool norm_stats_ = false;          // set around a normalise pass that reduces the conv's partials itself (vx_norm_act_drop_pool_stats)
  int norm_tiles_ = 0;
  auto norm = [&](const float* x, int C, float* out, int out_pitch, int out_coff, float* pool, const Level& L,
                  int drop_layer, int x_repeat, int out_xblk, int x_xblk = 0, const float* mean = nullptr,
                  const float* rstd = nullptr, bool normalise = true, int act = VX_ACT_LRELU) {
    vx_norm_args a = {};
    a.out_xblk = out_xblk; a.out_half = 1;
    a.x_xblk = x_xblk; a.x_half = 1;
    a.seed_dev = r->seed_dev;
    a.range_flag = normalise ? nullptr : r->range_flag;   // an un-normalised tensor on its way to a split-fp16 conv
    a.x = x; a.x_pitch = C;
    a.mean = normalise ? (mean ? mean : p.mean) : nullptr;
    a.rstd = normalise ? (rstd ? rstd : p.rstd) : nullptr;
    a.out = out; a.out_pitch = out_pitch; a.out_coff = out_coff;
    a.pool_out = pool; a.pool_pitch = C;
    a.N = N; a.D = L.D; a.H = L.H; a.W = L.W; a.C = C;
    a.act = act;
    a.drop_mode = drop_layer >= 0 ? dm : VX_DROP_NONE; a.drop_seed = r->seed; a.drop_layer = (uint32_t)(drop_layer >= 0 ? drop_layer : 0);
    a.drop_mask = drop_layer >= 0 ? mask(drop_layer) : nullptr;
    if (norm_stats_ && normalise && x_repeat == 1) {
      a.mean = nullptr; a.rstd = nullptr;
      vx_stat_src st = {p.stats, norm_tiles_, 1e-5f, (int64_t)L.nvox, p.mean, p.rstd};
      return vx_norm_act_drop_pool_stats(&a, &st, stream);
    }
    return vx_norm_act_drop_pool_bcast(&a, x_repeat, stream);
  };
  auto convT = [&](const float* in, int ui, float* out, int out_pitch, const Level& Lin, int Cin, int Cout, int act,
                   int drop_layer, bool dense = false) {
    vx_convT_args a = {};
    a.range_flag = r->range_flag;
    a.seed_dev = r->seed_dev;
    a.out_xblk = dense ? 0 : xblk_of(2 * Lin.W); a.out_half = 0;
    a.in = in; a.in_pitch = Cin; a.w_packed = w->up_w[ui]; a.bias = w->up_b[ui];
    a.out = out; a.out_pitch = out_pitch; a.out_coff = 0;
    a.N = N; a.D = Lin.D; a.H = Lin.H; a.W = Lin.W; a.Cin = Cin; a.Cout = Cout;
    a.act = act;
    a.drop_mode = drop_layer >= 0 ? dm : VX_DROP_NONE;
    a.drop_seed = r->seed; a.drop_layer = (uint32_t)(drop_layer >= 0 ? drop_layer : 0);
    a.drop_mask = drop_layer >= 0 ? mask(drop_layer) : nullptr;
    return vx_convT_k2s2(&a, stream);
  };

  static const char* kConv[18] = {"contr_1_1", "contr_1_2", "contr_2_1", "contr_2_2", "contr_3_1", "contr_3_2",
                                  "contr_4_1", "contr_4_2", "center.0", "center.2", "expand_4_1", "expand_4_2",
                                  "expand_3_1", "expand_3_2", "expand_2_1", "expand_2_2", "expand_1_1", "expand_1_2"};
  const char* kLast = fuse_head ? "expand_1_2+final" : kConv[17];   // a fused launch is labelled with every layer it computes
  static const char* kNorm[8] = {"norm:contr_1_1", "norm:contr_1_2", "norm:contr_2_1", "norm:contr_2_2",
                                 "norm:contr_3_1", "norm:contr_3_2", "norm:contr_4_1", "norm:contr_4_2"};
  static const char* kFin[8] = {"finalize:contr_1_1", "finalize:contr_1_2", "finalize:contr_2_1", "finalize:contr_2_2",
                                "finalize:contr_3_1", "finalize:contr_3_2", "finalize:contr_4_1", "finalize:contr_4_2"};
  static const char* kUp[4] = {"center.4", "upscale4", "upscale3", "upscale2"};
  // level 0 on the z-column kernel: contr_1_2 and expand_1_1 normalise their inputs themselves (no normalised
  // full-resolution tensor is ever written); needs the hash generator or no dropout -- injected masks take the
  // general kernels with their separate normalise passes
  const bool pre0 = dm != VX_DROP_MASK && !vx_cfg().s16_no_prenorm && F == 8 && !w->no_instancenorm &&
                    vx_conv3d_k3_prologue_ok(p.lv[0].D, p.lv[0].H, p.lv[0].W, F, F);
  // contr_1_2's raw output straight into the skip half + a pooling-only pass + expand_1_1 normalising its skip half
  // (vx_config.s16_skip_raw, default 1): the pass shrinks 1.10 -> 0.58 ms per 320 samples, expand_1_1 grows 2.38 -> 2.69
  // now that its staging runs in producer waves (on the kernel where every wave staged AND multiplied it grew 2.33 -> 2.95
  // and the fusion was neutral)
  // upscale2 inside expand_1_1: the up half of CAT_0 is computed from B_1 while expand_1_1 stages its tiles and never
  // exists in memory (conv3d_xp8w.hip, UP = 1)
  const bool fuse_up = dm != VX_DROP_MASK && F == 8 && vx_conv3d_k3_upfuse_ok(p.lv[0].D, p.lv[0].H, p.lv[0].W, 2 * F, F);
  // MaxPool of the first block out of contr_1_2's epilogue (window maxima of the kept raw values + any-dropped bits, then
  // vx_pool_finish on 1/8 of the voxels) instead of a pass that re-reads the full-resolution tensor
  const bool fuse_pool = dm != VX_DROP_MASK && vx_conv3d_k3_poolfuse_ok(p.lv[0].D, p.lv[0].H, p.lv[0].W, F, F);
  const bool fuse0 = pre0 && vx_cfg().s16_skip_raw && vx_conv3d_k3_prologue_ok(p.lv[0].D, p.lv[0].H, p.lv[0].W, 2 * F, F);
  // opt-in reduced-storage mode (vx_config.storage16, never the default): expand_1_1 -> expand_1_2 hand their tensor over as
  // fp16; needs the z-column kernels on both and the fused head
  const bool st16 = vx_cfg().storage16 && dm != VX_DROP_MASK && F == 8 && fuse_head && vx_cfg().conv_fp32 == 0 &&
                    vx_conv3d_k3_prologue_ok(p.lv[0].D, p.lv[0].H, p.lv[0].W, F, F) && dm == VX_DROP_HASH;
  bool skip_raw[4] = {false, false, false, false};   // level l: the skip half of CAT_l holds contr_l_2's RAW output (decoder normalises on load)
  // level l: CAT_l is used as TWO DENSE tensors (up = first half of the buffer, skip = second) and expand_l_1 runs as two
  // launches of the 16-channel z-column kernel over them (vx_conv3d_args.acc_in) -- level 1 of the F = 8 networks
  bool halves[4] = {false, false, false, false};
  bool fuse_up1 = false, usplit1_ = false;   // upscale3 inside expand_2_1's up-half launch; B_2 handed over as fp16 pairs
  bool planar1 = false;                      // expand_2_1's output left as the planar pre-split tensor expand_2_2 stages by LDS-DMA
  // ---------------- encoder ----------------
  const bool inorm = !w->no_instancenorm;
  const int ICH = w->in_channels > 1 ? w->in_channels : 1;
  // the first conv over n_out samples (raw output + bias; statistics if asked): Cin == 1 on its own VALU kernel, more
  // input channels zero-padded to 8 on the general kernels (input laid out channels-last in B_0, free until contr_1_2)
  auto first_conv = [&](float* out, int n_out, int rep, const int32_t* src, const int32_t* flip, float* stats) -> int {
    const Level& L = p.lv[0];
    if (ICH == 1)
      return vx_conv3d_k3_c1(r->x, w->conv_w[0], w->conv_b[0], out, L.C, n_out, L.D, L.H, L.W, L.C, rep, src, flip, stats, stream);
    VX_TRY(vx_pack_input_cl8(r->x, p.B[0], n_out, ICH, L.D, L.H, L.W, rep, src, flip, stream));
    return conv(p.B[0], 8, 0, out, L.C, 0, L, 8, L.C, VX_ACT_NONE, -1, stats, 0, -1, 1, nullptr, nullptr, 0, n_out);
  };
  for (int l = 0; l < 4; ++l) {
    const Level& L = p.lv[l];
    const int C = L.C;
    int ntiles;
    int pre_layer = -1, pre_rep = 1;       // contr_l_2 normalises its own input (no separate pass over A_l)
    const float* in2 = p.A[l];
    if (l == 0) {
      ntiles = ICH == 1 ? vx_conv3d_k3_c1_tiles(L.D, L.H, L.W) : vx_conv3d_k3_tiles_for(L.D, L.H, L.W, C);
      const int rep = r->repeat > 0 ? r->repeat : 1;
      const bool fuse_norm = pre0 && inorm;
      if (!r->src && !r->flip && rep > 1 && N % rep == 0) {
        // MC-dropout: the T samples of a volume share this conv and its statistics -> once per volume into a
        // scratch; contr_1_2 reads that scratch with T dropout patterns, or (general kernels) the norm kernel fans it out
        const int V = N / rep;
        // (scratch: A_0 when CAT_0 takes contr_1_2's raw output; else CAT_0, free until contr_1_2's norm)
        float* scratch = fuse0 ? p.A[0] : p.CAT[0];
        VX_STEP(kConv[0], first_conv(scratch, V, 1, nullptr, nullptr, inorm ? p.stats : nullptr));
        // (vx_prenorm_split_stats -- the pre-split pass reducing the 256 partial tiles of a 64^3 volume itself -- measured 0.095 ms
        // against 0.079 + 0.007 for the finalize launch and the plain pass: the launch stays here)
        const bool presplit = fuse_norm && C == 8 && dm != VX_DROP_MASK && !vx_cfg().s16_no_presplit && vx_conv3d_k3_presplit_ok(L.D, L.H, L.W, F, F);
        if (inorm) VX_STEP(kFin[0], vx_instnorm_finalize(p.stats, V, ntiles, C, L.nvox, 1e-5f, p.mean, p.rstd, stream));
        if (fuse_norm) {
          in2 = scratch; pre_layer = 0; pre_rep = rep;
          // InstanceNorm + LeakyReLU + the fp16 split of the shared tensor ONCE per volume (in place); contr_1_2's staging
          // waves then only AND sample n's dropout bits in -- a third of their vector work (they are that layer's critical
          // path: tools/stamp_s16.py)
          // (only where contr_1_2 runs on the z-column kernel: the tile kernel's prologue reads the RAW tensor through
          // in_repeat -- s16_no_xp = 1 with n_pred > 1 failed with VX_E_SHAPE after the scratch had been rewritten)
          if (presplit) {
            VX_STEP("presplit:contr_1_1", vx_prenorm_split(scratch, p.mean, p.rstd, V, L.nvox, dm == VX_DROP_HASH ? 2.f : 1.f, stream));
            pre_split_ = true;
          }
        }
        else VX_STEP(kNorm[0], norm(p.CAT[0], C, p.A[0], C, 0, nullptr, L, 0, rep, 0, 0, nullptr, nullptr, inorm));
      } else {
        VX_STEP(kConv[0], first_conv(p.A[0], N, rep, r->src, r->flip, inorm ? p.stats : nullptr));
        if (inorm) VX_STEP(kFin[0], vx_instnorm_finalize(p.stats, N, ntiles, C, L.nvox, 1e-5f, p.mean, p.rstd, stream));
        if (fuse_norm) pre_layer = 0;
        else VX_STEP(kNorm[0], norm(p.A[0], C, p.A[0], C, 0, nullptr, L, 0, 1, 0, 0, nullptr, nullptr, inorm));
      }
    } else if (inorm) {
      ntiles = vx_conv3d_k3_tiles_for(L.D, L.H, L.W, C);
      if (l == 1 && poolfin_on_load_) {
        pf_in_flags_ = poolfin_flags_;
        VX_STEP("poolfin+contr_2_1", conv(poolfin_raw_, C / 2, 2 * l, p.A[l], C, 0, L, C / 2, C, VX_ACT_NONE, -1, p.stats, 0, 1, 1, p.mean0,
                                          p.rstd0));
        pf_in_flags_ = nullptr;
      } else
      VX_STEP(kConv[2 * l], conv(p.P[l], C / 2, 2 * l, p.A[l], C, 0, L, C / 2, C, VX_ACT_NONE, -1, p.stats, 0));
      VX_STEP(kFin[2 * l], vx_instnorm_finalize(p.stats, N, ntiles, C, L.nvox, 1e-5f, p.mean, p.rstd, stream));
      // contr_l_2 normalises the raw A_l on load (tile kernel prologue) -- or a pass rewrites A_l in place
      if (dm != VX_DROP_MASK && !vx_cfg().s16_no_prenorm && vx_conv3d_k3_prologue_ok(L.D, L.H, L.W, C, C)) pre_layer = 2 * l;
      else VX_STEP(kNorm[2 * l], norm(p.A[l], C, p.A[l], C, 0, nullptr, L, 2 * l, 1, 0));
    } else {
      // do_instancenorm=False (unet3D_module.py:238-243): conv + LeakyReLU + Dropout, all in the conv's epilogue
      VX_STEP(kConv[2 * l], conv(p.P[l], C / 2, 2 * l, p.A[l], C, 0, L, C / 2, C, VX_ACT_LRELU, 2 * l, nullptr, 0));
    }
    ntiles = vx_conv3d_k3_tiles_for(L.D, L.H, L.W, C);
    struct ClearSplit { bool& f; ~ClearSplit() { f = false; } } clear_split_{pre_split_};   // the flag covers this level's second conv only
    if (!inorm) {
      // second conv of the block with its activation / dropout fused; one streaming pass copies it into the skip half of
      // the concat buffer and pools it (no normalisation, no activation)
      VX_STEP(kConv[2 * l + 1], conv(in2, C, 2 * l + 1, p.B[l], C, 0, L, C, C, VX_ACT_LRELU, 2 * l + 1, nullptr, 0));
      VX_STEP(kNorm[2 * l + 1], norm(p.B[l], C, p.CAT[l], 2 * C, C, p.P[l + 1], L, -1, 1, xblk_of(L.W), 0, nullptr, nullptr, false,
                                     VX_ACT_NONE));
      continue;
    }
    if (l == 0 && fuse0) {
      // contr_1_2's RAW output goes straight into the skip half of CAT_0; one pooling pass produces P_1 from it, and
      // expand_1_1 normalises the skip half while it stages its tiles (statistics kept in mean0 / rstd0 until then):
      // the full-resolution tensor is written once and read twice instead of written twice and read twice
      if (fuse_pool) {   // B_0 is free until the decoder: window maxima [N][nvox / 8][8] + flag words [N][nvox / 8][2]
        pool_raw_ = p.B[0];
        pool_flags_ = reinterpret_cast<uint32_t*>(p.B[0] + (size_t)N * p.lv[1].nvox * 8);
      }
      products_ = (st16 && vx_cfg().storage16 == 2 && pre_split_ && fuse_pool) ? 1 : 0;
      VX_STEP(kConv[1], conv(in2, C, 1, p.CAT[0], C, 0, L, C, C, VX_ACT_NONE, -1, p.stats, 0, pre_layer, pre_rep, nullptr,
                             nullptr, xblk_of(L.W)));
      products_ = 0;
      float* praw = pool_raw_;
      uint32_t* pfl = pool_flags_;
      pool_raw_ = nullptr; pool_flags_ = nullptr;
      VX_STEP(kFin[1], vx_instnorm_finalize(p.stats, N, ntiles, C, L.nvox, 1e-5f, p.mean0, p.rstd0, stream));
      // Round 4: contr_2_1 finishes the window maxima itself while it stages its tiles (vx_conv3d_args.in_pool_flags): the pass
      // over the pooled tensor, its launch and the tensor are gone
      poolfin_on_load_ = praw && !vx_cfg().s16_no_poolfin && vx_conv3d_k3_poolfin_ok(C, 2 * C);
      if (poolfin_on_load_) { poolfin_raw_ = praw; poolfin_flags_ = pfl; }
      else if (praw)
        VX_STEP("poolfin:contr_1_2", vx_pool_finish(praw, pfl, p.mean0, p.rstd0, p.P[1], C, N, p.lv[1].nvox, dm == VX_DROP_HASH, stream));
      else
        VX_STEP("pool:contr_1_2", norm(p.CAT[0], C, nullptr, 0, 0, p.P[1], L, 1, 1, 0, xblk_of(L.W), p.mean0, p.rstd0));
      continue;
    }
    // Round 5 (levels below full resolution on the 16-channel z-column kernel, conv3d_zc16.hip): the level-0 data flow -- the
    // second conv writes its RAW output straight into the skip half, leaves the (y, x) half of the block's MaxPool
    // (window maxima of the kept raw values + any-dropped bits) next to it, vx_pool_finish_z produces P_{l+1} from a quarter
    // of the voxels once the statistics exist, and the decoder's first conv of the level normalises the skip half on load.
    // The normalise + pool pass over the whole tensor (0.25 ms per 320 samples at level 1) is gone.
    if (l >= 1 && dm != VX_DROP_MASK && !vx_cfg().s16_no_prenorm && vx_cfg().s16_skip_raw &&
        vx_conv3d_k3_pool_layout(L.D, L.H, L.W, C, C) == 2 && vx_conv3d_k3_skip_prologue_ok(L.D, L.H, L.W, 2 * C, C, xblk_of(L.W))) {
      // B_l is free until the decoder: window maxima [N][D][H/2][W/2][C] + flag words [N][D][H/2][W/2][C/4]
      pool_raw_ = p.B[l];
      pool_flags_ = reinterpret_cast<uint32_t*>(p.B[l] + (size_t)N * (L.nvox / 4) * C);
      halves[l] = l == 1 && C == 16 && w->split_w[0] && w->split_w[1] && !vx_cfg().s16_no_halves &&
                  w->split_family == vx_conv3d_k3_family(16, 16) && vx_conv3d_k3_acc_ok(L.D, L.H, L.W, C, C);
      if (halves[l])
        VX_STEP(kConv[2 * l + 1], conv(in2, C, 2 * l + 1, p.CAT[l] + (size_t)N * L.nvox * C, C, 0, L, C, C, VX_ACT_NONE, -1, p.stats, 0,
                                       pre_layer, pre_rep));
      else
      VX_STEP(kConv[2 * l + 1], conv(in2, C, 2 * l + 1, p.CAT[l], C, 0, L, C, C, VX_ACT_NONE, -1, p.stats, 0, pre_layer, pre_rep, nullptr,
                                     nullptr, xblk_of(L.W)));
      float* praw = pool_raw_;
      uint32_t* pfl = pool_flags_;
      pool_raw_ = nullptr; pool_flags_ = nullptr;
      {   // (round 5: the pass reduces the partials itself and leaves meanS / rstdS for the decoder's prologue)
        vx_stat_src st = {p.stats, ntiles, 1e-5f, (int64_t)L.nvox, p.meanS[l], p.rstdS[l]};
        VX_STEP(l == 1 ? "poolfin:contr_2_2" : (l == 2 ? "poolfin:contr_3_2" : "poolfin:contr_4_2"),
                vx_pool_finish_z_stats(praw, pfl, &st, p.P[l + 1], C, N, L.D / 2, (int64_t)(L.H / 2) * (L.W / 2), dm == VX_DROP_HASH, stream));
      }
      skip_raw[l] = true;
      continue;
    }
    VX_STEP(kConv[2 * l + 1], conv(in2, C, 2 * l + 1, p.B[l], C, 0, L, C, C, VX_ACT_NONE, -1, p.stats, 0, pre_layer, pre_rep));
    if (C <= 512) {   // (round 5: the normalise + pool pass reduces the partials itself; no finalize launch)
      norm_stats_ = true; norm_tiles_ = ntiles;
      VX_STEP(kNorm[2 * l + 1], norm(p.B[l], C, p.CAT[l], 2 * C, C, p.P[l + 1], L, 2 * l + 1, 1, xblk_of(L.W)));
      norm_stats_ = false;
    } else {
    VX_STEP(kFin[2 * l + 1], vx_instnorm_finalize(p.stats, N, ntiles, C, L.nvox, 1e-5f, p.mean, p.rstd, stream));
    VX_STEP(kNorm[2 * l + 1], norm(p.B[l], C, p.CAT[l], 2 * C, C, p.P[l + 1], L, 2 * l + 1, 1, xblk_of(L.W)));
    }
  }
  // ---------------- center ----------------
  {
    const Level& L4 = p.lv[4];
    const int C3 = p.lv[3].C, C4 = L4.C;
    VX_STEP(kConv[8], conv(p.P[4], C3, 8, p.C0, C4, 0, L4, C3, C4, VX_ACT_RELU, -1, nullptr, 0));
    VX_STEP(kConv[9], conv(p.C0, C4, 9, p.C1, C4, 0, L4, C4, C4, VX_ACT_RELU, -1, nullptr, 0));
    VX_STEP(kUp[0], convT(p.C1, 0, p.CAT[3], 2 * C3, L4, C4, C3, VX_ACT_RELU, 8));
  }
  // ---------------- decoder ----------------
  for (int l = 3; l >= 0; --l) {
    const Level& L = p.lv[l];
    const int C = L.C;
    const int wi = 10 + 2 * (3 - l);
    const int dl = 9 + 2 * (3 - l);
    const float* up_in = l == 0 && fuse_up ? p.B[1] : nullptr;
    st16_ = (l == 0 && st16) ? 1 : 0;     // expand_1_1 leaves A_0 as fp16 (same buffer, half of it used)
    if (l == 0 && fuse0) {  // the skip half of CAT_0 is contr_1_2's raw output: normalise + LeakyReLU + dropout layer 1 on load
      products_ = (st16 && vx_cfg().storage16 == 2 && up_in && w->up_fused && !vx_cfg().s16_no_upcompose) ? 1 : 0;
      VX_STEP(up_in ? "upscale2+expand_1_1" : kConv[wi],
              conv(p.CAT[l], 2 * C, wi, p.A[l], C, 0, L, 2 * C, C, VX_ACT_LRELU, dl, nullptr, xblk_of(L.W), 1, 1,
                   p.mean0, p.rstd0, 0, 0, up_in, 3, 2 * C));
      products_ = 0;
    }
    else if (up_in)
      VX_STEP("upscale2+expand_1_1", conv(p.CAT[l], 2 * C, wi, p.A[l], C, 0, L, 2 * C, C, VX_ACT_LRELU, dl, nullptr,
                                          xblk_of(L.W), -1, 1, nullptr, nullptr, 0, 0, up_in, 3, 2 * C));
    else if (halves[l]) {
      // conv(cat([up, skip])) = conv_up(up) + conv_skip(skip) + bias, as two launches of the 16-channel z-column kernel over the two
      // DENSE halves: (1) the skip half, normalised on load (InstanceNorm + LeakyReLU + dropout layer 2 l + 1), + bias -> partial
      // sums in A_l; (2) the up half + the partial sums -> LeakyReLU -> dropout -> A_l in place.  The tile kernel's launch over the
      // x-blocked buffer paid 0.17 ms for the prologue in waves that also multiply (1.02 ms; the two launches: see DESIGN 5e).
      vx_conv3d_args a1 = {};
      a1.in = p.CAT[l] + (size_t)N * L.nvox * C; a1.w_packed = w->split_w[1]; a1.bias = w->conv_b[wi]; a1.out = p.A[l];
      a1.in_pitch = C; a1.out_pitch = C; a1.N = N; a1.D = L.D; a1.H = L.H; a1.W = L.W; a1.Cin = C; a1.Cout = C;
      a1.act = VX_ACT_NONE; a1.drop_mode = VX_DROP_NONE; a1.w_family = w->split_family;
      a1.in_mean = p.meanS[l]; a1.in_rstd = p.rstdS[l]; a1.in_drop_mode = dm; a1.in_drop_seed = r->seed; a1.in_drop_layer = (uint32_t)(2 * l + 1);
      a1.in_repeat = 1; a1.seed_dev = r->seed_dev; a1.out_half = 1;
      VX_STEP("expand_2_1(skip half)", vx_conv3d_k3(&a1, stream));
      vx_conv3d_args a2 = {};
      a2.in = p.CAT[l]; a2.w_packed = w->split_w[0]; a2.bias = w->conv_b[wi]; a2.out = p.A[l];
      a2.in_pitch = C; a2.out_pitch = C; a2.N = N; a2.D = L.D; a2.H = L.H; a2.W = L.W; a2.Cin = C; a2.Cout = C;
      a2.act = VX_ACT_LRELU; a2.drop_mode = dm; a2.drop_seed = r->seed; a2.drop_layer = (uint32_t)dl; a2.w_family = w->split_family;
      a2.acc_in = p.A[l]; a2.acc_pitch = C; a2.seed_dev = r->seed_dev; a2.range_flag = r->range_flag; a2.out_half = 1;
      if (fuse_up1) {   // upscale3 evaluated by this launch's staging waves from B_2 (the up tensor never exists)
        a2.in = p.B[l + 1]; a2.up_in = p.B[l + 1]; a2.up_pitch = 2 * C; a2.up_w = w->up3_zc16; a2.up_b = w->up_b[2];
        a2.up_split = usplit1_ ? 1 : 0;
        // Round 6: the block's second conv has no normalisation in front of it (unet3D_module.py:263-267), so this launch's epilogue
        // hands the tensor over as the fp16 (hi, lo) planes expand_2_2's matrix instructions take, in its LDS row order, and
        // expand_2_2 stages it by LDS-DMA.  The planar tensor goes into the up half of CAT_l, free since upscale3 lives inside
        // this launch (it cannot overwrite A_l in place: the partial sums there have the float layout).
        if (l == 1 && dm != VX_DROP_MASK && !vx_cfg().s16_no_l1dma && vx_conv3d_k3_planar_ok(L.D, L.H, L.W, C, C)) {
          a2.out = p.CAT[l]; a2.out_planar = 1;
          planar1 = true;
        }
      }
      VX_STEP(fuse_up1 ? "upscale3+expand_2_1(up half)" : "expand_2_1(up half)", vx_conv3d_k3(&a2, stream));
    }
    else if (skip_raw[l])   // the skip half of CAT_l is contr_l_2's raw output: InstanceNorm + LeakyReLU + dropout layer 2 l + 1 on load
      VX_STEP(kConv[wi], conv(p.CAT[l], 2 * C, wi, p.A[l], C, 0, L, 2 * C, C, VX_ACT_LRELU, dl, nullptr, xblk_of(L.W), 2 * l + 1, 1,
                              p.meanS[l], p.rstdS[l]));
    else
      VX_STEP(kConv[wi], conv(p.CAT[l], 2 * C, wi, p.A[l], C, 0, L, 2 * C, C, VX_ACT_LRELU, dl, nullptr, xblk_of(L.W)));
    st16_ = (l == 0 && st16) ? 2 : 0;
    // expand_2_2's output has ONE reader when upscale2 is fused into expand_1_1: hand it over pre-split
    osplit_ = l == 1 && fuse_up && dm != VX_DROP_MASK && !vx_cfg().s16_no_upsplit;   // (level 1 runs on the tile kernel)
    if (osplit_) usplit_ = true;
    // ... and expand_3_2's when upscale3 is evaluated inside expand_2_1's up-half launch (round 5)
    if (l == 2 && halves[1] && w->up3_zc16 && dm != VX_DROP_MASK && vx_conv3d_k3_upfuse_ok(p.lv[1].D, p.lv[1].H, p.lv[1].W, 16, 16) == 2) {
      fuse_up1 = true;
      if (!vx_cfg().s16_no_upsplit) { osplit_ = true; usplit1_ = true; }
    }
    in_planar_ = l == 1 && planar1;
    products_ = (l == 0 && st16 && vx_cfg().storage16 == 2) ? 1 : 0;
    VX_STEP(wi + 1 == 17 ? kLast : kConv[wi + 1], conv(in_planar_ ? p.CAT[l] : p.A[l], C, wi + 1, p.B[l], C, 0, L, C, C, VX_ACT_LRELU, dl + 1, nullptr, 0));
    in_planar_ = false;
    products_ = 0;
    st16_ = 0;
    osplit_ = false;
    if ((l > 1 || (l == 1 && !fuse_up)) && !(l == 2 && fuse_up1))
      VX_STEP(kUp[1 + (3 - l)], convT(p.B[l], 1 + (3 - l), p.CAT[l - 1], halves[l - 1] ? C / 2 : C, L, C, C / 2, VX_ACT_NONE, -1, halves[l - 1]));
  }
  // ---------------- head ----------------
  if (!fuse_head)
    VX_STEP("final", vx_conv1x1_ncdhw(p.B[0], F, w->final_w, w->final_b, r->logits, N, D, H, W, F, NC, r->dst, r->flip, stream));
  return VX_OK;
}

// Diagnostic entry for bench.py's roofline leg: runs the forward eagerly with a HIP event pair around every
// launch on `stream`, synchronises the stream, and returns per-launch milliseconds and labels.
extern "C" int vx_unet3d_forward_profiled(const vx_unet3d_weights* w, const vx_unet3d_run* r, vx_stream_t stream,
                                          int max_launches, float* ms, const char** labels, int* n_launches) {
  if (!ms || !labels || !n_launches) VX_FAIL(VX_E_NULL, "vx_unet3d_forward_profiled: null output");
  Prof pf;
  pf.s = (hipStream_t)stream;
  for (int i = 0; i < 2 * Prof::MAXL; ++i) {
    hipError_t e = hipEventCreate(&pf.ev[i]);
    if (e != hipSuccess) VX_FAIL((int)e, "hipEventCreate: %s", hipGetErrorString(e));
  }
  g_prof = &pf;
  int rc = vx_unet3d_forward(w, r, stream);
  g_prof = nullptr;
  if (rc == VX_OK) {
    hipError_t e = hipStreamSynchronize(pf.s);
    if (e != hipSuccess) { vx_set_error("hipStreamSynchronize: %s", hipGetErrorString(e)); rc = (int)e; }
  }
  int n = 0;
  if (rc == VX_OK) {
    for (; n < pf.n && n < max_launches; ++n) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, pf.ev[2 * n], pf.ev[2 * n + 1]);
      ms[n] = t;
      snprintf(g_prof_labels[n], sizeof(g_prof_labels[n]), "%s|%s", pf.name[n], pf.kernel[n] ? pf.kernel[n] : "?");
      labels[n] = g_prof_labels[n];
    }
  }
  *n_launches = n;
  for (int i = 0; i < 2 * Prof::MAXL; ++i) (void)hipEventDestroy(pf.ev[i]);
  return rc;
}
