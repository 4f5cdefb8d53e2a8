// libvalues_amd.so: version + thread-local error string.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/values_amd.h"

static thread_local char g_err[512] = "";

void vx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local const char* g_last_kernel = nullptr;
void vx_note_kernel(const char* name) { g_last_kernel = name; }
const char* vx_last_kernel() { return g_last_kernel; }
const char* vx_kname(const char* fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  char* p = (char*)malloc(strlen(buf) + 1);   // one per launcher instantiation, never freed
  strcpy(p, buf);
  return p;
}

// ---- configuration: read from the environment exactly once, replaced only by vx_set_config ----
#include <mutex>
#include <stdlib.h>
#include <string.h>

static vx_config g_cfg;
static std::once_flag g_cfg_once;

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  char* end = nullptr;
  const long v = strtol(e, &end, 0);
  return end == e ? 1 : (int)v;   // VX_S16_NO_DB=yes style switches count as 1
}

static void cfg_from_env() {
  memset(&g_cfg, 0, sizeof(g_cfg));
  g_cfg.conv_fp32 = env_int("VX_CONV_FP32", 0);
  g_cfg.s16_no_prenorm = env_int("VX_S16_NO_PRENORM", 0);
  g_cfg.s16_no_xp8 = env_int("VX_S16_NO_XP8", 0);
  g_cfg.s16_skip_raw = env_int("VX_S16_SKIP_RAW", 1);
  g_cfg.no_head_fusion = env_int("VX_NO_HEAD_FUSION", 0);
  g_cfg.s16_no_upfuse = env_int("VX_S16_NO_UPFUSE", 0);
  g_cfg.s16_no_poolfuse = env_int("VX_S16_NO_POOLFUSE", 0);
  g_cfg.s16_no_presplit = env_int("VX_S16_NO_PRESPLIT", 0);
  g_cfg.storage16 = env_int("VX_STORAGE16", 0);
  g_cfg.s16_no_dbplain = env_int("VX_S16_NO_DBPLAIN", 0);
  g_cfg.s16_generic = env_int("VX_S16_GENERIC", 0);
  g_cfg.s16_no_upcompose = env_int("VX_S16_NO_UPCOMPOSE", 0);
  g_cfg.s16_no_upsplit = env_int("VX_S16_NO_UPSPLIT", 0);
  g_cfg.s16_no_poolfin = env_int("VX_S16_NO_POOLFIN", 0);
  g_cfg.s16_no_zc16 = env_int("VX_S16_NO_ZC16", 0);
  g_cfg.s16_no_halves = env_int("VX_S16_NO_HALVES", 0);
  g_cfg.s16_no_deep = env_int("VX_S16_NO_DEEP", 0);
  g_cfg.s16_no_l1dma = env_int("VX_S16_NO_L1DMA", 0);
  g_cfg.c2s_no_wide = env_int("VX_C2S_NO_WIDE", 0);
  g_cfg.c2s_no_oct = env_int("VX_C2S_NO_OCT", 0);
}

const vx_config& vx_cfg() {
  std::call_once(g_cfg_once, cfg_from_env);
  return g_cfg;
}

extern "C" int vx_get_config(vx_config* out) {
  if (!out) { vx_set_error("vx_get_config: null"); return VX_E_NULL; }
  *out = vx_cfg();
  return VX_OK;
}

extern "C" int vx_set_config(const vx_config* cfg) {
  if (!cfg) { vx_set_error("vx_set_config: null"); return VX_E_NULL; }
  if (cfg->conv_fp32 < 0 || cfg->conv_fp32 > 2) { vx_set_error("vx_set_config: conv_fp32 %d", cfg->conv_fp32); return VX_E_DTYPE; }
  (void)vx_cfg();
  g_cfg = *cfg;
  return VX_OK;
}

extern "C" const char* vx_last_kernel_name(void) { return g_last_kernel ? g_last_kernel : ""; }
extern "C" int vx_version(void) { return 600; /* 0.6.0 (round 6): vx_config.s16_no_l1dma, vx_conv3d_args.out_planar / in_planar, vx_conv3d_k3_planar_ok; the dropout generator's key is two words (other masks for a given seed than 0.5.x); 0.5.1 (round 5, second half): vx_config.s16_no_deep, kernel family 7 (Cout % 32 == 0, Cin >= 16: the tile kernel's fragments + the deep-layer kernel's), vx_convT_k2s2 on split-fp16 products for Cin in {64, 128}, vx_stat_src + vx_norm_act_drop_pool_stats / vx_pool_finish_z_stats / vx_prenorm_split_stats; 0.5.0 (round 5): vx_config.s16_no_zc16 / s16_no_halves, vx_conv3d_args.acc_in, vx_unet3d_weights.split_w, vx_pool_finish_z, vx_conv3d_k3_pool_layout, kernel family 6 (Cout = 16, Cin in {8, 16}: the tile kernel's fragments + the z-column kernel's); 0.4.0 (round 4): vx_config lost conv_no_c8, conv_no_xcd, conv_per_cu, s16_per_cu, c8_per_cu, convt_wgs, s16_no_xp, s16_no_db, s16_no_db3, s16_no_epi, s16_no_ty8, s16_no_wall, c2s_no_nt5, convt_no_mfma, s16_range_check, s16_pw, s16_prio (measured-slower variants and tuning knobs, with their instances); + vx_conv3d_k3_presplit_ok; 0.3.3: vx_config.c2s_no_oct, vx_bilinear_softmax_nchw; 0.3.2: vx_config.c2s_no_wide; 0.3.1: vx_prenorm_split, vx_conv3d_args.in_split, vx_config.s16_no_presplit; 0.3.0: vx_config lost conv_dma, s16_ping, s16_dbg, c8_dbg, dma_dbg, dma_nw16, c8_tile16, s16_no_wspec (round 3) */ }
extern "C" const char* vx_last_error_string(void) { return g_err; }
