"""ctypes binding of libvalues_amd.so (include/values_amd.h).

The HIP library is the product: there is NO CPU or PyTorch fallback.  If the shared object is
missing or a symbol is absent, import of the compute entry points fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VX_LIB_PATH") or os.path.join(_HERE, "libvalues_amd.so")   # VX_LIB_PATH: A/B builds only

VX_F32, VX_F64 = 0, 1
VX_ACT_NONE, VX_ACT_LRELU, VX_ACT_RELU = 0, 1, 2
VX_DROP_NONE, VX_DROP_HASH, VX_DROP_MASK = 0, 1, 2

_p = C.c_void_p
_i = C.c_int
_i32 = C.c_int32
_u32 = C.c_uint32
_i64 = C.c_int64


class ConvArgs(C.Structure):
    _fields_ = [("in_", _p), ("w_packed", _p), ("bias", _p), ("out", _p),
                ("in_pitch", _i32), ("out_pitch", _i32), ("out_coff", _i32),
                ("N", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("Cin", _i32), ("Cout", _i32),
                ("act", _i32), ("drop_mode", _i32), ("drop_seed", _u32), ("drop_layer", _u32),
                ("drop_mask", _p), ("stats_partial", _p), ("in_xblk", _i32), ("w_family", _i32),
                ("head_out", _p), ("head_w", _p), ("head_b", _p), ("head_dst", _p), ("head_flip", _p),
                ("head_C", _i32),
                ("in_mean", _p), ("in_rstd", _p), ("in_drop_mode", _i32), ("in_drop_seed", _u32), ("in_drop_layer", _u32),
                ("in_repeat", _i32), ("out_xblk", _i32), ("out_half", _i32), ("range_flag", _p), ("seed_dev", _p),
                ("up_in", _p), ("up_w", _p), ("up_b", _p), ("up_pitch", _i32), ("pool_out", _p), ("pool_flags", _p),
                ("in_split", _i32), ("out_f16", _i32), ("in_f16", _i32),
                ("out_split", _i32), ("up_split", _i32), ("up_fused", _p), ("in_pool_flags", _p),
                ("acc_in", _p), ("acc_pitch", _i32), ("out_planar", _i32), ("in_planar", _i32), ("products", _i32)]


class NormArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i32), ("mean", _p), ("rstd", _p),
                ("out", _p), ("out_pitch", _i32), ("out_coff", _i32),
                ("pool_out", _p), ("pool_pitch", _i32),
                ("N", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("C", _i32),
                ("act", _i32), ("drop_mode", _i32), ("drop_seed", _u32), ("drop_layer", _u32), ("drop_mask", _p),
                ("out_xblk", _i32), ("out_half", _i32), ("x_xblk", _i32), ("x_half", _i32), ("seed_dev", _p),
                ("range_flag", _p)]


class StatSrc(C.Structure):
    """vx_stat_src: a streaming pass reduces the producing conv's statistics partials itself (round 5)."""
    _fields_ = [("stats_partial", _p), ("tiles", _i32), ("eps", C.c_float), ("count", _i64), ("mean_out", _p), ("rstd_out", _p)]


class ConvTArgs(C.Structure):
    _fields_ = [("in_", _p), ("in_pitch", _i32), ("w_packed", _p), ("bias", _p),
                ("out", _p), ("out_pitch", _i32), ("out_coff", _i32),
                ("N", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("Cin", _i32), ("Cout", _i32),
                ("act", _i32), ("drop_mode", _i32), ("drop_seed", _u32), ("drop_layer", _u32), ("drop_mask", _p),
                ("out_xblk", _i32), ("out_half", _i32), ("range_flag", _p), ("seed_dev", _p)]


class Conv2dArgs(C.Structure):
    _fields_ = [("in_", _p), ("in_pitch", _i32), ("w_packed", _p), ("bias", _p),
                ("out", _p), ("out_pitch", _i32), ("out_coff", _i32),
                ("N", _i32), ("H", _i32), ("W", _i32), ("Cin", _i32), ("Cout", _i32), ("KS", _i32), ("S", _i32),
                ("stats_partial", _p), ("w_family", _i32),
                ("in_scale", _p), ("in_shift", _p), ("in_cpitch", _i32), ("in_group_images", _i32), ("in_relu", _i32)]


class AffineArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i32), ("scale", _p), ("shift", _p), ("add", _p), ("add_pitch", _i32),
                ("out", _p), ("out_pitch", _i32), ("out_coff", _i32),
                ("N", _i32), ("H", _i32), ("W", _i32), ("C", _i32), ("OH", _i32), ("OW", _i32),
                ("act", _i32), ("drop_mode", _i32), ("drop_seed", _u32), ("drop_layer", _u32), ("drop_mask", _p),
                ("group_images", _i32)]


class FuseTerm(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i32), ("H", _i32), ("W", _i32), ("scale", _p), ("shift", _p)]


class FuseArgs(C.Structure):
    _fields_ = [("term", FuseTerm * 4), ("nterms", _i32), ("out", _p), ("out_pitch", _i32),
                ("N", _i32), ("OH", _i32), ("OW", _i32), ("C", _i32), ("act", _i32), ("group_images", _i32)]


class Config(C.Structure):
    """vx_config (include/values_amd.h): kernel-family selection and tuning knobs, read once from VX_* variables."""
    _fields_ = [(n, _i32) for n in (
        "conv_fp32", "s16_no_prenorm", "s16_no_xp8", "s16_skip_raw", "no_head_fusion", "s16_no_upfuse", "s16_no_poolfuse",
        "storage16", "s16_no_dbplain", "s16_generic", "s16_no_upcompose", "s16_no_upsplit", "s16_no_presplit", "s16_no_poolfin", "s16_no_zc16", "s16_no_halves", "s16_no_deep", "s16_no_l1dma", "c2s_no_wide", "c2s_no_oct")]


class UncOutputs(C.Structure):
    _fields_ = [("mean_prob", _p), ("pred_entropy", _p), ("exp_entropy", _p), ("mutual_info", _p), ("variance", _p),
                ("argmax", _p), ("sample_argmax", _p), ("in_count", _p), ("out_count", _p)]


class UNet3DWeights(C.Structure):
    _fields_ = [("conv_w", _p * 18), ("conv_b", _p * 18), ("up_w", _p * 4), ("up_b", _p * 4),
                ("final_w", _p), ("final_b", _p), ("F", _i32), ("num_classes", _i32), ("conv_family", _i32 * 18),
                ("in_channels", _i32), ("no_instancenorm", _i32), ("up_fused", _p), ("split_w", _p * 2), ("split_family", _i32), ("up3_zc16", _p)]


class UNet3DRun(C.Structure):
    _fields_ = [("x", _p), ("N", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("repeat", _i32),
                ("src", _p), ("flip", _p), ("dst", _p), ("drop_mode", _i32), ("seed", _u32),
                ("masks", _p * 17), ("logits", _p), ("workspace", _p), ("workspace_bytes", C.c_size_t),
                ("range_flag", _p), ("seed_dev", _p)]


# symbol -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "vx_version": (_i, []),
    "vx_last_error_string": (C.c_char_p, []),
    "vx_last_kernel_name": (C.c_char_p, []),
    "vx_drop_hash_mask": (_i, [_u32, _u32, _i, _i64, _p, _p]),
    "vx_get_config": (_i, [C.POINTER(Config)]),
    "vx_set_config": (_i, [C.POINTER(Config)]),
    "vx_conv3d_k3_family": (_i, [_i, _i]),
    "vx_conv2d_family": (_i, [_i, _i, _i]),
    "vx_unc_reduce": (_i, [_p, _i, _i, _i, _i, _i, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "vx_unc_reduce_ex": (_i, [_p, _i, _i, _i, _i, _i, _i64, C.POINTER(UncOutputs), _p]),
    "vx_unc_stats_accumulate": (_i, [_p, _i, _i, _i, _i64, _p, _p]),
    "vx_unc_stats_finalize": (_i, [_p, _i, _i, _i, _i64, _p, _p, _p, _p, _p, _p]),
    "vx_softmax_planar": (_i, [_p, _i64, _i, _i64, _p, _p]),
    "vx_one_minus_msr": (_i, [_p, _i, _i, _i64, _p, _p]),
    "vx_conv3d_k3_packed_floats": (_i64, [_i, _i]),
    "vx_pack_conv3d_k3": (_i, [_p, _p, _i, _i, _p]),
    "vx_convT_k2s2_packed_floats": (_i64, [_i, _i]),
    "vx_pack_convT_k2s2": (_i, [_p, _p, _i, _i, _p]),
    "vx_conv3d_k3_tiles": (_i, [_i, _i, _i]),
    "vx_conv3d_k3_tiles_for": (_i, [_i, _i, _i, _i]),
    "vx_conv3d_k3_head_fusable": (_i, [_i, _i]),
    "vx_conv3d_k3_prologue_ok": (_i, [_i, _i, _i, _i, _i]),
    "vx_conv3d_k3_upfuse_ok": (_i, [_i, _i, _i, _i, _i]),
    "vx_conv3d_k3_poolfuse_ok": (_i, [_i, _i, _i, _i, _i]),
    "vx_conv3d_k3_presplit_ok": (_i, [_i, _i, _i, _i, _i]),
    "vx_conv3d_k3_poolfin_ok": (_i, [_i, _i]),
    "vx_conv3d_upfused_packed_floats": (_i64, []),
    "vx_pack_conv3d_upfused": (_i, [_p, _p, _p, _p, _p, _p]),
    "vx_pool_finish": (_i, [_p, _p, _p, _p, _p, _i, _i, _i64, _i, _p]),
    "vx_pool_finish_z": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i64, _i, _p]),
    "vx_conv3d_k3_pool_layout": (_i, [_i, _i, _i, _i, _i]),
    "vx_conv3d_k3_skip_prologue_ok": (_i, [_i, _i, _i, _i, _i, _i]),
    "vx_conv3d_k3_acc_ok": (_i, [_i, _i, _i, _i, _i]),
    "vx_conv3d_k3_planar_ok": (_i, [_i, _i, _i, _i, _i]),
    "vx_convT_zc16_packed_floats": (_i64, []),
    "vx_pack_convT_zc16": (_i, [_p, _p, _p]),
    "vx_prenorm_split": (_i, [_p, _p, _p, _i, _i64, C.c_float, _p]),
    "vx_zero": (_i, [_p, _i64, _p]),
    "vx_conv3d_k3": (_i, [C.POINTER(ConvArgs), _p]),
    "vx_conv3d_k3_c1_tiles": (_i, [_i, _i, _i]),
    "vx_conv3d_k3_c1": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "vx_pack_input_cl8": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "vx_instnorm_finalize": (_i, [_p, _i, _i, _i, _i64, C.c_float, _p, _p, _p]),
    "vx_norm_act_drop_pool": (_i, [C.POINTER(NormArgs), _p]),
    "vx_norm_act_drop_pool_bcast": (_i, [C.POINTER(NormArgs), _i, _p]),
    "vx_norm_act_drop_pool_stats": (_i, [C.POINTER(NormArgs), C.POINTER(StatSrc), _p]),
    "vx_prenorm_split_stats": (_i, [_p, C.POINTER(StatSrc), _i, _i64, C.c_float, _p]),
    "vx_pool_finish_z_stats": (_i, [_p, _p, C.POINTER(StatSrc), _p, _i, _i, _i, _i64, _i, _p]),
    "vx_convT_k2s2": (_i, [C.POINTER(ConvTArgs), _p]),
    "vx_conv1x1_ncdhw": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "vx_unet3d_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "vx_unet3d_forward": (_i, [C.POINTER(UNet3DWeights), C.POINTER(UNet3DRun), _p]),
    "vx_unet3d_forward_profiled": (_i, [C.POINTER(UNet3DWeights), C.POINTER(UNet3DRun), _p, _i, C.POINTER(C.c_float),
                                        C.POINTER(C.c_char_p), C.POINTER(_i)]),
    "vx_softmax_accumulate": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vx_aleatoric_sample": (_i, [_p, _p, _u32, _i, _i, _i, _i64, _p, _p, _p]),
    "vx_colorize_u8": (_i, [_p, _p, _i64, _p, _i, _p, _p]),
    "vx_fuse_sum": (_i, [C.POINTER(FuseArgs), _p]),
    "vx_tta_views_2d": (_i, [_p, _i, _p, _p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, _i, _i, _i, _i,
                             C.POINTER(C.c_int32), _p, _p]),
    "vx_select_workspace_bytes": (_i64, []),
    "vx_select_kth": (_i, [_p, _i64, _i64, _p, _p, _p]),
    "vx_count_nonzero_u8": (_i, [_p, _i64, _p, _p]),
    "vx_mask_agreement": (_i, [_p, _i, _i, _i64, _p, _p]),
    "vx_soft_metric_workspace_bytes": (_i64, [_i, _i]),
    "vx_soft_metric_sums": (_i, [_p, _p, _i, _i, _i64, _p, _p, _p]),
    "vx_ssn2d_lowres": (_i, [_p, _i, _p, _i, _p, _u32, _i, _i64, _i, _i, _i, _p, _p, _p]),
    "vx_ssn2d_add_diag": (_i, [_p, _p, _p, _u32, _i, _i, _i64, C.c_float, _p]),
    "vx_softmax_variance": (_i, [_p, _i, _i, _i, _i, _i64, _p, _p]),
    "vx_ssn_sample": (_i, [_p, _p, _p, _u32, _i, _i, _i, _i, _i64, C.c_float, _p, _p]),
    "vx_conv2d_packed_floats": (_i64, [_i, _i, _i]),
    "vx_pack_conv2d": (_i, [_p, _p, _i, _i, _i, _p]),
    "vx_conv2d_tiles": (_i, [_i, _i, _i, _i]),
    "vx_conv2d": (_i, [C.POINTER(Conv2dArgs), _p]),
    "vx_bn_finalize": (_i, [_p, _i, _i, _i64, C.c_float, _p, _p, _p, _p, _p]),
    "vx_bn_finalize_groups": (_i, [_p, _i, _i, _i, _i, _i64, C.c_float, _p, _p, _p, _p, _p]),
    "vx_affine_gather": (_i, [C.POINTER(AffineArgs), _p]),
    "vx_bilinear_nchw": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "vx_bilinear_softmax_nchw": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "vx_evalmetrics_workspace_bytes": (_i64, []),
    "vx_ncc_sums": (_i, [_p, _i, _p, _i, _i64, _i, C.c_double, C.c_double, _p, _p, _p]),
    "vx_platt_sums": (_i, [_p, _i, _p, _p, _i, _i64, _i, C.c_double, C.c_double, C.c_double, C.c_double, _p, _p, _p]),
    "vx_calib_bins": (_i, [_p, _i, _p, _p, _i, _i64, _i, C.c_double, C.c_double, C.POINTER(C.c_double), _p, _p, _p]),
    "vx_box_max": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, C.c_size_t, _p]),
    "vx_sum_thr": (_i, [_p, _i, _i64, C.c_double, _p, _p]),
}

_lib = None


class VxError(RuntimeError):
    pass


def load():
    """Load libvalues_amd.so (once).  import torch first: torch ships the HIP runtime
    (libamdhip64.so.7) this library resolves against, so both share one runtime instance."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VxError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C values_amd/csrc`). values_amd has no CPU fallback.")
    import torch  # noqa: F401  (loads the HIP runtime)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is absent
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def get_config() -> "Config":
    c = Config()
    check(load().vx_get_config(C.byref(c)), "vx_get_config")
    return c


class config:
    """`with _lib.config(conv_fp32=1): ...` -- run a block under a changed vx_config (tests and A/B tools; the library
    reads its VX_* environment variables only once, at first use).  Restores the previous configuration on exit."""

    def __init__(self, **fields):
        self.fields = fields

    def __enter__(self):
        self.saved = get_config()
        c = get_config()
        for k, v in self.fields.items():
            if not hasattr(c, k):
                raise AttributeError(f"vx_config has no field {k!r}")
            setattr(c, k, int(v))
        check(load().vx_set_config(C.byref(c)), "vx_set_config")
        return c

    def __exit__(self, *exc):
        check(load().vx_set_config(C.byref(self.saved)), "vx_set_config")
        return False


def pack_mode():
    """The configuration fields that select the kernel family and with it the PACKED WEIGHT LAYOUT (conv_config() in
    conv3d_mfma.hip, s16_config() in conv3d_s16.hip, c2_split16() in conv2d_mfma.hip).  Models key their packed-weight
    cache on it, and every launch carries the family its weights were packed for (w_family): the library refuses a
    mismatch."""
    c = get_config()
    return (c.conv_fp32, c.c2s_no_oct)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().vx_last_error_string()
        raise VxError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise VxError("values_amd needs a ROCm device (torch.cuda.is_available() is False); there is no CPU fallback")


def zeros(shape, dtype=None, device=None):
    """torch.empty + vx_zero on the current stream: a zero tensor without an ATen fill kernel (the persistent zero-padded
    buffers of the 2D walk are created through this, once per geometry)"""
    import torch
    require_gpu()
    t = torch.empty(shape, dtype=dtype or torch.float32, device=device)
    check(load().vx_zero(ptr(t), t.numel() * t.element_size(), stream_ptr()), "vx_zero")
    return t


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
