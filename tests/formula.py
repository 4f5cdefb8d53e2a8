"""Closed-form, platform-independent tensor generators.

Golden fixtures (tests/golden/*.npz) store only the reference's OUTPUTS; the
inputs and the network weights that produced them are regenerated from these
integer-hash formulas, identically in tools/gen_golden.py (run once, in the
container that has /root/reference) and in the tests (run anywhere).  All
arithmetic is exact in uint64/float64, so the values do not depend on libm.

Test infrastructure (round 3: moved out of the product package): users are tests/, tools/ (gen_golden.py, the fuzzers)
and __graft_entry__.smoke().  Nothing in values_amd/ and nothing bench.py times imports it.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def hash_uniform(n: int, tag: int) -> np.ndarray:
    """n float64 values in [-1, 1), a pure function of (index, tag)."""
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(tag) * np.uint64(40503) + np.uint64(12345)) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x45D9F3B)) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x45D9F3B)) & _M32
    h ^= h >> np.uint64(16)
    return h.astype(np.float64) / 2147483648.0 - 1.0


def formula_tensor(shape, tag: int, scale: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape))
    return (hash_uniform(n, tag) * scale).reshape(shape)


def unet3d_param_shapes(num_classes=2, in_channels=1, f=8, aleatoric_loss=False):
    """State-dict names/shapes of the reference UNet3D
    (uncertainty_modeling/models/unet3D_module.py:36-210)."""
    shapes = OrderedDict()

    def conv(name, cin, cout, k=3):
        shapes[name + ".weight"] = (cout, cin, k, k, k)
        shapes[name + ".bias"] = (cout,)

    def convT(name, cin, cout):
        shapes[name + ".weight"] = (cin, cout, 2, 2, 2)
        shapes[name + ".bias"] = (cout,)

    conv("contr_1_1.0", in_channels, f)
    conv("contr_1_2.0", f, f)
    conv("contr_2_1.0", f, 2 * f)
    conv("contr_2_2.0", 2 * f, 2 * f)
    conv("contr_3_1.0", 2 * f, 4 * f)
    conv("contr_3_2.0", 4 * f, 4 * f)
    conv("contr_4_1.0", 4 * f, 8 * f)
    conv("contr_4_2.0", 8 * f, 8 * f)
    conv("center.0", 8 * f, 16 * f)
    conv("center.2", 16 * f, 16 * f)
    convT("center.4", 16 * f, 8 * f)
    conv("expand_4_1.0", 16 * f, 8 * f)
    conv("expand_4_2.0", 8 * f, 8 * f)
    convT("upscale4", 8 * f, 4 * f)
    conv("expand_3_1.0", 8 * f, 4 * f)
    conv("expand_3_2.0", 4 * f, 4 * f)
    convT("upscale3", 4 * f, 2 * f)
    conv("expand_2_1.0", 4 * f, 2 * f)
    conv("expand_2_2.0", 2 * f, 2 * f)
    convT("upscale2", 2 * f, f)
    conv("expand_1_1.0", 2 * f, f)
    conv("expand_1_2.0", f, f)
    conv("final", f, num_classes, k=1)
    if aleatoric_loss:
        conv("final_aleatoric", f, 2 * num_classes, k=1)
    conv("output_reconstruction_map", f, 1, k=1)
    return shapes


def formula_unet3d_state_dict(seed_tag: int = 0, **kw) -> "OrderedDict[str, np.ndarray]":
    """float32-representable weights with the magnitude of torch's default init
    (uniform(+-1/sqrt(fan_in))), so activations stay O(1) through the net."""
    sd = OrderedDict()
    for idx, (name, shape) in enumerate(unet3d_param_shapes(**kw).items()):
        if name.endswith(".weight"):
            if "upscale" in name or name == "center.4.weight":
                fan_in = shape[1] * 8  # ConvTranspose3d: weight (Cin, Cout, 2,2,2); torch uses size(1)*k^3
            else:
                fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            last_bound = bound
        else:
            bound = last_bound
        w = formula_tensor(shape, tag=1000 * (seed_tag + 1) + idx, scale=bound)
        sd[name] = w.astype(np.float32).astype(np.float64)
    return sd


def formula_ssn_state_dict(num_classes: int = 2, rank: int = 10, seed_tag: int = 0, f: int = 8):
    """Weights for the reference SsnUNet3D (ssn_unet3D_module.py:7-37): the parent UNet3D built with
    num_classes * (2 + rank) outputs, plus the mean / log_cov_diag / cov_factor 1x1x1 heads."""
    sd = formula_unet3d_state_dict(seed_tag=seed_tag, num_classes=num_classes * 2 + num_classes * rank, f=f)
    bound = 1.0 / math.sqrt(f)
    for i, (name, cout) in enumerate((("mean_conv", num_classes), ("log_cov_diag_conv", num_classes),
                                      ("cov_factor_conv", num_classes * rank))):
        for j, (suffix, shape) in enumerate(((".weight", (cout, f, 1, 1, 1)), (".bias", (cout,)))):
            w = formula_tensor(shape, tag=5000 * (seed_tag + 1) + 10 * i + j, scale=bound)
            sd[name + suffix] = w.astype(np.float32).astype(np.float64)
    return sd


def formula_volume(shape, tag: int = 7) -> np.ndarray:
    """Smooth-ish z-scored test volume: a few low-frequency bumps + hash noise."""
    shape = tuple(shape)
    sp = shape[-3:]
    z, y, x = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in sp], indexing="ij")
    r2 = ((z - sp[0] * 0.4) ** 2 + (y - sp[1] * 0.55) ** 2 + (x - sp[2] * 0.5) ** 2) / (0.08 * sum(s * s for s in sp))
    blob = 2.0 / (1.0 + 4.0 * r2)  # rational bump: exact ops only (no libm)
    vol = blob + 0.6 * hash_uniform(int(np.prod(sp)), tag).reshape(sp)
    vol = vol - vol.mean()
    vol = vol / np.sqrt((vol * vol).mean())
    out = np.broadcast_to(vol, shape).copy()
    return out.astype(np.float32).astype(np.float64)


def _name_tag(name: str) -> int:
    import zlib
    return zlib.crc32(name.encode()) & 0xFFFFF


def formula_state_dict_from_shapes(shapes, seed_tag: int = 0):
    """Weights for an arbitrary conv/BN network (the HRNet fixtures): a pure function of (parameter NAME, shape), so
    it does not depend on state-dict order.  Conv weights ~ U(+-sqrt(3/fan_in)), BN gamma in [0.7, 1.3], BN beta and
    conv biases in [-0.2, 0.2].  Buffers (running_mean/var, num_batches_tracked) are left at their defaults."""
    sd = OrderedDict()
    for name, shape in shapes.items():
        shape = tuple(shape)
        if name.endswith("running_mean") or name.endswith("running_var") or name.endswith("num_batches_tracked"):
            continue
        tag = _name_tag(name) + 7919 * seed_tag
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            w = formula_tensor(shape, tag, scale=math.sqrt(3.0 / fan_in))
        elif name.endswith(".weight"):
            w = 1.0 + formula_tensor(shape, tag, scale=0.3)
        else:
            w = formula_tensor(shape, tag, scale=0.2)
        sd[name] = w.astype(np.float32).astype(np.float64)
    return sd


# reduced HRNet layouts of the golden fixtures (tools/gen_golden.py); the shipped layouts live in values_amd.hrnet_configs
HRNET_SMALL_EXTRA = {
    "DROPOUT_FINAL": True, "FINAL_CONV_KERNEL": 1,
    "STAGE1": {"NUM_MODULES": 1, "NUM_BRANCHES": 1, "BLOCK": "BOTTLENECK", "NUM_BLOCKS": [2], "NUM_CHANNELS": [32],
               "FUSE_METHOD": "SUM"},
    "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "BLOCK": "BASIC", "NUM_BLOCKS": [2, 2], "NUM_CHANNELS": [16, 32],
               "FUSE_METHOD": "SUM"},
    "STAGE3": {"NUM_MODULES": 2, "NUM_BRANCHES": 3, "BLOCK": "BASIC", "NUM_BLOCKS": [2, 2, 2],
               "NUM_CHANNELS": [16, 32, 64], "FUSE_METHOD": "SUM"},
    "STAGE4": {"NUM_MODULES": 1, "NUM_BRANCHES": 4, "BLOCK": "BASIC", "NUM_BLOCKS": [2, 2, 2, 2],
               "NUM_CHANNELS": [16, 32, 64, 128], "FUSE_METHOD": "SUM"},
}


HRNET_W18S_EXTRA = {   # HRNet-W18 widths (BASELINE config 4) with fewer blocks / modules: none is a multiple of 16
    "DROPOUT_FINAL": True, "FINAL_CONV_KERNEL": 1,
    "STAGE1": {"NUM_MODULES": 1, "NUM_BRANCHES": 1, "BLOCK": "BOTTLENECK", "NUM_BLOCKS": [1], "NUM_CHANNELS": [32],
               "FUSE_METHOD": "SUM"},
    "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "BLOCK": "BASIC", "NUM_BLOCKS": [1, 1], "NUM_CHANNELS": [18, 36],
               "FUSE_METHOD": "SUM"},
    "STAGE3": {"NUM_MODULES": 1, "NUM_BRANCHES": 3, "BLOCK": "BASIC", "NUM_BLOCKS": [1, 1, 1],
               "NUM_CHANNELS": [18, 36, 72], "FUSE_METHOD": "SUM"},
    "STAGE4": {"NUM_MODULES": 1, "NUM_BRANCHES": 4, "BLOCK": "BASIC", "NUM_BLOCKS": [1, 1, 1, 1],
               "NUM_CHANNELS": [18, 36, 72, 144], "FUSE_METHOD": "SUM"},
}
