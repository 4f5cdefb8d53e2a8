"""Host-side input harness of the hot path (integer / byte work, no arithmetic kernels):

  * load_patch      DataCarrier3D.load_image (uncertainty_modeling/data_carrier_3D.py:59-97): memory-mapped .npy crop
  * tta_views_2d    the TTA branch of Cityscapes_dataset.__getitem__ (data/cityscapes_dataset.py:76-99): the four
                    views [img, HFlip(img), Noise(img), Noise(HFlip(img))] + the transform names test_2D.py uses to
                    un-flip (:304-309).  albumentations' GaussNoise (var_limit (10, 50), per-image RNG) is third-party
                    and absent: the noise field is an INPUT here (parity unpinned for the two noisy views; the flip
                    views are exact).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

TTA_2D_TRANSFORMS = [[], ["HorizontalFlip"], ["GaussNoise"], ["HorizontalFlip", "GaussNoise"]]


def load_patch(sample: Dict) -> Dict:
    """sample: {"image_path", "label_paths", "crop_idx": ((x0,x1),(y0,y1),(z0,z1))} as produced by
    get_val_test_data_samples -> the dict load_image returns (data (1,P,P,P), seg (R,1,P,P,P) int32)."""
    c = sample["crop_idx"]
    sl = tuple(slice(a, b) for a, b in c)
    arr = np.load(sample["image_path"], mmap_mode="r")
    out = {"image_paths": [sample["image_path"]], "label_paths": [sample["label_paths"]], "crop_idx": [c],
           "org_image_size": [arr.shape], "data": np.expand_dims(arr[sl], 0)}
    if sample["label_paths"] is not None:
        segs = np.array([np.load(p, mmap_mode="r")[sl] for p in sample["label_paths"]], dtype=np.intc)
        out["seg"] = np.expand_dims(segs, 1)
    return out


def tta_views_2d(img: np.ndarray, mean: Sequence[float], std: Sequence[float], noise: Optional[np.ndarray] = None,
                 noise_flipped: Optional[np.ndarray] = None, max_pixel_value: float = 255.0):
    """img: (H, W, 3) uint8.  Returns (views: list of 4 float32 arrays (3, H, W), transforms: list of name lists).
    Normalisation = albumentations.Normalize(mean, std, max_pixel_value) followed by ToTensorV2 (HWC -> CHW).
    noise / noise_flipped: additive fields (H, W, 3) applied to the uint8 image and its flip (clipped to 0..255 like
    albumentations does for uint8); None -> zeros (the view degenerates to the clean one)."""
    img = np.asarray(img)
    flipped = img[:, ::-1]

    def noisy(a, n):
        if n is None:
            return a
        return np.clip(a.astype(np.float32) + n, 0, 255).astype(np.uint8)

    raw = [img, flipped, noisy(img, noise), noisy(flipped, noise_flipped)]
    mean = np.asarray(mean, dtype=np.float32) * max_pixel_value
    inv = 1.0 / (np.asarray(std, dtype=np.float32) * max_pixel_value)
    views = [np.ascontiguousarray(((v.astype(np.float32) - mean) * inv).transpose(2, 0, 1)) for v in raw]
    return views, [list(t) for t in TTA_2D_TRANSFORMS]


# view codes of vx_tta_views_2d (include/values_amd.h): bit 0 HorizontalFlip, bit 1 VerticalFlip, bit 2 noisy,
# bit 3 noise field indexed at the source pixel, bit 4 noise slot
TTA_2D_VIEW_CODES = [0, 1, 4, 1 | 4 | 16]                      # the reference's four views, noise drawn after the flip
TTA_8_VIEW_CODES = [0, 1, 2, 3, 4 | 8, 5 | 8, 6 | 8, 7 | 8]    # config C4: {id, H, V, HV} of the clean and of the noisy image


def tta_views_2d_device(img, mean: Sequence[float], std: Sequence[float], noise=None, noise_flipped=None,
                        max_pixel_value: float = 255.0, view_codes: Optional[Sequence[int]] = None):
    """tta_views_2d on the device, one launch (vx_tta_views_2d): img (B, H, W, 3) or (H, W, 3) uint8 tensor (host or
    device), noise / noise_flipped float fields of the same shape or None.  Returns (views, transforms): views a float32
    device tensor (G, B, H, W, 4) -- channels-last at pitch 4, channel 3 zero: the layout HighResolutionNet's stem stages
    from (pass it on with nhwc=True) -- bit-exact with the host function's values; transforms as tta_views_2d."""
    import ctypes as C

    import torch

    from . import _lib
    _lib.require_gpu()
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    t = torch.as_tensor(img)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[-1] != 3:
        raise ValueError("tta_views_2d_device: img must be uint8 (B, H, W, 3)")
    t = t.to(dev).contiguous()
    B, H, W, _ = t.shape
    codes = list(view_codes) if view_codes is not None else list(TTA_2D_VIEW_CODES)

    def field(n):
        if n is None:
            return None
        n = torch.as_tensor(n).to(dev, torch.float32)
        if n.dim() == 3:
            n = n.unsqueeze(0)
        if tuple(n.shape) != (B, H, W, 3):
            raise ValueError("tta_views_2d_device: noise fields have the image's shape")
        return n.contiguous()
    n0, n1 = field(noise), field(noise_flipped)
    for g, c in enumerate(codes):      # a noisy view without its field degenerates to the clean one, as the host function's does
        if (c & 4) and (n1 if (c & 16) else n0) is None:
            codes[g] = c & 3
    out = torch.empty((len(codes), B, H, W, 4), dtype=torch.float32, device=dev)
    m3 = (C.c_float * 3)(*[float(v) for v in mean])
    s3 = (C.c_float * 3)(*[float(v) for v in std])
    vc = (C.c_int32 * len(codes))(*codes)
    _lib.check(lib.vx_tta_views_2d(_lib.ptr(t), 1, _lib.ptr(n0), _lib.ptr(n1), m3, s3, float(max_pixel_value), B, H, W,
                                   len(codes), vc, _lib.ptr(out), _lib.stream_ptr()), "vx_tta_views_2d")
    names = [[nm for bit, nm in ((1, "HorizontalFlip"), (2, "VerticalFlip"), (4, "GaussNoise")) if c & bit] for c in codes]
    return out, names


def tta_views_8_device(x, x_noisy, out=None):
    """values_amd.predict2d.tta_views_8 as one launch: x, x_noisy (B, 3, H, W) float32 device tensors (already normalised)
    -> ((8, B, H, W, 4) channels-last views, hflip flags, vflip flags) in pred order.  out: write into this tensor (the
    input of a captured graph: GraphedPredictor2D.x[0])."""
    import ctypes as C

    import torch

    from . import _lib
    _lib.require_gpu()
    lib = _lib.load()
    if x.dim() != 4 or x.shape[1] != 3 or tuple(x_noisy.shape) != tuple(x.shape):
        raise ValueError(f"tta_views_8_device: x and x_noisy must be (B, 3, H, W) tensors of one shape, got {tuple(x.shape)} "
                         f"and {tuple(x_noisy.shape)}")
    # (a host tensor's pointer would reach the kernel as it is: move the inputs like tta_views_2d_device does)
    dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    x = x.to(dev, torch.float32).contiguous()
    xn = x_noisy.to(dev, torch.float32).contiguous()
    B, _, H, W = x.shape
    codes = [0, 1, 2, 3, 4, 5, 6, 7]
    if out is None:
        out = torch.empty((8, B, H, W, 4), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (8, B, H, W, 4) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("tta_views_8_device: out must be a contiguous float32 (8, B, H, W, 4) tensor")
    vc = (C.c_int32 * 8)(*codes)
    _lib.check(lib.vx_tta_views_2d(_lib.ptr(x), 0, _lib.ptr(xn), None, None, None, 0.0, B, H, W, 8, vc, _lib.ptr(out),
                                   _lib.stream_ptr()), "vx_tta_views_2d")
    return out, [bool(c & 1) for c in codes], [bool(c & 2) for c in codes]


def hflip_flags(transforms: Sequence[Sequence[str]]) -> List[bool]:
    """which views test_2D.py flips back (:304-309)."""
    return [any("HorizontalFlip" in s for s in t) for t in transforms]
