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


def hflip_flags(transforms: Sequence[Sequence[str]]) -> List[bool]:
    """which views test_2D.py flips back (:304-309)."""
    return [any("HorizontalFlip" in s for s in t) for t in transforms]
