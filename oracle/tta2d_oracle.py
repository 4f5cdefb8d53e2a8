"""ORACLE (test infrastructure, not product code) -- CPU restatement of the 2D test-time-augmentation branch of the
reference's dataset class.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restates /root/reference/uncertainty_modeling/data/cityscapes_dataset.py:76-99 (`Cityscapes_dataset.__getitem__`, the
`self.tta` branch):

    images = [img, HorizontalFlip(img), GaussNoise(img), GaussNoise(HorizontalFlip(img))]
    images = [self.transforms(image=image)["image"].float() for image in images]      # Normalize + ToTensorV2
    transforms = [[], ["HorizontalFlip"], ["GaussNoise"], ["HorizontalFlip", "GaussNoise"]]

The arithmetic lives in a THIRD-PARTY dependency that is absent from /root/reference and from this image:
albumentations==1.3.0 (/root/reference/requirements.txt:4).  Its published algorithm, restated here function by function
(file names are those of the albumentations 1.3.0 source tree):

  * HorizontalFlip.apply -> augmentations/geometric/functional.py `hflip`: `img[:, ::-1, ...]` (cv2.flip(img, 1) for
    3-channel uint8 images: the same values).
  * GaussNoise (augmentations/transforms.py; defaults var_limit=(10.0, 50.0), mean=0, per_channel=True):
    `var = random.uniform(*var_limit); sigma = var ** 0.5; gauss = RandomState(random.randint(0, 2**32 - 1)).normal(mean,
    sigma, image.shape)`; apply -> augmentations/functional.py `gauss_noise`, decorated `@clipped`:
    `image.astype("float32") + gauss`, then `np.clip(., 0, MAX_VALUES_BY_DTYPE[uint8] = 255).astype(uint8)` (truncation).
    The DRAW (python `random` + a fresh RandomState per call) is not reproducible outside albumentations: the field `gauss`
    is an INPUT of this oracle, as it is of the product (values_amd.data.tta_views_2d, vx_tta_views_2d).  albumentations'
    field is float64; the product's boundary takes float32 fields -- this oracle adds the field in the dtype it is given
    (numpy promotion, like the library) so both can be stated.
  * Normalize.apply -> augmentations/functional.py `normalize(img, mean, std, max_pixel_value=255.0)`:
    `mean = float32(mean) * max_pixel_value; std = float32(std) * max_pixel_value; denominator = np.reciprocal(std,
    dtype=float32); img = img.astype(float32); img -= mean; img *= denominator` (the 3-channel path runs the same two
    float32 operations through cv2.subtract / cv2.multiply).
  * ToTensorV2.apply (pytorch/transforms.py): `torch.from_numpy(img.transpose(2, 0, 1))`.

PARITY UNPINNED: albumentations cannot be imported here, so no fixture of the library's own output exists;
this file pins the product to the library's PUBLISHED arithmetic only.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

TRANSFORMS = [[], ["HorizontalFlip"], ["GaussNoise"], ["HorizontalFlip", "GaussNoise"]]   # cityscapes_dataset.py:78-90


def hflip(img: np.ndarray) -> np.ndarray:
    """albumentations 1.3.0 geometric/functional.py `hflip`."""
    return np.ascontiguousarray(img[:, ::-1, ...])


def vflip(img: np.ndarray) -> np.ndarray:
    """albumentations 1.3.0 geometric/functional.py `vflip` (config C4's extra views; not used by the reference's branch)."""
    return np.ascontiguousarray(img[::-1, ...])


def gauss_noise(image: np.ndarray, gauss: np.ndarray) -> np.ndarray:
    """albumentations 1.3.0 functional.py `gauss_noise` under its `@clipped` decorator, for a uint8 image."""
    dtype = image.dtype
    maxval = 255.0 if dtype == np.uint8 else 1.0
    out = image.astype("float32") + gauss
    return np.clip(out, 0, maxval).astype(dtype)


def normalize(img: np.ndarray, mean: Sequence[float], std: Sequence[float], max_pixel_value: float = 255.0) -> np.ndarray:
    """albumentations 1.3.0 functional.py `normalize` / `normalize_numpy`."""
    m = np.array(mean, dtype=np.float32)
    m *= max_pixel_value
    s = np.array(std, dtype=np.float32)
    s *= max_pixel_value
    denominator = np.reciprocal(s, dtype=np.float32)
    out = img.astype(np.float32)
    out -= m
    out *= denominator
    return out


def to_tensor_v2(img: np.ndarray) -> np.ndarray:
    """albumentations 1.3.0 pytorch/transforms.py `ToTensorV2.apply`: HWC -> CHW (as a numpy array here)."""
    return np.ascontiguousarray(img.transpose(2, 0, 1))


def tta_branch(img: np.ndarray, mean: Sequence[float], std: Sequence[float], gauss: Optional[np.ndarray],
               gauss_flipped: Optional[np.ndarray], max_pixel_value: float = 255.0) -> Tuple[List[np.ndarray], List[List[str]]]:
    """cityscapes_dataset.py:76-99 with the two GaussNoise fields as inputs (gauss: drawn for `img`; gauss_flipped: drawn
    for the flipped image -- the reference calls noise_transform a second time ON `flipped["image"]`, :88).  A field that
    is None leaves the view un-noised.  Returns ([4 x (3, H, W) float32], transforms)."""
    img = np.asarray(img)
    images = [img]
    flipped = hflip(img)
    images.append(flipped)
    images.append(gauss_noise(img, gauss) if gauss is not None else img)
    images.append(gauss_noise(flipped, gauss_flipped) if gauss_flipped is not None else flipped)
    out = [to_tensor_v2(normalize(im, mean, std, max_pixel_value)).astype(np.float32) for im in images]   # `.float()`, :91
    return out, [list(t) for t in TRANSFORMS]


def tta_views_8(x: np.ndarray, x_noisy: np.ndarray) -> Tuple[List[np.ndarray], List[bool], List[bool]]:
    """BASELINE config C4's 8 views (an extension of the reference's four): {id, H, V, HV} of the clean and of the
    noisy normalised tensors (B, 3, H, W), in the order of values_amd.predict2d.tta_views_8."""
    views, hf, vf = [], [], []
    for src in (x, x_noisy):
        for code in range(4):
            v = src
            if code & 1:
                v = v[..., ::-1]
            if code & 2:
                v = v[..., ::-1, :]
            views.append(np.ascontiguousarray(v))
            hf.append(bool(code & 1))
            vf.append(bool(code & 2))
    return views, hf, vf
