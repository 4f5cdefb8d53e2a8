"""ORACLE (test infrastructure, not product code) -- CPU restatement of the
reference's map -> scalar aggregations.

Restates /root/reference/evaluation/uncertainty_aggregation/aggregate_uncertainties.py:
  * patch_level_aggregation  (:13-31)  box-sum ('valid') -> max, first bbox attaining it
  * image_level_aggregation  (:34-37)
  * threshold_aggregation    (:40-67)

The reference's box-sum is scipy.signal.convolve(image, ones, 'valid'), which may
choose fftconvolve; for a float32 map (what medpy returns for the saved NIfTI) scipy
transforms it in SINGLE precision, so the reference value carries ~1e-7 relative
noise (measured: 4e-8 at 24^3).  This restatement uses exact separable cumulative
sums in float64; comparisons against the golden values therefore use rtol 1e-6.

Parity pin: tests/test_oracle_golden.py vs tests/golden/agg_kat.json.
"""
from __future__ import annotations

import numpy as np


def box_sum_valid(image: np.ndarray, patch_size) -> np.ndarray:
    out = np.asarray(image, dtype=np.float64)
    for ax, k in enumerate(patch_size):
        c = np.cumsum(out, axis=ax)
        c = np.concatenate([np.zeros_like(np.take(c, [0], axis=ax)), c], axis=ax)
        n = out.shape[ax]
        hi = np.take(c, np.arange(k, n + 1), axis=ax)
        lo = np.take(c, np.arange(0, n - k + 1), axis=ax)
        out = hi - lo
    return out


def patch_level_aggregation(image, patch_size, mean=False, **kwargs):
    if type(patch_size) == int:
        patch_size = len(image.shape) * [patch_size]
    agg = box_sum_valid(image, patch_size)
    if mean:
        agg = agg / np.prod(patch_size)
    mx = np.max(agg)
    # np.where(np.isclose(agg, max)) -> first index in C order (:20-23)
    first = np.argwhere(np.isclose(agg, mx))[0]
    bbox = [(int(i), int(i + patch_size[d])) for d, i in enumerate(first)]
    return {"max_score": float(mx), "bounding_box": bbox}


def image_level_aggregation(image, mean=False, **kwargs):
    if mean:
        return float(np.sum(image) / image.size)  # NB: bare float, not a dict (:35-36)
    return {"max_score": float(np.sum(image))}


def threshold_aggregation(image, threshold=None, mean=True, **kwargs):
    if threshold is None:
        raise Exception("A threshold needs to be provided for threshold aggregation!")
    sel = image >= threshold
    s = image[sel].sum()
    count = sel.sum()
    if mean and count > 0:
        return {"max_score": s / count, "threshold": threshold}
    return {"max_score": s, "threshold": threshold}
