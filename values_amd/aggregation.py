"""Map -> scalar aggregations on MI355X; same names / arguments / return dicts as
evaluation/uncertainty_aggregation/aggregate_uncertainties.py:13-67, so the hydra `_target_`
strings of evaluation/configs/tasks/aggregation_*.yaml can be re-pointed here.
"""
from __future__ import annotations

import json

import numpy as np
import torch

from . import _lib


def _dev_map(image):
    _lib.require_gpu()
    if isinstance(image, np.ndarray):
        image = torch.from_numpy(np.ascontiguousarray(image))
    dev = image.device if image.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return image.to(dev, torch.float32).contiguous(), dev


def patch_level_aggregation(image, patch_size, mean=False, **kwargs):
    if type(patch_size) == int:
        patch_size = len(image.shape) * [patch_size]
    img, dev = _dev_map(image)
    nd = img.dim()
    if nd not in (2, 3):
        raise ValueError("patch_level_aggregation: 2D or 3D maps only")
    shape = (1,) * (3 - nd) + tuple(img.shape)
    patch = (1,) * (3 - nd) + tuple(int(p) for p in patch_size)
    n = img.numel()
    ws = torch.empty(2 * n + 2, dtype=torch.float64, device=dev)
    res = torch.empty(1, dtype=torch.float64, device=dev)
    idx = torch.empty(3, dtype=torch.int32, device=dev)
    rc = _lib.load().vx_box_max(_lib.ptr(img), *shape, *patch, _lib.ptr(res), _lib.ptr(idx), _lib.ptr(ws),
                                ws.numel() * 8, _lib.stream_ptr())
    _lib.check(rc, "vx_box_max")
    mx = float(res.item())
    first = idx.tolist()[3 - nd:]
    if mean:
        mx = mx / float(np.prod(patch_size))
    return {"max_score": mx, "bounding_box": [(int(i), int(i + patch_size[d])) for d, i in enumerate(first)]}


def _sums(image, thr):
    """(sum, sum of values >= thr, count of values >= thr) in float64; a float64 map (what medpy hands the reference
    after a NIfTI round trip) stays float64, so `map >= threshold` is the reference's comparison bit for bit."""
    _lib.require_gpu()
    if isinstance(image, np.ndarray):
        image = torch.from_numpy(np.ascontiguousarray(image))
    dev = image.device if image.is_cuda else torch.device("cuda", torch.cuda.current_device())
    f64 = image.dtype == torch.float64
    img = image.to(dev, torch.float64 if f64 else torch.float32).contiguous()
    sums = torch.empty(3, dtype=torch.float64, device=dev)
    rc = _lib.load().vx_sum_thr(_lib.ptr(img), _lib.VX_F64 if f64 else _lib.VX_F32, img.numel(), float(thr),
                                _lib.ptr(sums), _lib.stream_ptr())
    _lib.check(rc, "vx_sum_thr")
    return sums.tolist(), img.numel()


def image_level_aggregation(image, mean=False, **kwargs):
    (s, _, _), n = _sums(image, float("inf"))
    if mean:
        return float(s / n)  # the reference returns a bare float here (:35-36)
    return {"max_score": float(s)}


def threshold_aggregation(image, threshold=None, threshold_path=None, pred_model=None, unc_type=None, mean=True):
    if threshold is None:
        if threshold_path is None:
            raise Exception("A threshold needs to be provided for threshold aggregation!")
        with open(threshold_path) as f:
            threshold_json = json.load(f)
        if pred_model is None or unc_type is None:
            raise Exception("If you want to load the threshold from a json file, you have to provide the prediction "
                            "model and the uncertainty type")
        unc_type_split = unc_type.split("_")[0]
        threshold = threshold_json[pred_model][f"Mean {unc_type_split} threshold"]
    (_, st, ct), _ = _sums(image, threshold)
    if mean and ct > 0:
        return {"max_score": st / ct, "threshold": threshold}
    return {"max_score": st, "threshold": threshold}
