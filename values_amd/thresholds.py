"""Threshold search for the threshold aggregation (evaluation/uncertainty_aggregation/find_threshold.py).

Same function names and JSON files as the reference script:
    quantile_analysis.json   {pred_model: mean over images and versions of 1 - foreground / size of pred_seg}
    threshold_analysis.json  {pred_model: {"Mean <unc> threshold": quantile of ALL its validation maps}, "Mean": ...}
(`threshold_aggregation` reads "Mean {predictive|aleatoric|epistemic} threshold", aggregate_uncertainties.py:59-60).

np.quantile over every voxel of every validation map is a selection problem on tens of millions of floats; here the
two order statistics around q * (n - 1) come from `vx_select_kth` (radix select on the device) and numpy's linear
interpolation (`_lerp`, method="linear") is applied to them in float64 with a float64 virtual index -- what
numpy 1.24.3 (the reference's pin, requirements.txt:48) computes for a python-float q, and bit-identical to
np.quantile(maps.astype(float64), q) on any numpy.  (numpy >= 2 rounds q, the index and the interpolation to float32
when the data are float32; the two differ in the 8th digit.)

Note on the reference: `find_threshold` calls `calculate_threshold_image(np.array(unc_images), pred_model)` although
the function is defined as `(quantile_path, image, method)` (find_threshold.py:61-66 vs :93) -- as shipped it raises
a TypeError.  This module implements what the two pieces say together: threshold = quantile of the stacked maps at
the model's foreground quantile read from quantile_analysis.json.
"""
from __future__ import annotations

import json
import os
from itertools import chain
from pathlib import Path
from typing import Dict, Iterable

import numpy as np
import torch

from . import _lib, nifti


def count_nonzero(mask: torch.Tensor) -> int:
    _lib.require_gpu()
    lib = _lib.load()
    m = mask.reshape(-1)
    if m.dtype != torch.uint8:
        m = (m != 0).to(torch.uint8)
    m = m.contiguous()
    out = torch.empty(1, dtype=torch.int64, device=m.device)
    _lib.check(lib.vx_count_nonzero_u8(m.data_ptr(), m.numel(), out.data_ptr(), _lib.stream_ptr()), "vx_count_nonzero_u8")
    return int(out.item())


def calculate_foreground_quantile_image(image) -> float:
    """find_threshold.py:11-13"""
    t = image if isinstance(image, torch.Tensor) else torch.as_tensor(np.asarray(image))
    if not t.is_cuda:
        _lib.require_gpu()
        t = t.cuda()
    return 1 - (count_nonzero(t) / t.numel())


def quantile(values: torch.Tensor, q: float) -> float:
    """np.quantile(values, q) (method='linear') of a float32 device tensor, any shape."""
    _lib.require_gpu()
    lib = _lib.load()
    if not 0.0 <= q <= 1.0:
        raise ValueError("Quantiles must be in the range [0, 1]")
    x = values.reshape(-1)
    if not x.is_cuda:
        x = x.cuda()
    x = x.to(torch.float32).contiguous()
    n = x.numel()
    if n == 0:
        raise ValueError("quantile of an empty array")
    if bool(torch.isnan(x).any()):
        return float("nan")
    virt = q * (n - 1)                      # numpy: _compute_virtual_index(n, q, alpha=1, beta=1)
    lo = int(np.floor(virt))
    hi = min(lo + 1, n - 1)
    gamma = virt - lo
    ws = torch.empty(int(lib.vx_select_workspace_bytes()), dtype=torch.uint8, device=x.device)
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    _lib.check(lib.vx_select_kth(x.data_ptr(), n, lo, out.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), "vx_select_kth")
    _lib.check(lib.vx_select_kth(x.data_ptr(), n, hi, out.data_ptr() + 4, ws.data_ptr(), _lib.stream_ptr()), "vx_select_kth")
    a, b = (np.float64(v) for v in out.cpu().numpy())
    diff = b - a
    res = a + diff * gamma                  # numpy _lerp
    if gamma >= 0.5:
        res = b - diff * (1 - gamma)
    return float(res)


def get_foreground_quantile(exp_dataloader) -> Dict:
    """find_threshold.py:16-29"""
    all_quantiles = []
    for image_id in exp_dataloader.image_ids:
        for pred_seg in exp_dataloader.get_pred_segs(image_id):
            all_quantiles.append(calculate_foreground_quantile_image(np.asarray(pred_seg)))
    return {exp_dataloader.exp_version.pred_model: {exp_dataloader.exp_version.version_name: all_quantiles}}


def save_foreground_quantiles(results_dict: Dict, save_path) -> Dict:
    """find_threshold.py:32-41"""
    methods = {m: float(np.mean(list(chain.from_iterable(v.values())))) for m, v in results_dict.items()}
    if not os.path.isfile(save_path):
        save_path = Path(save_path) / "quantile_analysis.json"
    with open(save_path, "w") as f:
        json.dump(methods, f, indent=2)
    return methods


def threshold_images_paths(exp_dataloader) -> Dict:
    """find_threshold.py:44-58"""
    ev = exp_dataloader.exp_version
    d = {ev.pred_model: {ev.version_name: {}}}
    for unc_type in ev.unc_types:
        p = exp_dataloader.unc_path_dict[unc_type]
        d[ev.pred_model][ev.version_name][unc_type] = [p / f"{i}{ev.unc_ending}" for i in exp_dataloader.image_ids]
    return d


def calculate_threshold_image(quantile_path, image, method: str) -> float:
    """find_threshold.py:61-66; `image`: array / device tensor / iterable of maps (stacked)."""
    with open(quantile_path) as f:
        all_quantiles = json.load(f)
    if isinstance(image, torch.Tensor):
        t = image
    elif isinstance(image, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(image))
    else:
        t = torch.cat([torch.as_tensor(np.asarray(i)).reshape(-1) for i in image])
    return quantile(t, all_quantiles[method])


def find_threshold(results_dict: Dict, quantile_path, save_path, loader=None) -> Dict:
    """find_threshold.py:69-117.  results_dict: {pred_model: {version: {unc_type: [paths]}}} (threshold_images_paths,
    merged over versions).  Maps are read with the package's NIfTI reader unless `loader(path) -> array` is given."""
    if not os.path.isfile(quantile_path):
        quantile_path = Path(quantile_path) / "quantile_analysis.json"
    if not os.path.isfile(save_path):
        save_path = Path(save_path) / "threshold_analysis.json"
    load = loader or (lambda p: nifti.load(p)[0])
    per_model = {}
    for pred_model, versions in results_dict.items():
        per_model[pred_model] = {}
        for _version, uncs in versions.items():
            for unc, paths in uncs.items():
                per_model[pred_model].setdefault(unc, []).extend(paths)
    threshold_dict = {}
    for pred_model, uncs in per_model.items():
        threshold_dict[pred_model] = {}
        for unc, paths in uncs.items():
            maps = [np.asarray(load(p), dtype=np.float32) for p in paths]
            thr = calculate_threshold_image(quantile_path, maps, pred_model)
            threshold_dict[pred_model][f"Mean {unc.split('_')[0]} threshold"] = thr
    al, ep, pr = [], [], []
    for key, value in threshold_dict.items():
        if key != "Softmax":
            al.append(value["Mean aleatoric threshold"])
            ep.append(value["Mean epistemic threshold"])
        pr.append(value["Mean predictive threshold"])
    threshold_dict["Mean"] = {"Mean aleatoric threshold": float(np.mean(al)), "Mean epistemic threshold": float(np.mean(ep)),
                              "Mean predictive threshold": float(np.mean(pr))}
    with open(save_path, "w") as f:
        json.dump(threshold_dict, f, indent=2)
    return threshold_dict
