"""Results-directory writer: the tree DataCarrier3D.save_data / log_metrics produce
(uncertainty_modeling/data_carrier_3D.py:17-57, 181-391), fed from device tensors.

    <root>/<exp_name>/test_results/<version>/<split>/
        input/<id>.nii.gz                    gt_seg/<id>_<RR>.nii.gz
        pred_seg/<id>_mean.nii.gz, <id>_<NN>.nii.gz          (uint8 argmax; NN = 1-based prediction index)
        pred_prob/<id>_mean_<CC>.nii.gz, <id>_<NN>_<CC>.nii.gz   (float64 like the reference's numpy buffers)
        pred_entropy/, aleatoric_uncertainty/, epistemic_uncertainty/<id>.nii.gz   (float32 maps / clip(count,1))
        metrics.json
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import numpy as np

from . import nifti


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def save_case(save_dir: str, image_id: str, softmax_pred, maps: Optional[Dict] = None, data=None, gt_seg=None,
              num_predictions=None, header=False) -> None:
    """softmax_pred: (T, C, X,Y,Z) probabilities (sums if num_predictions is given: divided by clip(count, 1) like
    data_carrier_3D.py:208-217); maps: pred_entropy / aleatoric_uncertainty / epistemic_uncertainty, already
    normalised."""
    sub = {k: os.path.join(save_dir, k) for k in ("input", "gt_seg", "pred_seg", "pred_prob")}
    for d in sub.values():
        os.makedirs(d, exist_ok=True)
    sm = _np(softmax_pred).astype(np.float64)
    if num_predictions is not None:
        sm = sm / np.clip(_np(num_predictions), 1, None)
    if data is not None:
        nifti.save(_np(data), os.path.join(sub["input"], f"{image_id}.nii.gz"), header)
    if gt_seg is not None:
        for r, g in enumerate(_np(gt_seg)):
            nifti.save(g, os.path.join(sub["gt_seg"], f"{image_id}_{str(r).zfill(2)}.nii.gz"), header)
    T, C = sm.shape[:2]
    if T > 1:  # data_carrier_3D.py:253-279
        mean = sm.mean(axis=0)
        nifti.save(np.argmax(mean, axis=0).astype(np.uint8), os.path.join(sub["pred_seg"], f"{image_id}_mean.nii.gz"), header)
        for c in range(C):
            nifti.save(mean[c], os.path.join(sub["pred_prob"], f"{image_id}_mean_{str(c + 1).zfill(2)}.nii.gz"), header)
    for t in range(T):  # :281-307
        tag = str(t + 1).zfill(2)
        nifti.save(np.argmax(sm[t], axis=0).astype(np.uint8), os.path.join(sub["pred_seg"], f"{image_id}_{tag}.nii.gz"), header)
        for c in range(C):
            nifti.save(sm[t, c], os.path.join(sub["pred_prob"], f"{image_id}_{tag}_{str(c + 1).zfill(2)}.nii.gz"), header)
    for k in ("pred_entropy", "aleatoric_uncertainty", "epistemic_uncertainty"):  # :323-371
        if maps and k in maps:
            os.makedirs(os.path.join(save_dir, k), exist_ok=True)
            nifti.save(_np(maps[k]), os.path.join(save_dir, k, f"{image_id}.nii.gz"), header)


def results_dir(root_dir: str, exp_name: str, version, test_split: str = "id") -> str:
    return os.path.join(root_dir, exp_name, "test_results", str(version), test_split)  # data_carrier_3D.py:40-42


def log_metrics(save_dir: str, per_image_metrics: Dict[str, Dict[str, float]]) -> None:
    """metrics.json with a "mean" entry (data_carrier_3D.py:373-391)."""
    out = {k: dict(v) for k, v in per_image_metrics.items()}
    names = sorted({m for v in per_image_metrics.values() for m in v})
    out["mean"] = {m: float(np.mean([v[m] for v in per_image_metrics.values() if m in v])) for m in names}
    with open(os.path.join(save_dir, "metrics.json"), "w") as f:
        json.dump(out, f, indent=2)
