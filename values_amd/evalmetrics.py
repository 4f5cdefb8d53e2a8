"""Downstream scalars of the evaluation stage: same function names, arguments and result files as
evaluation/metrics/{aurc,ncc,ace,auroc}.py, so the task functions of evaluation/configs/tasks/*.yaml can be re-pointed.

Where the work is per voxel it runs on the GPU in float64 (values_amd/csrc/evalmetrics.hip, deterministic reductions):
  compute_ncc              ncc.py:9-25      two-pass mean / std(ddof=1) / cross product of two maps
  sigmoid_calibration      ace.py:13-41     Platt scaling of -uncertainty against "reference == prediction": loss,
                                            gradient and Hessian sums on the device, Newton steps on the host
  calc_ace                 ace.py:44-90     platt_scale_confid + the 20-bin statistics of calib_stats in one pass
Where it is one scalar per IMAGE (AURC / E-AURC over (risk, confidence) pairs, AUROC over (OoD label, score) pairs:
a few hundred numbers) it stays on the host, restated in numpy float64: aurc.py:14-67, and sklearn's roc_curve + auc
as auroc.py:126-127 calls them.  scikit-learn itself is not needed.
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np
import torch

from . import _lib

_ws = {}


def _workspace(dev):
    key = str(dev)
    if key not in _ws:
        _ws[key] = torch.empty(_lib.load().vx_evalmetrics_workspace_bytes(), dtype=torch.uint8, device=dev)
    return _ws[key]


def _dev():
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def _float_map(a, dev):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a.detach()
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    t = t.to(dev).contiguous()
    return t, (_lib.VX_F64 if t.dtype == torch.float64 else _lib.VX_F32)


# ------------------------------------------------------------------------------------------------ failure detection
def rc_curve_stats(risks, confids):
    """aurc.py:14-51: coverages, selective risks and weights of the risk-coverage curve (ties in the confidence
    collapse into one point)."""
    risks, confids = np.asarray(risks, dtype=np.float64), np.asarray(confids, dtype=np.float64)
    assert risks.ndim == 1 and confids.ndim == 1 and len(risks) == len(confids)
    n = len(risks)
    order = np.argsort(confids)
    r, c = risks[order], confids[order]
    coverage, err = n, float(sum(r))
    coverages, selective, weights = [coverage / n], [err / n], []
    pending = 0
    for i in range(n - 1):
        coverage -= 1
        err -= r[i]
        pending += 1
        if i == 0 or c[i] != c[i - 1]:
            coverages.append(coverage / n)
            selective.append(err / (n - 1 - i))
            weights.append(pending / n)
            pending = 0
    if pending > 0:
        coverages.append(0)
        selective.append(selective[-1])
        weights.append(pending / n)
    return coverages, selective, weights


def aurc(risks, confids):
    _, sel, w = rc_curve_stats(risks, confids)
    return sum((sel[i] + sel[i + 1]) * 0.5 * w[i] for i in range(len(w)))


def eaurc(risks, confids):
    """AURC minus the AURC of the optimal confidence ranking (aurc.py:61-67)"""
    risks = np.asarray(risks, dtype=np.float64)
    n = len(risks)
    best = np.sort(risks).cumsum() / np.arange(1, n + 1)
    return aurc(risks, confids) - best.sum() / n


def _metric_entry(metrics, image_id):
    if image_id not in metrics:
        keys = [k for k in metrics if k.split("/")[-1].split(".")[0] == image_id]
        if len(keys) > 1:
            print(f"Found multiple matches for image id {image_id}. Using the first match {keys[0]}")
        image_id = keys[0]
    e = metrics[image_id]
    return e["dice"] if "dice" in e else e["metrics"]["dice"]


def get_dice(image_id, metrics_file):
    with open(metrics_file) as f:
        return _metric_entry(json.load(f), image_id)


def get_risk(image_id, metrics_file):
    return 1 - get_dice(image_id, metrics_file)


def get_confid(image_name, aggregated_unc_file, aggregation_level, unc_file_ending):
    with open(aggregated_unc_file) as f:
        unc = json.load(f)
    return -unc[f"{image_name}{unc_file_ending}"][aggregation_level]["max_score"]


def get_risks_and_confids(dataset_path, image_ids, unc_type, aggregation, unc_file_ending):
    risks, confids, dices = [], [], []
    for image in image_ids:
        dice = get_dice(image, dataset_path / "metrics.json")
        dices.append(dice)
        risks.append(1 - dice)
        confids.append(get_confid(image, dataset_path / f"aggregated_{unc_type}.json", aggregation, unc_file_ending))
    return risks, confids, dices


def failure_detection(exp_dataloader):
    """aurc.py:128-153 (its `main`): failure_detection.json with AURC / E-AURC per uncertainty type and aggregation"""
    ev = exp_dataloader.exp_version
    res = {"mean": {}}
    for unc_type in ev.unc_types:
        res["mean"][unc_type] = {}
        for aggregation in ev.aggregations:
            risks, confids, _ = get_risks_and_confids(exp_dataloader.dataset_path, exp_dataloader.image_ids, unc_type,
                                                      aggregation, ev.unc_ending)
            res["mean"][unc_type][aggregation] = {"metrics": {"aurc": aurc(np.array(risks), np.array(confids)),
                                                              "eaurc": eaurc(np.array(risks), np.array(confids))}}
    with open(exp_dataloader.dataset_path / "failure_detection.json", "w") as f:
        json.dump(res, f, indent=2)
    return res


# ------------------------------------------------------------------------------------------------ ambiguity modelling
def compute_ncc(gt_unc_map, pred_unc_map):
    """ncc.py:9-25 on the device: numpy's two-pass moments, in float64 whatever the maps' dtype (numpy works in the
    dtype of the map: a float32 map read back from NIfTI gives the reference a float32-rounded value, ~1e-7 away)."""
    lib, dev = _lib.load(), _dev()
    g, gd = _float_map(gt_unc_map, dev)
    p, pd = _float_map(pred_unc_map, dev)
    n = g.numel()
    if p.numel() != n:
        raise ValueError("compute_ncc: maps of different size")
    sums = torch.empty(3, dtype=torch.float64, device=dev)
    ws = _workspace(dev)
    st = _lib.stream_ptr()
    _lib.check(lib.vx_ncc_sums(_lib.ptr(g), gd, _lib.ptr(p), pd, n, 0, 0.0, 0.0, _lib.ptr(sums), _lib.ptr(ws), st), "vx_ncc_sums")
    s0 = sums.tolist()
    mg, mp = s0[0] / n, s0[1] / n
    _lib.check(lib.vx_ncc_sums(_lib.ptr(g), gd, _lib.ptr(p), pd, n, 1, mg, mp, _lib.ptr(sums), _lib.ptr(ws), st), "vx_ncc_sums")
    vg, vp, prod = sums.tolist()
    sg, sp = np.sqrt(vg / (n - 1)), np.sqrt(vp / (n - 1))
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.float64(1.0) / (np.float64(n) * sg * sp) * np.float64(prod)


def ambiguity_modeling(exp_dataloader):
    """ncc.py:28-44 (its `main`): ambiguity_modeling.json"""
    res = {"mean": {}}
    for unc_type in exp_dataloader.exp_version.unc_types:
        vals = []
        for image_id in exp_dataloader.image_ids:
            res.setdefault(image_id, {})
            ncc = float(compute_ncc(exp_dataloader.get_gt_unc_map(image_id), exp_dataloader.get_unc_map(image_id, unc_type)))
            res[image_id][unc_type] = {"metrics": {"ncc": ncc}}
            vals.append(ncc)
        res["mean"][unc_type] = {"metrics": {"ncc": float(np.mean(np.array(vals)))}}
    with open(exp_dataloader.dataset_path / "ambiguity_modeling.json", "w") as f:
        json.dump(res, f, indent=2)
    return res


# ------------------------------------------------------------------------------------------------ calibration
class _RaterInputs:
    """reference segmentations (R, *spatial), mean prediction (*spatial) and uncertainty map on the device, shaped as
    ace.py:19-29 / :104-114 shape them (a 2D map loaded as (W, H) is swapped to the prediction's (H, W))"""

    def __init__(self, reference_segs, pred_seg, unc_map, ignore_value=None):
        dev = _dev()
        ref = np.asarray(reference_segs)
        pred = np.asarray(pred_seg)
        unc = unc_map.detach().cpu().numpy() if isinstance(unc_map, torch.Tensor) else np.asarray(unc_map)
        if pred.shape != unc.shape:
            unc = np.swapaxes(unc, 0, 1)
        if ref.shape[1:] != pred.shape or unc.shape != pred.shape:
            raise ValueError(f"reference {ref.shape}, prediction {pred.shape} and map {unc.shape} do not fit together")
        self.R, self.nvox = int(ref.shape[0]), int(pred.size)
        self.ref = torch.from_numpy(np.ascontiguousarray(ref, dtype=np.int32)).to(dev)
        self.pred = torch.from_numpy(np.ascontiguousarray(pred, dtype=np.int32)).to(dev)
        self.unc, self.dtype = _float_map(unc, dev)
        self.ignore = -1 if ignore_value is None else int(ignore_value)
        if ignore_value is not None and int(ignore_value) < 0:
            raise ValueError("ignore_value must be a non-negative label")
        self.dev = dev


def _platt_sums(x: _RaterInputs, A, B, t_pos, t_neg):
    sums = torch.empty(8, dtype=torch.float64, device=x.dev)
    _lib.check(_lib.load().vx_platt_sums(_lib.ptr(x.unc), x.dtype, _lib.ptr(x.ref), _lib.ptr(x.pred), x.R, x.nvox, x.ignore,
                                         float(A), float(B), float(t_pos), float(t_neg), _lib.ptr(sums),
                                         _lib.ptr(_workspace(x.dev)), _lib.stream_ptr()), "vx_platt_sums")
    return sums.tolist()


def sigmoid_calibration(reference_segs, pred_seg, unc_map, ignore_value=None, max_iter=100):
    """(a, b) of sklearn.calibration._sigmoid_calibration(-unc, reference == prediction) as ace.py:30-36 calls it:
    Platt's regularised targets, P = 1 / (1 + exp(a F + b)).  The objective is convex in (a, b); every evaluation
    (loss, gradient, Hessian: sums over all rater-voxels) is one device pass, the host does damped Newton steps to the
    optimum (|gradient| below 1e-10 per sample).  sklearn reaches the same optimum with BFGS (1.2.2, the reference's
    pin) or L-BFGS-B (>= 1.4) to ITS tolerance -- that difference is the only unpinned part."""
    x = _RaterInputs(reference_segs, pred_seg, unc_map, ignore_value)
    s = _platt_sums(x, 0.0, 0.0, 0.5, 0.5)
    n, n1 = s[0], s[1]
    if n <= 0:
        raise ValueError("sigmoid_calibration: no valid voxel")
    prior1, prior0 = n1, n - n1
    t_pos, t_neg = (prior1 + 1.0) / (prior1 + 2.0), 1.0 / (prior0 + 2.0)
    A, B = 0.0, float(np.log((prior0 + 1.0) / (prior1 + 1.0)))
    s = _platt_sums(x, A, B, t_pos, t_neg)
    for _ in range(max_iter):
        loss, g, H = s[2], np.array([s[3], s[4]]), np.array([[s[5], s[6]], [s[6], s[7]]])
        if np.abs(g).max() < 1e-10 * max(1.0, n):
            break
        step = np.linalg.solve(H + 1e-12 * np.eye(2), g)
        t = 1.0
        while True:
            s2 = _platt_sums(x, A - t * step[0], B - t * step[1], t_pos, t_neg)
            if s2[2] <= loss + 1e-12 * abs(loss) or t < 1e-10:
                break
            t *= 0.5
        A, B, s = A - t * step[0], B - t * step[1], s2
    return float(A), float(B)


def platt_scale_confid(uncalib_confid, platt_scale_file, uncertainty):
    """ace.py:44-48 (host form, for scalars / small arrays; calc_ace fuses it into the binning pass)"""
    with open(platt_scale_file) as f:
        params = json.load(f)[uncertainty]
    return 1 / (1 + np.exp(np.asarray(uncalib_confid, dtype=np.float64) * params["a"] + params["b"]))


def calib_stats(reference_segs, pred_seg, unc_map, a, b, ignore_value=None):
    """calib_stats (ace.py:51-82) of platt_scale_confid(-unc, a, b) against "reference == prediction": bin discrepancies,
    bin weights and the number of non-empty bins; the 20 bins of np.linspace(0, 1 + 1e-8, 21)."""
    x = _RaterInputs(reference_segs, pred_seg, unc_map, ignore_value)
    edges = np.linspace(0.0, 1.0 + 1e-8, 21)
    e = (C.c_double * 21)(*edges.tolist())
    out = torch.empty(63, dtype=torch.float64, device=x.dev)
    _lib.check(_lib.load().vx_calib_bins(_lib.ptr(x.unc), x.dtype, _lib.ptr(x.ref), _lib.ptr(x.pred), x.R, x.nvox, x.ignore,
                                         float(a), float(b), e, _lib.ptr(out), _lib.ptr(_workspace(x.dev)),
                                         _lib.stream_ptr()), "vx_calib_bins")
    h = out.cpu().numpy()
    bin_sums, bin_true, bin_total = h[:21], h[21:42], h[42:]
    n = bin_total.sum()
    if n <= 0:
        raise ValueError("calib_stats: no valid voxel")
    if bin_true.sum() == n or bin_true.sum() == 0:
        # one label only: sklearn's label_binarize(y, classes=[label])[:, 0] is a column of zeros (ace.py:68), even when
        # every voxel is correct -- reproduced
        bin_true = np.zeros_like(bin_true)
    nz = bin_total != 0
    disc = np.abs(bin_true[nz] / bin_total[nz] - bin_sums[nz] / bin_total[nz])
    return disc, bin_total[nz] / n, int(nz.sum())


def calc_ace(reference_segs, pred_seg, unc_map, a, b, ignore_value=None):
    """ace.py:85-87 on the device inputs: average calibration error over the non-empty bins"""
    disc, _, k = calib_stats(reference_segs, pred_seg, unc_map, a, b, ignore_value)
    return (1 / k) * np.sum(disc)


def platt_scale_params(val_exp_dataloader, ignore_value=None):
    """ace.py:13-41: per uncertainty type the mean (a, b) over the validation images -> platt_scale_params.json"""
    res = {}
    for unc_type in val_exp_dataloader.exp_version.unc_types:
        aa, bb = [], []
        for image_id in val_exp_dataloader.image_ids:
            a, b = sigmoid_calibration(val_exp_dataloader.get_reference_segs(image_id),
                                       val_exp_dataloader.get_mean_pred_seg(image_id),
                                       val_exp_dataloader.get_unc_map(image_id, unc_type), ignore_value)
            aa.append(a)
            bb.append(b)
        res[unc_type] = {"a": float(np.mean(np.array(aa))), "b": float(np.mean(np.array(bb)))}
    with open(val_exp_dataloader.exp_version.exp_path / "platt_scale_params.json", "w") as f:
        json.dump(res, f, indent=2)
    return res


def calibration_error(exp_dataloader, ignore_value=None):
    """ace.py:90-131: calibration.json with the ACE per image and uncertainty type"""
    with open(exp_dataloader.exp_version.exp_path / "platt_scale_params.json") as f:
        params = json.load(f)
    res = {"mean": {}}
    for unc_type in exp_dataloader.exp_version.unc_types:
        vals = []
        for image_id in exp_dataloader.image_ids:
            res.setdefault(image_id, {})
            ace = float(calc_ace(exp_dataloader.get_reference_segs(image_id), exp_dataloader.get_mean_pred_seg(image_id),
                                 exp_dataloader.get_unc_map(image_id, unc_type), params[unc_type]["a"], params[unc_type]["b"],
                                 ignore_value))
            res[image_id][unc_type] = {"metrics": {"ace": ace}}
            vals.append(ace)
        res["mean"][unc_type] = {"metrics": {"ace": float(np.mean(np.array(vals)))}}
    with open(exp_dataloader.dataset_path / "calibration.json", "w") as f:
        json.dump(res, f, indent=2)
    return res


def calibration(exp_dataloader, ignore_value=None):
    """ace.py:134-143 (its `main`): fit the Platt parameters on the validation split if they are not there yet"""
    if not os.path.isfile(exp_dataloader.exp_version.exp_path / "platt_scale_params.json"):
        from .experiment import ExperimentDataloader
        platt_scale_params(ExperimentDataloader(exp_dataloader.exp_version, "val"), ignore_value=ignore_value)
    return calibration_error(exp_dataloader, ignore_value=ignore_value)


# ------------------------------------------------------------------------------------------------ OoD detection
def roc_auc(y_true, y_score):
    """sklearn.metrics.roc_curve + auc as auroc.py:126-127 calls them (positive label 1): one threshold per distinct
    score, cumulative true / false positive rates, trapezoid."""
    y_true = np.asarray(y_true)
    y_score = np.asarray(y_score, dtype=np.float64)
    order = np.argsort(-y_score, kind="mergesort")
    score, pos = y_score[order], y_true[order] == 1
    last = np.r_[np.where(np.diff(score))[0], len(score) - 1]      # last index of every run of equal scores
    tps = np.r_[0.0, np.cumsum(pos)[last]]
    fps = np.r_[0.0, 1 + last - np.cumsum(pos)[last]]
    if tps[-1] <= 0 or fps[-1] <= 0:
        return float("nan")
    tpr, fpr = tps / tps[-1], fps / fps[-1]
    return float(np.sum((fpr[1:] - fpr[:-1]) * (tpr[1:] + tpr[:-1]) * 0.5))


def is_ood_toy(sample):
    """auroc.py:18-24: in the toy data sets samples numbered up to 20 are out of distribution"""
    return not int(sample.split(".")[0]) > 20


def get_auroc_input(uncertainties, aggregation, is_ood=is_ood_toy):
    """auroc.py:79-93: (OoD labels, scores) from an aggregated_<unc>.json dict"""
    y, s = [], []
    for sample, unc in uncertainties.items():
        y.append(1 if is_ood(f"{sample.split('.')[0]}.npy") else 0)
        s.append(unc[aggregation]["max_score"])
    return y, s


def ood_auroc(exp_dataloader, is_ood=is_ood_toy):
    """the AUROC half of ood_detection (auroc.py:95-139) for every aggregated_<unc>.json; the active-learning split
    files that decide `is_ood` for the non-toy data sets are outside this build (pass your own predicate)"""
    res = {"mean": {}}
    for unc, path in exp_dataloader.get_aggregated_unc_files_dict().items():
        with open(path) as f:
            uncertainties = json.load(f)
        res["mean"][unc] = {}
        for aggregation in exp_dataloader.exp_version.aggregations:
            y, s = get_auroc_input(uncertainties, aggregation, is_ood)
            res["mean"][unc][aggregation] = {"metrics": {"auroc": roc_auc(y, s)}}
    return res
