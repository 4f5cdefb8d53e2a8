"""ORACLE (test infrastructure, not product code) -- numpy restatement of the evaluation stage's downstream scalars.

Only tests/ may import this.  Restates /root/reference/evaluation/metrics:
  * aurc.py:14-67        rc_curve_stats / aurc / eaurc            (fd-shifts' risk-coverage statistics)
  * ncc.py:9-25          compute_ncc
  * ace.py:44-90         platt_scale_confid / calib_stats / calc_ace   (sklearn's column_or_1d / label_binarize /
                         np.digitize / np.bincount spelled out)
  * ace.py:13-41         the Platt fit: sklearn.calibration._sigmoid_calibration (scikit-learn 1.2.2 pinned by the
                         reference, requirements.txt:88).  Third-party and version dependent in its OPTIMISER (fmin_bfgs in
                         1.2.2, L-BFGS-B from 1.4), not in what it minimises: Platt's regularised cross entropy, convex in
                         (A, B).  Restated here as that objective + Newton iterations to its unique minimum; pinned to the
                         installed scikit-learn's result to the optimiser's own tolerance (tests/golden/evalmetrics_kat.npz).
  * auroc.py:79-139      get_auroc_input + sklearn.metrics.roc_curve / auc (trapezoid over the distinct thresholds)

Parity pin: tests/test_oracle_golden.py checks every function against tests/golden/evalmetrics_kat.npz, which
tools/gen_golden.py produced by importing the reference modules themselves.
"""
from __future__ import annotations

import numpy as np


def rc_curve_stats(risks, confids):
    """aurc.py:14-51"""
    risks, confids = np.asarray(risks, dtype=np.float64), np.asarray(confids, dtype=np.float64)
    n = len(risks)
    idx = np.argsort(confids)
    coverage, error_sum = n, float(sum(risks[idx]))
    coverages, selective, weights = [coverage / n], [error_sum / n], []
    tmp = 0
    for i in range(0, n - 1):
        coverage -= 1
        error_sum -= risks[idx[i]]
        tmp += 1
        if i == 0 or confids[idx[i]] != confids[idx[i - 1]]:
            coverages.append(coverage / n)
            selective.append(error_sum / (n - 1 - i))
            weights.append(tmp / n)
            tmp = 0
    if tmp > 0:
        coverages.append(0)
        selective.append(selective[-1])
        weights.append(tmp / n)
    return coverages, selective, weights


def aurc(risks, confids):
    """aurc.py:54-58"""
    _, r, w = rc_curve_stats(risks, confids)
    return sum((r[i] + r[i + 1]) * 0.5 * w[i] for i in range(len(w)))


def eaurc(risks, confids):
    """aurc.py:61-67"""
    risks = np.asarray(risks, dtype=np.float64)
    n = len(risks)
    sel = np.sort(risks).cumsum() / np.arange(1, n + 1)
    return aurc(risks, confids) - sel.sum() / n


def compute_ncc(gt, pred):
    """ncc.py:9-25"""
    gt, pred = np.asarray(gt), np.asarray(pred)
    mg, mp = np.mean(gt), np.mean(pred)
    sg, sp = np.std(gt, ddof=1), np.std(pred, ddof=1)
    return (1 / (np.size(gt) * sg * sp)) * np.sum(np.multiply(gt - mg, pred - mp))


def rater_correct(reference_segs, pred_seg, unc_map, ignore_value=None):
    """ace.py:19-34 / :104-121: flat (F = -unc repeated per rater, correct) with ignored voxels dropped"""
    reference_segs = np.asarray(reference_segs)
    pred = np.repeat(np.asarray(pred_seg)[np.newaxis, :], reference_segs.shape[0], 0)
    unc = np.repeat(np.asarray(unc_map)[np.newaxis, :], reference_segs.shape[0], 0)
    correct = (reference_segs == pred).astype(int)
    if ignore_value is not None:
        keep = reference_segs != ignore_value
        return -unc[keep], correct[keep]
    return -unc.flatten(), correct.flatten()


def sigmoid_calibration(F, y, iters=100):
    """Platt (2000) / sklearn.calibration._sigmoid_calibration: minimise -(T log P + (1 - T) log(1 - P)),
    P = 1 / (1 + exp(A F + B)), with the Bayesian targets T; Newton with step halving to the unique optimum."""
    F = np.asarray(F, dtype=np.float64)
    y = np.asarray(y)
    prior0 = float(np.sum(y <= 0))
    prior1 = y.shape[0] - prior0
    T = np.where(y > 0, (prior1 + 1.0) / (prior1 + 2.0), 1.0 / (prior0 + 2.0))

    def parts(A, B):
        z = A * F + B
        e = np.exp(-np.abs(z))
        P = np.where(z >= 0, e / (1 + e), 1 / (1 + e))
        loss = np.where(z >= 0, T * z + np.log1p(e), (T - 1) * z + np.log1p(e)).sum()
        d, w = T - P, P * (1 - P)
        return loss, np.array([d @ F, d.sum()]), np.array([[w @ (F * F), w @ F], [w @ F, w.sum()]])

    A, B = 0.0, np.log((prior0 + 1.0) / (prior1 + 1.0))
    loss, g, H = parts(A, B)
    for _ in range(iters):
        if np.abs(g).max() < 1e-10 * max(1.0, len(F)):
            break
        step = np.linalg.solve(H + 1e-12 * np.eye(2), g)
        t = 1.0
        while True:
            l2, g2, H2 = parts(A - t * step[0], B - t * step[1])
            if l2 <= loss + 1e-12 * abs(loss) or t < 1e-10:
                break
            t *= 0.5
        A, B, loss, g, H = A - t * step[0], B - t * step[1], l2, g2, H2
    return A, B


def platt_scale_confid(uncalib_confid, a, b):
    """ace.py:44-48"""
    return 1 / (1 + np.exp(np.asarray(uncalib_confid, dtype=np.float64) * a + b))


def calib_stats(correct, calib_confids, n_bins=20):
    """ace.py:51-82"""
    y_true = np.ravel(correct)
    y_prob = np.ravel(calib_confids)
    if y_prob.min() < 0 or y_prob.max() > 1:
        raise ValueError("y_prob has values outside [0, 1] and normalize is set to False.")
    labels = np.unique(y_true)
    if len(labels) > 2:
        raise ValueError(f"Only binary classification is supported. Provided labels {labels}.")
    # sklearn.preprocessing.label_binarize(y, classes=labels)[:, 0]: one class -> a column of zeros (neg_label);
    # two classes -> 1 where y equals the larger label
    y_bin = np.zeros(len(y_true)) if len(labels) < 2 else (y_true == labels[1]).astype(np.float64)
    bins = np.linspace(0.0, 1.0 + 1e-8, n_bins + 1)
    binids = np.digitize(y_prob, bins) - 1
    bin_sums = np.bincount(binids, weights=y_prob, minlength=len(bins))
    bin_true = np.bincount(binids, weights=y_bin, minlength=len(bins))
    bin_total = np.bincount(binids, minlength=len(bins))
    nz = bin_total != 0
    prob_true, prob_pred = bin_true[nz] / bin_total[nz], bin_sums[nz] / bin_total[nz]
    return np.abs(prob_true - prob_pred), bin_total[nz] / bin_total.sum(), int(nz.sum())


def calc_ace(correct, calib_confids):
    """ace.py:85-87"""
    d, _, k = calib_stats(correct, calib_confids)
    return (1 / k) * np.sum(d)


def roc_auc(y_true, y_score):
    """sklearn.metrics.roc_curve + auc as auroc.py:126-127 calls them: thresholds at the distinct scores (descending),
    cumulative true / false positives, trapezoid (dropping collinear points does not change the area)"""
    y_true = np.asarray(y_true)
    y_score = np.asarray(y_score, dtype=np.float64)
    order = np.argsort(-y_score, kind="mergesort")
    ys, yt = y_score[order], (y_true[order] == 1)
    distinct = np.where(np.diff(ys))[0]
    idx = np.r_[distinct, len(ys) - 1]
    tps = np.cumsum(yt)[idx].astype(np.float64)
    fps = (1 + idx - tps).astype(np.float64)
    tps, fps = np.r_[0, tps], np.r_[0, fps]
    if tps[-1] <= 0 or fps[-1] <= 0:
        return float("nan")
    tpr, fpr = tps / tps[-1], fps / fps[-1]
    return float(np.sum((fpr[1:] - fpr[:-1]) * (tpr[1:] + tpr[:-1]) * 0.5))
