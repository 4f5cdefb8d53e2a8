"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's per-image metrics.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restates /root/reference/uncertainty_modeling/test_3D.py:
  * calculate_test_metrics (:250-281)   SoftDiceLoss (loss_modules.py:7-97) + torch NLLLoss + Dice(ignore_index=0)
  * calculate_ged          (:284-358)   generalised energy distance from pooled Dice scores + max-Dice summaries
The hard Dice is third-party arithmetic that is NOT under /root/reference and NOT installed here:
torchmetrics==0.11.4 (requirements.txt:103), torchmetrics.functional.dice.  Its published algorithm, restated in
`tm_dice` below exactly as that version computes it for the call shapes used:
  1. float predictions (N, C, ...) -> arg-max over C; integer predictions are labels; both sides one-hot over
     max(2, num_classes) classes with all extra dimensions flattened into N (mdmc_average="global");
  2. ignore_index (micro average): that one-hot COLUMN is deleted from predictions and targets;
  3. tp = sum(pred & target), fp = sum(pred & ~target), fn = sum(~pred & target) over everything that is left;
  4. score = 2 tp / (2 tp + fp + fn), and 0 where the denominator is 0 (zero_division=0).
PARITY UNPINNED for the Dice / GED numbers (no torchmetrics here, no golden vector in the reference); SoftDiceLoss
and NLLLoss ARE pinned: tests/golden/metrics_kat.npz holds the outputs of the imported reference SoftDiceLoss and
torch.nn.NLLLoss on formula inputs (tools/gen_golden.py:gen_metrics).
"""
from __future__ import annotations

import numpy as np


def _onehot(labels: np.ndarray, n: int) -> np.ndarray:
    """(N, ...) int -> (N * prod(...), n) bool"""
    flat = labels.reshape(-1)
    return flat[:, None] == np.arange(n)[None, :]


def tm_dice(preds: np.ndarray, target: np.ndarray, ignore_index=None) -> float:
    preds, target = np.asarray(preds), np.asarray(target)
    if np.issubdtype(preds.dtype, np.floating):
        num_classes = preds.shape[1]
        labels = preds.argmax(axis=1)
    else:
        labels = preds
        num_classes = int(max(labels.max(), target.max())) + 1
    n = max(2, num_classes)
    p, t = _onehot(labels, n), _onehot(target, n)
    if ignore_index is not None and 0 <= ignore_index < n:
        keep = [c for c in range(n) if c != ignore_index]
        p, t = p[:, keep], t[:, keep]
    tp = int((p & t).sum()); fp = int((p & ~t).sum()); fn = int((~p & t).sum())
    den = 2 * tp + fp + fn
    return 0.0 if den == 0 else 2.0 * tp / den


def soft_dice_loss(x: np.ndarray, y: np.ndarray, smooth: float = 1e-5) -> float:
    """SoftDiceLoss()(x, y): x (B, C, ...) probabilities, y (B, ...) labels (loss_modules.py:35-97, defaults)."""
    x = np.asarray(x, dtype=np.float64)
    B, C = x.shape[:2]
    onehot = (np.asarray(y)[:, None] == np.arange(C).reshape((1, C) + (1,) * (x.ndim - 2)))
    axes = tuple(range(2, x.ndim))
    inter = (x * onehot).sum(axes)
    denom = (x + onehot).sum(axes)
    return float((-((2 * inter + smooth) / (denom + smooth))).mean())


def nll_loss(logp: np.ndarray, y: np.ndarray) -> float:
    """torch.nn.NLLLoss()(logp (B, C, ...), y (B, ...)) with mean reduction."""
    logp = np.asarray(logp, dtype=np.float64)
    picked = np.take_along_axis(logp, np.asarray(y)[:, None].astype(np.int64), axis=1)
    return float(-picked.mean())


def calculate_test_metrics(output_softmax: np.ndarray, ground_truth: np.ndarray) -> dict:
    losses, dices = [], []
    for r in range(ground_truth.shape[0]):
        gt = ground_truth[r][None].astype(np.int64)
        losses.append(soft_dice_loss(output_softmax, gt) + nll_loss(np.log(output_softmax), gt))
        dices.append(tm_dice(output_softmax, gt, ignore_index=0))
    return {"loss": float(np.mean(losses)), "dice": float(np.mean(dices))}


def calculate_ged(output_softmax: np.ndarray, ground_truth: np.ndarray, ignore_index: int = 0, ged_only: bool = False) -> dict:
    T, R = output_softmax.shape[0], ground_truth.shape[0]
    gt_repeat = np.repeat(ground_truth, T, axis=0)
    pred_repeat = np.tile(output_softmax, (R,) + (1,) * (output_softmax.ndim - 1))
    d_gp = 1 - tm_dice(pred_repeat, gt_repeat, ignore_index=ignore_index)
    am = output_softmax.argmax(axis=1)
    d_pp = 1 - tm_dice(np.repeat(am, T, axis=0), np.tile(am, (T,) + (1,) * (am.ndim - 1)),
                       ignore_index=ignore_index if ignore_index == 0 else None)
    g1, g2 = np.repeat(ground_truth, R, axis=0), np.tile(ground_truth, (R,) + (1,) * (ground_truth.ndim - 1))
    d_gg = 1 - (tm_dice(g1, g2, ignore_index=ignore_index) if np.any(g1 == ignore_index) else tm_dice(g1, g2))
    out = {"ged": 2 * d_gp - d_pp - d_gg}
    if R > 1 and not ged_only:
        pair = np.array([[np.float32(tm_dice(output_softmax[p][None], ground_truth[r][None], ignore_index=ignore_index))
                          for r in range(R)] for p in range(T)], dtype=np.float32)
        for r in range(R):
            out["max dice rater {}".format(r)] = float(max(np.float32(0), pair[:, r].max()))
        out["max dice pred"] = float(np.float32(sum(max(np.float32(0), pair[p].max()) for p in range(T))) / np.float32(T))
    return out
