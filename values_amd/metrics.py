"""Per-image segmentation metrics of the reference's test loop, computed from two device reductions.

Mirrors `calculate_test_metrics` and `calculate_ged` of uncertainty_modeling/test_3D.py:250-358 (same names,
arguments, result keys).  The reference evaluates them with torchmetrics' `dice` (torchmetrics==0.11.4,
functional/classification/dice.py; average="micro", mdmc_average="global", zero_division=0) on repeated tensors --
T*R + T*T + R*R + 2*T*R mask comparisons per image.  Here ONE pass (`vx_mask_agreement`) counts, for every pair of
masks, the voxels on which both carry class c; every Dice is then a ratio of sums of these integers:
    tp = sum_c I[a][b][c],  fp = sum_c (I[a][a][c] - I[a][b][c]),  fn = sum_c (I[b][b][c] - I[a][b][c])
with c running over the classes that survive `ignore_index` (torchmetrics deletes that one-hot column for micro
averaging).  SoftDiceLoss + NLLLoss (loss_modules.py:7-97) come from `vx_soft_metric_sums`.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import _lib


def mask_agreement(masks: torch.Tensor, num_classes: int) -> np.ndarray:
    """masks (M, *spatial) integer labels on the device -> counts (M, M, C) int64 (host)."""
    _lib.require_gpu()
    lib = _lib.load()
    m = masks.reshape(masks.shape[0], -1).to(torch.uint8).contiguous()
    M, nvox = m.shape
    out = torch.empty((M, M, num_classes), dtype=torch.int64, device=m.device)
    _lib.check(lib.vx_mask_agreement(m.data_ptr(), M, num_classes, nvox, out.data_ptr(), _lib.stream_ptr()), "vx_mask_agreement")
    return out.cpu().numpy()


def _micro_dice(I: np.ndarray, a_idx, b_idx, classes) -> float:
    """torchmetrics dice(average='micro', mdmc_average='global', zero_division=0) pooled over the listed (a, b) pairs."""
    tp = fp = fn = 0
    for a, b in zip(a_idx, b_idx):
        iab = int(I[a, b, classes].sum())
        tp += iab
        fp += int(I[a, a, classes].sum()) - iab
        fn += int(I[b, b, classes].sum()) - iab
    den = 2 * tp + fp + fn
    return 0.0 if den == 0 else 2.0 * tp / den


def _classes(num_classes: int, ignore_index: Optional[int]):
    return [c for c in range(num_classes) if c != ignore_index]


def _to_dev(t, dev):
    return t.to(dev) if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t), device=dev)


def calculate_test_metrics(output_softmax: torch.Tensor, ground_truth: torch.Tensor) -> Dict:
    """output_softmax (1, C, *spatial) mean softmax; ground_truth (R, *spatial) integer -> {"loss", "dice"}
    (test_3D.py:250-281: per rater SoftDiceLoss + NLLLoss and Dice(ignore_index=0), averaged over raters)."""
    _lib.require_gpu()
    lib = _lib.load()
    dev = output_softmax.device if output_softmax.is_cuda else torch.device("cuda", torch.cuda.current_device())
    p = _to_dev(output_softmax, dev).to(torch.float32)
    if p.shape[0] != 1:
        raise ValueError("calculate_test_metrics: output_softmax must be (1, C, ...) -- the mean prediction")
    C = p.shape[1]
    p = p[0].reshape(C, -1).contiguous()
    gt = _to_dev(ground_truth, dev)
    R = gt.shape[0]
    g8 = gt.reshape(R, -1).to(torch.uint8).contiguous()
    nvox = p.shape[1]
    sums = torch.empty((R, 3 * C + 1), dtype=torch.float64, device=dev)
    ws = torch.empty(max(int(lib.vx_soft_metric_workspace_bytes(C, R)), 8), dtype=torch.uint8, device=dev)
    _lib.check(lib.vx_soft_metric_sums(p.data_ptr(), g8.data_ptr(), C, R, nvox, sums.data_ptr(), ws.data_ptr(),
                                       _lib.stream_ptr()), "vx_soft_metric_sums")
    # hard Dice of the arg-max of the mean prediction against every rater
    # (the arg-max itself is one of vx_unc_reduce's outputs; recomputed here from p for a self-contained signature)
    from .uncertainty import uncertainty_maps
    am = uncertainty_maps(p.reshape(1, 1, C, nvox), from_logits=False)["argmax"].reshape(1, nvox)
    I = mask_agreement(torch.cat([am, g8], 0), C)
    s = sums.cpu().numpy()
    smooth = 1e-5
    losses, dices = [], []
    cls = _classes(C, 0)
    for r in range(R):
        inter, cnt, psum = s[r, 0:3 * C:3], s[r, 1:3 * C:3], s[r, 2:3 * C:3]
        soft = np.mean(-((2.0 * inter + smooth) / ((psum + cnt) + smooth)))      # soft_dice, B = 1
        nll = -s[r, 3 * C] / nvox                                                 # NLLLoss mean reduction
        losses.append(soft + nll)
        dices.append(_micro_dice(I, [0], [1 + r], cls))
    return {"loss": float(np.mean(np.array(losses))), "dice": float(np.mean(np.array(dices)))}


def calculate_ged(output_softmax: torch.Tensor, ground_truth: torch.Tensor, ignore_index: int = 0, ged_only: bool = False,
                  pred_masks: Optional[torch.Tensor] = None) -> Dict:
    """output_softmax (T, C, *spatial); ground_truth (R, *spatial) -> {"ged", "max dice rater i", "max dice pred"}
    (test_3D.py:284-358).  pred_masks (T, *spatial): the per-sample arg-max masks when the caller already has them
    (vx_unc_reduce's sample_argmax), otherwise they are taken from output_softmax."""
    _lib.require_gpu()
    dev = output_softmax.device if output_softmax.is_cuda else torch.device("cuda", torch.cuda.current_device())
    sm = _to_dev(output_softmax, dev)
    T, C = sm.shape[0], sm.shape[1]
    gt = _to_dev(ground_truth, dev)
    R = gt.shape[0]
    if pred_masks is None:
        from .uncertainty import uncertainty_maps
        nvox = sm[0, 0].numel()
        pred_masks = uncertainty_maps(sm.reshape(1, T, C, nvox).to(torch.float32), from_logits=False,
                                      want_sample_argmax=True)["sample_argmax"].reshape(T, nvox)
    pm = _to_dev(pred_masks, dev).reshape(T, -1).to(torch.uint8)
    g8 = gt.reshape(R, -1).to(torch.uint8)
    I = mask_agreement(torch.cat([pm, g8], 0), C)
    P = list(range(T))
    G = [T + r for r in range(R)]
    cls = _classes(C, ignore_index)
    # pooled distances over the repeated tensors of :290-320 (order of a pair does not change pooled tp / fp+fn)
    d_gp = 1.0 - _micro_dice(I, [p for _ in G for p in P], [g for g in G for _ in P], cls)
    d_pp = 1.0 - _micro_dice(I, [a for a in P for _ in P], [b for _ in P for b in P],
                             cls if ignore_index == 0 else _classes(C, None))
    gt_has_ignored = bool(ignore_index is not None and 0 <= ignore_index < C and any(I[g, g, ignore_index] > 0 for g in G))
    d_gg = 1.0 - _micro_dice(I, [a for a in G for _ in G], [b for _ in G for b in G],
                             cls if gt_has_ignored else _classes(C, None))
    out = {"ged": float(2 * d_gp - d_pp - d_gg)}
    if R > 1 and not ged_only:
        pair = np.array([[np.float32(_micro_dice(I, [p], [g], cls)) for g in G] for p in P], dtype=np.float32)
        # per rater: best prediction (starts from 0, strict >; :326-337); per prediction: best rater, averaged
        for r in range(R):
            out["max dice rater {}".format(r)] = float(max(np.float32(0), pair[:, r].max()))
        out["max dice pred"] = float(np.float32(sum(max(np.float32(0), pair[p].max()) for p in range(T))) / np.float32(T))
    return out
