"""ORACLE (test infrastructure, not product code) -- CPU restatement of the
reference's per-voxel uncertainty reduction.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this.

Restates /root/reference/uncertainty_modeling/test_3D.py:
  * calculate_uncertainty  (:486-518)
  * calculate_one_minus_msr (:521-525)
and /root/reference/uncertainty_modeling/data_carrier_3D.py:253-293 (mean softmax,
argmax masks).

Semantics that matter for parity:
  - mean over the prediction axis first, natural log;
  - products p*log(p) that are NaN (p == 0) are SKIPPED, not propagated;
  - the accumulators are created with torch.zeros(...) => float32, even when
    the input is float64: every `+=` rounds the running sum to float32;
  - `ssn=True` swaps the aleatoric/epistemic keys.

Parity pin: tests/test_oracle_golden.py vs tests/golden/unc_kat.npz (outputs of
the imported reference functions).
"""
from __future__ import annotations

import numpy as np


def calculate_uncertainty(softmax_preds: np.ndarray, ssn: bool = False) -> dict:
    """softmax_preds: (T, C, *spatial) float32 or float64 -> dict of float32 maps."""
    p = np.asarray(softmax_preds)
    T, C = p.shape[0], p.shape[1]
    spatial = p.shape[2:]
    # torch.mean(dim=0) in the input dtype (test_3D.py:489)
    mean_softmax = p.mean(axis=0, dtype=p.dtype)
    pred_entropy = np.zeros(spatial, dtype=np.float32)  # torch.zeros => float32 (:490)
    with np.errstate(divide="ignore", invalid="ignore"):
        for y in range(C):
            term = mean_softmax[y] * np.log(mean_softmax[y])
            ok = ~np.isnan(term)
            # f32 += f64 rounds the sum to f32 at every class (:494)
            pred_entropy[ok] = (pred_entropy[ok].astype(term.dtype) + term[ok]).astype(np.float32)
        pred_entropy *= np.float32(-1)
        expected = np.zeros((T,) + spatial, dtype=np.float32)
        for t in range(T):
            ent = np.zeros(spatial, dtype=np.float32)
            for y in range(C):
                term = p[t, y] * np.log(p[t, y])
                ok = ~np.isnan(term)
                ent[ok] = (ent[ok].astype(term.dtype) + term[ok]).astype(np.float32)
            ent *= np.float32(-1)
            expected[t] = ent
    expected_entropy = expected.mean(axis=0, dtype=np.float32)
    mutual_information = pred_entropy - expected_entropy
    out = {"pred_entropy": pred_entropy}
    if not ssn:
        out["aleatoric_uncertainty"] = expected_entropy
        out["epistemic_uncertainty"] = mutual_information
    else:  # test_3D.py:513-516
        out["aleatoric_uncertainty"] = mutual_information
        out["epistemic_uncertainty"] = expected_entropy
    return out


def calculate_one_minus_msr(softmax_pred: np.ndarray) -> dict:
    """(C, *spatial) -> {"pred_entropy": 1 - max_c p_c}  (test_3D.py:521-525)."""
    p = np.asarray(softmax_pred)
    return {"pred_entropy": 1 - p.max(axis=0)}


def mean_and_argmax(softmax_preds: np.ndarray):
    """data_carrier_3D.py:253-255, 281-283: mean over T, argmax over C (uint8),
    per-sample argmax (uint8).  np.argmax returns the FIRST maximal class."""
    p = np.asarray(softmax_preds)
    mean = p.mean(axis=0)
    return mean, np.argmax(mean, axis=0).astype(np.uint8), np.argmax(p, axis=1).astype(np.uint8)


def softmax(logits: np.ndarray, axis: int = 1) -> np.ndarray:
    """F.softmax(dim=1) (test_3D.py:435,448,472) in the input dtype."""
    z = logits - logits.max(axis=axis, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=axis, keepdims=True)
