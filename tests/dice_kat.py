"""Hand-computable known answers for the hard Dice / GED of test_3D.py:250-358.

The arithmetic behind those numbers is torchmetrics==0.11.4 `dice` (requirements.txt:103; absent from /root/reference
and from this image), so the row stays PARITY UNPINNED: no output of that package can be generated here.  What CAN be
pinned is its documented definition -- micro average over everything left after the `ignore_index` column is deleted,
score = 2 tp / (2 tp + fp + fn), 0 where the denominator is 0 -- on label volumes small enough to count by hand.  Both the
oracle (tests/test_oracle_golden.py) and the product (tests/test_gpu_results.py) must give exactly these numbers.

Each case: labels of T predictions (as one-hot "softmax" volumes), R raters, and the closed-form answers.
"""
import numpy as np

S = (2, 4, 4)          # 32 voxels


def _vol(ones):
    """binary label volume with label 1 on the listed flat voxel indices"""
    v = np.zeros(32, dtype=np.int64)
    v[list(ones)] = 1
    return v.reshape(S)


def onehot(labels, C):
    """(T, *S) labels -> (T, C, *S) float32 probabilities whose arg-max is the label"""
    lab = np.asarray(labels)
    return (lab[:, None] == np.arange(C).reshape((1, C) + (1,) * (lab.ndim - 1))).astype(np.float32)


def cases():
    out = []
    A = _vol(range(0, 8))            # 8 foreground voxels
    B = _vol(range(4, 12))           # 8 voxels, 4 shared with A
    D = _vol(range(16, 24))          # 8 voxels, disjoint from A and B
    E = _vol([])                     # empty
    # --- with ignore_index = 0 on two classes the micro Dice is the foreground Dice 2 |P n G| / (|P| + |G|)
    out.append(dict(name="perfect overlap", preds=[A], gts=[A], C=2, dice=1.0, ged=0.0))
    out.append(dict(name="disjoint", preds=[A], gts=[D], C=2, dice=0.0, ged=2.0 * 1.0 - 0.0 - 0.0))
    out.append(dict(name="half overlap", preds=[A], gts=[B], C=2, dice=2 * 4 / 16.0, ged=2 * (1 - 0.5) - 0.0 - 0.0))
    # empty prediction, non-empty rater: tp = fp = 0, fn = 8 -> 0;  d(pred, pred) pools an EMPTY foreground: 0 / 0 -> Dice 0
    # -> distance 1 (zero_division = 0: the quirk that makes GED of an all-background sample smaller)
    out.append(dict(name="empty prediction", preds=[E], gts=[A], C=2, dice=0.0, ged=2 * 1.0 - 1.0 - 0.0))
    # both empty: every Dice is 0 / 0 -> 0, every distance 1
    out.append(dict(name="empty both", preds=[E], gts=[E], C=2, dice=0.0, ged=2 * 1.0 - 1.0 - 1.0))
    # --- two predictions, two raters: pooled counts over the T * R (and T * T, R * R) pairs
    #  pairs (p, g): (A,A) tp 8 fp 0 fn 0; (A,B) tp 4 fp 4 fn 4; (B,A) 4/4/4; (B,B) 8/0/0 -> tp 24, fp 8, fn 8: Dice 48/64
    #  pred-pred and rater-rater pools are the same multiset: Dice 0.75 -> GED = 2 * 0.25 - 0.25 - 0.25 = 0
    out.append(dict(name="two by two", preds=[A, B], gts=[A, B], C=2, dice=None, ged=0.0,
                    max_dice_rater=[1.0, 1.0], max_dice_pred=1.0))
    #  predictions A, A against raters B, D: (A,B) 4/4/4 twice, (A,D) 0/8/8 twice -> tp 8, fp 24, fn 24: Dice 16/64 = 0.25
    #  pred-pred: four (A,A) -> 1 -> distance 0;  rater-rater: (B,B) 8/0/0, (B,D) 0/8/8, (D,B) 0/8/8, (D,D) 8/0/0 -> 32/64
    out.append(dict(name="identical predictions, disagreeing raters", preds=[A, A], gts=[B, D], C=2, dice=None,
                    ged=2 * 0.75 - 0.0 - 0.5, max_dice_rater=[0.5, 0.0], max_dice_pred=0.5))
    # --- three classes, ignore_index = 0: classes 1 and 2 pooled
    P3 = np.zeros(32, dtype=np.int64); P3[0:8] = 1; P3[8:16] = 2
    G3 = np.zeros(32, dtype=np.int64); G3[0:4] = 1; G3[4:8] = 2; G3[8:16] = 2; G3[16:20] = 1
    # class 1: P 8, G 8 (0..3, 16..19), tp 4, fp 4, fn 4;  class 2: P 8, G 12 (4..15), tp 8, fp 0, fn 4 -> 2 * 12 / (24 + 4 + 8)
    out.append(dict(name="three classes", preds=[P3.reshape(S)], gts=[G3.reshape(S)], C=3, dice=24 / 36.0,
                    ged=2 * (1 - 24 / 36.0) - 0.0 - 0.0))
    return out


def no_ignore_case():
    """ignore_index = None on labels: every wrong voxel is one fp and one fn -> micro Dice = accuracy"""
    P3 = np.zeros(32, dtype=np.int64); P3[0:8] = 1; P3[8:16] = 2
    G3 = np.zeros(32, dtype=np.int64); G3[0:4] = 1; G3[4:8] = 2; G3[8:16] = 2; G3[16:20] = 1
    wrong = int((P3 != G3).sum())          # voxels 4..7 and 16..19
    return P3.reshape((1,) + S), G3.reshape((1,) + S), (32 - wrong) / 32.0
