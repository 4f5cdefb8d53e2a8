#!/usr/bin/env python3
"""Randomised inputs through values_amd.metrics / values_amd.aggregation against the CPU oracle restatements
(oracle/metrics_oracle.py, oracle/aggregation_oracle.py -- which is why this checker lives under tests/).   python tests/fuzz/fuzz_metrics.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import aggregation_oracle as AO, metrics_oracle as MO
from values_amd import aggregation as AG, metrics as M

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    g = torch.Generator().manual_seed(case)
    kind = rng.choice(["ged", "test", "patch", "image", "thr"])
    tag = kind
    try:
        if kind in ("ged", "test"):
            C, R = rng.randint(2, 5), rng.randint(1, 4)
            spatial = tuple(rng.randint(1, 9) for _ in range(rng.choice([2, 3])))
            T = rng.randint(1, 6) if kind == "ged" else 1
            sm = torch.softmax(torch.randn((T, C) + spatial, generator=g) * 2, 1)
            gt = torch.randint(0, C, (R,) + spatial, generator=g)
            if rng.random() < 0.2:
                gt[:] = 0                      # empty foreground
            tag = f"{kind} T={T} C={C} R={R} {spatial}"
            if kind == "ged":
                ii = rng.choice([0, 0, 1])
                got = M.calculate_ged(sm, gt, ignore_index=ii)
                ref = MO.calculate_ged(sm.numpy(), gt.numpy(), ignore_index=ii)
            else:
                got = M.calculate_test_metrics(sm, gt)
                ref = MO.calculate_test_metrics(sm.numpy(), gt.numpy())
            assert set(got) == set(ref), (sorted(got), sorted(ref))
            for k in ref:
                a, b = float(got[k]), float(ref[k])
                assert (np.isnan(a) and np.isnan(b)) or abs(a - b) < 2e-5, (k, a, b)
        else:
            dims = rng.choice([2, 3])
            shape = tuple(rng.randint(10, 40) for _ in range(dims))
            img = (torch.rand(shape, generator=g) ** 3).numpy().astype(np.float32)
            mean = rng.random() < 0.5
            if kind == "patch":
                ps = [10] * dims
                tag = f"patch {shape} mean={mean}"
                got = AG.patch_level_aggregation(img, ps, mean=mean)
                ref = AO.patch_level_aggregation(img, ps, mean=mean)
            elif kind == "image":
                tag = f"image {shape} mean={mean}"
                got = AG.image_level_aggregation(img, mean=mean)
                ref = AO.image_level_aggregation(img, mean=mean)
            else:
                thr = float(np.quantile(img, rng.choice([0.5, 0.9, 0.99])))
                tag = f"thr {shape} mean={mean} thr={thr:.3f}"
                got = AG.threshold_aggregation(img, threshold=thr, mean=mean)
                ref = AO.threshold_aggregation(img, threshold=thr, mean=mean)
            if not isinstance(ref, dict):          # (the reference returns a bare score from some of these)
                assert not isinstance(got, dict)
                got, ref = {"score": got}, {"score": ref}
            for k in ref:
                if isinstance(ref[k], (int, float, np.floating)):
                    assert abs(float(got[k]) - float(ref[k])) <= 2e-6 * max(1.0, abs(float(ref[k]))), (k, got[k], ref[k])
                else:
                    assert np.array_equal(np.asarray(got[k]), np.asarray(ref[k])), (k, got[k], ref[k])
    except Exception as e:
        bad += 1
        print(f"FAIL case {case}: {tag}: {type(e).__name__}: {e}")
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
