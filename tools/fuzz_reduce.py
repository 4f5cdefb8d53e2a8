#!/usr/bin/env python3
"""Randomised shapes through the reductions: uncertainty maps (T, C, odd voxel counts, f32 / f64, logits / probs) against
a float64 torch restatement of test_3D.py:486-518, the radix-select quantile against np.quantile on float64 data, and
the mask-agreement counts / GED against a direct evaluation.     python tools/fuzz_reduce.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from values_amd.uncertainty import uncertainty_maps, softmax_variance
from values_amd.thresholds import quantile, count_nonzero
from values_amd.metrics import mask_agreement

dev = torch.device("cuda", 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    g = torch.Generator().manual_seed(case)
    kind = rng.choice(["unc", "unc", "quantile", "agree", "var"])
    tag = ""
    try:
        if kind in ("unc", "var"):
            B, T, C = rng.randint(1, 4), rng.randint(1, 20), rng.randint(2, 19)
            spatial = tuple(rng.randint(1, 13) for _ in range(rng.choice([1, 2, 3])))
            from_logits = rng.random() < 0.6
            f64 = rng.random() < 0.3 and kind == "unc" and not (from_logits and C > 8)   # (rejected loudly: f32 only)
            x = torch.randn((B, T, C) + spatial, generator=g, dtype=torch.float64) * rng.choice([0.5, 3.0, 30.0])
            if not from_logits:
                x = torch.softmax(x, 2)
            xin = (x if f64 else x.float()).to(dev)
            tag = f"{kind} B={B} T={T} C={C} {spatial} f64={f64} logits={from_logits}"
            p = torch.softmax(xin.double().cpu(), 2) if from_logits else xin.double().cpu()
            if kind == "var":
                got = softmax_variance(xin, from_logits=from_logits).cpu().double()
                ref = p.var(1, unbiased=False).mean(1)
                err = (got - ref).abs().max().item()
                ok = err < 1e-5
            else:
                m = uncertainty_maps(xin, from_logits=from_logits, want_sample_argmax=True)
                mean = p.mean(1)
                pe = -(mean * torch.log(mean.clamp_min(1e-300))).sum(1)
                ee = -(p * torch.log(p.clamp_min(1e-300))).sum(2).mean(1)
                errs = [(m["pred_entropy"].cpu().double() - pe).abs().max().item(),
                        (m["expected_entropy"].cpu().double() - ee).abs().max().item(),
                        (m["mutual_information"].cpu().double() - (pe - ee)).abs().max().item(),
                        (m["mean_softmax"].cpu().double() - mean).abs().max().item()]
                err = max(errs)
                # arg-max: identical outside ties
                top2 = mean.topk(2, 1).values
                clear = (top2[:, 0] - top2[:, 1]) > 1e-5
                am = (m["argmax"].cpu().long() == mean.argmax(1))[clear].all().item()
                t2 = p.topk(2, 2).values
                clear_s = (t2[:, :, 0] - t2[:, :, 1]) > 1e-5
                sam = (m["sample_argmax"].cpu().long() == p.argmax(2))[clear_s].all().item()
                ok = err < (2e-5 if not f64 else 1e-5) and am and sam
            if not ok:
                raise AssertionError(f"err {err:.2e}")
        elif kind == "quantile":
            n = rng.choice([1, 2, 3, 17, 1000, 65537, 300001])
            q = rng.choice([0.0, 1.0, 0.5, 0.95, 0.98, rng.random()])
            v = torch.randn(n, generator=g) * rng.choice([1e-3, 1.0, 1e4])
            if rng.random() < 0.4:
                v = (v * 4).round() / 4          # many ties
            if rng.random() < 0.3:
                v = v.abs()
            tag = f"quantile n={n} q={q}"
            got = quantile(v.to(dev), q)
            ref = float(np.quantile(v.numpy().astype(np.float64), q))
            if got != ref:
                raise AssertionError(f"{got!r} != {ref!r}")
            mask = (v > 0).to(torch.uint8)
            if count_nonzero(mask.to(dev)) != int(mask.sum()):
                raise AssertionError("count_nonzero")
        else:
            M, C = rng.randint(1, 12), rng.randint(2, 6)
            nvox = rng.choice([1, 7, 64, 1000, 4097])
            masks = torch.randint(0, C, (M, nvox), generator=g, dtype=torch.uint8)
            tag = f"agree M={M} C={C} nvox={nvox}"
            I = mask_agreement(masks.to(dev), C)
            ref = np.zeros_like(I)
            mn = masks.numpy()
            for a in range(M):
                for b in range(M):
                    for c in range(C):
                        ref[a, b, c] = np.sum((mn[a] == c) & (mn[b] == c))
            if I.shape != ref.shape or not np.array_equal(I, ref):
                raise AssertionError("counts differ")
    except Exception as e:
        bad += 1
        print(f"FAIL case {case}: {tag}: {type(e).__name__}: {e}")
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
