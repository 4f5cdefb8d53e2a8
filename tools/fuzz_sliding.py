#!/usr/bin/env python3
"""Randomised image shapes / patch sizes / overlaps through predict_image_sliding: crop list, softmax accumulation
(overlapping patches) and the normalised maps against a float64 torch restatement built from the same model's
per-patch logits.       python tools/fuzz_sliding.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from values_amd import UNet3D, predict_image_sliding
from values_amd.predict import crop_indices

dev = torch.device("cuda", 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 25
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
torch.manual_seed(3)
models = {c: UNet3D(num_classes=c, do_dropout=False).to(dev) for c in (2, 3)}
for case in range(cases):
    P = rng.choice([16, 32])
    ov = rng.choice([1, 0.5, 0.75, 0.25])
    dims = [rng.randint(P, 3 * P + 5) for _ in range(3)]
    C = rng.choice([2, 3])
    T = rng.randint(1, 3)
    pb = rng.choice([1, 3, 8])
    m = models[C]
    img = torch.randn(dims, generator=torch.Generator().manual_seed(case)).to(dev)
    tag = f"case {case}: image {dims} patch {P} overlap {ov} C={C} T={T} batch={pb}"
    try:
        out = predict_image_sliding([m], img, patch_size=P, patch_overlap=ov, n_pred=T, patch_batch=pb, compat=False)
        crops = crop_indices(tuple(dims), P, ov)
        ssum = torch.zeros((T, C, *dims), dtype=torch.float64)
        cnt = torch.zeros(dims, dtype=torch.float64)
        for c in crops:
            sl = tuple(slice(a, b) for a, b in c)
            lg = m(img[sl][None, None]).double().cpu()[0]          # (C, P,P,P)
            p = torch.softmax(lg, 0)
            for t in range(T):
                ssum[(t, slice(None)) + sl] += p
            cnt[sl] += 1
        e1 = (out["softmax_sum"].cpu().double() - ssum).abs().max().item()
        e2 = (out["num_predictions"].cpu().double() - cnt).abs().max().item()
        pn = ssum / cnt.clamp(min=1)
        mean = pn.mean(0)
        logm = torch.where(mean > 0, torch.log(mean.clamp_min(1e-300)), torch.zeros_like(mean))
        pe = -(mean * logm).sum(0)
        e3 = (out["pred_entropy"].cpu().double() - pe).abs().max().item()
        e4 = (out["mean_softmax"].cpu().double() - mean).abs().max().item()
        ok = e1 < 2e-5 and e2 == 0 and e3 < 2e-5 and e4 < 2e-5 and len(crops) > 0
        if not ok or os.environ.get("FUZZ_VERBOSE"):
            print(("ok   " if ok else "FAIL ") + f"{tag}: {len(crops)} crops, sum {e1:.2e} count {e2} pe {e3:.2e} mean {e4:.2e}", flush=True)
        bad += 0 if ok else 1
    except Exception as e:
        bad += 1
        print(f"ERROR {tag}: {type(e).__name__}: {e}")
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
