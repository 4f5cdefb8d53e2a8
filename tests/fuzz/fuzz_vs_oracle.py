#!/usr/bin/env python3
"""Randomised volume shapes, widths, class counts and injected dropout masks: the HIP forward + fused reduction against
the float64 CPU oracle (oracle/unet3d_oracle.py -- test infrastructure, which is why this checker lives under tests/).
    python tests/fuzz/fuzz_vs_oracle.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.set_num_threads(16)
from oracle.unet3d_oracle import DROPOUT_ORDER, unet3d_forward
from values_amd import UNet3D, predict_uncertainty
from tests.formula import formula_unet3d_state_dict

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    f = rng.choice([8, 8, 16])
    ncls = rng.choice([2, 3, 5, 9])
    dims = [16 * rng.randint(1, 3 if f == 8 else 2) for _ in range(3)]
    V, T = rng.randint(1, 2), rng.randint(1, 3)
    drop = rng.random() < 0.7
    sd = {k: torch.from_numpy(v) for k, v in formula_unet3d_state_dict(seed_tag=10 + case, num_classes=ncls, f=f).items()}
    m = UNet3D(num_classes=ncls, initial_filter_size=f, do_dropout=drop)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    m = m.cuda()
    g = torch.Generator().manual_seed(case)
    x = torch.randn((V, 1, *dims), generator=g, dtype=torch.float64)
    tag = f"case {case}: f={f} C={ncls} V={V} T={T} {dims} drop={drop}"
    # channel counts / resolutions of the 17 dropout sites
    lv = [(f * 2 ** l, [d // 2 ** l for d in dims]) for l in range(5)]
    site = {"contr_1_1": 0, "contr_1_2": 0, "contr_2_1": 1, "contr_2_2": 1, "contr_3_1": 2, "contr_3_2": 2,
            "contr_4_1": 3, "contr_4_2": 3, "center": 3, "expand_4_1": 3, "expand_4_2": 3, "expand_3_1": 2,
            "expand_3_2": 2, "expand_2_1": 1, "expand_2_2": 1, "expand_1_1": 0, "expand_1_2": 0}
    ref = torch.empty((V, T, ncls, *dims), dtype=torch.float64)
    masks_dev = None
    if drop:
        per = [[{n: torch.rand((1, lv[site[n]][0], *lv[site[n]][1]), generator=g) < 0.5 for n in DROPOUT_ORDER}
                for t in range(T)] for v in range(V)]
        for v in range(V):
            for t in range(T):
                ref[v, t] = unet3d_forward(sd, x[v:v + 1], masks=per[v][t])[0]
        # device layout: 17 masks, each (V*T, C, D,H,W), sample n = v*T + t
        masks_dev = [torch.cat([per[v][t][n] for v in range(V) for t in range(T)], 0) for n in DROPOUT_ORDER]
    else:
        r = unet3d_forward(sd, x, masks=None)
        for t in range(T):
            ref[:, t] = r
    kw = {"dropout_masks": [masks_dev]} if drop else {}
    out = predict_uncertainty([m], x.float().cuda(), n_pred=T, **kw)
    err = (out["logits"].cpu().double() - ref).abs().max().item()
    p = torch.softmax(ref, 2)
    mean = p.mean(1)
    pe = -(mean * torch.log(mean)).sum(1)
    ee = -(p * torch.log(p)).sum(2).mean(1)
    merr = max((out["pred_entropy"].cpu().double() - pe).abs().max().item(),
               (out["aleatoric_uncertainty"].cpu().double() - ee).abs().max().item(),
               (out["epistemic_uncertainty"].cpu().double() - (pe - ee)).abs().max().item(),
               (out["mean_softmax"].cpu().double() - mean).abs().max().item())
    top2 = mean.topk(2, 1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    seg_ok = (out["pred_seg_mean"].cpu().long() == mean.argmax(1))[clear].all().item()
    ok = err < 1e-4 and merr < 1e-4 and seg_ok
    print(("ok   " if ok else "FAIL ") + f"{tag}: logits {err:.2e} maps {merr:.2e} seg={seg_ok}", flush=True)
    bad += 0 if ok else 1
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
