#!/usr/bin/env python3
"""Randomised volume shapes / widths / class counts through the whole UNet3D forward: split-fp16 (default) against the
native-fp32 kernels on the same hash-dropout draws, and the chunked two-stream run against the one-stream run.
    python tools/fuzz_unet.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from values_amd import UNet3D, predict_uncertainty
from values_amd import _lib

dev = torch.device("cuda", 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    f = rng.choice([8, 8, 8, 16, 4])
    ncls = rng.choice([2, 2, 3, 5, 8, 9, 12, 19])
    dims = [16 * rng.randint(1, 5 if f <= 8 else 3) for _ in range(3)]
    V = rng.randint(1, 6)
    T = rng.choice([1, 2, 5, 10, 16])
    drop = rng.random() < 0.8
    torch.manual_seed(case)
    try:
        m = UNet3D(num_classes=ncls, do_dropout=drop, initial_filter_size=f).to(dev)
    except Exception as e:
        print(f"case {case}: f={f} unsupported: {e}")
        continue
    x = torch.randn((V, 1, *dims), device=dev)
    tag = f"case {case}: f={f} C={ncls} V={V} T={T} {dims} drop={drop}"
    if os.environ.get("FUZZ_VERBOSE"):
        print(tag, flush=True)

    m32 = UNet3D(num_classes=ncls, do_dropout=drop, initial_filter_size=f).to(dev)   # weights are packed per mode:
    m32.load_state_dict(m.state_dict())                                               # one instance per mode

    def run(env, **kw):
        with _lib.config(**{"conv_fp32": 0, **env}):
            o = predict_uncertainty([m32 if env else m], x, n_pred=T, seeds=[7], **kw)
            torch.cuda.synchronize()
            return o

    a = run({}, n_streams=1)
    b = run({"conv_fp32": 1}, n_streams=1)
    c = run({}, n_streams=2)
    err = (a["logits"] - b["logits"]).abs().max().item()
    scale = max(1.0, a["logits"].abs().max().item())
    merr = max((a[k] - b[k]).abs().max().item() for k in ("pred_entropy", "aleatoric_uncertainty", "epistemic_uncertainty", "mean_softmax"))
    nan = any(torch.isnan(a[k].float()).any().item() for k in a)
    first = torch.equal(a["logits"][:1], c["logits"][:1])      # the first chunk draws the same masks
    if nan or err > 2e-4 * scale or merr > 1e-4 or not first:
        bad += 1
        print(f"FAIL {tag}: logits err {err:.2e} (scale {scale:.1f}) maps err {merr:.2e} nan={nan} first-chunk-equal={first}")
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
