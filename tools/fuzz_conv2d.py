#!/usr/bin/env python3
"""Randomised shapes through vx_conv2d (HRNet path): split-fp16 (default) and native-fp32 kernels against a float64
F.conv2d on the device's host, statistics partials included.     python tools/fuzz_conv2d.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
torch.set_num_threads(16)
from tests.test_gpu_kernels2d import run_conv2d
from values_amd import _lib

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    ks, s = rng.choice([(3, 1), (3, 1), (3, 2), (1, 1)])
    cin = rng.choice([3, 5, 8, 16, 17, 18, 24, 25, 32, 36, 48, 64, 72, 96, 144, 192, 270]) if ks == 3 else rng.choice([16, 18, 48, 64, 96, 256, 270, 720])
    cout = rng.choice([16, 18, 32, 36, 48, 64, 72, 96, 144, 19, 4, 24, 128] + ([240, 320, 720] if ks == 1 else []))
    n = rng.randint(1, 3)
    h, w = rng.randint(1, 40), rng.randint(1, 70)
    g = torch.Generator().manual_seed(case)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, ks, ks), generator=g) * (1.0 / (ks * ks * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.2 if rng.random() < 0.4 else None
    ref = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), stride=s, padding=ks // 2)
    tag = f"case {case}: {cin}->{cout} k{ks} s{s} n={n} {h}x{w} bias={b is not None}"
    if os.environ.get("FUZZ_VERBOSE"):
        print(tag, flush=True)
    for mode in (0, 1):
        try:
            with _lib.config(conv_fp32=mode):
                got, st, _ = run_conv2d(x, wt, b, ks, s, narrow=bool(case & 1))
        except Exception as e:
            print(f"ERROR {tag} mode={mode}: {e}")
            bad += 1
            continue
        err = (got.double() - ref).abs().max().item()
        ssum = st.double().sum(0)
        e1 = (ssum[:, 0] - ref.sum((0, 2, 3))).abs().max().item()
        e2 = (ssum[:, 1] - (ref * ref).sum((0, 2, 3))).abs().max().item()
        tol = 3e-5 * max(1.0, ref.abs().max().item())
        stol = 2e-3 + 1e-4 * (ref * ref).sum((0, 2, 3)).max().item()
        if got.shape != ref.shape or err > tol or e1 > stol or e2 > stol or torch.isnan(got).any():
            bad += 1
            print(f"FAIL {tag} mode={mode}: err {err:.2e} stats {e1:.2e} {e2:.2e}")
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
