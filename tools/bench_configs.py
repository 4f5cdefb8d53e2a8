#!/usr/bin/env python3
"""Throughput of the non-headline configurations of BASELINE.json on one GPU (the headline C2 is bench.py):
  C3  64^3, 5-member deep ensemble, n_pred = 1, no dropout             -> volumes/s
  C3t 64^3, one member, 16-view TTA                                    -> volumes/s
  C5  128^3 volume, sliding window patch 64, T = 20 MC-dropout, overlap 1 (8 patches) and 0.5 (27 patches) -> volumes/s
  SSN 64^3, SsnUNet3D rank 10, 10 samples                              -> volumes/s
Synthetic z-scored inputs, torch default-init weights."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from values_amd import SsnUNet3D, UNet3D, predict_image_sliding, predict_uncertainty  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(123)
    V = 16
    x = torch.randn((V, 1, 64, 64, 64), device=dev)
    members = [UNet3D(num_classes=2, do_dropout=False).to(dev) for _ in range(5)]
    dt = timed(lambda: predict_uncertainty(members, x, n_pred=1), 10)
    print(f"C3  5-member ensemble, 64^3:            {V / dt:8.1f} volumes/s ({dt * 1e3:.2f} ms per {V} volumes)")
    dt = timed(lambda: predict_uncertainty(members[:1], x, tta=True), 10)
    print(f"C3t 16-view TTA, 64^3:                  {V / dt:8.1f} volumes/s")
    mc = UNet3D(num_classes=2, do_dropout=True).to(dev)
    big = torch.randn((128, 128, 128), device=dev)
    for ov, npatch in ((1, 8), (0.5, 27)):
        dt = timed(lambda: predict_image_sliding([mc], big, patch_size=64, patch_overlap=ov, n_pred=20, patch_batch=8), 5)
        print(f"C5  128^3, T=20, overlap {ov} ({npatch:2d} patches): {1 / dt:8.2f} volumes/s ({dt * 1e3:.1f} ms per volume)")
    ssn = SsnUNet3D(num_classes=2).to(dev)
    dt = timed(lambda: predict_uncertainty([ssn], x, n_pred=10, ssn=True), 10)
    print(f"SSN rank 10, 10 samples, 64^3:          {V / dt:8.1f} volumes/s")


if __name__ == "__main__":
    main()
