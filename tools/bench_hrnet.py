#!/usr/bin/env python3
"""Timing of the 2D path (config C4 shape): HRNet-W48, (B,3,256,478), DROPOUT_FINAL MC sampling."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from values_amd.hrnet_configs import hrnet_w18_extra, hrnet_w48_extra
from values_amd.hrnet import HighResolutionNet

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=12)
ap.add_argument("--T", type=int, default=4)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--h", type=int, default=256)
ap.add_argument("--w", type=int, default=478)
ap.add_argument("--width", type=int, default=48, choices=(18, 48), help="HRNet-W48 (shipped configs) or W18 (BASELINE config 4)")
ap.add_argument("--classes", type=int, default=24)
args = ap.parse_args()
extra = hrnet_w48_extra(True) if args.width == 48 else hrnet_w18_extra(True)
cfg = {"MODEL": {"EXTRA": extra, "ALIGN_CORNERS": False, "INPUT_CHANNELS": 3},
       "DATASET": {"NUM_CLASSES": args.classes}}
torch.manual_seed(0)
m = HighResolutionNet(cfg).cuda()
x = torch.randn(args.batch, 3, args.h, args.w, device="cuda")
for T in (1, args.T):
    m.forward_samples(x, T, seeds=list(range(T)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        y = m.forward_samples(x, T, seeds=list(range(T)))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    gflop = 87.65 * args.batch * (args.h * args.w) / (256 * 478)
    print(f"T={T}: {dt*1e3:.2f} ms per batch of {args.batch} -> {args.batch/dt:.1f} images/s; backbone+head ~{gflop/dt/1e3:.1f} TFLOP/s (T=1 work)", flush=True)

for T in (1, args.T):
    f = m.graphed(x, T)
    f(x); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        y = f(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    eager = m.forward_samples(x, T, seeds=list(range(T)))
    print(f"graphed T={T}: {dt*1e3:.2f} ms per batch of {args.batch} -> {args.batch/dt:.1f} images/s; max|graph-eager| = {(y-eager).abs().max().item():.1e}", flush=True)
