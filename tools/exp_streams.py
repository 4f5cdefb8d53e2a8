#!/usr/bin/env python3
"""Experiment: how to overlap the HBM-bound launches of one batch with the convolutions of another.
  a) one stream                      b) volume chunks on 2 streams inside a step (joined every step: predict's n_streams)
  c) whole steps alternating between 2 streams (no join between steps)      d) b + c"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from values_amd import UNet3D, predict_uncertainty

dev = torch.device("cuda", 0)
torch.manual_seed(123)
T = 10
V = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = UNet3D(num_classes=2, do_dropout=True).to(dev)
x = torch.randn((V, 1, 64, 64, 64), device=dev)
ss = [torch.cuda.Stream() for _ in range(2)]

def run(alt, chunks, n=20):
    def step(i):
        if alt:
            with torch.cuda.stream(ss[i % 2]):
                return predict_uncertainty([m], x, n_pred=T, seeds=[i], n_streams=chunks)
        return predict_uncertainty([m], x, n_pred=T, seeds=[i], n_streams=chunks)
    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    return V * n / (time.perf_counter() - t0)

for rep in range(2):
    print(f"V={V}  a) {run(False, 1):7.1f}   b) {run(False, 2):7.1f}   c) {run(True, 1):7.1f}   d) {run(True, 2):7.1f} vol/s")
