#!/usr/bin/env python3
"""Time the fused softmax -> {mean, entropy, MI, variance, argmax} reduction at the bench's shape (32 x 10 x 2 x 64^3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from values_amd.uncertainty import uncertainty_maps
V, T, C, S = 32, 10, 2, 64
x = torch.randn((V, T, C, S, S, S), device="cuda")
for _ in range(5):
    m = uncertainty_maps(x, from_logits=True, want_variance=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    m = uncertainty_maps(x, from_logits=True, want_variance=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
gb = (x.numel() * 4 + V * S ** 3 * ((3 + C + 1) * 4 + 1)) / 1e9
print(f"unc_reduce {ms:.4f} ms  {gb / ms * 1e3 / 1e3:.2f} TB/s algorithmic")
