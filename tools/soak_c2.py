#!/usr/bin/env python3
"""Soak: 300 steps of the C2 workload (32 volumes x T = 10) -- no growth of allocated memory, the fp16 range word stays 0,
the same seed gives the same bits before and after.    python tools/soak_c2.py"""
import sys, torch, time
sys.path.insert(0, "/root/repo")
from values_amd import UNet3D, predict_uncertainty
torch.manual_seed(1)
m = UNet3D(num_classes=2, do_dropout=True).cuda()
x = torch.randn(32, 1, 64, 64, 64, device="cuda")
ref = {k: v.clone() for k, v in predict_uncertainty([m], x, n_pred=10, seeds=[5]).items() if torch.is_tensor(v)}
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
t0 = time.time()
for i in range(300):
    out = predict_uncertainty([m], x, n_pred=10, seeds=[i])
torch.cuda.synchronize(); print("300 steps", round(time.time() - t0, 2), "s; mem delta MB", (torch.cuda.memory_allocated() - m0) / 1e6, "range", m.range_max())
out = predict_uncertainty([m], x, n_pred=10, seeds=[5])
print("deterministic:", all(torch.equal(out[k], ref[k]) for k in ref))
print("finite:", all(torch.isfinite(out[k].float()).all().item() for k in ref))
