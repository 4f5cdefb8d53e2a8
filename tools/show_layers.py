#!/usr/bin/env python3
"""Print the per-layer / per-kernel table of a `bench.py --detail` JSON."""
import json, sys
d = json.load(open(sys.argv[1]))
L = d["layers_ms"]; tot = sum(L.values())
print(f"forward {d['forward_ms_sum_of_launches']} ms over {d['samples']} samples")
groups = {}
for k, v in L.items():
    key = k.split(":")[0] if ":" in k else ("convT" if k.startswith("upscale") or k == "center.4" else k)
    groups[key] = groups.get(key, 0) + v
    print(f"  {k:24s} {v:8.4f} ms {100 * v / tot:5.1f}%")
for k, v in d["kernels"].items():
    print(f"{k:44s} {v['ms_per_step']:7.4f} ms x{v['launches']}  {v['tflops'] or 0:7.2f} TF  {v['alg_GBps'] or 0:7.1f} GB/s")
