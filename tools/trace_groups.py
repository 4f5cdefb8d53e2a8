#!/usr/bin/env python3
"""Group a rocprofv3 kernel_trace.csv by (kernel, grid size): launches, total and mean duration, share."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:50]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = (nm, r["Grid_Size_X"])
    agg[k][0] += 1; agg[k][1] += d; tot += d
print(f"{len(rows)} launches, {tot / 1e3:.2f} ms of kernel time")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:52s} grid {k[1]:>9s} n={v[0]:5d} total {v[1] / 1e3:8.2f} ms  mean {v[1] / v[0]:8.1f} us  {100 * v[1] / tot:5.1f} %")
