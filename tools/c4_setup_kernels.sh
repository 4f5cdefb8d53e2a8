#!/bin/bash
# Which kernels of the C4 trace are one-off setup: two kernel traces of `bench.py --config C4 --roofline-only` (8 and 2 timed forwards).
#   tools/c4_setup_kernels.sh <tag>     -> gpurun_out/<tag>_c4_setup_kernels.txt   (tools/profile_round.sh runs the same steps)
tag=${1:-rXX}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for r in 8 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_c4r$r -o ${tag}_c4r$r -- python3 bench.py --config C4 --roofline-only --roofline-reps $r > gpurun_out/${tag}_c4r${r}_prof.log 2>&1
  find gpurun_out/prof_${tag}_c4r$r -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_c4r${r}_kernel_stats.csv \;
done
python3 - <<PY > gpurun_out/${tag}_c4_setup_kernels.txt
import csv
a = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open("gpurun_out/${tag}_c4r8_kernel_stats.csv"))}
b = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open("gpurun_out/${tag}_c4r2_kernel_stats.csv"))}
print("# bench.py --config C4 --roofline-only: kernel call counts with 8 + 1 and with 2 + 1 forwards (rocprofv3 --kernel-trace --stats)")
print("# a kernel whose count does not move belongs to the one-off setup; per-forward kernels scale 9 : 3")
for k in sorted(a, key=lambda k: -a[k]):
    kind = "setup (count fixed)" if a[k] == b.get(k) else ("per forward" if b.get(k) and a[k] * 3 == b[k] * 9 else "other")
    print(f"{a[k]:6d} {b.get(k, 0):6d}  {kind:20s} {k[:110]}")
PY
cat gpurun_out/${tag}_c4_setup_kernels.txt | grep -v "conv2d_s16\|fuse_sum\|affine" | head -20
