#!/bin/bash
# Same-box A/B of two builds of the library through the bench's per-layer detail (base, new, base, new):
#   tools/ab_libs.sh tag [layer-name-regex]
tag=${1:-ab}; pat=${2:-.}
cd "$(dirname "$0")/.."
for rep in 0 1; do
  for which in base new; do
    lib=values_amd/libvalues_amd.so; [ $which = base ] && lib=values_amd/libvalues_amd_base.so
    VX_LIB_PATH=$lib python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-storage16 --detail gpurun_out/${tag}_${which}_layers.json 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['value'], d['ms_per_step'])"
  done
done
for which in base new; do echo "== $which"; python3 tools/show_layers.py gpurun_out/${tag}_${which}_layers.json | grep -E "$pat"; done
