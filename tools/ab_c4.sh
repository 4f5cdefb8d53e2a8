#!/bin/bash
# Same-box A/B of two builds of the library on config C4 (HRNet-W18, 1024x512, 8 views): base, new, base, new
cd "$(dirname "$0")/.."
for rep in 0 1; do
  for lib in values_amd/libvalues_amd_base.so values_amd/libvalues_amd.so; do
    VX_LIB_PATH=$lib python3 bench.py --config C4 --steps 4 --warmup 2 --repeats 2 --no-roofline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])"
  done
done
