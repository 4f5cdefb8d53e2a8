#!/bin/bash
# Same-box A/B of one vx_config knob through the bench (base = knob set, new = default; base, new, base, new):
#   tools/ab_knob.sh VX_S16_NO_UPCOMPOSE [layer-name-regex]
knob=${1:?knob}; pat=${2:-.}
cd "$(dirname "$0")/.."
for rep in 0 1; do
  for which in base new; do
    if [ $which = base ]; then export $knob=1; else unset $knob; fi
    python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-storage16 --detail gpurun_out/ab_${which}_layers.json 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['value'], d['ms_per_step'])"
  done
done
unset $knob
for which in base new; do echo "== $which"; python3 tools/show_layers.py gpurun_out/ab_${which}_layers.json | grep -E "$pat"; done
