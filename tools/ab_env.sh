#!/bin/bash
# Same-box A/B of environment settings through the bench's per-layer detail:
#   tools/ab_env.sh tag "" "VX_S16_NO_TY8=1 VX_S16_WG2=1" ...
# writes gpurun_out/<tag>_<i>.json (+ _layers.json) and prints value / ms per step of each setting, twice (A B A B).
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 0 1; do
  i=0
  for envs in "$@"; do
    env $envs python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --detail gpurun_out/${tag}_${i}_layers.json > gpurun_out/${tag}_${i}.json 2> gpurun_out/${tag}_${i}.err
    python3 - "$envs" gpurun_out/${tag}_${i}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("[%s]" % sys.argv[1], d["value"], d["ms_per_step"])
PY
    i=$((i+1))
  done
done
i=0
for envs in "$@"; do echo "== [$envs]"; python3 tools/show_layers.py gpurun_out/${tag}_${i}_layers.json | head -36; i=$((i+1)); done
