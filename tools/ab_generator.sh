#!/bin/bash
# one-off of round 6: whole-step cost of the dropout generator's variants, per-layer detail for contr_1_2
mkdir -p gpurun_out/r06
for i in 1 2; do for v in base two nob rfl p1 c1; do VX_LIB_PATH=values_amd/libvalues_amd_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-storage16 --min-gpu-seconds 2 --detail gpurun_out/r06/gen_${v}_$i.json 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); L=json.load(open('gpurun_out/r06/gen_${v}_$i.json'))['layers_ms']; print('$v', d['value'], d['ms_per_step'], 'contr_1_2', L.get('contr_1_2'), 'expand_1_1', L.get('expand_1_1'), 'expand_1_2', L.get('expand_1_2'))"; done; done
