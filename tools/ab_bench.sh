#!/bin/bash
# Same-box A/B on the GPU box: base (values_amd/libvalues_amd_base.so, tools/ab_build.sh) vs the working tree's build,
# alternating, per-layer times of each into gpurun_out/.   usage: tools/ab_bench.sh <tag> [bench args]
TAG=$1; shift
mkdir -p gpurun_out
for i in 1 2; do
  VX_LIB_PATH=values_amd/libvalues_amd_base.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-storage16 --min-gpu-seconds 2 --detail gpurun_out/${TAG}_base_layers_$i.json "$@" > gpurun_out/${TAG}_base_$i.json 2> gpurun_out/${TAG}_base.err
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-storage16 --min-gpu-seconds 2 --detail gpurun_out/${TAG}_new_layers_$i.json "$@" > gpurun_out/${TAG}_new_$i.json 2> gpurun_out/${TAG}_new.err
done
python - <<PY
import json
for k in ("base_1","new_1","base_2","new_2"):
    try:
        d=json.load(open("gpurun_out/${TAG}_%s.json"%k)); print(k, d["value"], d["ms_per_step"])
    except Exception as e: print(k, "failed", e)
try:
    L={k:[json.load(open("gpurun_out/${TAG}_%s_layers_%d.json"%(k,i)))["layers_ms"] for i in (1,2)] for k in ("base","new")}
    for k in L["new"][0]:
        a=[x.get(k,0) for x in L["base"]]; b=[x.get(k,0) for x in L["new"]]
        if abs(sum(b)-sum(a))>0.02: print("%-28s %.4f %.4f -> %.4f %.4f"%(k,a[0],a[1],b[0],b[1]))
    print("sum", [round(sum(x.values()),3) for x in L["base"]], [round(sum(x.values()),3) for x in L["new"]])
except Exception as e: print("layers failed", e)
PY
