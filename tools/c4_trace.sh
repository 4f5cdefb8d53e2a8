#!/bin/bash
# Kernel trace of config C4 with every launch on ONE stream (VX_HRNET_SINGLE_STREAM=1): per-kernel durations that are not
# inflated by the branches of a stage running side by side.  tools/c4_trace.sh <tag>; summary by tools/trace_groups.py
tag=${1:-c4s}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp VX_HRNET_SINGLE_STREAM=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_trace -o ${tag} -- python3 bench.py --config C4 --steps 2 --warmup 1 --repeats 1 --no-roofline > gpurun_out/${tag}_trace.log 2>&1
python3 tools/trace_groups.py gpurun_out/${tag}_trace/${tag}_kernel_trace.csv | head -45
tail -c 400 gpurun_out/${tag}_trace.log
