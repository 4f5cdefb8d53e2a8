#!/bin/bash
# Every randomised check on the GPU box, fixed seeds; the summary lines land in gpurun_out/<tag>_fuzz.txt.   tools/fuzz_all.sh <tag> [seed]
tag=${1:-rXX}; seed=${2:-6}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  for spec in "tools/fuzz_conv.py 400" "tools/fuzz_unet.py 40" "tools/fuzz_misc.py 150" "tools/fuzz_reduce.py 200" "tools/fuzz_sliding.py 25" \
              "tools/fuzz_hrnet.py 20" "tools/fuzz_conv2d.py 300" "tests/fuzz/fuzz_vs_oracle.py 12" "tests/fuzz/fuzz_metrics.py 120"; do
    set -- $spec
    echo "== $1 $2 cases, seed $seed"
    timeout 1500 python3 $1 $2 $seed 2>&1 | grep -v "amdgpu.ids" | tail -12
  done
} > gpurun_out/${tag}_fuzz.txt 2>&1
grep -E "^==|failures|FAIL|Error|error" gpurun_out/${tag}_fuzz.txt | tail -40
