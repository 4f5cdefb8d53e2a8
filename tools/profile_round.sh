#!/bin/bash
# Collect the per-round evidence under gpurun_out/ (copy the summaries into profiles/ afterwards):
#   tools/profile_round.sh r01g
tag=${1:-rXX}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py --steps 20 --warmup 3 --detail gpurun_out/${tag}_layers.json > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -o ${tag} -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_${tag}_$c --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${tag}_pmc_$c.log 2>&1
done
{
  echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 1; mean per launch"
  python3 tools/pmc_summary.py "gpurun_out/pmc_${tag}_FETCH_SIZE/**/*counter_collection.csv" conv3d
  python3 tools/pmc_summary.py "gpurun_out/pmc_${tag}_WRITE_SIZE/**/*counter_collection.csv" conv3d
} > gpurun_out/${tag}_pmc_hbm.txt
find gpurun_out/prof_${tag} -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
tail -1 gpurun_out/${tag}_bench.json | cut -c1-400
head -8 gpurun_out/${tag}_kernel_stats.csv
cat gpurun_out/${tag}_pmc_hbm.txt | head -40
