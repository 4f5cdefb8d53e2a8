#!/bin/bash
# Collect the per-round evidence under gpurun_out/ (copy the summaries into profiles/ afterwards):
#   tools/profile_round.sh r02g
tag=${1:-rXX}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py --steps 20 --warmup 5 --detail gpurun_out/${tag}_layers.json > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -o ${tag} -- python3 bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency > gpurun_out/${tag}_prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_${tag}_$c --output-format csv -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-roofline --no-latency > gpurun_out/${tag}_pmc_$c.log 2>&1
done
{
  echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 1; mean per launch"
  python3 tools/pmc_summary.py "gpurun_out/pmc_${tag}_FETCH_SIZE/**/*counter_collection.csv" conv3d
  python3 tools/pmc_summary.py "gpurun_out/pmc_${tag}_WRITE_SIZE/**/*counter_collection.csv" conv3d
} > gpurun_out/${tag}_pmc_hbm.txt
python3 tools/pmc_traffic.py "gpurun_out/pmc_${tag}_FETCH_SIZE/**/*counter_collection.csv" "gpurun_out/pmc_${tag}_WRITE_SIZE/**/*counter_collection.csv" gpurun_out/${tag}_traffic.json > /dev/null
find gpurun_out/prof_${tag} -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
# config C4 (HRNet-W18, 1024x512, 8-view TTA) and C3 (5-member ensemble): bench lines + the C4 kernel summary
python3 bench.py --config C4 --steps 5 --warmup 2 --repeats 3 > gpurun_out/${tag}_c4_bench.json 2> gpurun_out/${tag}_c4_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_c4 -o ${tag}_c4 -- python3 bench.py --config C4 --steps 3 --warmup 1 --repeats 1 --no-roofline > gpurun_out/${tag}_c4_prof.log 2>&1
find gpurun_out/prof_${tag}_c4 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_c4_kernel_stats.csv \;
python3 bench.py --config C4 --hrnet-width 48 --steps 4 --warmup 2 --repeats 3 > gpurun_out/${tag}_c4w48_bench.json 2> gpurun_out/${tag}_c4w48_bench.err
python3 bench.py --config C3 --steps 10 --warmup 3 --repeats 3 > gpurun_out/${tag}_c3_bench.json 2> gpurun_out/${tag}_c3_bench.err
python3 bench.py --config C5 --steps 4 --warmup 1 --repeats 3 > gpurun_out/${tag}_c5_bench.json 2> gpurun_out/${tag}_c5_bench.err
tail -1 gpurun_out/${tag}_bench.json | cut -c1-600
head -12 gpurun_out/${tag}_kernel_stats.csv | cut -c1-160
head -30 gpurun_out/${tag}_pmc_hbm.txt
tail -c 900 gpurun_out/${tag}_c4_bench.json
head -8 gpurun_out/${tag}_c4_kernel_stats.csv | cut -c1-160
tail -c 700 gpurun_out/${tag}_c4w48_bench.json
tail -c 500 gpurun_out/${tag}_c3_bench.json
