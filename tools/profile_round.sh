#!/bin/bash
# Collect the per-round evidence under gpurun_out/ (copy the summaries into profiles/ afterwards):
#   tools/profile_round.sh r02g
tag=${1:-rXX}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -o ${tag} -- python3 bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-batch64 --no-storage16 > gpurun_out/${tag}_prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_${tag}_$c --output-format csv -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-roofline --no-latency --no-batch64 --no-storage16 > gpurun_out/${tag}_pmc_$c.log 2>&1
done
{
  echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 1; mean per launch"
  python3 tools/pmc_summary.py "gpurun_out/pmc_${tag}_FETCH_SIZE/**/*counter_collection.csv" conv3d
  python3 tools/pmc_summary.py "gpurun_out/pmc_${tag}_WRITE_SIZE/**/*counter_collection.csv" conv3d
} > gpurun_out/${tag}_pmc_hbm.txt
python3 tools/pmc_traffic.py "gpurun_out/pmc_${tag}_FETCH_SIZE/**/*counter_collection.csv" "gpurun_out/pmc_${tag}_WRITE_SIZE/**/*counter_collection.csv" gpurun_out/${tag}_traffic.json > /dev/null
cp gpurun_out/${tag}_traffic.json profiles/traffic.json        # the bench line below quotes roofline.traffic from it (same sources = same hash)
python3 bench.py --steps 20 --warmup 5 --detail gpurun_out/${tag}_layers.json > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
find gpurun_out/prof_${tag} -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
# config C4 (HRNet-W18, 1024x512, 8-view TTA): the bench line, and -- for a `roofline` object that reproduces from profiles/ --
# the kernel summary + PMC traffic of the SAME single-stream launches the roofline leg times (`--roofline-only`; the graph replay
# overlaps branch kernels on side streams, where rocprofv3's per-kernel durations are inflated)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_c4 -o ${tag}_c4 -- python3 bench.py --config C4 --roofline-only > gpurun_out/${tag}_c4_prof.log 2>&1
find gpurun_out/prof_${tag}_c4 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_c4_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_${tag}_c4_$c --output-format csv -- python3 bench.py --config C4 --roofline-only > gpurun_out/${tag}_c4_pmc_$c.log 2>&1
done
python3 tools/pmc_traffic.py "gpurun_out/pmc_${tag}_c4_FETCH_SIZE/**/*counter_collection.csv" "gpurun_out/pmc_${tag}_c4_WRITE_SIZE/**/*counter_collection.csv" gpurun_out/${tag}_traffic_c4w18.json > /dev/null
cp gpurun_out/${tag}_traffic_c4w18.json profiles/traffic_c4w18.json     # (on the GPU box: the bench line below quotes it; copy it into profiles/ at home too)
# which kernels of the C4 trace are the model's one-off SETUP (parameter uploads, the persistent zero buffers, weight packing): a second
# kernel trace with 2 timed forwards instead of 8 -- their call counts do not move, the per-forward kernels' scale with the forwards
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_c4r2 -o ${tag}_c4r2 -- python3 bench.py --config C4 --roofline-only --roofline-reps 2 > gpurun_out/${tag}_c4r2_prof.log 2>&1
find gpurun_out/prof_${tag}_c4r2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_c4r2_kernel_stats.csv \;
python3 - <<PY > gpurun_out/${tag}_c4_setup_kernels.txt
import csv
a = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open("gpurun_out/${tag}_c4_kernel_stats.csv"))}
b = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open("gpurun_out/${tag}_c4r2_kernel_stats.csv"))}
print("# bench.py --config C4 --roofline-only: kernel call counts with 8 + 1 and with 2 + 1 forwards (rocprofv3 --kernel-trace --stats)")
print("# a kernel whose count does not move belongs to the one-off setup; per-forward kernels scale 9 : 3")
for k in sorted(a, key=lambda k: -a[k]):
    kind = "setup (count fixed)" if a[k] == b.get(k) else ("per forward" if b.get(k) and a[k] * 3 == b[k] * 9 else "other")
    print(f"{a[k]:6d} {b.get(k, 0):6d}  {kind:20s} {k[:110]}")
PY
python3 bench.py --config C4 --steps 5 --warmup 2 --repeats 3 > gpurun_out/${tag}_c4_bench.json 2> gpurun_out/${tag}_c4_bench.err
python3 bench.py --config C4 --volumes 4 --steps 5 --warmup 2 --repeats 3 > gpurun_out/${tag}_c4b4_bench.json 2> gpurun_out/${tag}_c4b4_bench.err
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
