#!/bin/bash
# PMC counters of the z-column kernels (product library), one rocprofv3 pass per counter group:
#   tools/pmc_xp8.sh <tag> <spec...>      (spec as tools/stamp_s16.py: cin:cout:edge:act:drop:head[:up])
cd /root/repo
export TMPDIR=/tmp STAMP_PRODUCT=1 STAMP_TERSE=1 STAMP_N=320 STAMP_REPS=6
tag=$1; shift
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmcx_${tag}_$n --output-format csv -- python3 tools/stamp_s16.py $SPECS > gpurun_out/pmcx_${tag}_$n.log 2>&1
}
SPECS="$*"
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run d GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
for n in a b c d; do python3 tools/pmc_summary.py "gpurun_out/pmcx_${tag}_$n/**/*counter_collection.csv" ${PMC_FILTER:-xp8}; tail -1 gpurun_out/pmcx_${tag}_$n.log | cut -c1-150; done
