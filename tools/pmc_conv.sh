cd /root/repo
export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_$n --output-format csv -- python3 tools/bench_conv.py --n 160 --reps 8 --act 0 --drop 0 --stats 1 contr_1_2:8:8:64 > gpurun_out/pmc_$n.log 2>&1; rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_${n}2 --output-format csv -- python3 tools/bench_conv.py --n 160 --reps 8 expand_1_1:16:8:64 > gpurun_out/pmc_$n.log 2>&1
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run d GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
for n in a b c d; do echo "== $n"; python3 tools/pmc_summary.py "gpurun_out/pmc_$n*/**/*counter_collection.csv" s16 ; tail -2 gpurun_out/pmc_$n.log; done
