#!/bin/bash
# Training-step profile (GPU box): rocprofv3 kernel-trace stats + SQ counter passes + FETCH/WRITE_SIZE of
# bench_train.py at one batch size.  usage: pmc_train.sh <batch> <outdir-name> [extra bench_train args]
R=$GRAFT_REPO_ROOT; B=$1; O=$R/gpurun_out/$2; shift 2; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="--batch $B --steps 3 --warmup 2 --cpu-steps 0 $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench_train.py --batch $B --steps 10 --warmup 3 --cpu-steps 0 "$@" > $O/kt.log 2>&1 || { echo "kernel trace failed"; tail -3 $O/kt.log; exit 1; }
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
run() { name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $O/$name --output-format csv -- python3 $R/bench_train.py $ARGS > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $O/$name.log; exit 1; }
}
run a SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE &&
run b SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR &&
run c SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE && run write WRITE_SIZE && python3 $R/microbench/pmc_summary.py $O/a $O/b $O/c $O/fetch $O/write > $O/counters.txt && echo passes ok
