#!/bin/bash
# Counter passes on the f16x2 forward kernel (400 k curves, one launch each); one rocprofv3 run per pass.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_h2; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $O/$name --output-format csv -- python3 $R/bench.py --precision ${PREC:-f16x2} --curves 400000 --cpu-curves 0 --steps 1 --warmup 0 > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $O/$name.log; exit 1; }
}
run a SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE &&
run b SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR &&
run c SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE && run write WRITE_SIZE && echo passes ok
