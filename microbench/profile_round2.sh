#!/bin/bash
# Round-2 profile run (GPU box): bench lines + rocprofv3 kernel-trace stats + counter passes, everything that
# profiles/r2/ holds.  Outputs under gpurun_out/r2prof/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2prof
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $O/bench_f32.json.log 2>&1
for prec in f32 f16x2; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$prec -- python3 $R/bench.py --precision $prec --cpu-curves 0 --steps 3 --warmup 1 --no-selfcheck --no-fast-path --no-end-to-end --no-secondary > $O/kt_$prec.log 2>&1
  cp $(ls $O/kt_$prec/*/*kernel_stats.csv | head -1) $O/bench_${prec}_kernel_stats.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c -d $O/pmc_$c --output-format csv -- python3 $R/bench.py --cpu-curves 0 --steps 1 --warmup 0 --no-selfcheck --no-fast-path --no-end-to-end --no-secondary > $O/pmc_$c.log 2>&1
done
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/pmc_sq --output-format csv -- python3 $R/bench.py --cpu-curves 0 --steps 1 --warmup 0 --no-selfcheck --no-fast-path --no-end-to-end --no-secondary > $O/pmc_sq.log 2>&1
python3 $R/microbench/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq > $O/forward_counters.txt
for b in 65536 8192 128; do
  timeout -k 10 120 python3 $R/bench_train.py --batch $b --steps 30 --warmup 5 --cpu-steps 0 > $O/bench_train_b$b.json.log 2>&1
done
bash $R/microbench/pmc_train.sh 65536 r2prof/train64k
bash $R/microbench/pmc_train.sh 8192 r2prof/train8k
echo profile ok
