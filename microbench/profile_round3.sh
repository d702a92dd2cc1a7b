#!/bin/bash
# Round-3 profile run (GPU box): what profiles/r3/ holds.  Part 1 (this file): the default bench line, rocprofv3
# kernel-trace stats of the headline command and of every secondary workload, FETCH_SIZE / WRITE_SIZE passes per
# workload (-> microbench/traffic_json.py), one SQ pass of the headline kernel.  Outputs under gpurun_out/r3prof/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
COMMON="--cpu-curves 0 --no-selfcheck --no-fast-path --no-end-to-end --no-secondary --no-train"
fail() { echo "FAILED: $1"; tail -5 $2; exit 1; }
if [ "$1" != "skipbench" ]; then
  timeout -k 10 560 python3 $R/bench.py > $O/bench_default.json.log 2>&1 || fail "default bench" $O/bench_default.json.log
fi
WLS=${WLS:-"des variational elasticc h64 uni gru rnn"}
for wl in $WLS; do
  EXTRA=""; [ $wl = h64 ] && EXTRA="--curves 312500"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$wl -- python3 $R/bench.py --workload $wl $EXTRA $COMMON --steps 3 --warmup 1 > $O/kt_$wl.log 2>&1 || fail "kernel trace $wl" $O/kt_$wl.log
  cp $(ls $O/kt_$wl/*/*kernel_stats.csv | head -1) $O/bench_${wl}_kernel_stats.csv
  grep '^{' $O/kt_$wl.log | tail -1 > $O/bench_${wl}_traced.json
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $c -d $O/pmc_${wl}_$c --output-format csv -- python3 $R/bench.py --workload $wl $EXTRA $COMMON --steps 1 --warmup 0 > $O/pmc_${wl}_$c.log 2>&1 || fail "pmc $c $wl" $O/pmc_${wl}_$c.log
  done
  grep '^{' $O/pmc_${wl}_FETCH_SIZE.log | tail -1 > $O/pmc_$wl.json
  echo "$wl done"
done
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/pmc_sq --output-format csv -- python3 $R/bench.py $COMMON --steps 1 --warmup 0 > $O/pmc_sq.log 2>&1 || fail "pmc sq" $O/pmc_sq.log
python3 $R/microbench/pmc_summary.py $O/pmc_sq > $O/forward_counters.txt
python3 $R/microbench/traffic_json.py $O > $O/traffic.json || fail "traffic json" $O/traffic.json
# the per-dispatch counter tables are large: keep the summaries only
rm -rf $O/kt_* $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE $O/pmc_sq
echo profile ok
