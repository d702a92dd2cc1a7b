#!/bin/bash
# Round-3 profile run, part 2 (GPU box): training steps (BASELINE configs[3] and the configs[4] shape) -- the bench_train
# line of each case and the rocprofv3 kernel-trace stats of the same command.  Outputs under gpurun_out/r3prof/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
fail() { echo "FAILED: $1"; tail -5 $2; exit 1; }
ONLY="$*"     # case names to run (default: all)
one() { name=$1; steps=$2; shift 2
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return 0; fi
  timeout -k 10 200 python3 $R/bench_train.py --steps $steps --warmup 5 --cpu-steps 0 "$@" > $O/train_$name.json.log 2>&1 || fail "bench_train $name" $O/train_$name.json.log
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt_$name -- python3 $R/bench_train.py --steps $steps --warmup 5 --cpu-steps 0 "$@" > $O/ktt_$name.log 2>&1 || fail "kernel trace $name" $O/ktt_$name.log
  cp $(ls $O/ktt_$name/*/*kernel_stats.csv | head -1) $O/train_${name}_kernel_stats.csv
  rm -rf $O/ktt_$name
  echo "$name done"
}
one b128 200 --batch 128
one b8192 30 --batch 8192
one b65536 20 --batch 65536
one h64_b65536 10 --batch 65536 --hidden 64
one h128_t300_b128 20 --batch 128 --hidden 128 --features 78 --classes 14 --max-len 300 --full-length
one h128_t300_b8192 5 --batch 8192 --hidden 128 --features 78 --classes 14 --max-len 300 --full-length
echo profile ok
