#!/bin/bash
# Training tests, then bench_train + kernel-trace statistics of the split-backward cases (H = 128 / T = 300 / 8192 curves,
# H = 64 / 65 536 curves).  usage: ab_big.sh <outdir-name> [notest]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
if [ "$2" != notest ]; then
  (cd $R && timeout -k 10 600 python3 -u -m pytest tests/test_gpu_train.py -x -q -m gpu --timeout 120 --timeout-method thread > $O/tests.txt 2>&1) || { echo "tests failed"; tail -15 $O/tests.txt; exit 1; }
  tail -1 $O/tests.txt
fi
one() { name=$1; steps=$2; shift 2
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt_$name -- python3 $R/bench_train.py --steps $steps --warmup 3 --cpu-steps 0 "$@" > $O/$name.log 2>&1 || { echo "$name failed"; tail -5 $O/$name.log; exit 1; }
  cp $(ls $O/ktt_$name/*/*kernel_stats.csv | head -1) $O/${name}_kernel_stats.csv; rm -rf $O/ktt_$name
  python3 $R/microbench/kstats.py $O/${name}_kernel_stats.csv | head -7
  grep -o '"ms_per_step": [0-9.]*' $O/$name.log | head -1
}
one h128 5 --batch 8192 --hidden 128 --features 78 --classes 14 --max-len 300 --full-length &&
one h64 10 --batch 65536 --hidden 64
