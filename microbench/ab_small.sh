#!/bin/bash
# All GPU tests, then the lines that run_phase (snn_fwd_small.h) carries: training steps at H = 32 / 64 and the
# one-tile inference workloads.  usage: ab_small.sh <outdir-name> [notest]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
if [ "$2" != notest ]; then
timeout -k 10 900 python3 -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > $O/tests.txt 2>&1 || { echo "tests failed"; tail -25 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
fi
tr() { name=$1; shift
  timeout -k 10 200 python3 bench_train.py --cpu-steps 0 "$@" > $O/train_$name.log 2>&1 || { echo "train $name failed"; tail -5 $O/train_$name.log; exit 1; }
  python3 - $O/train_$name.log $name <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); ph = j["roofline"]["phases_ms"]
        print("train %-12s step %.3f ms  fwd %.3f  bwd %.3f  frac %.3f" % (sys.argv[2], j["ms_per_step"], ph["forward"], ph["backward"], j["roofline"]["frac"]))
PY
}
tr b128 --steps 200 --warmup 20 --batch 128 && tr b8192 --steps 30 --warmup 5 --batch 8192 && tr b65536 --steps 20 --warmup 5 --batch 65536 && tr h64 --steps 10 --warmup 3 --batch 65536 --hidden 64 || exit 1
for w in h64 gru rnn; do
timeout -k 10 300 python3 bench.py --workload $w --steps 3 --warmup 1 --cpu-curves 0 --no-fast-path --no-end-to-end --no-secondary --no-train > $O/$w.log 2>&1 || { echo "$w failed"; tail -5 $O/$w.log; exit 1; }
python3 - $O/$w.log $w <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); r = j["roofline"]
        print("%s: %.0f curves/s  %.2f ms  kernel %.2f ms frac %.4f  %s" % (sys.argv[2], j["value"], j["ms_per_step"], r["kernel_ms"], r["frac"], r["kernel"]))
PY
done
