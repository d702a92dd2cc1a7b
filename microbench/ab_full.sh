#!/bin/bash
# All GPU tests, then the H = 128 lines: bench_train at the configs[4] shape and bench.py --workload elasticc.
# usage: ab_full.sh <outdir-name>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout -k 10 900 python3 -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > $O/tests.txt 2>&1 || { echo "tests failed"; tail -25 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
timeout -k 10 200 python3 bench_train.py --steps 5 --warmup 3 --cpu-steps 0 --batch 8192 --hidden 128 --features 78 --classes 14 --max-len 300 --full-length > $O/train_h128.log 2>&1 || { echo "train failed"; tail -5 $O/train_h128.log; exit 1; }
python3 - $O/train_h128.log <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); ph = j["roofline"]["phases_ms"]
        print("train h128: step %.2f ms  fwd %.2f  bwd %.2f  frac %.3f" % (j["ms_per_step"], ph["forward"], ph["backward"], j["roofline"]["frac"]))
PY
timeout -k 10 300 python3 bench.py --workload elasticc --steps 3 --warmup 1 --cpu-curves 0 --no-fast-path --no-end-to-end --no-secondary --no-train > $O/elasticc.log 2>&1 || { echo "elasticc failed"; tail -5 $O/elasticc.log; exit 1; }
python3 - $O/elasticc.log <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); r = j["roofline"]
        print("elasticc: %.0f curves/s  %.1f ms  frac %.4f  selfcheck %s  %s" % (j["value"], j["ms_per_step"], r["frac"], j.get("selfcheck"), r["kernel"]))
PY
