#!/bin/bash
# bench_train.py at one batch size for the in-tree library and for each variants/libsnn_<NAME>.so named; prints the
# phase times.  usage: ab_train.sh <batch> <name> [name ...]
R=$GRAFT_REPO_ROOT; B=$1; shift; O=$R/gpurun_out/ab_train; mkdir -p $O
for v in default "$@"; do
  if [ "$v" = default ]; then unset SNN_HIP_LIB SNN_BENCH_NOCHECK; else export SNN_HIP_LIB=$R/variants/libsnn_$v.so SNN_BENCH_NOCHECK=1; fi
  timeout -k 10 120 python3 -u $R/bench_train.py --batch $B --steps 20 --warmup 5 --cpu-steps 0 > $O/${v}_b$B.log 2>&1 || { echo "$v failed"; tail -5 $O/${v}_b$B.log; exit 1; }
  python3 - $O/${v}_b$B.log $v <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); ph = j["roofline"]["phases_ms"]
        print("%-10s step %.3f ms  fwd %.3f  bwd %.3f  (fwd frac %.3f)" % (sys.argv[2], j["ms_per_step"], ph["forward"], ph["backward"], j["roofline"]["forward"]["frac"]))
PY
done
