#!/bin/bash
# bench_train.py (same box, alternating) for the in-tree library and variants/libsnn_<NAME>.so: usage ab_train2.sh <name> -- <bench_train args>
R=$GRAFT_REPO_ROOT; N=$1; shift; shift; O=$R/gpurun_out/ab_train2; mkdir -p $O
for rep in 1 2; do for v in default $N; do
  if [ "$v" = default ]; then unset SNN_HIP_LIB SNN_BENCH_NOCHECK; else export SNN_HIP_LIB=$R/variants/libsnn_$v.so SNN_BENCH_NOCHECK=1; fi
  timeout -k 10 120 python3 -u $R/bench_train.py --cpu-steps 0 "$@" > $O/$v.log 2>&1 || { echo "$v failed"; tail -5 $O/$v.log; exit 1; }
  python3 - $O/$v.log $v <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); ph = j["roofline"]["phases_ms"]
        print("%-10s step %.3f ms  fwd %.3f  bwd %.3f" % (sys.argv[2], j["ms_per_step"], ph["forward"], ph["backward"]))
PY
done; done
