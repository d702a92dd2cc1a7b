#!/bin/bash
# bench_train.py phase times for the in-tree library and for variants/libsnn_<NAME>.so.
# usage: ab_variants.sh <batch> <name>...
R=$GRAFT_REPO_ROOT; B=$1; shift; O=$R/gpurun_out/ab_variants; mkdir -p $O
for v in default "$@"; do
  if [ "$v" = default ]; then unset SNN_HIP_LIB; else export SNN_HIP_LIB=$R/variants/libsnn_$v.so; fi
  timeout -k 10 120 python3 -u $R/bench_train.py --batch $B --steps 20 --warmup 5 --cpu-steps 0 > $O/${v}_b$B.log 2>&1 || { echo "$v failed"; tail -5 $O/${v}_b$B.log; exit 1; }
  python3 - $O/${v}_b$B.log $v $B <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); ph = j["roofline"]["phases_ms"]
        print("b=%-6s %-10s step %.3f ms  fwd %.3f  bwd %.3f" % (sys.argv[3], sys.argv[2], j["ms_per_step"], ph["forward"], ph["backward"]))
PY
done
