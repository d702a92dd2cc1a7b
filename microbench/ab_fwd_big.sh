#!/bin/bash
# Forward phase time of bench_train.py at the configs[4] shape for the in-tree library and each variants/libsnn_<NAME>.so
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_fwd_big; mkdir -p $O
for v in default "$@"; do
  if [ "$v" = default ]; then unset SNN_HIP_LIB SNN_BENCH_NOCHECK; else export SNN_HIP_LIB=$R/variants/libsnn_$v.so SNN_BENCH_NOCHECK=1; fi
  timeout -k 10 120 python3 -u $R/bench_train.py --batch 8192 --hidden 128 --features 78 --classes 14 --max-len 300 --full-length --steps 3 --warmup 2 --cpu-steps 0 > $O/$v.log 2>&1 || { echo "$v failed"; tail -5 $O/$v.log; exit 1; }
  python3 - $O/$v.log $v <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); ph = j["roofline"]["phases_ms"]
        print("%-10s step %.2f ms  fwd %.2f  bwd %.2f" % (sys.argv[2], j["ms_per_step"], ph["forward"], ph["backward"]))
PY
done
