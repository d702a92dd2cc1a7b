#!/bin/bash
# A/B of the streamed backward: bwd_stream_kernel vs the step-pipelined bwd_stream3_kernel (SNN_STREAM3=1).
# usage: ab_stream3.sh <batch>...   (runs the training parity tests under SNN_STREAM3=1 first)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_stream3; mkdir -p $O
SNN_STREAM3=1 timeout -k 10 300 python3 -u -m pytest $R/tests/test_gpu_train.py $R/tests/test_gpu_modules.py -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { echo "tests failed under SNN_STREAM3"; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() { tag=$1; B=$2; shift 2
  env "$@" timeout -k 10 120 python3 -u $R/bench_train.py --batch $B --steps 20 --warmup 5 --cpu-steps 0 > $O/${tag}_b$B.log 2>&1 || { echo "$tag failed"; tail -5 $O/${tag}_b$B.log; exit 1; }
  python3 - $O/${tag}_b$B.log $tag $B <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); ph = j["roofline"]["phases_ms"]
        print("b=%-6s %-10s step %.3f ms  fwd %.3f  bwd %.3f" % (sys.argv[3], sys.argv[2], j["ms_per_step"], ph["forward"], ph["backward"]))
PY
}
for B in "$@"; do run v1 $B SNN_X=1 && run v3 $B SNN_STREAM3=1 || exit 1; done
