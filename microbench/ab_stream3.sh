#!/bin/bash
# Development loop of the streamed backward: the training parity tests, then bench_train.py phase times per batch size.
# usage: [AB_ENV=<env switch of a second kernel>] ab_stream3.sh <batch>...   (A/B against another build: microbench/ab_variants.sh with variants/libsnn_<NAME>.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_stream3; mkdir -p $O
env ${AB_ENV:-SNN_X}=1 timeout -k 10 300 python3 -u -m pytest $R/tests/test_gpu_train.py $R/tests/test_gpu_modules.py -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
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
for B in "$@"; do run default $B SNN_X=1 && { [ -z "$AB_ENV" ] || run "$AB_ENV" $B $AB_ENV=1; } || exit 1; done
