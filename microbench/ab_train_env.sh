#!/bin/bash
# bench_train.py at one batch size under different path selections (env), prints phase times.  usage: ab_train_env.sh <batch>
R=$GRAFT_REPO_ROOT; B=$1; O=$R/gpurun_out/ab_train; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout -k 10 120 python3 -u $R/bench_train.py --batch $B --steps 20 --warmup 5 --cpu-steps 0 > $O/env_${tag}_b$B.log 2>&1 || { echo "$tag failed"; tail -5 $O/env_${tag}_b$B.log; exit 1; }
  python3 - $O/env_${tag}_b$B.log $tag $B <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); ph = j["roofline"]["phases_ms"]
        print("b=%-6s %-22s step %.3f ms  fwd %.3f  bwd %.3f" % (sys.argv[3], sys.argv[2], j["ms_per_step"], ph["forward"], ph["backward"]))
PY
}
run default SNN_X=1 && run one_tile SNN_NO_PAIR=1 SNN_NO_LAYERWISE=1 && run pair SNN_NO_LAYERWISE=1 && run layerwise SNN_FORCE_LAYERWISE=1
