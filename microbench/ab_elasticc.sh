#!/bin/bash
# bench.py --workload elasticc and the forward of the configs[4] training step for the in-tree library and variants
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_elasticc; mkdir -p $O
for v in default "$@"; do
  if [ "$v" = default ]; then unset SNN_HIP_LIB SNN_BENCH_NOCHECK; else export SNN_HIP_LIB=$R/variants/libsnn_$v.so SNN_BENCH_NOCHECK=1; fi
  timeout -k 10 200 python3 $R/bench.py --workload elasticc --steps 2 --warmup 1 --cpu-curves 0 --no-selfcheck --no-fast-path --no-end-to-end --no-secondary --no-train > $O/$v.log 2>&1 || { echo "$v failed"; tail -5 $O/$v.log; exit 1; }
  python3 - $O/$v.log $v <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); print("%-10s elasticc inference %.1f ms" % (sys.argv[2], j["roofline"]["kernel_ms"]))
PY
done
