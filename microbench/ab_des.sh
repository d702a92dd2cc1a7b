#!/bin/bash
# Headline workload (and variational / uni) for the in-tree library and variants/libsnn_<NAME>.so, alternating on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_des; mkdir -p $O
for wl in des variational uni; do for v in default "$@" default "$@"; do
  if [ "$v" = default ]; then unset SNN_HIP_LIB SNN_BENCH_NOCHECK; else export SNN_HIP_LIB=$R/variants/libsnn_$v.so SNN_BENCH_NOCHECK=1; fi
  timeout -k 10 200 python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --cpu-curves 0 --no-fast-path --no-end-to-end --no-secondary --no-train > $O/$v.log 2>&1 || { echo "$v failed"; tail -5 $O/$v.log; exit 1; }
  python3 - $O/$v.log $v $wl <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); print("%-12s %-8s kernel %.1f ms  frac %.4f  selfcheck %s" % (sys.argv[3], sys.argv[2], j["roofline"]["kernel_ms"], j["roofline"]["frac"], (j.get("selfcheck") or "")[-14:]))
PY
done; done
