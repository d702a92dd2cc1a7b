#!/bin/bash
# per-kernel times of a training step with bwd_stream_kernel and with bwd_stream3_kernel.  usage: prof_stream3.sh <batch>
R=$GRAFT_REPO_ROOT; B=${1:-65536}; O=$R/gpurun_out/prof_stream3/b$B; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in v1 v3; do
  if [ $v = v3 ]; then export SNN_STREAM3=1; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python3 $R/bench_train.py --batch $B --steps 10 --warmup 3 --cpu-steps 0 > $O/$v.log 2>&1 || { echo "$v failed"; tail -3 $O/$v.log; exit 1; }
  echo "== $v (batch $B)"; python3 - $O/$v <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0])):
    if any(t in r["Name"] for t in ("bwd_", "lstm_fwd", "head_", "reduce")):
        print("   %-52s %8.1f us x %s" % (r["Name"][:52], float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
done
