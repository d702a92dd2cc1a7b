#!/bin/bash
# Timing ablations of the streamed backward (variants/libsnn_<NAME>.so built with microbench/build_variant.sh and -DSNN_ABL_S3_L2 / -DSNN_ABL_S3_NOSTORE / -DSNN_S3_POSTX / -DSNN_STREAM_OCC=2):
# rocprofv3 kernel-trace of bench_train.py at one batch size per variant; prints the stream kernels' average times.
R=$GRAFT_REPO_ROOT; B=${1:-65536}; O=$R/gpurun_out/ablate_stream/b$B; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in default "$@"; do
  [ "$v" = "$B" ] && continue
  if [ "$v" = default ]; then unset SNN_HIP_LIB; else export SNN_HIP_LIB=$R/variants/libsnn_$v.so; export SNN_BENCH_NOCHECK=1; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python3 $R/bench_train.py --batch $B --steps 10 --warmup 3 --cpu-steps 0 > $O/$v.log 2>&1 || { echo "$v failed"; tail -3 $O/$v.log; exit 1; }
  echo "== $v (batch $B)"; python3 - $O/$v <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0])):
    if any(t in r["Name"] for t in ("bwd_", "lstm_fwd", "head_", "reduce")):
        print("   %-52s %8.1f us x %s" % (r["Name"][:52], float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
done
