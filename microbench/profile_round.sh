#!/bin/bash
# Round profile run (GPU box): bench lines for the three precisions + rocprofv3 kernel-trace stats of the same commands.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for prec in f32 bf16x3 f16x2; do
  timeout -k 10 200 python3 $R/bench.py --precision $prec > $O/bench_$prec.json.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$prec -- python3 $R/bench.py --precision $prec --cpu-curves 0 --steps 3 --warmup 1 --no-selfcheck --no-fast-path > $O/kt_$prec.log 2>&1
  cp $(ls $O/kt_$prec/*/*kernel_stats.csv | head -1) $O/bench_${prec}_kernel_stats.csv
  echo "== $prec"; tail -1 $O/bench_$prec.json.log | cut -c1-200; head -3 $O/bench_${prec}_kernel_stats.csv | cut -c1-160
done
