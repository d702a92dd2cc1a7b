#!/bin/bash
# GPU side of ablate_b3.sh: time every variant in variants/ on a 400 k-curve workload (kernel_ms of the bench line)
mkdir -p gpurun_out
out=gpurun_out/ablate_b3.txt
: > $out
for f in variants/*.so; do
  v=$(basename $f .so)
  echo "== $v" >> $out
  SNN_BENCH_NOCHECK=1 SNN_HIP_LIB=$PWD/$f timeout -k 10 120 python bench.py --precision ${PREC:-bf16x3} --curves 400000 --cpu-curves 0 --steps 2 --warmup 1 2>&1 | grep -o '"kernel_ms": [0-9.]*' >> $out || exit 1
done
cat $out
