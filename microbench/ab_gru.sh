#!/bin/bash
# GRU / RNN tests, then the gru / rnn lines of bench.py.  usage: ab_gru.sh <outdir-name>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_engine.py tests/test_gpu_modules.py -x -q -m gpu --timeout 120 --timeout-method thread -k "gru or rnn or wave_per or vanilla" > $O/tests.txt 2>&1 || { echo "tests failed"; tail -25 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
for w in gru rnn; do
timeout -k 10 300 python3 bench.py --workload $w --steps 3 --warmup 1 --cpu-curves 0 --no-fast-path --no-end-to-end --no-secondary --no-train > $O/$w.log 2>&1 || { echo "$w failed"; tail -5 $O/$w.log; exit 1; }
python3 - $O/$w.log $w <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); r = j["roofline"]
        print("%s: %.0f curves/s  %.2f ms  kernel %.2f ms frac %.4f  %s  selfcheck %s" % (sys.argv[2], j["value"], j["ms_per_step"], r["kernel_ms"], r["frac"], r["kernel"], j.get("selfcheck")))
PY
done
