R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3dw; mkdir -p $O
run() { tag=$1; shift; timeout -k 10 120 env "$@" python3 $R/bench_train.py --steps 10 --warmup 5 --cpu-steps 0 $ARGS > $O/$tag.log 2>&1 || { echo "fail $tag"; tail -3 $O/$tag.log; exit 1; }
  python3 - $O/$tag.log $tag <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
print(sys.argv[2], round(d['ms_per_step'],3), round(r['frac'],4), {k:round(v,2) for k,v in r['phases_ms'].items() if k in('forward','backward')})
PY
}
ARGS="--batch 65536 --hidden 64"
run h64_11 SNN_DW_FORM=11 && run h64_21 SNN_DW_FORM=21 && run h64_12 SNN_DW_FORM=12 && run h64_22 SNN_DW_FORM=22 && run h64_12_c64 SNN_DW_FORM=12 SNN_DW_CHUNKS=64 && run h64_12_c128 SNN_DW_FORM=12 SNN_DW_CHUNKS=128 && run h64_21_c64 SNN_DW_FORM=21 SNN_DW_CHUNKS=64 || exit 1
ARGS="--batch 8192 --hidden 128 --features 78 --classes 14 --max-len 300 --full-length --steps 4"
run h128_11 SNN_DW_FORM=11 && run h128_21 SNN_DW_FORM=21 && run h128_12 SNN_DW_FORM=12 || exit 1
ARGS="--batch 8192"
run h32_11 SNN_DW_FORM=11 SNN_BWD_PATH=split && run h32_21 SNN_DW_FORM=21 SNN_BWD_PATH=split && run h32_12 SNN_DW_FORM=12 SNN_BWD_PATH=split && run h32_22 SNN_DW_FORM=22 SNN_BWD_PATH=split
