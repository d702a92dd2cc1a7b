#!/bin/bash
# A/B builds for timing ablations: variants/libsnn_<NAME>.so = the library with the listed translation units recompiled
# with extra flags (the other objects are taken from the regular in-tree build).  Runs here (CPU, hipcc cross-compiles).
#   microbench/build_variant.sh <name> "<extra flags>" <file.hip> [file.hip ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/supernnova_amd/csrc; N=$1; F=$2; shift 2
make -s -C $C
mkdir -p $R/variants /tmp/snn_variant_$N
OBJS=""
for o in $C/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do
    if [ "$f" = "$b.hip" ]; then
      EX=""; case $b in snn_forward_h2|snn_forward_b3p) EX="-fno-slp-vectorize";; esac
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $EX $F -c $C/$f -o /tmp/snn_variant_$N/$b.o
      use=/tmp/snn_variant_$N/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/libsnn_$N.so $OBJS
echo "built variants/libsnn_$N.so"
