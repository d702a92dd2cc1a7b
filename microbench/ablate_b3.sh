#!/bin/bash
# Timing ablations of the bf16x3 forward kernels: builds variants of snn_forward_b3p.o (or SRC=snn_forward_b3) with -D
# flags into variants/ (run here, on the CPU box); timed on the GPU by run_ablate_b3.sh through SNN_HIP_LIB.
set -e
cd "$(dirname "$0")/../supernnova_amd/csrc"
SRC=${SRC:-snn_forward_b3p}
mkdir -p ../../variants
make -s -j8
OTHERS=$(ls *.o | grep -v "^$SRC.o$")
build() {  # name flags...
  name=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $SRC.hip -o ../../variants/$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$name.so ../../variants/$name.o $OTHERS
  rm ../../variants/$name.o
}
n=0
while read -r name flags; do
  [ -z "$name" ] && continue
  build $name $flags &
  n=$((n+1)); [ $((n % 4)) -eq 0 ] && wait
done <<LIST
${VARIANTS:-base
noyld_noyst -DSNN_ABL_NOYLD -DSNN_ABL_NOYST
math9 -DSNN_MATH=9
nomfma -DSNN_ABL_NOMFMA
nobar -DSNN_ABL_NOBAR
w1 -DSNN_B3P_WAVES=1
nomfma_noyld_noyst -DSNN_ABL_NOMFMA -DSNN_ABL_NOYLD -DSNN_ABL_NOYST
math9_noyld_noyst -DSNN_MATH=9 -DSNN_ABL_NOYLD -DSNN_ABL_NOYST}
LIST
wait
ls ../../variants
