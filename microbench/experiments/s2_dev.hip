// Development unit: the two instantiations of bwd_stream2_kernel the training bench runs, alone, so that
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -I supernnova_amd/csrc -I microbench/experiments microbench/experiments/s2_dev.hip -o /tmp/s2.s
// gives their ISA in seconds (the full snn_train_big.hip takes minutes).
#include "snn_bwd_stream2.h"
template __global__ void bwd_stream2_kernel<2, false, 2, false>(const Bwd2Params, int, int, float*, const float*);
template __global__ void bwd_stream2_kernel<1, true, 2, false>(const Bwd2Params, int, int, float*, const float*);
