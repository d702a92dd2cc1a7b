// Development unit for snn_bwd_stream4.h (see s2_dev.hip):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -I supernnova_amd/csrc -I microbench/experiments microbench/experiments/s4_dev.hip -o /tmp/s4.s
#include "snn_bwd_stream4.h"
template __global__ void bwd_stream4_kernel<2, false, 2, false>(const Bwd2Params, int, int, float*, const float*);
template __global__ void bwd_stream4_kernel<1, true, 2, false>(const Bwd2Params, int, int, float*, const float*);
