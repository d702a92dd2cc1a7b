// Development unit for snn_bwd_stream3.h (see s2_dev.hip):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -I supernnova_amd/csrc microbench/experiments/s3_dev.hip -o /tmp/s3.s
#include "snn_bwd_stream3.h"
template __global__ void bwd_stream3_kernel<2, false, 2, false>(const Bwd2Params, int, int, float*, const float*);
template __global__ void bwd_stream3_kernel<1, true, 2, false>(const Bwd2Params, int, int, float*, const float*);
