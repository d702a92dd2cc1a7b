// How do fp32 MFMAs (v_mfma_f32_32x32x2_f32, 64 cycles) and vector instructions share a SIMD?  One workgroup of 512
// threads per CU (two waves per SIMD, as in bwd_stream3_kernel).  Per loop iteration a wave issues M MFMAs and V
// DEPENDENT v_fma_f32; modes: 0 both waves run the same program; 1 waves 0-3 issue only the MFMAs, waves 4-7 only the
// vector chain.  Prints cycles per iteration for the older and the younger wave of SIMD 0.
//   hipcc -O3 --offload-arch=gfx950 microbench/mfma_two_waves.hip -o microbench/mfma_two_waves
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int M, int V, int MODE>
__global__ void __launch_bounds__(512, 1) k(float* out, long long* cyc, int iters)
{
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    f32x16 acc[2];
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    float a = threadIdx.x * 1e-3f, b = 1.0001f, x = a;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || role == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0 || role == 1) {
#pragma unroll
            for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(b));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    float s = x;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) cyc[role] = t1 - t0;
}

template <int M, int V, int MODE>
static void run(float* out, long long* cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k<M, V, MODE>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d  M=%d MFMAs + V=%2d dependent v_fma per iteration: older wave %6.1f  younger wave %6.1f cycles/iteration\n", MODE, M, V,
           (double)h[0] / iters, (double)h[1] / iters);
}

int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 2 * sizeof(long long));
    run<1, 0, 0>(out, cyc); run<1, 8, 0>(out, cyc); run<1, 16, 0>(out, cyc); run<1, 32, 0>(out, cyc); run<1, 64, 0>(out, cyc);
    run<2, 0, 0>(out, cyc); run<2, 16, 0>(out, cyc); run<2, 32, 0>(out, cyc); run<2, 64, 0>(out, cyc);
    run<0, 32, 0>(out, cyc);
    run<1, 8, 1>(out, cyc); run<1, 16, 1>(out, cyc); run<1, 32, 1>(out, cyc); run<2, 32, 1>(out, cyc); run<2, 64, 1>(out, cyc); run<4, 64, 1>(out, cyc);
    return 0;
}
