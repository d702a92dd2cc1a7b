// Does VALU work issue in the shadow of v_mfma_f32_32x32x2_f32 on gfx950, as it does for bf16 MFMA?
// One wave per SIMD (256 threads per CU-sized workgroup), chains of dependent MFMAs with K independent
// VALU (or transcendental) instructions placed behind each MFMA.  Prints cycles per MFMA slot.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int K, bool BF16>
__global__ void __launch_bounds__(256) probe(float* out, long long* cyc, int iters)
{
    f32x16 acc = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 ab = {0}, bb = {0};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + i * 0.01f + threadIdx.x * 1e-5f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (MODE != 1) {
                if (BF16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            if (MODE != 0) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float& x = v[k & 7];
                    if (MODE == 2 || MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
                    else asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int K, bool BF16>
static void run(const char* name, float* out, long long* cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL((probe<MODE, K, BF16>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, K, BF16>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-44s %8.1f counter ticks / slot   %8.3f ns / slot\n", name, (double)c / (iters * 16.0), ms * 1e6 / (iters * 16.0));
}

int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 8);
    run<0, 0, false>("f32 32x32x2 MFMA only", out, cyc);
    run<1, 8, false>("8 v_fma only", out, cyc);
    run<2, 4, false>("f32 MFMA + 4 v_fma", out, cyc);
    run<2, 8, false>("f32 MFMA + 8 v_fma", out, cyc);
    run<2, 12, false>("f32 MFMA + 12 v_fma", out, cyc);
    run<2, 16, false>("f32 MFMA + 16 v_fma", out, cyc);
    run<3, 4, false>("f32 MFMA + 4 v_exp", out, cyc);
    run<3, 8, false>("f32 MFMA + 8 v_exp", out, cyc);
    run<0, 0, true>("bf16 32x32x16 MFMA only", out, cyc);
    run<2, 4, true>("bf16 MFMA + 4 v_fma", out, cyc);
    run<2, 6, true>("bf16 MFMA + 6 v_fma", out, cyc);
    run<2, 8, true>("bf16 MFMA + 8 v_fma", out, cyc);
    return 0;
}
