// Does v_mfma_f32_32x32x16_f16 honour subnormal fp16 inputs on gfx950, and does v_cvt_pk_f16_f32 produce them?
//   hipcc --offload-arch=gfx950 -O3 -o f16_denorm_probe f16_denorm_probe.hip && ./f16_denorm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float* out, float tiny)
{
    unsigned int pk;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(pk) : "v"(tiny));
    _Float16 t = __builtin_bit_cast(_Float16, (unsigned short)(pk & 0xffff));
    f16x8 a = {t, 0, 0, 0, 0, 0, 0, 0}, b = {(_Float16)1024.0f, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    f32x16 acc2 = {0};
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc2, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = acc2[0]; out[2] = (float)t; out[3] = __uint_as_float(pk); }
}
int main()
{
    float* d; hipMalloc(&d, 16);
    const float tiny = 9.5367431640625e-07f;   // 2^-20: subnormal in fp16 (min normal 2^-14)
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, tiny);
    float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("tiny=%g  cvt->f16->f32=%g  mfma(A=tiny,B=1024)=%g  mfma(A=1024,B=tiny)=%g  expected %g (0 = flushed)\n", tiny, h[2], h[0], h[1], tiny * 1024.0f);
    return 0;
}
