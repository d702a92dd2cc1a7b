// snn_device.h -- device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- activations -----------------------------------------------------------
// v_exp_f32 / v_rcp_f32 based; absolute error ~1e-7 (checked against the oracle
// in tests/test_gpu_parity.py).  exp(-x) overflowing to +inf gives rcp(inf) = 0.
__device__ __forceinline__ float snn_exp(float x)
{
#ifdef SNN_ACCURATE_MATH
    return expf(x);
#else
    // 2^(x*log2e) with the product's rounding error compensated by an fma
    const float l2e_hi = 1.44269502162933349609375f, l2e_lo = 1.925963033500011e-08f;
    float ph = x * l2e_hi;
    float pl = fmaf(x, l2e_hi, -ph) + x * l2e_lo;
    return __builtin_amdgcn_exp2f(ph) * (1.0f + pl * 0.693147180559945f);
#endif
}
__device__ __forceinline__ float snn_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + snn_exp(-x));
}
__device__ __forceinline__ float snn_tanh(float x)
{
#ifdef SNN_ACCURATE_MATH
    return tanhf(x);
#else
    // 1 - 2/(1+e^{2x}); for small |x| the cancellation costs absolute 1e-7 only
    float ax = fabsf(x);
    float e = snn_exp(-2.0f * ax);                     // in (0, 1]
    float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
#endif
}

// ---- Philox4x32-10 ---------------------------------------------------------
struct Philox4 { uint32_t x, y, z, w; };
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return Philox4{c0, c1, c2, c3};
}
// uniform in (0,1]
__device__ __forceinline__ float u01(uint32_t r) { return ((r >> 8) + 1) * 5.9604644775390625e-08f; }

#define SNN_STREAM_BBB 0x0B
#define SNN_STREAM_MASK_IN 0x10
#define SNN_STREAM_MASK_WHH 0x11
#define SNN_STREAM_MASK_MID 0x12
#define SNN_STREAM_MASK_OUT 0x13

// N(0,1) for (draw, element): Box-Muller on one Philox block
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t draw, uint32_t stream, uint64_t elem)
{
    Philox4 r = philox4x32_10((uint32_t)elem, (uint32_t)(elem >> 32), (uint32_t)draw, stream,
                              (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(draw >> 32));
    float u1 = u01(r.x), u2 = u01(r.y);
    return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}
// Bernoulli(keep) * keep_scale for (draw, stream, a, b)
__device__ __forceinline__ float philox_mask(uint64_t seed, uint64_t draw, uint32_t stream, uint32_t a,
                                             uint32_t b, float keep_prob, float keep_scale)
{
    Philox4 r = philox4x32_10(a, b, (uint32_t)draw, stream, (uint32_t)seed,
                              (uint32_t)(seed >> 32) ^ (uint32_t)(draw >> 32));
    return ((r.x >> 8) * 5.9604644775390625e-08f) < keep_prob ? keep_scale : 0.0f;
}
