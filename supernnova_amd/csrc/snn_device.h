// snn_device.h -- device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- activations -----------------------------------------------------------
// SNN_MATH 0 (default): v_exp_f32 / v_rcp_f32, 4-5 instructions per activation.  exp(x) = 2^(x*log2e)
//   carries a relative error of ~|x|*1e-7 in the exponential, i.e. an ABSOLUTE error <= ~1e-7 in the
//   sigmoid / tanh value (they saturate where |x| is large); checked against the oracle in the GPU
//   parity tests (class probabilities within 1e-5 relative).
// SNN_MATH 1: exponent product compensated by an fma (abs error ~6e-8).   SNN_MATH 2: ocml expf/tanhf.
#ifndef SNN_MATH
#define SNN_MATH 0
#endif
__device__ __forceinline__ float snn_sigmoid(float x)
{
#if SNN_MATH == 9   // timing ablation only: no transcendental work
    return x * 0.125f;
#elif SNN_MATH == 0
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
#elif SNN_MATH == 1
    const float l2e_hi = 1.44269502162933349609375f, l2e_lo = 1.925963033500011e-08f;
    const float nx = -x, ph = nx * l2e_hi;
    const float pl = fmaf(nx, l2e_hi, -ph) + nx * l2e_lo;
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ph) * (1.0f + pl * 0.693147180559945f));
#else
    return 1.0f / (1.0f + expf(-x));
#endif
}
__device__ __forceinline__ float snn_tanh(float x)
{
#if SNN_MATH == 9
    return x * 0.0625f;
#elif SNN_MATH == 0
    // 1 - 2/(1 + e^{2x}): e -> inf gives 1, e -> 0 gives -1; cancellation near 0 costs absolute ~6e-8 only
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008177792681f)), 1.0f);
#elif SNN_MATH == 1
    return fmaf(2.0f, snn_sigmoid(2.0f * x), -1.0f);
#else
    return tanhf(x);
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
// Bernoulli(keep) * keep_scale for (draw, stream, a, b).  One Philox block serves the four elements b & ~3 .. b | 3
// (b is the feature index in every caller), so a thread that needs several consecutive features of one curve
// evaluates the block once (the compiler merges the identical calls).
__device__ __forceinline__ float philox_mask(uint64_t seed, uint64_t draw, uint32_t stream, uint32_t a,
                                             uint32_t b, float keep_prob, float keep_scale)
{
    Philox4 r = philox4x32_10(a, b >> 2, (uint32_t)draw, stream, (uint32_t)seed,
                              (uint32_t)(seed >> 32) ^ (uint32_t)(draw >> 32));
    const uint32_t sel = b & 3u;
    const uint32_t v = sel == 0 ? r.x : sel == 1 ? r.y : sel == 2 ? r.z : r.w;
    return ((v >> 8) * 5.9604644775390625e-08f) < keep_prob ? keep_scale : 0.0f;
}
