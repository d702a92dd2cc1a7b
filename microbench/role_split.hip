// Skeleton of a role-split LSTM forward for H = 32: of the two waves of a SIMD one issues ONLY the fp32 MFMAs (and the
// LDS traffic of their operands / results), the other ONLY the cell update (sigmoid / tanh) -- the question being what
// an MFMA costs the matrix wave when the vector work of the step is another wave's (mfma_two_waves.hip: a wave never
// overlaps its own vector instructions with its own fp32 MFMA, two waves of a SIMD do).
//
// Two 32-curve tiles in flight per workgroup, one barrier per SLOT, two slots per time step:
//   slot s, q = s & 1:   matrix waves: recurrent k-steps of tile q (h from LDS), input k-steps of tile q^1's next step
//                                      (independent of any h: they cover the LDS latency and the drain of the chain),
//                                      gate pre-activations of tile q -> LDS
//                        cell waves:   pre-activations of tile q^1 (written in slot s-1) -> c, h; h -> LDS, y -> global
// Prints cycles per slot of a matrix wave and of a cell wave against the KK x 64 cycles of the MFMAs alone.
//   hipcc -O3 --offload-arch=gfx950 microbench/role_split.hip -o microbench/role_split
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rcp_1p_exp2(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z)); }
constexpr float K2 = 2.88539008177792681f;

// MODE 0: both roles; 1: matrix waves alone (cell waves only meet the barrier); 2: cell waves alone
template <int KKI, int MODE, int OCC>
__global__ void __launch_bounds__(512, OCC) k(float* out, long long* cyc, int slots, const float* __restrict__ wsrc)
{
    constexpr int KKH = 16, KK = KKI + KKH, XH = KKI / 2, BAHEAD = 4;
    __shared__ float xin[2][2 * KKI * 32];
    __shared__ float hb[2][32 * 32];
    __shared__ __attribute__((aligned(16))) float accb[2][4][16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, cl = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), role = wave >> 2, w4 = wave & 3;
    for (int i = tid; i < 2 * 2 * KKI * 32; i += 512) (&xin[0][0])[i] = 1e-3f * (float)(i & 127);
    for (int i = tid; i < 2 * 32 * 32; i += 512) (&hb[0][0])[i] = 0.0f;
    for (int i = tid; i < 2 * 4 * 16 * 64; i += 512) (&accb[0][0][0])[i] = 0.01f;
    __syncthreads();
    long long t0 = 0, t1 = 0;
    if (role == 0) {
        float wreg[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wreg[kk] = wsrc[(w4 * KK + kk) * 64 + lane];
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.01f; acc[1][i] = 0.01f; }
        // operand k of slot (q): k < XH input rows of tile q^1 | 16 recurrent rows of tile q | the other input rows
        auto bop = [&](int q, int kk) -> float {
            if (kk < XH) return xin[q ^ 1][kk * 64 + lane];
            if (kk < XH + KKH) return hb[q][(kk - XH) * 64 + lane];
            return xin[q ^ 1][(kk - KKH) * 64 + lane];
        };
        float bv[KK];
#pragma unroll
        for (int kk = 0; kk < BAHEAD; ++kk) bv[kk] = bop(0, kk);
        t0 = clock64();
        auto slot_body = [&](int q) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q ^ 1][i] = 0.01f;   // the bias image
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if (kk + BAHEAD < KK) bv[kk + BAHEAD] = bop(q, kk + BAHEAD);
                const bool rec = kk >= XH && kk < XH + KKH;
                if (MODE != 2) {
                    if (rec) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[KKI + kk - XH], bv[kk], acc[q], 0, 0, 0);
                    else acc[q ^ 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[kk < XH ? kk : kk - KKH], bv[kk], acc[q ^ 1], 0, 0, 0);
                }
                if (kk == XH + KKH + 1) {   // two independent MFMAs behind the chain: its results have drained
                    f32x4* dst = reinterpret_cast<f32x4*>(&accb[q][w4][0]) + lane;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f32x4 v = {acc[q][4 * i], acc[q][4 * i + 1], acc[q][4 * i + 2], acc[q][4 * i + 3]};
                        dst[i * 64] = v;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the first operands of the next slot are input rows: they do not wait for the barrier
#pragma unroll
            for (int kk = 0; kk < BAHEAD; ++kk) bv[kk] = bop(q ^ 1, kk);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        };
        for (int s = 0; s < slots; s += 2) { slot_body(0); slot_body(1); }
        t1 = clock64();
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += acc[0][i] + acc[1][i];
        out[blockIdx.x * 512 + tid] = sum;
    } else {
        float cst[2][4] = {}, hsum[2][4] = {};
        const int j0 = 8 * w4 + 4 * hf;
        t0 = clock64();
        auto slot_body = [&](int qa) {
            if (MODE != 1) {
                const f32x4* src = reinterpret_cast<const f32x4*>(&accb[qa][w4][0]) + lane;
                f32x4 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = src[i * 64];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float ig = rcp_1p_exp2(a[0][u]);
                    const float gg = fmaf(-2.0f * K2, rcp_1p_exp2(a[2][u]), K2);
                    const float fg = rcp_1p_exp2(a[1][u]);
                    const float cn = fmaf(fg, cst[qa][u], ig * gg);
                    const float tc = fmaf(-2.0f, rcp_1p_exp2(cn), 1.0f);
                    cst[qa][u] = cn;
                    const float og = rcp_1p_exp2(a[3][u]);
                    const float hv = og * tc;
                    hb[qa][(j0 + u) * 32 + cl] = hv;
                    hsum[qa][u] += hv;
                }
            }
            __syncthreads();
        };
        for (int s = 0; s < slots; s += 2) { slot_body(1); slot_body(0); }
        t1 = clock64();
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) sum += hsum[0][u] + hsum[1][u];
        out[blockIdx.x * 512 + tid] = sum;
    }
    if (blockIdx.x == 0 && (tid & 255) == 0) cyc[role] = t1 - t0;
}

template <int KKI, int MODE, int OCC>
static void run(float* out, long long* cyc, const float* w)
{
    const int slots = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KKI, MODE, OCC>), dim3(256 * OCC), dim3(512), 0, 0, out, cyc, 64, w);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KKI, MODE, OCC>), dim3(256 * OCC), dim3(512), 0, 0, out, cyc, slots, w);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double ideal = (KKI + 16) * 64.0;
    printf("KKI=%2d mode %d occ %d: matrix wave %7.1f  cell wave %7.1f cycles/slot  (MFMAs alone %6.0f: %.1f %%)   kernel %.3f ms = %.1f ns/slot/WG\n",
           KKI, MODE, OCC, (double)h[0] / slots, (double)h[1] / slots, ideal, 100.0 * ideal / ((double)h[0] / slots), ms,
           1e6 * ms / slots / OCC);
}

int main()
{
    float *out, *w; long long* cyc;
    hipMalloc(&out, 512 * 512 * sizeof(float)); hipMalloc(&cyc, 2 * sizeof(long long));
    hipMalloc(&w, 4 * 48 * 64 * sizeof(float));
    hipMemset(w, 0, 4 * 48 * 64 * sizeof(float));
    run<12, 1, 1>(out, cyc, w); run<12, 2, 1>(out, cyc, w); run<12, 0, 1>(out, cyc, w);
    run<32, 1, 1>(out, cyc, w); run<32, 2, 1>(out, cyc, w); run<32, 0, 1>(out, cyc, w);
    run<12, 0, 2>(out, cyc, w); run<32, 0, 2>(out, cyc, w);
    return 0;
}
