// snn_bwd_dx.hip -- dy of the layer below for wide layers (dirs * H / 32 = 4 or 8 output tiles), chip-filling batches:
//     dy_below[t][f][curve] = mask (.) sum_dir sum_R W_ih[R][f] dG_t[dir][R][curve]
// a plain GEMM over (tile, t): M = dirs * H features, N = 32 curves x 4 steps, K = dirs * 4H gate rows.
//
// bwd_dx_kernel (snn_train_big.hip) streams both operands from L2 into registers, every wave for itself: the four
// waves of a workgroup each fetch the same dG rows (and fetch them again for their second output tile) -- 5.2 MB of
// L2 -> L1 traffic per (tile, 4 steps) at H = 128, 1.25 load instructions per MFMA, no read-ahead; its MFMA pipe was
// 55 % busy (profiles/r3).  Here
//   * a workgroup covers (tile, 4 steps) with ALL its output tiles at once: wave w holds tiles {w, w + 4} x 4 steps in
//     registers (8 accumulators at H = 128 bidirectional), so every dG row is used for all 8 output tiles;
//   * dG goes global -> registers -> LDS once per workgroup, in chunks of 16 k-steps x 4 steps (16 KB), double-buffered:
//     the loads of chunk c + 1 are issued before the MFMAs of chunk c, written to the other buffer after them, one
//     barrier per chunk.  The rows of a k-step are 64 contiguous floats in dG, which IS the B-operand order: the LDS
//     image is a straight copy (16-byte writes, 4-byte reads, no bank conflicts);
//   * W_ih^T (the A operand, 1 MB at H = 128, L2-resident) goes global -> registers one chunk ahead.
// Same products in the same order as bwd_dx_kernel: bit-identical dy.  Every load is unconditional at a clamped
// address (no branch in the chunk loop).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snn_kernels.h"
#include "snn_host.h"

constexpr int DX_KC = 16;     // k-steps per chunk
constexpr int DX_TS = 4;      // time steps per workgroup

template <int H, int MPW>     // MPW output tiles per wave: dirs * H / 32 = 4 * MPW
__global__ void __launch_bounds__(256, 2) bwd_dx_lds_kernel(const Bwd2Params p, int l, int n_tchunks)
{
    __shared__ __attribute__((aligned(16))) float bs[2][DX_TS][DX_KC * 64];
    const SnnLayout& y = p.lay;
    const int dirs = y.dirs;
    const int tile = blockIdx.x / n_tchunks, t0 = (blockIdx.x - tile * n_tchunks) * DX_TS;
    const int T = __builtin_amdgcn_readfirstlane(p.tile_T[tile]);
    if (t0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63, cl = lane & 31, hf = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NCD = 2 * H / DX_KC;          // chunks per direction
    const int NC = dirs * NCD;
    int tc[DX_TS];                              // steps beyond the buffer re-read its last one (never stored)
#pragma unroll
    for (int s = 0; s < DX_TS; ++s) tc[s] = t0 + s < p.max_T ? t0 + s : p.max_T - 1;

    f32x16 acc[MPW][DX_TS];
#pragma unroll
    for (int m = 0; m < MPW; ++m)
#pragma unroll
        for (int s = 0; s < DX_TS; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][s][i] = 0.0f;

    f32x4 bst[DX_TS];            // this thread's 16 bytes of each step's chunk image
    float an[MPW][DX_KC];        // A operands of the next chunk
    // the DX_TS + MPW * DX_KC loads of a chunk, dealt out one per k-step (and the rest behind the first MFMAs): issued in
    // one burst they queue at the compute unit's one address unit in front of the chunk's MFMAs
    const float *Bf = nullptr, *Af = nullptr;
    auto fetch_setup = [&](int c) __attribute__((always_inline)) {
        const int d = c / NCD, kc = c - d * NCD;
        const int ld = l * dirs + d;
        Bf = p.dG + ((int64_t)tile * dirs + d) * p.max_T * (4 * H * 32) + kc * (DX_KC * 64) + tid * 4;
        Af = p.WT + p.wt_ih[ld] + kc * (DX_KC * 64) + lane;
    };
    constexpr int NPC = DX_TS + MPW * DX_KC;     // pieces per chunk
    auto fetch_piece = [&](int j) __attribute__((always_inline)) {
        if (j < DX_TS) bst[j] = *reinterpret_cast<const f32x4*>(Bf + (int64_t)tc[j] * (4 * H * 32));
        else {
            const int m = (j - DX_TS) / DX_KC, k = (j - DX_TS) % DX_KC;
            an[m][k] = Af[(int64_t)(w + 4 * m) * 2 * H * 64 + k * 64];
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < DX_TS; ++s) *reinterpret_cast<f32x4*>(&bs[buf][s][tid * 4]) = bst[s];
    };

    fetch_setup(0);
#pragma unroll
    for (int j = 0; j < NPC; ++j) fetch_piece(j);
    commit(0);
    float ac[MPW][DX_KC];
#pragma unroll
    for (int m = 0; m < MPW; ++m)
#pragma unroll
        for (int k = 0; k < DX_KC; ++k) ac[m][k] = an[m][k];
    __syncthreads();
    constexpr int PPK = (NPC + DX_KC - 1) / DX_KC;   // pieces behind each k-step
    for (int c = 0; c < NC; ++c) {
        const int buf = c & 1;
        fetch_setup(c + 1 < NC ? c + 1 : c);
        __builtin_amdgcn_sched_barrier(0);
        const float* bc = &bs[buf][0][lane];
#pragma unroll
        for (int k = 0; k < DX_KC; ++k) {
            float b[DX_TS];
#pragma unroll
            for (int s = 0; s < DX_TS; ++s) b[s] = bc[s * (DX_KC * 64) + k * 64];
#pragma unroll
            for (int m = 0; m < MPW; ++m)
#pragma unroll
                for (int s = 0; s < DX_TS; ++s) {
                    acc[m][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[m][k], b[s], acc[m][s], 0, 0, 0);
                    if (m == 0 && s < PPK && k * PPK + s < NPC) fetch_piece(k * PPK + s);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        commit(buf ^ 1);
#pragma unroll
        for (int m = 0; m < MPW; ++m)
#pragma unroll
            for (int k = 0; k < DX_KC; ++k) ac[m][k] = an[m][k];
        __syncthreads();
    }

    const int orig = p.order[tile * SNN_TILE + cl];
    const bool masked = p.mk.mode != SNN_MASK_NONE;
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
        const int mt = w + 4 * m;
#pragma unroll
        for (int s = 0; s < DX_TS; ++s) {
            const int t = t0 + s;
            if (t < T) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int f = 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * hf;
                    float v = acc[m][s][i];
                    if (masked && orig >= 0) v = __fmul_rn(mask_mid(p.mk, 0, y.L, p.n_curves, dirs * H, l - 1, t, orig, f), v);
                    const int dd = f / H, j = f - dd * H;
                    p.dy_out[(((int64_t)tile * dirs + dd) * p.max_T + t) * H * 32 + j * 32 + cl] = v;
                }
            }
        }
    }
}

// true: launched.  false: this shape is bwd_dx_kernel's (fewer than 4 output tiles).
bool snn_launch_bwd_dx_lds(const Bwd2Params& p, int l, hipStream_t st)
{
    const SnnLayout& y = p.lay;
    const int mtx = y.dirs * y.H / 32;
    const int n_tchunks = (p.max_T + DX_TS - 1) / DX_TS;
    const dim3 grid(p.n_tiles * n_tchunks), block(256);
    if (y.H == 128 && mtx == 8) hipLaunchKernelGGL((bwd_dx_lds_kernel<128, 2>), grid, block, 0, st, p, l, n_tchunks);
    else if (y.H == 128 && mtx == 4) hipLaunchKernelGGL((bwd_dx_lds_kernel<128, 1>), grid, block, 0, st, p, l, n_tchunks);
    else if (y.H == 64 && mtx == 4) hipLaunchKernelGGL((bwd_dx_lds_kernel<64, 1>), grid, block, 0, st, p, l, n_tchunks);
    else return false;
    return true;
}
