// snn_forward_pair.hip -- persistent LSTM forward, two 32-curve tiles per workgroup (inference; MASKED variant for the
// locked variational masks, applied exactly as lstm_fwd_kernel applies them: same products, same bits).
//
// Same decomposition as lstm_fwd_kernel (wave w owns the 4 gates of hidden units 8w..8w+7, weights of the
// running (layer, direction) in registers, B operand = [x_t | h_{t-1}] of 32 curves from LDS), but every
// work item is (PAIR of adjacent tiles, MC sample) and the time loop is software-pipelined across the pair:
//
//     phase I : MFMA chain of tile 1, step t      ||  activations of tile 0, step t   -> h0_t to LDS
//     barrier
//     phase II: MFMA chain of tile 0, step t + 1  ||  activations of tile 1, step t   -> h1_t to LDS
//     barrier
//
// so one wave always has an MFMA chain in flight whose results it does not need yet, and the sigmoid/tanh
// VALU work of the other tile issues in its shadow (64 cycles per v_mfma_f32_32x32x2_f32 on the SIMD's matrix
// pipe against 4-8 cycles of vector issue per VALU instruction).  The one-tile kernel depends on OTHER waves
// of the SIMD to fill the pipe while a wave computes activations; measured, that leaves the pipe idle 25 % of
// the time (profiles/r1), and neither more waves nor removing the barrier changes it.
// Weight registers are shared by both tiles: half the weight loads per curve.  Because a tile's chain and its
// activations never run concurrently in different waves, x and h need ONE LDS buffer per tile (no ping-pong).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "snn_kernels.h"
#include "snn_host.h"

#ifndef SNN_PAIR_WAVES
#define SNN_PAIR_WAVES 3
#endif
#ifndef SNN_PAIR_MASKED_WAVES
#define SNN_PAIR_MASKED_WAVES 3   // 2 -> 3: 775 -> 744 ms per 1 M curves x 50 samples (spills and all)
#endif

struct PairCtx {
    int s, T, w, lane, cl, hf, tid;
    int tile[2];
    int Tt[2];              // steps stored in each tile's packed image (loads are clamped to it)
    int len[2], orig[2];    // of this lane's curve in each tile
    const float* wks;
    float* ybase;           // this workgroup's y scratch: [nbuf][2 tiles][dirs][max_T][H*32]
};

// UNIFORM: every curve of both tiles runs all T steps of the pair (the usual case: curves are sorted by length, and a
// length class of a large batch spans many tiles) -- the per-lane "still inside the curve" selects of the cell update
// drop out (3 of ~32 vector instructions per hidden unit and step, on the pipe the fp32 MFMAs share).  Same values.
// exp2 factor of gate block `gate` (0 i, 1 f, 2 g, 3 o), and the activations of pre-scaled arguments
__device__ __forceinline__ float gate_exp2_scale(int gate) { return gate == 2 ? 2.88539008177792681f : -1.44269504088896341f; }
__device__ __forceinline__ float rcp_1p_exp2(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z)); }
__device__ __forceinline__ float sigmoid_exp2(float z) { return rcp_1p_exp2(z); }     // z = -a log2 e
constexpr float K2 = 2.88539008177792681f;                                             // 2 log2 e

template <int H, int KKI, int DIRS, bool FIRST, bool LAST, bool MASKED, bool UNIFORM>
__device__ __forceinline__ void pair_phase(const FwdParams& p, const PairCtx& c, int l, int d, int k_A, int k_bias,
                                           int k_head_w, float* xin, float* hb, float* part, float* logit, float* bl)
{
    constexpr int NT = H * 8, NW = H / 8;
    constexpr int ROWS = 2 * KKI, NF4 = ROWS * 8, NPT = (NF4 + NT - 1) / NT, XIN = ROWS * 32, HB = H * 32;
    constexpr int KKH = H / 2, KK = KKI + KKH;
    const FwdParams& y = p;
    const int ld = l * DIRS + d;
    constexpr bool last_layer = LAST;
    const int T = c.T;
    const int64_t y_tile = (int64_t)DIRS * p.max_T * HB;   // floats of one tile's y block

    // ---- weights -> registers, shared by both tiles
    float wreg[KK];
    {
        const float* A = c.wks + k_A + c.w * KK * 64 + c.lane;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wreg[kk] = A[kk * 64];
    }
    const MaskP& mk = p.mk;
    // variational W_hh row mask: forward direction only (variational_rnn.py:45, 224-238)
    if (MASKED && d == 0) {
        const int r = c.lane & 31;
        const float m = mask_whh(mk, c.s, y.L, H, l, (r >> 3) * H + 8 * c.w + (r & 7));
#pragma unroll
        for (int kk = KKI; kk < KK; ++kk) wreg[kk] = __fmul_rn(m, wreg[kk]);
    }
    // The exp2 factor of each gate's activation goes into its weight rows and bias, once per phase: sigmoid(a) =
    // 1 / (1 + 2^(-a log2 e)), tanh(a) = 1 - 2 / (1 + 2^(2 a log2 e)), so rows of i, f, o carry -log2 e and rows of g
    // 2 log2 e, and the accumulator IS the exponent -- one multiplication less per activation in the step loop, on the
    // issue pipe the fp32 MFMAs share (the cell state is carried as 2 log2 e * c for the same reason, see act_piece).
    // One extra fp32 rounding per weight; parity bar unchanged (tests against the oracle and the reference fixtures).
    {
        const float sc = gate_exp2_scale((c.lane & 31) >> 3);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wreg[kk] = __fmul_rn(sc, wreg[kk]);
    }
    if (c.cl < 16) {
        const float* B = c.wks + k_bias + (c.w * 2 + c.hf) * 16;
        bl[(c.w * 2 + c.hf) * 16 + c.cl] = __fmul_rn(gate_exp2_scale(c.cl >> 2), B[c.cl] + B[NW * 32 + c.cl]);
    }
    const f32x16* binit_p = reinterpret_cast<const f32x16*>(bl + (c.w * 2 + c.hf) * 16);
    const int j0 = 8 * c.w + 4 * c.hf;
    // locked between-layer mask of this lane's 4 units, and the layer-0 input mask of the staging threads
    float mmid[2][4], min_[2][NPT][4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            mmid[q][u] = (MASKED && !last_layer && c.orig[q] >= 0)
                             ? mask_mid(mk, c.s, y.L, p.n_curves, DIRS * H, l, 0, c.orig[q], d * H + j0 + u) : 1.0f;
#pragma unroll
        for (int i = 0; i < NPT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) min_[q][i][e] = 1.0f;
        if (MASKED && FIRST) {
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const int f = c.tid + i * NT < NF4 ? c.tid + i * NT : NF4 - 1;
                const int k = f >> 3, c4 = (f & 7) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = p.order[c.tile[q] * SNN_TILE + c4 + e];
                    if (o >= 0 && k < y.D) min_[q][i][e] = mask_in(mk, c.s, p.n_curves, y.D, o, k);
                }
            }
        }
    }

    const float* in0[2];
    float* yout[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (FIRST) in0[q] = p.xp + p.tile_off[c.tile[q]];
        else in0[q] = c.ybase + (int64_t)((l - 1) % p.nbuf) * p.y_buf_floats + q * y_tile;
        yout[q] = last_layer ? nullptr : c.ybase + (int64_t)(l % p.nbuf) * p.y_buf_floats + q * y_tile + (int64_t)d * p.max_T * HB;
    }

    auto load_step = [&](int q, int t, f32x4 (&pre)[NPT]) {
        if (FIRST) t = t < c.Tt[q] ? t : c.Tt[q] - 1;   // tile q may hold fewer steps than the pair runs
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            // threads beyond the image duplicate its last float4 (same value to the same LDS address): no branch,
            // so a whole pipeline phase stays one scheduling region
            const int f = c.tid + i * NT < NF4 ? c.tid + i * NT : NF4 - 1;
            const float* src;
            if (FIRST) src = in0[q] + (int64_t)t * XIN + f * 4;
            else {
                const int dd = f / (H * 8), ff = f - dd * (H * 8);
                src = in0[q] + ((int64_t)dd * p.max_T + t) * HB + ff * 4;
            }
            pre[i] = *reinterpret_cast<const f32x4*>(src);
        }
    };
    auto store_step = [&](int q, const f32x4 (&pre)[NPT]) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = c.tid + i * NT < NF4 ? c.tid + i * NT : NF4 - 1;
            f32x4 v = pre[i];
            if (MASKED && FIRST) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __fmul_rn(min_[q][i][e], v[e]);
            }
            *reinterpret_cast<f32x4*>(xin + q * XIN + f * 4) = v;
        }
    };
    // B operand k of tile q: [x_t | h_{t-1}] rows 2k, 2k+1 (lane = (row & 1) * 32 + curve)
    auto bop = [&](int q, int kk) -> float {
        return kk < KKI ? xin[q * XIN + kk * 64 + c.lane] : hb[q * HB + (kk - KKI) * 64 + c.lane];
    };

    float cst[2][4], hst[2][4], hsum[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) cst[q][u] = hst[q][u] = hsum[q][u] = 0.0f;

    // The cell update of one tile-step, cut into NPIECE pieces placed behind the MFMAs of the other tile's chain.  A piece
    // works on TWO hidden units (j0 + 2 up, j0 + 2 up + 1) at the same stage: each unit's stage is a dependent chain
    // (v_exp -> v_add -> v_rcp -> v_fma), and a transcendental's consumer right behind it waits out its latency (the
    // compiler also has to put an s_nop between them); with two independent chains in one scheduling region the
    // instructions interleave and neither wait is paid.  SNN_PAIR_UNITS_PER_PIECE=1: one unit per piece (24 pieces).
#ifndef SNN_PAIR_UNITS_PER_PIECE
#define SNN_PAIR_UNITS_PER_PIECE 2
#endif
    constexpr int UPP = SNN_PAIR_UNITS_PER_PIECE;
    constexpr int NPIECE = 24 / UPP;
    static_assert(KK >= NPIECE, "one cell-update piece per MFMA at most");
    struct ActRegs { float ig[4], fg[4], gg[4], og[4], tc[4]; };
    auto act_piece = [&](int q, int t, const f32x16& acc, ActRegs& r, unsigned long long vmask, int piece) {
        const int up = piece / 6, st = piece % 6;
        // `pin`: an empty volatile asm on the piece's input.  Volatile asms and sched_barrier keep their program
        // order, so the piece cannot be hoisted to the head of the chain (which the compiler otherwise does with
        // everything that only depends on the previous chain's accumulator).
        auto pin = [](float v) { asm volatile("" : "+v"(v)); return v; };
        // acc holds the exp2 exponents (scaled rows); gg and the cell state are carried times K2 = 2 log2 e
#pragma unroll
        for (int uu = 0; uu < UPP; ++uu) {
            const int u = up * UPP + uu;
            if (st == 0) r.ig[u] = sigmoid_exp2(pin(acc[u]));
            else if (st == 1) r.gg[u] = fmaf(-2.0f * K2, rcp_1p_exp2(pin(acc[8 + u])), K2);                   // K2 * tanh(a_g)
            else if (st == 2) r.fg[u] = sigmoid_exp2(pin(acc[4 + u]));
            else if (st == 3) {
                const float cn = fmaf(pin(r.fg[u]), cst[q][u], __fmul_rn(r.ig[u], r.gg[u]));                    // K2 * c_t
                r.tc[u] = fmaf(-2.0f, rcp_1p_exp2(cn), 1.0f);                                                   // tanh(c_t)
                cst[q][u] = UNIFORM ? cn : lane_select(vmask, cn, cst[q][u]);
            } else if (st == 4) r.og[u] = sigmoid_exp2(pin(acc[12 + u]));
            else {
                const float hv = pin(r.og[u]) * r.tc[u];
                hst[q][u] = UNIFORM ? hv : lane_select(vmask, hv, hst[q][u]);
                const float outv = UNIFORM ? hv : lane_select(vmask, hv, 0.0f);
                hb[q * HB + (j0 + u) * 32 + c.cl] = hst[q][u];
                if (last_layer) hsum[q][u] += outv;
                else yout[q][(int64_t)t * HB + (j0 + u) * 32 + c.cl] = MASKED ? __fmul_rn(mmid[q][u], outv) : outv;
            }
        }
    };
    // One pipeline phase: the MFMA chain of tile qc (into accc), with the cell update of tile qa (from acca, step ta)
    // issued piecewise behind the MFMAs.  sched_barrier(0) pins the program order: nothing crosses it.
    constexpr int BAHEAD = 4;   // B operands are read from LDS this many MFMAs ahead
    auto phase = [&](int qc, f32x16& accc, bool do_act, int qa, int ta, const f32x16& acca) {
        const unsigned long long vmask = __builtin_amdgcn_ballot_w64(ta < c.len[qa]);
        ActRegs r;
        float bv[KK];
#pragma unroll
        for (int kk = 0; kk < BAHEAD; ++kk) bv[kk] = bop(qc, kk);
        accc = *binit_p;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if (kk + BAHEAD < KK) bv[kk + BAHEAD] = bop(qc, kk + BAHEAD);
            accc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[kk], bv[kk], accc, 0, 0, 0);
            // piece p goes behind MFMA (p * KK) / NPIECE: evenly spread over the chain
            if (do_act) {   // KK >= NPIECE: at most one piece per MFMA
                const int pc = (kk * NPIECE + KK - 1) / KK;
                if (pc < NPIECE && (pc * KK) / NPIECE == kk) act_piece(qa, ta, acca, r, vmask, pc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto act_only = [&](int qa, int ta, const f32x16& acca) {
        const unsigned long long vmask = __builtin_amdgcn_ballot_w64(ta < c.len[qa]);
        ActRegs r;
#pragma unroll
        for (int pc = 0; pc < NPIECE; ++pc) act_piece(qa, ta, acca, r, vmask, pc);
    };

    // ---- prologue: x(t0) of both tiles, h = 0, first chain of tile 0
    const int t0 = d == 0 ? 0 : T - 1;
    {
        f32x4 pre[NPT];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            load_step(q, t0, pre);
            store_step(q, pre);
#pragma unroll
            for (int u = 0; u < 4; ++u) hb[q * HB + (j0 + u) * 32 + c.cl] = 0.0f;
        }
    }
    lds_barrier();
    f32x16 acc0, acc1;
    phase(0, acc0, false, 0, 0, acc0);
    lds_barrier();   // every wave has read h0 = 0 before the first activations overwrite it

    for (int step = 0; step + 1 < T; ++step) {
        const int t = d == 0 ? step : T - 1 - step;
        const int tn = d == 0 ? t + 1 : t - 1;
        f32x4 pre0[NPT], pre1[NPT];
        load_step(0, tn, pre0);
        // phase I: chain of tile 1 (step t) || cell update of tile 0 (step t)
        phase(1, acc1, true, 0, t, acc0);
        store_step(0, pre0);
        load_step(1, tn, pre1);
        lds_barrier();
        // phase II: chain of tile 0 (step t+1) || cell update of tile 1 (step t)
        phase(0, acc0, true, 1, t, acc1);
        store_step(1, pre1);
        lds_barrier();
    }
    {   // last step: nothing left to prefetch
        const int t = d == 0 ? T - 1 : 0;
        phase(1, acc1, true, 0, t, acc0);
        lds_barrier();   // tile 1's cell update overwrites the h image the other waves' chains may still be reading
        act_only(1, t, acc1);
    }

    // ---- head contribution of this phase, per tile
    const bool contributes = y.pool == SNN_POOL_STANDARD || last_layer;
    if (contributes) {
        const float* Wo = c.wks + k_head_w;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float feat[4];
            int fidx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (y.pool == SNN_POOL_MEAN) {
                    feat[u] = __fdiv_rn(hsum[q][u], (float)(c.len[q] > 0 ? c.len[q] : 1));
                    fidx[u] = d * H + j0 + u;
                } else {
                    feat[u] = hst[q][u];
                    fidx[u] = (j0 + u) * (y.L * DIRS) + ld;
                }
            }
            if (MASKED && c.orig[q] >= 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) feat[u] = __fmul_rn(mask_out(mk, c.s, p.n_curves, y.F, c.orig[q], fidx[u]), feat[u]);
            }
            for (int cq = 0; cq < y.C; ++cq) {
                float a = 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u) a = fmaf(Wo[cq * y.F + fidx[u]], feat[u], a);
                a += __shfl_xor(a, 32);
                if (c.hf == 0) part[((q * NW + c.w) * y.C + cq) * 32 + c.cl] = a;
            }
        }
        __syncthreads();
        for (int idx = c.tid; idx < 2 * y.C * 32; idx += NT) {
            const int q = idx / (y.C * 32), r = idx - q * (y.C * 32);
            const int cq = r >> 5, cc = r & 31;
            float a = logit[idx];
            for (int w = 0; w < NW; ++w) a += part[((q * NW + w) * y.C + cq) * 32 + cc];
            logit[idx] = a;
        }
    }
    __syncthreads();  // also publishes this phase's y stores to the workgroup
}

template <int H, int KKI0, int DIRS, bool MASKED>
__global__ void __launch_bounds__(H * 8, MASKED ? SNN_PAIR_MASKED_WAVES : SNN_PAIR_WAVES) lstm_fwd_pair_kernel(const FwdParams p)
{
    constexpr int NT = H * 8, NW = H / 8;
    constexpr int KKI1 = DIRS * H / 2;
    constexpr int ROWS = (2 * KKI0 > 2 * KKI1) ? 2 * KKI0 : 2 * KKI1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdParams& y = p;
    constexpr int BLK0 = NW * (KKI0 + H / 2) * 64 + 2 * NW * 32;
    constexpr int BLK1 = NW * (KKI1 + H / 2) * 64 + 2 * NW * 32;
    const int k_head_w = DIRS * BLK0 + (y.L - 1) * DIRS * BLK1;
    float* xin = smem;                           // [2 tiles][ROWS*32]
    float* hb = xin + 2 * ROWS * 32;             // [2 tiles][H*32]
    float* part = hb + 2 * H * 32;               // [2][NW][C][32]
    float* logit = part + 2 * NW * y.C * 32;     // [2][C][32]
    float* bl = logit + 2 * y.C * 32;            // [NW][2][16]
    int* item_slot = reinterpret_cast<int*>(bl + NW * 32);

    PairCtx c;
    c.tid = threadIdx.x;
    c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.cl = c.lane & 31;
    c.hf = c.lane >> 5;
    c.ybase = p.yscratch + (int64_t)blockIdx.x * p.nbuf * p.y_buf_floats;
    const int n_pairs = (p.n_tiles + 1) >> 1;
    const int64_t n_items = (int64_t)n_pairs * p.n_samples;

    for (;;) {
        if (c.tid == 0) *item_slot = (int)atomicAdd(p.counter, 1u);
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(*item_slot);
        if (item >= n_items) break;
        const int pair = item / p.n_samples;
        c.s = item - pair * p.n_samples;
        const bool has1 = 2 * pair + 1 < p.n_tiles;
        c.tile[0] = 2 * pair;
        c.tile[1] = has1 ? 2 * pair + 1 : 2 * pair;      // a missing partner aliases tile 0's data, with no live curve
        c.Tt[0] = __builtin_amdgcn_readfirstlane(p.tile_T[c.tile[0]]);
        c.Tt[1] = __builtin_amdgcn_readfirstlane(p.tile_T[c.tile[1]]);
        c.T = c.Tt[0];                                   // tiles are sorted by length, descending
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pos = c.tile[q] * SNN_TILE + c.cl;
            const bool live = q == 0 || has1;
            c.len[q] = live ? p.len_sorted[pos] : 0;
            c.orig[q] = live ? p.order[pos] : -1;
        }
        c.wks = p.wk + (int64_t)c.s * p.wk_stride;
        for (int idx = c.tid; idx < 2 * y.C * 32; idx += NT) logit[idx] = 0.0f;
        __syncthreads();

        // all 64 curves as long as the pair's first one (lengths are sorted, descending): the UNIFORM instantiation
        const bool uniform = has1 && __builtin_amdgcn_readfirstlane(p.len_sorted[c.tile[1] * SNN_TILE + SNN_TILE - 1]) == c.T;
        for (int l = 0; l < y.L; ++l) {
            const bool last = l == y.L - 1;
            for (int d = 0; d < DIRS; ++d) {
#define SNN_PAIR_PHASE(KKI_, FIRST_, LAST_)                                                                                   \
    do {                                                                                                                      \
        if (uniform) pair_phase<H, KKI_, DIRS, FIRST_, LAST_, MASKED, true>(p, c, l, d, kA, kB, k_head_w, xin, hb, part, logit, bl);  \
        else pair_phase<H, KKI_, DIRS, FIRST_, LAST_, MASKED, false>(p, c, l, d, kA, kB, k_head_w, xin, hb, part, logit, bl);         \
    } while (0)
                if (l == 0) {
                    const int kA = d * BLK0, kB = kA + NW * (KKI0 + H / 2) * 64;
                    if (last) SNN_PAIR_PHASE(KKI0, true, true);
                    else SNN_PAIR_PHASE(KKI0, true, false);
                } else {
                    const int kA = DIRS * BLK0 + ((l - 1) * DIRS + d) * BLK1, kB = kA + NW * (KKI1 + H / 2) * 64;
                    if (last) SNN_PAIR_PHASE(KKI1, false, true);
                    else SNN_PAIR_PHASE(KKI1, false, false);
                }
#undef SNN_PAIR_PHASE
            }
        }

        // ---- head bias + softmax, one thread per curve (64 curves)
        if (c.tid < 64) {
            const int q = c.tid >> 5, cc = c.tid & 31;   // tid < 64: lane == tid, cl == cc, hf == q
            const int orig = q == 0 ? c.orig[0] : c.orig[1];
            if (orig >= 0) {
                const float* lg_ = logit + q * y.C * 32;
                const float* bo = c.wks + k_head_w + y.C * y.F;
                float mx = -INFINITY;
                for (int k = 0; k < y.C; ++k) mx = fmaxf(mx, lg_[k * 32 + cc] + bo[k]);
                float z = 0.0f;
                for (int k = 0; k < y.C; ++k) z += expf(lg_[k * 32 + cc] + bo[k] - mx);
                const int64_t o = (int64_t)c.s * p.os_s + (int64_t)orig * p.os_n;
                for (int k = 0; k < y.C; ++k) {
                    const float lg = lg_[k * 32 + cc] + bo[k];
                    if (p.out_logits) p.out_logits[o + k] = lg;
                    if (p.out_probs) p.out_probs[o + k] = __fdiv_rn(expf(lg - mx), z);
                }
            }
        }
        __syncthreads();
    }
}

// ---- registry hook used by snn_forward.hip --------------------------------------------------------
typedef void (*fwd_kernel_t)(const FwdParams);
fwd_kernel_t snn_pair_kernel(int H, int KKI0, int dirs, bool masked)
{
    if (H == 32 && (dirs == 2 || dirs == 1)) {   // every layer-0 width bucket of snn_layout.h, both direction counts
#define SNN_PAIR_CASE(K)                                                                                                   \
    if (KKI0 == K) {                                                                                                       \
        if (dirs == 2) return masked ? lstm_fwd_pair_kernel<32, K, 2, true> : lstm_fwd_pair_kernel<32, K, 2, false>;        \
        return masked ? lstm_fwd_pair_kernel<32, K, 1, true> : lstm_fwd_pair_kernel<32, K, 1, false>;                      \
    }
        SNN_PAIR_CASE(12)
        SNN_PAIR_CASE(13)
        SNN_PAIR_CASE(24)
        SNN_PAIR_CASE(38)
        SNN_PAIR_CASE(39)
#undef SNN_PAIR_CASE
    }
    return nullptr;
}
