// snn_fwd_small.h -- the persistent LSTM forward kernel for hidden sizes whose (layer, direction) weights fit
// the register file (H <= 64): template + per-H instantiation lists.  Included by the snn_fwd_h<H>.hip
// translation units (one per hidden size, so the instantiations compile in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snn_kernels.h"

#ifndef SNN_FWD_WAVES
#define SNN_FWD_WAVES 4  // waves per SIMD the un-masked inference kernels are register-allocated for (A/B: 2/3/4 -> 67.5/67.6/69.4 % of peak)
#endif

// ============================================================================
// Persistent LSTM forward
// ============================================================================
template <int H, int KKI, int DIRS>
struct PhaseCfg {
    static constexpr int NT = H * 8;                 // threads per workgroup
    static constexpr int ROWS = 2 * KKI;             // input rows staged per step
    static constexpr int NF4 = ROWS * 8;             // float4 per step
    static constexpr int NPT = (NF4 + NT - 1) / NT;  // float4 per thread per step
};

// One (layer, direction) pass over the tile's T steps.
//   FIRST: input comes from the packed x tile; else from the previous layer's output.
//   TRAIN: activations go to the tape (which also carries the layer outputs, unmasked; the
//          between-layer mask is then applied by the consumer while staging).
//   CELL:  0 LSTM, 1 GRU, 2 tanh RNN (snn_layout.h: the weight image has the LSTM's 4 row groups for every cell).
//   GX:    the input half (bias + W_ih x_t) of every step was computed by fwd_gx_kernel (snn_fwd_gx.hip) and is read
//          from c.gx; the step then issues only the H / 2 recurrent k-steps (small batches: a shorter dependent chain).
template <int H, int KKI, int DIRS, bool FIRST, bool TRAIN, bool MASKED, int CELL = 0, bool GX = false>
__device__ __forceinline__ void run_phase(const FwdParams& p, const ItemCtx& c, int l, int d, int k_A, int k_bias,
                                          int k_head_w, float* xin, float* hb, float* part, float* logit, float* bl)
{
    using PC = PhaseCfg<H, KKI, DIRS>;
    constexpr int NT = PC::NT, NPT = PC::NPT, NF4 = PC::NF4, KKH = H / 2, KK = KKI + KKH;
    constexpr int XIN_STRIDE = PC::ROWS * 32;  // floats per xin buffer for this phase
    constexpr int NW = H / 8;
    const FwdParams& y = p;  // D, L, C, pool, F
    const MaskP& mk = p.mk;
    const int ld = l * DIRS + d;
    const bool last_layer = l == y.L - 1;
    const int T = c.T;
    const bool masked = MASKED && mk.mode != SNN_MASK_NONE;

    // ---- weights -> registers (A operand of every MFMA of this phase)
    float wreg[KK];
    {
        const float* A = c.wks + k_A + c.w * KK * 64 + c.lane;
#pragma unroll
        for (int kk = GX ? KKI : 0; kk < KK; ++kk) wreg[kk] = A[kk * 64];   // GX: the recurrent columns only
    }
    // variational W_hh row mask: forward direction only (variational_rnn.py:45, 224-238)
    if (d == 0 && masked) {
        const int r = c.lane & 31;
        const float m = mask_whh(mk, c.s, y.L, H, l, (r >> 3) * H + 8 * c.w + (r & 7));
#pragma unroll
        for (int kk = KKI; kk < KK; ++kk) wreg[kk] = __fmul_rn(m, wreg[kk]);
    }
    // b_ih + b_hh in C/D register order, parked in LDS (re-read per step: 4 broadcast ds_read_b128)
    if (!GX && c.cl < 16) {
        const float* B = c.wks + k_bias + (c.w * 2 + c.hf) * 16;
        bl[(c.w * 2 + c.hf) * 16 + c.cl] = B[c.cl] + B[NW * 32 + c.cl];
    }
    const f32x16* binit_p = reinterpret_cast<const f32x16*>(bl + (c.w * 2 + c.hf) * 16);

    // ---- per-lane masks
    const int j0 = 8 * c.w + 4 * c.hf;  // first of this lane's 4 hidden units
    float mmid[4] = {1.f, 1.f, 1.f, 1.f};
    const bool mid_step = masked && mk.mid_T > 1;  // per-step between-layer mask (torch.nn.LSTM dropout, train)
    if (!TRAIN && !last_layer && masked && c.orig >= 0 && !mid_step) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            mmid[u] = mask_mid(mk, c.s, y.L, p.n_curves, DIRS * H, l, 0, c.orig, d * H + j0 + u);
    }
    // Inference: staging is branch-free -- a thread beyond the image duplicates its last float4 (same value to the same
    // LDS address) and the step after the last re-stages the last one; loads under `if (f < NF4)` / `if (more)` made the
    // compiler put s_waitcnt vmcnt at the joins (snn_forward_big.hip has the measurement; GRU inference 55.7 -> 59.0 % of
    // peak).  TRAIN keeps the guarded loads: with two workgroups per compute unit covering each other and the tape
    // stores in flight, the same-box comparison read 1.41 ms (guarded) against 1.45 ms (branch-free) for the forward of
    // 65 536 curves.
    int fi[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) fi[i] = c.tid + i * NT < NF4 ? c.tid + i * NT : NF4 - 1;
    // mask registers of the staging threads: layer 0 input mask, or (TRAIN) the between-layer mask
    float min_[NPT][4];
#pragma unroll
    for (int i = 0; i < NPT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) min_[i][e] = 1.0f;
    if (!GX && masked && (FIRST || (TRAIN && !mid_step))) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = fi[i];
            const int k = f >> 3, c4 = (f & 7) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = p.order[c.tile * SNN_TILE + c4 + e];
                if (o >= 0) {
                    if (FIRST) { if (k < y.D) min_[i][e] = mask_in(mk, c.s, p.n_curves, y.D, o, k); }
                    else min_[i][e] = mask_mid(mk, c.s, y.L, p.n_curves, DIRS * H, l - 1, 0, o, k);
                }
            }
        }
    }

    // ---- input source
    const float* in0;  // FIRST: tile base; TRAIN: tape; else y buffer of layer l-1
    if (FIRST) in0 = p.xp + p.tile_off[c.tile];
    else if (TRAIN) in0 = c.tape;
    else in0 = c.ybase + (int64_t)((l - 1) % p.nbuf) * p.y_buf_floats;
    float* yout = (last_layer || TRAIN) ? nullptr
                                        : c.ybase + (int64_t)(l % p.nbuf) * p.y_buf_floats + (int64_t)d * p.max_T * H * 32;
    float* tape_ld = TRAIN ? c.tape + (int64_t)ld * p.max_T * tape_step_floats(H) : nullptr;

    auto load_step = [&](int t, f32x4 (&pre)[NPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = fi[i];
            if (TRAIN && c.tid + i * NT >= NF4) continue;
            const float* src;
            if (FIRST) src = in0 + (int64_t)t * XIN_STRIDE + f * 4;
            else {
                const int dd = f / (H * 8), ff = f - dd * (H * 8);
                if (TRAIN)
                    src = in0 + ((int64_t)((l - 1) * DIRS + dd) * p.max_T + t) * tape_step_floats(H) + 5 * H * 32 + ff * 4;
                else
                    src = in0 + ((int64_t)dd * p.max_T + t) * (H * 32) + ff * 4;
            }
            pre[i] = *reinterpret_cast<const f32x4*>(src);
        }
    };
    auto store_step = [&](float* dstbuf, const f32x4 (&pre)[NPT], int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = fi[i];
            if (TRAIN && c.tid + i * NT >= NF4) continue;
            f32x4 v = pre[i];
            if (MASKED && (FIRST || TRAIN)) {
                if (!FIRST && mid_step) {
                    const int k = f >> 3, c4 = (f & 7) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int o = p.order[c.tile * SNN_TILE + c4 + e];
                        if (o >= 0) v[e] = __fmul_rn(mask_mid(mk, c.s, y.L, p.n_curves, DIRS * H, l - 1, t, o, k), v[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __fmul_rn(min_[i][e], v[e]);
                }
            }
            *reinterpret_cast<f32x4*>(dstbuf + f * 4) = v;
        }
    };

    // ---- state
    float cst[4] = {0.f, 0.f, 0.f, 0.f}, hst[4] = {0.f, 0.f, 0.f, 0.f}, hsum[4] = {0.f, 0.f, 0.f, 0.f};
    // GX: Gx[t][wave's row tile][16][64] of this (tile, direction), fetched one step ahead of its use
    const float* gxw = GX ? c.gx + (int64_t)c.w * 1024 + c.lane : nullptr;
    f32x16 gxn;
    {
        const int t0 = d == 0 ? 0 : T - 1;
        if (GX) {
#pragma unroll
            for (int i = 0; i < 16; ++i) gxn[i] = gxw[(int64_t)t0 * NW * 1024 + i * 64];
        } else {
            f32x4 pre[NPT];
            load_step(t0, pre);
            store_step(xin, pre, t0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) hb[(j0 + u) * 32 + c.cl] = 0.0f;
    }
    lds_barrier();

    for (int step = 0; step < T; ++step) {
        const int t = d == 0 ? step : T - 1 - step;
        const int cur = step & 1;
        f32x4 pre[NPT];
        const bool more = step + 1 < T;
        const int tn = more ? (d == 0 ? t + 1 : t - 1) : t;      // the step after the last: this one again (unused)
        if (!GX && (!TRAIN || more)) load_step(tn, pre);

        f32x16 acc;
        if (GX) {
            acc = gxn;
            if (more) {
#pragma unroll
                for (int i = 0; i < 16; ++i) gxn[i] = gxw[(int64_t)tn * NW * 1024 + i * 64];
            }
            // tape row of the step before: staged in LDS by its cell update (below), written here as 16-byte words --
            // 6 store instructions per thread beside this step's MFMAs instead of 24 4-byte ones between the cell
            // update and the barrier (with one workgroup per compute unit nothing else hides their issue time)
            if (TRAIN && step > 0) {
                const int tprev = d == 0 ? t - 1 : t + 1;
                const float* sg = xin + ((step - 1) & 1) * (SNN_TAPE_Q * H * 32);
                float* dst = tape_ld + (int64_t)tprev * tape_step_floats(H);
#pragma unroll
                for (int i = 0; i < SNN_TAPE_Q; ++i)
                    *reinterpret_cast<f32x4*>(dst + (c.tid + i * NT) * 4) = *reinterpret_cast<const f32x4*>(sg + (c.tid + i * NT) * 4);
            }
        } else {
            acc = *binit_p;
            const float* xc = xin + cur * XIN_STRIDE + c.lane;
#pragma unroll
            for (int kk = 0; kk < KKI; ++kk)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[kk], xc[kk * 64], acc, 0, 0, 0);
        }
        const float* hc = hb + cur * (H * 32) + c.lane;
#pragma unroll
        for (int kk = 0; kk < KKH; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[KKI + kk], hc[kk * 64], acc, 0, 0, 0);

        const bool valid = t < c.len;
        float* hn = hb + (cur ^ 1) * (H * 32) + j0 * 32 + c.cl;
        float outv[4];
        float* tp = TRAIN ? tape_ld + (int64_t)t * tape_step_floats(H) + j0 * 32 + c.cl : nullptr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cn = 0.f, hv;
            if (CELL == 1) {
                // torch.nn.GRU: r, z from the summed pre-activations; group 2 carries W_in x + b_in, group 3 W_hn h + b_hn
                const float rg = snn_sigmoid(acc[u]);
                const float zg = snn_sigmoid(acc[4 + u]);
                const float ng = snn_tanh(fmaf(rg, acc[12 + u], acc[8 + u]));
                hv = fmaf(zg, hst[u] - ng, ng);   // (1 - z) n + z h
                if (TRAIN) { ig = rg; fg = zg; gg = ng; og = acc[12 + u]; }   // tape slots 0..3: r, z, n, W_hn h + b_hn
            } else if (CELL == 2) {
                hv = snn_tanh(acc[u]);
            } else {
            ig = snn_sigmoid(acc[u]);
            fg = snn_sigmoid(acc[4 + u]);
            gg = snn_tanh(acc[8 + u]);
            og = snn_sigmoid(acc[12 + u]);
            // explicit contraction: every copy of the loop body (peeled last step, TRAIN variant) must round the
            // same way, so a curve's result does not depend on the tile it lands in
            cn = fmaf(fg, cst[u], __fmul_rn(ig, gg));
            hv = og * snn_tanh(cn);
            }
            cst[u] = valid ? cn : cst[u];
            hst[u] = valid ? hv : hst[u];
            outv[u] = valid ? hv : 0.0f;
            hn[u * 32] = hst[u];
            if (TRAIN && GX) {
                float* sg = xin + (step & 1) * (SNN_TAPE_Q * H * 32) + (j0 + u) * 32 + c.cl;
                sg[0 * H * 32] = ig;
                sg[1 * H * 32] = fg;
                sg[2 * H * 32] = gg;
                sg[3 * H * 32] = og;
                sg[4 * H * 32] = cn;
                sg[5 * H * 32] = outv[u];
            }
            if (TRAIN && !GX) {
#ifdef SNN_ABL_NOTAPE   // timing ablation only (the backward then reads garbage): what the gate / cell tape stores cost
                if (p.n_curves < 0)
#endif
                {
                tp[(0 * H + u) * 32] = ig;
                tp[(1 * H + u) * 32] = fg;
                tp[(2 * H + u) * 32] = gg;
                tp[(3 * H + u) * 32] = og;
                tp[(4 * H + u) * 32] = cn;
                }
                tp[(5 * H + u) * 32] = outv[u];
            }
        }
        if (last_layer) {
#pragma unroll
            for (int u = 0; u < 4; ++u) hsum[u] += outv[u];
        } else if (!TRAIN) {
#ifdef SNN_ABL_NOYST   // timing ablation only: no layer-output stores
            if (p.n_curves < 0)
#endif
            {
            float* yo = yout + (int64_t)t * (H * 32) + j0 * 32 + c.cl;
            if (mid_step && c.orig >= 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    yo[u * 32] = __fmul_rn(mask_mid(mk, c.s, y.L, p.n_curves, DIRS * H, l, t, c.orig, d * H + j0 + u), outv[u]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) yo[u * 32] = __fmul_rn(mmid[u], outv[u]);
            }
            }
        }
        if (!GX && (!TRAIN || more)) store_step(xin + (cur ^ 1) * XIN_STRIDE, pre, tn);
        lds_barrier();
    }
    if (GX && TRAIN && T > 0) {      // the last step's tape row (its staging writes are behind the loop's closing barrier)
        const int tlast = d == 0 ? T - 1 : 0;
        const float* sg = xin + ((T - 1) & 1) * (SNN_TAPE_Q * H * 32);
        float* dst = tape_ld + (int64_t)tlast * tape_step_floats(H);
#pragma unroll
        for (int i = 0; i < SNN_TAPE_Q; ++i)
            *reinterpret_cast<f32x4*>(dst + (c.tid + i * NT) * 4) = *reinterpret_cast<const f32x4*>(sg + (c.tid + i * NT) * 4);
    }

    // ---- head contribution of this phase
    const bool contributes = y.pool == SNN_POOL_STANDARD || last_layer;
    if (contributes) {
        float feat[4];
        int fidx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (y.pool == SNN_POOL_MEAN) {
                feat[u] = __fdiv_rn(hsum[u], (float)(c.len > 0 ? c.len : 1));
                fidx[u] = d * H + j0 + u;
            } else {
                feat[u] = hst[u];
                fidx[u] = (j0 + u) * (y.L * DIRS) + ld;
            }
        }
        if (masked && c.orig >= 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) feat[u] = __fmul_rn(mask_out(mk, c.s, p.n_curves, y.F, c.orig, fidx[u]), feat[u]);
        }
        if (TRAIN) {
            float* fs = c.tape + (int64_t)y.L * DIRS * p.max_T * tape_step_floats(H);
#pragma unroll
            for (int u = 0; u < 4; ++u) fs[fidx[u] * 32 + c.cl] = feat[u];
        }
        const float* Wo = c.wks + k_head_w;
        for (int q = 0; q < y.C; ++q) {
            float a = 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) a = fmaf(Wo[q * y.F + fidx[u]], feat[u], a);
            a += __shfl_xor(a, 32);
            if (c.hf == 0) part[(c.w * y.C + q) * 32 + c.cl] = a;
        }
        __syncthreads();
        for (int idx = c.tid; idx < y.C * 32; idx += NT) {
            const int q = idx >> 5, cc = idx & 31;
            float a = logit[idx];
            for (int w = 0; w < H / 8; ++w) a += part[(w * y.C + q) * 32 + cc];
            logit[idx] = a;
        }
    }
    __syncthreads();  // also publishes this phase's y / tape stores to the workgroup
}

#ifndef SNN_TRAIN_OCC
#define SNN_TRAIN_OCC 2
#endif
template <int H, int KKI0, int DIRS, bool TRAIN, bool MASKED, int CELL = 0>
__global__ void __launch_bounds__(H * 8, ((TRAIN || MASKED) ? SNN_TRAIN_OCC : SNN_FWD_WAVES) * 32 / H > 0 ? ((TRAIN || MASKED) ? SNN_TRAIN_OCC : SNN_FWD_WAVES) * 32 / H : 1)
lstm_fwd_kernel(const FwdParams p)
{
    constexpr int NT = H * 8;
    constexpr int KKI1 = DIRS * H / 2;
    constexpr int ROWS = (2 * KKI0 > 2 * KKI1) ? 2 * KKI0 : 2 * KKI1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdParams& y = p;
    // kernel-layout offsets of snn_layout.h, compile-time given the template parameters
    constexpr int NW = H / 8;
    constexpr int BLK0 = NW * (KKI0 + H / 2) * 64 + 2 * NW * 32;  // one (layer 0, dir) block
    constexpr int BLK1 = NW * (KKI1 + H / 2) * 64 + 2 * NW * 32;  // one (layer >= 1, dir) block
    const int k_head_w = DIRS * BLK0 + (y.L - 1) * DIRS * BLK1;
    float* xin = smem;                      // [2][ROWS*32]
    float* hb = xin + 2 * ROWS * 32;        // [2][H*32]
    float* part = hb + 2 * H * 32;          // [NW][C][32]
    float* logit = part + (H / 8) * y.C * 32;  // [C][32]
    float* bl = logit + y.C * 32;               // [NW][2][16] bias image of the running phase
    int* item_slot = reinterpret_cast<int*>(bl + NW * 32);

    ItemCtx c;
    c.tid = threadIdx.x;
    c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.cl = c.lane & 31;
    c.hf = c.lane >> 5;
    c.ybase = p.yscratch + (int64_t)blockIdx.x * p.nbuf * p.y_buf_floats;
    const int64_t n_items = (int64_t)p.n_tiles * p.n_samples;

    for (;;) {
        if (c.tid == 0) *item_slot = (int)atomicAdd(p.counter, 1u);
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(*item_slot);
        if (item >= n_items) break;
        c.tile = item / p.n_samples;
        c.s = item - c.tile * p.n_samples;
        c.T = __builtin_amdgcn_readfirstlane(p.tile_T[c.tile]);
        const int pos = c.tile * SNN_TILE + c.cl;
        c.len = p.len_sorted[pos];
        c.orig = p.order[pos];
        c.wks = p.wk + (int64_t)c.s * p.wk_stride;
        c.tape = TRAIN ? p.tape + (int64_t)c.tile * p.tape_tile : nullptr;
        for (int idx = c.tid; idx < y.C * 32; idx += NT) logit[idx] = 0.0f;
        __syncthreads();  // item_slot consumed, logit zeroed

        for (int l = 0; l < y.L; ++l) {
            for (int d = 0; d < DIRS; ++d) {
                if (l == 0) {
                    const int kA = d * BLK0;
                    run_phase<H, KKI0, DIRS, true, TRAIN, MASKED, CELL>(p, c, l, d, kA, kA + NW * (KKI0 + H / 2) * 64, k_head_w,
                                                                  xin, hb, part, logit, bl);
                } else {
                    const int kA = DIRS * BLK0 + ((l - 1) * DIRS + d) * BLK1;
                    run_phase<H, KKI1, DIRS, false, TRAIN, MASKED, CELL>(p, c, l, d, kA, kA + NW * (KKI1 + H / 2) * 64, k_head_w,
                                                                   xin, hb, part, logit, bl);
                }
            }
        }

        // ---- head bias + softmax, one thread per curve
        if (c.tid < 32) {
            const int orig = c.orig;  // tid < 32: lane == cl
            if (orig >= 0) {
                const float* bo = c.wks + k_head_w + y.C * y.F;
                float mx = -INFINITY;
                for (int q = 0; q < y.C; ++q) mx = fmaxf(mx, logit[q * 32 + c.tid] + bo[q]);
                float z = 0.0f;
                for (int q = 0; q < y.C; ++q) z += expf(logit[q * 32 + c.tid] + bo[q] - mx);
                const int64_t o = (int64_t)c.s * p.os_s + (int64_t)orig * p.os_n;
                for (int q = 0; q < y.C; ++q) {
                    const float lg = logit[q * 32 + c.tid] + bo[q];
                    if (p.out_logits) p.out_logits[o + q] = lg;
                    if (p.out_probs) p.out_probs[o + q] = __fdiv_rn(expf(lg - mx), z);
                }
            }
        }
        __syncthreads();
    }
}


// ---- training forward of ONE layer, a workgroup per (tile, direction) ------------------------------------
// Small batches are a latency chain: 4 tiles of a 128-curve batch occupy 4 workgroups for L * DIRS * T steps in the
// persistent kernel above.  The two directions of a layer are independent, so this variant runs them side by side
// (grid = n_tiles * DIRS, one launch per layer: the launch boundary is the wait for both directions of the layer
// below) and halves the chain.  Same run_phase, same tape; each phase's per-wave head partials go to `head_part`
// ([tile][ld][NW][C][32]) and head_combine_kernel (snn_forward.hip) adds them in the persistent kernel's order, so
// the logits are bit-identical to the one-launch path.
#ifndef SNN_TRAIN_LAYER_OCC
#define SNN_TRAIN_LAYER_OCC 2
#endif
#ifdef SNN_ABL_LAYER_UNMASKED   // timing ablation only: what the mask code of the MASKED instantiation costs an un-masked model
#define SNN_LAYER_MASKED false
#else
#define SNN_LAYER_MASKED true
#endif
template <int H, int KKI0, int DIRS>
__global__ void __launch_bounds__(H * 8, SNN_TRAIN_LAYER_OCC * 32 / H > 0 ? SNN_TRAIN_LAYER_OCC * 32 / H : 1)
lstm_fwd_train_layer_kernel(const FwdParams p, int l, float* __restrict__ head_part)
{
    constexpr int NT = H * 8;
    constexpr int KKI1 = DIRS * H / 2;
    constexpr int ROWS = (2 * KKI0 > 2 * KKI1) ? 2 * KKI0 : 2 * KKI1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdParams& y = p;
    constexpr int NW = H / 8;
    constexpr int BLK0 = NW * (KKI0 + H / 2) * 64 + 2 * NW * 32;
    constexpr int BLK1 = NW * (KKI1 + H / 2) * 64 + 2 * NW * 32;
    const int k_head_w = DIRS * BLK0 + (y.L - 1) * DIRS * BLK1;
    float* xin = smem;
    float* hb = xin + 2 * ROWS * 32;
    float* part = hb + 2 * H * 32;
    float* logit = part + NW * y.C * 32;
    float* bl = logit + y.C * 32;

    ItemCtx c;
    c.tid = threadIdx.x;
    c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.cl = c.lane & 31;
    c.hf = c.lane >> 5;
    c.ybase = nullptr;   // TRAIN: layer outputs travel through the tape
    c.tile = blockIdx.x / DIRS;
    const int d = blockIdx.x - c.tile * DIRS;
    c.s = 0;
    c.T = __builtin_amdgcn_readfirstlane(p.tile_T[c.tile]);
    const int pos = c.tile * SNN_TILE + c.cl;
    c.len = p.len_sorted[pos];
    c.orig = p.order[pos];
    c.wks = p.wk;
    c.tape = p.tape + (int64_t)c.tile * p.tape_tile;
    for (int idx = c.tid; idx < y.C * 32; idx += NT) logit[idx] = 0.0f;
    __syncthreads();
    if (l == 0) {
        const int kA = d * BLK0;
        run_phase<H, KKI0, DIRS, true, true, SNN_LAYER_MASKED>(p, c, l, d, kA, kA + NW * (KKI0 + H / 2) * 64, k_head_w, xin, hb, part, logit, bl);
    } else {
        const int kA = DIRS * BLK0 + ((l - 1) * DIRS + d) * BLK1;
        run_phase<H, KKI1, DIRS, false, true, SNN_LAYER_MASKED>(p, c, l, d, kA, kA + NW * (KKI1 + H / 2) * 64, k_head_w, xin, hb, part, logit, bl);
    }
    if (y.pool == SNN_POOL_STANDARD || l == y.L - 1) {   // this phase feeds the head: `part` holds its per-wave partials
        float* dst = head_part + ((int64_t)c.tile * (y.L * DIRS) + l * DIRS + d) * (NW * y.C * 32);
        for (int idx = c.tid; idx < NW * y.C * 32; idx += NT) dst[idx] = part[idx];
    }
}

// ---- training forward of ONE layer from precomputed input products: the recurrent half ------------------------
// lstm_fwd_train_layer_kernel with the input k-steps taken out of the step (GX): fwd_gx_kernel (snn_fwd_gx.hip) has
// written Gx[t] = b + W_ih x_t for every step of every (tile, direction); a step here is H / 2 recurrent MFMAs + the
// cell.  Chosen for small batches (snn_forward.hip), where the step chain IS the run time: H = 32: 16 MFMAs per step
// instead of 28 (layer 0) / 48 (layer 1).  Same products in the same order: bit-identical logits and tape.
template <int H, int KKI0, int DIRS, int CELL>
__global__ void __launch_bounds__(H * 8, SNN_TRAIN_LAYER_OCC * 32 / H > 0 ? SNN_TRAIN_LAYER_OCC * 32 / H : 1)
lstm_fwd_rec_kernel(const FwdParams p, int l, float* __restrict__ head_part, const float* __restrict__ gx)
{
    constexpr int NT = H * 8;
    constexpr int KKI1 = DIRS * H / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdParams& y = p;
    constexpr int NW = H / 8;
    constexpr int BLK0 = NW * (KKI0 + H / 2) * 64 + 2 * NW * 32;
    constexpr int BLK1 = NW * (KKI1 + H / 2) * 64 + 2 * NW * 32;
    const int k_head_w = DIRS * BLK0 + (y.L - 1) * DIRS * BLK1;
    float* stage = smem;                 // [2][6][H*32] tape rows of the last two steps (run_phase<GX>: its `xin`)
    float* hb = stage + 2 * SNN_TAPE_Q * H * 32;   // [2][H*32]
    float* part = hb + 2 * H * 32;       // [NW][C][32]
    float* logit = part + NW * y.C * 32; // [C][32]

    ItemCtx c;
    c.tid = threadIdx.x;
    c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.cl = c.lane & 31;
    c.hf = c.lane >> 5;
    c.ybase = nullptr;
    c.tile = blockIdx.x / DIRS;
    const int d = blockIdx.x - c.tile * DIRS;
    c.s = 0;
    c.T = __builtin_amdgcn_readfirstlane(p.tile_T[c.tile]);
    const int pos = c.tile * SNN_TILE + c.cl;
    c.len = p.len_sorted[pos];
    c.orig = p.order[pos];
    c.wks = p.wk;
    c.tape = p.tape + (int64_t)c.tile * p.tape_tile;
    c.gx = gx + ((int64_t)c.tile * DIRS + d) * p.max_T * (4 * H * 32);
    for (int idx = c.tid; idx < y.C * 32; idx += NT) logit[idx] = 0.0f;
    __syncthreads();
    if (l == 0) {
        const int kA = d * BLK0;
        run_phase<H, KKI0, DIRS, true, true, SNN_LAYER_MASKED, CELL, true>(p, c, l, d, kA, kA + NW * (KKI0 + H / 2) * 64, k_head_w, stage, hb, part, logit, nullptr);
    } else {
        const int kA = DIRS * BLK0 + ((l - 1) * DIRS + d) * BLK1;
        run_phase<H, KKI1, DIRS, false, true, SNN_LAYER_MASKED, CELL, true>(p, c, l, d, kA, kA + NW * (KKI1 + H / 2) * 64, k_head_w, stage, hb, part, logit, nullptr);
    }
    if (y.pool == SNN_POOL_STANDARD || l == y.L - 1) {
        float* dst = head_part + ((int64_t)c.tile * (y.L * DIRS) + l * DIRS + d) * (NW * y.C * 32);
        for (int idx = c.tid; idx < NW * y.C * 32; idx += NT) dst[idx] = part[idx];
    }
}

// ---- instantiation lists ------------------------------------------------------------------------------
typedef void (*fwd_kernel_t)(const FwdParams);
typedef void (*fwd_layer_kernel_t)(const FwdParams, int, float*);
typedef void (*fwd_rec_kernel_t)(const FwdParams, int, float*, const float*);
struct FwdSmallEntry { int H, KKI0, dirs; fwd_kernel_t fn, fn_masked, fn_train, fn_gru, fn_rnn, fn_gru_train, fn_rnn_train; fwd_layer_kernel_t fn_train_layer; fwd_rec_kernel_t fn_rec[3]; };
template <int H, int K, int Dd>
constexpr fwd_layer_kernel_t train_layer_kernel()
{
    if constexpr (Dd == 2) return lstm_fwd_train_layer_kernel<H, K, Dd>;
    else return nullptr;   // one direction: nothing to run side by side
}
#define SNN_FWD_ENTRY(H, K, Dd)                                                                       \
    {H, K, Dd, lstm_fwd_kernel<H, K, Dd, false, false>, lstm_fwd_kernel<H, K, Dd, false, true>, \
     lstm_fwd_kernel<H, K, Dd, true, true>, lstm_fwd_kernel<H, K, Dd, false, false, 1>,         \
     lstm_fwd_kernel<H, K, Dd, false, false, 2>, lstm_fwd_kernel<H, K, Dd, true, true, 1>,      \
     lstm_fwd_kernel<H, K, Dd, true, true, 2>, train_layer_kernel<H, K, Dd>(),                   \
     {lstm_fwd_rec_kernel<H, K, Dd, 0>, lstm_fwd_rec_kernel<H, K, Dd, 1>, lstm_fwd_rec_kernel<H, K, Dd, 2>}}
// layer-0 k-step buckets of snn_layout.h (D <= 24, 26, 48, 76, 78), both direction counts
#define SNN_FWD_ENTRIES(H)                                                                              \
    SNN_FWD_ENTRY(H, 12, 2), SNN_FWD_ENTRY(H, 13, 2), SNN_FWD_ENTRY(H, 24, 2), SNN_FWD_ENTRY(H, 38, 2), \
    SNN_FWD_ENTRY(H, 39, 2), SNN_FWD_ENTRY(H, 12, 1), SNN_FWD_ENTRY(H, 13, 1), SNN_FWD_ENTRY(H, 24, 1), \
    SNN_FWD_ENTRY(H, 38, 1), SNN_FWD_ENTRY(H, 39, 1)
