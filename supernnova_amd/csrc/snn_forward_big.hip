// snn_forward_big.hip -- persistent LSTM forward for hidden sizes whose gate weights do not fit the
// register file next to each other (H = 128: one (layer, dir) is 512 x (I + 128) floats = 0.4-0.8 MB).
//
// Same work decomposition as lstm_fwd_kernel (item = (32-curve tile, MC sample), 4 waves, the 32x32 MFMA
// tile of row-tile tau holds the 4 gates of hidden units 8tau..8tau+7), but one workgroup per CU with the
// full 512-register budget, each wave owning RT = H/32 row tiles, and the phase split in two passes:
//   1. input pass (no recurrence): Gx[t] = W_ih x_t + b_ih + b_hh for ALL steps, W_ih row tiles held in
//      registers TP tiles at a time, results streamed to a per-workgroup scratch slab in C/D register order;
//   2. recurrent pass: W_hh (RT x H/2 = 256 registers) stays resident, each step starts its RT accumulator
//      chains from Gx[t] and adds W_hh h_{t-1}; activations, pooling and the head as in the small kernel.
// Algorithmic FLOPs are unchanged (every weight is used once per curve-step); the extra traffic is the Gx
// round trip, 4 KB per curve-step-direction (1.2 MFLOP per step at H=128, D=78).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <type_traits>

#include "snn_kernels.h"
#include "snn_host.h"

#ifndef SNN_BIG_AG
#define SNN_BIG_AG 2     // row tiles (of 4 per wave) whose W_hh stays in VGPRs in the recurrent pass; the others are read from AGPRs
#endif
// One (layer, direction) phase.  TP = row tiles per wave handled by one input pass.
// GX: pass 1 was done by fwd_gx_kernel (snn_fwd_gx.hip) for all tiles at once; `gx` is this (tile, direction)'s block.
// MASKED = false is launched for models without dropout masks only: no Philox code in the kernel at all.
template <int H, int KKI, int DIRS, bool FIRST, int TP, bool TRAIN, bool MASKED, bool GX = false>
__device__ __forceinline__ void big_phase(const FwdParams& p, const ItemCtx& c, int l, int d, int k_A, int k_bias,
                                          int k_head_w, float* xin, float* hb, float* part, float* logit,
                                          float* mlds, float* gx)
{
    constexpr int NT = 256, RT = H / 32, KKH = H / 2, KK = KKI + KKH;
    constexpr int ROWS = 2 * KKI, NF4 = ROWS * 8, NPT = (NF4 + NT - 1) / NT, XIN_STRIDE = ROWS * 32;
    constexpr int NWT = H / 8;  // row tiles of the layer
    static_assert(RT % TP == 0, "tiles per pass must divide the tiles per wave");
    const FwdParams& y = p;
    const MaskP& mk = p.mk;
    const bool masked = MASKED && mk.mode != SNN_MASK_NONE;
    const int ld = l * DIRS + d;
    const bool last_layer = l == y.L - 1;
    const int T = c.T;
    const float* Abase = c.wks + k_A;
    // TRAIN: the layer outputs live on the tape (unmasked); the between-layer mask is applied while staging
    const float* in0 = FIRST ? p.xp + p.tile_off[c.tile] : TRAIN ? c.tape : c.ybase + (int64_t)((l - 1) % p.nbuf) * p.y_buf_floats;
    float* yout = (last_layer || TRAIN) ? nullptr : c.ybase + (int64_t)(l % p.nbuf) * p.y_buf_floats + (int64_t)d * p.max_T * H * 32;
    float* tape_ld = TRAIN ? c.tape + (int64_t)ld * p.max_T * tape_step_floats(H) : nullptr;
    const bool mid_step = masked && mk.mid_T > 1;   // per-step between-layer mask (torch.nn.LSTM dropout in train mode)

    // Staging of x_t (or the layer below's y_t) is branch-free: a thread beyond the image duplicates its last float4
    // (same value to the same LDS address) and the step after the last re-stages the last one.  With the loads under
    // `if (f < NF4)` / `if (more)` the compiler put s_waitcnt vmcnt at the joins, so every step of the input pass waited
    // for the Gx stores of the step before it and for its own staging loads (20.5 ms of the 34.4 ms forward of 8192
    // curves x 300 steps, for 57 % of its FLOPs).
    int fi[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) fi[i] = c.tid + i * NT < NF4 ? c.tid + i * NT : NF4 - 1;
    // input-mask registers of the staging threads (layer 0, variational)
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
    auto load_piece = [&](int i, int t) __attribute__((always_inline)) -> f32x4 {
        const int f = fi[i];
        const float* src;
        if (FIRST) src = in0 + (int64_t)t * XIN_STRIDE + f * 4;
        else {
            const int dd = f / (H * 8), ff = f - dd * (H * 8);
            if (TRAIN) src = in0 + ((int64_t)((l - 1) * DIRS + dd) * p.max_T + t) * tape_step_floats(H) + 5 * H * 32 + ff * 4;
            else src = in0 + ((int64_t)dd * p.max_T + t) * (H * 32) + ff * 4;
        }
        return *reinterpret_cast<const f32x4*>(src);
    };
    auto store_step = [&](float* dst, const f32x4 (&pre)[NPT], int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = fi[i];
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
            *reinterpret_cast<f32x4*>(dst + f * 4) = v;
        }
    };

    // ================= pass 1: Gx[t][tile][i][lane] = W_ih x_t + b ===================================
#ifdef SNN_ABL_BIG_NOP1   // timing ablation only (results are wrong): the recurrent pass alone
    if (p.n_curves >= 0) goto pass2;
#endif
    for (int r0 = 0; r0 < (GX ? 0 : RT); r0 += TP) {
        float wx[TP][KKI];
        f32x16 binit[TP];
#pragma unroll
        for (int r = 0; r < TP; ++r) {
            const int tau = c.w * RT + r0 + r;
            const float* A = Abase + (int64_t)tau * KK * 64 + c.lane;
#pragma unroll
            for (int kk = 0; kk < KKI; ++kk) wx[r][kk] = A[kk * 64];
            const float* B = c.wks + k_bias + (tau * 2 + c.hf) * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) binit[r][i] = B[i] + B[NWT * 32 + i];
        }
        {
            f32x4 pre[NPT];
#pragma unroll
            for (int i = 0; i < NPT; ++i) pre[i] = load_piece(i, 0);
            store_step(xin, pre, 0);
        }
        lds_barrier();
        static_assert(NPT <= KKI, "one staging load behind each of the first k-steps");
        for (int t = 0; t < T; ++t) {
            const int cur = t & 1;
            const int tn = t + 1 < T ? t + 1 : t;
            f32x4 pre[NPT];
            f32x16 acc[TP];
#pragma unroll
            for (int r = 0; r < TP; ++r) acc[r] = binit[r];
            const float* xc = xin + cur * XIN_STRIDE + c.lane;
            // B operands are read from LDS BA k-steps ahead of their MFMAs (sched_barrier keeps the order): with one or
            // two MFMAs per k-step the LDS latency is otherwise exposed at every k-step.  The staging loads of the next
            // step are dealt out one per k-step behind the first MFMAs (issued in one burst, the four waves' loads queue
            // at the compute unit's one address unit in front of the MFMAs).
            constexpr int BA = 4;
            float bq[BA];
#pragma unroll
            for (int kk = 0; kk < BA && kk < KKI; ++kk) bq[kk] = xc[kk * 64];
#pragma unroll
            for (int kk = 0; kk < KKI; ++kk) {
                const float b = bq[kk % BA];
                if (kk + BA < KKI) bq[kk % BA] = xc[(kk + BA) * 64];
#pragma unroll
                for (int r = 0; r < TP; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wx[r][kk], b, acc[r], 0, 0, 0);
                if (kk < NPT) pre[kk] = load_piece(kk, tn);
                __builtin_amdgcn_sched_barrier(0);
            }
            // Gx of this step: 16-byte words [tile][gate q][lane][unit 0..3] (4 stores per row tile instead of 16)
#pragma unroll
            for (int r = 0; r < TP; ++r) {
                f32x4* g = reinterpret_cast<f32x4*>(gx + ((int64_t)t * NWT + (c.w * RT + r0 + r)) * 1024) + c.lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q * 64] = f32x4{acc[r][4 * q], acc[r][4 * q + 1], acc[r][4 * q + 2], acc[r][4 * q + 3]};
            }
            store_step(xin + (cur ^ 1) * XIN_STRIDE, pre, tn);
            lds_barrier();
        }
    }
#ifdef SNN_ABL_BIG_NOP1
pass2:
#endif
    __syncthreads();  // Gx visible to the whole workgroup (each wave reads its own tiles, but keep it simple)

    // ================= pass 2: recurrence ==========================================================
    float wh[RT][KKH];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int tau = c.w * RT + r;
        const float* A = Abase + ((int64_t)tau * KK + KKI) * 64 + c.lane;
#pragma unroll
        for (int kk = 0; kk < KKH; ++kk) wh[r][kk] = A[kk * 64];
        if (d == 0 && masked) {  // variational W_hh row mask (forward direction only)
            const int rr = c.lane & 31;
            const float m = mask_whh(mk, c.s, y.L, H, l, (rr >> 3) * H + 8 * tau + (rr & 7));
#pragma unroll
            for (int kk = 0; kk < KKH; ++kk) wh[r][kk] = __fmul_rn(m, wh[r][kk]);
        }
    }
    // locked between-layer mask of this lane's units, parked in LDS: mlds[unit][curve] (1.0 without masks: the y
    // stores below multiply unconditionally, x * 1.0f == x bit for bit)
    if (MASKED && !last_layer && !TRAIN) {
        for (int idx = c.tid; idx < H * 32; idx += NT) {
            const int j = idx >> 5, cc = idx & 31;
            const int o = p.order[c.tile * SNN_TILE + cc];
            mlds[idx] = (masked && o >= 0) ? mask_mid(mk, c.s, y.L, p.n_curves, DIRS * H, l, 0, o, d * H + j) : 1.0f;
        }
    }
    // Gx[t] of row tile r into its accumulator: fwd_gx_kernel's layout [i][lane] (GX) or pass 1's 16-byte words
    auto load_gx = [&](int t, int r, f32x16& a) __attribute__((always_inline)) {
        const float* g = gx + ((int64_t)t * NWT + (c.w * RT + r)) * 1024;
        if (GX) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = g[i * 64 + c.lane];
        } else {
            const f32x4* g4 = reinterpret_cast<const f32x4*>(g) + c.lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = g4[q * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) a[4 * q + e] = v[e];
            }
        }
    };
    float cst[RT][4], hst[RT][4], hsum[RT][4];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            cst[r][u] = hst[r][u] = hsum[r][u] = 0.0f;
            hb[(8 * (c.w * RT + r) + 4 * c.hf + u) * 32 + c.cl] = 0.0f;
        }
    f32x16 acc[RT];
    const int t0 = d == 0 ? 0 : T - 1;
    if constexpr (GX) {
#pragma unroll
        for (int r = 0; r < RT; ++r) load_gx(t0, r, acc[r]);
        lds_barrier();
        // The step body is ONE straight-line block: the "still inside the curve" selects are v_cndmask on a ballot (a C++
        // select let the compiler sink each cell update under its own exec-mask branch: ~40 branches per step), the Gx of
        // the step after the last is a (clamped, unused) reload of the last one, and the only branch left -- last layer or
        // not -- is uniform and sits behind the activations.
        for (int step = 0; step < T; ++step) {
            const int t = d == 0 ? step : T - 1 - step;
            const int tn = d == 0 ? (t + 1 < T ? t + 1 : t) : (t > 0 ? t - 1 : 0);
            const int cur = step & 1;
            const float* hc = hb + cur * (H * 32) + c.lane;
    #pragma unroll
            for (int kk = 0; kk < KKH; ++kk) {
                const float b = hc[kk * 64];
    #pragma unroll
                for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[r][kk], b, acc[r], 0, 0, 0);
            }
            const unsigned long long vmask = __builtin_amdgcn_ballot_w64(t < c.len);
            float outv[RT][4];
    #pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int j0 = 8 * (c.w * RT + r) + 4 * c.hf;
                float* hn = hb + (cur ^ 1) * (H * 32) + j0 * 32 + c.cl;
    #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float ig = snn_sigmoid(acc[r][u]);
                    const float fg = snn_sigmoid(acc[r][4 + u]);
                    const float gg = snn_tanh(acc[r][8 + u]);
                    const float og = snn_sigmoid(acc[r][12 + u]);
                    const float cn = fmaf(fg, cst[r][u], __fmul_rn(ig, gg));
                    const float hv = og * snn_tanh(cn);
                    cst[r][u] = lane_select(vmask, cn, cst[r][u]);
                    hst[r][u] = lane_select(vmask, hv, hst[r][u]);
                    outv[r][u] = lane_select(vmask, hv, 0.0f);
                    hn[u * 32] = hst[r][u];
                    if (TRAIN) {
                        float* tp = tape_ld + (int64_t)t * tape_step_floats(H) + (j0 + u) * 32 + c.cl;
                        tp[0 * H * 32] = ig;
                        tp[1 * H * 32] = fg;
                        tp[2 * H * 32] = gg;
                        tp[3 * H * 32] = og;
                        tp[4 * H * 32] = cn;
                        tp[5 * H * 32] = outv[r][u];
                    }
                }
                // the next step's Gx for this tile goes straight into the now-free accumulator
    #ifndef SNN_ABL_BIG_NOGX   // (timing ablation only, results are wrong: no Gx reload)
                load_gx(tn, r, acc[r]);
    #endif
            }
            if (last_layer) {
    #pragma unroll
                for (int r = 0; r < RT; ++r)
    #pragma unroll
                    for (int u = 0; u < 4; ++u) hsum[r][u] += outv[r][u];
            } else if (!TRAIN) {
    #pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const int j0 = 8 * (c.w * RT + r) + 4 * c.hf;
                    float* yo = yout + (int64_t)t * (H * 32) + j0 * 32 + c.cl;
    #pragma unroll
                    for (int u = 0; u < 4; ++u) yo[u * 32] = MASKED ? __fmul_rn(mlds[(j0 + u) * 32 + c.cl], outv[r][u]) : outv[r][u];
                }
            }
            lds_barrier();
        }
    } else {
    // Large batches (one workgroup per compute unit, nothing else to run under a wave's memory instructions): what a step
    // stores is dealt out over the 64 k-steps of the NEXT step's MFMA block:
    // the layer outputs (inference) or the tape row (training), which the cell update parks in registers / in LDS
    // (`stage`: the x image of the input pass is free by now; rows are wave-private, so no barrier guards them) and which
    // leave as 16-byte words.  Issued inside the cell update, the 4 waves' 96 four-byte tape stores per lane (or 16 y
    // stores + 16 Gx loads) queued at the compute unit's one address unit behind the MFMAs: the recurrent pass ran at
    // 55 % (training) / 70 % (inference) of its MFMA time.
    constexpr bool STG = TRAIN;
    float* stage = xin;                                   // [6][H*32] tape row of the step before (TRAIN)
#ifdef SNN_PROBE_BIG   // clock64 probes, wave 0 of workgroup 0: cycles per step of the recurrent pass (MFMA block | cell | barrier)
    long long pcyc[3] = {0, 0, 0};
#endif
#pragma unroll
    for (int r = 0; r < RT; ++r) load_gx(t0, r, acc[r]);
    lds_barrier();
    const int lane_row = (32 * c.w + (c.lane >> 3)) * 32 + (c.lane & 7) * 4;   // flush: this lane's 16-byte word of piece 0
    auto steps = [&](auto ysf) __attribute__((always_inline)) {
        constexpr bool YS = decltype(ysf)::value;         // inference, not the last layer: layer outputs are stored
        float yprev[RT][4];
        auto one_step = [&](auto flf, int step) __attribute__((always_inline)) {
            constexpr bool FL = decltype(flf)::value;     // a step before this one exists: its stores go out beside these MFMAs
            const int t = d == 0 ? step : T - 1 - step;
            const int tn = d == 0 ? (t + 1 < T ? t + 1 : t) : (t > 0 ? t - 1 : 0);
            const int tprev = d == 0 ? t - 1 : t + 1;
            const int cur = step & 1;
            const float* hc = hb + cur * (H * 32) + c.lane;
#ifdef SNN_PROBE_BIG
            long long pt_ = clock64();
#define PROBE(i) { const long long now_ = clock64(); pcyc[i] += now_ - pt_; pt_ = now_; }
#else
#define PROBE(i)
#endif
            constexpr int BA = 4;
            float bq[BA];
#pragma unroll
            for (int kk = 0; kk < BA; ++kk) bq[kk] = hc[kk * 64];
            f32x4 fv = {0.f, 0.f, 0.f, 0.f};
            float* trow = TRAIN ? tape_ld + (int64_t)tprev * tape_step_floats(H) + lane_row : nullptr;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KKH; ++kk) {
                const float b = bq[kk % BA];
                if (kk + BA < KKH) bq[kk % BA] = hc[(kk + BA) * 64];
                // W_hh is 256 registers per lane: with the accumulators and the cell state beside it, the compiler parks part
                // of it in AGPRs and copies each value to a VGPR in front of its MFMA (~150 v_accvgpr_read + hazard nops per
                // step, on the pipe the MFMAs need: clock64 probes read 20 000 cycles for the 256 MFMAs of a step).  The
                // matrix instruction can read its A operand from an AGPR directly; the compiler does not select that form, so
                // the upper half of the row tiles issues it as inline asm with the weight constrained to an AGPR.
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (!TRAIN && r >= SNN_BIG_AG)      // (training: the tape staging changes the allocation, and the form costs there: 28.2 -> 32.8 ms)
                        asm volatile(SNN_AGPR_MFMA : "+a"(acc[r]) : "a"(wh[r][kk]), "v"(b));
                    else
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[r][kk], b, acc[r], 0, 0, 0);
                }
                if (FL && YS && kk >= 24 && kk < 24 + 4 * RT) {   // layer outputs of the step before
                    const int r = (kk - 24) >> 2, u = (kk - 24) & 3;
                    yout[(int64_t)tprev * (H * 32) + (8 * (c.w * RT + r) + 4 * c.hf + u) * 32 + c.cl] = yprev[r][u];
                }
                if (FL && STG && kk >= 16) {                // tape row of the step before: LDS -> 16-byte store, 24 words per lane
                    const int pc = (kk - 16) >> 1, off = ((pc >> 2) * H + 8 * (pc & 3)) * 32;
                    if (((kk - 16) & 1) == 0) fv = *reinterpret_cast<const f32x4*>(stage + lane_row + off);
                    else *reinterpret_cast<f32x4*>(trow + off) = fv;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // (the compiler does not know the asm statements are matrix instructions: the wait states it would put between
            // a 16-pass MFMA and the first vector read of its result are given here)
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
            PROBE(0)
            const unsigned long long vmask = __builtin_amdgcn_ballot_w64(t < c.len);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int j0 = 8 * (c.w * RT + r) + 4 * c.hf;
                float* hn = hb + (cur ^ 1) * (H * 32) + j0 * 32 + c.cl;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float ig = snn_sigmoid(acc[r][u]);
                    const float fg = snn_sigmoid(acc[r][4 + u]);
                    const float gg = snn_tanh(acc[r][8 + u]);
                    const float og = snn_sigmoid(acc[r][12 + u]);
                    const float cn = fmaf(fg, cst[r][u], __fmul_rn(ig, gg));
                    const float hv = og * snn_tanh(cn);
                    cst[r][u] = lane_select(vmask, cn, cst[r][u]);
                    hst[r][u] = lane_select(vmask, hv, hst[r][u]);
                    const float outv = lane_select(vmask, hv, 0.0f);
                    hn[u * 32] = hst[r][u];
                    if (TRAIN) {
                        float* sg = stage + (j0 + u) * 32 + c.cl;
                        sg[0 * H * 32] = ig;
                        sg[1 * H * 32] = fg;
                        sg[2 * H * 32] = gg;
                        sg[3 * H * 32] = og;
                        sg[4 * H * 32] = cn;
                        sg[5 * H * 32] = outv;
                    }
                    if (!YS && last_layer) hsum[r][u] += outv;
                    if (YS) yprev[r][u] = MASKED ? __fmul_rn(mlds[(j0 + u) * 32 + c.cl], outv) : outv;
                }
                // the next step's Gx for this tile goes straight into the now-free accumulator (a second register set
                // for it, filled beside the MFMAs, spilled: 1 262 against 1 208 ms on the elasticc workload)
#ifndef SNN_ABL_BIG_NOGX   // (timing ablation only, results are wrong: no Gx reload)
                load_gx(tn, r, acc[r]);
#endif
            }
            PROBE(1)
            lds_barrier();
            PROBE(2)
#undef PROBE
        };
        one_step(std::false_type{}, 0);
        for (int step = 1; step < T; ++step) one_step(std::true_type{}, step);
        // the last step's stores
        const int tl = d == 0 ? T - 1 : 0;
        if (YS) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    yout[(int64_t)tl * (H * 32) + (8 * (c.w * RT + r) + 4 * c.hf + u) * 32 + c.cl] = yprev[r][u];
        }
        if (STG) {
            float* trow = tape_ld + (int64_t)tl * tape_step_floats(H) + lane_row;
#pragma unroll
            for (int pc = 0; pc < 24; ++pc) {
                const int off = ((pc >> 2) * H + 8 * (pc & 3)) * 32;
                *reinterpret_cast<f32x4*>(trow + off) = *reinterpret_cast<const f32x4*>(stage + lane_row + off);
            }
        }
    };
    if (!TRAIN && !last_layer) steps(std::true_type{});
    else steps(std::false_type{});
#ifdef SNN_PROBE_BIG
    if (blockIdx.x == 0 && c.tid == 0 && c.s == 0)
        printf("big pass 2 (l %d d %d, %d steps): mfma %lld  cell %lld  barrier %lld cycles per step\n", l, d, T, pcyc[0] / T, pcyc[1] / T, pcyc[2] / T);
#endif
    }

    // ---- head contribution of this phase
    const bool contributes = y.pool == SNN_POOL_STANDARD || last_layer;
    if (contributes) {
        const float* Wo = c.wks + k_head_w;
        for (int q = 0; q < y.C; ++q) {
            float a = 0.0f;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int j0 = 8 * (c.w * RT + r) + 4 * c.hf;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float feat;
                    int fidx;
                    if (y.pool == SNN_POOL_MEAN) {
                        feat = __fdiv_rn(hsum[r][u], (float)(c.len > 0 ? c.len : 1));
                        fidx = d * H + j0 + u;
                    } else {
                        feat = hst[r][u];
                        fidx = (j0 + u) * (y.L * DIRS) + ld;
                    }
                    if (masked && c.orig >= 0) feat = __fmul_rn(mask_out(mk, c.s, p.n_curves, y.F, c.orig, fidx), feat);
                    if (TRAIN && q == 0) c.tape[(int64_t)y.L * DIRS * p.max_T * tape_step_floats(H) + fidx * 32 + c.cl] = feat;
                    a = fmaf(Wo[q * y.F + fidx], feat, a);
                }
            }
            a += __shfl_xor(a, 32);
            if (c.hf == 0) part[(c.w * y.C + q) * 32 + c.cl] = a;
        }
        __syncthreads();
        for (int idx = c.tid; idx < y.C * 32; idx += NT) {
            const int q = idx >> 5, cc = idx & 31;
            float a = logit[idx];
            for (int w = 0; w < 4; ++w) a += part[(w * y.C + q) * 32 + cc];
            logit[idx] = a;
        }
    }
    __syncthreads();
}

template <int H, int KKI0, int DIRS, bool TRAIN, bool MASKED>
__global__ void __launch_bounds__(256, 1) lstm_fwd_big_kernel(const FwdParams p)
{
    constexpr int NT = 256;
    constexpr int KKI1 = DIRS * H / 2;
    constexpr int ROWS = (2 * KKI0 > 2 * KKI1) ? 2 * KKI0 : 2 * KKI1;
    constexpr int NWT = H / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdParams& y = p;
    constexpr int BLK0 = NWT * (KKI0 + H / 2) * 64 + 2 * NWT * 32;
    constexpr int BLK1 = NWT * (KKI1 + H / 2) * 64 + 2 * NWT * 32;
    const int k_head_w = DIRS * BLK0 + (y.L - 1) * DIRS * BLK1;
    // TRAIN: the x image doubles as the staging row of the tape in the recurrent pass (6 x H x 32 floats); no mask image
    constexpr int XIN = TRAIN ? (2 * ROWS * 32 > SNN_TAPE_Q * H * 32 ? 2 * ROWS * 32 : SNN_TAPE_Q * H * 32) : 2 * ROWS * 32;
    float* xin = smem;                     // [2][ROWS*32] (TRAIN: at least [6][H*32])
    float* hb = xin + XIN;                 // [2][H*32]
    float* mlds = hb + 2 * H * 32;         // [H*32] (inference only)
    float* part = mlds + (TRAIN ? 0 : H * 32);   // [4][C][32]
    float* logit = part + 4 * y.C * 32;    // [C][32]
    int* item_slot = reinterpret_cast<int*>(logit + y.C * 32);

    ItemCtx c;
    c.tid = threadIdx.x;
    c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.cl = c.lane & 31;
    c.hf = c.lane >> 5;
    // per-workgroup scratch: [nbuf y buffers][Gx: max_T * NWT * 1024]
    const int64_t gx_floats = (int64_t)p.max_T * NWT * 1024;
    float* wg = p.yscratch + (int64_t)blockIdx.x * (p.nbuf * p.y_buf_floats + gx_floats);
    c.ybase = wg;
    float* gx = wg + p.nbuf * p.y_buf_floats;
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
        __syncthreads();

        for (int l = 0; l < y.L; ++l) {
            for (int d = 0; d < DIRS; ++d) {
                if (l == 0) {
                    const int kA = d * BLK0;
                    big_phase<H, KKI0, DIRS, true, (KKI0 <= 48 ? H / 32 : 2), TRAIN, MASKED>(p, c, l, d, kA, kA + NWT * (KKI0 + H / 2) * 64,
                                                                            k_head_w, xin, hb, part, logit, mlds, gx);
                } else {
                    const int kA = DIRS * BLK0 + ((l - 1) * DIRS + d) * BLK1;
                    big_phase<H, KKI1, DIRS, false, (KKI1 <= 48 ? H / 32 : 2), TRAIN, MASKED>(p, c, l, d, kA, kA + NWT * (KKI1 + H / 2) * 64,
                                                                             k_head_w, xin, hb, part, logit, mlds, gx);
                }
            }
        }
        if (c.tid < 32) {
            const int orig = c.orig;
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

// ---- training forward of ONE layer from precomputed input products: a workgroup per (tile, direction) -----------
// The persistent kernel above walks a tile's L * DIRS phases one after the other and, per phase, the input pass
// before the recurrent one: 128 curves x 300 steps at H = 128 are 4 workgroups x 4 phases x (2 + 1) x 300 dependent
// steps = 28 ms on 4 of 256 compute units.  Here the input pass of a whole layer is ONE launch over the chip
// (fwd_gx_kernel), the two directions of a layer run side by side, and the chain per layer is 300 recurrent steps.
// Same accumulation order (bias + input k-steps, then recurrent k-steps): bit-identical tape and logits.
template <int H, int KKI0, int DIRS>
__global__ void __launch_bounds__(256, 1) lstm_fwd_big_rec_kernel(const FwdParams p, int l, float* __restrict__ head_part,
                                                                 const float* __restrict__ gx)
{
    constexpr int NT = 256;
    constexpr int KKI1 = DIRS * H / 2;
    constexpr int NWT = H / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdParams& y = p;
    constexpr int BLK0 = NWT * (KKI0 + H / 2) * 64 + 2 * NWT * 32;
    constexpr int BLK1 = NWT * (KKI1 + H / 2) * 64 + 2 * NWT * 32;
    const int k_head_w = DIRS * BLK0 + (y.L - 1) * DIRS * BLK1;
    float* hb = smem;                      // [2][H*32]
    float* mlds = hb + 2 * H * 32;         // [H*32] (inference only; unused here)
    float* part = mlds + H * 32;           // [4][C][32]
    float* logit = part + 4 * y.C * 32;    // [C][32]

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
    float* gxb = const_cast<float*>(gx) + ((int64_t)c.tile * DIRS + d) * p.max_T * (4 * H * 32);
    for (int idx = c.tid; idx < y.C * 32; idx += NT) logit[idx] = 0.0f;
    __syncthreads();
    if (l == 0) {
        const int kA = d * BLK0;
        big_phase<H, KKI0, DIRS, true, 1, true, true, true>(p, c, l, d, kA, kA + NWT * (KKI0 + H / 2) * 64, k_head_w, nullptr, hb, part,
                                                      logit, mlds, gxb);
    } else {
        const int kA = DIRS * BLK0 + ((l - 1) * DIRS + d) * BLK1;
        big_phase<H, KKI1, DIRS, false, 1, true, true, true>(p, c, l, d, kA, kA + NWT * (KKI1 + H / 2) * 64, k_head_w, nullptr, hb, part,
                                                       logit, mlds, gxb);
    }
    if (y.pool == SNN_POOL_STANDARD || l == y.L - 1) {   // per-wave head partials of this phase, for head_combine_kernel
        float* dst = head_part + ((int64_t)c.tile * (y.L * DIRS) + l * DIRS + d) * (4 * y.C * 32);
        for (int idx = c.tid; idx < 4 * y.C * 32; idx += NT) dst[idx] = part[idx];
    }
}

// ---- registry hooks used by snn_forward.hip -------------------------------------------------------
typedef void (*fwd_kernel_t)(const FwdParams);
#define BIG_PICK(K, Dd)                                                                                                   \
    if (KKI0 == K && dirs == Dd)                                                                                          \
        return train ? (masked ? lstm_fwd_big_kernel<128, K, Dd, true, true> : lstm_fwd_big_kernel<128, K, Dd, true, false>)  \
                     : (masked ? lstm_fwd_big_kernel<128, K, Dd, false, true> : lstm_fwd_big_kernel<128, K, Dd, false, false>)
fwd_kernel_t snn_big_kernel(int H, int KKI0, int dirs, bool train, bool masked)
{
    if (H != 128) return nullptr;
    BIG_PICK(12, 2); BIG_PICK(13, 2); BIG_PICK(24, 2); BIG_PICK(38, 2); BIG_PICK(39, 2);   // D <= 24, 26, 48, 76, 78
    BIG_PICK(12, 1); BIG_PICK(13, 1); BIG_PICK(24, 1); BIG_PICK(38, 1); BIG_PICK(39, 1);
    return nullptr;
}
typedef void (*fwd_rec_kernel_t)(const FwdParams, int, float*, const float*);
#define BIG_REC_PICK(K, Dd) if (KKI0 == K && dirs == Dd) return lstm_fwd_big_rec_kernel<128, K, Dd>
fwd_rec_kernel_t snn_big_rec_kernel(int H, int KKI0, int dirs)
{
    if (H != 128) return nullptr;
    BIG_REC_PICK(12, 2); BIG_REC_PICK(13, 2); BIG_REC_PICK(24, 2); BIG_REC_PICK(38, 2); BIG_REC_PICK(39, 2);
    BIG_REC_PICK(12, 1); BIG_REC_PICK(13, 1); BIG_REC_PICK(24, 1); BIG_REC_PICK(38, 1); BIG_REC_PICK(39, 1);
    return nullptr;
}
