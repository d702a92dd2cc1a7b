// snn_bwd_stream3.h -- streamed LSTM backward for H = 32, pipelined over consecutive steps (included by snn_train_big.hip).
//
// bwd_stream_kernel (snn_train_big.hip) runs a step as  algebra | barrier | products | barrier | reduction : measured
// with clock64 probes, ~2 000 of its 9 400 cycles per step (layer 1) are the algebra / reduction phases and their
// barriers -- LDS round trips and a dependent chain, with the matrix pipe idle.  Of the 48 / 96 MFMAs a SIMD issues per
// step only W_hh^T dG_t (16) is on the recurrence.  This variant keeps the two roles and the register-resident dW
// accumulators and re-orders a step so that the chain's vector work runs INSIDE the products that are off the chain:
//   barrier X   dG_t complete (buffer b)
//   phase P     role 0: dh partials = W_hh^T dG_t (16 MFMAs) -> red;  role 1: dy partials = W_ih^T dG_t -> red2;
//               between the MFMAs: loads of the gate values of step t-2, stores of dy_{t+1}
//   barrier Y   red, red2 complete
//   phase Q     dW += dG_t [h | x]^T (both roles), and dealt out BETWEEN those MFMAs (sched_barrier-pinned stages, as
//               lstm_fwd_pair_kernel does it): the reduction of red -> dh_{t-1}, the gate gradients of step t-1 into dG
//               buffer b^1, the reduction of red2 -> dy_t, and the loads of [h | x] of step t-1.
// fp32 MFMA and the vector ALU are one pipe, so the vector instructions still cost their issue slots; what disappears
// is every wait between them (LDS latency, transcendental latency, one barrier) -- they now pass under MFMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snn_kernels.h"

template <int NTX, bool FIRST, int DIRS, bool MASKED, int ROLE>
__device__ __forceinline__ void bwd_stream3_role(const Bwd2Params& p, int l, int G, float* dy_out_base,
                                                 const float* dy_in_base, float* smem)
{
    constexpr int H = 32, UPT = 2, SP = 33;
    constexpr int MTX = FIRST ? 1 : DIRS;     // output tiles of dy (unused when FIRST)
    constexpr int NT = 1 + NTX;               // column tiles of [h | x]
    constexpr int NT0 = (NT + 1) / 2;         // tiles [0, NT0) belong to role 0, [NT0, NT) to role 1
    constexpr int T0 = ROLE == 0 ? 0 : NT0, MYNT = ROLE == 0 ? NT0 : NT - NT0;
    constexpr bool DO_REC = ROLE == 0, DO_DX = ROLE == 1 && !FIRST;
    float* dGl = smem;                        // [2][4H][33]
    float* red = dGl + 2 * 4 * H * SP;        // [4 SIMDs][H][32]
    float* red2 = red + 4 * H * 32;           // [4 SIMDs][MTX * 32][32]
    int* origs = reinterpret_cast<int*>(red2 + (FIRST ? 0 : 4 * MTX * 32 * 32));   // [32]
    const SnnLayout& y = p.lay;
    const int d = blockIdx.x / G, g = blockIdx.x - d * G;
    const int ld = l * DIRS + d;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6) & 3, lane = tid & 63, cl = lane & 31, hf = lane >> 5;
    const int c = tid & 31, jg = tid >> 5;    // element-wise stage: curve c, hidden units jg and jg + 16
    const bool last_layer = l == y.L - 1;
    constexpr bool masked = MASKED;           // the un-masked instantiation carries no mask code
    const int64_t ts = tape_step_floats(H);
    const int64_t dy_dir = (int64_t)p.n_tiles * DIRS * p.max_T * H * 32;   // floats of one producer's dy block

    float wT[DO_REC ? 16 : 1], aih[DO_DX ? MTX : 1][16];
    if (DO_REC) {
        const float* A = p.WT + p.wt_hh[ld] + (int64_t)(16 * w) * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) wT[kk] = A[kk * 64];
    }
    if (DO_DX) {
#pragma unroll
        for (int mt = 0; mt < MTX; ++mt) {
            const float* B = p.WT + p.wt_ih[ld] + ((int64_t)mt * 2 * H + 16 * w) * 64 + lane;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) aih[mt][kk] = B[kk * 64];
        }
    }
#ifndef SNN_S3_PRIO
#define SNN_S3_PRIO 0
#endif
    if (SNN_S3_PRIO == 1 && ROLE == 1) __builtin_amdgcn_s_setprio(1);
    if (SNN_S3_PRIO == 3 && ROLE == 0) __builtin_amdgcn_s_setprio(1);
    f32x16 acc[MYNT > 0 ? MYNT : 1];
#pragma unroll
    for (int nt = 0; nt < MYNT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;
    float db[4 * UPT];
#pragma unroll
    for (int i = 0; i < 4 * UPT; ++i) db[i] = 0.0f;

#ifdef SNN_PROBE_STREAM     // measurement build (microbench/build_variant.sh probe -DSNN_PROBE_STREAM): cycles per phase of workgroup 0
    long long probe[5] = {0, 0, 0, 0, 0}, pst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long pk0 = clock64();
#endif
    for (int band = 0;; ++band) {
        const int tile = band * G + ((band & 1) ? G - 1 - g : g);
        if (band * G >= p.n_tiles) break;
        if (tile >= p.n_tiles) continue;
        const int T = __builtin_amdgcn_readfirstlane(p.tile_T[tile]);
        if (T < 1) continue;
        const int len = p.len_sorted[tile * SNN_TILE + c];
        const float inv_len = 1.0f / (float)(len > 0 ? len : 1);
        const float* tape = p.tape + (int64_t)tile * p.tape_tile;
        const float* tape_ld = tape + (int64_t)ld * p.max_T * ts;
        const float* dfeat = p.dfeat + (int64_t)tile * y.F * 32;
        const float* xtile = p.xp + p.tile_off[tile];
        const float* dy_in = last_layer ? nullptr : dy_in_base + ((int64_t)tile * DIRS + d) * p.max_T * H * 32;
        __syncthreads();                       // previous tile done with LDS
        if (tid < 32) origs[tid] = p.order[tile * SNN_TILE + tid];
        __syncthreads();

        float dc[UPT], dhr[UPT], dfm[UPT];
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int j = jg + 16 * i;
            dc[i] = 0.0f;
            dhr[i] = y.pool == SNN_POOL_STANDARD ? dfeat[(j * (y.L * DIRS) + ld) * 32 + c] : 0.0f;
            dfm[i] = (last_layer && y.pool == SNN_POOL_MEAN) ? dfeat[(d * H + j) * 32 + c] * inv_len : 0.0f;
        }
        // gate values (q 0-3: i, f, g, o; q 5: c of the step before -- each c_t is read once and carried) and dy of one
        // step; [h | x] operand rows of one step.  Two of each, alternating: the step loop is unrolled by two.
        struct QSet { float q[6][UPT], dy[DIRS][UPT]; };
        struct XSet { f32x4 xh[MYNT > 0 ? MYNT : 1][4]; };
        QSet qA, qB;
        XSet xA, xB;
        const float* dy0 = dy_in ? dy_in : tape_ld;          // last layer: any valid address, value unused
        const bool has_dy = dy_in != nullptr;
        const int64_t dy_tstep = has_dy ? (int64_t)H * 32 : 0, dy_pstep = has_dy ? dy_dir : 0;
        unsigned off_q[5][UPT], off_dy[UPT], off_x[MYNT > 0 ? MYNT : 1];
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int j = jg + 16 * i;
#pragma unroll
            for (int q = 0; q < 5; ++q) off_q[q][i] = (unsigned)((q * H + j) * 32 + c);
            off_dy[i] = (unsigned)(j * 32 + c);
        }
#pragma unroll
        for (int m = 0; m < MYNT; ++m) {
            const int nt = T0 + m;
            if (nt == 0) off_x[m] = (unsigned)(cl * 32 + 16 * hf);
            else {
                const int k = 32 * (nt - 1) + cl;
                if (FIRST) off_x[m] = (unsigned)((k < y.Dp ? k : y.Dp - 1) * 32 + 16 * hf);   // clamped: columns >= Dp give dW columns nobody reads
                else {
                    const int kc = k < DIRS * H ? k : DIRS * H - 1;
                    const int dd = kc / H, kk = kc - dd * H;
                    off_x[m] = (unsigned)((int64_t)dd * p.max_T * ts + kk * 32 + 16 * hf);
                }
            }
        }
        const float* x_base = FIRST ? xtile : tape + (int64_t)((l - 1) * DIRS) * p.max_T * ts + 5 * H * 32;
        const int64_t x_step = FIRST ? (int64_t)y.Dp * 32 : ts;
        struct FetchBase { const float *tq, *tqp, *dyb, *xb; int t; };
        auto fetch_base = [&](int step) {                    // step < 0 (nothing left to fetch): step 0's addresses, values unused
            FetchBase fb;
            if (step < 0) step = 0;
            fb.t = d == 0 ? step : T - 1 - step;
            const int tpc = step > 0 ? (d == 0 ? fb.t - 1 : fb.t + 1) : fb.t;      // clamped: step 0 has no predecessor
            fb.tq = tape_ld + (int64_t)fb.t * ts;
            fb.tqp = tape_ld + (int64_t)tpc * ts;
            fb.dyb = dy0 + (int64_t)fb.t * dy_tstep;
            fb.xb = x_base + (int64_t)fb.t * x_step;
            return fb;
        };
        constexpr int NQ = UPT * 5, NDY = UPT * DIRS, NQL = NQ + NDY, NXH = MYNT * 4, NST = FIRST ? 0 : MTX * 2;
        float pend[MTX * 2];
#pragma unroll
        for (int i = 0; i < MTX * 2; ++i) pend[i] = 0.0f;
        int pend_t = -1;
        auto piece_q = [&](int k, const FetchBase& fb, QSet& nv) {
            if (k < NQ) {
                const int i = k / 5, q = k - 5 * i;
                if (q < 4) nv.q[q][i] = fb.tq[off_q[q][i]];
                else nv.q[5][i] = fb.tqp[off_q[4][i]];
            } else if (k < NQL) {
                const int kk = k - NQ, i = kk / DIRS, pd = kk - DIRS * i;
                nv.dy[pd][i] = (fb.dyb + pd * dy_pstep)[off_dy[i]];
            }
        };
        auto piece_x = [&](int k, const FetchBase& fb, XSet& nv) {
            if (k < NXH) {
                const int m = k >> 2, q = k & 3, nt = T0 + m;
                const float* src = (nt == 0 ? fb.tqp + 5 * H * 32 : fb.xb) + off_x[m];
                f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * q);   // (h_prev of step 0: clamped address, zeroed before use)
                if (masked && nt > 0) {
                    const int kcol = 32 * (nt - 1) + cl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int o = origs[16 * hf + 4 * q + e];
                        if (o >= 0 && kcol < (FIRST ? y.D : DIRS * H))
                            v[e] = __fmul_rn(FIRST ? mask_in(p.mk, 0, p.n_curves, y.D, o, kcol)
                                                   : mask_mid(p.mk, 0, y.L, p.n_curves, DIRS * H, l - 1, fb.t, o, kcol), v[e]);
                    }
                }
                nv.xh[m][q] = v;
            }
        };
        auto piece_s = [&](int k, float* st_base) {         // one store of the dy the step before produced
            if (k < NST) {
                const int f = jg + 16 * k, dd = f / H, j = f - dd * H;
                st_base[(int64_t)dd * p.max_T * H * 32 + j * 32 + c] = pend[k];
            }
        };
        auto dy_slot = [&](int tt) { return dy_out_base + (int64_t)d * dy_dir + ((int64_t)tile * DIRS * p.max_T + tt) * H * 32; };
        float c_carry[UPT];
#pragma unroll
        for (int i = 0; i < UPT; ++i) c_carry[i] = (tape_ld + (int64_t)(d == 0 ? T - 1 : 0) * ts)[off_q[4][i]];   // c of the first processed step
        bool valid_prev = false;
        int t_prev = 0;

        float bv[16];                          // phase P's B operand (dG_t rows), fetched right after dG_t is complete
        // after the dG writes of a step into buffer wb: barrier X, then this wave's dG operand of phase P (two halves)
        auto post_dg = [&](int wb, int part) {
            if (part == 0) __syncthreads();                                   // X: dG complete
            if (DO_REC || DO_DX) {
                const float* bsrc = dGl + wb * 4 * H * SP + (32 * w + hf) * SP + cl;
#pragma unroll
                for (int kk = 8 * part; kk < 8 * part + 8; ++kk) bv[kk] = bsrc[2 * kk * SP];
            }
        };
        // The gate gradients of step `step` (live: it exists) into dG buffer wb; with_red: first the reduction of the
        // step processed before it (dh from red, its dy from red2).  Cut into 16 slices; mf(k) is called after slice k.
        auto algebra = [&](int step, bool live, const QSet& cv, bool with_red, int wb, auto&& mf) {
            const int t = d == 0 ? step : T - 1 - step;
            const int tp = d == 0 ? t - 1 : t + 1;
            const bool valid = live && t < len;
            const bool prev_live = step > 0 && tp < len;
            float r[UPT][4], r2[MTX * 2][4];
#pragma unroll
            for (int i = 0; i < UPT; ++i)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) r[i][s4] = red[(s4 * H + jg + 16 * i) * 32 + c];
            mf(0);
            if (!FIRST) {
#pragma unroll
                for (int i = 0; i < MTX * 2; ++i) {
                    const int f = jg + 16 * i, mt = f >> 5, rr = f & 31;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) r2[i][s4] = red2[((s4 * MTX + mt) * 32 + rr) * 32 + c];
                }
            }
            mf(1);
            // dh of this step (beyond the curve the state is frozen: the gradient passes through)
            const bool take = with_red && valid_prev;
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                const float s = ((r[i][0] + r[i][1]) + r[i][2]) + r[i][3];
                dhr[i] = take ? s : dhr[i];
            }
            mf(2);
            float tc[UPT], dh[UPT];
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                float dyv = cv.dy[0][i];
                if (DIRS == 2) dyv += cv.dy[DIRS - 1][i];
                dh[i] = dhr[i] + (has_dy ? dyv : 0.0f);
                if (last_layer && y.pool == SNN_POOL_MEAN) dh[i] += dfm[i];
            }
            mf(3);
#pragma unroll
            for (int i = 0; i < UPT; ++i) tc[i] = snn_tanh(c_carry[i]);
            mf(4);
            float dct[UPT], dao[UPT];
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                const float og = cv.q[3][i];
                dct[i] = dc[i] + dh[i] * og * (1.0f - tc[i] * tc[i]);
                dao[i] = valid ? dh[i] * tc[i] * og * (1.0f - og) : 0.0f;
            }
            mf(5);
            float dai[UPT], daf[UPT], dag[UPT];
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                const float ig = cv.q[0][i], gg = cv.q[2][i];
                dai[i] = valid ? dct[i] * gg * ig * (1.0f - ig) : 0.0f;
                dag[i] = valid ? dct[i] * ig * (1.0f - gg * gg) : 0.0f;
            }
            mf(6);
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                const float fg = cv.q[1][i];
                const float cp = prev_live ? cv.q[5][i] : 0.0f;
                daf[i] = valid ? dct[i] * cp * fg * (1.0f - fg) : 0.0f;
                dc[i] = valid ? dct[i] * fg : dc[i];
            }
            mf(7);
            float* dg = dGl + wb * 4 * H * SP;
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                const int j = jg + 16 * i;
                dg[(0 * H + j) * SP + c] = dai[i];
                dg[(1 * H + j) * SP + c] = daf[i];
                dg[(2 * H + j) * SP + c] = dag[i];
                dg[(3 * H + j) * SP + c] = dao[i];
            }
            mf(8);
            post_dg(wb, 0);
            mf(9);
            post_dg(wb, 1);
            mf(10);
            // dy of the step before (its stores go out during the next phase P)
            if (!FIRST) {
                const int o = masked ? origs[c] : 0;
#pragma unroll
                for (int i = 0; i < MTX * 2; ++i) {
                    float s = ((r2[i][0] + r2[i][1]) + r2[i][2]) + r2[i][3];
                    if (masked) {
                        const float mm = o >= 0 ? mask_mid(p.mk, 0, y.L, p.n_curves, DIRS * H, l - 1, t_prev, o, jg + 16 * i) : 1.0f;
                        s = __fmul_rn(mm, s);
                    }
                    pend[i] = with_red ? s : pend[i];
                }
                pend_t = with_red ? t_prev : pend_t;
            }
            mf(11);
#pragma unroll
            for (int i = 0; i < UPT; ++i) { db[0 * UPT + i] += dai[i]; db[1 * UPT + i] += daf[i]; }
            mf(12);
#pragma unroll
            for (int i = 0; i < UPT; ++i) { db[2 * UPT + i] += dag[i]; db[3 * UPT + i] += dao[i]; }
            mf(13);
#pragma unroll
            for (int i = 0; i < UPT; ++i) c_carry[i] = cv.q[5][i];           // this step's q 5 is the next step's c
            mf(14);
            mf(15);
            valid_prev = valid;
            t_prev = t;
        };

        // one step: phase P and phase Q above.  cq: the (consumed) gate values of `step`, free for step - 2;
        // nq: those of step - 1; cx: [h | x] of `step`; nx: free for step - 1; b: dG buffer of `step`.
        // Barrier X of the NEXT step sits inside phase Q (post_dg), so phase P starts with its operand in registers.
        auto do_step = [&](const int step, QSet& cq, const QSet& nq, XSet& cx, XSet& nx, const int b) {
            const int t = d == 0 ? step : T - 1 - step;
#ifdef SNN_PROBE_STREAM
            const long long pt1 = clock64();
#endif
            const FetchBase fq = fetch_base(step - 2), fx = fetch_base(step - 1);
            float* st_base = FIRST ? nullptr : dy_slot(pend_t >= 0 ? pend_t : t);
            constexpr int NPP = NQL + NST;                                    // memory pieces of phase P
            const float* dgb = dGl + b * 4 * H * SP;
            float av[16];                                                     // phase Q's A operand (dG_t columns), fetched during phase P
            auto p_slot = [&](int sl) {
#pragma unroll
                for (int k = (sl * NPP) / 16; k < ((sl + 1) * NPP) / 16; ++k) {
                    if (k < NQL) piece_q(k, fq, cq);
                    else piece_s(k - NQL, st_base);
                }
                if (MYNT > 0 && sl >= 8) {
                    const float* a = dgb + (32 * w + cl) * SP + 16 * hf;
                    av[2 * (sl - 8)] = a[2 * (sl - 8)];
                    av[2 * (sl - 8) + 1] = a[2 * (sl - 8) + 1];
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // ---- phase P
            if (DO_REC || DO_DX) {
                f32x16 ax[DO_DX ? MTX : 1];
#pragma unroll
                for (int mt = 0; mt < (DO_DX ? MTX : 1); ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ax[mt][i] = 0.0f;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    if (DO_REC) ax[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wT[kk], bv[kk], ax[0], 0, 0, 0);
                    if (DO_DX) {
#pragma unroll
                        for (int mt = 0; mt < MTX; ++mt) ax[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aih[mt][kk], bv[kk], ax[mt], 0, 0, 0);
                    }
                    p_slot(kk);
                }
#ifdef SNN_PROBE_STREAM
                pst[1] += clock64() - pt1;
#endif
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = (i & 3) + 8 * (i >> 2) + 4 * hf;
                    if (DO_REC) red[(w * H + r) * 32 + cl] = ax[0][i];
                    if (DO_DX) {
#pragma unroll
                        for (int mt = 0; mt < MTX; ++mt) red2[((w * MTX + mt) * 32 + r) * 32 + cl] = ax[mt][i];
                    }
                }
            } else {
#pragma unroll
                for (int sl = 0; sl < 16; ++sl) p_slot(sl);                   // (layer 0, role 1: no product in this phase)
            }
#ifdef SNN_PROBE_STREAM
            const long long pt2 = clock64();
#endif
            __syncthreads();                                                  // Y: red, red2 complete
#ifdef SNN_PROBE_STREAM
            const long long pt3 = clock64();
#endif
            // ---- phase Q
            if (SNN_S3_PRIO == 2 && ROLE == 1) __builtin_amdgcn_s_setprio(1);
            if (ROLE == 0 && step == 0) {       // the tile's last step: h before it is the zero initial state (uniform branch)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) cx.xh[0][q][e] = 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NM = 16 * MYNT;
            auto mf = [&](int k) {
                __builtin_amdgcn_sched_barrier(0);
#ifdef SNN_PROBE_STREAM
                if ((k & 1) == 0) {
                    pst[2 + (k >> 1)] += clock64() - pt3;
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
#pragma unroll
                for (int m = (k * NM) / 16; m < ((k + 1) * NM) / 16; ++m) {
                    const int kk = m / (MYNT > 0 ? MYNT : 1), mi = m - kk * (MYNT > 0 ? MYNT : 1);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], cx.xh[mi][kk >> 2][kk & 3], acc[mi], 0, 0, 0);
                }
#pragma unroll
                for (int q = (k * NXH) / 16; q < ((k + 1) * NXH) / 16; ++q) piece_x(q, fx, nx);
                __builtin_amdgcn_sched_barrier(0);
            };
            algebra(step - 1, step > 0, nq, true, b ^ 1, mf);
            if (SNN_S3_PRIO == 2 && ROLE == 1) __builtin_amdgcn_s_setprio(0);
#ifdef SNN_PROBE_STREAM
            probe[1] += pt2 - pt1; probe[2] += pt3 - pt2; probe[3] += clock64() - pt3; probe[4] += 1;
#endif
        };
        // ---- prologue: operands of the first two steps, gate gradients of the first
        {
            const FetchBase f0 = fetch_base(T - 1), f1 = fetch_base(T - 2);
#pragma unroll
            for (int k = 0; k < NQL; ++k) piece_q(k, f0, qA);
#pragma unroll
            for (int k = 0; k < NXH; ++k) piece_x(k, f0, xA);
#pragma unroll
            for (int k = 0; k < NQL; ++k) piece_q(k, f1, qB);
            algebra(T - 1, true, qA, false, 0, [](int) {});
        }
        int step = T - 1;
        for (; step >= 1; step -= 2) {
            do_step(step, qA, qB, xA, xB, 0);
            do_step(step - 1, qB, qA, xB, xA, 1);
        }
        if (step == 0) do_step(0, qA, qB, xA, xB, 0);
        if (!FIRST && pend_t >= 0) {           // the dy of the tile's last step
            float* st_last = dy_slot(pend_t);
#pragma unroll
            for (int k = 0; k < NST; ++k) piece_s(k, st_last);
        }
    }

#ifdef SNN_PROBE_STREAM
    if (blockIdx.x == 0 && (threadIdx.x & 255) == 0)
        printf("probe3 l=%d role=%d steps=%lld  phase P %lld (products %lld)  barrier Y %lld  phase Q with barrier X %lld (slice starts 0, 2, .. 14: %lld %lld %lld %lld %lld %lld %lld %lld)  per step; loop total %lld per step\n",
               l, ROLE, probe[4], probe[1] / probe[4], pst[1] / probe[4], probe[2] / probe[4], probe[3] / probe[4],
               pst[2] / probe[4], pst[3] / probe[4], pst[4] / probe[4], pst[5] / probe[4], pst[6] / probe[4], pst[7] / probe[4], pst[8] / probe[4], pst[9] / probe[4],
               (clock64() - pk0) / probe[4]);
#endif
    // ---- this workgroup's partial sums -> its slab (canonical layout, only the (layer, direction) slices it owns)
    float* slab = p.slabs + (int64_t)blockIdx.x * y.P;
    const int I = y.I[l];
#pragma unroll
    for (int m = 0; m < MYNT; ++m) {
        const int nt = T0 + m;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int R = 32 * w + (i & 3) + 8 * (i >> 2) + 4 * hf;
            const float v = acc[m][i];
            if (nt == 0) {
                const float mm = (d == 0 && masked) ? mask_whh(p.mk, 0, y.L, H, l, R) : 1.0f;   // gradient of weight_hh_l0_raw
                slab[y.c_whh[ld] + (int64_t)R * H + cl] = mm * v;
            } else {
                const int k = 32 * (nt - 1) + cl;
                if (k < I) slab[y.c_wih[ld] + (int64_t)R * I + k] = v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4 * UPT; ++i) {
        float v = db[i];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (c == 0) {
            const int R = (i / UPT) * H + jg + 16 * (i % UPT);
            slab[y.c_bih[ld] + R] = v;
            slab[y.c_bhh[ld] + R] = v;
        }
    }
}

template <int NTX, bool FIRST, int DIRS, bool MASKED>
__global__ void __launch_bounds__(512, 1) bwd_stream3_kernel(const Bwd2Params p, int l, int G, float* dy_out_base,
                                                            const float* dy_in_base)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // wave-uniform role: both roles execute the same barriers (two per step, two per tile)
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) bwd_stream3_role<NTX, FIRST, DIRS, MASKED, 0>(p, l, G, dy_out_base, dy_in_base, smem);
    else bwd_stream3_role<NTX, FIRST, DIRS, MASKED, 1>(p, l, G, dy_out_base, dy_in_base, smem);
}
