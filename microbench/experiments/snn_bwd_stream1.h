// snn_bwd_stream1.h -- the streamed LSTM backward as it was before the step-pipelined version (snn_bwd_stream3.h)
// replaced it: algebra | barrier | products | barrier | reduction per step, memory pieces between the MFMAs.
// Kept for the record (not built): 2.76 ms of backward at 65 536 curves against 2.55 ms.  It was included by
// snn_train_big.hip where snn_bwd_stream3.h is now.
#pragma once
#include "snn_kernels.h"

template <int NTX, bool FIRST, int DIRS, bool MASKED, int ROLE>
__device__ __forceinline__ void bwd_stream_role(const Bwd2Params& p, int l, int G, float* dy_out_base,
                                                const float* dy_in_base, float* smem)
{
    constexpr int H = 32, UPT = 2, SP = 33;
    constexpr int MTX = FIRST ? 1 : DIRS;     // output tiles of dy (unused when FIRST)
    constexpr int NT = 1 + NTX;               // column tiles of [h | x]
    constexpr int NT0 = (NT + 1) / 2;         // tiles [0, NT0) belong to role 0, [NT0, NT) to role 1
    constexpr int T0 = ROLE == 0 ? 0 : NT0, MYNT = ROLE == 0 ? NT0 : NT - NT0;
    constexpr bool DO_REC = ROLE == 0, DO_DX = ROLE == 1 && !FIRST;
    float* dGl = smem;                        // [4H][33]
    float* red = dGl + 4 * H * SP;            // [4 SIMDs][H][32]
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
    f32x16 acc[MYNT > 0 ? MYNT : 1];
#pragma unroll
    for (int nt = 0; nt < MYNT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;
    float db[4 * UPT];
#pragma unroll
    for (int i = 0; i < 4 * UPT; ++i) db[i] = 0.0f;

#ifdef SNN_PROBE_STREAM
    long long probe[6] = {0, 0, 0, 0, 0, 0};
    const long long pk0 = clock64();
#endif
    for (int band = 0;; ++band) {
        const int tile = band * G + ((band & 1) ? G - 1 - g : g);
        if (band * G >= p.n_tiles) break;
        if (tile >= p.n_tiles) continue;
        const int T = __builtin_amdgcn_readfirstlane(p.tile_T[tile]);
        const int len = p.len_sorted[tile * SNN_TILE + c];
        const float inv_len = 1.0f / (float)(len > 0 ? len : 1);
        const float* tape = p.tape + (int64_t)tile * p.tape_tile;
        const float* tape_ld = tape + (int64_t)ld * p.max_T * ts;
        const float* dfeat = p.dfeat + (int64_t)tile * y.F * 32;
        const float* xtile = p.xp + p.tile_off[tile];
        const float* dy_in = last_layer ? nullptr : dy_in_base + ((int64_t)tile * DIRS + d) * p.max_T * H * 32;
        __syncthreads();                       // previous item done with LDS
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
        // q 5 = c of the step before, fetched WITH the step (taken from the next fetch it would make every step wait for
        // loads it has just issued); dy of both producing directions, added when consumed (an add at fetch time waits too)
        // Two register sets, used alternately (the step loop is unrolled by two): no copies between "next" and "current".
        struct StepVals { float q[6][UPT], dy[DIRS][UPT]; f32x4 xh[MYNT > 0 ? MYNT : 1][4]; };
        StepVals setA, setB;
        const float* dy0 = dy_in ? dy_in : tape_ld;          // last layer: any valid address, value unused
        const bool has_dy = dy_in != nullptr;
        const int64_t dy_tstep = has_dy ? (int64_t)H * 32 : 0, dy_pstep = has_dy ? dy_dir : 0;   // no branch inside fetch
        // Addressing: every load is  uniform base (SGPR pair, advanced per step by scalar adds) + a 32-bit per-thread
        // offset that does not change over the tile  -- no 64-bit vector address arithmetic inside the step loop.
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
        // One step's operands = NLD load instructions per thread; its dy output = NST stores.  They are issued one by one
        // BETWEEN the matrix products of the step before (piece()): a wave pushes the 64 addresses of a memory instruction
        // to the texture addresser in ~8-16 cycles, and with all 8 waves issuing their ~25 at the top of a step that
        // queue was 1 300 cycles per step during which the matrix pipe idled (measured with clock64 probes).
        struct FetchBase { const float *tq, *tqp, *dyb, *xb; int t; };
        auto fetch_base = [&](int step) {
            FetchBase fb;
            fb.t = d == 0 ? step : T - 1 - step;
            const int tpc = step > 0 ? (d == 0 ? fb.t - 1 : fb.t + 1) : fb.t;      // clamped: step 0 has no predecessor
            fb.tq = tape_ld + (int64_t)fb.t * ts;
            fb.tqp = tape_ld + (int64_t)tpc * ts;
            fb.dyb = dy0 + (int64_t)fb.t * dy_tstep;
            fb.xb = x_base + (int64_t)fb.t * x_step;
            return fb;
        };
        constexpr int NQ = UPT * 5, NDY = UPT * DIRS, NXH = MYNT * 4, NLD = NQ + NDY + NXH, NST = FIRST ? 0 : MTX * 2;
        constexpr int NPIECE = NLD + NST;
        float pend[MTX * 2];
#pragma unroll
        for (int i = 0; i < MTX * 2; ++i) pend[i] = 0.0f;
        int pend_t = -1;
        // piece k < NLD: one load of step fb into nv; NLD <= k < NPIECE: one store of the dy the step before produced
        // (st_base: its time slot; before a tile's first dy exists, the slot of the running step, which the same thread
        // overwrites with the real value one step later)
        auto piece = [&](int k, const FetchBase& fb, StepVals& nv, float* st_base) {
            if (k < NQ) {
                const int i = k / 5, q = k - 5 * i;
                if (q < 4) nv.q[q][i] = fb.tq[off_q[q][i]];
                else nv.q[5][i] = fb.tqp[off_q[4][i]];    // c of the step before; the next step takes it over as ITS c (c_carry)
            } else if (k < NQ + NDY) {
                const int kk = k - NQ, i = kk / DIRS, pd = kk - DIRS * i;
                nv.dy[pd][i] = (fb.dyb + pd * dy_pstep)[off_dy[i]];
            } else if (k < NLD) {
                // [h_{prev} | x_t] rows of this lane's columns (this role's tiles), curves 16 hf .. 16 hf + 15
                const int kk = k - NQ - NDY, m = kk >> 2, q = kk & 3, nt = T0 + m;
                const float* src = (nt == 0 ? fb.tqp + 5 * H * 32 : fb.xb) + off_x[m];
                f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * q);   // (h_prev of step 0: clamped address, never multiplied)
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
            } else if (k < NPIECE) {
                const int i = k - NLD, f = jg + 16 * i, dd = f / H, j = f - dd * H;
                st_base[(int64_t)dd * p.max_T * H * 32 + j * 32 + c] = pend[i];
            }
        };
        auto dy_slot = [&](int tt) { return dy_out_base + (int64_t)d * dy_dir + ((int64_t)tile * DIRS * p.max_T + tt) * H * 32; };
        // Columns of [h | x] beyond the input width hold a clamped (valid, finite) column's data: the dW columns they
        // produce are dropped at the write-out, so they need no select.
        // dy of the layer below is stored one step late: gfx9's vmcnt counts loads and stores together, so the wait for
        // the tape values at the top of a step also waits for every store issued before it -- a store issued at the end
        // of the previous step would put its full round trip on the chain.  Issued after that wait instead, it has a
        // whole step to complete.
        float c_carry[UPT];
#pragma unroll
        for (int i = 0; i < UPT; ++i) c_carry[i] = T > 0 ? (tape_ld + (int64_t)(d == 0 ? T - 1 : 0) * ts)[off_q[4][i]] : 0.0f;   // c of the first processed step
        auto do_step = [&](const int step, const StepVals& cv, StepVals& nv) {
#ifdef SNN_PROBE_STREAM     // measurement build (microbench/build_variant.sh probe -DSNN_PROBE_STREAM): cycles per phase of workgroup 0
            const long long pt0 = clock64();
#endif
            const int t = d == 0 ? step : T - 1 - step;
            const int tp = d == 0 ? t - 1 : t + 1;
            const bool valid = t < len;
            const bool prev_live = step > 0 && tp < len;
            const auto& q_ = cv.q;
            const auto& xh = cv.xh;
            float dyv[UPT];
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                dyv[i] = cv.dy[0][i];
                if (DIRS == 2) dyv[i] += cv.dy[DIRS - 1][i];
                if (!has_dy) dyv[i] = 0.0f;
            }
            // ---- gate gradients of this thread's 2 units (curve c)
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                const int j = jg + 16 * i;
                float dai = 0.f, daf = 0.f, dag = 0.f, dao = 0.f;
                if (valid) {
                    const float ig = q_[0][i], fg = q_[1][i], gg = q_[2][i], og = q_[3][i], cn = c_carry[i];
                    float dh = dhr[i] + dyv[i];
                    if (last_layer && y.pool == SNN_POOL_MEAN) dh += dfm[i];
                    const float cp = prev_live ? q_[5][i] : 0.0f;
                    const float tc = snn_tanh(cn);
                    const float dct = dc[i] + dh * og * (1.0f - tc * tc);
                    dao = dh * tc * og * (1.0f - og);
                    dai = dct * gg * ig * (1.0f - ig);
                    daf = dct * cp * fg * (1.0f - fg);
                    dag = dct * ig * (1.0f - gg * gg);
                    dc[i] = dct * fg;
                }
                dGl[(0 * H + j) * SP + c] = dai;
                dGl[(1 * H + j) * SP + c] = daf;
                dGl[(2 * H + j) * SP + c] = dag;
                dGl[(3 * H + j) * SP + c] = dao;
                db[0 * UPT + i] += dai; db[1 * UPT + i] += daf; db[2 * UPT + i] += dag; db[3 * UPT + i] += dao;
                c_carry[i] = q_[5][i];     // every c_t is read from the tape once: as the predecessor's c here, as its own c next step
            }
#ifdef SNN_PROBE_STREAM
            const long long pt1 = clock64();
#endif
            __syncthreads();
#ifdef SNN_PROBE_STREAM
            const long long pt2 = clock64();
#endif
            // ---- the matrix products of this step, with the loads of the next step (clamped at the tile's last step: the
            // values are then unused) and the stores of the previous step's dy dealt out between them
            const FetchBase fb = fetch_base(step > 0 ? step - 1 : 0);
            float* st_base = FIRST ? nullptr : dy_slot(pend_t >= 0 ? pend_t : t);
            constexpr int NSLOT = ((DO_REC || DO_DX) ? 16 : 0) + (MYNT > 0 ? 16 : 0);
            auto slot = [&](int sl) {      // the pieces of slot sl, pinned behind the products issued so far
#pragma unroll
                for (int k = (sl * NPIECE) / NSLOT; k < ((sl + 1) * NPIECE) / NSLOT; ++k) piece(k, fb, nv, st_base);
                __builtin_amdgcn_sched_barrier(0);
            };
            // ---- dh_{prev} (role 0) / dy of the layer below (role 1): K = this SIMD's 32 gate rows
            if (DO_REC || DO_DX) {
                f32x16 ax[DO_DX ? MTX : 1];
#pragma unroll
                for (int mt = 0; mt < (DO_DX ? MTX : 1); ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ax[mt][i] = 0.0f;
                const float* b = dGl + (32 * w + hf) * SP + cl;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const float bv = b[2 * kk * SP];
                    if (DO_REC) ax[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wT[kk], bv, ax[0], 0, 0, 0);
                    if (DO_DX) {
#pragma unroll
                        for (int mt = 0; mt < MTX; ++mt) ax[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aih[mt][kk], bv, ax[mt], 0, 0, 0);
                    }
                    slot(kk);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = (i & 3) + 8 * (i >> 2) + 4 * hf;
                    if (DO_REC) red[(w * H + r) * 32 + cl] = ax[0][i];
                    if (DO_DX) {
#pragma unroll
                        for (int mt = 0; mt < MTX; ++mt) red2[((w * MTX + mt) * 32 + r) * 32 + cl] = ax[mt][i];
                    }
                }
            }
            // ---- dW += dG_t [h_prev | x_t]^T : A = this SIMD's 32 gate rows, k-step kk = curves {kk, kk + 16}
            if (MYNT > 0) {
                const float* a = dGl + (32 * w + cl) * SP + 16 * hf;
                // h_prev of the first processed step is 0: that step (once per tile, uniform) skips the products of tile 0
                auto dw = [&](auto skip0) {
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) {
                        const float av = a[kk];
#pragma unroll
                        for (int m = 0; m < MYNT; ++m)
                            if (!(decltype(skip0)::value && T0 + m == 0))
                                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xh[m][kk >> 2][kk & 3], acc[m], 0, 0, 0);
                        slot(((DO_REC || DO_DX) ? 16 : 0) + kk);
                    }
                };
                if (ROLE == 0 && step == 0) dw(std::true_type{});
                else dw(std::false_type{});
            }
#ifdef SNN_PROBE_STREAM
            const long long pt3 = clock64();
#endif
            __syncthreads();
#ifdef SNN_PROBE_STREAM
            const long long pt4 = clock64();
#endif
            if (valid) {   // beyond the curve the state is frozen: the gradient passes through unchanged
#pragma unroll
                for (int i = 0; i < UPT; ++i) {
                    const int j = jg + 16 * i;
                    dhr[i] = ((red[(0 * H + j) * 32 + c] + red[(1 * H + j) * 32 + c]) + red[(2 * H + j) * 32 + c]) + red[(3 * H + j) * 32 + c];
                }
            }
            if (!FIRST) {
                const int o = origs[c];
#pragma unroll
                for (int i = 0; i < MTX * 2; ++i) {
                    const int f = jg + 16 * i, mt = f >> 5, r = f & 31;
                    float s = ((red2[((0 * MTX + mt) * 32 + r) * 32 + c] + red2[((1 * MTX + mt) * 32 + r) * 32 + c]) +
                               red2[((2 * MTX + mt) * 32 + r) * 32 + c]) + red2[((3 * MTX + mt) * 32 + r) * 32 + c];
                    if (masked && o >= 0) s = __fmul_rn(mask_mid(p.mk, 0, y.L, p.n_curves, DIRS * H, l - 1, t, o, f), s);
                    pend[i] = s;
                }
                pend_t = t;
            }
#ifdef SNN_PROBE_STREAM
            const long long pt5 = clock64();
            probe[0] += pt1 - pt0; probe[1] += pt2 - pt1; probe[2] += pt3 - pt2; probe[3] += pt4 - pt3; probe[4] += pt5 - pt4; probe[5] += 1;
#endif
        };
        if (T > 0) {
            const FetchBase fb0 = fetch_base(T - 1);
#pragma unroll
            for (int k = 0; k < NLD; ++k) piece(k, fb0, setA, nullptr);
        }
        int step = T - 1;
        for (; step >= 1; step -= 2) {
            do_step(step, setA, setB);
            do_step(step - 1, setB, setA);
        }
        if (step == 0) do_step(0, setA, setB);
        if (!FIRST && pend_t >= 0) {           // the dy of the tile's last step
            float* st_last = dy_slot(pend_t);
#pragma unroll
            for (int k = NLD; k < NPIECE; ++k) piece(k, FetchBase{}, setA, st_last);
        }
    }

#ifdef SNN_PROBE_STREAM
    if (blockIdx.x == 0 && (threadIdx.x & 255) == 0)
        printf("probe l=%d role=%d steps=%lld  algebra %lld  barrier1 %lld  products %lld  barrier2 %lld  reduce %lld  per step; loop total %lld per step\n",
               l, ROLE, probe[5], probe[0] / probe[5], probe[1] / probe[5], probe[2] / probe[5], probe[3] / probe[5], probe[4] / probe[5],
               (clock64() - pk0) / probe[5]);
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
__global__ void __launch_bounds__(512, 1) bwd_stream_kernel(const Bwd2Params p, int l, int G, float* dy_out_base,
                                                           const float* dy_in_base)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // wave-uniform role: both roles execute the same barriers (two per step, two per tile)
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) bwd_stream_role<NTX, FIRST, DIRS, MASKED, 0>(p, l, G, dy_out_base, dy_in_base, smem);
    else bwd_stream_role<NTX, FIRST, DIRS, MASKED, 1>(p, l, G, dy_out_base, dy_in_base, smem);
}

