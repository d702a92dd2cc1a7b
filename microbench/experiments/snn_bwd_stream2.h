// snn_bwd_stream2.h -- streamed LSTM backward for H = 32, producer / consumer formulation (included by snn_train_big.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snn_kernels.h"

// ---- streamed backward, producer / consumer formulation ---------------------------------------------------------
// bwd_stream_kernel (snn_train_big.hip) runs every wave through the same phases (gate-gradient algebra, barrier, matrix
// products, barrier, reduction), so the matrix pipe idles while the algebra runs and the vector ALU idles during the
// products: its MFMA pipe is ~51 % busy.  Only W_hh^T dG_t (16 of the 48 / 96 MFMAs a SIMD issues per step) is on the
// recurrence; dW and the dy of the layer below are not.  Here the 8 waves split by what the recurrence needs:
//   waves 0-3 (R, one per SIMD): the chain.  Wave w owns hidden units 8w .. 8w+7 of all 32 curves: it computes their
//       gate gradients, writes them to LDS as ITS K-chunk of dG (rows {gate g, unit 8w + jj}: the K split of
//       W_hh^T dG is by unit, so a wave multiplies what it has just written -- no workgroup barrier in between),
//       issues its 16 MFMAs and publishes the partial dh; behind them (off the chain) it adds its rows of
//       dW_hh += dG_t h_{t-1}^T when the layer has a layer below (the W waves then carry the dy products).
//   waves 4-7 (W): everything else, ONE STEP BEHIND: dW_ih += dG_t x_t^T (and dW_hh in layer 0), accumulators in
//       registers over the whole stripe; dy_below = W_ih^T dG_t (partials through LDS, reduced and stored one more
//       step later).
// dG, the dh partials and the dy partials are double-buffered in LDS, so ONE barrier per step orders everything, and
// the W waves' long MFMA blocks run under the R waves' algebra of the next step.  Every operand from HBM is fetched one
// step ahead by loads that every path issues (clamped addresses when there is nothing to fetch): the compiler's
// s_waitcnt counts stay exact, so nothing waits for a load issued in the same step.  The stripe of tiles is one
// continuous sequence of steps (the W waves finish a tile while the R waves have started the next).  Same slabs, same
// reduction kernel, same results to summation order as bwd_stream_kernel.
struct StreamCursor { int band, tile, T, step; bool on; };
#define SNN_S2_BANDS 256
#ifndef SNN_S2_PRIO_R
#define SNN_S2_PRIO_R 3
#endif
#ifndef SNN_S2_PRIO_W
#define SNN_S2_PRIO_W 0
#endif

template <int NTX, bool FIRST, int DIRS, bool MASKED, int ROLE>
__device__ __forceinline__ void bwd_stream2_role(const Bwd2Params& p, int l, int G, float* dy_out_base,
                                                 const float* dy_in_base, float* smem)
{
    constexpr int H = 32, SP = 33;
    constexpr int MTX = FIRST ? 1 : DIRS;     // 32-row output tiles of dy (unused when FIRST)
    constexpr bool R_HAS_H = !FIRST;          // who accumulates dW_hh (column tile 0 of [h | x])
    constexpr int NTW = R_HAS_H ? NTX : 1 + NTX, WT0 = R_HAS_H ? 1 : 0;   // the W waves' column tiles: [WT0, WT0 + NTW)
    float* dGl = smem;                        // [2][4H][33], rows in chunk order: 32 w + 8 gate + (unit - 8 w)
    float* red = dGl + 2 * 4 * H * SP;        // [2][4 chunks][H][32]
    float* red2 = red + 2 * 4 * H * 32;       // [2][4 chunks][MTX * 32][32]
    int* origs = reinterpret_cast<int*>(red2 + (FIRST ? 0 : 2 * 4 * MTX * 32 * 32));   // [4 bands in flight][32]
    int* Tband = origs + 4 * 32;              // [SNN_S2_BANDS] steps of this workgroup's tile in each band (0: none)
    const SnnLayout& y = p.lay;
    const int d = blockIdx.x / G, g = blockIdx.x - d * G;
    const int ld = l * DIRS + d;
    const int tid = threadIdx.x & 255, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, cl = lane & 31, hf = lane >> 5;
    const bool last_layer = l == y.L - 1;
    constexpr bool masked = MASKED;
    const int64_t ts = tape_step_floats(H);
    const int64_t dy_dir = (int64_t)p.n_tiles * DIRS * p.max_T * H * 32;   // floats of one producer's dy block

    // ---- the stripe as one sequence of (tile, step) items; interval n: R works on item n, W multiplies item n - 1
    // and stores the dy of item n - 2
    // The stripe's step counts sit in LDS: looking one up from global memory in the middle of the stream would wait for
    // every load in flight (the counter that orders them is shared).
    const int n_bands = (p.n_tiles + G - 1) / G;
    for (int b = threadIdx.x; b < n_bands && b < SNN_S2_BANDS; b += blockDim.x) {
        const int tile = b * G + ((b & 1) ? G - 1 - g : g);
        Tband[b] = tile < p.n_tiles ? p.tile_T[tile] : 0;
    }
    __syncthreads();
    auto band_T = [&](int b, int tile) {
        return __builtin_amdgcn_readfirstlane(b < SNN_S2_BANDS ? Tband[b] : (tile < p.n_tiles ? p.tile_T[tile] : 0));
    };
    auto seek = [&](StreamCursor& c) {
        for (;; ++c.band) {
            if (c.band >= n_bands) { c.on = false; return; }                // (keeps the last item's coordinates: valid addresses)
            const int tile = c.band * G + ((c.band & 1) ? G - 1 - g : g);
            const int T = band_T(c.band, tile);
            if (T > 0) { c.tile = tile; c.T = T; c.step = T - 1; c.on = true; return; }
        }
    };
    auto advance = [&](StreamCursor& c) {
        if (!c.on) return;
        if (c.step > 0) --c.step;
        else { ++c.band; seek(c); }
    };
    int n_items = 0;
    for (int band = 0; band < n_bands; ++band) n_items += band_T(band, band * G + ((band & 1) ? G - 1 - g : g));
    const int n_int = n_items + 2;
    // k-step kk of chunk w holds rows {gate kk >> 2, units 8 w + 2 (kk & 3), + 1}: k-step 16 (kk >> 2) + 4 w + (kk & 3)
    // of the transposed images (whose K runs over the canonical gate rows)
    auto chunk_kstep = [&](int kk) { return 16 * (kk >> 2) + 4 * w + (kk & 3); };
    float* slab = p.slabs + (int64_t)blockIdx.x * y.P;
    // this workgroup's tile 0 exists (G <= n_tiles): the coordinates a cursor starts from
    StreamCursor first{0, 0, 1, 0, false};
    seek(first);

    if (ROLE == 0) {
        // =========================================== R: the recurrence ===========================================
        __builtin_amdgcn_s_setprio(SNN_S2_PRIO_R);
        float wT[16];
        {
            const float* A = p.WT + p.wt_hh[ld] + lane;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) wT[kk] = A[(int64_t)chunk_kstep(kk) * 64];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) asm volatile("" : "+v"(wT[kk]));      // opaque: kept in its register, never re-loaded inside the step loop
        }
        const int j0 = 8 * w + 4 * hf;            // this lane: units j0 .. j0 + 3 of curve cl
        float db[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) db[i] = 0.0f;
        f32x16 acch;                              // dW_hh rows of this chunk (R_HAS_H)
#pragma unroll
        for (int i = 0; i < 16; ++i) acch[i] = 0.0f;
        struct RSet { float q[6][4], dy[DIRS][4]; f32x4 xh[R_HAS_H ? 4 : 1]; };
        RSet setA, setB;
        float dc[4] = {0.f, 0.f, 0.f, 0.f}, dhr[4] = {0.f, 0.f, 0.f, 0.f}, dfm[4] = {0.f, 0.f, 0.f, 0.f};
        int len = 0;
        bool prev_valid = false;
        StreamCursor cur = first;
        unsigned off_q[5][4], off_dy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int q = 0; q < 5; ++q) off_q[q][u] = (unsigned)((q * H + j0 + u) * 32 + cl);
            off_dy[u] = (unsigned)((j0 + u) * 32 + cl);
        }
        const unsigned off_h = (unsigned)(5 * H * 32 + cl * 32 + 16 * hf);
        const bool has_dy = !last_layer;
        const int64_t dy_pstep = has_dy ? dy_dir : 0;
        auto fetch = [&](const StreamCursor& c, RSet& nv) {          // every path runs it: c's coordinates are always valid
            const int T = c.T, step = c.step;
            const int t = d == 0 ? step : T - 1 - step;
            const int tpc = step > 0 ? (d == 0 ? t - 1 : t + 1) : t;      // clamped: the last step of a tile has no predecessor
            const float* tape_ld = p.tape + (int64_t)c.tile * p.tape_tile + (int64_t)ld * p.max_T * ts;
            const float* tq = tape_ld + (int64_t)t * ts;
            const float* tqp = tape_ld + (int64_t)tpc * ts;
            const float* dyb = has_dy ? dy_in_base + (((int64_t)c.tile * DIRS + d) * p.max_T + t) * H * 32 : tq;   // last layer: any valid address
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int q = 0; q < 5; ++q) nv.q[q][u] = tq[off_q[q][u]];
                nv.q[5][u] = tqp[off_q[4][u]];
#pragma unroll
                for (int pd = 0; pd < DIRS; ++pd) nv.dy[pd][u] = (dyb + pd * dy_pstep)[off_dy[u]];
            }
            if (R_HAS_H) {
#pragma unroll
                for (int q = 0; q < 4; ++q) nv.xh[q] = *reinterpret_cast<const f32x4*>(tqp + off_h + 4 * q);
            }
        };
        auto interval = [&](int n, RSet& cv, RSet& nv) {
#ifdef SNN_ABL_S2_NOR      // timing ablation only (results are wrong): the W waves alone
            if (p.n_curves > 0) { __syncthreads(); advance(cur); return; }
#endif
            const bool live = cur.on;             // the two trailing intervals run on the last item's coordinates, results dropped
            const int T = cur.T, step = cur.step;
            const int t = d == 0 ? step : T - 1 - step;
            const int tp = d == 0 ? t - 1 : t + 1;
            StreamCursor nxt = cur;
            advance(nxt);
            if (live && step == T - 1) {          // first step of a tile (uniform)
                len = p.len_sorted[cur.tile * SNN_TILE + cl];
                const float inv_len = 1.0f / (float)(len > 0 ? len : 1);
                const float* dfeat = p.dfeat + (int64_t)cur.tile * y.F * 32;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u;
                    dc[u] = 0.0f;
                    dhr[u] = y.pool == SNN_POOL_STANDARD ? dfeat[(j * (y.L * DIRS) + ld) * 32 + cl] : 0.0f;
                    dfm[u] = (last_layer && y.pool == SNN_POOL_MEAN) ? dfeat[(d * H + j) * 32 + cl] * inv_len : 0.0f;
                }
                if (w == 0 && lane < 32) origs[(cur.band & 3) * 32 + lane] = p.order[cur.tile * SNN_TILE + lane];
            } else {                              // dh from the step before: the four chunks' partials, fixed order
                const float* r = red + ((n - 1) & 1) * 4 * H * 32 + j0 * 32 + cl;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float s = ((r[(0 * H + u) * 32] + r[(1 * H + u) * 32]) + r[(2 * H + u) * 32]) + r[(3 * H + u) * 32];
                    asm volatile("" : "+v"(s));            // (computed on every path: no branch around the LDS reads)
                    dhr[u] = prev_valid ? s : dhr[u];      // beyond the curve the state is frozen: the gradient passes through
                }
            }
#ifdef SNN_ABL_S2_NOLOADS  // timing ablation only (results are wrong): the steps without their global loads
            if (p.n_curves < 0)
#endif
            fetch(nxt, nv);
            __builtin_amdgcn_sched_barrier(0);    // the loads of the next step are issued HERE: a whole step ahead of their use
            const bool valid = live && t < len;
            const bool prev_live = step > 0 && tp < len;
            float* dg = dGl + (n & 1) * 4 * H * SP + (32 * w + 4 * hf) * SP + cl;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ig = cv.q[0][u], fg = cv.q[1][u], gg = cv.q[2][u], og = cv.q[3][u], cn = cv.q[4][u];
                float dyv = cv.dy[0][u];
                if (DIRS == 2) dyv += cv.dy[DIRS - 1][u];
                float dh = dhr[u] + (has_dy ? dyv : 0.0f);
                if (last_layer && y.pool == SNN_POOL_MEAN) dh += dfm[u];
                const float cp = prev_live ? cv.q[5][u] : 0.0f;
                const float tc = snn_tanh(cn);
                const float dct = dc[u] + dh * og * (1.0f - tc * tc);
                const float dao = valid ? dh * tc * og * (1.0f - og) : 0.0f;
                const float dai = valid ? dct * gg * ig * (1.0f - ig) : 0.0f;
                const float daf = valid ? dct * cp * fg * (1.0f - fg) : 0.0f;
                const float dag = valid ? dct * ig * (1.0f - gg * gg) : 0.0f;
                dc[u] = valid ? dct * fg : dc[u];
                dg[(0 * 8 + u) * SP] = dai;
                dg[(1 * 8 + u) * SP] = daf;
                dg[(2 * 8 + u) * SP] = dag;
                dg[(3 * 8 + u) * SP] = dao;
                db[0 * 4 + u] += dai; db[1 * 4 + u] += daf; db[2 * 4 + u] += dag; db[3 * 4 + u] += dao;
            }
            // dh partial of this chunk: the B operand is what this wave has just written (LDS keeps a wave's order)
            {
                const float* b = dGl + (n & 1) * 4 * H * SP + (32 * w + hf) * SP + cl;
                float bv[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) bv[kk] = b[2 * kk * SP];
                f32x16 ax;
#pragma unroll
                for (int i = 0; i < 16; ++i) ax[i] = 0.0f;
                __builtin_amdgcn_sched_barrier(0);    // all 16 operand reads in flight before the first product
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) ax = __builtin_amdgcn_mfma_f32_32x32x2f32(wT[kk], bv[kk], ax, 0, 0, 0);
                float* ro = red + (n & 1) * 4 * H * 32 + w * H * 32 + cl;
#pragma unroll
                for (int i = 0; i < 16; ++i) ro[((i & 3) + 8 * (i >> 2) + 4 * hf) * 32] = ax[i];
            }
            if (R_HAS_H) {                        // dW_hh rows of this chunk: A = dG rows, k-step kk = curves {kk, kk + 16}
                const float* a = dGl + (n & 1) * 4 * H * SP + (32 * w + cl) * SP + 16 * hf;
                float av[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) av[kk] = a[kk];
                if (step == 0) {                  // the tile's last step: h before it is the zero initial state (uniform branch)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) cv.xh[q][e] = 0.0f;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) acch = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], cv.xh[kk >> 2][kk & 3], acch, 0, 0, 0);
            }
            prev_valid = valid;
            __syncthreads();
            advance(cur);
        };
        fetch(cur, setA);
        int n = 0;
        for (; n + 1 < n_int; n += 2) {
            interval(n, setA, setB);
            interval(n + 1, setB, setA);
        }
        if (n < n_int) interval(n, setA, setB);
        // bias gradients: this lane's 16 (gate, unit) sums over the 32 curves
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = db[i];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (cl == 0) {
                const int R = (i >> 2) * H + j0 + (i & 3);
                slab[y.c_bih[ld] + R] = v;
                slab[y.c_bhh[ld] + R] = v;
            }
        }
        if (R_HAS_H) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = (i & 3) + 8 * (i >> 2) + 4 * hf;
                const int R = (r >> 3) * H + 8 * w + (r & 7);
                const float mm = (d == 0 && masked) ? mask_whh(p.mk, 0, y.L, H, l, R) : 1.0f;   // gradient of weight_hh_l0_raw
                slab[y.c_whh[ld] + (int64_t)R * H + cl] = mm * acch[i];
            }
        }
    } else {
        // =========================================== W: dW and dy_below ==========================================
        __builtin_amdgcn_s_setprio(SNN_S2_PRIO_W);
        float aih[FIRST ? 1 : MTX][16];
        if (!FIRST) {
#pragma unroll
            for (int mt = 0; mt < MTX; ++mt) {
                const float* B = p.WT + p.wt_ih[ld] + (int64_t)mt * 2 * H * 64 + lane;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) aih[mt][kk] = B[(int64_t)chunk_kstep(kk) * 64];
            }
#pragma unroll
            for (int mt = 0; mt < MTX; ++mt)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) asm volatile("" : "+v"(aih[mt][kk]));      // opaque: kept in its register, never re-loaded inside the step loop
        }
        f32x16 acc[NTW];
#pragma unroll
        for (int m = 0; m < NTW; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][i] = 0.0f;
        struct WSet { f32x4 xh[NTW][4]; };
        WSet setA, setB;
        StreamCursor cN = first, cM = first, cS = first;
        cM.on = cS.on = false;
        // [h_prev | x_t] rows of this lane's columns, curves 16 hf .. 16 hf + 15 (an MFMA k-step takes curves {kk, kk + 16})
        auto fetch = [&](const StreamCursor& c, WSet& nv) {          // every path runs it: c's coordinates are always valid
            const int T = c.T, step = c.step;
            const int t = d == 0 ? step : T - 1 - step;
            const int tpc = step > 0 ? (d == 0 ? t - 1 : t + 1) : t;      // clamped; that step's h products are zeroed below
            const float* tape = p.tape + (int64_t)c.tile * p.tape_tile;
#pragma unroll
            for (int m = 0; m < NTW; ++m) {
                const int nt = WT0 + m;
                const float* src;
                if (nt == 0) src = tape + ((int64_t)ld * p.max_T + tpc) * ts + 5 * H * 32 + cl * 32 + 16 * hf;
                else {
                    const int k = 32 * (nt - 1) + cl;
                    if (FIRST) src = p.xp + p.tile_off[c.tile] + (int64_t)t * y.Dp * 32 + (k < y.Dp ? k : y.Dp - 1) * 32 + 16 * hf;   // clamped: columns >= Dp give dW columns nobody reads
                    else {
                        const int kc = k < DIRS * H ? k : DIRS * H - 1;
                        const int dd = kc / H, kk = kc - dd * H;
                        src = tape + ((int64_t)((l - 1) * DIRS + dd) * p.max_T + t) * ts + 5 * H * 32 + kk * 32 + 16 * hf;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) nv.xh[m][q] = *reinterpret_cast<const f32x4*>(src + 4 * q);
            }
        };
        auto interval = [&](int n, WSet& cv, WSet& nv) {
#ifdef SNN_ABL_S2_NOLOADS
            if (p.n_curves < 0)
#endif
            fetch(cN, nv);
            __builtin_amdgcn_sched_barrier(0);    // the loads of the next step are issued HERE: a whole step ahead of their use
            // ---- dy of item n - 2: the four chunks' partials, fixed order; mask; store (nothing to store: the stores go
            // to this workgroup's own slab, which the write-out at the end overwrites)
            if (!FIRST) {
                const int t = d == 0 ? cS.step : cS.T - 1 - cS.step;
                const float* r2 = red2 + ((n - 1) & 1) * 4 * MTX * 32 * 32;
                const int c = tid & 31, jq = tid >> 5;
                const int o = masked ? origs[(cS.band & 3) * 32 + c] : 0;
                float* dst = cS.on ? dy_out_base + (int64_t)d * dy_dir + ((int64_t)cS.tile * DIRS * p.max_T + t) * H * 32
                                   : slab + y.c_wih[ld];
                const int64_t dd_stride = cS.on ? (int64_t)p.max_T * H * 32 : 0;
#pragma unroll
                for (int i = 0; i < MTX * 4; ++i) {
                    const int f = jq + 8 * i, mt = f >> 5, r = f & 31;
                    float sum = ((r2[((0 * MTX + mt) * 32 + r) * 32 + c] + r2[((1 * MTX + mt) * 32 + r) * 32 + c]) +
                                 r2[((2 * MTX + mt) * 32 + r) * 32 + c]) + r2[((3 * MTX + mt) * 32 + r) * 32 + c];
                    if (masked && o >= 0) sum = __fmul_rn(mask_mid(p.mk, 0, y.L, p.n_curves, DIRS * H, l - 1, t, o, f), sum);
                    const int dd = f / H, j = f - dd * H;
                    dst[dd * dd_stride + j * 32 + c] = sum;
                }
            }
            // ---- products of item n - 1 (uniform branch; no global memory operation inside)
#ifdef SNN_ABL_S2_NOW      // timing ablation only (results are wrong): the R waves alone
            if (cM.on && p.n_curves < 0) {
#else
            if (cM.on) {
#endif
                const int t = d == 0 ? cM.step : cM.T - 1 - cM.step;
                if (masked) {     // input / between-layer masks on the x columns (the tile's curve ids were published a step ago)
#pragma unroll
                    for (int m = 0; m < NTW; ++m) {
                        const int nt = WT0 + m;
                        if (nt == 0) continue;
                        const int k = 32 * (nt - 1) + cl;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int o = origs[(cM.band & 3) * 32 + 16 * hf + 4 * q + e];
                                if (o >= 0 && k < (FIRST ? y.D : DIRS * H))
                                    cv.xh[m][q][e] = __fmul_rn(FIRST ? mask_in(p.mk, 0, p.n_curves, y.D, o, k)
                                                                     : mask_mid(p.mk, 0, y.L, p.n_curves, DIRS * H, l - 1, t, o, k), cv.xh[m][q][e]);
                            }
                    }
                }
                if (!R_HAS_H && cM.step == 0) {   // the tile's last step: h before it is the zero initial state (uniform branch)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) cv.xh[0][q][e] = 0.0f;
                }
                const float* dgb = dGl + ((n - 1) & 1) * 4 * H * SP;
                if (!FIRST) {
                    const float* b = dgb + (32 * w + hf) * SP + cl;
                    float bv[16];
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) bv[kk] = b[2 * kk * SP];
                    __builtin_amdgcn_sched_barrier(0);    // all 16 operand reads in flight before the first product
                    float* r2 = red2 + (n & 1) * 4 * MTX * 32 * 32 + w * MTX * 32 * 32 + cl;
#pragma unroll
                    for (int mt = 0; mt < MTX; ++mt) {
                        f32x16 ax;
#pragma unroll
                        for (int i = 0; i < 16; ++i) ax[i] = 0.0f;
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) ax = __builtin_amdgcn_mfma_f32_32x32x2f32(aih[mt][kk], bv[kk], ax, 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 16; ++i) r2[(mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hf) * 32] = ax[i];
                    }
                }
                const float* a = dgb + (32 * w + cl) * SP + 16 * hf;
                float av[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) av[kk] = a[kk];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
                    for (int m = 0; m < NTW; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], cv.xh[m][kk >> 2][kk & 3], acc[m], 0, 0, 0);
                }
            }
            __syncthreads();
            cS = cM;
            cM = cN;
            advance(cN);
        };
        int n = 0;
        for (; n + 1 < n_int; n += 2) {
            interval(n, setB, setA);          // interval n multiplies what interval n - 1 fetched (into setB), fetches into setA
            interval(n + 1, setA, setB);
        }
        if (n < n_int) interval(n, setB, setA);
        // ---- this workgroup's dW sums -> its slab: chunk row 8 gate + jj of chunk w is canonical row gate * H + 8 w + jj
        const int I = y.I[l];
#pragma unroll
        for (int m = 0; m < NTW; ++m) {
            const int nt = WT0 + m;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = (i & 3) + 8 * (i >> 2) + 4 * hf;
                const int R = (r >> 3) * H + 8 * w + (r & 7);
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
    }
}

template <int NTX, bool FIRST, int DIRS, bool MASKED>
__global__ void __launch_bounds__(512, 1) bwd_stream2_kernel(const Bwd2Params p, int l, int G, float* dy_out_base,
                                                            const float* dy_in_base)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // wave-uniform role; both roles execute the same number of barriers (one per interval)
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) bwd_stream2_role<NTX, FIRST, DIRS, MASKED, 0>(p, l, G, dy_out_base, dy_in_base, smem);
    else bwd_stream2_role<NTX, FIRST, DIRS, MASKED, 1>(p, l, G, dy_out_base, dy_in_base, smem);
}

