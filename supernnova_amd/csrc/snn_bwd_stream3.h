// snn_bwd_stream3.h -- streamed LSTM backward for H = 32, pipelined over consecutive steps and consecutive tiles
// (included by snn_train_big.hip, which describes the formulation: who owns what, where dG, dW and db live).
//
// Its predecessor (microbench/experiments/snn_bwd_stream1.h) ran a step as  algebra | barrier | products | barrier |
// reduction : measured with clock64 probes, ~2 000 of its 9 400 cycles per step (layer 1) were the algebra / reduction
// phases and their barriers -- LDS round trips and a dependent chain, with the matrix pipe idle -- and a tile boundary
// cost ~20 000 cycles (three dependent trips to HBM).  Of the 48 / 96 MFMAs a SIMD issues per step only W_hh^T dG_t (16)
// is on the recurrence.  This kernel keeps the two roles and the register-resident dW accumulators and orders a step as
//   barrier X   dG_t complete (buffer b); sits INSIDE phase Q of the step before, followed by the fetch of the
//               dG operand of phase P
//   phase P     role 0: dh partials = W_hh^T dG_t (16 MFMAs) -> red;  role 1: dy partials = W_ih^T dG_t -> red2;
//               before the MFMAs: everything of step t-1's gate gradients that does not depend on dh (precompute);
//               between the MFMAs: loads of the gate values of step t-2, stores of dy_{t+1}, the dG operand of phase Q
//   barrier Y   red, red2 complete
//   phase Q     dW += dG_t [h | x]^T (both roles), and dealt out BETWEEN those MFMAs in 16 slices pinned with
//               sched_barrier: the reduction of red -> dh_{t-1}, the gate gradients of step t-1 into dG buffer b^1,
//               barrier X, the reduction of red2 -> dy_t, and the loads of [h | x] of step t-1.
// Across a tile's end the same fetches address the first two steps of the stripe's next tile.
// What this does not buy (microbench/mfma_two_waves.hip): a wave does not overlap its own vector instructions with its
// own fp32 MFMA, so phase Q still costs its MFMAs plus its chain; DESIGN.md lists what was tried against that.
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
    int* origs_lds = reinterpret_cast<int*>(red2 + (FIRST ? 0 : 4 * MTX * 32 * 32));   // [2][32]: this tile's curves, the next tile's
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
    // (s_setprio 1 for either role, or for role 1 during phase Q only: no measurable difference)
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
    // gate values (q 0-3: i, f, g, o; q 5: c of the step before -- each c_t is read once and carried) and dy of one
    // step; [h | x] operand rows of one step.  Two of each, alternating: the step loop is unrolled by two.
    struct QSet { float q[6][UPT], dy[DIRS][UPT]; };
    struct XSet { f32x4 xh[MYNT > 0 ? MYNT : 1][4]; };
    QSet qA, qB;
    XSet xA, xB;
    const bool has_dy = !last_layer;
    const int64_t dy_tstep = has_dy ? (int64_t)H * 32 : 0, dy_pstep = has_dy ? dy_dir : 0;
    // element (hidden unit jg + 16 i, curve c) of a [H][32] block: one per-lane offset, the rest goes into the (scalar) base
    // Per-lane offsets are kept in BYTES in 32 bits and added to a wave-uniform base: the load then takes the form
    // global_load v, v_off, s[base:base+1] with no vector arithmetic (an offset in floats has to be widened and
    // shifted per load -- 2 - 3 vector instructions each, on a pipe the fp32 MFMAs share).
    const unsigned off0 = (unsigned)(jg * 32 + c) * 4u;
    // (The empty asm keeps the widening of the offset next to its use -- hoisted out of the loop it lands in another basic
    //  block, where instruction selection no longer sees "uniform base + zero-extended 32-bit offset" -- and the base whole:
    //  the compiler otherwise moves a constant part of it behind the vector add.)
    typedef const __attribute__((address_space(1))) char* gcptr;     // (behind the asm the compiler no longer knows the pointer is global)
    typedef __attribute__((address_space(1))) char* gptr;
    auto ldf = [](const float* sb, unsigned boff) {
#ifdef SNN_S3_PLAIN_ADDR
        return sb[boff >> 2];
#endif
        asm volatile("" : "+s"(sb), "+v"(boff));
        return *reinterpret_cast<const __attribute__((address_space(1))) float*>((gcptr)sb + boff);
    };
    auto ld4 = [](const float* sb, unsigned boff) {
#ifdef SNN_S3_PLAIN_ADDR
        return *reinterpret_cast<const f32x4*>(sb + (boff >> 2));
#endif
        asm volatile("" : "+s"(sb), "+v"(boff));
        return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>((gcptr)sb + boff);
    };
    auto stf = [](float* sb, unsigned boff, float v) {
#ifdef SNN_S3_PLAIN_ADDR
        sb[boff >> 2] = v;
        return;
#endif
        asm volatile("" : "+s"(sb), "+v"(boff));
        *reinterpret_cast<__attribute__((address_space(1))) float*>((gptr)sb + boff) = v;
    };
    unsigned off_x[MYNT > 0 ? MYNT : 1];
#pragma unroll
    for (int m = 0; m < MYNT; ++m) {
        const int nt = T0 + m;
        if (nt == 0) off_x[m] = (unsigned)(cl * 32 + 16 * hf) * 4u;
        else {
            const int k = 32 * (nt - 1) + cl;
            if (FIRST) off_x[m] = (unsigned)((k < y.Dp ? k : y.Dp - 1) * 32 + 16 * hf) * 4u;   // clamped: columns >= Dp give dW columns nobody reads
            else {
                const int kc = k < DIRS * H ? k : DIRS * H - 1;
                const int dd = kc / H, kk = kc - dd * H;
                off_x[m] = (unsigned)((int64_t)dd * p.max_T * ts + kk * 32 + 16 * hf) * 4u;
            }
        }
    }
    const int64_t x_step = FIRST ? (int64_t)y.Dp * 32 : ts;
    constexpr int NQ = UPT * 5, NDY = UPT * DIRS, NQL = NQ + NDY, NXH = MYNT * 4, NST = FIRST ? 0 : MTX * 2;

    // One tile's addresses.  The operands of a tile's first two steps are fetched during the last two steps of the
    // tile before it (the fetches that would otherwise run past the tile's end), and its per-curve values at that
    // tile's start: a tile boundary then costs the fill of the step pipeline, not three dependent trips to HBM.
    struct TileCtx { int tile, T; const float *tape_ld, *x_base, *dy0; };
    const int NB = (p.n_tiles + G - 1) / G;
    auto tile_of = [&](int band) {
        if (band >= NB) return -1;
        const int tl = band * G + ((band & 1) ? G - 1 - g : g);
        return tl < p.n_tiles ? tl : -1;
    };
    auto make_ctx = [&](int tile, int T, int64_t x_off) {
        TileCtx cx;
        cx.tile = tile; cx.T = T;
        const float* tape = p.tape + (int64_t)tile * p.tape_tile;
        cx.tape_ld = tape + (int64_t)ld * p.max_T * ts;
        cx.x_base = FIRST ? p.xp + x_off : tape + (int64_t)((l - 1) * DIRS) * p.max_T * ts + 5 * H * 32;
        cx.dy0 = has_dy ? dy_in_base + ((int64_t)tile * DIRS + d) * p.max_T * H * 32 : cx.tape_ld;   // last layer: any valid address, value unused
        return cx;
    };
    auto uniform64 = [](int64_t v) {
        const int lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
        return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    };
    int tile0 = tile_of(0);
    TileCtx cur = make_ctx(0, 1, 0);
    if (tile0 >= 0) cur = make_ctx(tile0, __builtin_amdgcn_readfirstlane(p.tile_T[tile0]), FIRST ? uniform64(p.tile_off[tile0]) : 0);
    // values of the NEXT tile, fetched with vector loads (in-order vmcnt; a scalar load would be waited for at the next LDS wait)
    int nT_v = 0, n_len = 0, n_order = 0;
    int64_t n_xoff_v = 0;
    float n_dhr[UPT], n_cc[UPT];
    int T_n2 = 0;                              // T and x offset of the tile after the current one, as scalars
    int64_t xoff_n2 = 0;
    {
        const int t1 = tile_of(1);
        nT_v = t1 >= 0 ? p.tile_T[t1] : 0;
        n_xoff_v = (FIRST && t1 >= 0) ? p.tile_off[t1] : 0;
    }
    // dy of the step before, stored during the next step's phase P -- across a tile boundary too: the last step's dy of a
    // tile goes out in the first step of the next (a store at a tile's end would sit in front of the waits of its start)
    float pend[MTX * 2];
#pragma unroll
    for (int i = 0; i < MTX * 2; ++i) pend[i] = 0.0f;
    float* pend_slot = nullptr;
    bool pref = false;                         // this tile's first operands and per-curve values are already in registers
    for (int band = 0; tile0 >= 0 && band < NB; ++band) {
        const int tile = cur.tile, T = cur.T;
        if (tile_of(band) < 0) break;
#ifdef SNN_PROBE_STREAM
        const long long ph0 = clock64();
#endif
        // ---- the tile after this one
        const int tile_n = tile_of(band + 1);
        const int T_n = __builtin_amdgcn_readfirstlane(nT_v);
        xoff_n2 = FIRST ? uniform64(n_xoff_v) : 0;
        const bool has_next = tile_n >= 0 && T_n >= 1;
        const TileCtx nxt_real = make_ctx(tile_n >= 0 ? tile_n : tile, T_n, xoff_n2);
        const TileCtx nxt = has_next ? nxt_real : cur;   // for the fetches; no next tile: valid addresses, values unused
        {
            const int t2 = tile_of(band + 2);
            nT_v = t2 >= 0 ? p.tile_T[t2] : 0;
            n_xoff_v = (FIRST && t2 >= 0) ? p.tile_off[t2] : 0;
        }
        if (T < 1) {
            cur = nxt_real; pref = false;
            if (tile_n < 0) break;
            continue;
        }
        int* origs_c = origs_lds + 32 * (band & 1);
        int* origs_n = origs_lds + 32 * ((band & 1) ^ 1);
        int len, order_c;
        float dc[UPT], dhr[UPT], dfm[UPT], c_carry[UPT];
        const float* dfeat = p.dfeat + (int64_t)tile * y.F * 32;
        if (pref) {
            len = n_len; order_c = n_order;
#pragma unroll
            for (int i = 0; i < UPT; ++i) { dhr[i] = n_dhr[i]; c_carry[i] = n_cc[i]; }
        } else {
            len = p.len_sorted[tile * SNN_TILE + c];
            order_c = p.order[tile * SNN_TILE + c];
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                dhr[i] = y.pool == SNN_POOL_STANDARD ? dfeat[((jg + 16 * i) * (y.L * DIRS) + ld) * 32 + c] : 0.0f;
                c_carry[i] = ldf(cur.tape_ld + (int64_t)(d == 0 ? T - 1 : 0) * ts + (4 * H + 16 * i) * 32, off0);   // c of the first processed step
            }
        }
        const float inv_len = 1.0f / (float)(len > 0 ? len : 1);
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            dc[i] = 0.0f;
            dfm[i] = (last_layer && y.pool == SNN_POOL_MEAN) ? dfeat[(d * H + jg + 16 * i) * 32 + c] * inv_len : 0.0f;
        }
        {   // the next tile's per-curve values (no next tile: this tile's, unused)
            const float* dfeat_n = p.dfeat + (int64_t)nxt.tile * y.F * 32;
            n_len = p.len_sorted[nxt.tile * SNN_TILE + c];
            if (masked) n_order = p.order[nxt.tile * SNN_TILE + c];
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                n_dhr[i] = y.pool == SNN_POOL_STANDARD ? dfeat_n[((jg + 16 * i) * (y.L * DIRS) + ld) * 32 + c] : 0.0f;
                n_cc[i] = ldf(nxt.tape_ld + (int64_t)(d == 0 ? nxt.T - 1 : 0) * ts + (4 * H + 16 * i) * 32, off0);
            }
        }
#ifdef SNN_PROBE_STREAM
        const long long ph1 = clock64();
#endif
        __syncthreads();                       // previous tile done with LDS
#ifdef SNN_PROBE_STREAM
        const long long ph2 = clock64();
#endif
        if (!pref) {
            if (tid < 32) origs_c[tid] = order_c;
            __syncthreads();
        }

        struct FetchBase { const float *tq, *tqp, *dyb, *xb; const int* og; int t; };
        auto fetch_base = [&](int step) {      // step < 0: the first steps of the next tile (no next tile: a valid address of this one, values unused)
            FetchBase fb;
            const bool nx = step < 0;
            const TileCtx& cx = nx ? nxt : cur;
            if (nx) { step += cx.T; if (step < 0) step = 0; }
            fb.t = d == 0 ? step : cx.T - 1 - step;
            const int tpc = step > 0 ? (d == 0 ? fb.t - 1 : fb.t + 1) : fb.t;      // clamped: step 0 has no predecessor
            fb.tq = cx.tape_ld + (int64_t)fb.t * ts;
            fb.tqp = cx.tape_ld + (int64_t)tpc * ts;
            fb.dyb = cx.dy0 + (int64_t)fb.t * dy_tstep;
            fb.xb = cx.x_base + (int64_t)fb.t * x_step;
            fb.og = nx ? origs_n : origs_c;
#ifdef SNN_ABL_S3_L2          // timing ablation: every fetch from the same few KB (cache hits), results wrong
            fb.tq = fb.tqp = p.tape + (int64_t)ld * p.max_T * ts; fb.dyb = dy_in_base ? dy_in_base : p.tape; fb.xb = FIRST ? p.xp : p.tape;
#endif
            return fb;
        };
        auto piece_q = [&](int k, const FetchBase& fb, QSet& nv) {
            if (k < NQ) {
                const int i = k / 5, q = k - 5 * i;
                if (q < 4) nv.q[q][i] = ldf(fb.tq + (q * H + 16 * i) * 32, off0);
                else nv.q[5][i] = ldf(fb.tqp + (4 * H + 16 * i) * 32, off0);
            } else if (k < NQL) {
                const int kk = k - NQ, i = kk / DIRS, pd = kk - DIRS * i;
                nv.dy[pd][i] = ldf(fb.dyb + pd * dy_pstep + 16 * i * 32, off0);
            }
        };
        auto piece_x = [&](int k, const FetchBase& fb, XSet& nv) {
            if (k < NXH) {
                const int m = k >> 2, q = k & 3, nt = T0 + m;
                f32x4 v = ld4((nt == 0 ? fb.tqp + 5 * H * 32 : fb.xb) + 4 * q, off_x[m]);   // (h_prev of step 0: clamped address, zeroed before use)
                if (masked && nt > 0) {
                    const int kcol = 32 * (nt - 1) + cl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int o = fb.og[16 * hf + 4 * q + e];
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
                // feature jg + 16 k of the layer below: direction k >> 1, hidden unit jg + 16 (k & 1)
                stf(st_base + (int64_t)(k >> 1) * p.max_T * H * 32 + 16 * (k & 1) * 32, off0, pend[k]);
            }
        };
#ifdef SNN_ABL_S3_NOSTORE      // timing ablation: all dy stores into one slot
        auto dy_slot = [&](int tt) { return dy_out_base + (int64_t)d * dy_dir + 0 * tt; };
#else
        auto dy_slot = [&](int tt) { return dy_out_base + (int64_t)d * dy_dir + ((int64_t)tile * DIRS * p.max_T + tt) * H * 32; };
#endif
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
        // Everything of a step's gate gradients that does not depend on dh -- tanh(c), the derivative factors of the four
        // gates, dy from above -- is computed ahead (during phase P, from gate values fetched a step earlier), so that
        // between barrier Y and barrier X only  dh -> dc -> dG  remains: on this pipe a vector instruction waits for the
        // MFMA in flight, and the waves of a workgroup all sit in that chain at the same time.
        struct Fac { float A[UPT], Bv[UPT], Gi[UPT], Gf[UPT], Gg[UPT], fg[UPT], dyv[UPT]; };
        Fac fac;
        auto precompute = [&](int step, bool live, const QSet& cv) {
            const int t = d == 0 ? step : T - 1 - step;
            const int tp = d == 0 ? t - 1 : t + 1;
            const bool valid = live && t < len;
            const bool prev_live = step > 0 && tp < len;
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                const float ig = cv.q[0][i], fg = cv.q[1][i], gg = cv.q[2][i], og = cv.q[3][i];
                const float cp = prev_live ? cv.q[5][i] : 0.0f;
                const float tc = snn_tanh(c_carry[i]);
                c_carry[i] = cv.q[5][i];                                      // this step's q 5 is the next step's c
                fac.A[i] = valid ? og * (1.0f - tc * tc) : 0.0f;             // (selects, not products: tape and dy beyond a curve's end are never written)
                fac.Bv[i] = valid ? tc * og * (1.0f - og) : 0.0f;
                fac.Gi[i] = valid ? gg * ig * (1.0f - ig) : 0.0f;
                fac.Gf[i] = valid ? cp * fg * (1.0f - fg) : 0.0f;
                fac.Gg[i] = valid ? ig * (1.0f - gg * gg) : 0.0f;
                fac.fg[i] = fg;
                float dyv = cv.dy[0][i];
                if (DIRS == 2) dyv += cv.dy[DIRS - 1][i];
                dyv = has_dy ? dyv : 0.0f;
                if (last_layer && y.pool == SNN_POOL_MEAN) dyv += dfm[i];
                fac.dyv[i] = valid ? dyv : 0.0f;
            }
        };
        auto algebra = [&](int step, bool live, bool with_red, int wb, auto&& mf) {
            const int t = d == 0 ? step : T - 1 - step;
            const bool valid = live && t < len;
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
            float dh[UPT], dct[UPT], dao[UPT], dai[UPT], daf[UPT], dag[UPT];
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                dh[i] = dhr[i] + fac.dyv[i];
                dct[i] = dc[i] + dh[i] * fac.A[i];
            }
            mf(3);
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                dao[i] = dh[i] * fac.Bv[i];
                dai[i] = dct[i] * fac.Gi[i];
            }
            mf(4);
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                daf[i] = dct[i] * fac.Gf[i];
                dag[i] = dct[i] * fac.Gg[i];
            }
            mf(5);
#pragma unroll
            for (int i = 0; i < UPT; ++i) dc[i] = valid ? dct[i] * fac.fg[i] : dc[i];
            mf(6);
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
                const int o = masked ? origs_c[c] : 0;
#pragma unroll
                for (int i = 0; i < MTX * 2; ++i) {
                    float s = ((r2[i][0] + r2[i][1]) + r2[i][2]) + r2[i][3];
                    if (masked) {
                        const float mm = o >= 0 ? mask_mid(p.mk, 0, y.L, p.n_curves, DIRS * H, l - 1, t_prev, o, jg + 16 * i) : 1.0f;
                        s = __fmul_rn(mm, s);
                    }
                    pend[i] = with_red ? s : pend[i];
                }
                pend_slot = with_red ? dy_slot(t_prev) : pend_slot;
            }
            mf(11);
#pragma unroll
            for (int i = 0; i < UPT; ++i) { db[0 * UPT + i] += dai[i]; db[1 * UPT + i] += daf[i]; }
            mf(12);
#pragma unroll
            for (int i = 0; i < UPT; ++i) { db[2 * UPT + i] += dag[i]; db[3 * UPT + i] += dao[i]; }
            mf(13);
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
            if (masked && tid < 32) origs_n[tid] = n_order;   // the next tile's curves, for the masks of its first [h | x] fetch (in phase Q, behind barrier Y)
            precompute(step - 1, step > 0, nq);
            __builtin_amdgcn_sched_barrier(0);
            const FetchBase fq = fetch_base(step - 2), fx = fetch_base(step - 1);
            float* st_base = FIRST ? nullptr : (pend_slot ? pend_slot : dy_slot(t));   // (the very first step: zeros into its own slot, overwritten one step later by the same thread)
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
                // (the MFMAs evenly over the slices; issuing few of them before barrier X and the rest back to back after
                //  it -- 0 + 0, 18 + 0, 9 + 4 of role 0 + role 1 -- measured 2 to 4 % slower)
#ifdef SNN_S3_POSTX      // measurement build: all of the phase's MFMAs after barrier X (slices 9 .. 15)
                const int m0 = k <= 8 ? 0 : ((k - 9) * NM) / 7, m1 = k <= 8 ? 0 : ((k - 8) * NM) / 7;
#else
                const int m0 = (k * NM) / 16, m1 = ((k + 1) * NM) / 16;
#endif
#pragma unroll
                for (int m = m0; m < m1; ++m) {
                    const int kk = m / (MYNT > 0 ? MYNT : 1), mi = m - kk * (MYNT > 0 ? MYNT : 1);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], cx.xh[mi][kk >> 2][kk & 3], acc[mi], 0, 0, 0);
                }
#pragma unroll
                for (int q = (k * NXH) / 16; q < ((k + 1) * NXH) / 16; ++q) piece_x(q, fx, nx);
                __builtin_amdgcn_sched_barrier(0);
            };
            algebra(step - 1, step > 0, true, b ^ 1, mf);
#ifdef SNN_PROBE_STREAM
            probe[1] += pt2 - pt1; probe[2] += pt3 - pt2; probe[3] += clock64() - pt3; probe[4] += 1;
#endif
        };
        // ---- prologue: operands of the first two steps (unless the tile before fetched them), gate gradients of the first
        {
            if (!pref) {
                const FetchBase f0 = fetch_base(T - 1);
#pragma unroll
                for (int k = 0; k < NQL; ++k) piece_q(k, f0, qA);
#pragma unroll
                for (int k = 0; k < NXH; ++k) piece_x(k, f0, xA);
            }
            if (!pref || T == 1) {             // (a one-step tile: what the tile before fetched as "second step" is not the next tile's first)
                const FetchBase f1 = fetch_base(T - 2);
#pragma unroll
                for (int k = 0; k < NQL; ++k) piece_q(k, f1, qB);
            }
            precompute(T - 1, true, qA);
            algebra(T - 1, true, false, 0, [](int) {});
        }
#ifdef SNN_PROBE_STREAM
        const long long ph3 = clock64();
        pst[10] += ph1 - ph0; pst[11] += ph2 - ph1; pst[0] += ph3 - ph2;
#endif
        int step = T - 1;
        for (; step >= 1; step -= 2) {
            do_step(step, qA, qB, xA, xB, 0);
            do_step(step - 1, qB, qA, xB, xA, 1);
        }
        if (step == 0) do_step(0, qA, qB, xA, xB, 0);
#ifdef SNN_PROBE_STREAM
        const long long ph4 = clock64();
#endif
        if (T & 1) {                           // the sets alternate per step: after an odd tile the next tile's first step sits in the B sets
            const QSet tq = qA; qA = qB; qB = tq;    // (moving a register waits for its fetch: done before the stores below are
            const XSet tx = xA; xA = xB; xB = tx;    //  issued, when the youngest outstanding fetch is half a step old)
        }
#ifdef SNN_PROBE_STREAM
        probe[0] += clock64() - ph4;
#endif
        cur = nxt_real;
        pref = has_next;
        if (tile_n < 0) break;
    }

    if (!FIRST && pend_slot) {                 // the dy of the last step of the last tile
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            stf(pend_slot + (int64_t)(k >> 1) * p.max_T * H * 32 + 16 * (k & 1) * 32, off0, pend[k]);
        }
    }
#ifdef SNN_PROBE_STREAM
    if (blockIdx.x == 0 && (threadIdx.x & 255) == 0)
        printf("probe3 l=%d role=%d steps=%lld  phase P %lld (products %lld)  barrier Y %lld  phase Q with barrier X %lld (slice starts 0, 2, .. 14: %lld %lld %lld %lld %lld %lld %lld %lld)  per step; loop total %lld per step\n",
               l, ROLE, probe[4], probe[1] / probe[4], pst[1] / probe[4], probe[2] / probe[4], probe[3] / probe[4],
               pst[2] / probe[4], pst[3] / probe[4], pst[4] / probe[4], pst[5] / probe[4], pst[6] / probe[4], pst[7] / probe[4], pst[8] / probe[4], pst[9] / probe[4],
               (clock64() - pk0) / probe[4]);
    if (blockIdx.x == 0 && (threadIdx.x & 255) == 0)
        printf("probe3t l=%d role=%d per tile (16 tiles): head loads %lld  barrier %lld  prologue %lld  tail %lld\n", l, ROLE, pst[10] / 16, pst[11] / 16, pst[0] / 16, probe[0] / 16);
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
