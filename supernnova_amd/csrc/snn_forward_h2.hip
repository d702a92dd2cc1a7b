// snn_forward_h2.hip -- persistent LSTM forward with the gate products on the fp16 matrix pipe (SNN_PREC_F16X2):
// every operand (x_t, h_{t-1}, W) is carried as two fp16 pieces, v = hi + lo * 2^-11 (split2, snn_kernels.h), and each
// fp32 product is rebuilt from three piece products
//     main  += hi * hi            cross += hi * lo + lo * hi            gate = main + 2^-11 * cross
// (fp16 x fp16 is exact in fp32; the dropped lo * lo term is < 2^-22 relative, 2^-25 RMS), accumulated in fp32 by
// v_mfma_f32_32x32x16_f16 in two independent accumulator chains.  Why, after bf16 x 3 (snn_forward_b3p.hip): that
// kernel is limited by the power its 24 / 36 MFMAs per step draw (the clock falls from 2.4 to 1.6 GHz,
// profiles/r1/README.md) and by ~145 VALU instructions per step; here the matrix work is halved, the exp2 scale of
// every gate is folded into the weight rows and biases (h2_gate_scale), and the cell update runs on packed fp32
// instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) over pairs of hidden units.
//
// Same decomposition and pipeline as snn_forward_b3p.hip: item = (group of 4 tiles, MC sample), two tiles at a time,
//     phase I : MFMA chains of tile 1, step t      ||  cell update of tile 0, step t   -> h0_t pieces to LDS
//     phase II: MFMA chains of tile 0, step t + 1  ||  cell update of tile 1, step t   -> h1_t pieces to LDS
// One kernel for every batch size (a lone tile runs with a dead partner), so results do not depend on how the
// curves are tiled.  Inference only, no masks, H = 32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "snn_kernels.h"
#include "snn_host.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#ifndef SNN_H2_WAVES
#define SNN_H2_WAVES 2
#endif

constexpr int BLK_U4 = 2 * 64;   // uint4 (16-byte) words of one K=16 block: 2 pieces x 64 lanes
constexpr int B3_G = SNN_H2_G;   // tiles of one work item (snn_kernels.h)
static_assert(B3_G % 2 == 0, "a work item is a whole number of tile pairs (the y slab has B3_G tile slots)");

// two floats -> two fp16 (round to nearest even) in one register, low half = a
__device__ __forceinline__ unsigned int cvt_pk_f16(float a, float b)
{
    unsigned int r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 exp2_2(f2 v) { return f2{__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)}; }
__device__ __forceinline__ f2 rcp_2(f2 v) { return f2{__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)}; }
__device__ __forceinline__ f2 fma_2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

struct H2Ctx {
    int s, w, lane, cl, hf, tid;
    const float* wks;
};

template <int H, int XB, int DIRS, bool FIRST, bool LAST, bool MASKED>
__device__ __forceinline__ void h2_phase(const FwdParams& p, const H2Ctx& c, int n_live, int tile0, const int* info,
                                         int l, int d, int64_t k_A, int64_t k_bias, int64_t k_head_w, u32x4* xin,
                                         u32x4* hb, float* part, float* logit_all, float* bl, u32x4* ybase_all)
{
    constexpr int NT = H * 8, NW = H / 8, HB = H / 16, NB = XB + HB, NM = NB * 3;
    constexpr int NF4 = XB * BLK_U4, NPT = (NF4 + NT - 1) / NT;   // 16-byte words of one step's x image
    constexpr int YB = DIRS * HB;                                   // blocks of one step of the y image
    constexpr int XS = XB * BLK_U4, HS = HB * BLK_U4;               // LDS words of one tile's x / h image
    const FwdParams& y = p;
    const int ld = l * DIRS + d;
    // Variational masks (MASKED): the kernel applies the Bernoulli BITS (mask value != 0) with ANDs on the fp16 pieces,
    // which is exact; the 1/(1-p) factor of the input, W_hh-row and between-layer masks is folded into the weights by
    // the caller (include/snn_hip.h, SNN_PREC_F16X2).  The output mask multiplies the pooled features in fp32.
    const MaskP& mk = p.mk;
    const bool masked = MASKED && mk.mode != SNN_MASK_NONE;

    // ---- weight pieces -> registers, shared by both tiles: wr[b][piece] = 8 fp16 (4 VGPRs)
    f16x8 wr[NB][2];
    {
        const u32x4* A = reinterpret_cast<const u32x4*>(c.wks + k_A) + (int64_t)c.w * NB * BLK_U4 + c.lane;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) wr[b][q] = __builtin_bit_cast(f16x8, A[(b * 2 + q) * 64]);
    }
    if (MASKED && masked && d == 0) {   // W_hh row mask, forward direction only (variational_rnn.py:45, 224-238)
        const int r = c.lane & 31;      // this lane's A-operand row: gate r >> 3 of hidden unit 8w + (r & 7)
        const unsigned int keep = mask_whh(mk, c.s, y.L, H, l, (r >> 3) * H + 8 * c.w + (r & 7)) != 0.0f ? 0xffffffffu : 0u;
#pragma unroll
        for (int b = XB; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                wr[b][q] = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4, wr[b][q]) & keep);
    }
    if (c.cl < 16) {   // b_ih + b_hh in C/D register order (entry i: gate i >> 2), times the gate's exp2 factor
        const float* B = c.wks + k_bias + (c.w * 2 + c.hf) * 16;
        bl[(c.w * 2 + c.hf) * 16 + c.cl] = __fmul_rn(B[c.cl] + B[NW * 32 + c.cl], h2_gate_scale(c.cl >> 2));
    }
    const f32x16* binit_p = reinterpret_cast<const f32x16*>(bl + (c.w * 2 + c.hf) * 16);
    const int j0 = 8 * c.w + 4 * c.hf;
    const int64_t y_tile_u4 = p.y_buf_floats / 4 / B3_G;   // 16-byte words of one tile's y slab
    // this lane's 4 hidden units sit in block w >> 1, lane' = (w & 1) * 32 + curve, elements 4 hf .. 4 hf + 3
    const int hblk = c.w >> 1, hlane = (c.w & 1) * 32 + c.cl;

  for (int pr = 0; 2 * pr < n_live; ++pr) {
    const bool has1 = 2 * pr + 1 < n_live;
    int tile[2], Tt[2], len[2];
    tile[0] = tile0 + 2 * pr;
    tile[1] = has1 ? tile[0] + 1 : tile[0];          // a missing partner aliases tile 0's input, with no live curve
    Tt[0] = __builtin_amdgcn_readfirstlane(p.tile_T[tile[0]]);
    Tt[1] = __builtin_amdgcn_readfirstlane(p.tile_T[tile[1]]);
    const int T = Tt[0];                             // tiles are sorted by length, descending
    len[0] = info[(2 * pr) * 64 + c.cl];
    len[1] = has1 ? info[(2 * pr + 1) * 64 + c.cl] : 0;

    const u32x4* in0[2];
    u32x4* yout[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        u32x4* ybase = ybase_all + (2 * pr + q) * y_tile_u4;
        // (tile_off is in fp32-image floats, T * Dp * 32 per tile; this image holds XB * BLK_U4 16-byte words per step)
        in0[q] = FIRST ? reinterpret_cast<const u32x4*>(p.xp) + p.tile_off[tile[q]] / ((int64_t)p.Dp * SNN_TILE) * NF4
                       : ybase + (int64_t)((l - 1) % p.nbuf) * p.y_buf_floats / 4;
        yout[q] = LAST ? nullptr : ybase + (int64_t)(l % p.nbuf) * p.y_buf_floats / 4;
    }
    __syncthreads();   // the previous pair's head reduction has finished with part / logit before this pair's staging
    // the lane's 16 bias values stay in VGPRs (4 fewer ds_read_b128 per phase: -1.7 %); the masked variants have no
    // registers left for them and re-read the LDS copy
    f32x16 binit_v;
    if (!MASKED) binit_v = *binit_p;

    // mask bits of this pair: the between-layer mask of the lane's own 4 outputs (applied by the producer), and the
    // input mask of the words this thread stages (8 features of one curve each); both locked over time
    unsigned int ymask[2][2] = {{~0u, ~0u}, {~0u, ~0u}};
    u32x4 xm[2][NPT];
    if (MASKED) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool live = q == 0 || has1;
            if (!LAST && masked) {
                const int orig = live ? info[(2 * pr + q) * 64 + 32 + c.cl] : -1;
                if (orig >= 0) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const bool b0 = mask_mid(mk, c.s, y.L, p.n_curves, DIRS * H, l, 0, orig, d * H + j0 + 2 * k) != 0.0f;
                        const bool b1 = mask_mid(mk, c.s, y.L, p.n_curves, DIRS * H, l, 0, orig, d * H + j0 + 2 * k + 1) != 0.0f;
                        ymask[q][k] = (b0 ? 0x0000ffffu : 0u) | (b1 ? 0xffff0000u : 0u);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                xm[q][i] = u32x4{~0u, ~0u, ~0u, ~0u};
                if (FIRST && masked) {
                    const int f = c.tid + i * NT < NF4 ? c.tid + i * NT : NF4 - 1;
                    const int b = f / BLK_U4, lp = f & 63;
                    const int orig = live ? info[(2 * pr + q) * 64 + 32 + (lp & 31)] : -1;
                    if (orig >= 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int k = 16 * b + 8 * (lp >> 5) + e;
                            if (k < y.D && mask_in(mk, c.s, p.n_curves, y.D, orig, k) == 0.0f)
                                xm[q][i][e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
                        }
                    }
                }
            }
        }
    }

    // branch-free staging: threads beyond the image duplicate its last word (same value to the same address)
    auto load_step = [&](int q, int t, u32x4 (&pre)[NPT]) {
        if (FIRST) t = t < Tt[q] ? t : Tt[q] - 1;   // tile q may hold fewer steps than the pair runs
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = c.tid + i * NT < NF4 ? c.tid + i * NT : NF4 - 1;
#ifdef SNN_ABL_NOYLD   // timing ablation only: every step re-reads the first image (L2-resident)
            pre[i] = in0[q][(int64_t)(FIRST ? t : 0) * NF4 + f];
#else
            pre[i] = in0[q][(int64_t)t * NF4 + f];
#endif
        }
    };
    auto store_step = [&](int q, const u32x4 (&pre)[NPT]) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = c.tid + i * NT < NF4 ? c.tid + i * NT : NF4 - 1;
            xin[q * XS + f] = (MASKED && FIRST) ? (pre[i] & xm[q][i]) : pre[i];
        }
    };

    // state of the lane's 4 hidden units as two pairs (packed fp32 math works on register pairs)
    f2 cst[2][2], hst[2][2], hsum[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < 2; ++k) cst[q][k] = hst[q][k] = hsum[q][k] = f2{0.0f, 0.0f};

    // The cell update of one tile-step in NPIECE pieces: 8 per pair of hidden units, then the stores.  They are spread
    // evenly over the NM MFMAs of the other tile's chains.  All gate pre-activations arrive scaled for exp2.
    constexpr int NPIECE = 17;
    struct ActRegs { f2 di[2], dg[2], df[2], dc[2], dq[2], ig[2], gg[2]; unsigned int ph[2], pl[2]; };
    auto act_piece = [&](int qa, int ta, const f32x16& aM, const f32x16& aX, ActRegs& r, unsigned long long vmask,
                         unsigned int vbits, int piece) {
        // `pin`: an empty volatile asm on the piece's input keeps the piece behind the sched_barrier before it
        auto pin = [](float v) { asm volatile("" : "+v"(v)); return v; };
        // Scalar fp32 instructions on the two units of a pair, interleaved: beside MFMAs a packed fp32 instruction
        // costs far more issue time than two scalar ones (MI355X_MICROARCH.md), and with two independent chains the
        // consumer of a v_exp_f32 / v_rcp_f32 result is never the next instruction issued (no `s_nop` for the
        // transcendental-use hazard).  Built with -fno-slp-vectorize so the compiler does not re-pack them.
        // gate g of unit u: main + 2^-11 * cross
        auto gate = [&](int g, int u) { return fmaf(aX[4 * g + u], SNN_H2_LO_INV, aM[4 * g + u]); };
        constexpr float KC = 2.88539008177792681f;   // 2 log2(e)
        auto ex = [](float v) { return __builtin_amdgcn_exp2f(v); };
        auto rc = [](float v) { return __builtin_amdgcn_rcpf(v); };
        if (piece == 16) {
            u32x4* hn = hb + qa * HS;
            reinterpret_cast<uint2*>(hn + (hblk * 2 + 0) * 64 + hlane)[c.hf] = uint2{r.ph[0], r.ph[1]};
            reinterpret_cast<uint2*>(hn + (hblk * 2 + 1) * 64 + hlane)[c.hf] = uint2{r.pl[0], r.pl[1]};
            if (!LAST) {
                u32x4* yo = yout[qa] + ((int64_t)ta * YB + d * HB + hblk) * BLK_U4 + hlane;
#ifdef SNN_ABL_NOYST   // timing ablation only
                if (p.n_curves < 0)
#endif
                {
                const unsigned int v0 = MASKED ? vbits & ymask[qa][0] : vbits, v1 = MASKED ? vbits & ymask[qa][1] : vbits;
                reinterpret_cast<uint2*>(yo + 0 * 64)[c.hf] = uint2{r.ph[0] & v0, r.ph[1] & v1};
                reinterpret_cast<uint2*>(yo + 1 * 64)[c.hf] = uint2{r.pl[0] & v0, r.pl[1] & v1};
                }
            }
            return;
        }
        const int k = piece >> 3, st = piece & 7, ua = 2 * k, ub = 2 * k + 1;
        if (st == 0) {                                                            // 1 + e^-i
            const float a = ex(pin(gate(0, ua))), b = ex(pin(gate(0, ub)));
            r.di[k] = f2{1.0f + a, 1.0f + b};
        } else if (st == 1) {                                                     // 1 + e^2g
            const float a = ex(pin(gate(2, ua))), b = ex(pin(gate(2, ub)));
            r.dg[k] = f2{1.0f + a, 1.0f + b};
        } else if (st == 2) {                                                     // 1 + e^-f
            const float a = ex(pin(gate(1, ua))), b = ex(pin(gate(1, ub)));
            r.df[k] = f2{1.0f + a, 1.0f + b};
        } else if (st == 3) {
            const float ia = rc(pin(r.di[k].x)), ib = rc(pin(r.di[k].y));
            const float ga = rc(r.dg[k].x), gb = rc(r.dg[k].y);
            r.ig[k] = f2{ia, ib};
            // K tanh(g), K = 2 log2(e): the cell state is carried as K c, ready for the exp2 of tanh(c)
            r.gg[k] = f2{fmaf(-2.0f * KC, ga, KC), fmaf(-2.0f * KC, gb, KC)};
        } else if (st == 4) {
            const float fa = rc(pin(r.df[k].x)), fb = rc(pin(r.df[k].y));
            const float pa = r.ig[k].x * r.gg[k].x, pb = r.ig[k].y * r.gg[k].y;
            const float ca = fmaf(fa, cst[qa][k].x, pa), cb = fmaf(fb, cst[qa][k].y, pb);      // K c_t
            r.dc[k] = f2{ca, cb};
            cst[qa][k] = f2{lane_select(vmask, ca, cst[qa][k].x), lane_select(vmask, cb, cst[qa][k].y)};
        } else if (st == 5) {
            const float ca = ex(pin(r.dc[k].x)), cb = ex(pin(r.dc[k].y));            // e^2c
            const float oa = ex(gate(3, ua)), ob = ex(gate(3, ub));                  // e^-o
            r.dc[k] = f2{1.0f + ca, 1.0f + cb};
            r.dq[k] = f2{1.0f + oa, 1.0f + ob};
        } else if (st == 6) {
            const float oa = rc(pin(r.dq[k].x)), ob = rc(pin(r.dq[k].y));
            const float ta_ = rc(r.dc[k].x), tb_ = rc(r.dc[k].y);
            const float ha = oa * fmaf(-2.0f, ta_, 1.0f), hb_ = ob * fmaf(-2.0f, tb_, 1.0f);   // sigmoid(o) tanh(c)
            hst[qa][k] = f2{lane_select(vmask, ha, hst[qa][k].x), lane_select(vmask, hb_, hst[qa][k].y)};
            if (LAST) {
                hsum[qa][k].x += lane_select(vmask, ha, 0.0f);
                hsum[qa][k].y += lane_select(vmask, hb_, 0.0f);
            }
        } else {   // split2 of the pair
            const float ha = pin(hst[qa][k].x), hb_ = pin(hst[qa][k].y);
            const unsigned int hi = cvt_pk_f16(ha, hb_);
            const float fa = (float)__builtin_bit_cast(_Float16, (unsigned short)(hi & 0xffffu));
            const float fb = (float)__builtin_bit_cast(_Float16, (unsigned short)(hi >> 16));
            r.ph[k] = hi;
            // (h - hi) * 2^11, exact either way; written as an fma on the fp16 operand (v_fma_mix_f32: no separate cvt)
            r.pl[k] = cvt_pk_f16(fmaf(fa, -SNN_H2_LO_SCALE, ha * SNN_H2_LO_SCALE), fmaf(fb, -SNN_H2_LO_SCALE, hb_ * SNN_H2_LO_SCALE));
        }
    };
    // One pipeline phase: the MFMA chains of tile qc (into mc / xc), with the cell update of tile qa (from ma / xa, step
    // ta) issued piecewise behind the MFMAs.  sched_barrier(0) pins the program order: nothing crosses it.
    auto phase = [&](int qc, f32x16& mc, f32x16& xc, bool do_act, int qa, int ta, const f32x16& ma, const f32x16& xa) {
        const unsigned long long vmask = __builtin_amdgcn_ballot_w64(ta < len[qa]);
        const unsigned int vbits = ta < len[qa] ? 0xffffffffu : 0u;
        ActRegs r;
        f16x8 bv[NB][2];
        const u32x4* xs = xin + qc * XS + c.lane;
        const u32x4* hs = hb + qc * HS + c.lane;
        auto bload = [&](int b) {
            const u32x4* src = b < XB ? xs + b * BLK_U4 : hs + (b - XB) * BLK_U4;
#pragma unroll
            for (int q = 0; q < 2; ++q) bv[b][q] = __builtin_bit_cast(f16x8, src[q * 64]);
        };
        bload(0);
        mc = MASKED ? *binit_p : binit_v;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b + 1 < NB) bload(b + 1);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
#if defined(SNN_ABL_NOMFMA)     // timing ablation only: one product per block instead of three
                if (j == 2 || b == 0)
#endif
                {
                if (j == 0) {
                    const f32x16 zero = {0};
                    xc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[b][0], bv[b][1], b == 0 ? zero : xc, 0, 0, 0);   // hi * lo
                } else if (j == 1) xc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[b][1], bv[b][0], xc, 0, 0, 0);   // lo * hi
                else mc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[b][0], bv[b][0], mc, 0, 0, 0);                 // hi * hi
                }
                if (do_act) {
                    const int m = b * 3 + j;
                    const int lo = (m * NPIECE + NM - 1) / NM, hi = ((m + 1) * NPIECE + NM - 1) / NM;
#pragma unroll
                    for (int pc = lo; pc < hi; ++pc) act_piece(qa, ta, ma, xa, r, vmask, vbits, pc);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto act_only = [&](int qa, int ta, const f32x16& ma, const f32x16& xa) {
        const unsigned long long vmask = __builtin_amdgcn_ballot_w64(ta < len[qa]);
        const unsigned int vbits = ta < len[qa] ? 0xffffffffu : 0u;
        ActRegs r;
#pragma unroll
        for (int pc = 0; pc < NPIECE; ++pc) act_piece(qa, ta, ma, xa, r, vmask, vbits, pc);
    };

    // ---- prologue: x(t0) of both tiles, h = 0, request x(t1), first chains of tile 0
    const int dt = d == 0 ? 1 : -1;
    const int t0 = d == 0 ? 0 : T - 1;
    u32x4 pre0[NPT], pre1[NPT];
    {
        load_step(0, t0, pre0);
        load_step(1, t0, pre1);
        store_step(0, pre0);
        store_step(1, pre1);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int k = 0; k < 2; ++k)
                reinterpret_cast<uint2*>(hb + q * HS + (hblk * 2 + k) * 64 + hlane)[c.hf] = uint2{0u, 0u};
        const int t1 = T > 1 ? t0 + dt : t0;
        load_step(0, t1, pre0);
        load_step(1, t1, pre1);
    }
    lds_barrier();
    f32x16 m0, x0, m1, x1;
    phase(0, m0, x0, false, 0, 0, m0, x0);
    lds_barrier();   // every wave has read h0 = 0 before the first cell update overwrites it

    for (int step = 0; step + 1 < T; ++step) {
        const int t = t0 + dt * step;
        const int tnn = step + 2 < T ? t + 2 * dt : t + dt;
        // phase I: chains of tile 1 (step t) || cell update of tile 0 (step t)
        phase(1, m1, x1, true, 0, t, m0, x0);
        store_step(0, pre0);          // x0(t + 1)
        load_step(0, tnn, pre0);
        lds_barrier();
        // phase II: chains of tile 0 (step t + 1) || cell update of tile 1 (step t)
        phase(0, m0, x0, true, 1, t, m1, x1);
        store_step(1, pre1);
        load_step(1, tnn, pre1);
        lds_barrier();
    }
    {   // last step
        const int t = t0 + dt * (T - 1);
        phase(1, m1, x1, true, 0, t, m0, x0);
        lds_barrier();   // tile 1's cell update overwrites the h image the other waves' chains may still be reading
        act_only(1, t, m1, x1);
    }

    // ---- head contribution of this phase (fp32, as lstm_fwd_kernel), tile by tile
    const bool contributes = y.pool == SNN_POOL_STANDARD || LAST;
    if (contributes) {
        const float* Wo = c.wks + k_head_w;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float* logit = logit_all + (2 * pr + q) * y.C * 32;
            float feat[4];
            int fidx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (y.pool == SNN_POOL_MEAN) {
                    feat[u] = __fdiv_rn(hsum[q][u >> 1][u & 1], (float)(len[q] > 0 ? len[q] : 1));
                    fidx[u] = d * H + j0 + u;
                } else {
                    feat[u] = hst[q][u >> 1][u & 1];
                    fidx[u] = (j0 + u) * (y.L * DIRS) + ld;
                }
            }
            if (MASKED && masked) {
                const int orig = (q == 0 || has1) ? info[(2 * pr + q) * 64 + 32 + c.cl] : -1;
                if (orig >= 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) feat[u] = __fmul_rn(mask_out(mk, c.s, p.n_curves, y.F, orig, fidx[u]), feat[u]);
                }
            }
            for (int cq = 0; cq < y.C; ++cq) {
                float a = 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u) a = fmaf(Wo[cq * y.F + fidx[u]], feat[u], a);
                a += __shfl_xor(a, 32);
                if (c.hf == 0) part[(c.w * y.C + cq) * 32 + c.cl] = a;
            }
            __syncthreads();
            for (int idx = c.tid; idx < y.C * 32; idx += NT) {
                const int cq = idx >> 5, cc = idx & 31;
                float a = logit[idx];
                for (int w = 0; w < NW; ++w) a += part[(w * y.C + cq) * 32 + cc];
                logit[idx] = a;
            }
            __syncthreads();
        }
    }
    __syncthreads();  // also publishes this phase's y stores to the workgroup
  }
}

template <int H, int XB0, int DIRS, bool MASKED>
__global__ void __launch_bounds__(H * 8, SNN_H2_WAVES) lstm_fwd_h2_kernel(const FwdParams p, const B3Offsets o)
{
    constexpr int NT = H * 8, NW = H / 8, HB = H / 16;
    constexpr int XB1 = DIRS * HB;
    constexpr int XBM = XB0 > XB1 ? XB0 : XB1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdParams& y = p;
    u32x4* xin = reinterpret_cast<u32x4*>(smem);            // [2 tiles][XBM blocks]
    u32x4* hb = xin + 2 * XBM * BLK_U4;                     // [2 tiles][HB blocks]
    float* part = reinterpret_cast<float*>(hb + 2 * HB * BLK_U4);   // [NW][C][32]
    float* logit = part + NW * y.C * 32;                    // [B3_G][C][32]
    float* bl = logit + B3_G * y.C * 32;                    // [NW][2][16]
    int* info = reinterpret_cast<int*>(bl + NW * 32);       // [B3_G][64]
    int* item_slot = info + B3_G * 64;

    H2Ctx c;
    c.tid = threadIdx.x;
    c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.cl = c.lane & 31;
    c.hf = c.lane >> 5;
    u32x4* ybase = reinterpret_cast<u32x4*>(p.yscratch) + (int64_t)blockIdx.x * p.nbuf * p.y_buf_floats / 4;
    // Work queues, one per XCD (workgroups are dispatched round-robin over the 8 XCDs, each with its own L2):
    // queue x holds the tile groups x, x + 8, ... with all MC samples of a group adjacent, so the x image of a
    // group is fetched into ONE L2 and shared by its S items.  An XCD that runs dry steals from the next queue.
    const int n_groups = (p.n_tiles + B3_G - 1) / B3_G;
    const int xcd = blockIdx.x & 7;

    for (int k = 0; k < 8; ++k) {
        const int qx = (xcd + k) & 7;
        const int64_t n_items = (int64_t)((n_groups - qx + 7) / 8) * p.n_samples;
        for (;;) {
            if (c.tid == 0) *item_slot = (int)atomicAdd(p.counter + qx * 16, 1u);
            __syncthreads();
            const int item = __builtin_amdgcn_readfirstlane(*item_slot);
            if (item >= n_items) break;
            const int g = qx + 8 * (item / p.n_samples);
            c.s = item % p.n_samples;
            const int tile0 = g * B3_G;
            const int n_live = min(B3_G, p.n_tiles - tile0);
            c.wks = p.wk + (int64_t)c.s * p.wk_stride;
            for (int idx = c.tid; idx < n_live * 64; idx += NT) {
                const int q = idx >> 6, r = idx & 63;
                const int pos = (tile0 + q) * SNN_TILE + (r & 31);
                info[idx] = r < 32 ? p.len_sorted[pos] : p.order[pos];
            }
            for (int idx = c.tid; idx < B3_G * y.C * 32; idx += NT) logit[idx] = 0.0f;
            __syncthreads();

            for (int l = 0; l < y.L; ++l) {
                const bool last = l == y.L - 1;
                for (int d = 0; d < DIRS; ++d) {
                    const int ld = l * DIRS + d;
                    if (l == 0) {
                        if (last) h2_phase<H, XB0, DIRS, true, true, MASKED>(p, c, n_live, tile0, info, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
                        else h2_phase<H, XB0, DIRS, true, false, MASKED>(p, c, n_live, tile0, info, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
                    } else {
                        if (last) h2_phase<H, XB1, DIRS, false, true, MASKED>(p, c, n_live, tile0, info, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
                        else h2_phase<H, XB1, DIRS, false, false, MASKED>(p, c, n_live, tile0, info, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
                    }
                }
            }

            // ---- head bias + softmax, one thread per curve of the group
            for (int idx = c.tid; idx < n_live * 32; idx += NT) {
                const int q = idx >> 5, cc = idx & 31;
                const int orig = info[q * 64 + 32 + cc];
                if (orig >= 0) {
                    const float* lg_ = logit + q * y.C * 32;
                    const float* bo = c.wks + o.k_head_w + y.C * y.F;
                    float mx = -INFINITY;
                    for (int k2 = 0; k2 < y.C; ++k2) mx = fmaxf(mx, lg_[k2 * 32 + cc] + bo[k2]);
                    float z = 0.0f;
                    for (int k2 = 0; k2 < y.C; ++k2) z += expf(lg_[k2 * 32 + cc] + bo[k2] - mx);
                    const int64_t off = (int64_t)c.s * p.os_s + (int64_t)orig * p.os_n;
                    for (int k2 = 0; k2 < y.C; ++k2) {
                        const float lg = lg_[k2 * 32 + cc] + bo[k2];
                        if (p.out_logits) p.out_logits[off + k2] = lg;
                        if (p.out_probs) p.out_probs[off + k2] = __fdiv_rn(expf(lg - mx), z);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- registry hook used by snn_forward.hip -----------------------------------------------------------
typedef void (*fwd_b3_kernel_t)(const FwdParams, const B3Offsets);
fwd_b3_kernel_t snn_h2_kernel(int H, int XB0, int dirs, bool masked)
{
    if (H == 32 && dirs == 2) {
        if (masked) {   // (the D <= 78 variant has no registers left for the staged input-mask words)
            if (XB0 == 2) return lstm_fwd_h2_kernel<32, 2, 2, true>;
            if (XB0 == 3) return lstm_fwd_h2_kernel<32, 3, 2, true>;
            return nullptr;
        }
        if (XB0 == 2) return lstm_fwd_h2_kernel<32, 2, 2, false>;   // D <= 26 (buckets 24, 26)
        if (XB0 == 3) return lstm_fwd_h2_kernel<32, 3, 2, false>;   // D <= 48
        if (XB0 == 5) return lstm_fwd_h2_kernel<32, 5, 2, false>;   // D <= 78
    }
    return nullptr;
}
