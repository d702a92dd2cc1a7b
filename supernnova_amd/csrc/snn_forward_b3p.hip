// snn_forward_b3p.hip -- bf16 x 3 LSTM forward (see snn_forward_b3.hip for the arithmetic), two 32-curve tiles per
// workgroup with the time loop software-pipelined across the pair:
//
//     phase I : MFMA chain of tile 1, step t      ||  cell update of tile 0, step t   -> h0_t pieces to LDS
//     barrier
//     phase II: MFMA chain of tile 0, step t + 1  ||  cell update of tile 1, step t   -> h1_t pieces to LDS
//     barrier
//
// Why: timing ablations of the one-tile kernel (profiles/r1/README.md) show its time is the SUM of the MFMA chain
// (24 / 36 dependent v_mfma_f32_32x32x16_bf16 per step), the cell's VALU work and the y traffic: other waves of the SIMD
// do not fill the matrix pipe while a wave computes its cell.  The bf16 MFMA does run concurrently with VALU issued
// by the SAME wave (microbench/mfma_valu_overlap.hip), so here every wave always has a chain in flight whose result
// it does not need yet, and the other tile's cell update, bf16 split and stores are cut into pieces placed behind
// its MFMAs.  Same products in the same order as the one-tile kernel: results are bit-identical.
// The x / y image of step t + 2 is requested right after the image of step t + 1 went to LDS (two phases of latency
// hiding).  Inference only, no masks, H = 32 (SNN_PREC_BF16X3; include/snn_hip.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "snn_kernels.h"
#include "snn_host.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef SNN_B3P_WAVES
#define SNN_B3P_WAVES 2
#endif

constexpr int BLK_U4 = 3 * 64;   // uint4 (16-byte) words of one K=16 block: 3 pieces x 64 lanes
constexpr int B3_G = 4;          // tiles of one work item (two pairs), as in snn_forward_b3.hip

// two floats -> two bf16 (round to nearest even) in one register, low half = a
__device__ __forceinline__ unsigned int cvt_pk_bf16(float a, float b)
{
    unsigned int r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct B3PCtx {
    int s, w, lane, cl, hf, tid;
    const float* wks;
};

template <int H, int XB, int DIRS, bool FIRST, bool LAST>
__device__ __forceinline__ void b3p_phase(const FwdParams& p, const B3PCtx& c, int n_live, int tile0, const int* info,
                                          int l, int d, int64_t k_A, int64_t k_bias, int64_t k_head_w, u32x4* xin,
                                          u32x4* hb, float* part, float* logit_all, float* bl, u32x4* ybase_all)
{
    constexpr int NT = H * 8, NW = H / 8, HB = H / 16, NB = XB + HB, NM = NB * 6;
    constexpr int NF4 = XB * BLK_U4, NPT = (NF4 + NT - 1) / NT;   // 16-byte words of one step's x image
    constexpr int YB = DIRS * HB;                                   // blocks of one step of the y image
    constexpr int XS = XB * BLK_U4, HS = HB * BLK_U4;               // LDS words of one tile's x / h image
    const FwdParams& y = p;
    const int ld = l * DIRS + d;

    // ---- weight pieces -> registers, shared by both tiles: wr[b][piece] = 8 bf16 (4 VGPRs)
    bf16x8 wr[NB][3];
    {
        const u32x4* A = reinterpret_cast<const u32x4*>(c.wks + k_A) + (int64_t)c.w * NB * BLK_U4 + c.lane;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 3; ++q) wr[b][q] = __builtin_bit_cast(bf16x8, A[(b * 3 + q) * 64]);
    }
    if (c.cl < 16) {
        const float* B = c.wks + k_bias + (c.w * 2 + c.hf) * 16;
        bl[(c.w * 2 + c.hf) * 16 + c.cl] = B[c.cl] + B[NW * 32 + c.cl];
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
            xin[q * XS + f] = pre[i];
        }
    };

    float cst[2][4], hst[2][4], hsum[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) cst[q][u] = hst[q][u] = hsum[q][u] = 0.0f;

    // The cell update of one tile-step in NPIECE pieces: 6 per hidden unit (the three gates, the cell, tanh(c), h),
    // the bf16 split of units 0-1 / 2-3, and the stores.  They are spread evenly over the NM MFMAs of the other
    // tile's chain (~32 cycles of matrix time each: room for about 6 VALU instructions).
    constexpr int NPIECE = 27;
    struct ActRegs { float ig[4], fg[4], gg[4], og[4], tc[4]; unsigned int ph[2], pm[2], pl[2]; };
    auto split_pair = [&](int qa, ActRegs& r, int k) {   // units 2k, 2k+1 -> hi / mid / lo pieces
        const float a = hst[qa][2 * k], b = hst[qa][2 * k + 1];
        const unsigned int hi = cvt_pk_bf16(a, b);
        const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
        const unsigned int mid = cvt_pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
        r.ph[k] = hi; r.pm[k] = mid; r.pl[k] = cvt_pk_bf16(sa, sb);
    };
    auto act_piece = [&](int qa, int ta, const f32x16& acc, ActRegs& r, unsigned long long vmask, unsigned int vbits,
                         int piece) {
        // `pin`: an empty volatile asm on the piece's input keeps the piece behind the sched_barrier before it
        auto pin = [](float v) { asm volatile("" : "+v"(v)); return v; };
        if (piece == 12 || piece == 25) { split_pair(qa, r, piece == 12 ? 0 : 1); return; }
        if (piece == 26) {
            u32x4* hn = hb + qa * HS;
            reinterpret_cast<uint2*>(hn + (hblk * 3 + 0) * 64 + hlane)[c.hf] = uint2{r.ph[0], r.ph[1]};
            reinterpret_cast<uint2*>(hn + (hblk * 3 + 1) * 64 + hlane)[c.hf] = uint2{r.pm[0], r.pm[1]};
            reinterpret_cast<uint2*>(hn + (hblk * 3 + 2) * 64 + hlane)[c.hf] = uint2{r.pl[0], r.pl[1]};
            if (!LAST) {
                u32x4* yo = yout[qa] + ((int64_t)ta * YB + d * HB + hblk) * BLK_U4 + hlane;
#ifdef SNN_ABL_NOYST   // timing ablation only
                if (p.n_curves < 0)
#endif
                {
                reinterpret_cast<uint2*>(yo + 0 * 64)[c.hf] = uint2{r.ph[0] & vbits, r.ph[1] & vbits};
                reinterpret_cast<uint2*>(yo + 1 * 64)[c.hf] = uint2{r.pm[0] & vbits, r.pm[1] & vbits};
                reinterpret_cast<uint2*>(yo + 2 * 64)[c.hf] = uint2{r.pl[0] & vbits, r.pl[1] & vbits};
                }
            }
            return;
        }
        const int pu = piece < 12 ? piece : piece - 1;   // 0..23
        const int u = pu / 6, st = pu % 6;
        if (st == 0) r.ig[u] = snn_sigmoid(pin(acc[u]));
        else if (st == 1) r.gg[u] = snn_tanh(pin(acc[8 + u]));
        else if (st == 2) r.fg[u] = snn_sigmoid(pin(acc[4 + u]));
        else if (st == 3) {
            const float cn = fmaf(pin(r.fg[u]), cst[qa][u], __fmul_rn(r.ig[u], r.gg[u]));   // as the one-tile kernel
            r.tc[u] = snn_tanh(cn);
            cst[qa][u] = lane_select(vmask, cn, cst[qa][u]);
        } else if (st == 4) r.og[u] = snn_sigmoid(pin(acc[12 + u]));
        else {
            const float hv = pin(r.og[u]) * r.tc[u];
            hst[qa][u] = lane_select(vmask, hv, hst[qa][u]);
            if (LAST) hsum[qa][u] += lane_select(vmask, hv, 0.0f);
        }
    };
    // One pipeline phase: the MFMA chain of tile qc (into accc), with the cell update of tile qa (from acca, step ta)
    // issued piecewise behind the MFMAs.  sched_barrier(0) pins the program order: nothing crosses it.
    auto phase = [&](int qc, f32x16& accc, bool do_act, int qa, int ta, const f32x16& acca) {
        const unsigned long long vmask = __builtin_amdgcn_ballot_w64(ta < len[qa]);
        const unsigned int vbits = ta < len[qa] ? 0xffffffffu : 0u;
        ActRegs r;
        bf16x8 bv[NB][3];
        const u32x4* xc = xin + qc * XS + c.lane;
        const u32x4* hc = hb + qc * HS + c.lane;
        auto bload = [&](int b) {
            const u32x4* src = b < XB ? xc + b * BLK_U4 : hc + (b - XB) * BLK_U4;
#pragma unroll
            for (int q = 0; q < 3; ++q) bv[b][q] = __builtin_bit_cast(bf16x8, src[q * 64]);
        };
        bload(0);
        accc = *binit_p;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b + 1 < NB) bload(b + 1);
#pragma unroll
            for (int j = 0; j < 6; ++j) {

                // small terms first: (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi), as snn_forward_b3.hip
                constexpr int WQ[6] = {1, 2, 0, 1, 0, 0}, BQ[6] = {1, 0, 2, 0, 1, 0};
#if defined(SNN_ABL_NOMFMA)     // timing ablations only: one / three products per block instead of six
                if (j == 0)
#elif defined(SNN_ABL_HALFMFMA)
                if (j >= 3)
#endif
                accc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[b][WQ[j]], bv[b][BQ[j]], accc, 0, 0, 0);
                if (do_act) {
                    const int m = b * 6 + j;
                    const int lo = (m * NPIECE + NM - 1) / NM, hi = ((m + 1) * NPIECE + NM - 1) / NM;
#pragma unroll
                    for (int pc = lo; pc < hi; ++pc) act_piece(qa, ta, acca, r, vmask, vbits, pc);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto act_only = [&](int qa, int ta, const f32x16& acca) {
        const unsigned long long vmask = __builtin_amdgcn_ballot_w64(ta < len[qa]);
        const unsigned int vbits = ta < len[qa] ? 0xffffffffu : 0u;
        ActRegs r;
#pragma unroll
        for (int pc = 0; pc < NPIECE; ++pc) act_piece(qa, ta, acca, r, vmask, vbits, pc);
    };

    // ---- prologue: x(t0) of both tiles, h = 0, request x(t1), first chain of tile 0
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
            for (int k = 0; k < 3; ++k)
                reinterpret_cast<uint2*>(hb + q * HS + (hblk * 3 + k) * 64 + hlane)[c.hf] = uint2{0u, 0u};
        const int t1 = T > 1 ? t0 + dt : t0;
        load_step(0, t1, pre0);
        load_step(1, t1, pre1);
    }
    lds_barrier();
    f32x16 acc0, acc1;
    phase(0, acc0, false, 0, 0, acc0);
    lds_barrier();   // every wave has read h0 = 0 before the first cell update overwrites it

    for (int step = 0; step + 1 < T; ++step) {
        const int t = t0 + dt * step;
        const int tnn = step + 2 < T ? t + 2 * dt : t + dt;
        // phase I: chain of tile 1 (step t) || cell update of tile 0 (step t)
        phase(1, acc1, true, 0, t, acc0);
        store_step(0, pre0);          // x0(t + 1)
        load_step(0, tnn, pre0);
        lds_barrier();
        // phase II: chain of tile 0 (step t + 1) || cell update of tile 1 (step t)
        phase(0, acc0, true, 1, t, acc1);
        store_step(1, pre1);
        load_step(1, tnn, pre1);
        lds_barrier();
    }
    {   // last step
        const int t = t0 + dt * (T - 1);
        phase(1, acc1, true, 0, t, acc0);
        lds_barrier();   // tile 1's cell update overwrites the h image the other waves' chains may still be reading
        act_only(1, t, acc1);
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
                    feat[u] = __fdiv_rn(hsum[q][u], (float)(len[q] > 0 ? len[q] : 1));
                    fidx[u] = d * H + j0 + u;
                } else {
                    feat[u] = hst[q][u];
                    fidx[u] = (j0 + u) * (y.L * DIRS) + ld;
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

template <int H, int XB0, int DIRS>
__global__ void __launch_bounds__(H * 8, SNN_B3P_WAVES) lstm_fwd_b3p_kernel(const FwdParams p, const B3Offsets o)
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

    B3PCtx c;
    c.tid = threadIdx.x;
    c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.cl = c.lane & 31;
    c.hf = c.lane >> 5;
    u32x4* ybase = reinterpret_cast<u32x4*>(p.yscratch) + (int64_t)blockIdx.x * p.nbuf * p.y_buf_floats / 4;
    // work queues, one per XCD, with stealing: see snn_forward_b3.hip
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
                        if (last) b3p_phase<H, XB0, DIRS, true, true>(p, c, n_live, tile0, info, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
                        else b3p_phase<H, XB0, DIRS, true, false>(p, c, n_live, tile0, info, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
                    } else {
                        if (last) b3p_phase<H, XB1, DIRS, false, true>(p, c, n_live, tile0, info, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
                        else b3p_phase<H, XB1, DIRS, false, false>(p, c, n_live, tile0, info, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
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
fwd_b3_kernel_t snn_b3p_kernel(int H, int XB0, int dirs)
{
    if (H == 32 && dirs == 2) {
        if (XB0 == 2) return lstm_fwd_b3p_kernel<32, 2, 2>;   // D <= 26 (buckets 24, 26)
        if (XB0 == 3) return lstm_fwd_b3p_kernel<32, 3, 2>;   // D <= 48
        if (XB0 == 5) return lstm_fwd_b3p_kernel<32, 5, 2>;   // D <= 78
    }
    return nullptr;
}
