// snn_forward_b3.hip -- persistent LSTM forward with the gate products on the bf16 matrix pipe, fp32-equivalent:
// every operand (x_t, h_{t-1}, W) is carried as three bf16 pieces hi + mid + lo (= the fp32 value to 2^-27 relative)
// and each fp32 product is rebuilt from the six piece products of order <= 2
//     hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid
// (bf16 x bf16 is exact in fp32; the dropped terms are < 2^-24 relative, below one fp32 rounding), accumulated in
// fp32 by v_mfma_f32_32x32x16_bf16.  Why: on gfx950 the f32 MFMA shares the vector ALU (microbench/
// mfma_valu_overlap.hip) so the LSTM cell's VALU work adds to the matrix time; the bf16 MFMA runs on the separate
// matrix pipe at 16x the rate, so 6 products per block still cost 2.7x less than one f32 MFMA chain AND the cell
// update overlaps them.  Inference only, no masks, H = 32 (SNN_PREC_BF16X3; include/snn_hip.h).
//
// Same decomposition as lstm_fwd_kernel: item = (32-curve tile, MC sample), wave w owns the 4 gates of hidden
// units 8w..8w+7, weights of the running (layer, direction) in registers, B operands from LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "snn_kernels.h"
#include "snn_host.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef SNN_B3_WAVES
#define SNN_B3_WAVES 3
#endif

constexpr int BLK_U4 = 3 * 64;   // uint4 (16-byte) words of one K=16 block: 3 pieces x 64 lanes

// Tiles of one work item: the weights of a (layer, direction) are loaded once and G tiles run through them.
constexpr int B3_G = 4;
struct B3Group {
    int n;             // live tiles of the group (1..B3_G)
    int tile0;         // first tile
    int* info;         // LDS [B3_G][64]: lane cl -> len, 32 + cl -> orig
};

template <int H, int XB, int DIRS, bool FIRST, bool LAST>
__device__ __forceinline__ void b3_phase(const FwdParams& p, ItemCtx& c, const B3Group& grp, int l, int d, int64_t k_A,
                                         int64_t k_bias, int64_t k_head_w, u32x4* xin, u32x4* hb, float* part,
                                         float* logit_all, float* bl, u32x4* ybase_all)
{
    constexpr int NT = H * 8, NW = H / 8, HB = H / 16, NB = XB + HB;
    constexpr int NF4 = XB * BLK_U4, NPT = (NF4 + NT - 1) / NT;   // 16-byte words of one step's x image
    constexpr int YB = DIRS * HB;                                   // blocks of one step of the y image
    const FwdParams& y = p;
    const int ld = l * DIRS + d;

    // ---- weight pieces -> registers: wr[b][piece] = 8 bf16 (4 VGPRs)
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

  for (int q = 0; q < grp.n; ++q) {
    c.tile = grp.tile0 + q;
    c.T = __builtin_amdgcn_readfirstlane(p.tile_T[c.tile]);
    c.len = grp.info[q * 64 + c.cl];
    c.orig = grp.info[q * 64 + 32 + c.cl];
    const int T = c.T;
    float* logit = logit_all + q * y.C * 32;
    u32x4* ybase = ybase_all + q * y_tile_u4;

    // input image of step t: FIRST: the packed tile; else the y image the previous layer wrote
    // (tile_off is in fp32-image floats, T * Dp * 32 per tile; this image holds XB * BLK_U4 16-byte words per step)
    const u32x4* in0 = FIRST ? reinterpret_cast<const u32x4*>(p.xp) + p.tile_off[c.tile] / ((int64_t)p.Dp * SNN_TILE) * NF4
                             : ybase + (int64_t)((l - 1) % p.nbuf) * p.y_buf_floats / 4;
    u32x4* yout = LAST ? nullptr : ybase + (int64_t)(l % p.nbuf) * p.y_buf_floats / 4;
    __syncthreads();   // the previous tile's head reduction has finished with part / logit before this tile's staging

    // branch-free staging: threads beyond the image duplicate its last word (same value to the same address), and
    // the last step re-stages its own image, so the loads issue at the top of the step and nothing is predicated
    auto load_step = [&](int t, u32x4 (&pre)[NPT]) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = c.tid + i * NT < NF4 ? c.tid + i * NT : NF4 - 1;
#ifdef SNN_ABL_NOYLD   // timing ablation only: every step re-reads the first image (L2-resident)
            pre[i] = in0[(int64_t)(FIRST ? t : 0) * NF4 + f];
#else
            pre[i] = in0[(int64_t)t * NF4 + f];
#endif
        }
    };
    auto store_step = [&](u32x4* dst, const u32x4 (&pre)[NPT]) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = c.tid + i * NT < NF4 ? c.tid + i * NT : NF4 - 1;
            dst[f] = pre[i];
        }
    };

    float cst[4] = {0.f, 0.f, 0.f, 0.f}, hst[4] = {0.f, 0.f, 0.f, 0.f}, hsum[4] = {0.f, 0.f, 0.f, 0.f};
    // this lane's 4 hidden units sit in block w >> 1, lane' = (w & 1) * 32 + curve, elements 4 hf .. 4 hf + 3
    const int hblk = c.w >> 1, hlane = (c.w & 1) * 32 + c.cl;
    {
        u32x4 pre[NPT];
        const int t0 = d == 0 ? 0 : T - 1;
        load_step(t0, pre);
        store_step(xin, pre);
#pragma unroll
        for (int q = 0; q < 3; ++q)
            reinterpret_cast<uint2*>(hb + (hblk * 3 + q) * 64 + hlane)[c.hf] = uint2{0u, 0u};
    }
    lds_barrier();

    for (int step = 0; step < T; ++step) {
        const int t = d == 0 ? step : T - 1 - step;
        const int cur = step & 1;
        u32x4 pre[NPT];
        const bool more = step + 1 < T;
        const int tn = more ? (d == 0 ? t + 1 : t - 1) : t;
        load_step(tn, pre);

        f32x16 acc = *binit_p;
        const u32x4* xc = xin + cur * (XB * BLK_U4) + c.lane;
        const u32x4* hc = hb + cur * (HB * BLK_U4) + c.lane;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#ifdef SNN_ABL_NOMFMA   // timing ablation only: one block of products instead of NB
            if (b > 0) break;
#endif
            const u32x4* src = b < XB ? xc + b * BLK_U4 : hc + (b - XB) * BLK_U4;
            const bf16x8 bh = __builtin_bit_cast(bf16x8, src[0]);
            const bf16x8 bm = __builtin_bit_cast(bf16x8, src[64]);
            const bf16x8 bq = __builtin_bit_cast(bf16x8, src[128]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[b][1], bm, acc, 0, 0, 0);   // small terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[b][2], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[b][0], bq, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[b][1], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[b][0], bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[b][0], bh, acc, 0, 0, 0);
        }

        // (opaque v_cndmask selects: a C++ select lets the compiler put each unit's cell update under a branch)
        const unsigned long long vmask = __builtin_amdgcn_ballot_w64(t < c.len);
        const unsigned int vbits = t < c.len ? 0xffffffffu : 0u;
        float outv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float ig = snn_sigmoid(acc[u]);
            const float fg = snn_sigmoid(acc[4 + u]);
            const float gg = snn_tanh(acc[8 + u]);
            const float og = snn_sigmoid(acc[12 + u]);
            const float cn = fmaf(fg, cst[u], __fmul_rn(ig, gg));
            const float hv = og * snn_tanh(cn);
            cst[u] = lane_select(vmask, cn, cst[u]);
            hst[u] = lane_select(vmask, hv, hst[u]);
            outv[u] = lane_select(vmask, hv, 0.0f);
        }
        // h_t as three bf16 pieces -> the other LDS buffer (and, masked by `valid`, the y image of the next layer)
        bf16x4 ph, pm, pl;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __bf16 a, b, q;
            split3(hst[u], a, b, q);
            ph[u] = a; pm[u] = b; pl[u] = q;
        }
        const uint2 wh = __builtin_bit_cast(uint2, ph), wm = __builtin_bit_cast(uint2, pm), wl = __builtin_bit_cast(uint2, pl);
        u32x4* hn = hb + (cur ^ 1) * (HB * BLK_U4);
        reinterpret_cast<uint2*>(hn + (hblk * 3 + 0) * 64 + hlane)[c.hf] = wh;
        reinterpret_cast<uint2*>(hn + (hblk * 3 + 1) * 64 + hlane)[c.hf] = wm;
        reinterpret_cast<uint2*>(hn + (hblk * 3 + 2) * 64 + hlane)[c.hf] = wl;
        if (LAST) {
#pragma unroll
            for (int u = 0; u < 4; ++u) hsum[u] += outv[u];
        } else {
            u32x4* yo = yout + ((int64_t)t * YB + d * HB + hblk) * BLK_U4 + hlane;
#ifdef SNN_ABL_NOYST   // timing ablation only
            if (p.n_curves < 0)
#endif
            {
            reinterpret_cast<uint2*>(yo + 0 * 64)[c.hf] = uint2{wh.x & vbits, wh.y & vbits};
            reinterpret_cast<uint2*>(yo + 1 * 64)[c.hf] = uint2{wm.x & vbits, wm.y & vbits};
            reinterpret_cast<uint2*>(yo + 2 * 64)[c.hf] = uint2{wl.x & vbits, wl.y & vbits};
            }
        }
        store_step(xin + (cur ^ 1) * (XB * BLK_U4), pre);
        lds_barrier();
    }

    // ---- head contribution of this phase (fp32, as lstm_fwd_kernel)
    const bool contributes = y.pool == SNN_POOL_STANDARD || LAST;
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
            for (int w = 0; w < NW; ++w) a += part[(w * y.C + q) * 32 + cc];
            logit[idx] = a;
        }
    }
    __syncthreads();  // also publishes this phase's y stores to the workgroup
  }
}

template <int H, int XB0, int DIRS>
__global__ void __launch_bounds__(H * 8, SNN_B3_WAVES) lstm_fwd_b3_kernel(const FwdParams p, const B3Offsets o)
{
    constexpr int NT = H * 8, NW = H / 8, HB = H / 16;
    constexpr int XB1 = DIRS * HB;
    constexpr int XBM = XB0 > XB1 ? XB0 : XB1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdParams& y = p;
    u32x4* xin = reinterpret_cast<u32x4*>(smem);            // [2][XBM blocks]
    u32x4* hb = xin + 2 * XBM * BLK_U4;                     // [2][HB blocks]
    float* part = reinterpret_cast<float*>(hb + 2 * HB * BLK_U4);   // [NW][C][32]
    float* logit = part + NW * y.C * 32;                    // [B3_G][C][32]
    float* bl = logit + B3_G * y.C * 32;                    // [NW][2][16]
    int* info = reinterpret_cast<int*>(bl + NW * 32);       // [B3_G][64]
    int* item_slot = info + B3_G * 64;

    ItemCtx c;
    c.tid = threadIdx.x;
    c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.cl = c.lane & 31;
    c.hf = c.lane >> 5;
    c.tape = nullptr;
    c.ybase = nullptr;
    u32x4* ybase = reinterpret_cast<u32x4*>(p.yscratch) + (int64_t)blockIdx.x * p.nbuf * p.y_buf_floats / 4;
    // Work queues, one per XCD (workgroups are dispatched round-robin over the 8 XCDs, each with its own L2):
    // queue x holds the tile groups x, x + 8, ... with all MC samples of a group adjacent, so the x image of a
    // group is fetched into ONE L2 and shared by its S items.  An XCD that runs dry steals from the next queue.
    const int n_groups = (p.n_tiles + B3_G - 1) / B3_G;
    const int xcd = blockIdx.x & 7;
    B3Group grp;
    grp.info = info;

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
            grp.tile0 = g * B3_G;
            grp.n = min(B3_G, p.n_tiles - grp.tile0);
            c.wks = p.wk + (int64_t)c.s * p.wk_stride;
            for (int idx = c.tid; idx < grp.n * 64; idx += NT) {
                const int q = idx >> 6, r = idx & 63;
                const int pos = (grp.tile0 + q) * SNN_TILE + (r & 31);
                info[idx] = r < 32 ? p.len_sorted[pos] : p.order[pos];
            }
            for (int idx = c.tid; idx < B3_G * y.C * 32; idx += NT) logit[idx] = 0.0f;
            __syncthreads();

            for (int l = 0; l < y.L; ++l) {
                const bool last = l == y.L - 1;
                for (int d = 0; d < DIRS; ++d) {
                    const int ld = l * DIRS + d;
                    if (l == 0) {
                        if (last) b3_phase<H, XB0, DIRS, true, true>(p, c, grp, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
                        else b3_phase<H, XB0, DIRS, true, false>(p, c, grp, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
                    } else {
                        if (last) b3_phase<H, XB1, DIRS, false, true>(p, c, grp, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
                        else b3_phase<H, XB1, DIRS, false, false>(p, c, grp, l, d, o.k_A[ld], o.k_bias[ld], o.k_head_w, xin, hb, part, logit, bl, ybase);
                    }
                }
            }

            // ---- head bias + softmax, one thread per curve of the group
            for (int idx = c.tid; idx < grp.n * 32; idx += NT) {
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

// ---- registry hooks used by snn_forward.hip -----------------------------------------------------------
typedef void (*fwd_b3_kernel_t)(const FwdParams, const B3Offsets);
fwd_b3_kernel_t snn_b3_kernel(int H, int XB0, int dirs)
{
    if (H == 32 && dirs == 2) {
        if (XB0 == 2) return lstm_fwd_b3_kernel<32, 2, 2>;   // D <= 26 (buckets 24, 26)
        if (XB0 == 3) return lstm_fwd_b3_kernel<32, 3, 2>;   // D <= 48
        if (XB0 == 5) return lstm_fwd_b3_kernel<32, 5, 2>;   // D <= 78
    }
    return nullptr;
}
