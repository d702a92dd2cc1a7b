// snn_bwd_small.h -- BPTT kernel for hidden sizes 32 and 64 (template + instantiation lists); included by the
// snn_bwd_h<H>.hip translation units.  See snn_train.hip for the host side.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snn_kernels.h"

struct BwdParams {
    SnnLayout lay;
    const float* xp;
    const int32_t* len_sorted;
    const int32_t* order;
    const int32_t* tile_T;
    const int64_t* tile_off;
    int n_curves, n_tiles, max_T;
    const float* wk;       // kernel-layout weights used by the forward (one set)
    MaskP mk;
    const float* tape;
    int64_t tape_tile;
    const float* dlogits;  // (N, C), original curve order
    float* slabs;          // [grid][P] canonical flat gradients (zeroed by the host)
    float* dyscratch;      // [grid][2][dirs][max_T][H*32]
};

constexpr int LDS_PAD = 33;  // row stride of the transposed-read LDS images (conflict-free column reads)

template <int H, int KKI0, int DIRS>
struct BwdLds {
    static constexpr int NW = H / 8;
    static constexpr int XROWS = DIRS * H > 2 * KKI0 ? DIRS * H : 2 * KKI0;  // x rows of the widest layer
    static constexpr int KROWS = H + XROWS;                       // [h | x] rows staged per step
    static constexpr int MT_MAX = (KROWS + 31) / 32;
    static constexpr int XH = 0;                                  // [2][KROWS][33]
    static constexpr int DG = XH + 2 * KROWS * LDS_PAD;           // [4H][33]
    static constexpr int RED = DG + 4 * H * LDS_PAD;              // [NW][MT_MAX][1024]
    static constexpr int DHREC = RED + NW * MT_MAX * 1024;        // [H][32]
    static constexpr int LENS = DHREC + H * 32;                   // [32] ints + [32] origs
    static constexpr int FIXED = LENS + 64;                       // then dlg[C][32], dfeat[F][32]
};

// One (layer, direction) backward pass over the tile.
//   FIRST: layer 0 (x from the packed tile, no dx needed)
template <int H, int KKI0, int KKI, int DIRS, bool FIRST>
__device__ __forceinline__ void bwd_phase(const BwdParams& p, float* smem, int tile, int T, int l, int d, int w,
                                          int lane, int tid, int len, int orig, const float* tape,
                                          float* slab, float* dybuf_in, float* dybuf_out, const float* dfeat)
{
    using LD = BwdLds<H, KKI0, DIRS>;
    constexpr int NT = H * 8, NW = H / 8;
    constexpr int XR = 2 * KKI;              // x rows staged (incl. even padding)
    constexpr int KR = H + XR;               // rows of [h | x]
    constexpr int MT_H = H / 32;             // tiles of dxh that are dh
    constexpr int MT_X = FIRST ? 0 : (XR + 31) / 32;
    constexpr int MT = MT_H + MT_X;          // tiles of W^T dgates computed
    constexpr int MT_W = (KR + 31) / 32;     // tiles of dW^T accumulated
    constexpr int KKH = H / 2, KK = KKI + KKH;
    constexpr int NF4 = KR * 8;              // float4 loads to stage one step
    constexpr int NPT = (NF4 + NT - 1) / NT;
    const SnnLayout& y = p.lay;
    const MaskP& mk = p.mk;
    const bool masked = mk.mode != SNN_MASK_NONE;
    const int ld = l * DIRS + d;
    const bool last_layer = l == y.L - 1;
    const int cl = lane & 31, hf = lane >> 5;
    const int j0 = 8 * w + 4 * hf;
    const int I = y.I[l];

    float* xh = smem + LD::XH;
    float* dG = smem + LD::DG;
    float* red = smem + LD::RED;
    float* dhrec = smem + LD::DHREC;
    const int* lens = reinterpret_cast<const int*>(smem + LD::LENS);
    const int* origs = lens + 32;

    // ---- W^T operand registers, gathered from the forward (kernel-layout) image:
    //      abw[mt][i] = Wcat[row(w, r_i)][col], r_i = (i&3)+8(i>>2)+4hf, col = 32mt + (lane&31) in [h | x] order
    float abw[MT][16];
    {
        const float* A = p.wk + y.k_A[ld] + (int64_t)w * KK * 64;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int col = 32 * mt + cl;
            const int k = col < H ? 2 * KKI + col : col - H;   // index in the forward's [x | h] order
            const bool live = col < KR;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = (i & 3) + 8 * (i >> 2) + 4 * hf;
                abw[mt][i] = live ? A[(k >> 1) * 64 + (k & 1) * 32 + rr] : 0.0f;
            }
        }
    }
    // variational W_hh row mask (forward direction only): effective W_hh = m (.) raw
    float whm_lane = 1.0f;  // mask of gate row R(w, lane&31): scales dW_hh at write-out
    if (d == 0 && masked) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = (i & 3) + 8 * (i >> 2) + 4 * hf;
            const float m = mask_whh(mk, 0, y.L, H, l, (rr >> 3) * H + 8 * w + (rr & 7));
#pragma unroll
            for (int mt = 0; mt < MT_H; ++mt) abw[mt][i] = __fmul_rn(m, abw[mt][i]);
        }
        whm_lane = mask_whh(mk, 0, y.L, H, l, (cl >> 3) * H + 8 * w + (cl & 7));
    }

    f32x16 dw[MT_W];
#pragma unroll
    for (int mt = 0; mt < MT_W; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dw[mt][i] = 0.0f;
    float db[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) db[i] = 0.0f;
    float dc[4] = {0.f, 0.f, 0.f, 0.f};

    const float* tape_ld = tape + (int64_t)ld * p.max_T * tape_step_floats(H);
    const float* xtile = p.xp + p.tile_off[tile];
    const int64_t dy_src = (int64_t)DIRS * p.max_T * H * 32;   // floats of one producing direction's dy block
    const float* dy_in = last_layer ? nullptr : dybuf_in + (int64_t)d * p.max_T * H * 32;

    // staging of [h_prev | x_t] for processing step `step` into buffer `buf` (transposed-read image, stride 33)
    auto stage_load = [&](int step, f32x4 (&pre)[NPT]) {
        const int t = d == 0 ? step : T - 1 - step;
        const int tp = d == 0 ? t - 1 : t + 1;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = tid + i * NT;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f < NF4) {
                const int row = f >> 3, c4 = (f & 7) * 4;
                if (row < H) {
                    if (step > 0) v = *reinterpret_cast<const f32x4*>(tape_ld + (int64_t)tp * tape_step_floats(H) + 5 * H * 32 + row * 32 + c4);
                } else {
                    const int k = row - H;
                    if (FIRST) {
                        v = *reinterpret_cast<const f32x4*>(xtile + ((int64_t)t * XR + k) * 32 + c4);
                        if (masked && k < y.D) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (origs[c4 + e] >= 0) v[e] = __fmul_rn(mask_in(mk, 0, p.n_curves, y.D, origs[c4 + e], k), v[e]);
                        }
                    } else {
                        const int dd = k / H, kk = k - dd * H;
                        v = *reinterpret_cast<const f32x4*>(tape + ((int64_t)((l - 1) * DIRS + dd) * p.max_T + t) * tape_step_floats(H) + 5 * H * 32 + kk * 32 + c4);
                        if (masked) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (origs[c4 + e] >= 0)
                                    v[e] = __fmul_rn(mask_mid(mk, 0, y.L, p.n_curves, DIRS * H, l - 1, t, origs[c4 + e], k), v[e]);
                        }
                    }
                }
            }
            pre[i] = v;
        }
    };
    auto stage_store = [&](float* buf, const f32x4 (&pre)[NPT]) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = tid + i * NT;
            if (f < NF4) {
                const int row = f >> 3, c4 = (f & 7) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) buf[row * LDS_PAD + c4 + e] = pre[i][e];
            }
        }
    };

    // ---- phase init: recurrent gradient entering at the last processed step
    for (int idx = tid; idx < H * 32; idx += NT) {
        const int j = idx >> 5, c = idx & 31;
        dhrec[idx] = (y.pool == SNN_POOL_STANDARD) ? dfeat[(j * (y.L * DIRS) + ld) * 32 + c] : 0.0f;
    }
    {
        f32x4 pre[NPT];
        stage_load(T - 1, pre);
        stage_store(xh, pre);
    }
    __syncthreads();

    const float inv_len = 1.0f / (float)(len > 0 ? len : 1);
    // tape values of one step for this lane's 4 units: i, f, g, o, c (5 x 4) and the gradient arriving from
    // the layer above (both of its directions); loaded one step ahead so the L2/HBM latency hides behind the
    // MFMAs of the running step (the barriers inside the loop are LDS-only and do not drain vector memory)
    struct StepVals { float q[5][4]; float dy[4]; };
    auto load_vals = [&](int step, StepVals& v) {
        const int t = d == 0 ? step : T - 1 - step;
        const float* tq = tape_ld + (int64_t)t * tape_step_floats(H) + j0 * 32 + cl;
#pragma unroll
        for (int g = 0; g < 5; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) v.q[g][u] = tq[(g * H + u) * 32];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v.dy[u] = 0.0f;
            if (!last_layer) {
                const int64_t o = (int64_t)t * H * 32 + (j0 + u) * 32 + cl;
                v.dy[u] = dy_in[o];
                if (DIRS == 2) v.dy[u] += dy_in[dy_src + o];
            }
        }
    };
    StepVals curv, nxtv;
    load_vals(T - 1, curv);
    for (int sb = 0; sb < T; ++sb) {
        const int step = T - 1 - sb;
        const int t = d == 0 ? step : T - 1 - step;
        const int tp = d == 0 ? t - 1 : t + 1;
        const int cur = sb & 1;
        const bool valid = t < len;
        const bool prev_live = step > 0 && tp < len;
        f32x4 pre[NPT];
        if (step > 0) {
            load_vals(step - 1, nxtv);
            stage_load(step - 1, pre);
        }

        // ---- dgates of this lane's 4 units (C/D register order i = gate*4 + u)
        float dga[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float dh = dhrec[(j0 + u) * 32 + cl] + curv.dy[u];
            if (last_layer && y.pool == SNN_POOL_MEAN && valid) dh += dfeat[(d * H + j0 + u) * 32 + cl] * inv_len;
            float dai = 0.f, daf = 0.f, dag = 0.f, dao = 0.f;
            if (valid) {
                const float ig = curv.q[0][u], fg = curv.q[1][u], gg = curv.q[2][u], og = curv.q[3][u], cn = curv.q[4][u];
                const float cp = prev_live ? nxtv.q[4][u] : 0.0f;   // c of the previous processed step
                const float tc = snn_tanh(cn);
                const float dct = dc[u] + dh * og * (1.0f - tc * tc);
                dao = dh * tc * og * (1.0f - og);
                dai = dct * gg * ig * (1.0f - ig);
                daf = dct * cp * fg * (1.0f - fg);
                dag = dct * ig * (1.0f - gg * gg);
                dc[u] = dct * fg;
            }
            dga[0 + u] = dai; dga[4 + u] = daf; dga[8 + u] = dag; dga[12 + u] = dao;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            db[i] += dga[i];
            dG[(w * 32 + (i & 3) + 8 * (i >> 2) + 4 * hf) * LDS_PAD + cl] = dga[i];
        }
        // ---- W^T dgates: this wave's 32 gate rows contribute a partial [KR x 32 curves]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(abw[mt][i], dga[i], acc, 0, 0, 0);
            float* r = red + (w * LD::MT_MAX + mt) * 1024 + cl;
#pragma unroll
            for (int i = 0; i < 16; ++i) r[((i & 3) + 8 * (i >> 2) + 4 * hf) * 32] = acc[i];
        }
        lds_barrier();  // A: dG and red complete

        // ---- dW^T += [h_prev | x_t] (KR x 32 curves) . dgates^T (32 curves x this wave's 32 gate rows)
        {
            const float* xb = xh + cur * LD::KROWS * LDS_PAD;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float b = dG[(w * 32 + cl) * LDS_PAD + 2 * kk + hf];
#pragma unroll
                for (int mt = 0; mt < MT_W; ++mt) {
                    const int row = 32 * mt + cl;
                    const float a = row < KR ? xb[row * LDS_PAD + 2 * kk + hf] : 0.0f;
                    dw[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, dw[mt], 0, 0, 0);
                }
            }
        }
        // ---- reduce the partials over waves: dh -> dhrec (next step), dx -> dy of the layer below
        for (int e = tid; e < MT * 1024; e += NT) {
            const int mt = e >> 10, m = (e >> 5) & 31, c = e & 31;
            float s = 0.0f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) s += red[(ww * LD::MT_MAX + mt) * 1024 + m * 32 + c];
            if (mt < MT_H) {
                const int j = 32 * mt + m;
                if (t < lens[c]) dhrec[j * 32 + c] = s;   // beyond the curve the state is frozen: gradient passes through
            } else if (!FIRST) {
                const int f = 32 * (mt - MT_H) + m;
                if (f < DIRS * H) {
                    if (masked && origs[c] >= 0) s = __fmul_rn(mask_mid(mk, 0, y.L, p.n_curves, DIRS * H, l - 1, t, origs[c], f), s);
                    const int dd = f / H, jj = f - dd * H;
                    // one buffer per producing direction (summed by the consumer): no read-modify-write
                    dybuf_out[(int64_t)d * dy_src + ((int64_t)dd * p.max_T + t) * H * 32 + jj * 32 + c] = s;
                }
            }
        }
        if (step > 0) stage_store(xh + (cur ^ 1) * LD::KROWS * LDS_PAD, pre);
        curv = nxtv;
        lds_barrier();  // B
    }

    // ---- write-out into this workgroup's private slab (canonical layout)
    {
        const int R = (cl >> 3) * H + 8 * w + (cl & 7);  // gate row of this lane's column
#pragma unroll
        for (int mt = 0; mt < MT_W; ++mt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kr = 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * hf;
                if (kr < H) slab[y.c_whh[ld] + (int64_t)R * H + kr] += whm_lane * dw[mt][i];
                else if (kr - H < I) slab[y.c_wih[ld] + (int64_t)R * I + (kr - H)] += dw[mt][i];
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = db[i];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (cl == 0) {
                const int rr = (i & 3) + 8 * (i >> 2) + 4 * hf;
                const int Rb = (rr >> 3) * H + 8 * w + (rr & 7);
                slab[y.c_bih[ld] + Rb] += v;
                slab[y.c_bhh[ld] + Rb] += v;
            }
        }
    }
    __syncthreads();
}

template <int H, int KKI0, int DIRS>
__global__ void __launch_bounds__(H * 8) lstm_bwd_kernel(const BwdParams p)
{
    using LD = BwdLds<H, KKI0, DIRS>;
    constexpr int NT = H * 8;
    constexpr int KKI1 = DIRS * H / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SnnLayout& y = p.lay;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = tid & 63, cl = lane & 31;
    int* lens = reinterpret_cast<int*>(smem + LD::LENS);
    int* origs = lens + 32;
    float* dlg = smem + LD::FIXED;        // [C][32]
    float* dfeat = dlg + y.C * 32;        // [F][32]
    float* slab = p.slabs + (int64_t)blockIdx.x * y.P;
    const int64_t dy_buf = (int64_t)DIRS * DIRS * p.max_T * H * 32;   // [producer dir][dir][max_T][H*32]
    float* dyb = p.dyscratch + (int64_t)blockIdx.x * 2 * dy_buf;
    const float* Wo = p.wk + y.k_head_w;

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int T = __builtin_amdgcn_readfirstlane(p.tile_T[tile]);
        const int pos = tile * SNN_TILE + cl;
        const int len = p.len_sorted[pos], orig = p.order[pos];
        const float* tape = p.tape + (int64_t)tile * p.tape_tile;
        const float* feat = tape + (int64_t)y.L * DIRS * p.max_T * tape_step_floats(H);
        if (tid < 32) { lens[tid] = len; origs[tid] = orig; }
        for (int idx = tid; idx < y.C * 32; idx += NT) {
            const int q = idx >> 5, c = idx & 31;
            const int o = p.order[tile * SNN_TILE + c];
            dlg[idx] = o >= 0 ? p.dlogits[(int64_t)o * y.C + q] : 0.0f;
        }
        __syncthreads();
        // ---- head backward: dW_out, db_out, dfeat (BayesLinear / nn.Linear + output dropout)
        for (int idx = tid; idx < y.F * 32; idx += NT) {
            const int f = idx >> 5, c = idx & 31;
            float s = 0.0f;
            for (int q = 0; q < y.C; ++q) s = fmaf(Wo[q * y.F + f], dlg[q * 32 + c], s);
            if (p.mk.mode != SNN_MASK_NONE && origs[c] >= 0) s = __fmul_rn(mask_out(p.mk, 0, p.n_curves, y.F, origs[c], f), s);
            dfeat[idx] = s;
        }
        for (int idx = tid; idx < y.C * y.F; idx += NT) {
            const int q = idx / y.F, f = idx - q * y.F;
            float s = 0.0f;
            for (int c = 0; c < 32; ++c) s = fmaf(dlg[q * 32 + c], feat[f * 32 + c], s);
            slab[y.c_head_w + idx] += s;
        }
        if (tid < y.C) {
            float s = 0.0f;
            for (int c = 0; c < 32; ++c) s += dlg[tid * 32 + c];
            slab[y.c_head_b + tid] += s;
        }
        __syncthreads();

        for (int l = y.L - 1; l >= 0; --l) {
            float* dy_in = dyb + (int64_t)((l + 1) & 1) * dy_buf;
            float* dy_out = dyb + (int64_t)(l & 1) * dy_buf;
            for (int d = 0; d < DIRS; ++d) {
                if (l == 0) bwd_phase<H, KKI0, KKI0, DIRS, true>(p, smem, tile, T, l, d, w, lane, tid, len, orig, tape, slab, dy_in, dy_out, dfeat);
                else bwd_phase<H, KKI0, KKI1, DIRS, false>(p, smem, tile, T, l, d, w, lane, tid, len, orig, tape, slab, dy_in, dy_out, dfeat);
            }
        }
    }
}


// ---- instantiation lists ------------------------------------------------------------------------------
typedef void (*bwd_kernel_t)(const BwdParams);
struct BwdEntry { int H, KKI0, dirs; bwd_kernel_t fn; };
#define SNN_BWD_ENTRY(H, K, Dd) {H, K, Dd, lstm_bwd_kernel<H, K, Dd>}
#define SNN_BWD_ENTRIES(H)                                                                              \
    SNN_BWD_ENTRY(H, 12, 2), SNN_BWD_ENTRY(H, 13, 2), SNN_BWD_ENTRY(H, 24, 2), SNN_BWD_ENTRY(H, 38, 2), \
    SNN_BWD_ENTRY(H, 39, 2), SNN_BWD_ENTRY(H, 12, 1), SNN_BWD_ENTRY(H, 13, 1), SNN_BWD_ENTRY(H, 24, 1), \
    SNN_BWD_ENTRY(H, 38, 1), SNN_BWD_ENTRY(H, 39, 1)
