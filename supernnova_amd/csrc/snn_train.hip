// snn_train.hip -- gfx950 kernels + C ABI for the training half of the hot path:
//   lstm_bwd_kernel     BPTT through the LSTM stack + pooling + head, per 32-curve tile
//   grad_reduce_kernel  fixed-order sum of the per-workgroup gradient slabs
//   bbb_backward_kernel reparameterisation + KL gradients (dmu, drho)
//   adam_kernel         fused Adam over one flat buffer
// Replaces autograd of `loss.backward()` and `optimizer.step()`
// (utils/training_utils.py:523-524) for the three model classes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>

#include "snn_kernels.h"
#include "snn_host.h"

#define HIP_CHECK(expr)                                                             \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            snn_set_error("%s failed: %s", #expr, hipGetErrorString(e_));           \
            return SNN_ERR_HIP;                                                     \
        }                                                                           \
    } while (0)

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

template <int H, int DIRS>
struct BwdLds {
    static constexpr int NW = H / 8;
    static constexpr int XROWS = DIRS * H > 32 ? DIRS * H : 32;  // >= 2*KKI0 (D <= 32) and DIRS*H
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
template <int H, int KKI, int DIRS, bool FIRST>
__device__ __forceinline__ void bwd_phase(const BwdParams& p, float* smem, int tile, int T, int l, int d, int w,
                                          int lane, int tid, int len, int orig, const float* tape,
                                          float* slab, float* dybuf_in, float* dybuf_out, const float* dfeat)
{
    using LD = BwdLds<H, DIRS>;
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
    using LD = BwdLds<H, DIRS>;
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
                if (l == 0) bwd_phase<H, KKI0, DIRS, true>(p, smem, tile, T, l, d, w, lane, tid, len, orig, tape, slab, dy_in, dy_out, dfeat);
                else bwd_phase<H, KKI1, DIRS, false>(p, smem, tile, T, l, d, w, lane, tid, len, orig, tape, slab, dy_in, dy_out, dfeat);
            }
        }
    }
}

// out[i] = sum over slabs in fixed order (deterministic)
__global__ void __launch_bounds__(256) grad_reduce_kernel(int64_t P, int n_slabs, const float* __restrict__ slabs,
                                                         float* __restrict__ out, int accumulate)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float s = accumulate ? out[i] : 0.0f;
    for (int k = 0; k < n_slabs; ++k) s += slabs[(int64_t)k * P + i];
    out[i] = s;
}

// ---- backward kernel registry --------------------------------------------------
typedef void (*bwd_kernel_t)(const BwdParams);
struct BwdEntry { int H, KKI0, dirs; bwd_kernel_t fn; };
#define BWD_ENTRY(H, K, Dd) {H, K, Dd, lstm_bwd_kernel<H, K, Dd>}
static const BwdEntry g_bwd[] = {
    BWD_ENTRY(32, 12, 2), BWD_ENTRY(32, 13, 2), BWD_ENTRY(32, 12, 1), BWD_ENTRY(64, 12, 2), BWD_ENTRY(64, 13, 2),
};
static const BwdEntry* find_bwd(const SnnLayout& y)
{
    for (const BwdEntry& e : g_bwd)
        if (e.H == y.H && e.KKI0 == y.KKI[0] && e.dirs == y.dirs) return &e;
    return nullptr;
}

static const int BWD_MAX_GRID = 256;

static size_t bwd_lds_bytes(const SnnLayout& y)
{
    const int xrows = std::max(y.dirs * y.H, 32), krows = y.H + xrows, mt = (krows + 31) / 32;
    const size_t fixed = (size_t)2 * krows * LDS_PAD + 4 * y.H * LDS_PAD + (size_t)y.NW * mt * 1024 + y.H * 32 + 64;
    return sizeof(float) * (fixed + (size_t)(y.C + y.F) * 32);
}

extern "C" int64_t snn_lstm_backward_workspace_bytes(const snn_cfg* cfg, int32_t n_tiles, int32_t max_T)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if (n_tiles < 1 || max_T < 1) { snn_set_error("snn_lstm_backward_workspace_bytes: bad sizes"); return SNN_ERR_ARG; }
    const int grid = std::min(n_tiles, BWD_MAX_GRID);
    return (int64_t)sizeof(float) * grid * (y.P + (int64_t)2 * y.dirs * y.dirs * max_T * y.H * 32);
}

extern "C" int snn_lstm_backward(const snn_cfg* cfg, const float* xp, const int32_t* len_sorted, const int32_t* order,
                                 const int32_t* tile_T, const int64_t* tile_off, int32_t n_curves, int32_t n_tiles,
                                 int32_t max_T, const float* wk, const snn_masks* masks, const float* tape,
                                 int64_t tape_floats, const float* dlogits, float* dflat, int32_t accumulate,
                                 void* workspace, int64_t workspace_bytes, void* stream)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    const BwdEntry* e = find_bwd(y);
    if (!e) { snn_set_error("no backward kernel for H=%d D=%d dirs=%d", y.H, y.D, y.dirs); return SNN_ERR_UNSUPPORTED; }
    if (!xp || !len_sorted || !order || !tile_T || !tile_off || !wk || !tape || !dlogits || !dflat || !workspace) {
        snn_set_error("snn_lstm_backward: null pointer");
        return SNN_ERR_ARG;
    }
    if (n_curves < 1 || n_tiles != (n_curves + SNN_TILE - 1) / SNN_TILE || max_T < 1) {
        snn_set_error("snn_lstm_backward: inconsistent sizes");
        return SNN_ERR_ARG;
    }
    const int64_t tape_tile = tape_tile_floats(y.H, y.L * y.dirs, max_T, y.F);
    if (tape_floats < tape_tile * n_tiles) { snn_set_error("snn_lstm_backward: tape too small"); return SNN_ERR_ARG; }
    const int grid = std::min(n_tiles, BWD_MAX_GRID);
    const int64_t need = (int64_t)sizeof(float) * grid * (y.P + (int64_t)2 * y.dirs * y.dirs * max_T * y.H * 32);
    if (workspace_bytes < need) {
        snn_set_error("snn_lstm_backward: workspace %lld < required %lld bytes", (long long)workspace_bytes, (long long)need);
        return SNN_ERR_WORKSPACE;
    }
    const size_t lds = bwd_lds_bytes(y);
    if (lds > 160 * 1024) { snn_set_error("snn_lstm_backward: LDS request %zu exceeds 160 KiB", lds); return SNN_ERR_UNSUPPORTED; }
    BwdParams p;
    memset(&p, 0, sizeof(p));
    p.lay = y;
    p.xp = xp; p.len_sorted = len_sorted; p.order = order; p.tile_T = tile_T; p.tile_off = tile_off;
    p.n_curves = n_curves; p.n_tiles = n_tiles; p.max_T = max_T; p.wk = wk;
    rc = snn_masks_to_device(masks, max_T, &p.mk);
    if (rc) return rc;
    p.tape = tape; p.tape_tile = tape_tile; p.dlogits = dlogits;
    p.slabs = (float*)workspace;
    p.dyscratch = p.slabs + (int64_t)grid * y.P;
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipMemsetAsync(p.slabs, 0, sizeof(float) * (size_t)grid * y.P, st));
    HIP_CHECK(hipFuncSetAttribute((const void*)e->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(e->fn, dim3(grid), dim3(y.H * 8), lds, st, p);
    HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((y.P + 255) / 256)), dim3(256), 0, st, y.P, grid, p.slabs, dflat,
                       accumulate);
    HIP_CHECK(hipGetLastError());
    return SNN_OK;
}

// ============================================================================
// Bayes-by-Backprop backward: reparameterisation + KL gradients
// ============================================================================
// loss = CE + kl_scale * KL,  w = mu + eps*sigma, sigma = softplus(rho) + 1e-5.
//   dKL/dw      = -(w-mu)/sigma^2 + w (r1/s1^2 + r2/s2^2)      r_k: per-tensor mixture responsibilities
//   dmu         = dCE/dw + kl_scale * [dKL/dw + (w-mu)/sigma^2]
//   dsigma      = (dCE/dw + kl_scale*dKL/dw) * eps + kl_scale * [(w-mu)^2/sigma^3 - 1/sigma]
//   drho        = dsigma * sigmoid(rho)
// which is what autograd derives for bayesian_rnn.py:472-528, 595-616, 794-815.
__global__ void __launch_bounds__(256) bbb_backward_kernel(SnnLayout y, const float* __restrict__ mu,
                                                          const float* __restrict__ rho,
                                                          const float* __restrict__ eps, int mode, uint64_t seed,
                                                          uint64_t offset, snn_prior prior,
                                                          const float* __restrict__ kl_terms, float kl_scale_host,
                                                          const float* __restrict__ kl_scale_dev,
                                                          const float* __restrict__ dw, float* __restrict__ dmu,
                                                          float* __restrict__ drho, int accumulate)
{
    const int64_t Ps = y.P - y.C;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= y.P) return;
    const float kl_scale = kl_scale_dev ? *kl_scale_dev : kl_scale_host;
    if (p >= Ps) {  // deterministic head bias
        dmu[p] = (accumulate ? dmu[p] : 0.0f) + dw[p];
        return;
    }
    // which tensor?
    int tau = 0;
    {
        const int nld = y.L * y.dirs;
        if (p >= y.c_head_w) tau = 4 * nld;
        else {
            int ld = 0;
            while (ld + 1 < nld && p >= y.c_wih[ld + 1]) ++ld;
            tau = 4 * ld + (p >= y.c_bhh[ld] ? 3 : p >= y.c_bih[ld] ? 2 : p >= y.c_whh[ld] ? 1 : 0);
        }
    }
    const bool head = tau == 4 * y.L * y.dirs;
    const float s1 = head ? prior.sigma1_out : prior.sigma1, s2 = head ? prior.sigma2_out : prior.sigma2;
    const float r1 = kl_terms[tau * 3 + 1], r2 = kl_terms[tau * 3 + 2];
    const float m = mu[p];
    const float rh = rho[p];
    const float sg = (rh > 20.0f ? rh : log1pf(expf(rh))) + 1e-5f;
    float e = 0.0f, w = m;
    if (mode != 1) {
        e = mode == 0 ? eps[p] : philox_normal(seed, offset, SNN_STREAM_BBB, (uint64_t)p);
        w = __fadd_rn(m, __fmul_rn(e, sg));
    }
    const float d = w - m;
    const float inv_var = 1.0f / (sg * sg);
    const float dkl_dw = -d * inv_var + w * (r1 / (s1 * s1) + r2 / (s2 * s2));
    const float gw = dw[p] + kl_scale * dkl_dw;
    const float gmu = gw + kl_scale * d * inv_var;
    const float gsig = gw * e + kl_scale * (d * d * inv_var / sg - 1.0f / sg);
    const float sgm = 1.0f / (1.0f + expf(-rh));
    dmu[p] = (accumulate ? dmu[p] : 0.0f) + gmu;
    drho[p] = (accumulate ? drho[p] : 0.0f) + gsig * sgm;
}

extern "C" int snn_bbb_backward(const snn_cfg* cfg, const float* mu, const float* rho, const float* eps, int32_t mode,
                                uint64_t seed, uint64_t offset, const snn_prior* prior, const float* kl_terms,
                                float kl_scale, const float* kl_scale_dev, const float* dw, float* dmu, float* drho,
                                int32_t accumulate, void* stream)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if (!mu || !rho || !prior || !kl_terms || !dw || !dmu || !drho || mode < 0 || mode > 2 || (mode == 0 && !eps)) {
        snn_set_error("snn_bbb_backward: null pointer or bad mode");
        return SNN_ERR_ARG;
    }
    hipLaunchKernelGGL(bbb_backward_kernel, dim3((unsigned)((y.P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, mu,
                       rho, eps, mode, seed, offset, *prior, kl_terms, kl_scale, kl_scale_dev, dw, dmu, drho, accumulate);
    HIP_CHECK(hipGetLastError());
    return SNN_OK;
}

// ============================================================================
// Cross-entropy (mean over the batch) and its gradient w.r.t. the logits
// (nn.CrossEntropyLoss() at utils/training_utils.py:515): one workgroup, fixed-order sum.
// ============================================================================
__global__ void __launch_bounds__(256) ce_loss_grad_kernel(int N, int C, const float* __restrict__ logits,
                                                          const int64_t* __restrict__ target, float scale,
                                                          float* __restrict__ loss, float* __restrict__ dlogits)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* lg = logits + (int64_t)n * C;
        float mx = -INFINITY;
        for (int q = 0; q < C; ++q) mx = fmaxf(mx, lg[q]);
        float z = 0.0f;
        for (int q = 0; q < C; ++q) z += expf(lg[q] - mx);
        const float lz = logf(z);
        const int tq = (int)target[n];
        acc += (double)(-(lg[tq] - mx - lz));
        for (int q = 0; q < C; ++q)
            dlogits[(int64_t)n * C + q] = (expf(lg[q] - mx - lz) - (q == tq ? 1.0f : 0.0f)) * scale;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (float)((sh[0] + sh[1] + sh[2] + sh[3]) / (double)N);
}

extern "C" int snn_ce_loss_grad(int32_t n, int32_t n_classes, const float* logits, const int64_t* target,
                                float grad_scale, float* loss, float* dlogits, void* stream)
{
    if (!logits || !target || !loss || !dlogits || n < 1 || n_classes < 1) {
        snn_set_error("snn_ce_loss_grad: null pointer or bad size");
        return SNN_ERR_ARG;
    }
    hipLaunchKernelGGL(ce_loss_grad_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n, n_classes, logits, target,
                       grad_scale, loss, dlogits);
    HIP_CHECK(hipGetLastError());
    return SNN_OK;
}

// ============================================================================
// Fused Adam over one flat buffer (torch.optim.Adam semantics, utils/training_utils.py:365-369)
// ============================================================================
__global__ void __launch_bounds__(256) adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ m, float* __restrict__ v, float lr, float beta1,
                                                  float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt,
                                                  float grad_scale)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * grad_scale;
    const float pi = p[i];
    if (weight_decay != 0.0f) gi = fmaf(weight_decay, pi, gi);
    const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);          // lerp, as torch
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

extern "C" int snn_adam_step(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                             int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                             float grad_scale, void* stream)
{
    if (!params || !grads || !exp_avg || !exp_avg_sq || n < 1 || step < 1) {
        snn_set_error("snn_adam_step: null pointer, n < 1 or step < 1");
        return SNN_ERR_ARG;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, params, grads,
                       exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    HIP_CHECK(hipGetLastError());
    return SNN_OK;
}
