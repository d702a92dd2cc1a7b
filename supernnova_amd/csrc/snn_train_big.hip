// snn_train_big.hip -- BPTT for hidden sizes 64 and 128, where the fused kernel of snn_bwd_small.h does not
// fit (its per-wave register image of [W_hh | W_ih]^T plus the dW accumulators needs > 512 VGPRs, its LDS
// image > 160 KB).  The work is split the way the forward's two-pass kernel splits it -- only the recurrent
// product stays inside the sequential loop, everything that is a plain GEMM over (curve, step) is deferred:
//
//   bwd_prep_kernel   W_hh^T / W_ih^T as MFMA A-operand images, gathered once from the forward's weight image
//   head_*_kernel     output layer backward: dfeat per tile, dW_out / db_out
//   per layer, top down:
//     bwd_rec_kernel  one workgroup per (tile, direction): gate gradients from the tape, dh_{t-1} = W_hh^T dG_t
//                     with W_hh^T resident in registers (256 VGPRs at H = 128); dG_t streamed to HBM
//     bwd_dx_kernel   (layer > 0) dy of the layer below = mask (.) sum_dir W_ih^T dG_t, tiles x time in parallel
//     bwd_dw_kernel   dW_ih, dW_hh, db = sum over (tile, t, curve) of dG_t [h_{t-1} | x_t | 1]^T, split over
//                     gate-row groups and chunks of (tile, t); chunk partials summed in fixed order
// Replaces autograd's traversal of aten::lstm in `loss.backward()` (utils/training_utils.py:523), as
// snn_lstm_backward does for H = 32.  Deterministic: no atomics, fixed summation order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>

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


constexpr int PAD = 33;   // LDS row stride of transposed-read images

// ---- weight images -------------------------------------------------------------------------------------
// WT_hh[ld][mt][kk][lane] = W_hh_eff[R = 2kk + (lane >> 5)][j = 32 mt + (lane & 31)]       (mt < H/32, kk < 2H)
// WT_ih[ld][mt][kk][lane] = W_ih[R][f = 32 mt + (lane & 31)]                               (layer >= 1, mt < I/32)
__device__ __forceinline__ void bwd_prep_body(const Bwd2Params& p, int bx, int nbx, int by)
{
    const SnnLayout& y = p.lay;
    const int ld = by >> 1, which = by & 1;   // 0: hh, 1: ih
    const int l = ld / y.dirs, d = ld - l * y.dirs;
    const int H = y.H, I = y.I[l];
    if (which == 1 && l == 0) return;
    const int cols = which == 0 ? H : I;
    const int64_t n = (int64_t)(cols / 32) * 2 * H * 64;
    float* dst = p.WT + (which == 0 ? p.wt_hh[ld] : p.wt_ih[ld]);
    const bool masked = p.mk.mode != SNN_MASK_NONE;
    for (int64_t idx = (int64_t)bx * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)nbx * blockDim.x) {
        const int lane = (int)(idx & 63);
        const int kk = (int)((idx >> 6) % (2 * H));
        const int mt = (int)(idx / ((int64_t)2 * H * 64));
        const int R = 2 * kk + (lane >> 5), col = 32 * mt + (lane & 31);
        // R is a row of the kernel image: row group R / H (LSTM: the gate; GRU: r, z, n input part, n hidden part; RNN:
        // group 0 only, the rest zero) of hidden unit R % H -- snn_layout.h
        const int grp = R / H, j = R - grp * H;
        const int k = which == 0 ? 2 * y.KKI[l] + col : col;
        float v = p.wk[y.k_A[ld] + ((int64_t)((j >> 3) * y.KK[l] + (k >> 1))) * 64 + (k & 1) * 32 + grp * 8 + (j & 7)];
        if (which == 0 && d == 0 && masked) v = __fmul_rn(mask_whh(p.mk, 0, y.L, H, l, R), v);   // variational_rnn.py:224-238
        dst[idx] = v;
    }
}

// ---- output layer backward ------------------------------------------------------------------------------
__device__ __forceinline__ void head_dfeat_body(const Bwd2Params& p, int tile)
{
    const SnnLayout& y = p.lay;
    const float* Wo = p.wk + y.k_head_w;
    float* out = p.dfeat + (int64_t)tile * y.F * 32;
    for (int idx = threadIdx.x; idx < y.F * 32; idx += blockDim.x) {
        const int f = idx >> 5, c = idx & 31;
        const int o = p.order[tile * SNN_TILE + c];
        float s = 0.0f;
        if (o >= 0) {
            for (int q = 0; q < y.C; ++q) s = fmaf(Wo[q * y.F + f], p.dlogits[(int64_t)o * y.C + q], s);
            if (p.mk.mode != SNN_MASK_NONE) s = __fmul_rn(mask_out(p.mk, 0, p.n_curves, y.F, o, f), s);
        }
        out[idx] = s;
    }
}

// dW_out[q][f] = sum over curves dlogits[curve][q] feat[curve][f];  db_out[q] = sum dlogits.  One workgroup per output
// element: thread k adds sorted positions k, k + 256, ... in order, then a fixed shuffle / LDS tree -- the order depends
// on the batch only, never on scheduling.
__device__ __forceinline__ void head_dw_body(const Bwd2Params& p, int n_chunks, int bx)
{
    // n_chunks == 1: the sum goes straight to dflat.  n_chunks > 1 (streamed path, large batches): chunk ch adds its
    // share of the sorted positions into slab ch (the head entries of the gradient slabs are otherwise unused) and
    // stream_reduce_kernel adds the chunks in index order.
    __shared__ float sh[4];
    const SnnLayout& y = p.lay;
    const int n_out = y.C * y.F + y.C;
    const int idx = bx % n_out, ch = bx / n_out;
    const int64_t feat0 = (int64_t)y.L * y.dirs * p.max_T * tape_step_floats(y.H);
    const int npos = p.n_tiles * SNN_TILE;
    const int per = ((npos + n_chunks - 1) / n_chunks + 255) / 256 * 256;
    const int pos0 = ch * per, pos1 = min(npos, pos0 + per);
    const bool is_w = idx < y.C * y.F;
    const int q = is_w ? idx / y.F : idx - y.C * y.F, f = is_w ? idx - q * y.F : 0;
    float s = 0.0f;
    for (int pos = pos0 + threadIdx.x; pos < pos1; pos += 256) {
        const int o = p.order[pos];
        if (o < 0) continue;
        const float dl = p.dlogits[(int64_t)o * y.C + q];
        s = is_w ? fmaf(dl, p.tape[(int64_t)(pos >> 5) * p.tape_tile + feat0 + f * 32 + (pos & 31)], s) : s + dl;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        const int64_t at = is_w ? y.c_head_w + idx : y.c_head_b + q;
        if (n_chunks == 1) p.dflat[at] = (p.accumulate ? p.dflat[at] : 0.0f) + tot;
        else p.slabs[(int64_t)ch * y.P + at] = tot;
    }
}

// The three preparations of a backward -- transposed weight images, dfeat per tile, output-layer gradient -- depend only
// on the forward's results and on each other not at all: ONE launch, a block range each (they were three launches of
// ~5 us; a 128-curve step is ~25 launches).
#define SNN_PREP_BX 64
__global__ void __launch_bounds__(256) bwd_head_prep_kernel(const Bwd2Params p, int head_chunks)
{
    const SnnLayout& y = p.lay;
    const int n_prep = SNN_PREP_BX * 2 * y.L * y.dirs;
    int b = blockIdx.x;
    if (b < n_prep) { bwd_prep_body(p, b % SNN_PREP_BX, SNN_PREP_BX, b / SNN_PREP_BX); return; }
    b -= n_prep;
    if (b < p.n_tiles) { head_dfeat_body(p, b); return; }
    head_dw_body(p, head_chunks, b - p.n_tiles);
}
static void launch_head_prep(const Bwd2Params& p, int head_chunks, hipStream_t st)
{
    const SnnLayout& y = p.lay;
    const int grid = SNN_PREP_BX * 2 * y.L * y.dirs + p.n_tiles + (y.C * y.F + y.C) * head_chunks;
    hipLaunchKernelGGL(bwd_head_prep_kernel, dim3(grid), dim3(256), 0, st, p, head_chunks);
}

// ---- the recurrent part ------------------------------------------------------------------------------
// CELL 0 LSTM (tape slots i, f, g, o, c, h), 1 GRU (r, z, n, W_hn h + b_hn, -, h), 2 tanh RNN (-, -, -, -, -, h).  The four
// row groups of dG follow the kernel image: GRU  d(r pre), d(z pre), d(n pre), d(n pre) * r   (input / hidden part of n);
// RNN  d(pre), 0, 0, 0.
template <int H, int CELL>
__global__ void __launch_bounds__(256, 1) bwd_rec_kernel(const Bwd2Params p, int l)
{
    constexpr int MT = H / 32;          // output tiles of dh
    constexpr int KSPLIT = 4 / MT;      // waves sharing one output tile (split over the gate rows)
    constexpr int KW = 2 * H / KSPLIT;  // MFMA k-steps per wave
    constexpr int UPT = H / 32;         // (hidden unit, 4 curves) items per thread in the element-wise stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dGl = smem;                  // [4H][32]  gate gradients of the running step (B operand image)
    float* red = dGl + 4 * H * 32;      // [KSPLIT][H][32]
    const SnnLayout& y = p.lay;
    const int dirs = y.dirs;
    const int tile = blockIdx.x / dirs, d = blockIdx.x - tile * dirs;
    const int ld = l * dirs + d;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 31, hf = lane >> 5;
    const int T = p.tile_T[tile];
    const bool last_layer = l == y.L - 1;
    // Element-wise stage: a thread owns 4 CONSECUTIVE curves of hidden units jg, jg + 32, ...: every tape row, dy row
    // and dG row is then read / written as one 16-byte word per thread (a step moved 24 loads + 16 stores of 4 bytes
    // per thread before; with 4 waves pushing their addresses through the compute unit's one address unit those
    // 40 instructions, not the 16 MFMAs, were the step's time at small batches).
    const int c4 = (tid & 7) * 4, jg = tid >> 3;
    int len[4];
    float inv_len[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        len[k] = p.len_sorted[tile * SNN_TILE + c4 + k];
        inv_len[k] = 1.0f / (float)(len[k] > 0 ? len[k] : 1);
    }
    const int64_t ts = tape_step_floats(H);
    const float* tape_ld = p.tape + (int64_t)tile * p.tape_tile + (int64_t)ld * p.max_T * ts;
    const float* dfeat = p.dfeat + (int64_t)tile * y.F * 32;
    const float* dy = last_layer ? nullptr : p.dy_in + ((int64_t)tile * dirs + d) * p.max_T * H * 32;
    float* dGg = p.dG + ((int64_t)tile * dirs + d) * p.max_T * 4 * H * 32;

    const int mt = w / KSPLIT, ks = w - mt * KSPLIT;
    float wT[KW];
    {
        const float* A = p.WT + p.wt_hh[ld] + ((int64_t)mt * 2 * H + ks * KW) * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) wT[kk] = A[kk * 64];
    }
    f32x4 dc[UPT], dhr[UPT], dfm[UPT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        const int j = jg + 32 * i;
        dc[i] = zero4;
        dhr[i] = y.pool == SNN_POOL_STANDARD ? *reinterpret_cast<const f32x4*>(dfeat + (j * (y.L * dirs) + ld) * 32 + c4) : zero4;
        dfm[i] = zero4;
        if (last_layer && y.pool == SNN_POOL_MEAN) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(dfeat + (d * H + j) * 32 + c4);
#pragma unroll
            for (int k = 0; k < 4; ++k) dfm[i][k] = v[k] * inv_len[k];
        }
    }
    // The tape rows (and dy) of a step are fetched one step ahead: the recurrence is a latency chain, and an
    // un-prefetched global load per step was most of it.  Rows t < T of the tape and of dy are always written
    // (forward / bwd_dx store every step of the tile), so the fetch needs no length test.
    f32x4 n_ig[UPT], n_fg[UPT], n_gg[UPT], n_og[UPT], n_cn[UPT], n_dy[UPT];
    auto fetch = [&](int t) __attribute__((always_inline)) {
        const float* tq = tape_ld + (int64_t)t * ts + c4;
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int j = jg + 32 * i;
            n_ig[i] = *reinterpret_cast<const f32x4*>(tq + (0 * H + j) * 32);
            n_fg[i] = *reinterpret_cast<const f32x4*>(tq + (1 * H + j) * 32);
            n_gg[i] = *reinterpret_cast<const f32x4*>(tq + (2 * H + j) * 32);
            n_og[i] = *reinterpret_cast<const f32x4*>(tq + (3 * H + j) * 32);
            n_cn[i] = *reinterpret_cast<const f32x4*>(tq + ((CELL == 0 ? 4 : 5) * H + j) * 32);   // LSTM: c_t; GRU / RNN: h_t
            n_dy[i] = dy ? *reinterpret_cast<const f32x4*>(dy + (int64_t)t * H * 32 + j * 32 + c4) : zero4;
        }
    };
    if (T > 0) fetch(d == 0 ? T - 1 : 0);

    for (int sb = 0; sb < T; ++sb) {
        const int step = T - 1 - sb;
        const int t = d == 0 ? step : T - 1 - step;
        const int tp = d == 0 ? t - 1 : t + 1;
        f32x4 ig[UPT], fg[UPT], gg[UPT], og[UPT], cn[UPT], dyv[UPT];
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            ig[i] = n_ig[i]; fg[i] = n_fg[i]; gg[i] = n_gg[i]; og[i] = n_og[i]; cn[i] = n_cn[i]; dyv[i] = n_dy[i];
        }
        if (step > 0) fetch(tp);   // n_cn is then c of the previous step: the cell state this step started from
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int j = jg + 32 * i;
            f32x4 dai = zero4, daf = zero4, dag = zero4, dao = zero4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool valid = t < len[k];
                const bool prev_live = step > 0 && tp < len[k];
                if (valid && CELL != 0) {
                    float dh = dhr[i][k];
                    if (dy) dh += dyv[i][k];
                    if (last_layer && y.pool == SNN_POOL_MEAN) dh += dfm[i][k];
                    if (CELL == 1) {   // h = (1 - z) n + z h_prev,  n = tanh(a2 + r a3)
                        const float rg = ig[i][k], zg = fg[i][k], ng = gg[i][k], a3 = og[i][k];
                        const float hp = prev_live ? n_cn[i][k] : 0.0f;
                        const float dnp = dh * (1.0f - zg) * (1.0f - ng * ng);
                        dai[k] = dnp * a3 * rg * (1.0f - rg);
                        daf[k] = dh * (hp - ng) * zg * (1.0f - zg);
                        dag[k] = dnp;
                        dao[k] = dnp * rg;
                        dc[i][k] = dh * zg;        // the direct path to h_prev, added to W_hh^T dG below
                    } else {           // h = tanh(a0); cn[] holds h_t
                        dai[k] = dh * (1.0f - cn[i][k] * cn[i][k]);
                        dc[i][k] = 0.0f;
                    }
                }
                if (valid && CELL == 0) {
                    float dh = dhr[i][k];
                    if (dy) dh += dyv[i][k];
                    if (last_layer && y.pool == SNN_POOL_MEAN) dh += dfm[i][k];
                    const float cp = prev_live ? n_cn[i][k] : 0.0f;
                    const float tc = snn_tanh(cn[i][k]);
                    const float dct = dc[i][k] + dh * og[i][k] * (1.0f - tc * tc);
                    dao[k] = dh * tc * og[i][k] * (1.0f - og[i][k]);
                    dai[k] = dct * gg[i][k] * ig[i][k] * (1.0f - ig[i][k]);
                    daf[k] = dct * cp * fg[i][k] * (1.0f - fg[i][k]);
                    dag[k] = dct * ig[i][k] * (1.0f - gg[i][k] * gg[i][k]);
                    dc[i][k] = dct * fg[i][k];
                }
            }
            *reinterpret_cast<f32x4*>(dGl + (0 * H + j) * 32 + c4) = dai;
            *reinterpret_cast<f32x4*>(dGl + (1 * H + j) * 32 + c4) = daf;
            *reinterpret_cast<f32x4*>(dGl + (2 * H + j) * 32 + c4) = dag;
            *reinterpret_cast<f32x4*>(dGl + (3 * H + j) * 32 + c4) = dao;
            float* o = dGg + (int64_t)t * 4 * H * 32 + j * 32 + c4;
            *reinterpret_cast<f32x4*>(o + 0 * H * 32) = dai;
            *reinterpret_cast<f32x4*>(o + 1 * H * 32) = daf;
            *reinterpret_cast<f32x4*>(o + 2 * H * 32) = dag;
            *reinterpret_cast<f32x4*>(o + 3 * H * 32) = dao;
        }
        __syncthreads();
        // dh_{prev}[j'][curve] = sum_R W_hh[R][j'] dG[R][curve]: this wave's output tile mt, gate rows of split ks
        {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            const float* b = dGl + (int64_t)ks * KW * 64 + lane;   // row 2kk + hf, column cl  ==  kk * 64 + lane
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wT[kk], b[kk * 64], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                red[((ks * H) + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * hf) * 32 + cl] = acc[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int j = jg + 32 * i;
            f32x4 s = zero4;
#pragma unroll
            for (int k = 0; k < KSPLIT; ++k) {
                const f32x4 r = *reinterpret_cast<const f32x4*>(red + (k * H + j) * 32 + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += r[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)      // beyond the curve the state is frozen: the gradient passes through unchanged
                if (t < len[e]) dhr[i][e] = CELL == 1 ? s[e] + dc[i][e] : s[e];
        }
    }
}

// ---- dy of the layer below ------------------------------------------------------------------------------
// TSW = time steps per wave.  A workgroup covers 4 steps; with fewer than 4 output tiles (H = 32: dirs * H / 32 = 1 or 2)
// the steps are split over the waves instead of leaving waves idle: wave w takes tile w % MTX, steps (w / MTX) * TSW ...
template <int H, int TSW>
__global__ void __launch_bounds__(256) bwd_dx_kernel(const Bwd2Params p, int l, int n_tchunks)
{
    // the 4 waves cover min(MTX, 4) output tiles x (4 / that) groups of TSW time steps: a workgroup covers
    // TS = TSW * 4 / min(MTX, 4) steps, and every W_ih^T operand load feeds TSW MFMAs.  Large batches take the largest
    // TSW (fewest operand loads); small ones TSW = 1, so that the steps of a tile spread over more compute units.
    const SnnLayout& y = p.lay;
    const int dirs = y.dirs;
    const int MTX = dirs * H / 32, MTXw = MTX < 4 ? MTX : 4;
    const int TS = TSW * (4 / MTXw);
    const int tile = blockIdx.x / n_tchunks, tw = (blockIdx.x - tile * n_tchunks) * TS;
    const int T = p.tile_T[tile];
    if (tw >= T) return;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 31, hf = lane >> 5;
    const int orig = p.order[tile * SNN_TILE + cl];
    const bool masked = p.mk.mode != SNN_MASK_NONE;
    const int mt0 = w % MTXw, mt_step = MTXw;
    const int t0 = tw + (w / MTXw) * TSW;
    for (int mt = mt0; mt < MTX; mt += mt_step) {
        f32x16 acc[TSW];
#pragma unroll
        for (int s = 0; s < TSW; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s][i] = 0.0f;
        for (int d = 0; d < dirs; ++d) {
            const int ld = l * dirs + d;
            const float* A = p.WT + p.wt_ih[ld] + (int64_t)mt * 2 * H * 64 + lane;
            const float* B = p.dG + (((int64_t)tile * dirs + d) * p.max_T + t0) * 4 * H * 32 + lane;
            // operands stream from L2: issue a batch of KB k-steps' loads, then their MFMAs (steps beyond the
            // tile's last one read the next step's rows of the same buffer and are never stored; steps beyond the
            // buffer re-read its last step: unconditional loads, no branch and no wait between them)
            constexpr int KB = TSW == 4 ? 16 : 32;
            int sc[TSW];
#pragma unroll
            for (int s = 0; s < TSW; ++s) sc[s] = t0 + s < p.max_T ? s : p.max_T - 1 - t0;
            for (int k0 = 0; k0 < 2 * H; k0 += KB) {
                float a[KB], b[TSW][KB];
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    a[k] = A[(k0 + k) * 64];
#pragma unroll
                    for (int s = 0; s < TSW; ++s) b[s][k] = B[(int64_t)sc[s] * 4 * H * 32 + (k0 + k) * 64];
                }
#pragma unroll
                for (int k = 0; k < KB; ++k)
#pragma unroll
                    for (int s = 0; s < TSW; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[s][k], acc[s], 0, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < TSW; ++s) {
            const int t = t0 + s;
            if (t < T) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int f = 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * hf;
                float v = acc[s][i];
                if (masked && orig >= 0) v = __fmul_rn(mask_mid(p.mk, 0, y.L, p.n_curves, dirs * H, l - 1, t, orig, f), v);
                const int dd = f / H, j = f - dd * H;
                p.dy_out[(((int64_t)tile * dirs + dd) * p.max_T + t) * H * 32 + j * 32 + cl] = v;
            }
            }
        }
    }
}

// ---- weight gradients -----------------------------------------------------------------------------------
// NTN = column tiles of [h_prev | x_t | 1] (ceil((H + I + 1) / 32)) is a template parameter: with a run-time tile count
// and accumulators for the widest layer, every MFMA of the loop sat behind a branch and the compiler kept the
// accumulators in scratch memory (2.3 KB per lane at H = 128: 857 scratch instructions per tile-step, 4.8 ms per
// layer for 128 curves x 300 steps -- 20 x the MFMA time).  One instantiation per (H, NTN) that occurs.
// OCC = workgroups per compute unit: 2 while the accumulators leave room (<= 7 column tiles; H=64 layer 1 14.6 against
// 15.6 ms per step at 65 536 curves).  Double-buffered LDS images (one barrier per tile-step) measured the same as
// single ones at either occupancy and are not kept.
template <int H, int NTN, int OCC, bool MASKED>
__global__ void __launch_bounds__(256, OCC) bwd_dw_kernel(const Bwd2Params p, int l, int n_chunks)
{
    constexpr int MG = 4 * H / 128;   // groups of 128 gate rows (one 32-row tile per wave)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dGs = smem;                // [128][PAD]
    float* xhs = dGs + 128 * PAD;     // [NTN*32][PAD]: rows h_prev (H), x_t (I), ones (1), zero padding
    const SnnLayout& y = p.lay;
    const int dirs = y.dirs;
    int b = blockIdx.x;
    const int chunk = b % n_chunks; b /= n_chunks;
    const int mg = b % MG, d = b / MG;
    const int ld = l * dirs + d;
    const int I = y.I[l];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 31, hf = lane >> 5;
    // MASKED = false is launched for models without dropout masks only: the Philox mask code (a few thousand
    // instructions behind run-time branches in the fetch) is not in that kernel at all
    const bool masked = MASKED && p.mk.mode != SNN_MASK_NONE;
    const int64_t ts = tape_step_floats(H);
    f32x16 acc[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;

    // A chunk is an equal share of the VALID tile-steps (tile_off is their running count times the packed x step: tiles
    // are sorted by length, so equal shares of the (tile, t < max_T) grid gave the chunks of the long tiles twice the
    // mean work and the kernel ran at the pace of those).
    const int64_t xsf = (int64_t)y.Dp * SNN_TILE;
    const int64_t total = p.tile_off[p.n_tiles] / xsf, per = (total + n_chunks - 1) / n_chunks;
    const int64_t v0 = std::min<int64_t>(total, chunk * per), v1 = std::min<int64_t>(total, v0 + per);
    int n_left = (int)(v1 - v0);
    int tile = 0, t = 0;
    if (n_left > 0) {      // the tile that holds valid step v0: the last one whose running count is <= v0
        int lo = 0, hi = p.n_tiles - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (p.tile_off[mid] <= v0 * xsf) lo = mid; else hi = mid - 1;
        }
        tile = lo;
        t = (int)(v0 - p.tile_off[lo] / xsf);
    }
    // Per-tile values of the fetch cursor, read when it enters a tile and kept in scalar registers: read per tile-step
    // (tile_T twice, tile_off once, each a dependent scalar load in front of the step's fetch) they stood between every
    // two MFMA blocks.
    int T_f = 0;
    const float *tape_f = nullptr, *x_f = nullptr, *dG_f = nullptr;
    auto enter_tile = [&](int tile) __attribute__((always_inline)) {
        T_f = p.tile_T[tile];
        tape_f = p.tape + (int64_t)tile * p.tape_tile;
        x_f = p.xp + p.tile_off[tile];
        dG_f = p.dG + ((int64_t)tile * dirs + d) * p.max_T * 4 * H * 32 + (int64_t)mg * 128 * 32;
    };
    // Software pipeline over the chunk's tile-steps: the operands of the next tile-step are fetched into registers
    // while the MFMAs of the current one run from LDS (an un-overlapped fetch was ~60 % of the kernel's time).
    f32x4 gv[4], xv[NTN];   // 128 dG rows and NTN * 32 [h_prev | x_t | 1 | 0] rows, 8 float4 per row, 256 threads
    // Branch-free: every load is unconditional at a clamped address, and what depends on the loaded VALUES -- zeros for
    // h_prev at a direction's first step, the ones row, the padding rows, the dropout masks -- is applied by commit(),
    // after the MFMAs the fetch runs beside.  With the rows' cases written as branches the compiler put s_waitcnt
    // vmcnt(0) at the joins, and with selects inside the fetch it waited for the loads right there: clock64 probes read
    // 5 200 / 3 300 cycles per tile-step for the 17 loads of H = 128 / layer 1 in front of 13 300 cycles of MFMAs.
    bool first_c = false;     // of the fetched tile-step, for commit()
    int tile_c = 0, t_c = 0;
    // The 4 + NTN loads of a fetch are dealt out over the 16 k-steps of the MFMA block (fetch_piece): issued in one
    // burst at the top of the tile-step, the four waves' 64-address instructions queued at the compute unit's one
    // address unit for ~2 200 cycles (H = 128, layer 1) before the first MFMA could issue.
    constexpr int NP = 4 + NTN;
    const float *src_f = nullptr, *hsrc_f = nullptr, *xsrc0_f = nullptr, *ysrc_f = nullptr;
    auto fetch_setup = [&](int tile, int t) __attribute__((always_inline)) {
        src_f = dG_f + (int64_t)t * 4 * H * 32;
        const bool first = d == 0 ? t == 0 : t == T_f - 1;   // first processed step of this direction: h_prev = 0
        const int tp = first ? t : (d == 0 ? t - 1 : t + 1);
        hsrc_f = tape_f + ((int64_t)ld * p.max_T + tp) * ts + 5 * H * 32;
        xsrc0_f = x_f + (int64_t)t * y.Dp * 32;                       // layer 0: packed input rows of step t
        ysrc_f = tape_f + (int64_t)t * ts + 5 * H * 32;               // layer > 0: outputs of the layer below
        first_c = first; tile_c = tile; t_c = t;
    };
    auto fetch_piece = [&](int j) __attribute__((always_inline)) {
        if (j < 4) {
            const int f = tid + j * 256, row = f >> 3, c4 = (f & 7) * 4;
            gv[j] = *reinterpret_cast<const f32x4*>(src_f + row * 32 + c4);
        } else {
            const int i = j - 4;
            const int f = tid + i * 256, row = f >> 3, c4 = (f & 7) * 4;
            if (32 * i + 31 < H) {        // (compile-time: H is a multiple of 32) rows of h_prev
                xv[i] = *reinterpret_cast<const f32x4*>(hsrc_f + row * 32 + c4);
            } else {
                const int k = row - H, kc = k < I ? k : I - 1;
                const int dd = kc / H, kk = kc - dd * H;
                const float* s0 = xsrc0_f + kc * 32 + c4;
                const float* s1 = ysrc_f + (int64_t)((l - 1) * dirs + dd) * p.max_T * ts + kk * 32 + c4;
                xv[i] = *reinterpret_cast<const f32x4*>(l == 0 ? s0 : s1);
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + i * 256, row = f >> 3, c4 = (f & 7) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) dGs[row * PAD + c4 + e] = gv[i][e];
        }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int i = 0; i < NTN; ++i) {
            const int f = tid + i * 256, row = f >> 3, c4 = (f & 7) * 4;
            f32x4 v = xv[i];
            if (32 * i + 31 < H) {
                v = first_c ? zero4 : v;
            } else {
                const int k = row - H;
                if (masked) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int o = p.order[tile_c * SNN_TILE + c4 + e];
                        if (o >= 0 && k < I)
                            v[e] = __fmul_rn(l == 0 ? mask_in(p.mk, 0, p.n_curves, y.D, o, k)
                                                    : mask_mid(p.mk, 0, y.L, p.n_curves, dirs * H, l - 1, t_c, o, k), v[e]);
                    }
                }
                v = k < I ? v : (k == I ? one4 : zero4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) xhs[row * PAD + c4 + e] = v[e];
        }
    };
    if (n_left > 0) {
        enter_tile(tile);
        fetch_setup(tile, t);
#pragma unroll
        for (int j = 0; j < NP; ++j) fetch_piece(j);
        commit();
    }
    __syncthreads();
    const float* dGc = dGs + (32 * w + cl) * PAD + hf;
    const float* xhc = xhs + cl * PAD + hf;
#ifdef SNN_PROBE_DW   // clock64 probes of workgroup 0, wave 0: cycles per tile-step in each part of the loop
    long long pc[5] = {0, 0, 0, 0, 0}; int pn = 0;
#define PROBE(i) { const long long now_ = clock64(); pc[i] += now_ - pt_; pt_ = now_; }
#else
#define PROBE(i)
#endif
    while (n_left > 0) {
#ifdef SNN_PROBE_DW
        long long pt_ = clock64(); ++pn;
#endif
        const bool more = n_left > 1;
        if (more) {                                        // the next tile-step of the chunk
            if (++t >= T_f) { ++tile; t = 0; enter_tile(tile); }
        }
        fetch_setup(tile, t);      // (the chunk's last tile-step fetches itself again: no branch in the MFMA block)
        PROBE(0)
        // operands of k-step kk + 1 are read from LDS before the MFMAs of k-step kk are issued
        float an = dGc[0], bn[NTN];
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) bn[nt] = xhc[32 * nt * PAD];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float a = an;
            float bc[NTN];
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) bc[nt] = bn[nt];
            if (kk + 1 < 16) {
                an = dGc[2 * kk + 2];
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) bn[nt] = xhc[32 * nt * PAD + 2 * kk + 2];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bc[nt], acc[nt], 0, 0, 0);
                if (nt == 0) {
#pragma unroll
                    for (int j = 0; j < NP; ++j)
                        if ((j * 16) / NP == kk) fetch_piece(j);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        PROBE(1)
        __syncthreads();
        PROBE(2)
        if (more) commit();
        PROBE(3)
        __syncthreads();
        PROBE(4)
        --n_left;
    }
#ifdef SNN_PROBE_DW
    if (blockIdx.x == 0 && tid == 0 && pn > 0)
        printf("bwd_dw<%d,%d> %d tile-steps: fetch %lld  mfma %lld  barrier %lld  commit %lld  barrier %lld cycles per tile-step\n", H, NTN, pn,
               pc[0] / pn, pc[1] / pn, pc[2] / pn, pc[3] / pn, pc[4] / pn);
#endif
#undef PROBE

    // this lane's 16 gate rows: image row -> canonical row of the hidden-side (w_hh, b_hh) / input-side (w_ih, b_ih)
    // tensors; -1: no such row (GRU: group 2 is the input part of n, group 3 its hidden part; RNN: group 0 only)
    float* slab = p.slabs + (int64_t)chunk * y.P;
    int Rh[16], Ri[16];
    float mh[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int R = 128 * mg + 32 * w + (i & 3) + 8 * (i >> 2) + 4 * hf;
        const int grp = R / H, jr = R - grp * H;
        Rh[i] = y.cell == 0 ? R : y.cell == 1 ? (grp == 2 ? -1 : (grp == 3 ? 2 : grp) * H + jr) : (grp == 0 ? jr : -1);
        Ri[i] = y.cell == 0 ? R : y.cell == 1 ? (grp == 3 ? -1 : R) : (grp == 0 ? jr : -1);
        mh[i] = (d == 0 && masked) ? mask_whh(p.mk, 0, y.L, H, l, R) : 1.0f;   // gradient of weight_hh_l0_raw
    }
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int col = 32 * nt + cl;
        if (col < H) {
            float* dst = slab + y.c_whh[ld] + col;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (Rh[i] >= 0) dst[(int64_t)Rh[i] * H] = mh[i] * acc[nt][i];
        } else if (col - H < I) {
            float* dst = slab + y.c_wih[ld] + (col - H);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (Ri[i] >= 0) dst[(int64_t)Ri[i] * I] = acc[nt][i];
        } else if (col == H + I) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (Ri[i] >= 0) slab[y.c_bih[ld] + Ri[i]] = acc[nt][i];
                if (Rh[i] >= 0) slab[y.c_bhh[ld] + Rh[i]] = acc[nt][i];
            }
        }
    }
}

// one instantiation per column-tile count that occurs for the hidden size (layer 0: D <= 78; layer >= 1: I = H or 2H)
typedef void (*bwd_dw_t)(const Bwd2Params, int, int);
#ifndef SNN_DW_OCC
#define SNN_DW_OCC(N) ((N) <= 3 ? 4 : (N) <= 7 ? 2 : 1)   // workgroups per compute unit (the grid comes in rounds of 256 x 2^k slots: bwd2_geom)
#endif
template <int H, int N>
static bwd_dw_t dw_pick(bool masked)
{
    constexpr int OCC = SNN_DW_OCC(N);
    return masked ? bwd_dw_kernel<H, N, OCC, true> : bwd_dw_kernel<H, N, OCC, false>;
}
template <int H>
static bwd_dw_t dw_kernel(int ntn, bool masked)
{
    switch (ntn) {
#define SNN_DW(N) case N: if constexpr (32 * (N) >= H + 2 && 32 * ((N) - 1) < H + 2 * H + 1) return dw_pick<H, N>(masked); else return nullptr;
        SNN_DW(2) SNN_DW(3) SNN_DW(4) SNN_DW(5) SNN_DW(6) SNN_DW(7) SNN_DW(8) SNN_DW(9) SNN_DW(10) SNN_DW(11) SNN_DW(12) SNN_DW(13)
#undef SNN_DW
    }
    return nullptr;
}

// dflat[i] (+)= sum over chunks, fixed order; LSTM parameters only (the head is written by head_dw_kernel)
__global__ void __launch_bounds__(256) chunk_reduce_kernel(int64_t n, int64_t P, int n_chunks, const float* __restrict__ slabs,
                                                          float* __restrict__ out, int accumulate)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? out[i] : 0.0f;
    // loads in batches of 8 (independent, in flight together), additions in index order: one dependent load per chunk
    // made this 20 us for the 80 chunks of a 128-curve batch
    int k = 0;
    for (; k + 8 <= n_chunks; k += 8) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = slabs[(int64_t)(k + e) * P + i];
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
    }
    for (; k < n_chunks; ++k) s += slabs[(int64_t)k * P + i];
    out[i] = s;
}

// ---- streamed backward (H = 32): everything of one (layer, direction) in ONE pass over the tape -------------------
// The split path above writes the gate gradients dG to HBM (16 KB per tile-step and direction) and reads them back
// twice (bwd_dx, bwd_dw): at 65 536 curves that is 7 of the 13.5 GB the backward moves, and bwd_rec alone is HBM-bound.
// Here a workgroup owns a direction and a fixed stripe of tiles (serpentine over the length-sorted tiles, so the
// stripes carry equal work and the summation order is fixed), walks each tile's tape backwards and keeps dG_t in LDS:
//   dh_{t-1}  = W_hh^T dG_t               16 MFMAs per step on each SIMD (K split over the 4 SIMDs: gate rows 32s..32s+31)
//   dy_below  = W_ih^T dG_t  (layer > 0)  16 x DIRS
//   dW       += dG_t [h_{t-1} | x_t]^T    16 x (1 + NTX); accumulators stay in registers for the whole stripe; the
//                                         [h | x] operand goes global -> registers directly: an MFMA k-step takes curves
//                                         {kk, kk + 16}, so a lane needs 16 CONTIGUOUS floats of its row
//   db       += dG_t                      per-thread registers, reduced over the 32 curves once at the end
// dG never leaves the chip.  8 waves: waves s and s + 4 share SIMD s and split its products by ROLE -- role 0 takes
// dh and the first half of the dW column tiles, role 1 dy_below and the rest -- so each wave stays under 256
// registers and two waves per SIMD cover each other's LDS / memory waits (with 4 waves of 455 registers every wait of
// the step was exposed: 13.8 k cycles per step against 6.1 k cycles of MFMA).  Everything a step reads from HBM is
// fetched one or two steps ahead by loads that are unconditional (clamped addresses, values discarded by selects),
// issued one by one between the MFMAs.  The kernel itself -- steps pipelined so that the gate gradients of step t - 1
// are computed between the dW MFMAs of step t, tiles pipelined so that a tile's first operands are fetched during the
// last steps of the tile before it -- is in snn_bwd_stream3.h; its predecessor (step = algebra | barrier | products |
// barrier | reduction) is kept, not built, in microbench/experiments/snn_bwd_stream1.h.
#ifndef SNN_STREAM_OCC
#define SNN_STREAM_OCC 1
#endif
#include "snn_bwd_stream3.h"

// dflat[i] (+)= sum over the G workgroups of the direction parameter i belongs to, fixed order
__global__ void __launch_bounds__(64) stream_reduce_kernel(SnnLayout y, int G, int head_chunks,
                                                           const float* __restrict__ slabs, float* __restrict__ out,
                                                           int accumulate)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= y.P) return;
    if (i >= y.c_head_w) {          // output layer: head_dw_kernel's chunks (one chunk wrote dflat itself)
        if (head_chunks == 1) return;
        float s = accumulate ? out[i] : 0.0f;
        for (int k = 0; k < head_chunks; ++k) s += slabs[(int64_t)k * y.P + i];
        out[i] = s;
        return;
    }
    int ld = 0;
    const int nld = y.L * y.dirs;
    while (ld + 1 < nld && i >= y.c_wih[ld + 1]) ++ld;
    const int d = ld % y.dirs;
    float s = accumulate ? out[i] : 0.0f;
    const float* src = slabs + (int64_t)d * G * y.P + i;
    // loads in batches of 8 (independent, in flight together), additions in slab order; 64-thread workgroups so that the
    // ~42 k parameters of the H = 32 model give every compute unit a few waves (one dependent load per slab on 165
    // workgroups was 52 us, 8 % of an 8192-curve step)
    int k = 0;
    for (; k + 8 <= G; k += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)(k + j) * y.P];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; k < G; ++k) s += src[(int64_t)k * y.P];
    out[i] = s;
}

// ---- host side ------------------------------------------------------------------------------------------
struct Bwd2Geom { int n_chunks, n_dG; int64_t wt_floats, dfeat_floats, dG_floats, dy_floats, slab_floats, total_bytes; };

// Small batch: even half tiles get a compute unit each.  The split backward then (a) runs its recurrent kernels on half
// tiles where they exist (snn_rec16.hip), (b) spreads bwd_dx over single time steps, and (c) keeps the gate gradients of
// EVERY layer (n_dG = L buffers instead of one) so that the weight-gradient GEMM of a layer runs on a side stream beside
// the recurrent chain of the layer below instead of in front of it.  SNN_REC16=0|1 forces the rule off / on (A/B, tests).
static bool small_batch(const SnnLayout& y, int n_tiles)
{
    const char* f16 = getenv("SNN_REC16");
    if (f16) return f16[0] == '1';
    static thread_local int cached_dev = -1, cached_cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (dev != cached_dev) {
        if (hipDeviceGetAttribute(&cached_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        cached_dev = dev;
    }
    return (int64_t)n_tiles * y.dirs * 2 <= cached_cus;
}

static void bwd2_geom(const SnnLayout& y, int n_tiles, int max_T, Bwd2Geom* g, int64_t* wt_hh, int64_t* wt_ih, bool small = false)
{
    int64_t off = 0;
    for (int ld = 0; ld < y.L * y.dirs; ++ld) {
        const int l = ld / y.dirs;
        if (wt_hh) wt_hh[ld] = off;
        off += (int64_t)(y.H / 32) * 2 * y.H * 64;
        if (wt_ih) wt_ih[ld] = off;
        if (l > 0) off += (int64_t)(y.I[l] / 32) * 2 * y.H * 64;
    }
    g->wt_floats = off;
    g->dfeat_floats = (int64_t)n_tiles * y.F * 32;
    g->dG_floats = (int64_t)n_tiles * y.dirs * max_T * 4 * y.H * 32;
    g->dy_floats = (int64_t)n_tiles * y.dirs * max_T * y.H * 32;
    const int64_t total = (int64_t)n_tiles * max_T;
    // weight-gradient partial sums: one slab of P floats per chunk of tile-steps; up to 256 chunks while the slabs stay
    // under 96 MB (small models, large batches), never fewer than the 64 the large models ran with
    const int64_t cap = std::max<int64_t>(64, std::min<int64_t>(256, ((int64_t)96 << 20) / ((int64_t)sizeof(float) * y.P)));
    // (a chunk is at least ~8 tile-steps: every chunk costs a slab of P floats that chunk_reduce_kernel reads back)
    g->n_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(cap, std::max<int64_t>(total / 8, std::min<int64_t>(total / 2, 16))));
    // chunks are equal shares of the work, so the grid (dirs x row groups x chunks) should come in whole rounds of the
    // chip's workgroup slots: 256 compute units x 2 (this geometry is also the workspace size, so it is not asked of
    // the device; on another part the rounding is merely not the best one)
    const int unit = std::max(1, 512 / (y.dirs * std::max(1, 4 * y.H / 128)));
    if (g->n_chunks > unit) g->n_chunks -= g->n_chunks % unit;
    if (const char* e = getenv("SNN_DW_CHUNKS")) g->n_chunks = std::max(1, atoi(e));
    g->slab_floats = (int64_t)g->n_chunks * y.P;
    g->n_dG = small ? y.L : 1;
    g->total_bytes = (int64_t)sizeof(float) * (g->wt_floats + g->dfeat_floats + g->n_dG * g->dG_floats + 2 * g->dy_floats + g->slab_floats) + 1024;
}

// ---- streamed path, host side
struct StreamGeom { int G; int64_t wt_floats, dfeat_floats, dy_floats, slab_floats, total_bytes; };
typedef void (*bwd_stream_t)(const Bwd2Params, int, int, float*, const float*);

bool snn_bwd_stream_supported(const SnnLayout& y) { return y.cell == 0 && y.H == 32 && y.D <= SNN_MAX_D && y.Dp <= 96; }

static int stream_geom(const SnnLayout& y, int n_tiles, int max_T, StreamGeom* g, int64_t* wt_hh, int64_t* wt_ih)
{
    Bwd2Geom b;
    bwd2_geom(y, n_tiles, max_T, &b, wt_hh, wt_ih);
    static thread_local int cached_dev = -1, cached_cus = 0;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev != cached_dev) {
        HIP_CHECK(hipDeviceGetAttribute(&cached_cus, hipDeviceAttributeMultiprocessorCount, dev));
        cached_dev = dev;
    }
    g->G = std::max(1, std::min(n_tiles, SNN_STREAM_OCC * cached_cus / y.dirs));   // workgroups per direction
    g->wt_floats = b.wt_floats;
    g->dfeat_floats = b.dfeat_floats;
    g->dy_floats = y.L > 1 ? (int64_t)y.dirs * b.dy_floats : 0;      // one block per producing direction
    g->slab_floats = (int64_t)y.dirs * g->G * y.P;
    g->total_bytes = (int64_t)sizeof(float) * (g->wt_floats + g->dfeat_floats + (y.L > 2 ? 2 : 1) * g->dy_floats + g->slab_floats) + 1024;
    return SNN_OK;
}

int64_t snn_bwd_stream_workspace_bytes(const SnnLayout& y, int n_tiles, int max_T)
{
    StreamGeom g;
    int rc = stream_geom(y, n_tiles, max_T, &g, nullptr, nullptr);
    return rc ? rc : g.total_bytes;
}

template <int DIRS, bool MASKED>
static bwd_stream_t stream_kernel(int ntx, bool first)
{
    if (first) return ntx == 1 ? bwd_stream3_kernel<1, true, DIRS, MASKED> : ntx == 2 ? bwd_stream3_kernel<2, true, DIRS, MASKED>
                                                                                    : bwd_stream3_kernel<3, true, DIRS, MASKED>;
    return bwd_stream3_kernel<DIRS, false, DIRS, MASKED>;
}

int snn_lstm_backward_stream(const SnnLayout& y, const float* xp, const int32_t* len_sorted, const int32_t* order,
                             const int32_t* tile_T, const int64_t* tile_off, int32_t n_curves, int32_t n_tiles, int32_t max_T,
                             const float* wk, const MaskP& mk, const float* tape, int64_t tape_tile, const float* dlogits,
                             float* dflat, int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream)
{
    Bwd2Params p;
    memset(&p, 0, sizeof(p));
    StreamGeom g;
    int rc = stream_geom(y, n_tiles, max_T, &g, p.wt_hh, p.wt_ih);
    if (rc) return rc;
    if (workspace_bytes < g.total_bytes) {
        snn_set_error("snn_lstm_backward: workspace %lld < required %lld bytes", (long long)workspace_bytes, (long long)g.total_bytes);
        return SNN_ERR_WORKSPACE;
    }
    p.lay = y;
    p.xp = xp; p.len_sorted = len_sorted; p.order = order; p.tile_T = tile_T; p.tile_off = tile_off;
    p.n_curves = n_curves; p.n_tiles = n_tiles; p.max_T = max_T; p.wk = wk; p.mk = mk;
    p.tape = tape; p.tape_tile = tape_tile; p.dlogits = dlogits;
    p.WT = (float*)workspace;
    p.dfeat = p.WT + g.wt_floats;
    float* dyA = p.dfeat + g.dfeat_floats;
    float* dyB = y.L > 2 ? dyA + g.dy_floats : dyA;
    p.slabs = dyA + (y.L > 2 ? 2 : 1) * g.dy_floats;
    p.dflat = dflat;
    p.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    const int n_ld = y.L * y.dirs;
    // output-layer gradient: one chunk of sorted positions per 2048 curves, at most one per gradient slab
    const int head_chunks = std::max(1, std::min(y.dirs * g.G, p.n_tiles * SNN_TILE / 2048));
    launch_head_prep(p, head_chunks, st);
    HIP_CHECK(hipGetLastError());
    for (int l = y.L - 1; l >= 0; --l) {
        const bool first = l == 0;
        const int ntx = first ? (y.Dp + 31) / 32 : y.dirs;
        const bool masked = mk.mode != SNN_MASK_NONE;
        bwd_stream_t fn = y.dirs == 2 ? (masked ? stream_kernel<2, true>(ntx, first) : stream_kernel<2, false>(ntx, first))
                                      : (masked ? stream_kernel<1, true>(ntx, first) : stream_kernel<1, false>(ntx, first));
        const int mtx = first ? 0 : y.dirs;
        // LDS: dG of two steps [2][4H][33], dh partials [4][H][32], dy partials [4][mtx * 32][32], the curves of two tiles [2][32]
        const size_t lds = sizeof(float) * ((size_t)2 * 4 * 32 * 33 + 4 * 32 * 32 + (size_t)4 * mtx * 32 * 32 + 64);
        HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        float* dy_out = (l & 1) ? dyA : dyB;          // written by layer l for layer l - 1
        const float* dy_in = (l & 1) ? dyB : dyA;     // written by layer l + 1
        hipLaunchKernelGGL(fn, dim3(y.dirs * g.G), dim3(512), lds, st, p, l, g.G, dy_out, dy_in);
        HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(stream_reduce_kernel, dim3((unsigned)((y.P + 63) / 64)), dim3(64), 0, st, y, g.G, head_chunks, p.slabs,
                       p.dflat, p.accumulate);
    HIP_CHECK(hipGetLastError());
    return SNN_OK;
}

bool snn_bwd_big_supported(const SnnLayout& y)
{
    return (y.H == 32 || y.H == 64 || (y.H == 128 && y.cell == 0)) && y.D <= SNN_MAX_D;
}

int64_t snn_bwd_big_workspace_bytes(const SnnLayout& y, int n_tiles, int max_T)
{
    Bwd2Geom g;
    bwd2_geom(y, n_tiles, max_T, &g, nullptr, nullptr, small_batch(y, n_tiles));
    return g.total_bytes;
}

bool snn_rec16_supported(const SnnLayout& y);                          // snn_rec16.hip
int snn_launch_bwd_rec16(const Bwd2Params& p, int l, hipStream_t st);
bool snn_launch_bwd_dx_lds(const Bwd2Params& p, int l, hipStream_t st);   // snn_bwd_dx.hip

// side stream + events of the overlapped weight-gradient GEMMs (small batches), one set per host thread and device
struct SideStream { int dev = -1; hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static int side_stream(SideStream** out)
{
    static thread_local SideStream ss[SNN_NOALLOC_MAX_DEVICES];
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= SNN_NOALLOC_MAX_DEVICES) { snn_set_error("device ordinal %d out of range", dev); return SNN_ERR_ARG; }
    SideStream& x = ss[dev];
    if (!x.s) {
        HIP_CHECK(hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&x.fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&x.join, hipEventDisableTiming));
        x.dev = dev;
    }
    *out = &x;
    return SNN_OK;
}

template <int H, int CELL>
static int launch_big(const Bwd2Params& p0, const Bwd2Geom& g, hipStream_t st)
{
    Bwd2Params p = p0;
    const SnnLayout& y = p.lay;
    launch_head_prep(p, 1, st);
    HIP_CHECK(hipGetLastError());
    constexpr int KSPLIT = 4 / (H / 32);
    const size_t lds_rec = sizeof(float) * ((size_t)4 * H * 32 + (size_t)KSPLIT * H * 32);
    HIP_CHECK(hipFuncSetAttribute((const void*)bwd_rec_kernel<H, CELL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rec));
    float* dG0 = p.dG;
    float* dyA = p.dfeat + g.dfeat_floats + g.n_dG * g.dG_floats;
    float* dyB = dyA + g.dy_floats;
    const bool small = g.n_dG > 1 || (y.L == 1 && small_batch(y, p.n_tiles));   // (see small_batch)
    // half tiles: always for small batches; at H = 128 for every size (two waves per SIMD with 128 k-steps of W_hh^T
    // each cover each other's tape loads where bwd_rec_kernel<128> has one wave per SIMD: backward 73.4 -> 68.2 ms at
    // 8192 curves x 300 steps; at H = 64 the 32-curve kernel is the faster one at size: 10.5 against 11.7 ms)
    const char* f16 = getenv("SNN_REC16");      // "0": the 32-curve kernels everywhere (A/B, tests)
    const bool rec16 = snn_rec16_supported(y) && (small || (H == 128 && !(f16 && f16[0] == '0')));
    const bool overlap = g.n_dG == y.L && y.L > 1;
    SideStream* side = nullptr;
    if (overlap && side_stream(&side) != SNN_OK) return SNN_ERR_HIP;
    for (int l = y.L - 1; l >= 0; --l) {
        p.dy_in = (l & 1) ? dyA : dyB;    // written by the layer above
        p.dy_out = (l & 1) ? dyB : dyA;
        p.dG = dG0 + (overlap ? (int64_t)l * g.dG_floats : 0);
        if (rec16) snn_launch_bwd_rec16(p, l, st);
        else hipLaunchKernelGGL((bwd_rec_kernel<H, CELL>), dim3(p.n_tiles * y.dirs), dim3(256), lds_rec, st, p, l);
        HIP_CHECK(hipGetLastError());
        // chip-filling batches of wide layers: the LDS-staged kernel of snn_bwd_dx.hip (SNN_DX_LDS=0: A/B, tests)
        const char* dxl = getenv("SNN_DX_LDS");
        if (l > 0 && !small && !(dxl && dxl[0] == '0') && snn_launch_bwd_dx_lds(p, l, st)) {
            HIP_CHECK(hipGetLastError());
        } else if (l > 0) {
            const int mtx = y.dirs * H / 32, mtxw = std::min(mtx, 4);
            // time steps per wave: 4 * mtxw / 4 fills a 4-step workgroup (fewest operand loads); 1 for small batches
            int tsw = mtx == 1 ? 1 : mtx == 2 ? 2 : 4;
            if (small && mtx <= 2) tsw = 1;
            const int ts = tsw * (4 / mtxw);
            const int n_tchunks = (p.max_T + ts - 1) / ts;
            auto dx = tsw == 1 ? bwd_dx_kernel<H, 1> : tsw == 2 ? bwd_dx_kernel<H, 2> : bwd_dx_kernel<H, 4>;
            hipLaunchKernelGGL(dx, dim3(p.n_tiles * n_tchunks), dim3(256), 0, st, p, l, n_tchunks);
            HIP_CHECK(hipGetLastError());
        }
        const int ntn = (H + y.I[l] + 1 + 31) / 32;
        bwd_dw_t dw = dw_kernel<H>(ntn, p.mk.mode != SNN_MASK_NONE);
        if (!dw) { snn_set_error("no weight-gradient kernel for H=%d with input width %d", H, y.I[l]); return SNN_ERR_UNSUPPORTED; }
        const size_t lds_dw = sizeof(float) * ((size_t)128 * PAD + (size_t)ntn * 32 * PAD);
        HIP_CHECK(hipFuncSetAttribute((const void*)dw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dw));
        // the weight gradients of layer l need only this layer's dG (kept: one buffer per layer) and the tape: on the side
        // stream they run beside bwd_dx / the recurrent kernel of layer l - 1; the last layer's run on the main stream
        hipStream_t dws = st;
        if (overlap && l > 0) {
            HIP_CHECK(hipEventRecord(side->fork, st));
            HIP_CHECK(hipStreamWaitEvent(side->s, side->fork, 0));
            dws = side->s;
        }
        hipLaunchKernelGGL(dw, dim3(y.dirs * (4 * H / 128) * g.n_chunks), dim3(256), lds_dw, dws, p, l, g.n_chunks);
        HIP_CHECK(hipGetLastError());
    }
    if (overlap) {      // join: everything the side stream was given precedes the reduction
        HIP_CHECK(hipEventRecord(side->join, side->s));
        HIP_CHECK(hipStreamWaitEvent(st, side->join, 0));
    }
    hipLaunchKernelGGL(chunk_reduce_kernel, dim3((unsigned)((y.c_head_w + 255) / 256)), dim3(256), 0, st, y.c_head_w, y.P,
                       g.n_chunks, p.slabs, p.dflat, p.accumulate);
    HIP_CHECK(hipGetLastError());
    return SNN_OK;
}

int snn_lstm_backward_big(const SnnLayout& y, const float* xp, const int32_t* len_sorted, const int32_t* order,
                          const int32_t* tile_T, const int64_t* tile_off, int32_t n_curves, int32_t n_tiles, int32_t max_T,
                          const float* wk, const MaskP& mk, const float* tape, int64_t tape_tile, const float* dlogits,
                          float* dflat, int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream)
{
    Bwd2Params p;
    memset(&p, 0, sizeof(p));
    Bwd2Geom g;
    bwd2_geom(y, n_tiles, max_T, &g, p.wt_hh, p.wt_ih, small_batch(y, n_tiles));
    if (workspace_bytes < g.total_bytes) {
        snn_set_error("snn_lstm_backward: workspace %lld < required %lld bytes", (long long)workspace_bytes, (long long)g.total_bytes);
        return SNN_ERR_WORKSPACE;
    }
    p.lay = y;
    p.xp = xp; p.len_sorted = len_sorted; p.order = order; p.tile_T = tile_T; p.tile_off = tile_off;
    p.n_curves = n_curves; p.n_tiles = n_tiles; p.max_T = max_T; p.wk = wk; p.mk = mk;
    p.tape = tape; p.tape_tile = tape_tile; p.dlogits = dlogits;
    p.WT = (float*)workspace;
    p.dfeat = p.WT + g.wt_floats;
    p.dG = p.dfeat + g.dfeat_floats;
    p.slabs = p.dG + g.n_dG * g.dG_floats + 2 * g.dy_floats;
    p.dflat = dflat;
    p.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    if (y.cell == 1 && y.H == 64) return launch_big<64, 1>(p, g, st);
    if (y.cell == 1 && y.H == 32) return launch_big<32, 1>(p, g, st);
    if (y.cell == 2 && y.H == 64) return launch_big<64, 2>(p, g, st);
    if (y.cell == 2 && y.H == 32) return launch_big<32, 2>(p, g, st);
    if (y.cell != 0) { snn_set_error("no backward kernel for cell %d at H=%d", y.cell, y.H); return SNN_ERR_UNSUPPORTED; }
    if (y.H == 128) return launch_big<128, 0>(p, g, st);
    if (y.H == 64) return launch_big<64, 0>(p, g, st);
    if (y.H == 32) return launch_big<32, 0>(p, g, st);
    snn_set_error("no backward kernel for H=%d", y.H);
    return SNN_ERR_UNSUPPORTED;
}
