// snn_rec16.hip -- the recurrent halves of a small-batch training step on HALF tiles (16 curves per workgroup).
//
// At the reference's batch size (128 curves, conf.py:517) a training step is a chain of L * T dependent time steps on
// a handful of workgroups, and what a step costs is the MFMA chain of one wave + the cell update of its lanes: with a
// 32-curve tile per workgroup (lstm_fwd_rec_kernel, bwd_rec_kernel) 16 x v_mfma_f32_32x32x2_f32 = 1024 cycles and
// 4 hidden units per lane.  Curves are independent, so a tile's two halves can run on two compute units: 16 curves per
// workgroup on v_mfma_f32_16x16x4_f32 (16 x 32 cycles per wave and step) and 2 hidden units per lane -- half the
// chain, twice the workgroups.  Memory layouts stay those of the 32-curve kernels (tape [q][H][32], Gx in the
// 32x32 C/D image, dG [4H][32]): a workgroup reads and writes its 16 columns, so fwd_gx_kernel, bwd_dx_kernel,
// bwd_dw_kernel and the head kernels are unchanged.  LSTM cell, H = 32 / 64 / 128 (H = 128: 16 waves forward, each with
// its 2 x 32 k-steps of W_hh in registers; 8 waves backward with 128 k-steps of W_hh^T each -- the chain per step is
// 64 / 128 MFMAs of 32 cycles where the 32-curve kernels of snn_forward_big.hip / bwd_rec_kernel<128> run 256 of 64).  v_mfma_f32_16x16x4_f32 adds its four
// products in another order than two v_mfma_f32_32x32x2_f32, so results differ from the 32-curve kernels by fp32
// rounding (same parity bar against the reference fixtures; tests/test_gpu_train.py).
//
// Operand layouts of v_mfma_f32_16x16x4_f32 (lane l): A[m = l % 16][k = l / 16], B[k = l / 16][n = l % 16],
// C/D register i: [m = 4 (l / 16) + i][n = l % 16].  Forward: a 16-row block is 4 hidden units x 4 gates with row
// 4 u + gate, so lane (g = l / 16, c = l % 16) ends up with i, f, g, o of unit 4 b + g for curve c.
// Replaces, as the kernels it stands in for, aten::lstm in `rnn(packed_tensor)` / `loss.backward()`
// (utils/training_utils.py:514, 523).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snn_kernels.h"
#include "snn_host.h"

struct Rec16Params {
    int dirs, KK, KKI;
    int64_t k_A[2];
    int64_t k_head_w;
};

template <int H, bool MASKED>
__global__ void __launch_bounds__(H * 8) lstm_fwd_rec16_kernel(const FwdParams p, const Rec16Params r, int l,
                                                               float* __restrict__ head_part, const float* __restrict__ gx)
{
    constexpr int NT = H * 8, NW = H / 8, KS = H / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* stage = smem;                          // [2][6][H][16]: tape rows of the last two steps, this half's columns
    float* hb = stage + 2 * SNN_TAPE_Q * H * 16;  // [2][H][16]
    const int dirs = r.dirs;
    int b = blockIdx.x;
    const int half = b & 1; b >>= 1;
    const int d = b % dirs, tile = b / dirs;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int cg = 16 * half + c;
    const int T = __builtin_amdgcn_readfirstlane(p.tile_T[tile]);
    const int pos = tile * SNN_TILE + cg;
    const int len = p.len_sorted[pos], orig = p.order[pos];
    const int ld = l * dirs + d;
    const bool last_layer = l == p.L - 1;
    const MaskP& mk = p.mk;
    const bool masked = MASKED && mk.mode != SNN_MASK_NONE;

    // ---- W_hh rows of this wave's two blocks (units 8 w .. 8 w + 7), gathered from the forward weight image
    float wA[2][KS];
    {
        const int rr = lane & 15, ul = rr >> 2, gate = rr & 3;
#pragma unroll
        for (int bk = 0; bk < 2; ++bk) {
            const int jr = 8 * w + 4 * bk + ul;
            float m = 1.0f;
            if (MASKED && d == 0 && masked) m = mask_whh(mk, 0, p.L, H, l, gate * H + jr);   // variational_rnn.py:224-238
            const float* A = p.wk + r.k_A[d] + (int64_t)(jr >> 3) * r.KK * 64 + gate * 8 + (jr & 7);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int k = 2 * r.KKI + 4 * kk + g;
                const float v = A[(k >> 1) * 64 + (k & 1) * 32];
                wA[bk][kk] = MASKED ? __fmul_rn(m, v) : v;
            }
        }
    }
    const int64_t ts = tape_step_floats(H);
    float* tape_ld = p.tape + (int64_t)tile * p.tape_tile + (int64_t)ld * p.max_T * ts;
    // Gx of (unit 8 w + 4 bk + g, gate i, curve cg) in the 32x32 C/D image: row tile w, register 4 i + g, lane 32 bk + cg
    const float* gxw = gx + (((int64_t)tile * dirs + d) * p.max_T * NW + w) * 1024 + g * 64 + cg;
    f32x4 gxn[2];
    auto fetch_gx = [&](int t) __attribute__((always_inline)) {
        const float* q = gxw + (int64_t)t * NW * 1024;
#pragma unroll
        for (int bk = 0; bk < 2; ++bk)
#pragma unroll
            for (int i = 0; i < 4; ++i) gxn[bk][i] = q[i * 256 + bk * 32];
    };
    auto flush = [&](int step_done, int t_done) __attribute__((always_inline)) {   // staged tape row -> global, 16-byte words
        const float* sg = stage + (step_done & 1) * (SNN_TAPE_Q * H * 16);
        float* dst = tape_ld + (int64_t)t_done * ts + 16 * half;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = (tid + i * NT) * 4, row = e >> 4, c4 = e & 15;
            *reinterpret_cast<f32x4*>(dst + row * 32 + c4) = *reinterpret_cast<const f32x4*>(sg + e);
        }
    };

    float cst[2] = {0.f, 0.f}, hst[2] = {0.f, 0.f}, hsum[2] = {0.f, 0.f};
    if (T > 0) fetch_gx(d == 0 ? 0 : T - 1);
#pragma unroll
    for (int bk = 0; bk < 2; ++bk) hb[(8 * w + 4 * bk + g) * 16 + c] = 0.0f;
    lds_barrier();

    for (int step = 0; step < T; ++step) {
        const int t = d == 0 ? step : T - 1 - step;
        const int cur = step & 1;
        f32x4 acc[2] = {gxn[0], gxn[1]};
        if (step + 1 < T) fetch_gx(d == 0 ? t + 1 : t - 1);
        if (step > 0) flush(step - 1, d == 0 ? t - 1 : t + 1);
        const float* hc = hb + cur * (H * 16) + lane;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const float bv = hc[kk * 64];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[0][kk], bv, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[1][kk], bv, acc[1], 0, 0, 0);
        }
        const bool valid = t < len;
        float* sgw = stage + cur * (SNN_TAPE_Q * H * 16) + c;
#pragma unroll
        for (int bk = 0; bk < 2; ++bk) {
            const int j = 8 * w + 4 * bk + g;
            const float ig = snn_sigmoid(acc[bk][0]);
            const float fg = snn_sigmoid(acc[bk][1]);
            const float gg = snn_tanh(acc[bk][2]);
            const float og = snn_sigmoid(acc[bk][3]);
            const float cn = fmaf(fg, cst[bk], __fmul_rn(ig, gg));
            const float hv = og * snn_tanh(cn);
            cst[bk] = valid ? cn : cst[bk];
            hst[bk] = valid ? hv : hst[bk];
            const float outv = valid ? hv : 0.0f;
            hb[(cur ^ 1) * (H * 16) + j * 16 + c] = hst[bk];
            sgw[(0 * H + j) * 16] = ig;
            sgw[(1 * H + j) * 16] = fg;
            sgw[(2 * H + j) * 16] = gg;
            sgw[(3 * H + j) * 16] = og;
            sgw[(4 * H + j) * 16] = cn;
            sgw[(5 * H + j) * 16] = outv;
            hsum[bk] += outv;
        }
        lds_barrier();
    }
    if (T > 0) flush(T - 1, d == 0 ? T - 1 : 0);

    // ---- head contribution of this (layer, direction): pooled features to the tape, per-wave partial logits
    if (p.pool == SNN_POOL_STANDARD || last_layer) {
        float feat[2];
        int fidx[2];
#pragma unroll
        for (int bk = 0; bk < 2; ++bk) {
            const int j = 8 * w + 4 * bk + g;
            if (p.pool == SNN_POOL_MEAN) {
                feat[bk] = __fdiv_rn(hsum[bk], (float)(len > 0 ? len : 1));
                fidx[bk] = d * H + j;
            } else {
                feat[bk] = hst[bk];
                fidx[bk] = j * (p.L * dirs) + ld;
            }
            if (MASKED && masked && orig >= 0) feat[bk] = __fmul_rn(mask_out(mk, 0, p.n_curves, p.F, orig, fidx[bk]), feat[bk]);
        }
        float* fs = p.tape + (int64_t)tile * p.tape_tile + (int64_t)p.L * dirs * p.max_T * ts;
        fs[fidx[0] * 32 + cg] = feat[0];
        fs[fidx[1] * 32 + cg] = feat[1];
        const float* Wo = p.wk + r.k_head_w;
        float* dst = head_part + (((int64_t)tile * (p.L * dirs) + ld) * NW + w) * (p.C * 32) + cg;
        for (int q = 0; q < p.C; ++q) {
            float a = fmaf(Wo[q * p.F + fidx[0]], feat[0], 0.0f);
            a = fmaf(Wo[q * p.F + fidx[1]], feat[1], a);
            a += __shfl_xor(a, 16);
            a += __shfl_xor(a, 32);
            if (g == 0) dst[q * 32] = a;
        }
    }
}

typedef void (*rec16_fwd_t)(const FwdParams, const Rec16Params, int, float*, const float*);

bool snn_rec16_supported(const SnnLayout& y) { return y.cell == 0 && y.prec == 0 && (y.H == 32 || y.H == 64 || y.H == 128) && y.dirs <= 2; }

// recurrent half of layer l of a training forward on half tiles; gx from fwd_gx_kernel; head_part as lstm_fwd_rec_kernel's
int snn_launch_fwd_rec16(const SnnLayout& y, const FwdParams& p, int l, float* head_part, const float* gx, hipStream_t st)
{
    const bool masked = p.mk.mode != SNN_MASK_NONE;
    rec16_fwd_t fn = y.H == 32 ? (masked ? lstm_fwd_rec16_kernel<32, true> : lstm_fwd_rec16_kernel<32, false>)
                   : y.H == 64 ? (masked ? lstm_fwd_rec16_kernel<64, true> : lstm_fwd_rec16_kernel<64, false>)
                               : (masked ? lstm_fwd_rec16_kernel<128, true> : lstm_fwd_rec16_kernel<128, false>);
    Rec16Params r;
    r.dirs = y.dirs; r.KK = y.KK[l]; r.KKI = y.KKI[l]; r.k_head_w = y.k_head_w;
    for (int d = 0; d < y.dirs; ++d) r.k_A[d] = y.k_A[l * y.dirs + d];
    const size_t lds = sizeof(float) * ((size_t)2 * SNN_TAPE_Q * y.H * 16 + (size_t)2 * y.H * 16);
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        snn_set_error("hipFuncSetAttribute failed for lstm_fwd_rec16_kernel");
        return SNN_ERR_HIP;
    }
    hipLaunchKernelGGL(fn, dim3(p.n_tiles * y.dirs * 2), dim3(y.H * 8), lds, st, p, r, l, head_part, gx);
    return SNN_OK;
}

// ============================================================================================================
// backward: gate gradients from the tape + dh_{t-1} = W_hh^T dG_t, half a tile per workgroup (bwd_rec_kernel<H, 0>)
// ============================================================================================================
template <int H>
__global__ void __launch_bounds__(H == 128 ? 512 : 256) bwd_rec16_kernel(const Bwd2Params p, int l)
{
    constexpr int NTH = H == 128 ? 512 : 256, NWV = NTH / 64;
    constexpr int MB = H / 16;          // 16-unit blocks of dh
    constexpr int KSPLIT = NWV / MB;    // waves sharing one block (split over the gate rows)
    constexpr int KW = H / KSPLIT;      // k-steps (of 4 gate rows) per wave
    // element-wise stage: a thread owns W consecutive curves of hidden unit jg (one 8- or 16-byte word per tape row)
    constexpr int W = H == 32 ? 2 : 4;
    constexpr int TPR = 16 / W;         // threads per unit row of the half tile
    static_assert(NTH / TPR == H, "one (unit, W curves) item per thread");
    static_assert(KSPLIT >= 1 && MB * KSPLIT == NWV, "one 16-unit block (or a K share of one) per wave");
    typedef float fW __attribute__((ext_vector_type(W)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dGl = smem;                  // [4H][16] gate gradients of the running step (B operand image)
    float* red = dGl + 4 * H * 16;      // [KSPLIT][H][16]
    const SnnLayout& y = p.lay;
    const int dirs = y.dirs;
    int b = blockIdx.x;
    const int half = b & 1; b >>= 1;
    const int tile = b / dirs, d = b - tile * dirs;
    const int ld = l * dirs + d;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int T = p.tile_T[tile];
    const bool last_layer = l == y.L - 1;
    const int c0 = (tid % TPR) * W, j = tid / TPR;
    const int cg = 16 * half + c0;
    int len[W];
    float inv_len[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        len[k] = p.len_sorted[tile * SNN_TILE + cg + k];
        inv_len[k] = 1.0f / (float)(len[k] > 0 ? len[k] : 1);
    }
    const int64_t ts = tape_step_floats(H);
    const float* tape_ld = p.tape + (int64_t)tile * p.tape_tile + (int64_t)ld * p.max_T * ts;
    const float* dfeat = p.dfeat + (int64_t)tile * y.F * 32;
    const float* dy = last_layer ? nullptr : p.dy_in + ((int64_t)tile * dirs + d) * p.max_T * H * 32;
    float* dGg = p.dG + ((int64_t)tile * dirs + d) * p.max_T * 4 * H * 32;

    // A operand: W_hh^T, rows = this wave's 16 hidden units, columns = its KW * 4 gate rows; from the image of bwd_prep
    const int mb = w / KSPLIT, ks = w - mb * KSPLIT;
    float wT[KW];
    {
        const int ju = 16 * mb + c;
        const float* A = p.WT + p.wt_hh[ld] + (int64_t)(ju >> 5) * 2 * H * 64 + (ju & 31);
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            const int R = ks * KW * 4 + 4 * kk + g;
            wT[kk] = A[(R >> 1) * 64 + (R & 1) * 32];
        }
    }
    fW zeroW, dc, dhr, dfm;
#pragma unroll
    for (int k = 0; k < W; ++k) zeroW[k] = 0.0f;
    dc = zeroW;
    dhr = y.pool == SNN_POOL_STANDARD ? *reinterpret_cast<const fW*>(dfeat + (j * (y.L * dirs) + ld) * 32 + cg) : zeroW;
    dfm = zeroW;
    if (last_layer && y.pool == SNN_POOL_MEAN) {
        const fW v = *reinterpret_cast<const fW*>(dfeat + (d * H + j) * 32 + cg);
#pragma unroll
        for (int k = 0; k < W; ++k) dfm[k] = v[k] * inv_len[k];
    }
    // Everything a step reads from the tape is fetched one step ahead by unconditional loads at clamped addresses: the
    // gates and dy of step t, and the cell state its predecessor left (c_{t-1}: the forget-gate gradient of step t needs
    // it).  A step's own c is the predecessor state fetched for the step processed before it, carried in a register.
    // The 6 loads of that fetch and the 4 stores of this step's gate gradients are dealt out over the MFMAs of the step
    // (vmem_piece): issued inside the element-wise stage, the 8 waves' memory instructions queued at the compute unit's
    // one address unit -- clock64 probes read 3 700 cycles for that stage and as many at the barrier behind it, beside
    // 8 200 cycles of MFMAs per SIMD and step (H = 128).
    fW n_ig, n_fg, n_gg, n_og, n_cp, n_dy, ccur;
    const float* dyp = dy ? dy : tape_ld;      // (no dy: any readable address, the values are not used)
    const float *tq_f = nullptr, *tc_f = nullptr, *dy_f = nullptr;
    auto fetch_setup = [&](int t) __attribute__((always_inline)) {
        const int tpp = d == 0 ? (t > 0 ? t - 1 : 0) : (t + 1 < T ? t + 1 : T - 1);
        tq_f = tape_ld + (int64_t)t * ts + j * 32 + cg;
        tc_f = tape_ld + (int64_t)tpp * ts + (4 * H + j) * 32 + cg;
        dy_f = dyp + (int64_t)t * H * 32 + j * 32 + cg;
    };
    auto fetch_piece = [&](int q) __attribute__((always_inline)) {
        if (q == 0) n_ig = *reinterpret_cast<const fW*>(tq_f + 0 * H * 32);
        else if (q == 1) n_fg = *reinterpret_cast<const fW*>(tq_f + 1 * H * 32);
        else if (q == 2) n_gg = *reinterpret_cast<const fW*>(tq_f + 2 * H * 32);
        else if (q == 3) n_og = *reinterpret_cast<const fW*>(tq_f + 3 * H * 32);
        else if (q == 4) n_cp = *reinterpret_cast<const fW*>(tc_f);
        else n_dy = *reinterpret_cast<const fW*>(dy_f);
    };
    if (T > 0) {
        const int tl = d == 0 ? T - 1 : 0;
        fetch_setup(tl);
#pragma unroll
        for (int q = 0; q < 6; ++q) fetch_piece(q);
        ccur = *reinterpret_cast<const fW*>(tape_ld + (int64_t)tl * ts + (4 * H + j) * 32 + cg);
    }

#ifdef SNN_PROBE_REC16   // clock64 probes of workgroup 0, wave 0: cycles per step in each part of the loop
    long long pc[6] = {0, 0, 0, 0, 0, 0};
#define PROBE(i) { const long long now_ = clock64(); pc[i] += now_ - pt_; pt_ = now_; }
#else
#define PROBE(i)
#endif
    for (int sb = 0; sb < T; ++sb) {
#ifdef SNN_PROBE_REC16
        long long pt_ = clock64();
#endif
        const int step = T - 1 - sb;
        const int t = d == 0 ? step : T - 1 - step;
        const int tp = d == 0 ? t - 1 : t + 1;
        const fW ig = n_ig, fg = n_fg, gg = n_gg, og = n_og, cn = ccur, cpv = n_cp, dyv = n_dy;
        ccur = cpv;                          // the state this step started from is the next processed step's own
        fetch_setup(step > 0 ? tp : t);      // (the last processed step fetches itself again: unused)
        fW dai = zeroW, daf = zeroW, dag = zeroW, dao = zeroW;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (t < len[k]) {
                float dh = dhr[k];
                if (dy) dh += dyv[k];
                if (last_layer && y.pool == SNN_POOL_MEAN) dh += dfm[k];
                const float cp = (step > 0 && tp < len[k]) ? cpv[k] : 0.0f;
                const float tc = snn_tanh(cn[k]);
                const float dct = dc[k] + dh * og[k] * (1.0f - tc * tc);
                dao[k] = dh * tc * og[k] * (1.0f - og[k]);
                dai[k] = dct * gg[k] * ig[k] * (1.0f - ig[k]);
                daf[k] = dct * cp * fg[k] * (1.0f - fg[k]);
                dag[k] = dct * ig[k] * (1.0f - gg[k] * gg[k]);
                dc[k] = dct * fg[k];
            }
        }
        *reinterpret_cast<fW*>(dGl + (0 * H + j) * 16 + c0) = dai;
        *reinterpret_cast<fW*>(dGl + (1 * H + j) * 16 + c0) = daf;
        *reinterpret_cast<fW*>(dGl + (2 * H + j) * 16 + c0) = dag;
        *reinterpret_cast<fW*>(dGl + (3 * H + j) * 16 + c0) = dao;
        float* o = dGg + (int64_t)t * 4 * H * 32 + j * 32 + cg;
        auto vmem_piece = [&](int q) __attribute__((always_inline)) {      // q < 6: next step's loads; then the 4 stores
            if (q < 6) fetch_piece(q);
            else if (q == 6) *reinterpret_cast<fW*>(o + 0 * H * 32) = dai;
            else if (q == 7) *reinterpret_cast<fW*>(o + 1 * H * 32) = daf;
            else if (q == 8) *reinterpret_cast<fW*>(o + 2 * H * 32) = dag;
            else *reinterpret_cast<fW*>(o + 3 * H * 32) = dao;
        };
        PROBE(0)
        __syncthreads();
        PROBE(1)
        {   // dh_{prev}[unit][curve] = sum_R W_hh[R][unit] dG[R][curve]: this wave's unit block mb, gate rows of split ks
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* bq = dGl + ks * KW * 64 + lane;      // row 4 kk + g, column c  ==  kk * 64 + lane
            constexpr int NVP = 10, EVERY = KW / NVP > 0 ? KW / NVP : 1;
            static_assert(KW >= NVP, "one memory instruction behind an MFMA at most");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wT[kk], bq[kk * 64], acc, 0, 0, 0);
                if (kk % EVERY == 0 && kk / EVERY < NVP) {
                    vmem_piece(kk / EVERY);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(ks * H + 16 * mb + 4 * g + i) * 16 + c] = acc[i];
        }
        PROBE(2)
        __syncthreads();
        PROBE(3)
        {
            fW s = zeroW;
#pragma unroll
            for (int k = 0; k < KSPLIT; ++k) {
                const fW rv = *reinterpret_cast<const fW*>(red + (k * H + j) * 16 + c0);
#pragma unroll
                for (int e = 0; e < W; ++e) s[e] += rv[e];
            }
#pragma unroll
            for (int e = 0; e < W; ++e)      // beyond the curve the state is frozen: the gradient passes through unchanged
                if (t < len[e]) dhr[e] = s[e];
        }
        PROBE(4)
    }
#ifdef SNN_PROBE_REC16
    if (blockIdx.x == 0 && tid == 0 && T > 0)
        printf("bwd_rec16<%d> %d steps: algebra %lld  barrier %lld  mfma %lld  barrier %lld  reduce %lld cycles per step\n", H, T,
               pc[0] / T, pc[1] / T, pc[2] / T, pc[3] / T, pc[4] / T);
#endif
#undef PROBE
}

int snn_launch_bwd_rec16(const Bwd2Params& p, int l, hipStream_t st)
{
    const SnnLayout& y = p.lay;
    const size_t lds = sizeof(float) * ((size_t)4 * y.H * 16 + (size_t)(y.H == 32 ? 2 : 1) * y.H * 16);
    const dim3 grid(p.n_tiles * y.dirs * 2);
    if (y.H == 32) hipLaunchKernelGGL(bwd_rec16_kernel<32>, grid, dim3(256), lds, st, p, l);
    else if (y.H == 64) hipLaunchKernelGGL(bwd_rec16_kernel<64>, grid, dim3(256), lds, st, p, l);
    else hipLaunchKernelGGL(bwd_rec16_kernel<128>, grid, dim3(512), lds, st, p, l);
    return SNN_OK;
}
