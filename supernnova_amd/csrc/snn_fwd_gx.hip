// snn_fwd_gx.hip -- the input half of a training forward, taken off the recurrent chain.
//
// A small batch is a latency chain: 128 curves are 4 tiles, i.e. 8 workgroups per layer that walk T dependent steps,
// and in the one-kernel forms (lstm_fwd_train_layer_kernel, lstm_fwd_big_kernel<TRAIN>) every step issues the input
// products W_ih x_t next to the recurrent ones although they depend on nothing the step before produced (H = 32:
// 12 + 16 / 32 + 16 MFMAs per step; H = 128: 3 x 256 per step in two passes).  Here the input half
//     Gx[tile][dir][t] = b_ih + b_hh + W_ih x_t          for ALL t
// is one launch over (tile, direction, chunk of steps, group of 4 row tiles) that uses the whole chip, and the
// recurrent kernels (lstm_fwd_rec_kernel in snn_fwd_small.h, lstm_fwd_big_rec_kernel in snn_forward_big.hip: a
// workgroup per (tile, direction)) start every step's accumulators from Gx[t] and add only W_hh h_{t-1}.  The products
// are the same MFMAs in the same order (bias, input k-steps ascending, then recurrent k-steps), so logits and tape are
// bit-identical to the one-kernel forms.
//
// Gx layout: [tile][dir][t][row tile tau < H/8][16][64] -- the C/D register image of v_mfma_f32_32x32x2_f32 for row tile
// tau (the 4 gates of hidden units 8 tau .. 8 tau + 7), what the recurrent kernels load with 16 coalesced dword loads.
// Replaces, together with the recurrent kernels, `output = rnn(packed_tensor)` of utils/training_utils.py:514
// (torch.nn.LSTM / torch._VF.lstm at vanilla_rnn.py:47, variational_rnn.py:84,244, bayesian_rnn.py:530-553).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "snn_kernels.h"
#include "snn_host.h"

struct GxParams {
    int H, dirs, l, KK, n_tchunks, TC, n_rg;
    int64_t k_A[2], k_bias[2];   // weight-image offsets of (layer l, direction d)
    float* gx;
};

// KKI: input k-steps of the layer (layer 0: snn_kki0_bucket(D); layers above: dirs * H / 2).
// FIRST: the input is the packed x image; otherwise the h rows of layer l - 1 on the tape.
// MASKED: variational input / between-layer masks are applied to the operand (as the staging of run_phase does).
template <int KKI, bool FIRST, bool MASKED>
__global__ void __launch_bounds__(256, KKI >= 64 ? 1 : 2) fwd_gx_kernel(const FwdParams p, const GxParams g)
{
    constexpr int KB = KKI < 32 ? KKI : 32;   // operand loads issued per batch, ahead of their MFMAs
    const int H = g.H, dirs = g.dirs, l = g.l, NWT = H / 8;
    int b = blockIdx.x;
    const int rg = b % g.n_rg; b /= g.n_rg;
    const int chunk = b % g.n_tchunks; b /= g.n_tchunks;
    const int d = b % dirs, tile = b / dirs;
    const int T = p.tile_T[tile];
    const int t0 = chunk * g.TC, t1 = min(T, t0 + g.TC);
    if (t0 >= t1) return;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 31, hf = lane >> 5;
    const int tau = rg * 4 + w;
    const MaskP& mk = p.mk;
    const bool masked = MASKED && mk.mode != SNN_MASK_NONE;
    const bool mid_step = masked && !FIRST && mk.mid_T > 1;
    // locked masks of the operand image, [2 KKI features][32 curves] in LDS, shared by the four waves
    // (a per-step between-layer mask -- torch.nn.LSTM's dropout in train mode -- is refilled at every step)
    __shared__ float mlds[MASKED ? 2 * KKI * 32 : 1];
    auto fill_masks = [&](int t) {
        for (int idx = threadIdx.x; idx < 2 * KKI * 32; idx += 256) {
            const int k = idx >> 5, o = p.order[tile * SNN_TILE + (idx & 31)];
            float m = 1.0f;
            if (masked && o >= 0) {
                if (FIRST) { if (k < p.D) m = mask_in(mk, 0, p.n_curves, p.D, o, k); }
                else m = mask_mid(mk, 0, p.L, p.n_curves, dirs * H, l - 1, t, o, k);
            }
            mlds[idx] = m;
        }
    };
    if (MASKED && !mid_step) {
        fill_masks(0);
        __syncthreads();
    }
    const bool active = tau < NWT;     // (H / 8 is a multiple of 4 for H = 32, 64, 128: always true there)

    float wx[KKI];
    {
        const float* A = p.wk + g.k_A[d] + (int64_t)(active ? tau : 0) * g.KK * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < KKI; ++kk) wx[kk] = A[kk * 64];
    }
    f32x16 binit;
    {
        const float* B = p.wk + g.k_bias[d] + ((active ? tau : 0) * 2 + hf) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) binit[i] = B[i] + B[NWT * 32 + i];
    }
    const int64_t ts = tape_step_floats(H);
    const float* in0 = FIRST ? p.xp + p.tile_off[tile] : p.tape + (int64_t)tile * p.tape_tile + (int64_t)(l - 1) * dirs * p.max_T * ts + 5 * H * 32;
    float* out0 = g.gx + ((((int64_t)tile * dirs + d) * p.max_T) * NWT + tau) * 1024 + lane;

    // The operands of batch b + 1 (KB k-steps; the next step's first batch after a step's last) are requested before the
    // MFMAs of batch b are issued: a wave then waits for memory once per chunk instead of once per batch (with one
    // batch in flight at a time the 256-wide layer-1 input of H = 128 ran at a sixth of the MFMA rate: 4 dependent
    // round trips per step beside 4 x 32 MFMAs).  NB batches per step; buffers alternate, statically: for an odd NB the
    // step loop is unrolled by two.
    constexpr int NB = (KKI + KB - 1) / KB;
    auto load_batch = [&](int t, int b, float (&bq)[KB]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int kk = b * KB + k;
            if (kk < KKI) {
                const float* src;
                if (FIRST) src = in0 + (int64_t)t * (2 * KKI * 32) + kk * 64 + lane;
                else {
                    const int dd = 2 * kk >= H ? 1 : 0;     // (dirs <= 2) rows 2 kk and 2 kk + 1 lie in the same direction block: H is even
                    src = in0 + ((int64_t)dd * p.max_T + t) * ts + (2 * kk - dd * H) * 32 + lane;
                }
                bq[k] = *src;
            }
        }
    };
    auto mfma_batch = [&](int b, const float (&bq)[KB], f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int kk = b * KB + k;
            if (kk < KKI) {
                float bv = bq[k];
                if (MASKED) bv = __fmul_rn(mlds[kk * 64 + lane], bv);   // feature 2 kk + hf of curve cl
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wx[kk], bv, acc, 0, 0, 0);
            }
        }
    };
    // one step: its NB batches from buffers (first, other, first, ...); the last batch prefetches step t + 1 into `nxt`
    auto step = [&](int t, float (&first)[KB], float (&other)[KB], float (&nxt)[KB]) __attribute__((always_inline)) {
        if (MASKED && mid_step) {       // block-uniform
            __syncthreads();
            fill_masks(t);
            __syncthreads();
        }
        f32x16 acc = binit;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float (&cur)[KB] = (b & 1) ? other : first;
            if (b + 1 < NB) load_batch(t, b + 1, (b & 1) ? first : other);
            else if (t + 1 < t1) load_batch(t + 1, 0, nxt);
            mfma_batch(b, cur, acc);
        }
        if (active) {
            float* o = out0 + (int64_t)t * NWT * 1024;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i * 64] = acc[i];
        }
    };
    float bqa[KB], bqb[KB];
    load_batch(t0, 0, bqa);
    if (NB % 2 == 0) {
        for (int t = t0; t < t1; ++t) step(t, bqa, bqb, bqa);
    } else {
        int t = t0;
        for (; t + 1 < t1; t += 2) {
            step(t, bqa, bqb, bqb);
            step(t + 1, bqb, bqa, bqa);
        }
        if (t < t1) step(t, bqa, bqb, bqb);
    }
}

typedef void (*gx_kernel_t)(const FwdParams, const GxParams);
template <int KKI>
static gx_kernel_t pick(bool first, bool masked)
{
    if (first) return masked ? fwd_gx_kernel<KKI, true, true> : fwd_gx_kernel<KKI, true, false>;
    return masked ? fwd_gx_kernel<KKI, false, true> : fwd_gx_kernel<KKI, false, false>;
}
static gx_kernel_t gx_kernel(int KKI, bool first, bool masked)
{
    switch (KKI) {
        case 12: return pick<12>(first, masked);
        case 13: return pick<13>(first, masked);
        case 24: return pick<24>(first, masked);
        case 38: return pick<38>(first, masked);
        case 39: return pick<39>(first, masked);
        case 16: return pick<16>(first, masked);     // layers above the first: dirs * H / 2
        case 32: return pick<32>(first, masked);
        case 64: return pick<64>(first, masked);
        case 128: return pick<128>(first, masked);
    }
    return nullptr;
}

bool snn_gx_supported(const SnnLayout& y)
{
    for (int l = 0; l < y.L; ++l)
        if (!gx_kernel(y.KKI[l], l == 0, false)) return false;
    return true;
}

int64_t snn_gx_floats(const SnnLayout& y, int n_tiles, int max_T) { return (int64_t)n_tiles * y.dirs * max_T * 4 * y.H * 32; }

// Launch the input half of layer l for all tiles; `gx` holds snn_gx_floats(...) floats.  cus: compute units of the device.
int snn_launch_gx(const SnnLayout& y, const FwdParams& p, int l, float* gx, int cus, hipStream_t st)
{
    const bool masked = p.mk.mode != SNN_MASK_NONE;
    gx_kernel_t fn = gx_kernel(y.KKI[l], l == 0, masked);
    if (!fn) { snn_set_error("no input-product kernel for %d k-steps", y.KKI[l]); return SNN_ERR_UNSUPPORTED; }
    GxParams g;
    g.H = y.H; g.dirs = y.dirs; g.l = l; g.KK = y.KK[l];
    g.n_rg = (y.H / 8 + 3) / 4;
    for (int d = 0; d < y.dirs; ++d) { g.k_A[d] = y.k_A[l * y.dirs + d]; g.k_bias[d] = y.k_bias[l * y.dirs + d]; }
    // chunks of steps: enough workgroups for ~2 per compute unit, (each workgroup loads its
    // rows of W_ih once and amortises them over the chunk; a chunk's steps are a chain of dependent memory round trips)
    const int64_t base = (int64_t)p.n_tiles * y.dirs * g.n_rg;
    int n_tchunks = (int)std::max<int64_t>(1, std::min<int64_t>(p.max_T, (2 * (int64_t)cus + base - 1) / base));
    g.TC = (p.max_T + n_tchunks - 1) / n_tchunks;
    g.n_tchunks = (p.max_T + g.TC - 1) / g.TC;
    g.gx = gx;
    hipLaunchKernelGGL(fn, dim3((unsigned)(base * g.n_tchunks)), dim3(256), 0, st, p, g);
    if (hipGetLastError() != hipSuccess) { snn_set_error("input-product kernel launch failed"); return SNN_ERR_HIP; }
    return SNN_OK;
}
