// snn_forward_wpt.hip -- GRU / tanh-RNN inference at H = 32 with ONE WAVE PER TILE (VanillaRNN, layer_type gru | rnn:
// training/vanilla_rnn.py:26-32).
//
// lstm_fwd_kernel<.., CELL> runs these cells on the LSTM's decomposition: wave w owns a 32-row tile = 4 row groups x the
// 8 hidden units 8w .. 8w+7.  The LSTM fills the four groups with its gates; the GRU needs r, z and the two halves of n
// (input part, hidden part: r multiplies only the latter), so a quarter of every tile's products multiply weights that
// are structurally zero; the tanh RNN uses one group of four.  They executed the LSTM's MFMA count for 3/4 and 1/4 of
// its algorithmic FLOPs (profiles/r3: 59 % / 21 % of the peak in algorithmic FLOPs).
//
// Here a row tile is 32 HIDDEN UNITS of one row group, so with H = 32 a single wave holds every row of a (layer,
// direction): the RNN is one accumulator chain of K = (I + H) / 2 MFMAs per step, the GRU four chains -- r and z over
// the whole K, n's input half over the I / 2 input k-steps only, n's hidden half over the 16 recurrent ones: 3 (I + H) / 2
// MFMAs instead of 4.  All gates of a (unit, curve) land in the same lane (the four tiles share the C/D layout), so the
// cell update needs no exchange between waves and the time loop has NO workgroup barrier: a wave redistributes its own
// h_t from C/D order to B-operand order through a wave-private LDS image (in-order LDS, a counter wait).  x_t and the
// layer-below outputs are B operands as they lie in memory ([row][32 curves]: the rows of a k-step are 64 contiguous
// floats), fetched one step ahead into registers.  A workgroup is 4 such waves on 4 consecutive tiles of one MC sample.
// The weights are gathered from the standard kernel image at the start of a phase (same values, no host change).
// Same products in the same order per output element: logits equal lstm_fwd_kernel<.., CELL>'s up to the head's
// summation order (16 units per lane here).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snn_kernels.h"
#include "snn_host.h"

namespace {
constexpr int H = 32, NWS = 4, KKH = 16;   // NWS: waves of the STANDARD image (8 hidden units each)

struct WaveCtx {
    int tile, s, T, lane, cl, hf, len, orig;
    const float* wks;
    float* ybase;      // this wave's y scratch: [nbuf][dirs][max_T][H*32]
};

// standard image -> unit-major A operand: row group g, k-step kk, this lane (k & 1 = lane >> 5, unit = lane & 31)
__device__ __forceinline__ float gather_w(const float* wks, int k_A, int KK, int g, int kk, int lane)
{
    const int r = lane & 31, kh = lane >> 5;
    return wks[k_A + ((r >> 3) * KK + kk) * 64 + kh * 32 + g * 8 + (r & 7)];
}
// b_ih + b_hh of row group g, C/D register i of half hf: register i <-> unit (i & 3) + 8 (i >> 2) + 4 hf
__device__ __forceinline__ float gather_bias(const float* wks, int k_bias, int g, int hf, int i)
{
    const int u = (i & 3) + 8 * (i >> 2) + 4 * hf;
    const float* B = wks + k_bias + ((u >> 3) * 2 + ((u & 7) >> 2)) * 16 + g * 4 + (u & 3);
    return B[0] + B[NWS * 32];
}

template <int KKI, int DIRS, int CELL, bool FIRST>
__device__ __forceinline__ void wpt_phase(const FwdParams& p, const WaveCtx& c, int l, int d, int k_A, int k_bias, int k_head_w,
                                          float* hl, float* lg, float* bl)
{
    constexpr int KK = KKI + KKH;
    constexpr int NG = CELL == 1 ? 4 : 1;            // accumulator chains: GRU r, z, n (input half), n (hidden half); RNN one
    const FwdParams& y = p;
    const int ld = l * DIRS + d;
    const bool last_layer = l == y.L - 1;
    const int T = c.T;

    // ---- weights -> registers
    float wx[CELL == 1 ? 3 : 1][KKI];               // input k-steps of r, z, n_x (RNN: the one group)
    float wh[CELL == 1 ? 3 : 1][KKH];               // recurrent k-steps of r, z, n_h
#pragma unroll
    for (int q = 0; q < (CELL == 1 ? 3 : 1); ++q) {
#pragma unroll
        for (int kk = 0; kk < KKI; ++kk) wx[q][kk] = gather_w(c.wks, k_A, KK, q, kk, c.lane);               // groups 0, 1, 2
#pragma unroll
        for (int kk = 0; kk < KKH; ++kk) wh[q][kk] = gather_w(c.wks, k_A, KK, q == 2 ? 3 : q, KKI + kk, c.lane);   // groups 0, 1, 3
    }
    // bias images in C/D register order, parked in the wave's LDS (re-read per step: broadcast 16-byte reads)
    if (c.cl < 16) {
#pragma unroll
        for (int g = 0; g < NG; ++g) bl[(g * 2 + c.hf) * 16 + c.cl] = gather_bias(c.wks, k_bias, g, c.hf, c.cl);
    }

    // ---- B operands of the input k-steps: x_t rows / the layer below's outputs, as they lie in memory.  One base address
    // per step (and per direction of the layer below); the k-steps are immediate offsets of it.
    const float* in0 = FIRST ? p.xp + p.tile_off[c.tile] : c.ybase + (int64_t)((l - 1) % p.nbuf) * p.y_buf_floats;
    constexpr int NXB = FIRST ? 1 : DIRS;           // base pointers per step
    auto xbase = [&](int t, const float* (&xb)[NXB]) __attribute__((always_inline)) {
        if (FIRST) xb[0] = in0 + (int64_t)t * (2 * KKI * 32) + c.lane;
        else {
#pragma unroll
            for (int dd = 0; dd < NXB; ++dd) xb[dd] = in0 + ((int64_t)dd * p.max_T + t) * (H * 32) + c.lane;
        }
    };
    auto xload = [&](const float* const (&xb)[NXB], int kk) __attribute__((always_inline)) -> float {
        if (FIRST) return xb[0][kk * 64];
        return xb[kk / KKH][(kk % KKH) * 64];         // (KKI = DIRS * 16: k-step kk is row pair kk % 16 of direction kk / 16)
    };
    float* yout = last_layer ? nullptr : c.ybase + (int64_t)(l % p.nbuf) * p.y_buf_floats + (int64_t)d * p.max_T * (H * 32);

    float hst[16], hsum[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        hst[i] = hsum[i] = 0.0f;
        hl[((i & 3) + 8 * (i >> 2) + 4 * c.hf) * 32 + c.cl] = 0.0f;
    }
    float bx[KKI];
    {
        const float* xb[NXB];
        xbase(d == 0 ? 0 : T - 1, xb);
#pragma unroll
        for (int kk = 0; kk < KKI; ++kk) bx[kk] = xload(xb, kk);
    }
    for (int step = 0; step < T; ++step) {
        const int t = d == 0 ? step : T - 1 - step;
        const int tn = step + 1 < T ? (d == 0 ? t + 1 : t - 1) : t;      // (after the last step: this one again, unused)
        f32x16 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = *reinterpret_cast<const f32x16*>(bl + (g * 2 + c.hf) * 16);
        // input k-steps (independent of h: they run while the h image of the step before settles); each operand register
        // is refilled for the next step as soon as its MFMAs are issued
        const float* xbn[NXB];
        xbase(tn, xbn);
        // GRU: the weights (144 registers in layer 1), four accumulators and the operand registers do not fit 256 VGPRs; the
        // compiler then parks weights in AGPRs and copies each one to a VGPR in front of its MFMA.  The z chain and n's hidden
        // half read their weights from AGPRs directly (inline asm, as in snn_forward_big.hip).
#pragma unroll
        for (int kk = 0; kk < KKI; ++kk) {
            const float b = bx[kk];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wx[0][kk], b, acc[0], 0, 0, 0);
            if (CELL == 1) {
                asm volatile(SNN_AGPR_MFMA : "+a"(acc[1]) : "a"(wx[1][kk]), "v"(b));
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wx[2][kk], b, acc[2], 0, 0, 0);
            }
            bx[kk] = xload(xbn, kk);
        }
        // recurrent k-steps: h_{t-1} from the wave's own LDS image, rows 2kk, 2kk+1 = 64 contiguous floats
#pragma unroll
        for (int kk = 0; kk < KKH; ++kk) {
            const float b = hl[kk * 64 + c.lane];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[0][kk], b, acc[0], 0, 0, 0);
            if (CELL == 1) {
                asm volatile(SNN_AGPR_MFMA : "+a"(acc[1]) : "a"(wh[1][kk]), "v"(b));
                asm volatile(SNN_AGPR_MFMA : "+a"(acc[3]) : "a"(wh[2][kk]), "v"(b));
            }
        }
        if (CELL == 1) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // (wait states between a 16-pass MFMA and a vector read of its result)
        const bool valid = t < c.len;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float hv;
            if (CELL == 1) {      // torch.nn.GRU: n = tanh(W_in x + b_in + r (W_hn h + b_hn)),  h = (1 - z) n + z h
                const float rg = snn_sigmoid(acc[0][i]);
                const float zg = snn_sigmoid(acc[1][i]);
                const float ng = snn_tanh(fmaf(rg, acc[3][i], acc[2][i]));
                hv = fmaf(zg, hst[i] - ng, ng);
            } else {
                hv = snn_tanh(acc[0][i]);
            }
            hst[i] = valid ? hv : hst[i];
            const float outv = valid ? hv : 0.0f;
            const int u = (i & 3) + 8 * (i >> 2) + 4 * c.hf;
            hl[u * 32 + c.cl] = hst[i];
            if (last_layer) hsum[i] += outv;
            else yout[(int64_t)t * (H * 32) + u * 32 + c.cl] = outv;
        }
    }

    // ---- head contribution of this phase
    if (y.pool == SNN_POOL_STANDARD || last_layer) {
        const float* Wo = c.wks + k_head_w;
        const float flen = (float)(c.len > 0 ? c.len : 1);
        for (int q = 0; q < y.C; ++q) {
            float a = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int u = (i & 3) + 8 * (i >> 2) + 4 * c.hf;
                const float feat = y.pool == SNN_POOL_MEAN ? __fdiv_rn(hsum[i], flen) : hst[i];
                const int fidx = y.pool == SNN_POOL_MEAN ? d * H + u : u * (y.L * DIRS) + ld;
                a = fmaf(Wo[q * y.F + fidx], feat, a);
            }
            a += __shfl_xor(a, 32);
            if (c.hf == 0) lg[q * 32 + c.cl] += a;
        }
    }
    // the next phase reads this one's y rows (written by this wave, other lanes): make them visible
    __threadfence_block();
}

template <int KKI0, int DIRS, int CELL>
__global__ void __launch_bounds__(256, CELL == 1 ? 1 : 2) wpt_fwd_kernel(const FwdParams p)
{
    constexpr int KKI1 = DIRS * H / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdParams& y = p;
    constexpr int BLK0 = NWS * (KKI0 + KKH) * 64 + 2 * NWS * 32;
    constexpr int BLK1 = NWS * (KKI1 + KKH) * 64 + 2 * NWS * 32;
    const int k_head_w = DIRS * BLK0 + (y.L - 1) * DIRS * BLK1;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* hl = smem + w * (H * 32);                         // [4 waves][H*32]   wave-private h image
    float* bl = smem + 4 * (H * 32) + w * 128;               // [4 waves][4 groups][2][16] bias images of the running phase
    float* lg = smem + 4 * (H * 32) + 4 * 128 + w * (y.C * 32);   // [4 waves][C][32]  wave-private logits
    int* item_slot = reinterpret_cast<int*>(smem + 4 * (H * 32) + 4 * 128 + 4 * y.C * 32);

    WaveCtx c;
    c.lane = threadIdx.x & 63;
    c.cl = c.lane & 31;
    c.hf = c.lane >> 5;
    // per-workgroup scratch: 4 per-wave blocks of [nbuf][dirs][max_T][H*32]
    c.ybase = p.yscratch + ((int64_t)blockIdx.x * 4 + w) * p.nbuf * p.y_buf_floats;
    const int n_groups = (p.n_tiles + 3) >> 2;
    const int64_t n_items = (int64_t)n_groups * p.n_samples;
    for (;;) {
        if (threadIdx.x == 0) *item_slot = (int)atomicAdd(p.counter, 1u);
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(*item_slot);
        __syncthreads();            // every wave has read the slot before thread 0 draws again
        if (item >= n_items) break;
        const int grp = item / p.n_samples;
        c.s = item - grp * p.n_samples;
        c.tile = 4 * grp + w;
        if (c.tile >= p.n_tiles) continue;      // (a group's missing tiles: the wave waits at the next draw)
        c.T = __builtin_amdgcn_readfirstlane(p.tile_T[c.tile]);
        const int pos = c.tile * SNN_TILE + c.cl;
        c.len = p.len_sorted[pos];
        c.orig = p.order[pos];
        c.wks = p.wk + (int64_t)c.s * p.wk_stride;
        for (int idx = c.lane; idx < y.C * 32; idx += 64) lg[idx] = 0.0f;

        for (int l = 0; l < y.L; ++l)
            for (int d = 0; d < DIRS; ++d) {
                if (l == 0) {
                    const int kA = d * BLK0;
                    wpt_phase<KKI0, DIRS, CELL, true>(p, c, l, d, kA, kA + NWS * (KKI0 + KKH) * 64, k_head_w, hl, lg, bl);
                } else {
                    const int kA = DIRS * BLK0 + ((l - 1) * DIRS + d) * BLK1;
                    wpt_phase<KKI1, DIRS, CELL, false>(p, c, l, d, kA, kA + NWS * (KKI1 + KKH) * 64, k_head_w, hl, lg, bl);
                }
            }

        // ---- head bias + softmax, one lane per curve
        if (c.hf == 0 && c.orig >= 0) {
            const float* bo = c.wks + k_head_w + y.C * y.F;
            float mx = -INFINITY;
            for (int q = 0; q < y.C; ++q) mx = fmaxf(mx, lg[q * 32 + c.cl] + bo[q]);
            float z = 0.0f;
            for (int q = 0; q < y.C; ++q) z += expf(lg[q * 32 + c.cl] + bo[q] - mx);
            const int64_t o = (int64_t)c.s * p.os_s + (int64_t)c.orig * p.os_n;
            for (int q = 0; q < y.C; ++q) {
                const float v = lg[q * 32 + c.cl] + bo[q];
                if (p.out_logits) p.out_logits[o + q] = v;
                if (p.out_probs) p.out_probs[o + q] = __fdiv_rn(expf(v - mx), z);
            }
        }
    }
}
}  // namespace

// ---- registry hook used by snn_forward.hip: un-masked inference, H = 32, cell 1 (GRU) / 2 (tanh RNN) --------------
typedef void (*fwd_kernel_t)(const FwdParams);
fwd_kernel_t snn_wpt_kernel(int Hh, int KKI0, int dirs, int cell)
{
    if (Hh != 32 || (cell != 1 && cell != 2)) return nullptr;
#define SNN_WPT_CASE(K)                                                                                         \
    if (KKI0 == K) {                                                                                            \
        if (dirs == 2) return cell == 1 ? wpt_fwd_kernel<K, 2, 1> : wpt_fwd_kernel<K, 2, 2>;                      \
        return cell == 1 ? wpt_fwd_kernel<K, 1, 1> : wpt_fwd_kernel<K, 1, 2>;                                    \
    }
    SNN_WPT_CASE(12)
    SNN_WPT_CASE(13)
    SNN_WPT_CASE(24)
    SNN_WPT_CASE(38)
    SNN_WPT_CASE(39)
#undef SNN_WPT_CASE
    return nullptr;
}
