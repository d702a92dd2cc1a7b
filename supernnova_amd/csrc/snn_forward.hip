// snn_forward.hip -- gfx950 kernels + C ABI for the inference hot path:
//   pack_x_kernel      light curves -> 32-curve tiles in MFMA B-operand order
//   bbb_sample_kernel  w = mu + eps*softplus(rho) into kernel layout, + KL sums
//   lstm_fwd_kernel    persistent LSTM stack + pooling + head + softmax
// See include/snn_hip.h for the contract and the reference call sites replaced.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <numeric>
#include <vector>

#include "../../include/snn_hip.h"
#include "snn_kernels.h"
#include "snn_host.h"

// ============================================================================
// pack_x: one workgroup per tile (split-precision images: a thread per 16-byte word; fp32 image: LDS transposition)
// ============================================================================
// precision 1: xp3[tile][t][XB][3 pieces][64 lanes][8 bf16], lane (curve = lane & 31, khalf = lane >> 5) holds features
// 16 b + 8 khalf + e -- the B operand of v_mfma_f32_32x32x16_bf16, three times (hi, mid, lo)
// precision 2 (NPC == 2): the same image with two fp16 pieces (split2)
template <int NPC>
__global__ void __launch_bounds__(256) pack_x3_kernel(int D, int Dp, int XB, const float* __restrict__ x,
                                                     int64_t stride_t, const int64_t* __restrict__ step_off, int64_t stride_n,
                                                     const int64_t* __restrict__ curve_off,
                                                     const int32_t* __restrict__ order,
                                                     const int32_t* __restrict__ len_sorted,
                                                     const int32_t* __restrict__ tile_T,
                                                     const int64_t* __restrict__ tile_off,
                                                     __bf16* __restrict__ xp, int64_t xp_words)
{
    // one thread per 16-byte output word: 8 consecutive features of one curve at one step, read as 32 contiguous
    // bytes, written as NPC full words (lanes of a wave write consecutive words)
    const int tile = blockIdx.x;
    const int T = tile_T[tile];
    const int64_t step_u4 = (int64_t)XB * NPC * 64;
    const int64_t step0 = tile_off[tile] / ((int64_t)Dp * SNN_TILE);
    if ((step0 + T) * step_u4 * 4 > xp_words) return;   // caller's buffer is smaller than the plan: never write past it
    uint4* dst = reinterpret_cast<uint4*>(xp) + step0 * step_u4;
    for (int idx = threadIdx.x; idx < T * XB * 64; idx += blockDim.x) {
        const int t = idx / (XB * 64), r = idx - t * (XB * 64), b = r >> 6, lp = r & 63, c = lp & 31;
        const int pos = tile * SNN_TILE + c;
        const int orig = order[pos];
        const bool live = orig >= 0 && t < len_sorted[pos];
        const int on = live ? orig : 0;
        const float* src = x + (step_off ? step_off[t] : (int64_t)t * stride_t) + (curve_off ? (live ? curve_off[on] : 0) : (int64_t)on * stride_n);
        const int k0 = 16 * b + 8 * (lp >> 5);
        uint4* o = dst + (int64_t)t * step_u4 + (int64_t)b * NPC * 64 + lp;
        if (NPC == 3) {
            alignas(16) __bf16 h[8], m[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) split3((live && k0 + e < D) ? src[k0 + e] : 0.0f, h[e], m[e], l[e]);
            o[0] = *reinterpret_cast<const uint4*>(h);
            o[64] = *reinterpret_cast<const uint4*>(m);
            o[128] = *reinterpret_cast<const uint4*>(l);
        } else {
            alignas(16) _Float16 h[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) split2((live && k0 + e < D) ? src[k0 + e] : 0.0f, h[e], l[e]);
            o[0] = *reinterpret_cast<const uint4*>(h);
            o[64] = *reinterpret_cast<const uint4*>(l);
        }
    }
}

// Transposition through LDS: the rows of the source are read along their feature axis (a curve's D floats are
// contiguous, so a wave touches ~3 rows instead of 64), the tile image is written along its curve axis (32 curves
// contiguous).  PACK_TS time steps per barrier pair; dynamic LDS = PACK_TS * Dp * 33 floats.
#define PACK_TS 4
__global__ void __launch_bounds__(256) pack_x_kernel(int D, int Dp, const float* __restrict__ x,
                                                    int64_t stride_t, const int64_t* __restrict__ step_off, int64_t stride_n,
                                                    const int64_t* __restrict__ curve_off,
                                                    const int32_t* __restrict__ order,
                                                    const int32_t* __restrict__ len_sorted,
                                                    const int32_t* __restrict__ tile_T,
                                                    const int64_t* __restrict__ tile_off,
                                                    float* __restrict__ xp, int64_t xp_words)
{
    extern __shared__ float pk_lds[];
    __shared__ int64_t s_base[SNN_TILE];
    __shared__ int s_len[SNN_TILE];
    const int tile = blockIdx.x;
    const int T = tile_T[tile];
    float* dst = xp + tile_off[tile];
    if (tile_off[tile] + (int64_t)T * SNN_TILE * Dp > xp_words) return;   // buffer smaller than the plan: never write past it
    if (threadIdx.x < SNN_TILE) {
        const int pos = tile * SNN_TILE + threadIdx.x;
        const int orig = order[pos];
        s_len[threadIdx.x] = orig >= 0 ? len_sorted[pos] : 0;
        s_base[threadIdx.x] = orig >= 0 ? (curve_off ? curve_off[orig] : (int64_t)orig * stride_n) : 0;
    }
    __syncthreads();
    const int row = SNN_TILE * D, img = SNN_TILE * Dp;
    // gridDim.y > 1 for small batches (a handful of tiles is a latency chain otherwise): the tile's steps are dealt out
    for (int t0 = blockIdx.y * PACK_TS; t0 < T; t0 += PACK_TS * gridDim.y) {
        const int nt = min(PACK_TS, T - t0);
        for (int e = threadIdx.x; e < nt * row; e += blockDim.x) {
            const int tt = e / row, r = e - tt * row;
            const int c = r / D, k = r - c * D;
            const int t = t0 + tt;
            pk_lds[(tt * Dp + k) * (SNN_TILE + 1) + c] = t < s_len[c] ? x[s_base[c] + (step_off ? step_off[t] : (int64_t)t * stride_t) + k] : 0.0f;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nt * img; e += blockDim.x) {
            const int tt = e / img, r = e - tt * img;
            const int k = r >> 5, c = r & 31;
            dst[(int64_t)(t0 + tt) * img + r] = k < D ? pk_lds[(tt * Dp + k) * (SNN_TILE + 1) + c] : 0.0f;
        }
        __syncthreads();
    }
}

// ============================================================================
// Bayes-by-Backprop sampling + KL (also the plain re-layout when rho == NULL)
// ============================================================================
__device__ __forceinline__ float softplus_f(float r) { return r > 20.0f ? r : log1pf(expf(r)); }

__device__ __forceinline__ double block_sum(double v, double* sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
    return s;
}

// the three log-probability sums of compute_KL in one pass: same shuffle tree and the same wave order as block_sum
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); c += __shfl_down(c, o); }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (int)(blockDim.x >> 6);
    if (lane == 0) { sh[w] = a; sh[16 + w] = b; sh[32 + w] = c; }
    __syncthreads();
    double sa = 0, sb = 0, sc = 0;
    for (int i = 0; i < nw; ++i) { sa += sh[i]; sb += sh[16 + i]; sc += sh[32 + i]; }
    a = sa; b = sb; c = sc;
}

// chunks a tensor's elements are dealt over (one workgroup each): a training step samples ONE weight set, and with a
// workgroup per tensor the 8192 elements of the largest one (Philox + Box-Muller + softplus + three log-densities each)
// were a 26 us chain on one compute unit
#define SNN_BBB_NCH 8
__host__ __device__ inline int64_t kl_final_floats(int n_tensors, int S) { return ((int64_t)S * n_tensors * 3 + 1) & ~(int64_t)1; }

__global__ void __launch_bounds__(256) bbb_sample_kernel(SnnLayout y, const float* __restrict__ mu,
                                                        const float* __restrict__ rho,
                                                        const float* __restrict__ eps, int mode,
                                                        uint64_t seed, uint64_t offset, snn_prior prior,
                                                        float* __restrict__ wk, float* __restrict__ kl_terms,
                                                        int64_t mu_stride, int S)
{
    __shared__ double sh[48];
    const int tau = blockIdx.x, s = blockIdx.y, ch = blockIdx.z;
    mu += (int64_t)s * mu_stride;   // snn_prepare_weights_multi: one canonical weight set per sample
    const TensorGeom g = tensor_geom(y, tau);
    const int64_t Ps = y.P - y.C;
    float* dst = wk + (int64_t)s * y.PK;
    const bool sampled = rho != nullptr && g.kind != 5;
    const bool want_kl = kl_terms != nullptr && sampled;
    const float lsq = 0.91893853320467274f;  // log(sqrt(2 pi))
    const float s1 = g.kind == 4 ? prior.sigma1_out : prior.sigma1;
    const float s2 = g.kind == 4 ? prior.sigma2_out : prior.sigma2;
    const float v1 = s1 * s1, v2 = s2 * s2, l1 = logf(s1), l2 = logf(s2);
    double q = 0, p1 = 0, p2 = 0;
    const int n = (int)g.n, per = (n + SNN_BBB_NCH - 1) / SNN_BBB_NCH;
    const int i1 = min(n, (ch + 1) * per);
    if (tau == y.n_tensors && ch == 0 && threadIdx.x < 4) {      // alignment words behind the head bias
        const int64_t at = y.k_head_b + y.C + threadIdx.x;
        if (at < y.PK) dst[at] = 0.0f;
    }
    for (int i = ch * per + threadIdx.x; i < i1; i += blockDim.x) {
        const int64_t p = g.cbase + i;
        const float m = mu[p];
        float w = m;
        if (sampled) {
            const float sg = softplus_f(rho[p]) + 1e-5f;
            if (mode != 1) {
                const float e = mode == 0 ? eps[(int64_t)s * Ps + p]
                                          : philox_normal(seed, offset + s, SNN_STREAM_BBB, (uint64_t)p);
                w = __fadd_rn(m, __fmul_rn(e, sg));  // two roundings, as torch's mul then add
            }
            if (want_kl) {
                // torch.distributions.Normal.log_prob, element-wise in fp32 (bayesian_rnn.py:796-811)
                const float d = w - m;
                const float var = sg * sg;
                q += (double)(__fdiv_rn(-(d * d), 2.0f * var) - logf(sg) - lsq);
                const float x2 = w * w;
                p1 += (double)(__fdiv_rn(-x2, 2.0f * v1) - l1 - lsq);
                p2 += (double)(__fdiv_rn(-x2, 2.0f * v2) - l2 - lsq);
            }
        }
        if (y.prec == 1 && g.kind <= 1) {
            __bf16 wh, wm, wl;
            split3(w, wh, wm, wl);
            __bf16* d3 = reinterpret_cast<__bf16*>(dst) + kernel_index3(y, g, i);
            d3[0] = wh;
            d3[64 * 8] = wm;
            d3[2 * 64 * 8] = wl;
        } else if (y.prec == 2 && g.kind <= 1) {
            // the row's gate factor goes into the weight (one fp32 rounding), then the two fp16 pieces
            _Float16 wh, wl;
            split2(__fmul_rn(w, h2_gate_scale((i / g.cols) / y.H)), wh, wl);
            _Float16* d2 = reinterpret_cast<_Float16*>(dst) + kernel_index3(y, g, i);
            d2[0] = wh;
            d2[64 * 8] = wl;
        } else {
            dst[kernel_index(y, g, i)] = w;
        }
    }
    if (want_kl) {      // this chunk's three log-probability sums (f64), added up per tensor by kl_finish_kernel
        block_sum3(q, p1, p2, sh);
        if (threadIdx.x == 0) {
            double* part = reinterpret_cast<double*>(kl_terms + kl_final_floats(y.n_tensors, S)) +
                           (((int64_t)s * y.n_tensors + tau) * SNN_BBB_NCH + ch) * 3;
            part[0] = q; part[1] = p1; part[2] = p2;
        }
    }
}

// per sample: chunk sums -> per-tensor KL term and mixture responsibilities (sum-then-logsumexp, bayesian_rnn.py:810-813),
// then kl[s] = fp32 running sum over tensors in reference order (bayesian_rnn.py:501-528, 137)
__global__ void __launch_bounds__(64) kl_finish_kernel(int n_tensors, int S, snn_prior prior, float* __restrict__ terms,
                                                      float* __restrict__ kl)
{
    __shared__ float sh_term[64];
    const int s = blockIdx.x, tau = threadIdx.x;
    if (tau < n_tensors) {
        const double* part = reinterpret_cast<const double*>(terms + kl_final_floats(n_tensors, S)) +
                             ((int64_t)s * n_tensors + tau) * SNN_BBB_NCH * 3;
        double q = 0, p1 = 0, p2 = 0;
        for (int ch = 0; ch < SNN_BBB_NCH; ++ch) { q += part[ch * 3]; p1 += part[ch * 3 + 1]; p2 += part[ch * 3 + 2]; }
        const float m1 = (float)p1 + logf(prior.pi), m2 = (float)p2 + logf(1.0f - prior.pi);
        const float mx = fmaxf(m1, m2);
        const float lse = mx + logf(expf(m1 - mx) + expf(m2 - mx));
        const float term = (float)q - lse;
        float* o = terms + ((int64_t)s * n_tensors + tau) * 3;
        o[0] = term;
        o[1] = expf(m1 - lse);  // mixture responsibilities: d logsumexp / d m_k (used by snn_bbb_backward)
        o[2] = expf(m2 - lse);
        sh_term[tau] = term;
    }
    __syncthreads();
    if (tau == 0) {
        float acc = 0.0f;
        for (int t = 0; t < n_tensors; ++t) acc = __fadd_rn(acc, sh_term[t]);
        kl[s] = acc;
    }
}

// Head of the per-layer training forward (lstm_fwd_train_layer_kernel): per curve, the per-wave partial logits of the
// phases that feed the head, added in the persistent kernel's order ((l, d) outer, wave inner, from 0.0f), then bias
// (+ softmax): the same fp32 additions, so the same bits.
__global__ void __launch_bounds__(256) head_combine_kernel(const FwdParams p, int NW, int dirs, int64_t k_head_b,
                                                          const float* __restrict__ head_part)
{
    // 8 groups of 32 threads: group qs sums the partials of classes qs, qs + 8, ... (loads four at a time, additions in
    // the fixed order ld, then wave: what the persistent kernel's in-LDS reduction does), then the first group applies
    // the softmax.  One thread per curve doing all of it was 107 us at C = 14 with 16 partials per phase.
    extern __shared__ float lgs[];      // [C][32]
    const int tile = blockIdx.x, cl = threadIdx.x & 31, qs = threadIdx.x >> 5;
    const int orig = p.order[tile * SNN_TILE + cl];
    const int n_ld = p.L * dirs;     // NW: per-wave partials per phase (H / 8 waves; 4 in the H = 128 kernels)
    const float* bo = p.wk + k_head_b;
    for (int q = qs; q < p.C; q += 8) {
        float a = 0.0f;
        for (int ld = (p.pool == SNN_POOL_STANDARD ? 0 : n_ld - dirs); ld < n_ld; ++ld) {
            const float* pp = head_part + ((int64_t)tile * n_ld + ld) * (NW * p.C * 32) + q * 32 + cl;
            int w = 0;
            for (; w + 4 <= NW; w += 4) {
                const float v0 = pp[(int64_t)w * p.C * 32], v1 = pp[(int64_t)(w + 1) * p.C * 32];
                const float v2 = pp[(int64_t)(w + 2) * p.C * 32], v3 = pp[(int64_t)(w + 3) * p.C * 32];
                a += v0; a += v1; a += v2; a += v3;
            }
            for (; w < NW; ++w) a += pp[(int64_t)w * p.C * 32];
        }
        lgs[q * 32 + cl] = a + bo[q];
    }
    __syncthreads();
    if (qs != 0 || orig < 0) return;
    const int64_t o = (int64_t)orig * p.os_n;
    float mx = -INFINITY;
    for (int q = 0; q < p.C; ++q) {
        const float lg = lgs[q * 32 + cl];
        mx = fmaxf(mx, lg);
        if (p.out_logits) p.out_logits[o + q] = lg;
    }
    if (!p.out_probs) return;
    float z = 0.0f;
    for (int q = 0; q < p.C; ++q) z += expf(lgs[q * 32 + cl] - mx);
    for (int q = 0; q < p.C; ++q) p.out_probs[o + q] = __fdiv_rn(expf(lgs[q * 32 + cl] - mx), z);
}

// ============================================================================
// host side
// ============================================================================
// name and launch geometry of the forward kernel this thread dispatched last (reports and benchmarks name what ran)
static thread_local char g_last_kernel[160] = "";
extern "C" const char* snn_last_forward_kernel(void) { return g_last_kernel; }

// ---- forward kernel registry ----------------------------------------------------
typedef void (*fwd_kernel_t)(const FwdParams);
typedef void (*fwd_layer_kernel_t)(const FwdParams, int, float*);
typedef void (*fwd_rec_kernel_t)(const FwdParams, int, float*, const float*);
struct FwdSmallEntry { int H, KKI0, dirs; fwd_kernel_t fn, fn_masked, fn_train, fn_gru, fn_rnn, fn_gru_train, fn_rnn_train; fwd_layer_kernel_t fn_train_layer; fwd_rec_kernel_t fn_rec[3]; };
struct FwdEntry { int H, KKI0, dirs; fwd_kernel_t fn, fn_masked, fn_train; bool big; fwd_layer_kernel_t fn_train_layer; fwd_rec_kernel_t fn_rec; };
fwd_rec_kernel_t snn_big_rec_kernel(int H, int KKI0, int dirs);   // snn_forward_big.hip
// snn_fwd_gx.hip: the input half of a training forward as one launch over the chip
bool snn_gx_supported(const SnnLayout& y);
int64_t snn_gx_floats(const SnnLayout& y, int n_tiles, int max_T);
int snn_launch_gx(const SnnLayout& y, const FwdParams& p, int l, float* gx, int cus, hipStream_t st);
// snn_rec16.hip: the recurrent half on half tiles (16 curves per workgroup), LSTM, H = 32 / 64 / 128
bool snn_rec16_supported(const SnnLayout& y);
int snn_launch_fwd_rec16(const SnnLayout& y, const FwdParams& p, int l, float* head_part, const float* gx, hipStream_t st);
const FwdSmallEntry* snn_fwd_small_h32(int KKI0, int dirs);   // snn_fwd_h32.hip
const FwdSmallEntry* snn_fwd_small_h64(int KKI0, int dirs);   // snn_fwd_h64.hip
fwd_kernel_t snn_big_kernel(int H, int KKI0, int dirs, bool train, bool masked);  // snn_forward_big.hip
fwd_kernel_t snn_pair_kernel(int H, int KKI0, int dirs, bool masked);      // snn_forward_pair.hip
fwd_kernel_t snn_wpt_kernel(int H, int KKI0, int dirs, int cell);          // snn_forward_wpt.hip (GRU / tanh RNN, one wave per tile)
typedef void (*fwd_b3_kernel_t)(const FwdParams, const B3Offsets);
fwd_b3_kernel_t snn_b3_kernel(int H, int XB0, int dirs);      // snn_forward_b3.hip (SNN_PREC_BF16X3)
fwd_b3_kernel_t snn_b3p_kernel(int H, int XB0, int dirs);     // snn_forward_b3p.hip (same, two tiles per workgroup)
fwd_b3_kernel_t snn_h2_kernel(int H, int XB0, int dirs, bool masked);   // snn_forward_h2.hip (SNN_PREC_F16X2)
static const FwdEntry* find_fwd(const SnnLayout& y)
{
    static thread_local FwdEntry ent;
    if (y.prec != 0) return nullptr;   // the bf16 x 3 kernels have their own launcher
    const FwdSmallEntry* e = y.H == 32 ? snn_fwd_small_h32(y.KKI[0], y.dirs)
                           : y.H == 64 ? snn_fwd_small_h64(y.KKI[0], y.dirs) : nullptr;
    if (e && y.cell != 0) {   // GRU / tanh RNN: un-masked inference kernel + the training kernel (which takes masks)
        ent = FwdEntry{e->H, e->KKI0, e->dirs, y.cell == 1 ? e->fn_gru : e->fn_rnn, nullptr,
                       y.cell == 1 ? e->fn_gru_train : e->fn_rnn_train, false, nullptr, e->fn_rec[y.cell]};
        return &ent;
    }
    if (y.cell != 0) return nullptr;
    if (e) {
        ent = FwdEntry{e->H, e->KKI0, e->dirs, e->fn, e->fn_masked, e->fn_train, false, e->fn_train_layer, e->fn_rec[0]};
        return &ent;
    }
    // hidden sizes whose weights do not fit the register file: two-pass kernel
    if (fwd_kernel_t fn = snn_big_kernel(y.H, y.KKI[0], y.dirs, false, false)) {
        ent = FwdEntry{y.H, y.KKI[0], y.dirs, fn, snn_big_kernel(y.H, y.KKI[0], y.dirs, false, true),
                       snn_big_kernel(y.H, y.KKI[0], y.dirs, true, true), true, nullptr,
                       snn_big_rec_kernel(y.H, y.KKI[0], y.dirs)};
        return &ent;
    }
    return nullptr;
}
extern "C" int snn_supported(const snn_cfg* cfg)
{
    SnnLayout y;
    if (snn_make_layout(cfg, &y)) return 0;
    if (y.prec == 1) return snn_b3_kernel(y.H, y.XB[0], y.dirs) != nullptr;
    if (y.prec == 2) return snn_h2_kernel(y.H, y.XB[0], y.dirs, false) != nullptr;
    return find_fwd(y) != nullptr;
}

#define HIP_CHECK(expr)                                                             \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            snn_set_error("%s failed: %s", #expr, hipGetErrorString(e_));           \
            return SNN_ERR_HIP;                                                     \
        }                                                                           \
    } while (0)

static int pack_x_launch(const snn_cfg* cfg, const float* x, int64_t stride_t, const int64_t* step_off, int64_t stride_n,
                         const int64_t* curve_off, const int32_t* order, const int32_t* len_sorted, const int32_t* tile_T,
                         const int64_t* tile_off, int32_t n_tiles, int64_t xp_floats, float* xp, void* stream)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if (!x || !order || !len_sorted || !tile_T || !tile_off || !xp || n_tiles < 1 || xp_floats < 0) {
        snn_set_error("snn_pack_x: null pointer or bad size");
        return SNN_ERR_ARG;
    }
    // xp_floats is the caller's statement of the buffer size in 4-byte words.  The exact need (total tile-steps x
    // snn_x_step_floats) is only known from the DEVICE plan arrays, so the host checks the lower bound (one step per
    // tile) and the kernels refuse to write a tile that would end beyond xp_floats.
    const int64_t step_words = y.prec != 0 ? (int64_t)y.XB[0] * y.NPC * 64 * 4 : (int64_t)y.Dp * SNN_TILE;
    if (xp_floats < (int64_t)n_tiles * step_words) {
        snn_set_error("snn_pack_x: xp_floats %lld < %d tiles x %lld words per tile-step", (long long)xp_floats, n_tiles,
                      (long long)step_words);
        return SNN_ERR_ARG;
    }
    if (y.prec != 0) {
        hipLaunchKernelGGL(y.prec == 1 ? pack_x3_kernel<3> : pack_x3_kernel<2>, dim3(n_tiles), dim3(256), 0,
                           (hipStream_t)stream, y.D, y.Dp, y.XB[0], x, stride_t, step_off, stride_n, curve_off, order, len_sorted,
                           tile_T, tile_off, reinterpret_cast<__bf16*>(xp), xp_floats);
        HIP_CHECK(hipGetLastError());
        return SNN_OK;
    }
    // small batches: a tile's steps are dealt over up to 64 workgroups (those beyond the tile's length leave at once;
    // with 8, the 300 steps x 78 features of 128 ELAsTiCC-like curves took 354 us on 32 workgroups)
    const int ny = std::max(1, std::min(64, 8192 / n_tiles));      // (tiles are sorted by length: small workgroups balance)
    hipLaunchKernelGGL(pack_x_kernel, dim3(n_tiles, ny), dim3(256),
                       sizeof(float) * PACK_TS * y.Dp * (SNN_TILE + 1), (hipStream_t)stream, y.D, y.Dp, x, stride_t,
                       step_off, stride_n, curve_off, order, len_sorted, tile_T, tile_off, xp, xp_floats);
    HIP_CHECK(hipGetLastError());
    return SNN_OK;
}

extern "C" int snn_pack_x(const snn_cfg* cfg, const float* x, int64_t stride_t, int64_t stride_n,
                          const int64_t* curve_off, const int32_t* order, const int32_t* len_sorted, const int32_t* tile_T,
                          const int64_t* tile_off, int32_t n_tiles, int64_t xp_floats, float* xp, void* stream)
{
    return pack_x_launch(cfg, x, stride_t, nullptr, stride_n, curve_off, order, len_sorted, tile_T, tile_off, n_tiles, xp_floats,
                         xp, stream);
}

extern "C" int snn_pack_x_steps(const snn_cfg* cfg, const float* x, const int64_t* step_off, int64_t stride_n,
                                const int64_t* curve_off, const int32_t* order, const int32_t* len_sorted,
                                const int32_t* tile_T, const int64_t* tile_off, int32_t n_tiles, int64_t xp_floats, float* xp,
                                void* stream)
{
    if (!step_off) { snn_set_error("snn_pack_x_steps: step_off is NULL"); return SNN_ERR_ARG; }
    return pack_x_launch(cfg, x, 0, step_off, stride_n, curve_off, order, len_sorted, tile_T, tile_off, n_tiles, xp_floats, xp,
                         stream);
}

static int launch_sample(const SnnLayout& y, const float* mu, const float* rho, const float* eps, int mode,
                         uint64_t seed, uint64_t offset, int S, const snn_prior& prior, float* wk, float* kl,
                         float* kl_terms, hipStream_t st, int64_t mu_stride = 0)
{
    if (y.Dp != y.D || y.prec != 0 || y.cell != 0)  // padding words inside the image: define them (the alignment words
        HIP_CHECK(hipMemsetAsync(wk, 0, sizeof(float) * (size_t)y.PK * S, st));   // at the tail are written by the kernel)
    if (y.n_tensors > 64) { snn_set_error("snn_bbb_sample: %d tensors (more than 64)", y.n_tensors); return SNN_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(bbb_sample_kernel, dim3(y.n_tensors + 1, S, SNN_BBB_NCH), dim3(256), 0, st, y, mu, rho, eps, mode, seed,
                       offset, prior, wk, kl ? kl_terms : nullptr, mu_stride, S);
    HIP_CHECK(hipGetLastError());
    if (kl) {
        hipLaunchKernelGGL(kl_finish_kernel, dim3(S), dim3(64), 0, st, y.n_tensors, S, prior, kl_terms, kl);
        HIP_CHECK(hipGetLastError());
    }
    return SNN_OK;
}

extern "C" int64_t snn_kl_terms_floats(const snn_cfg* cfg, int32_t n_samples)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if (n_samples < 1) { snn_set_error("snn_kl_terms_floats: n_samples < 1"); return SNN_ERR_ARG; }
    return kl_final_floats(y.n_tensors, n_samples) + (int64_t)n_samples * y.n_tensors * SNN_BBB_NCH * 3 * 2;
}

extern "C" int snn_prepare_weights(const snn_cfg* cfg, const float* flat, float* wk, void* stream)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if (!flat || !wk) { snn_set_error("snn_prepare_weights: null pointer"); return SNN_ERR_ARG; }
    snn_prior pr = {0.5f, 1.f, 1.f, 1.f, 1.f};
    return launch_sample(y, flat, nullptr, nullptr, 1, 0, 0, 1, pr, wk, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int snn_prepare_weights_multi(const snn_cfg* cfg, const float* flat, int64_t sample_stride, int32_t n_samples,
                                         float* wk, void* stream)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if (!flat || !wk || n_samples < 1 || n_samples > 65535 || sample_stride < y.P) {
        snn_set_error("snn_prepare_weights_multi: null pointer, n_samples out of [1, 65535] or sample_stride < %lld",
                      (long long)y.P);
        return SNN_ERR_ARG;
    }
    snn_prior pr = {0.5f, 1.f, 1.f, 1.f, 1.f};
    return launch_sample(y, flat, nullptr, nullptr, 1, 0, 0, n_samples, pr, wk, nullptr, nullptr, (hipStream_t)stream,
                         sample_stride);
}

extern "C" int snn_bbb_sample(const snn_cfg* cfg, const float* mu, const float* rho, const float* eps, int32_t mode,
                              uint64_t seed, uint64_t offset, int32_t S, const snn_prior* prior, float* wk,
                              float* kl, float* kl_terms, void* stream)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if ((rc = snn_lstm_only(y, "snn_bbb_sample"))) return rc;
    if (!mu || !rho || !wk || !prior || S < 1 || S > 65535) {
        snn_set_error("snn_bbb_sample: null pointer or n_samples out of [1, 65535]");
        return SNN_ERR_ARG;
    }
    if (mode < 0 || mode > 2 || (mode == 0 && !eps)) {
        snn_set_error("snn_bbb_sample: mode %d invalid or eps missing", mode);
        return SNN_ERR_ARG;
    }
    if (kl && (!kl_terms || ((uintptr_t)kl_terms & 7))) {
        snn_set_error("snn_bbb_sample: kl requested without kl_terms scratch (snn_kl_terms_floats floats, 8-byte aligned)");
        return SNN_ERR_ARG;
    }
    if (!(prior->pi > 0.f && prior->pi < 1.f) || !(prior->sigma1 > 0.f) || !(prior->sigma2 > 0.f) ||
        !(prior->sigma1_out > 0.f) || !(prior->sigma2_out > 0.f)) {
        snn_set_error("snn_bbb_sample: prior out of range");
        return SNN_ERR_ARG;
    }
    return launch_sample(y, mu, rho, eps, mode, seed, offset, S, *prior, wk, kl, kl_terms, (hipStream_t)stream);
}

// ---- persistent forward launch geometry ------------------------------------
struct FwdGeom { int grid, nbuf, cus, threads; size_t lds; int64_t y_buf_floats; int64_t ws_bytes; };

static int fwd_geom(const SnnLayout& y, fwd_kernel_t fn, bool big, int max_T, FwdGeom* g, bool pair = false, bool big_train = false,
                    bool wpt = false)
{
    const int rows = std::max(2 * y.KKI[0], y.dirs * y.H);
    const int threads = big ? 256 : y.H * 8;
    if (wpt) {   // wpt_fwd_kernel: four waves, a tile each -- wave-private h images and logits, four y slabs per workgroup
        g->lds = sizeof(float) * ((size_t)4 * y.H * 32 + 4 * 128 + (size_t)4 * y.C * 32) + 16;
        if (g->lds > 160 * 1024) { snn_set_error("LDS request %zu exceeds 160 KiB (nb_classes too large)", g->lds); return SNN_ERR_UNSUPPORTED; }
        int dev = 0, cus = 0, per_cu = 0;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds));
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fn, 256, g->lds));
        per_cu = std::max(1, std::min(8, per_cu));
        g->grid = cus * per_cu; g->cus = cus; g->threads = 256;
        g->nbuf = y.L > 2 ? 2 : 1;
        g->y_buf_floats = (int64_t)y.dirs * max_T * y.H * 32;
        g->ws_bytes = 256 + (int64_t)sizeof(float) * g->grid * 4 * g->nbuf * g->y_buf_floats;
        return SNN_OK;
    }
    // (lstm_fwd_big_kernel<TRAIN>: the x image is at least one tape row -- its staging buffer -- and there is no mask image)
    if (big && big_train) g->lds = sizeof(float) * ((size_t)std::max(2 * rows * 32, SNN_TAPE_Q * y.H * 32) + 2 * y.H * 32 + (size_t)5 * y.C * 32) + 16;
    else if (big) g->lds = sizeof(float) * ((size_t)2 * rows * 32 + 3 * y.H * 32 + (size_t)5 * y.C * 32) + 16;
    else if (pair) g->lds = sizeof(float) * ((size_t)2 * rows * 32 + 2 * y.H * 32 + (size_t)2 * (y.NW + 1) * y.C * 32 + y.NW * 32) + 16;
    else g->lds = sizeof(float) * ((size_t)2 * rows * 32 + 2 * y.H * 32 + (size_t)(y.NW + 1) * y.C * 32 + y.NW * 32) + 16;
    if (g->lds > 160 * 1024) { snn_set_error("LDS request %zu exceeds 160 KiB (nb_classes too large)", g->lds); return SNN_ERR_UNSUPPORTED; }
    static thread_local int cached_dev = -1, cached_cus = 0;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev != cached_dev) {
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        cached_cus = prop.multiProcessorCount;
        cached_dev = dev;
    }
    int per_cu = 0;
    HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds));
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fn, threads, g->lds));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    g->grid = cached_cus * per_cu;
    g->cus = cached_cus;
    g->nbuf = y.L > 2 ? 2 : 1;
    g->y_buf_floats = (int64_t)y.dirs * max_T * y.H * 32 * (pair ? 2 : 1);
    g->threads = threads;
    // the two-pass kernel also parks W_ih x_t + b for every step: max_T x 4H x 32 floats per workgroup
    const int64_t gx = big ? (int64_t)max_T * (y.H / 8) * 1024 : 0;
    g->ws_bytes = 256 + (int64_t)sizeof(float) * g->grid * (g->nbuf * g->y_buf_floats + gx);
    return SNN_OK;
}

// snn_lstm_forward_noalloc: the workspace is a fixed scratch, so the persistent grid shrinks to the number of
// per-workgroup slabs that fit (any grid >= 1 drains the work queue; results do not depend on the grid)
static int fit_grid_to_workspace(FwdGeom* g, int64_t header, int64_t workspace_bytes, int max_T)
{
    const int64_t per_wg = (g->ws_bytes - header) / g->grid;
    const int64_t cap = per_wg > 0 ? (workspace_bytes - header) / per_wg : g->grid;
    if (cap < 1) {
        snn_set_error("snn_lstm_forward_noalloc: one workgroup's layer-output slab for max_T=%d is %lld bytes, beyond the "
                      "%lld-byte scratch of this entry point: use snn_lstm_forward with a caller workspace",
                      max_T, (long long)per_wg, (long long)(workspace_bytes - header));
        return SNN_ERR_WORKSPACE;
    }
    if (cap < g->grid) {
        g->grid = (int)cap;
        g->ws_bytes = header + per_wg * cap;
    }
    return SNN_OK;
}

// ---- bf16 x 3 forward (SNN_PREC_BF16X3) ----------------------------------------------------------------
static int b3_geom(const SnnLayout& y, fwd_b3_kernel_t fn, int max_T, FwdGeom* g)
{
    const int HB = y.H / 16, XBM = std::max(y.XB[0], y.dirs * HB);
    g->threads = y.H * 8;
    const int G = y.prec == 2 ? SNN_H2_G : 4;   // tiles per work item (B3_G of the kernels)
    g->lds = (size_t)(2 * XBM + 2 * HB) * y.NPC * 64 * 16 + sizeof(float) * ((size_t)(y.NW + G) * y.C * 32 + y.NW * 32 + G * 64) + 16;
    if (g->lds > 160 * 1024) { snn_set_error("LDS request %zu exceeds 160 KiB (nb_classes too large)", g->lds); return SNN_ERR_UNSUPPORTED; }
    int dev = 0, cus = 0, per_cu = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds));
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fn, g->threads, g->lds));
    per_cu = std::max(1, std::min(per_cu, 8));
    g->grid = cus * per_cu;
    g->cus = cus;
    g->nbuf = y.L > 2 ? 2 : 1;
    g->y_buf_floats = (int64_t)G * max_T * y.dirs * HB * y.NPC * 64 * 4;   // [tile][t][dir * HB blocks][pieces][64 lanes][8 x 16 bit]
    g->ws_bytes = 1024 + (int64_t)sizeof(float) * g->grid * g->nbuf * g->y_buf_floats;
    return SNN_OK;
}

static int launch_forward_b3(const SnnLayout& y, const float* xp, const int32_t* len_sorted, const int32_t* order,
                             const int32_t* tile_T, const int64_t* tile_off, int32_t n_curves, int32_t n_tiles,
                             int32_t max_T, const float* wk, int64_t wk_sample_stride, int32_t n_samples,
                             const snn_masks* masks, float* out_logits, float* out_probs, int64_t out_stride_sample,
                             int64_t out_stride_curve, void* workspace, int64_t workspace_bytes, void* stream,
                             bool fit_grid)
{
    const bool masked = masks && masks->mode != SNN_MASK_NONE;
    if (masked && (y.prec != 2 || masks->mid_T != 1)) {
        snn_set_error("split-precision kernels: masks are implemented for SNN_PREC_F16X2 with locked (mid_T = 1) masks only");
        return SNN_ERR_UNSUPPORTED;
    }
    fwd_b3_kernel_t fn = y.prec == 2 ? snn_h2_kernel(y.H, y.XB[0], y.dirs, masked) : snn_b3_kernel(y.H, y.XB[0], y.dirs);
    if (!fn) {
        snn_set_error("no split-precision forward kernel for H=%d D=%d dirs=%d%s", y.H, y.D, y.dirs, masked ? " with masks" : "");
        return SNN_ERR_UNSUPPORTED;
    }
    if (y.prec == 1 && n_tiles > 1) {   // two tiles per workgroup, software-pipelined (snn_forward_b3p.hip); bit-identical results
        static const bool no_pair = getenv("SNN_NO_PAIR") != nullptr;   // A/B measurements only
        fwd_b3_kernel_t fp = no_pair ? nullptr : snn_b3p_kernel(y.H, y.XB[0], y.dirs);
        if (fp) fn = fp;
    }
    FwdGeom g;
    int rc = b3_geom(y, fn, max_T, &g);
    if (rc) return rc;
    if (fit_grid && (rc = fit_grid_to_workspace(&g, 1024, workspace_bytes, max_T))) return rc;
    if (workspace_bytes < g.ws_bytes) {
        snn_set_error("snn_lstm_forward: workspace %lld < required %lld bytes", (long long)workspace_bytes, (long long)g.ws_bytes);
        return SNN_ERR_WORKSPACE;
    }
    FwdParams p;
    memset(&p, 0, sizeof(p));
    p.D = y.D; p.Dp = y.Dp; p.L = y.L; p.C = y.C; p.pool = y.pool; p.F = y.F;
    p.xp = xp; p.len_sorted = len_sorted; p.order = order; p.tile_T = tile_T; p.tile_off = tile_off;
    p.n_curves = n_curves; p.n_tiles = n_tiles; p.max_T = max_T; p.n_samples = n_samples;
    p.wk = wk; p.wk_stride = wk_sample_stride;
    rc = snn_masks_to_device(masks, max_T, &p.mk);
    if (rc) return rc;
    p.out_logits = out_logits; p.out_probs = out_probs; p.os_s = out_stride_sample; p.os_n = out_stride_curve;
    p.counter = (unsigned int*)workspace;                 // 8 per-XCD queue heads, 64 bytes apart
    p.yscratch = (float*)((char*)workspace + 1024);
    p.y_buf_floats = g.y_buf_floats;
    p.nbuf = g.nbuf;
    B3Offsets o;
    for (int ld = 0; ld < y.L * y.dirs; ++ld) { o.k_A[ld] = y.k_A[ld]; o.k_bias[ld] = y.k_bias[ld]; }
    o.k_head_w = y.k_head_w;
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipMemsetAsync(workspace, 0, 1024, st));
    const int G = y.prec == 2 ? SNN_H2_G : 4;
    const int64_t n_items = (int64_t)((n_tiles + G - 1) / G) * n_samples;
    snprintf(g_last_kernel, sizeof(g_last_kernel), "%s<%d,%d,%d%s> grid=%d block=%d lds=%zu",
             y.prec == 2 ? "lstm_fwd_h2_kernel" : (y.prec == 1 && n_tiles > 1 && snn_b3p_kernel(y.H, y.XB[0], y.dirs) == fn)
             ? "lstm_fwd_b3p_kernel" : "lstm_fwd_b3_kernel", y.H, y.XB[0], y.dirs, y.prec == 2 ? (masked ? ",true" : ",false") : "",
             (int)std::min<int64_t>(g.grid, n_items), g.threads, g.lds);
    hipLaunchKernelGGL(fn, dim3((int)std::min<int64_t>(g.grid, n_items)), dim3(g.threads), g.lds, st, p, o);
    HIP_CHECK(hipGetLastError());
    return SNN_OK;
}

extern "C" int64_t snn_lstm_forward_workspace_bytes(const snn_cfg* cfg, int32_t max_T)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if (y.prec != 0) {
        fwd_b3_kernel_t fn = y.prec == 2 ? snn_h2_kernel(y.H, y.XB[0], y.dirs, false) : snn_b3_kernel(y.H, y.XB[0], y.dirs);
        if (!fn) { snn_set_error("no split-precision forward kernel for H=%d D=%d dirs=%d", y.H, y.D, y.dirs); return SNN_ERR_UNSUPPORTED; }
        if (max_T < 1) { snn_set_error("max_T < 1"); return SNN_ERR_ARG; }
        FwdGeom g, gp;
        rc = b3_geom(y, fn, max_T, &g);
        if (fwd_b3_kernel_t fp = (rc || y.prec != 1) ? (rc ? nullptr : snn_h2_kernel(y.H, y.XB[0], y.dirs, true))
                                                     : snn_b3p_kernel(y.H, y.XB[0], y.dirs)) {
            rc = b3_geom(y, fp, max_T, &gp);
            if (!rc) g.ws_bytes = std::max(g.ws_bytes, gp.ws_bytes);
        }
        return rc ? rc : g.ws_bytes;
    }
    const FwdEntry* e = find_fwd(y);
    if (!e) { snn_set_error("no forward kernel for H=%d D=%d dirs=%d cell=%d", y.H, y.D, y.dirs, y.cell); return SNN_ERR_UNSUPPORTED; }
    if (max_T < 1) { snn_set_error("max_T < 1"); return SNN_ERR_ARG; }
    FwdGeom g, gt;
    rc = fwd_geom(y, e->fn, e->big, max_T, &g);
    if (!rc && !e->big && y.cell != 0) {
        rc = fwd_geom(y, e->fn_train, false, max_T, &gt);
        if (!rc) g.ws_bytes = std::max(g.ws_bytes, gt.ws_bytes);
        if (fwd_kernel_t fw = rc ? nullptr : snn_wpt_kernel(y.H, y.KKI[0], y.dirs, y.cell)) {
            rc = fwd_geom(y, fw, false, max_T, &gt, false, false, true);
            if (!rc) g.ws_bytes = std::max(g.ws_bytes, gt.ws_bytes);
        }
    }
    if (!rc && !e->big && y.cell == 0) {
        rc = fwd_geom(y, e->fn_masked, false, max_T, &gt);
        if (!rc) g.ws_bytes = std::max(g.ws_bytes, gt.ws_bytes);
        if (!rc) rc = fwd_geom(y, e->fn_train, false, max_T, &gt);
        if (!rc) g.ws_bytes = std::max(g.ws_bytes, gt.ws_bytes);
        for (int m = 0; m < 2; ++m)
            if (fwd_kernel_t fp = rc ? nullptr : snn_pair_kernel(y.H, y.KKI[0], y.dirs, m == 1)) {
                rc = fwd_geom(y, fp, false, max_T, &gt, true);
                if (!rc) g.ws_bytes = std::max(g.ws_bytes, gt.ws_bytes);
            }
    }
    return rc ? rc : g.ws_bytes;
}

int snn_masks_to_device(const snn_masks* masks, int max_T, MaskP* mk)
{
    memset(mk, 0, sizeof(*mk));
    if (!masks || masks->mode == SNN_MASK_NONE) return SNN_OK;
    if (masks->mode != SNN_MASK_ARRAYS && masks->mode != SNN_MASK_PHILOX) {
        snn_set_error("masks.mode %d invalid", masks->mode);
        return SNN_ERR_ARG;
    }
    if (masks->mid_T != 1 && masks->mid_T != max_T && (masks->mode == SNN_MASK_PHILOX || masks->mid)) {
        snn_set_error("masks.mid_T must be 1 or max_T (%d), got %d", max_T, masks->mid_T);
        return SNN_ERR_ARG;
    }
    if (masks->mode == SNN_MASK_PHILOX && !(masks->keep_prob > 0.f && masks->keep_prob <= 1.f)) {
        snn_set_error("masks.keep_prob out of (0,1]");
        return SNN_ERR_ARG;
    }
    mk->mode = masks->mode; mk->which = masks->which; mk->mid_T = masks->mid_T < 1 ? 1 : masks->mid_T;
    mk->keep_scale = masks->keep_scale; mk->keep_prob = masks->keep_prob;
    mk->seed = masks->seed; mk->offset = masks->offset; mk->curve_base = masks->curve_base;
    mk->m_in = masks->in; mk->m_whh = masks->whh; mk->m_mid = masks->mid; mk->m_out = masks->out;
    return SNN_OK;
}

static int launch_forward(const snn_cfg* cfg, bool train, const float* xp, const int32_t* len_sorted,
                          const int32_t* order, const int32_t* tile_T, const int64_t* tile_off, int32_t n_curves,
                          int32_t n_tiles, int32_t max_T, const float* wk, int64_t wk_sample_stride,
                          int32_t n_samples, const snn_masks* masks, float* out_logits, float* out_probs,
                          int64_t out_stride_sample, int64_t out_stride_curve, float* tape, int64_t tape_floats,
                          void* workspace, int64_t workspace_bytes, void* stream, bool fit_grid = false)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    const FwdEntry* e = find_fwd(y);
    if (!e && y.prec == 0) { snn_set_error("no forward kernel for H=%d D=%d dirs=%d cell=%d", y.H, y.D, y.dirs, y.cell); return SNN_ERR_UNSUPPORTED; }
    if (!xp || !len_sorted || !order || !tile_T || !tile_off || !wk || !workspace || (!out_logits && !out_probs)) {
        snn_set_error("snn_lstm_forward: null pointer");
        return SNN_ERR_ARG;
    }
    if (y.prec != 0 && train) {
        snn_set_error("SNN_PREC_BF16X3 / SNN_PREC_F16X2 are inference-only");
        return SNN_ERR_UNSUPPORTED;
    }
    if (n_curves < 1 || n_tiles != (n_curves + SNN_TILE - 1) / SNN_TILE || max_T < 1 || n_samples < 1 ||
        (int64_t)n_tiles * n_samples > 0x7fffffff || wk_sample_stride < 0 ||
        (wk_sample_stride != 0 && wk_sample_stride < y.PK)) {
        snn_set_error("snn_lstm_forward: inconsistent sizes (n_curves=%d n_tiles=%d max_T=%d n_samples=%d)", n_curves,
                      n_tiles, max_T, n_samples);
        return SNN_ERR_ARG;
    }
    if (y.prec != 0)
        return launch_forward_b3(y, xp, len_sorted, order, tile_T, tile_off, n_curves, n_tiles, max_T, wk, wk_sample_stride,
                                 n_samples, masks, out_logits, out_probs, out_stride_sample, out_stride_curve, workspace,
                                 workspace_bytes, stream, fit_grid);
    const int64_t tape_tile = tape_tile_floats(y.H, y.L * y.dirs, max_T, y.F);
    if (train && (!tape || n_samples != 1 || tape_floats < tape_tile * n_tiles)) {
        snn_set_error("snn_lstm_forward_train: tape missing or too small (%lld < %lld floats) or n_samples != 1",
                      (long long)tape_floats, (long long)(tape_tile * n_tiles));
        return SNN_ERR_ARG;
    }
    const bool masked = masks && masks->mode != SNN_MASK_NONE;
    fwd_kernel_t fn = train ? e->fn_train : masked ? e->fn_masked : e->fn;
    if (e->big && train && !masked) fn = snn_big_kernel(y.H, y.KKI[0], y.dirs, true, false);   // no mask code in the kernel
    if (!fn) { snn_set_error("no %s kernel for H=%d cell=%d", train ? "training" : "masked", y.H, y.cell); return SNN_ERR_UNSUPPORTED; }
    if (e->big && !train && masked && masks->mid_T != 1) { snn_set_error("H=%d kernels take locked between-layer masks only", y.H); return SNN_ERR_UNSUPPORTED; }
    // un-masked inference: two tiles per workgroup, software-pipelined (snn_forward_pair.hip)
    bool pair = false;
    // (also for a single tile -- a missing partner is a tile without live curves -- so that every inference launch of a
    // shape runs the same arithmetic: the pair kernel folds the activations' exp2 factors into its weight registers)
    if (!train && (!masked || masks->mid_T <= 1) && !e->big && y.cell == 0) {   // locked masks only
        static const bool no_pair = getenv("SNN_NO_PAIR") != nullptr;   // A/B measurements only
        fwd_kernel_t fp = no_pair ? nullptr : snn_pair_kernel(y.H, y.KKI[0], y.dirs, masked);
        if (fp) { fn = fp; pair = true; }
    }
    // GRU / tanh RNN inference at H = 32: one wave per tile (snn_forward_wpt.hip); SNN_NO_WPT: the LSTM-shaped kernel (A/B, tests)
    bool wpt = false;
    if (!train && !masked && y.cell != 0 && !getenv("SNN_NO_WPT")) {
        if (fwd_kernel_t fw = snn_wpt_kernel(y.H, y.KKI[0], y.dirs, y.cell)) { fn = fw; wpt = true; }
    }
    FwdGeom g;
    rc = fwd_geom(y, fn, e->big, max_T, &g, pair, e->big && train, wpt);
    if (rc) return rc;
    if (fit_grid && (rc = fit_grid_to_workspace(&g, 256, workspace_bytes, max_T))) return rc;
    if (workspace_bytes < g.ws_bytes) {
        snn_set_error("snn_lstm_forward: workspace %lld < required %lld bytes", (long long)workspace_bytes, (long long)g.ws_bytes);
        return SNN_ERR_WORKSPACE;
    }
    FwdParams p;
    memset(&p, 0, sizeof(p));
    p.D = y.D; p.Dp = y.Dp; p.L = y.L; p.C = y.C; p.pool = y.pool; p.F = y.F;
    p.xp = xp; p.len_sorted = len_sorted; p.order = order; p.tile_T = tile_T; p.tile_off = tile_off;
    p.n_curves = n_curves; p.n_tiles = n_tiles; p.max_T = max_T; p.n_samples = n_samples;
    p.wk = wk; p.wk_stride = wk_sample_stride;
    rc = snn_masks_to_device(masks, max_T, &p.mk);
    if (rc) return rc;
    p.out_logits = out_logits; p.out_probs = out_probs; p.os_s = out_stride_sample; p.os_n = out_stride_curve;
    p.counter = (unsigned int*)workspace;
    p.yscratch = (float*)((char*)workspace + 256);
    p.y_buf_floats = g.y_buf_floats;
    p.nbuf = g.nbuf;
    p.tape = tape; p.tape_tile = tape_tile;
    hipStream_t st = (hipStream_t)stream;
    // training: one launch per layer, a workgroup per (tile, direction) (snn_fwd_small.h).  Written for small batches
    // (it halves their latency chain), it is also the fastest path at every larger size measured (forward ms, one-tile
    // persistent / two-tile persistent / layer-wise: 12 288 curves 0.54 / 0.61 / 0.32; 32 768: 0.84 / 1.07 / 0.69;
    // 65 536: 1.55 / 1.40 / 1.30; 131 072: 3.06 / 2.72 / 2.63): independent workgroups leave the balancing to the
    // dispatcher and run the two directions side by side.  The persistent kernel remains for unidirectional models,
    // the GRU / tanh-RNN cells and batches whose head partials do not fit the workspace.
    // while even the half tiles get a compute unit each: 16 curves per workgroup (snn_rec16.hip), half the chain
    // per step.  SNN_REC16=0|1 forces it off / on (A/B, tests); the backward follows the same rule.
    const char* f16 = getenv("SNN_REC16");
    const bool rec16 = train && snn_rec16_supported(y) && (f16 ? f16[0] == '1' : (int64_t)n_tiles * y.dirs * 2 <= g.cus);
    const int n_part = (e->big && !rec16) ? 4 : y.NW;      // per-wave head partials per phase
    const int64_t part_bytes = (int64_t)sizeof(float) * n_tiles * y.L * y.dirs * n_part * y.C * 32;
    const bool no_layerwise = getenv("SNN_NO_LAYERWISE") != nullptr;   // A/B measurements and tests (read per call)
    // Small batches: input products of a whole layer in one launch over the chip (fwd_gx_kernel), then the recurrent
    // half with a workgroup per (tile, direction) -- the dependent chain per step shrinks to the H / 2 recurrent k-steps.
    // Chosen while every (tile, direction) gets a compute unit to itself; SNN_FWD_SPLIT=0|1 forces it off / on (A/B,
    // tests).  Results are bit-identical to the other training forwards.
    {
        const char* force = getenv("SNN_FWD_SPLIT");
        const int64_t gx_bytes = (int64_t)sizeof(float) * snn_gx_floats(y, n_tiles, max_T);
        const int64_t part_off = 256, gx_off = (part_off + part_bytes + 255) & ~(int64_t)255;
        const bool want = force ? force[0] == '1' : (int64_t)n_tiles * y.dirs <= g.cus;
        if (train && e->fn_rec && want && !no_layerwise && snn_gx_supported(y) && gx_off + gx_bytes <= workspace_bytes) {
            float* head_part = (float*)((char*)workspace + part_off);
            float* gx = (float*)((char*)workspace + gx_off);
            const size_t lds = e->big ? sizeof(float) * ((size_t)3 * y.H * 32 + (size_t)5 * y.C * 32) + 16
                                      : sizeof(float) * ((size_t)2 * SNN_TAPE_Q * y.H * 32 + (size_t)2 * y.H * 32 + (size_t)(y.NW + 1) * y.C * 32) + 16;
            HIP_CHECK(hipFuncSetAttribute((const void*)e->fn_rec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            snprintf(g_last_kernel, sizeof(g_last_kernel), "fwd_gx_kernel + %s<%d,%d,%d,cell%d> x %d layers grid=%d block=%d lds=%zu",
                     rec16 ? "lstm_fwd_rec16_kernel" : e->big ? "lstm_fwd_big_rec_kernel" : "lstm_fwd_rec_kernel", y.H, y.KKI[0],
                     y.dirs, y.cell, y.L, n_tiles * y.dirs * (rec16 ? 2 : 1), rec16 ? y.H * 8 : g.threads,
                     rec16 ? sizeof(float) * ((size_t)2 * SNN_TAPE_Q * y.H * 16 + (size_t)2 * y.H * 16) : lds);
            for (int l = 0; l < y.L; ++l) {
                if ((rc = snn_launch_gx(y, p, l, gx, g.cus, st))) return rc;
                if (rec16) { if ((rc = snn_launch_fwd_rec16(y, p, l, head_part, gx, st))) return rc; }
                else hipLaunchKernelGGL(e->fn_rec, dim3(n_tiles * y.dirs), dim3(g.threads), lds, st, p, l, head_part, (const float*)gx);
            }
            hipLaunchKernelGGL(head_combine_kernel, dim3(n_tiles), dim3(256), sizeof(float) * y.C * 32, st, p, n_part, y.dirs, y.k_head_b, head_part);
            HIP_CHECK(hipGetLastError());
            return SNN_OK;
        }
    }
    if (train && e->fn_train_layer && !no_layerwise && 256 + part_bytes <= workspace_bytes) {
        float* head_part = (float*)((char*)workspace + 256);
        HIP_CHECK(hipFuncSetAttribute((const void*)e->fn_train_layer, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));
        snprintf(g_last_kernel, sizeof(g_last_kernel), "lstm_fwd_train_layer_kernel<%d,%d,%d> x %d layers grid=%d block=%d lds=%zu",
                 y.H, y.KKI[0], y.dirs, y.L, n_tiles * y.dirs, g.threads, g.lds);
        for (int l = 0; l < y.L; ++l)
            hipLaunchKernelGGL(e->fn_train_layer, dim3(n_tiles * y.dirs), dim3(g.threads), g.lds, st, p, l, head_part);
        hipLaunchKernelGGL(head_combine_kernel, dim3(n_tiles), dim3(256), sizeof(float) * y.C * 32, st, p, y.NW, y.dirs, y.k_head_b, head_part);
        HIP_CHECK(hipGetLastError());
        return SNN_OK;
    }
    // (n_part / rec16 above concern the split path only: the layer-wise kernel exists for H = 32 / 64, where n_part is NW)
    HIP_CHECK(hipMemsetAsync(workspace, 0, 256, st));
    const int64_t n_items = (int64_t)(wpt ? (n_tiles + 3) / 4 : pair ? (n_tiles + 1) / 2 : n_tiles) * n_samples;
    const int grid = (int)std::min<int64_t>(g.grid, n_items);
    snprintf(g_last_kernel, sizeof(g_last_kernel), "%s<%d,%d,%d,%s> grid=%d block=%d lds=%zu",
             wpt ? "wpt_fwd_kernel" : pair ? "lstm_fwd_pair_kernel" : e->big ? "lstm_fwd_big_kernel" : "lstm_fwd_kernel", y.H, y.KKI[0], y.dirs,
             train ? "train" : masked ? "masked" : wpt ? (y.cell == 1 ? "gru" : "rnn") : "plain", grid, g.threads, g.lds);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(g.threads), g.lds, st, p);
    HIP_CHECK(hipGetLastError());
    return SNN_OK;
}

extern "C" int snn_lstm_forward(const snn_cfg* cfg, const float* xp, const int32_t* len_sorted,
                                const int32_t* order, const int32_t* tile_T, const int64_t* tile_off,
                                int32_t n_curves, int32_t n_tiles, int32_t max_T, const float* wk,
                                int64_t wk_sample_stride, int32_t n_samples, const snn_masks* masks,
                                float* out_logits, float* out_probs, int64_t out_stride_sample,
                                int64_t out_stride_curve, void* workspace, int64_t workspace_bytes, void* stream)
{
    return launch_forward(cfg, false, xp, len_sorted, order, tile_T, tile_off, n_curves, n_tiles, max_T, wk,
                          wk_sample_stride, n_samples, masks, out_logits, out_probs, out_stride_sample,
                          out_stride_curve, nullptr, 0, workspace, workspace_bytes, stream);
}

// ---- the one entry point that allocates nothing (include/snn_hip.h) ---------------------------------------
// Work-queue heads + layer-output slabs live in this file-scope array instead of a caller workspace.  Every GPU the
// library runs on holds its own copy at its own address: the address is looked up once per device (with that device
// current) and only that pointer -- never the symbol -- is handed to hipMemsetAsync and the kernels.
static __device__ __align__(256) unsigned char scratch[SNN_NOALLOC_SCRATCH_BYTES];
static void* g_scratch_ptr[SNN_NOALLOC_MAX_DEVICES];
static std::mutex g_scratch_mu;

static int noalloc_scratch(void** out)
{
    int dev = -1;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= SNN_NOALLOC_MAX_DEVICES) {
        snn_set_error("snn_lstm_forward_noalloc: device ordinal %d outside [0, %d)", dev, SNN_NOALLOC_MAX_DEVICES);
        return SNN_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) { snn_set_error("snn_lstm_forward_noalloc: hipGetSymbolAddress returned a null scratch address"); return SNN_ERR_HIP; }
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return SNN_OK;
}

extern "C" int64_t snn_lstm_forward_noalloc_scratch_bytes(void) { return SNN_NOALLOC_SCRATCH_BYTES; }

extern "C" int snn_lstm_forward_noalloc(const snn_cfg* cfg, const float* xp, const int32_t* len_sorted,
                                        const int32_t* order, const int32_t* tile_T, const int64_t* tile_off,
                                        int32_t n_curves, int32_t n_tiles, int32_t max_T, const float* wk,
                                        int64_t wk_sample_stride, int32_t n_samples, const snn_masks* masks,
                                        float* out_logits, float* out_probs, int64_t out_stride_sample,
                                        int64_t out_stride_curve, void* stream)
{
    void* ws = nullptr;
    int rc = noalloc_scratch(&ws);
    if (rc) return rc;
    return launch_forward(cfg, false, xp, len_sorted, order, tile_T, tile_off, n_curves, n_tiles, max_T, wk,
                          wk_sample_stride, n_samples, masks, out_logits, out_probs, out_stride_sample,
                          out_stride_curve, nullptr, 0, ws, SNN_NOALLOC_SCRATCH_BYTES, stream, /*fit_grid=*/true);
}

extern "C" int64_t snn_train_tape_floats(const snn_cfg* cfg, int32_t n_tiles, int32_t max_T)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if (n_tiles < 1 || max_T < 1) { snn_set_error("snn_train_tape_floats: bad sizes"); return SNN_ERR_ARG; }
    return tape_tile_floats(y.H, y.L * y.dirs, max_T, y.F) * n_tiles;
}

extern "C" int snn_lstm_forward_train(const snn_cfg* cfg, const float* xp, const int32_t* len_sorted,
                                      const int32_t* order, const int32_t* tile_T, const int64_t* tile_off,
                                      int32_t n_curves, int32_t n_tiles, int32_t max_T, const float* wk,
                                      const snn_masks* masks, float* out_logits, float* tape, int64_t tape_floats,
                                      void* workspace, int64_t workspace_bytes, void* stream)
{
    return launch_forward(cfg, true, xp, len_sorted, order, tile_T, tile_off, n_curves, n_tiles, max_T, wk, 0, 1,
                          masks, out_logits, nullptr, 0, cfg ? cfg->nb_classes : 0, tape, tape_floats, workspace,
                          workspace_bytes, stream);
}
