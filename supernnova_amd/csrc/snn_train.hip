// snn_train.hip -- gfx950 kernels + C ABI for the training half of the hot path:
//   lstm_bwd_kernel     BPTT through the LSTM stack + pooling + head, per 32-curve tile
//   grad_reduce_kernel  fixed-order sum of the per-workgroup gradient slabs
//   bbb_backward_kernel reparameterisation + KL gradients (dmu, drho)
//   adam_kernel         fused Adam over one flat buffer
// Replaces autograd of `loss.backward()` and `optimizer.step()`
// (utils/training_utils.py:523-524) for the three model classes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
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

#include "snn_bwd_small.h"   // BwdParams (the kernels are instantiated in snn_bwd_h<H>.hip)

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
const BwdEntry* snn_bwd_small_h32(int KKI0, int dirs);   // snn_bwd_h32.hip: fused kernel, H = 32
static const BwdEntry* find_bwd(const SnnLayout& y)
{
    return (y.H == 32 && y.cell == 0) ? snn_bwd_small_h32(y.KKI[0], y.dirs) : nullptr;   // GRU / tanh RNN: split path only
}
// H = 32 has both paths.  Measured on MI355X (BayesianRNN train step, lengths U{3..40}), ms per step at batch
// 128 / 8192 / 16384 / 32768 / 65536: fused 1.32 / 1.66 / 2.57 / 4.05 / 7.43, split 0.59 / 1.33 / 2.25 / 3.73 / 7.08: the
// split path runs both directions and the weight-gradient GEMMs in parallel; the fused one keeps dG on chip and needs a
// workspace that does not grow with the batch (the split path's dG / dy buffers take 64 KB per tile-step: 4 GB at 65536
// curves x 40 steps), so it takes over above 8192 tiles.  SNN_BWD_PATH=fused|split overrides (A/B measurements).
static bool use_split_path(const SnnLayout& y, const BwdEntry* fused, int n_tiles)
{
    if (!fused) return true;
    const char* force = getenv("SNN_BWD_PATH");
    if (force && force[0] == 'f') return false;
    if (force && force[0] == 's') return true;
    return n_tiles <= 8192;
}
// H = 32, round 2: the streamed path (snn_train_big.hip: bwd_stream_kernel) -- one pass over the tape per (layer,
// direction), dG kept in LDS, dW accumulated in registers over a fixed stripe of tiles.  It moves half the bytes of
// the split path and needs no workspace that grows with the gate gradients; it is the default wherever it exists.
// SNN_BWD_PATH=split|fused selects the round-1 paths (A/B measurements and the equivalence tests).
bool snn_bwd_stream_supported(const SnnLayout& y);
int64_t snn_bwd_stream_workspace_bytes(const SnnLayout& y, int n_tiles, int max_T);
int snn_lstm_backward_stream(const SnnLayout& y, const float* xp, const int32_t* len_sorted, const int32_t* order,
                             const int32_t* tile_T, const int64_t* tile_off, int32_t n_curves, int32_t n_tiles, int32_t max_T,
                             const float* wk, const MaskP& mk, const float* tape, int64_t tape_tile, const float* dlogits,
                             float* dflat, int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream);
static bool use_stream_path(const SnnLayout& y, int n_tiles)
{
    const char* force = getenv("SNN_BWD_PATH");     // read per call: the tests switch paths inside one process
    if (force && (force[0] == 'f' || (force[0] == 's' && force[1] == 'p'))) return false;
    if (!snn_bwd_stream_supported(y)) return false;
    if (force && force[0] == 's' && force[1] == 't') return true;
    // a handful of tiles is a latency chain either way, and the split path's chain (recurrent product only) is the
    // shorter one: batch 128 backward 0.23 ms split / 0.43 ms streamed; 8192: 1.00 / 0.79; 65536: 5.29 / 4.42
    return n_tiles * y.dirs >= 128;
}
// H = 64, 128: recurrent kernel + deferred GEMMs (snn_train_big.hip)
bool snn_bwd_big_supported(const SnnLayout& y);
int64_t snn_bwd_big_workspace_bytes(const SnnLayout& y, int n_tiles, int max_T);
int snn_lstm_backward_big(const SnnLayout& y, const float* xp, const int32_t* len_sorted, const int32_t* order,
                          const int32_t* tile_T, const int64_t* tile_off, int32_t n_curves, int32_t n_tiles, int32_t max_T,
                          const float* wk, const MaskP& mk, const float* tape, int64_t tape_tile, const float* dlogits,
                          float* dflat, int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream);

static const int BWD_MAX_GRID = 256;

// which of the three backward paths the calling thread's last snn_lstm_backward took (reports, tests)
static thread_local const char* g_last_bwd_path = "";
extern "C" const char* snn_last_backward_path(void) { return g_last_bwd_path; }

static size_t bwd_lds_bytes(const SnnLayout& y)
{
    const int xrows = std::max(y.dirs * y.H, 2 * y.KKI[0]), krows = y.H + xrows, mt = (krows + 31) / 32;
    const size_t fixed = (size_t)2 * krows * LDS_PAD + 4 * y.H * LDS_PAD + (size_t)y.NW * mt * 1024 + y.H * 32 + 64;
    return sizeof(float) * (fixed + (size_t)(y.C + y.F) * 32);
}

extern "C" int64_t snn_lstm_backward_workspace_bytes(const snn_cfg* cfg, int32_t n_tiles, int32_t max_T)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if (n_tiles < 1 || max_T < 1) { snn_set_error("snn_lstm_backward_workspace_bytes: bad sizes"); return SNN_ERR_ARG; }
    const BwdEntry* fused = find_bwd(y);
    if (use_stream_path(y, n_tiles)) return snn_bwd_stream_workspace_bytes(y, n_tiles, max_T);
    if (use_split_path(y, fused, n_tiles)) {
        if (snn_bwd_big_supported(y)) return snn_bwd_big_workspace_bytes(y, n_tiles, max_T);
        snn_set_error("no backward kernel for H=%d D=%d dirs=%d", y.H, y.D, y.dirs);
        return SNN_ERR_UNSUPPORTED;
    }
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
    if (!e && !snn_bwd_big_supported(y)) { snn_set_error("no backward kernel for H=%d D=%d dirs=%d", y.H, y.D, y.dirs); return SNN_ERR_UNSUPPORTED; }
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
    if (use_stream_path(y, n_tiles)) {
        MaskP mk;
        rc = snn_masks_to_device(masks, max_T, &mk);
        if (rc) return rc;
        g_last_bwd_path = "stream";
        return snn_lstm_backward_stream(y, xp, len_sorted, order, tile_T, tile_off, n_curves, n_tiles, max_T, wk, mk, tape,
                                        tape_tile, dlogits, dflat, accumulate, workspace, workspace_bytes, stream);
    }
    if (use_split_path(y, e, n_tiles)) {
        MaskP mk;
        rc = snn_masks_to_device(masks, max_T, &mk);
        if (rc) return rc;
        g_last_bwd_path = "split";
        return snn_lstm_backward_big(y, xp, len_sorted, order, tile_T, tile_off, n_curves, n_tiles, max_T, wk, mk, tape,
                                     tape_tile, dlogits, dflat, accumulate, workspace, workspace_bytes, stream);
    }
    const int grid = std::min(n_tiles, BWD_MAX_GRID);
    const int64_t need = (int64_t)sizeof(float) * grid * (y.P + (int64_t)2 * y.dirs * y.dirs * max_T * y.H * 32);
    if (workspace_bytes < need) {
        snn_set_error("snn_lstm_backward: workspace %lld < required %lld bytes", (long long)workspace_bytes, (long long)need);
        return SNN_ERR_WORKSPACE;
    }
    g_last_bwd_path = "fused";
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
    if ((rc = snn_lstm_only(y, "snn_bbb_backward"))) return rc;
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
// (nn.CrossEntropyLoss() at utils/training_utils.py:515).  Up to CE_MAX_BLOCKS workgroups, each a fixed-order
// partial sum in fp64 over its curves, written to the CALLER's `partials`; a one-thread kernel then adds the
// partials in index order, so the loss does not depend on scheduling and the library keeps no state of its own.
// A batch of up to 1024 curves is one workgroup and one launch.  A label outside [0, C) (torch raises there) makes
// the loss NaN and zeroes that curve's gradient row.
// ============================================================================
#define CE_MAX_BLOCKS 256

__global__ void __launch_bounds__(256) ce_loss_grad_kernel(int N, int C, const float* __restrict__ logits,
                                                          const int64_t* __restrict__ target, float scale,
                                                          float* __restrict__ loss, float* __restrict__ dlogits,
                                                          double* __restrict__ partials, const float* __restrict__ kl,
                                                          float kl_scale)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const float* lg = logits + (int64_t)n * C;
        float mx = -INFINITY;
        for (int q = 0; q < C; ++q) mx = fmaxf(mx, lg[q]);
        float z = 0.0f;
        for (int q = 0; q < C; ++q) z += expf(lg[q] - mx);
        const float lz = logf(z);
        const int64_t t64 = target[n];
        const bool valid = t64 >= 0 && t64 < C;
        const int tq = valid ? (int)t64 : 0;
        acc += valid ? (double)(-(lg[tq] - mx - lz)) : (double)NAN;
        for (int q = 0; q < C; ++q)
            dlogits[(int64_t)n * C + q] = valid ? (expf(lg[q] - mx - lz) - (q == tq ? 1.0f : 0.0f)) * scale : 0.0f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double part = sh[0] + sh[1] + sh[2] + sh[3];
        if (gridDim.x == 1) {
            const float ce = (float)(part / (double)N);
            loss[0] = ce;
            if (kl) loss[1] = ce + kl[0] * kl_scale;      // the step's loss, utils/training_utils.py:515-518
        } else partials[blockIdx.x] = part;
    }
}

__global__ void ce_loss_sum_kernel(int N, int blocks, const double* __restrict__ partials, float* __restrict__ loss,
                                   const float* __restrict__ kl, float kl_scale)
{
    double tot = 0.0;
    for (int b = 0; b < blocks; ++b) tot += partials[b];
    const float ce = (float)(tot / (double)N);
    loss[0] = ce;
    if (kl) loss[1] = ce + kl[0] * kl_scale;
}

extern "C" int64_t snn_ce_loss_partials(int32_t n) { return n > 1024 ? CE_MAX_BLOCKS : 0; }

// snn_ce_loss_grad, and -- with kl != NULL -- also loss[1] = CE + kl_scale * kl[0], the loss train_step returns for a
// BayesianRNN (utils/training_utils.py:515-518): used by snn_train_step_grad, whose `loss` then holds two floats
int snn_ce_loss_grad_total(int32_t n, int32_t n_classes, const float* logits, const int64_t* target, float grad_scale,
                           float* loss, float* dlogits, double* partials, const float* kl, float kl_scale, void* stream);
extern "C" int snn_ce_loss_grad(int32_t n, int32_t n_classes, const float* logits, const int64_t* target,
                                float grad_scale, float* loss, float* dlogits, double* partials, void* stream)
{
    return snn_ce_loss_grad_total(n, n_classes, logits, target, grad_scale, loss, dlogits, partials, nullptr, 0.0f, stream);
}
int snn_ce_loss_grad_total(int32_t n, int32_t n_classes, const float* logits, const int64_t* target, float grad_scale,
                           float* loss, float* dlogits, double* partials, const float* kl, float kl_scale, void* stream)
{
    if (!logits || !target || !loss || !dlogits || n < 1 || n_classes < 1) {
        snn_set_error("snn_ce_loss_grad: null pointer or bad size");
        return SNN_ERR_ARG;
    }
    // one workgroup per 1024 curves (4 per thread): a batch of 128 stays a single workgroup and a single launch
    const int blocks = std::min(CE_MAX_BLOCKS, (n + 1023) / 1024);
    if (blocks > 1 && (!partials || ((uintptr_t)partials & 7))) {
        snn_set_error("snn_ce_loss_grad: n = %d > 1024 needs `partials` (snn_ce_loss_partials(n) = %d doubles, 8-byte aligned)",
                      n, CE_MAX_BLOCKS);
        return SNN_ERR_ARG;
    }
    hipLaunchKernelGGL(ce_loss_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, n_classes, logits, target,
                       grad_scale, loss, dlogits, partials, kl, kl_scale);
    if (blocks > 1)
        hipLaunchKernelGGL(ce_loss_sum_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, n, blocks, partials, loss, kl, kl_scale);
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

// ============================================================================
// Fused SGD with momentum over one flat buffer (torch.optim.SGD semantics, dampening 0, no Nesterov:
// training/train_rnn.py:541-546 -- the optimizer of train_swag_finetune)
// ============================================================================
__global__ void __launch_bounds__(256) sgd_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                 float* __restrict__ buf, float lr, float momentum, float weight_decay,
                                                 int first, float grad_scale)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float pi = p[i];
    float gi = g[i] * grad_scale;
    if (weight_decay != 0.0f) gi = fmaf(weight_decay, pi, gi);
    if (momentum != 0.0f) {
        gi = first ? gi : fmaf(momentum, buf[i], gi);      // the first step copies the gradient into the buffer
        buf[i] = gi;
    }
    p[i] = pi - lr * gi;
}

extern "C" int snn_sgd_step(int64_t n, float* params, const float* grads, float* momentum_buf, int64_t step, float lr,
                            float momentum, float weight_decay, float grad_scale, void* stream)
{
    if (!params || !grads || (momentum != 0.0f && !momentum_buf) || n < 1 || step < 1) {
        snn_set_error("snn_sgd_step: null pointer, n < 1 or step < 1");
        return SNN_ERR_ARG;
    }
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, params, grads,
                       momentum_buf, lr, momentum, weight_decay, step == 1 ? 1 : 0, grad_scale);
    HIP_CHECK(hipGetLastError());
    return SNN_OK;
}
