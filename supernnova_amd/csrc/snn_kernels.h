// snn_kernels.h -- structures and device helpers shared by the forward and the
// training kernels (mask evaluation, barriers, tape layout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/snn_hip.h"
#include "snn_device.h"
#include "snn_layout.h"

// v_mfma_f32_32x32x2_f32 with its A operand read from an AGPR, as inline asm (the builtin only takes VGPR sources):
//   asm volatile(SNN_AGPR_MFMA : "+a"(acc) : "a"(w), "v"(b));
// The compiler does not know this is a matrix instruction, so it inserts none of the wait states one needs.  Before it:
// when the value lives in a VGPR the compiler copies it with v_accvgpr_write_b32 right in front of the asm, and an
// MFMA that reads a register a VALU instruction wrote needs 2 wait states (a version whose copies sat directly in front
// of the MFMAs computed wrong W_hh products, caught by the fixtures) -- the s_nop 1 here provides them whatever the
// allocation; it issues in the shadow of the previous MFMA.  After it: the caller puts `s_nop 15; s_nop 3` between the
// last such MFMA and the first VALU read of its accumulator.  microbench/check_agpr_hazard.py scans a listing for both.
#define SNN_AGPR_MFMA "s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0"

// Dropout masks as the kernels see them (host: snn_masks).
struct MaskP {
    int mode, which, mid_T;
    float keep_scale, keep_prob;
    uint64_t seed, offset;
    uint32_t curve_base;
    const float *m_in, *m_whh, *m_mid, *m_out;
};

// All four return 1.0 when the mask does not exist.  `s` = MC sample, `orig` = original curve index.
__device__ __forceinline__ float mask_in(const MaskP& m, int s, int n_curves, int D, int orig, int k)
{
    if (m.mode == SNN_MASK_ARRAYS) return m.m_in ? m.m_in[((int64_t)s * n_curves + orig) * D + k] : 1.0f;
    if (m.mode == SNN_MASK_PHILOX && (m.which & 1))
        return philox_mask(m.seed, m.offset + s, SNN_STREAM_MASK_IN, m.curve_base + (uint32_t)orig, (uint32_t)k,
                           m.keep_prob, m.keep_scale);
    return 1.0f;
}
__device__ __forceinline__ float mask_whh(const MaskP& m, int s, int L, int H, int l, int R)
{
    if (m.mode == SNN_MASK_ARRAYS) return m.m_whh ? m.m_whh[((int64_t)s * L + l) * 4 * H + R] : 1.0f;
    if (m.mode == SNN_MASK_PHILOX && (m.which & 2))
        return philox_mask(m.seed, m.offset + s, SNN_STREAM_MASK_WHH, (uint32_t)R, (uint32_t)l, m.keep_prob,
                           m.keep_scale);
    return 1.0f;
}
// feature = index inside the dirs*H output of layer l (the input of layer l+1); t ignored for locked masks
__device__ __forceinline__ float mask_mid(const MaskP& m, int s, int L, int n_curves, int dirsH, int l, int t,
                                          int orig, int feature)
{
    if (m.mode == SNN_MASK_ARRAYS) {
        if (!m.m_mid) return 1.0f;
        const int tt = m.mid_T > 1 ? t : 0;
        return m.m_mid[((((int64_t)s * (L - 1) + l) * m.mid_T + tt) * n_curves + orig) * dirsH + feature];
    }
    if (m.mode == SNN_MASK_PHILOX && (m.which & 4)) {
        const uint32_t tkey = m.mid_T > 1 ? (uint32_t)(t + 1) : 0u;
        return philox_mask(m.seed, m.offset + s, SNN_STREAM_MASK_MID, m.curve_base + (uint32_t)orig,
                           (uint32_t)(l * dirsH + feature) | (tkey << 16), m.keep_prob, m.keep_scale);
    }
    return 1.0f;
}
__device__ __forceinline__ float mask_out(const MaskP& m, int s, int n_curves, int F, int orig, int fidx)
{
    if (m.mode == SNN_MASK_ARRAYS) return m.m_out ? m.m_out[((int64_t)s * n_curves + orig) * F + fidx] : 1.0f;
    if (m.mode == SNN_MASK_PHILOX && (m.which & 8))
        return philox_mask(m.seed, m.offset + s, SNN_STREAM_MASK_OUT, m.curve_base + (uint32_t)orig, (uint32_t)fidx,
                           m.keep_prob, m.keep_scale);
    return 1.0f;
}

// ---- canonical <-> kernel layout index math (snn_layout.h) -------------------------
struct TensorGeom {
    int64_t cbase;  // canonical offset
    int64_t n;      // elements
    int cols;       // row length (1 for biases)
    int kind;       // 0 w_ih, 1 w_hh, 2 b_ih, 3 b_hh, 4 head w, 5 head b
    int ld;
};

__device__ __forceinline__ TensorGeom tensor_geom(const SnnLayout& y, int tau)
{
    TensorGeom g;
    const int nld = y.L * y.dirs;
    if (tau < 4 * nld) {
        g.ld = tau >> 2;
        g.kind = tau & 3;
        const int l = g.ld / y.dirs;
        if (g.kind == 0) { g.cbase = y.c_wih[g.ld]; g.cols = y.I[l]; g.n = (int64_t)y.NG * y.H * y.I[l]; }
        else if (g.kind == 1) { g.cbase = y.c_whh[g.ld]; g.cols = y.H; g.n = (int64_t)y.NG * y.H * y.H; }
        else if (g.kind == 2) { g.cbase = y.c_bih[g.ld]; g.cols = 1; g.n = y.NG * y.H; }
        else { g.cbase = y.c_bhh[g.ld]; g.cols = 1; g.n = y.NG * y.H; }
    } else if (tau == 4 * nld) {
        g.ld = 0; g.kind = 4; g.cbase = y.c_head_w; g.cols = y.F; g.n = (int64_t)y.C * y.F;
    } else {
        g.ld = 0; g.kind = 5; g.cbase = y.c_head_b; g.cols = 1; g.n = y.C;
    }
    return g;
}

// canonical element i of tensor g -> index in the kernel layout
__device__ __forceinline__ int64_t kernel_index(const SnnLayout& y, const TensorGeom& g, int i)   // (32-bit: a 64-bit division per element was most of bbb_sample_kernel)
{
    if (g.kind == 4) return y.k_head_w + i;
    if (g.kind == 5) return y.k_head_b + i;
    const int l = g.ld / y.dirs;
    const int H = y.H;
    if (g.kind <= 1) {
        const int R = i / g.cols, col = i - R * g.cols;
        const int gate = R / H, j = R - gate * H;
        const int w = j >> 3, r = snn_row_group(y.cell, gate, g.kind == 1) * 8 + (j & 7);
        const int k = g.kind == 0 ? col : 2 * y.KKI[l] + col;
        const int kk = k >> 1, lane = (k & 1) * 32 + r;
        return y.k_A[g.ld] + ((int64_t)(w * y.KK[l] + kk)) * 64 + lane;
    }
    const int R = (int)i, gate = R / H, j = R - gate * H;
    const int w = j >> 3, hf = (j & 7) >> 2, ii = snn_row_group(y.cell, gate, g.kind == 3) * 4 + (j & 3);
    return y.k_bias[g.ld] + (g.kind == 3 ? y.NW * 32 : 0) + (w * 2 + hf) * 16 + ii;
}

// precision 1: index (in bf16 units from the start of the weight set) of piece 0 of canonical element i of a
// weight matrix g (kind 0 / 1); piece p sits 64 * 8 * p further
__device__ __forceinline__ int64_t kernel_index3(const SnnLayout& y, const TensorGeom& g, int i)
{
    const int l = g.ld / y.dirs, H = y.H;
    const int R = i / g.cols, col = i - R * g.cols;
    const int gate = R / H, j = R - gate * H;
    const int w = j >> 3, r = gate * 8 + (j & 7);
    const int k = g.kind == 0 ? col : 16 * y.XB[l] + col;
    const int b = k >> 4, lane = ((k >> 3) & 1) * 32 + r, e = k & 7;
    return y.k_A[g.ld] * 2 + ((((int64_t)w * y.NB[l] + b) * y.NPC) * 64 + lane) * 8 + e;
}
// precision 2: x = hi + lo * 2^-11 with hi = fp16(x) and lo = fp16((x - hi) * 2^11): the remainder is exact in fp32
// and has at most 12 significant bits, so the pair carries x to one fp32 ulp at worst (exactly for 3 values in 4).
// fp16 subnormals are produced by the conversion and honoured by v_mfma_f32_32x32x16_f16 on gfx950
// (microbench/f16_denorm_probe.hip), so small values need no special case; |x| >= 65520 has no fp16 and gives
// inf - inf = NaN downstream (inputs are normalised features).
#define SNN_H2_LO_SCALE 2048.0f
#define SNN_H2_LO_INV 4.8828125e-04f
__device__ __forceinline__ void split2(float x, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * SNN_H2_LO_SCALE);
}
// exp2 factor folded into the weight rows and biases of gate `gate` (0 i, 1 f, 2 g, 3 o) in precision 2:
// sigmoid(a) = 1 / (1 + 2^(-a log2 e)), tanh(a) = 1 - 2 / (1 + 2^(2 a log2 e))
__device__ __forceinline__ float h2_gate_scale(int gate) { return gate == 2 ? 2.88539008177792681f : -1.44269504088896341f; }
// x = hi + mid + lo to 2^-27 relative, each piece a bf16 (round-to-nearest-even; the two remainders are exact in fp32)
__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo)
{
    hi = (__bf16)x;
    const float r1 = x - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}

// LDS-only barrier: global stores of the step stay in flight (they are only
// read after a later __syncthreads()).
// mask bit of the lane ? a : b, as one opaque v_cndmask.  Written as (pure) inline asm because a C++ select lets
// the compiler sink the whole cell update under an `if (valid)` branch, which splits a pipeline phase into
// basic blocks and forbids the MFMA/VALU interleave the phase exists for.
__device__ __forceinline__ float lane_select(unsigned long long mask, float a, float b)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}

__device__ __forceinline__ void lds_barrier()
{
#ifdef SNN_ABL_NOBAR   // timing ablation only (results are wrong): what the per-step rendezvous costs
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// ---- training tape -----------------------------------------------------------
// Written by the training forward, read by the backward.  Per tile:
//   tape[ld][t][q][H][32]   q: 0 i, 1 f, 2 g, 3 o (post-activation), 4 c_t, 5 h_t (0 beyond the curve, unmasked)
//   then feat[F][32] (pooled features after the output mask), then logits are the regular output.
#define SNN_TAPE_Q 6
__host__ __device__ inline int64_t tape_step_floats(int H) { return (int64_t)SNN_TAPE_Q * H * 32; }
__host__ __device__ inline int64_t tape_tile_floats(int H, int n_ld, int max_T, int F)
{
    return (int64_t)n_ld * max_T * tape_step_floats(H) + (int64_t)F * 32;
}

// ---- forward kernel parameters (snn_forward.hip, snn_forward_big.hip) ---------
// tiles of one work item of the fp16 x 2 forward kernel (one pair): the layer-0 output of the items in flight
// (2 workgroups x 256 CUs x 2 tiles x T x 8 KB) then fits the 256 MB Infinity Cache instead of streaming through HBM
#ifndef SNN_H2_G
#define SNN_H2_G 2
#endif
// weight-image offsets of the bf16 x 3 forward kernel (4-byte words from the start of a weight set)
struct B3Offsets { int64_t k_A[SNN_MAX_LD], k_bias[SNN_MAX_LD], k_head_w; };

struct FwdParams {
    int D, Dp, L, C, pool, F;
    const float* xp;
    const int32_t* len_sorted;
    const int32_t* order;
    const int32_t* tile_T;
    const int64_t* tile_off;
    int n_curves, n_tiles, max_T, n_samples;
    const float* wk;
    int64_t wk_stride;
    MaskP mk;
    // outputs
    float* out_logits;
    float* out_probs;
    int64_t os_s, os_n;
    // workspace
    unsigned int* counter;
    float* yscratch;       // [grid][nbuf][dirs][max_T][H*32]
    int64_t y_buf_floats;  // dirs*max_T*H*32
    int nbuf;
    // training tape (TRAIN kernels only): [n_tiles][tape_tile_floats]
    float* tape;
    int64_t tape_tile;
};

struct ItemCtx {
    int tile, s, T, w, lane, cl, hf, tid;
    int len, orig;          // of this lane's curve
    const float* wks;       // weight set of this sample
    float* ybase;           // this workgroup's y scratch
    float* tape;            // TRAIN: this tile's tape
    const float* gx;        // GX kernels: input products b + W_ih x_t of this (tile, direction), [t][H/8][16][64]
};

// host-side mask validation / conversion (snn_forward.hip)
int snn_masks_to_device(const snn_masks* masks, int max_T, MaskP* out);

// ---- backward kernel parameters (snn_train_big.hip, snn_bwd_stream2.h) --------
struct Bwd2Params {
    SnnLayout lay;
    const float* xp;
    const int32_t* len_sorted;
    const int32_t* order;
    const int32_t* tile_T;
    const int64_t* tile_off;
    int n_curves, n_tiles, max_T;
    const float* wk;
    MaskP mk;
    const float* tape;
    int64_t tape_tile;
    const float* dlogits;
    float* WT;                                      // transposed weight images
    int64_t wt_hh[SNN_MAX_LD], wt_ih[SNN_MAX_LD];   // float offsets into WT
    float* dfeat;                                   // [tile][F][32]
    float* dG;                                      // [tile][dir][max_T][4H][32], the layer being processed
    const float* dy_in;                             // [tile][dir][max_T][H][32]: d loss / d (outputs of this layer)
    float* dy_out;                                  // same layout, for the layer below
    float* slabs;                                   // [n_chunks][P]
    float* dflat;
    int accumulate;
};

