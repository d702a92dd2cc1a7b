// snn_layout.h -- parameter layouts shared by host code and kernels.
//
// Canonical flat order (torch _flat_weights; eps draw order of the reference,
// training/bayesian_rnn.py:472-528): per (layer, dir) w_ih, w_hh, b_ih, b_hh;
// then head weight, head bias.
//
// Kernel layout of one weight set (what the persistent LSTM kernel loads
// straight into MFMA A-operand registers):
//   per (layer, dir):
//     A image  [NW][KK][64]   A[w][kk][lane] = Wcat[row(w, lane&31)][2*kk + (lane>>5)]
//              Wcat = [w_ih | zero pad to 2*KKI columns | w_hh] (4H x (2*KKI + H)), KKI of layer 0 from
//              snn_kki0_bucket(D),
//              row(w, r) = (r>>3)*H + 8*w + (r&7)   (gate r>>3 of hidden unit 8w + (r&7)),
//              so the 32x32 MFMA tile of wave w holds the 4 gates of 8 units and each
//              lane of the C/D tile ends up with all 4 gates of 4 units;
//     bias image [2][NW][2][16]  (b_ih then b_hh) in C/D register order:
//              entry (w, hf, i) = bias[(i>>2)*H + 8*w + 4*hf + (i&3)]
//   then head weight (C x F) and head bias (C), canonical.
#pragma once
#include <stdint.h>

#define SNN_MAX_LAYERS 4
#define SNN_MAX_LD (2 * SNN_MAX_LAYERS)
#define SNN_TILE 32

struct SnnLayout {
    int D, H, L, dirs, C, pool, F;
    int Dp;  // D padded to the layer-0 k-step bucket (2 * snn_kki0_bucket(D))
    int NW;  // waves per workgroup = H / 8
    int I[SNN_MAX_LAYERS];    // input width of layer l
    int KKI[SNN_MAX_LAYERS];  // ceil(I/2): MFMA k-steps of the input part
    int KK[SNN_MAX_LAYERS];   // KKI + H/2
    int64_t c_wih[SNN_MAX_LD], c_whh[SNN_MAX_LD], c_bih[SNN_MAX_LD], c_bhh[SNN_MAX_LD];
    int64_t k_A[SNN_MAX_LD], k_bias[SNN_MAX_LD];
    int64_t c_head_w, c_head_b, k_head_w, k_head_b;
    int64_t P, PK;
    int n_tensors;  // sampled tensors: 4*L*dirs + 1
    // precision 1 (bf16 x 3 pieces, snn_forward_b3.hip): the A image of (layer, dir) is
    //   [NW][NB][3 pieces][64 lanes][8 bf16]   lane (r = lane & 31, khalf = lane >> 5) holds
    //   Wcat3[row(w, r)][16 b + 8 khalf + e], Wcat3 = [w_ih | zero pad to 16*XB | w_hh], NB = XB + H/16;
    // k_A / k_bias / k_head_* / PK then describe that image (in 4-byte words); bias and head stay fp32.
    // precision 2 (fp16 x 2 pieces, snn_forward_h2.hip): same image with 2 pieces of 8 fp16 per lane, hi = fp16(v),
    //   lo = fp16((v - hi) * 2^11); the weight rows are pre-scaled by the gate's exp2 factor (-log2 e for i, f, o;
    //   2 log2 e for g), see snn_split2_scale.
    int prec;
    int NPC;   // pieces per value: 3 (precision 1), 2 (precision 2)
    // recurrent cell (snn_cfg.cell): 0 LSTM, 1 GRU, 2 tanh RNN.  NG = gate blocks of the canonical tensors (4 / 3 / 1).
    // The kernel image always has the LSTM's 4 row groups per hidden unit, so one MFMA tiling serves all cells:
    //   GRU   group 0 = r (W_ir | W_hr), 1 = z (W_iz | W_hz), 2 = n, input part (W_in | 0), 3 = n, hidden part
    //         (0 | W_hn); biases alike (b_in in group 2, b_hn in group 3), so that n = tanh(a2 + r * a3);
    //   RNN   group 0 = (W_ih | W_hh), groups 1-3 zero.
    int cell, NG;
    int XB[SNN_MAX_LAYERS], NB[SNN_MAX_LAYERS];
};

// MFMA k-steps of the layer-0 input part.  The kernels are instantiated for a few widths; an input size in
// between is zero-padded up to the next one (zero weight columns x zero inputs: results unchanged), so any
// feature count up to SNN_MAX_D runs: 4 filters (D = 24, +2 with redshift), 5 filters (42/44), 6 filters (76/78).
#define SNN_MAX_D 78
static inline int snn_kki0_bucket(int D)
{
    static const int buckets[] = {12, 13, 24, 38, 39};
    for (int i = 0; i < 5; ++i)
        if (D <= 2 * buckets[i]) return buckets[i];
    return (D + 1) / 2;
}

// kernel row group of canonical gate block `gate` of a hidden-side tensor (w_hh, b_hh) / input-side tensor
static inline __host__ __device__ int snn_row_group(int cell, int gate, bool hidden_side)
{
    return (cell == 1 && gate == 2 && hidden_side) ? 3 : gate;
}

static inline int snn_layout_init(SnnLayout* y, int D, int H, int L, int bidir, int C, int pool, int prec = 0, int cell = 0)
{
    if (D < 1 || H < 8 || (H % 8) != 0 || L < 1 || L > SNN_MAX_LAYERS || C < 1 || pool < 0 || pool > 1 || prec < 0 ||
        prec > 2 || (prec >= 1 && (H % 16) != 0) || cell < 0 || cell > 2 || (cell != 0 && prec != 0))
        return -1;
    y->prec = prec;
    y->cell = cell;
    y->NG = cell == 0 ? 4 : cell == 1 ? 3 : 1;
    const int NG = y->NG;
    y->NPC = prec == 2 ? 2 : 3;
    y->D = D; y->H = H; y->L = L; y->dirs = bidir ? 2 : 1; y->C = C; y->pool = pool;
    y->F = pool == 0 ? y->dirs * H : y->dirs * H * L;
    y->Dp = 2 * snn_kki0_bucket(D);
    y->NW = H / 8;
    int64_t c = 0, k = 0;
    for (int l = 0; l < L; ++l) {
        y->I[l] = l == 0 ? D : y->dirs * H;
        y->KKI[l] = l == 0 ? y->Dp / 2 : (y->I[l] + 1) / 2;
        y->KK[l] = y->KKI[l] + H / 2;
        y->XB[l] = (2 * y->KKI[l] + 15) / 16;
        y->NB[l] = y->XB[l] + H / 16;
        for (int d = 0; d < y->dirs; ++d) {
            int ld = l * y->dirs + d;
            y->c_wih[ld] = c; c += (int64_t)NG * H * y->I[l];
            y->c_whh[ld] = c; c += (int64_t)NG * H * H;
            y->c_bih[ld] = c; c += NG * H;
            y->c_bhh[ld] = c; c += NG * H;
            y->k_A[ld] = k; k += prec == 0 ? (int64_t)y->NW * y->KK[l] * 64 : (int64_t)y->NW * y->NB[l] * y->NPC * 64 * 4;
            y->k_bias[ld] = k; k += (int64_t)2 * y->NW * 32;
        }
    }
    y->c_head_w = c; c += (int64_t)C * y->F;
    y->c_head_b = c; c += C;
    y->k_head_w = k; k += (int64_t)C * y->F;
    y->k_head_b = k; k += C;
    y->P = c;
    y->PK = (k + 3) & ~(int64_t)3;  // keep weight sets 16-byte aligned
    y->n_tensors = 4 * L * y->dirs + 1;
    return 0;
}
