/*
 * oracle/snn_oracle.c -- CPU restatement of SuperNNova's RNN hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in supernnova_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Plain scalar C loops (no BLAS, no vectorisation assumptions) restating, for
 * ONE set of weights:
 *   - the LSTM equations of torch.nn.LSTM / torch._VF.lstm (third-party
 *     arithmetic, un-vendored; reference pins pytorch>=2.1,<2.5 in
 *     env/conda_env.yml:13; call sites
 *     python/supernnova/training/vanilla_rnn.py:26-32,47,
 *     variational_rnn.py:37-43,244, bayesian_rnn.py:531-553):
 *       gates = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh, split (i,f,g,o);
 *       c_t = sig(f) c_{t-1} + sig(i) tanh(g);  h_t = sig(o) tanh(c_t);
 *       zero initial state; packed semantics (each curve runs for its own
 *       length, the reverse direction starts at the curve's last step);
 *   - pooling "mean" (sum over valid steps / length, vanilla_rnn.py:74-83) and
 *     "standard" (hn.permute(1,2,0).view(B,-1): hidden-major,
 *     vanilla_rnn.py:60-72);
 *   - the variational masks (variational_rnn.py:161-169 locked (1,B,D) /
 *     (1,B,2H); :224-238 W_hh row mask on the forward direction only;
 *     :182-188 output dropout), all supplied by the caller;
 *   - the output Linear / BayesLinear (bayesian_rnn.py:595-616) and softmax
 *     (validation/validate_rnn.py:59-60);
 *   - Bayes-by-Backprop sampling w = mu + eps*(softplus(rho)+1e-5)
 *     (bayesian_rnn.py:472-528, 595-616) and compute_KL (:794-815).
 *
 * Pinned by tests/golden/ fixtures generated from the imported reference
 * (tools/make_golden.py); see tests/test_oracle_golden.py.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (REAL=float, the default) and
 * a second time with -DSNN_REAL=double for an f64 "truth" used in error
 * budgets.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef SNN_REAL
#define SNN_REAL float
#endif
typedef SNN_REAL real;

typedef struct {
    int32_t input_size;    /* D */
    int32_t hidden_size;   /* H */
    int32_t num_layers;    /* L */
    int32_t bidirectional; /* 0/1 */
    int32_t nb_classes;    /* C */
    int32_t pool;          /* 0 = mean, 1 = standard */
    int32_t cell;          /* 0 = lstm, 1 = gru, 2 = rnn (tanh): VanillaRNN's layer_type, vanilla_rnn.py:26-32 */
} snn_oracle_cfg;

/* gate rows per hidden unit: LSTM i,f,g,o; GRU r,z,n; RNN one pre-activation (torch.nn.{LSTM,GRU,RNN} layouts) */
static int n_gates(const snn_oracle_cfg *c) { return c->cell == 1 ? 3 : c->cell == 2 ? 1 : 4; }

static real sigm(real x) { return (real)1 / ((real)1 + (real)exp(-(double)x)); }
static real tanhr(real x) { return (real)tanh((double)x); }

/* canonical flat parameter layout (the order of torch's _flat_weights):
 * for layer, for direction: w_ih (4H x I_l), w_hh (4H x H), b_ih (4H), b_hh (4H);
 * then head weight (C x F), head bias (C). */
static int64_t layer_in(const snn_oracle_cfg *c, int l)
{
    int dirs = c->bidirectional ? 2 : 1;
    return l == 0 ? c->input_size : (int64_t)dirs * c->hidden_size;
}

int64_t snn_oracle_feat(const snn_oracle_cfg *c)
{
    int dirs = c->bidirectional ? 2 : 1;
    return c->pool == 0 ? (int64_t)dirs * c->hidden_size
                        : (int64_t)dirs * c->hidden_size * c->num_layers;
}

int64_t snn_oracle_param_count(const snn_oracle_cfg *c)
{
    int dirs = c->bidirectional ? 2 : 1;
    int64_t H = c->hidden_size, p = 0;
    for (int l = 0; l < c->num_layers; ++l)
        p += dirs * (n_gates(c) * H * (layer_in(c, l) + H + 2));
    p += (int64_t)c->nb_classes * snn_oracle_feat(c) + c->nb_classes;
    return p;
}

/*
 * Forward of one weight set over a padded, time-major batch.
 *   x          (T, B, D) fp32, zero padded
 *   lengths    (B) true lengths (1..T), any order
 *   w          canonical flat parameters
 *   mask_in    (B, D) or NULL            -- variational_rnn.py:165-169
 *   mask_whh   (L, 4H) or NULL           -- forward-direction W_hh row mask
 *   mask_mid   ((L-1), mid_T, B, dirs*H) or NULL; mid_T == 1 is the locked
 *              mask of variational_rnn.py:88-91, mid_T == T a per-step mask
 *              (torch.nn.LSTM's own inter-layer dropout in train mode)
 *   mask_out   (B, F) or NULL            -- output dropout
 *   logits, probs (B, C) (either may be NULL)
 *   pooled     (B, F) or NULL: features fed to the head (after mask_out)
 */
int snn_oracle_forward(const snn_oracle_cfg *cfg, const float *x, int T, int B,
                       const int32_t *lengths, const float *w,
                       const float *mask_in, const float *mask_whh,
                       const float *mask_mid, int mid_T, const float *mask_out,
                       float *logits, float *probs, float *pooled_out)
{
    const int D = cfg->input_size, H = cfg->hidden_size, L = cfg->num_layers;
    const int dirs = cfg->bidirectional ? 2 : 1, C = cfg->nb_classes;
    const int F = (int)snn_oracle_feat(cfg);
    const int G = n_gates(cfg) * H;
    real *gih = (real *)malloc(sizeof(real) * G), *ghh = (real *)malloc(sizeof(real) * G);
    int maxI = D > dirs * H ? D : dirs * H;
    real *in = (real *)malloc(sizeof(real) * (size_t)T * maxI);
    real *out = (real *)malloc(sizeof(real) * (size_t)T * dirs * H);
    real *gates = (real *)malloc(sizeof(real) * G);
    real *h = (real *)malloc(sizeof(real) * H), *c = (real *)malloc(sizeof(real) * H);
    real *hn = (real *)malloc(sizeof(real) * (size_t)L * dirs * H);
    real *feat = (real *)malloc(sizeof(real) * F);
    real *lg = (real *)malloc(sizeof(real) * C);
    if (!in || !out || !gates || !h || !c || !hn || !feat || !lg) return -1;

    for (int b = 0; b < B; ++b) {
        const int len = lengths[b];
        if (len < 1 || len > T) return -2;
        for (int t = 0; t < len; ++t)
            for (int k = 0; k < D; ++k) {
                real v = x[((size_t)t * B + b) * D + k];
                if (mask_in) v = (real)mask_in[(size_t)b * D + k] * v;
                in[(size_t)t * maxI + k] = v;
            }
        const float *p = w;
        for (int l = 0; l < L; ++l) {
            const int I = (int)layer_in(cfg, l);
            for (int d = 0; d < dirs; ++d) {
                const float *w_ih = p; p += (size_t)G * I;
                const float *w_hh = p; p += (size_t)G * H;
                const float *b_ih = p; p += G;
                const float *b_hh = p; p += G;
                for (int j = 0; j < H; ++j) h[j] = c[j] = 0;
                for (int s = 0; s < len; ++s) {
                    const int t = d ? len - 1 - s : s;
                    const real *xt = in + (size_t)t * maxI;
                    for (int r = 0; r < G; ++r) {
                        real ig = 0, hg = 0;
                        for (int k = 0; k < I; ++k) ig += (real)w_ih[(size_t)r * I + k] * xt[k];
                        ig += (real)b_ih[r];
                        real m = (mask_whh && d == 0) ? (real)mask_whh[(size_t)l * G + r] : (real)1;
                        for (int k = 0; k < H; ++k) {
                            real wv = (real)w_hh[(size_t)r * H + k];
                            if (mask_whh && d == 0) wv = (real)(float)(m * wv); /* mask*raw_w rounded to fp32 */
                            hg += wv * h[k];
                        }
                        hg += (real)b_hh[r];
                        gates[r] = ig + hg;
                        gih[r] = ig;
                        ghh[r] = hg;
                    }
                    for (int j = 0; j < H; ++j) {
                        if (cfg->cell == 1) {
                            /* torch.nn.GRU: r, z, n rows; n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1-z) n + z h */
                            real rg = sigm(gates[j]), zg = sigm(gates[H + j]);
                            real ng = tanhr(gih[2 * H + j] + rg * ghh[2 * H + j]);
                            c[j] = ((real)1 - zg) * ng + zg * h[j];
                        } else if (cfg->cell == 2) {
                            c[j] = tanhr(gates[j]);                       /* torch.nn.RNN, nonlinearity tanh */
                        } else {
                            real ig = sigm(gates[j]), fg = sigm(gates[H + j]);
                            real gg = tanhr(gates[2 * H + j]);
                            c[j] = fg * c[j] + ig * gg;
                        }
                    }
                    for (int j = 0; j < H; ++j) {
                        if (cfg->cell == 0) h[j] = sigm(gates[3 * H + j]) * tanhr(c[j]);
                        else h[j] = c[j];                                /* GRU / RNN: c held the new h of all units */
                        out[(size_t)t * dirs * H + d * H + j] = h[j];
                    }
                }
                for (int j = 0; j < H; ++j) hn[((size_t)l * dirs + d) * H + j] = h[j];
            }
            if (l < L - 1) {
                for (int t = 0; t < len; ++t)
                    for (int k = 0; k < dirs * H; ++k) {
                        real v = out[(size_t)t * dirs * H + k];
                        if (mask_mid) {
                            int tt = mid_T == 1 ? 0 : t;
                            v = (real)mask_mid[(((size_t)l * mid_T + tt) * B + b) * dirs * H + k] * v;
                        }
                        in[(size_t)t * maxI + k] = v;
                    }
            }
        }
        const float *w_out = p; p += (size_t)C * F;
        const float *b_out = p;
        if (cfg->pool == 0) {
            for (int k = 0; k < dirs * H; ++k) {
                real s = 0;
                for (int t = 0; t < len; ++t) s += out[(size_t)t * dirs * H + k];
                feat[k] = s / (real)len;
            }
        } else {
            /* hn (L*dirs, B, H).permute(1,2,0).view(B,-1): index = j*(L*dirs) + (l*dirs+d) */
            for (int ld = 0; ld < L * dirs; ++ld)
                for (int j = 0; j < H; ++j) feat[(size_t)j * L * dirs + ld] = hn[(size_t)ld * H + j];
        }
        if (mask_out)
            for (int k = 0; k < F; ++k) feat[k] = (real)mask_out[(size_t)b * F + k] * feat[k];
        if (pooled_out)
            for (int k = 0; k < F; ++k) pooled_out[(size_t)b * F + k] = (float)feat[k];
        real mx = -INFINITY;
        for (int q = 0; q < C; ++q) {
            real s = 0;
            for (int k = 0; k < F; ++k) s += (real)w_out[(size_t)q * F + k] * feat[k];
            s += (real)b_out[q];
            lg[q] = s;
            if (s > mx) mx = s;
            if (logits) logits[(size_t)b * C + q] = (float)s;
        }
        if (probs) {
            real z = 0;
            for (int q = 0; q < C; ++q) z += (real)exp((double)(lg[q] - mx));
            for (int q = 0; q < C; ++q)
                probs[(size_t)b * C + q] = (float)((real)exp((double)(lg[q] - mx)) / z);
        }
    }
    free(gih); free(ghh);
    free(in); free(out); free(gates); free(h); free(c); free(hn); free(feat); free(lg);
    return 0;
}

/* softplus as torch.nn.functional.softplus(beta=1, threshold=20) */
static float softplus_f(float r)
{
    return r > 20.0f ? r : (float)log1p(exp((double)r));
}

/*
 * Bayes-by-Backprop sampling of n parameters (bayesian_rnn.py:480-523, 598-609):
 *   sigma = softplus(rho) + 1e-5;  w = mu + eps * sigma   (eps == NULL: w = mu,
 *   the mean_field_inference branch).  fp32, product and sum rounded
 *   separately as torch's two element-wise kernels do.
 */
void snn_oracle_bbb_sample(int64_t n, const float *mu, const float *rho,
                           const float *eps, float *w, float *sigma_out)
{
    for (int64_t i = 0; i < n; ++i) {
        float sg = softplus_f(rho[i]) + 1e-5f;
        if (sigma_out) sigma_out[i] = sg;
        if (eps) {
            volatile float pr = eps[i] * sg;
            w[i] = mu[i] + pr;
        } else {
            w[i] = mu[i];
        }
    }
}

/*
 * compute_KL of ONE tensor (bayesian_rnn.py:794-815):
 *   log q = sum log N(w; mu, sigma)
 *   log p = logsumexp([sum log N(w;0,s1) + log pi, sum log N(w;0,s2) + log(1-pi)])
 * Element terms in fp32 like torch.distributions.Normal.log_prob
 *   -((x-mu)^2)/(2 var) - log(scale) - log(sqrt(2 pi)); sums carried in double
 * and rounded to fp32 once (torch's cascade sum is within 1 ulp of that).
 */
float snn_oracle_kl_tensor(int64_t n, const float *w, const float *mu, const float *sigma,
                           double pi_mix, double sigma1, double sigma2)
{
    const float lsq = (float)log(sqrt(2.0 * M_PI));
    const float s1 = (float)sigma1, s2 = (float)sigma2;
    const float v1 = s1 * s1, v2 = s2 * s2, l1 = logf(s1), l2 = logf(s2);
    double q = 0, p1 = 0, p2 = 0;
    for (int64_t i = 0; i < n; ++i) {
        float d = w[i] - mu[i];
        float var = sigma[i] * sigma[i];
        q += (double)(-(d * d) / (2.0f * var) - logf(sigma[i]) - lsq);
        float x2 = w[i] * w[i];
        p1 += (double)(-(x2) / (2.0f * v1) - l1 - lsq);
        p2 += (double)(-(x2) / (2.0f * v2) - l2 - lsq);
    }
    float m1 = (float)p1 + (float)log(pi_mix), m2 = (float)p2 + (float)log(1.0 - pi_mix);
    float mx = m1 > m2 ? m1 : m2;
    float lse = mx + logf(expf(m1 - mx) + expf(m2 - mx));
    return (float)q - lse;
}

/*
 * KL of a whole BayesianRNN (bayesian_rnn.py:470-528 + :614, summed at :137):
 * the 16 LSTM tensors use the LSTM prior, the head weight the output prior;
 * the head bias is deterministic (bayesian_rnn.py:587-588) and excluded.
 * w/mu/sigma: canonical flat buffers of length param_count - C.
 */
float snn_oracle_kl(const snn_oracle_cfg *cfg, const float *w, const float *mu, const float *sigma,
                    double pi_mix, double sigma1, double sigma2,
                    double sigma1_out, double sigma2_out)
{
    const int dirs = cfg->bidirectional ? 2 : 1;
    const int64_t H = cfg->hidden_size, G = 4 * H;
    float kl = 0;
    int64_t off = 0;
    for (int l = 0; l < cfg->num_layers; ++l)
        for (int d = 0; d < dirs; ++d) {
            int64_t sz[4] = {G * layer_in(cfg, l), G * H, G, G};
            for (int q = 0; q < 4; ++q) {
                kl += snn_oracle_kl_tensor(sz[q], w + off, mu + off, sigma + off, pi_mix, sigma1, sigma2);
                off += sz[q];
            }
        }
    int64_t nh = (int64_t)cfg->nb_classes * snn_oracle_feat(cfg);
    kl += snn_oracle_kl_tensor(nh, w + off, mu + off, sigma + off, pi_mix, sigma1_out, sigma2_out);
    return kl;
}
