/*
 * snn_hip.h -- C ABI of libsnn_hip.so: the MI355X (gfx950) implementation of
 * SuperNNova's RNN hot path (LSTM forward/backward x MC samples, Bayes-by-
 * Backprop sampling + KL, variational masks, head + softmax, Adam).
 *
 * The reference is pure Python on PyTorch and has no FFI of its own; every
 * entry point below replaces a *call site* of the reference, cited per
 * function as <file>:<line> relative to the reference checkout.  The Python
 * host layer (supernnova_amd/) binds these with ctypes and mirrors the
 * reference's nn.Module API; see INTEGRATION.md for the binding a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every device buffer is owned by the caller
 *     (the library never allocates device memory);
 *   - all tensors fp32 contiguous, lengths/indices int32, offsets int64;
 *   - every function returns 0 on success, a negative snn_status otherwise, and
 *     never throws; snn_last_error() returns a thread-local message;
 *   - launches are asynchronous on the given hipStream_t (passed as void*);
 *     calls on different streams with different workspaces may overlap.
 *
 * Canonical flat parameter order ("flat", P floats) -- the order of torch's
 * _flat_weights (torch.nn.LSTM) and of the eps draws in
 * training/bayesian_rnn.py:472-528:
 *   for layer l, for direction d:  w_ih (4H x I_l), w_hh (4H x H), b_ih (4H), b_hh (4H)
 *   then head weight (C x F), head bias (C)
 * with I_0 = D, I_l = dirs*H, F = dirs*H (pool mean) or dirs*H*L (pool standard).
 * Gate order inside 4H is torch's (i, f, g, o).
 */
#ifndef SNN_HIP_H
#define SNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum snn_status {
    SNN_OK = 0,
    SNN_ERR_ARG = -1,         /* bad argument (null pointer, negative size, ...) */
    SNN_ERR_UNSUPPORTED = -2, /* configuration this build has no kernel for */
    SNN_ERR_WORKSPACE = -3,   /* workspace too small */
    SNN_ERR_HIP = -4          /* a HIP runtime call failed */
} snn_status;

#define SNN_POOL_MEAN 0     /* rnn_output_option == "mean"     (vanilla_rnn.py:74-83) */
#define SNN_POOL_STANDARD 1 /* rnn_output_option == "standard" (vanilla_rnn.py:60-72) */

/* settings the modules read: training/vanilla_rnn.py:9-16 */
typedef struct snn_cfg {
    int32_t input_size;    /* D  */
    int32_t hidden_size;   /* H  (settings.hidden_dim) */
    int32_t num_layers;    /* L  */
    int32_t bidirectional; /* 0/1 */
    int32_t nb_classes;    /* C  */
    int32_t pool;          /* SNN_POOL_* */
    int32_t precision;     /* SNN_PREC_*: arithmetic of the gate products in snn_lstm_forward */
    int32_t cell;          /* SNN_CELL_*: settings.layer_type (vanilla_rnn.py:30-40) */
} snn_cfg;

/*
 * SNN_CELL_LSTM  torch.nn.LSTM stack (gate blocks i, f, g, o): every entry point.
 * SNN_CELL_GRU   torch.nn.GRU stack (gate blocks r, z, n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),
 *                h' = (1 - z) * n + z * h), and
 * SNN_CELL_RNN   torch.nn.RNN stack with tanh: VanillaRNN's other two layer types (vanilla_rnn.py:26-32), SNN_PREC_F32,
 *                hidden sizes 32 and 64.  snn_param_count / snn_kparam_count / snn_prepare_weights(_multi) /
 *                snn_lstm_forward (no masks), snn_lstm_forward_train (masks allowed), snn_train_tape_floats and
 *                snn_lstm_backward take them (canonical tensors have 3H resp. H rows; the tape keeps r, z, n and
 *                W_hn h + b_hn resp. h in the LSTM's slots); the Bayesian entry points return SNN_ERR_UNSUPPORTED.
 */
#define SNN_CELL_LSTM 0
#define SNN_CELL_GRU 1
#define SNN_CELL_RNN 2

/*
 * SNN_PREC_F32     every product in fp32 (v_mfma_f32_32x32x2_f32): the default, and the only mode of the
 *                  training entry points.
 * SNN_PREC_BF16X3  inference only: x, h and W are cut into 3 bf16 pieces each (hi + mid + lo = the fp32 value
 *                  to 2^-27) and every fp32 product is rebuilt from 6 bf16 products accumulated in fp32
 *                  (v_mfma_f32_32x32x16_bf16; the dropped cross terms are below 2^-24 relative, i.e. below fp32
 *                  rounding).  Same parity bar; different rounding order.  The packed x image and the weight
 *                  image have a different size and layout in this mode: snn_x_step_floats / snn_kparam_count
 *                  report it, and snn_pack_x, snn_prepare_weights, snn_bbb_sample and snn_lstm_forward must all
 *                  be called with the same precision.
 * SNN_PREC_F16X2   inference only: x, h and W are cut into 2 fp16 pieces, v = hi + lo * 2^-11 (hi = fp16(v),
 *                  lo = fp16((v - hi) * 2^11): one fp32 ulp at worst, exact for 3 values in 4), and every fp32
 *                  product is rebuilt from 3 fp16 products (hi*hi into one fp32 accumulator, hi*lo + lo*hi into a
 *                  second one, combined as main + 2^-11 * cross; the dropped lo*lo term is below 2^-22 relative,
 *                  2^-25 RMS) on v_mfma_f32_32x32x16_f16.  Half the matrix work of BF16X3 at an error of about
 *                  twice fp32 rounding; same parity bar.  Requires |x| < 65504 (normalised features): a larger
 *                  input has no fp16 and yields NaN probabilities for its light curve.  Layout rules as BF16X3.
 *                  Masks (variational dropout) are supported for input widths <= 48 with locked masks (mid_T = 1):
 *                  the kernel applies the Bernoulli BITS of the input, W_hh-row and between-layer masks (mask value
 *                  != 0) exactly, as ANDs on the fp16 pieces, and the CALLER folds their keep_scale = 1/(1-p) into
 *                  the weights before snn_prepare_weights: W_ih of layer 0 (input mask), W_hh of the forward
 *                  directions (W_hh row mask), W_ih of layers >= 1 (between-layer mask); the output mask is applied
 *                  by value (supernnova_amd/modules.py: VariationalRNN._infer_weights).
 */
#define SNN_PREC_F32 0
#define SNN_PREC_BF16X3 1
#define SNN_PREC_F16X2 2

/* Prior of training/bayesian_rnn.py:753-763 (LSTM prior, then the head's). */
typedef struct snn_prior {
    float pi;
    float sigma1, sigma2;         /* exp(log_sigma1), exp(log_sigma2)               */
    float sigma1_out, sigma2_out; /* exp(log_sigma1_output), exp(log_sigma2_output) */
} snn_prior;

#define SNN_MASK_NONE 0   /* eval-mode vanilla, mean_field_inference=True             */
#define SNN_MASK_ARRAYS 1 /* parity mode: masks drawn by the host in reference order  */
#define SNN_MASK_PHILOX 2 /* sampled mode: Bernoulli masks from Philox4x32-10 in-kernel */

/*
 * Dropout masks of training/variational_rnn.py (values are 0 or 1/(1-p)), all
 * indexed by the ORIGINAL curve index n in [0, n_curves):
 *   in   (S, N, D)            locked input mask          :161-169 via :78-80
 *   whh  (S, L, 4H)           W_hh row mask, forward direction only  :224-238
 *   mid  (S, L-1, mid_T, N, dirs*H)  between layers; mid_T == 1 is the locked mask
 *                             of :88-91, mid_T == max_T the per-step mask of
 *                             torch.nn.LSTM(dropout=p) in train mode (vanilla_rnn.py:26-32)
 *   out  (S, N, F)            output dropout             :138-140, 182-188
 * Any pointer may be NULL (= no such mask).  In PHILOX mode the pointers are
 * ignored, `which` selects the masks to draw (bit 0 in, 1 whh, 2 mid, 3 out) and
 * mid_T > 1 asks for a per-step between-layer mask.
 */
typedef struct snn_masks {
    int32_t mode;  /* SNN_MASK_* */
    int32_t which; /* PHILOX mode: bitmask of masks to draw */
    int32_t mid_T; /* 1 (locked) or max_T (per step) */
    float keep_scale; /* PHILOX mode: 1/(1-p) as fp32 */
    float keep_prob;  /* PHILOX mode: 1-p */
    uint64_t seed;    /* PHILOX mode */
    uint64_t offset;  /* PHILOX mode: draw counter of sample 0 (sample s uses offset+s) */
    uint32_t curve_base; /* PHILOX mode: global index of local curve 0 (data-parallel ranks draw distinct per-curve masks) */
    uint32_t reserved;
    const float *in, *whh, *mid, *out;
} snn_masks;

const char *snn_last_error(void);
int snn_version(void);
/* Name, template arguments and launch geometry of the forward kernel the calling thread dispatched last
 * (snn_lstm_forward / _noalloc / _train), e.g. "lstm_fwd_pair_kernel<32,12,2,plain> grid=768 block=256 lds=...";
 * "" before the first launch.  For reports: benchmarks print what ran instead of a hard-coded name. */
const char *snn_last_forward_kernel(void);

/* ---- sizes (host only, no device access) -------------------------------- */
int64_t snn_param_count(const snn_cfg *cfg);  /* P: canonical flat floats incl. head bias, <0 on error */
int64_t snn_kparam_count(const snn_cfg *cfg); /* PK: floats of ONE weight set in kernel layout        */
int32_t snn_feat(const snn_cfg *cfg);         /* F */
int snn_supported(const snn_cfg *cfg);        /* 1 if this build has forward kernels for cfg          */
int64_t snn_x_step_floats(const snn_cfg *cfg); /* 4-byte words of ONE time step of ONE tile in the packed x image */

/*
 * Host-side batching plan; replaces the sort/pad bookkeeping of
 * utils/training_utils.py:449-467 (argsort by length, descending) for ANY
 * number of curves: curves are sorted by length (stable, descending) and cut
 * into tiles of 32; tile i holds sorted positions [32i, 32i+32) and runs for
 * tile_T[i] = its longest curve.  All outputs are HOST arrays:
 *   order      (n_tiles*32) sorted position -> original index (-1 = padding)
 *   len_sorted (n_tiles*32) length at sorted position (0 = padding)
 *   tile_T     (n_tiles)
 *   tile_off   (n_tiles+1)  float offset of each tile in the fp32 packed x buffer
 *                           (tile i occupies tile_T[i] * Dp * 32 floats, Dp = D padded to the next supported width:
 *                           24, 26, 48, 76 or 78); tile_off[n_tiles] / (Dp*32) is the total number of tile-steps,
 *                           and the packed buffer holds that many times snn_x_step_floats(cfg) words
 * n_tiles = ceil(n_curves/32).  Lengths must lie in [1, SNN_MAX_STEPS] (anything else is refused with SNN_ERR_ARG;
 * the bound keeps the sort's bucket table small whatever the caller passes).
 */
#define SNN_MAX_STEPS (1 << 22)
int snn_plan_build(const snn_cfg *cfg, const int32_t *lengths_host, int32_t n_curves,
                   int32_t *order, int32_t *len_sorted, int32_t *tile_T, int64_t *tile_off);

/*
 * Pack padded light curves into the tile layout the LSTM kernel streams:
 *   xp[tile_off[i] + (t*Dp + k)*32 + c] = x[t, order[32i+c], k]   (0 beyond the curve / padding)
 * x is addressed as x[t*stride_t + n*stride_n + k]; (T,B,D) time-major has
 * stride_t = B*D, stride_n = D (the X_tensor of utils/training_utils.py:454-461).
 * If curve_off != NULL the batch is RAGGED (no padding, the layout of the
 * vlen rows in database.h5 and of list_data, utils/training_utils.py:127-191):
 * curve n starts at float offset curve_off[n] and x[curve_off[n] + t*stride_t + k]
 * is read (stride_t = D for concatenated (sum(len), D) rows; stride_n ignored).
 * order/len_sorted/tile_T/tile_off are DEVICE copies of the plan.
 */
int snn_pack_x(const snn_cfg *cfg, const float *x, int64_t stride_t, int64_t stride_n,
               const int64_t *curve_off, const int32_t *order, const int32_t *len_sorted, const int32_t *tile_T,
               const int64_t *tile_off, int32_t n_tiles, int64_t xp_floats, float *xp, void *stream);

/*
 * snn_pack_x for a source whose time steps are not equally spaced -- torch's PackedSequence.data, what
 * get_data_batch hands the model (utils/training_utils.py:479; torch.nn.utils.rnn.pack_padded_sequence): step t
 * holds batch_sizes[t] rows, the rows of the batch columns that are still running.  Row t of curve n is read at
 *   x[step_off[t] + (curve_off ? curve_off[n] : n * stride_n) + k]
 * with step_off (max_T, DEVICE) = D * sum(batch_sizes[:t]) and curve n in column n (stride_n = D) or, for a batch
 * packed with enforce_sorted=False, in column unsorted_indices[n] (curve_off[n] = D * unsorted_indices[n]).
 * Replaces pad_packed_sequence (vanilla_rnn.py:75-76) + snn_pack_x: no padded copy is made.
 */
int snn_pack_x_steps(const snn_cfg *cfg, const float *x, const int64_t *step_off, int64_t stride_n,
                     const int64_t *curve_off, const int32_t *order, const int32_t *len_sorted,
                     const int32_t *tile_T, const int64_t *tile_off, int32_t n_tiles, int64_t xp_floats,
                     float *xp, void *stream);

/*
 * Host-only: the row-building half of the on-the-fly (broker) featuriser (validation/validate_onthefly.py:65-120 ->
 * utils/data_utils.compute_delta_time :364-390, data/make_dataset.pivot_dataframe_single_from_df :800-938).
 * snn_featurize_plan: observations (curve index, MJD) -> `order` (sorted by curve, then MJD, stable), and for the j-th
 * observation in that order its output row `row_of[j]` and the first time of its 8-hour group `row_mjd_of[j]`
 * (a new group opens at the first observation of a curve, at delta_time == 0, or more than min_dt after the running
 * group's first time); rows_per_curve[n_curves]; returns the number of rows (< 0: error).
 * snn_featurize_fill: per (row, filter) flux and error of the observation with the lowest error (earliest on
 * ties); filt < 0 fills nothing; flux_out / err_out / present_out must arrive zeroed.
 */
int64_t snn_featurize_plan(int64_t n, int64_t n_curves, const int64_t *curve, const double *mjd, double min_dt,
                           int64_t *order, int64_t *row_of, double *row_mjd_of, int64_t *rows_per_curve);
int snn_featurize_fill(int64_t n, int64_t n_rows, int32_t F, const int64_t *order, const int64_t *row_of,
                       const double *row_mjd_of, const double *flux, const double *err, const int64_t *filt,
                       double *flux_out, double *err_out, uint8_t *present_out, double *row_mjd);

/*
 * Host-only: gather the elements of a variable-length HDF5 dataset -- the light curves `hf["data"]` of database.h5
 * (written at utils/data_utils.py:691-733, read back one Python object per curve by h5py at utils/training_utils.py:271)
 * -- from the global heap of the memory-mapped file into ONE caller buffer.  file / file_size: the mapped file;
 * desc: n 16-byte element records of the dataset {uint32 length, uint64 heap collection address, uint32 object index};
 * elem_size: bytes per element (4 for float32); out receives sum(length) * elem_size bytes, element after element.
 * Used by supernnova_amd/utils/h5reader.py (the reader for images without h5py).
 */
int snn_h5_vlen_gather(const void *file, int64_t file_size, const void *desc, int64_t n, int32_t elem_size,
                       void *out, int64_t out_bytes);

/*
 * Re-lay ONE canonical flat weight set (P floats) into kernel layout (PK
 * floats).  Used for torch.nn.LSTM / nn.Linear weights of VanillaRNN and
 * VariationalRNN (vanilla_rnn.py:26-34, variational_rnn.py:37-52).
 */
int snn_prepare_weights(const snn_cfg *cfg, const float *flat, float *wk, void *stream);

/*
 * The same for n_samples canonical weight sets at once (set s starts at flat + s * sample_stride; wk receives
 * n_samples x PK floats): the per-sample models SwagModel.sample draws one deepcopy at a time in
 * utils/swag_utils.py:268-335, called per MC sample by validate_rnn.get_batch_predictions_SWAG (:89-112).
 */
int snn_prepare_weights_multi(const snn_cfg *cfg, const float *flat, int64_t sample_stride, int32_t n_samples,
                              float *wk, void *stream);

/*
 * Bayes-by-Backprop draw of S weight sets + their KL; replaces
 * BayesLSTM.forward's sampling loop and BayesLinear.forward
 * (training/bayesian_rnn.py:472-528, 595-616) and compute_KL (:794-815):
 *   sigma = softplus(rho) + 1e-5;  w = mu + eps * sigma;  kl[s] = sum over the
 *   4*L*dirs LSTM tensors and the head weight of  log q(w) - log p_mix(w).
 *   mu  (P)      canonical flat, head bias (deterministic, :587-588) at the tail
 *   rho (P - C)  canonical flat
 *   eps (S, P-C) host-drawn N(0,1) in reference draw order (parity mode), or NULL
 *   mode: 0 = use eps; 1 = mean_field_inference (w = mu, :486-488); 2 = Philox N(0,1)
 *         keyed by (seed, offset + s, parameter index)
 *   wk  (S, PK)  out: kernel-layout weights;  kl (S) out, may be NULL
 *   kl_terms     snn_kl_terms_floats(cfg, S) floats (caller-owned, 8-byte aligned), may be NULL iff kl is NULL: first,
 *                per sample and tensor, (KL term, r1, r2) with r_k the mixture responsibilities snn_bbb_backward needs
 *                -- S*(4*L*dirs+1)*3 floats --, then the scratch of the per-tensor sums (fp64 partial sums of the
 *                workgroups a tensor is dealt over; added in index order: the KL does not depend on scheduling)
 */
int64_t snn_kl_terms_floats(const snn_cfg *cfg, int32_t n_samples);
int snn_bbb_sample(const snn_cfg *cfg, const float *mu, const float *rho, const float *eps,
                   int32_t mode, uint64_t seed, uint64_t offset, int32_t n_samples,
                   const snn_prior *prior, float *wk, float *kl, float *kl_terms, void *stream);

/* Bytes of workspace snn_lstm_forward needs for sequences up to max_T steps. */
int64_t snn_lstm_forward_workspace_bytes(const snn_cfg *cfg, int32_t max_T);

/*
 * The hot path: LSTM stack + pooling + head + softmax for n_samples MC draws of
 * every curve, one persistent launch.  Replaces the S sequential calls of
 *   rnn.forward(X) -> softmax          validation/validate_rnn.py:59-60 (loop :319-331)
 * i.e. torch.nn.LSTM / torch._VF.lstm at vanilla_rnn.py:47, variational_rnn.py:84,244,
 * bayesian_rnn.py:530-553, the pooling of vanilla_rnn.py:60-83 and the output layer.
 *   xp, len_sorted, order, tile_T, tile_off : from snn_plan_build / snn_pack_x (device)
 *   wk : weight sets in kernel layout; set s is at wk + s*wk_sample_stride
 *        (stride 0: all samples share one set -- vanilla / variational)
 *   masks : NULL or dropout masks (see snn_masks)
 *   out_logits / out_probs : either may be NULL; element (s, n, q) is written at
 *        s*out_stride_sample + n*out_stride_curve + q with n the ORIGINAL curve index
 */
int snn_lstm_forward(const snn_cfg *cfg, const float *xp, const int32_t *len_sorted,
                     const int32_t *order, const int32_t *tile_T, const int64_t *tile_off,
                     int32_t n_curves, int32_t n_tiles, int32_t max_T, const float *wk,
                     int64_t wk_sample_stride, int32_t n_samples, const snn_masks *masks,
                     float *out_logits, float *out_probs, int64_t out_stride_sample,
                     int64_t out_stride_curve, void *workspace, int64_t workspace_bytes,
                     void *stream);

/*
 * snn_lstm_forward for callers that cannot allocate.  No counterpart in the reference: it exists for callers of this
 * library that may not call hipMalloc around the launch -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation -- and therefore cannot provide (or size) a workspace.  Same arguments, results
 * and errors as snn_lstm_forward minus workspace / workspace_bytes: the work-queue heads and the layer-output slabs
 * live in a device array of the library, SNN_NOALLOC_SCRATCH_BYTES (64 MiB) on every GPU the library runs on, held
 * until the process ends.  The persistent grid is cut to the number of per-workgroup slabs that fit
 * (dirs * max_T * H * 32 * 4 bytes per tile in flight: 102 workgroups at H=32, max_T=40; results do not depend on the
 * grid); a (cfg, max_T) whose single slab exceeds the scratch returns SNN_ERR_WORKSPACE.  The call itself performs no
 * hipMalloc / hipFree (nor does any other entry point: here there is also no caller allocation to size).  All calls on
 * one GPU share that GPU's array: calls that overlap on one GPU (different streams) race -- serialise them.
 */
#define SNN_NOALLOC_SCRATCH_BYTES (64ll << 20)
#define SNN_NOALLOC_MAX_DEVICES 64
int64_t snn_lstm_forward_noalloc_scratch_bytes(void);
int snn_lstm_forward_noalloc(const snn_cfg *cfg, const float *xp, const int32_t *len_sorted,
                             const int32_t *order, const int32_t *tile_T, const int64_t *tile_off,
                             int32_t n_curves, int32_t n_tiles, int32_t max_T, const float *wk,
                             int64_t wk_sample_stride, int32_t n_samples, const snn_masks *masks,
                             float *out_logits, float *out_probs, int64_t out_stride_sample,
                             int64_t out_stride_curve, void *stream);

/* ---- training ------------------------------------------------------------- */

/* Floats of the activation tape snn_lstm_forward_train writes for n_tiles tiles. */
int64_t snn_train_tape_floats(const snn_cfg *cfg, int32_t n_tiles, int32_t max_T);

/*
 * Training forward of ONE weight set (the reference draws one set per mini-batch,
 * bayesian_rnn.py:492-496): as snn_lstm_forward with n_samples = 1, and additionally records
 * gates, cell and hidden states of every step plus the pooled features on `tape` for the
 * backward.  Replaces `output = rnn(packed_tensor)` in utils/training_utils.py:514.
 * out_logits (N, C) in original curve order.
 */
int snn_lstm_forward_train(const snn_cfg *cfg, const float *xp, const int32_t *len_sorted,
                           const int32_t *order, const int32_t *tile_T, const int64_t *tile_off,
                           int32_t n_curves, int32_t n_tiles, int32_t max_T, const float *wk,
                           const snn_masks *masks, float *out_logits, float *tape, int64_t tape_floats,
                           void *workspace, int64_t workspace_bytes, void *stream);

/*
 * Workspace of snn_lstm_backward.  The backward has three implementations with results equal to the gradient bar
 * (streamed / split / fused; they differ in summation order, ~1e-5 relative), chosen per call from (cfg, n_tiles) and
 * the environment variable SNN_BWD_PATH=stream|split|fused (A/B measurements, read at every call): the size returned
 * here is that of the path THIS call would pick, so query it again whenever n_tiles or the variable may have changed
 * instead of caching it by (cfg, max_T) -- snn_lstm_backward refuses a workspace that is too small for its path with
 * SNN_ERR_WORKSPACE.  snn_last_backward_path() names the path the calling thread's last snn_lstm_backward took.
 */
int64_t snn_lstm_backward_workspace_bytes(const snn_cfg *cfg, int32_t n_tiles, int32_t max_T);
const char *snn_last_backward_path(void);

/*
 * BPTT: gradients of a scalar loss w.r.t. the weight set used by snn_lstm_forward_train, given
 * dlogits = dloss/dlogits (N, C) in original curve order; replaces autograd's traversal of
 * aten::lstm + pooling + output layer in `loss.backward()` (utils/training_utils.py:523).
 * dflat (P) receives (accumulate = 0) or accumulates (1) the gradient in canonical flat order;
 * for VariationalRNN the W_hh gradient is that of weight_hh_l0_raw (variational_rnn.py:237).
 * masks must be the ones of the forward.  Deterministic (fixed summation order).
 */
int snn_lstm_backward(const snn_cfg *cfg, const float *xp, const int32_t *len_sorted, const int32_t *order,
                      const int32_t *tile_T, const int64_t *tile_off, int32_t n_curves, int32_t n_tiles,
                      int32_t max_T, const float *wk, const snn_masks *masks, const float *tape,
                      int64_t tape_floats, const float *dlogits, float *dflat, int32_t accumulate,
                      void *workspace, int64_t workspace_bytes, void *stream);

/*
 * Bayes-by-Backprop backward for one draw: from dw = dCE/dw (P, canonical; head bias at the tail)
 * to dmu (P) and drho (P - C) of  loss = CE + kl_scale * KL  (utils/training_utils.py:517-518 with
 * kl_scale = 1/(num_batches*batch_size)); autograd of bayesian_rnn.py:472-528, 595-616, 794-815.
 * eps/mode/seed/offset and kl_terms must be those of the forward's snn_bbb_sample (sample 0).
 * kl_scale_dev, if not NULL, is a device scalar that overrides kl_scale (avoids a host sync when the
 * scale arrives as an autograd gradient).
 */
int snn_bbb_backward(const snn_cfg *cfg, const float *mu, const float *rho, const float *eps, int32_t mode,
                     uint64_t seed, uint64_t offset, const snn_prior *prior, const float *kl_terms,
                     float kl_scale, const float *kl_scale_dev, const float *dw, float *dmu, float *drho,
                     int32_t accumulate, void *stream);

/*
 * Mean cross-entropy over n curves and dlogits = grad_scale * (softmax - onehot)  [grad_scale = 1/n
 * for the plain mean]; nn.CrossEntropyLoss at utils/training_utils.py:515 and its backward.
 * target int64 (n), loss: device scalar.  The sum is taken in a fixed order (per-workgroup fp64 partials, added by
 * index), so the loss does not depend on scheduling.  `partials`: caller-owned device scratch of
 * snn_ce_loss_partials(n) doubles (0 for n <= 1024, where it may be NULL; 256 otherwise) -- the library keeps no state,
 * so calls on different streams with different scratch may overlap.  A target outside [0, n_classes) -- where
 * nn.CrossEntropyLoss raises, and train_rnn.py:318-323 raises ValueError beforehand -- makes the loss NaN and zeroes
 * that curve's row of dlogits (no out-of-bounds read).
 */
int64_t snn_ce_loss_partials(int32_t n);
int snn_ce_loss_grad(int32_t n, int32_t n_classes, const float *logits, const int64_t *target,
                     float grad_scale, float *loss, float *dlogits, double *partials, void *stream);


/*
 * Fused Adam step over one flat buffer, torch.optim.Adam semantics (betas, eps, L2 weight_decay
 * added to the gradient; utils/training_utils.py:365-369, :524).  grads are multiplied by
 * grad_scale first (1/world_size after a data-parallel sum).  step counts from 1.
 */
int snn_adam_step(int64_t n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq,
                  int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                  float grad_scale, void *stream);

/* torch.optim.SGD (momentum, L2 weight decay, dampening 0, no Nesterov) over a flat buffer: the optimizer of the
 * reference's train_swag_finetune (training/train_rnn.py:541-546).  `step` counts from 1 (step 1 initialises
 * momentum_buf with the gradient, as torch does); momentum_buf may be NULL when momentum is 0. */
int snn_sgd_step(int64_t n, float *params, const float *grads, float *momentum_buf, int64_t step, float lr,
                 float momentum, float weight_decay, float grad_scale, void *stream);

/*
 * One training step up to the flat gradient, as ONE call: the reference's
 *     rnn.train(); optimizer.zero_grad(); output = rnn(packed_tensor); loss = criterion(output.squeeze(), target_tensor)
 *     [+ rnn.kl / (num_batches * batch_size)]; loss.backward()                    utils/training_utils.py:508-523
 * i.e. snn_pack_x_steps (or snn_pack_x), snn_bbb_sample (BayesianRNN) or snn_prepare_weights, snn_lstm_forward_train,
 * snn_ce_loss_grad, snn_lstm_backward and snn_bbb_backward enqueued back to back on `stream` -- the same launches in the
 * same order as the separate entry points (results are identical), without a host round trip per launch: at the
 * reference's batch size (128, conf.py:517) the step is a few hundred microseconds of GPU work and the host side of
 * twelve separate calls took longer than that.  The optimizer step stays separate (snn_adam_step / snn_sgd_step): under
 * data parallelism the all-reduce of `grad` sits between the two.  All buffers are the caller's:
 *   x, step_off, stride_n, curve_off   the batch as snn_pack_x_steps takes it (step_off != NULL), or as snn_pack_x takes it
 *                                      (step_off == NULL: stride_t is used)
 *   order .. tile_off, n_curves, n_tiles, max_T, xp_floats     device copy of the snn_plan_build plan
 *   target (n_curves) int64
 *   params   canonical flat parameters: P floats (VanillaRNN / VariationalRNN), or mu (P) when rho != NULL
 *   rho      NULL, or rho (P - C): Bayes-by-Backprop draw with eps / bbb_mode / seed / offset / prior as snn_bbb_sample
 *   masks    NULL or the dropout masks of forward and backward
 *   ce_grad_scale   dlogits = ce_grad_scale * (softmax - onehot): 1 / n_curves, or 1 / global batch under data parallelism
 *   kl_scale        weight of the KL gradient (1 / (num_batches * batch_size), divided by the world size under DP)
 *   loss_kl_scale   weight of the KL term in the reported loss (1 / (num_batches * batch_size))
 *   out: xp (xp_floats), wk (PK), kl (1) + kl_terms (snn_kl_terms_floats(cfg, 1)) [Bayesian], logits (n_curves, C), tape
 *        (snn_train_tape_floats), dlogits (n_curves, C), loss (2 floats: mean CE of this batch, and for BayesianRNN
 *        CE + loss_kl_scale * KL), ce_partials (snn_ce_loss_partials doubles, may be NULL for n <= 1024), dw (P: the
 *        gradient w.r.t. the weight set -- for VanillaRNN / VariationalRNN THE gradient), grad (BayesianRNN only:
 *        P + P - C floats, dmu then drho), fwd_ws / bwd_ws as the separate entry points size them.
 */
typedef struct snn_train_io {
    const float *x; const int64_t *step_off; int64_t stride_t, stride_n; const int64_t *curve_off;
    const int32_t *order, *len_sorted, *tile_T; const int64_t *tile_off;
    int32_t n_curves, n_tiles, max_T, bbb_mode;
    int64_t xp_floats;
    const int64_t *target;
    const float *params, *rho, *eps;
    uint64_t seed, offset;
    const snn_prior *prior;
    const snn_masks *masks;
    float ce_grad_scale, kl_scale, loss_kl_scale, reserved;
    float *xp, *wk, *kl, *kl_terms, *logits, *tape;
    int64_t tape_floats;
    float *dlogits, *loss;
    double *ce_partials;
    float *dw, *grad;
    void *fwd_ws; int64_t fwd_ws_bytes;
    void *bwd_ws; int64_t bwd_ws_bytes;
} snn_train_io;
int snn_train_step_grad(const snn_cfg *cfg, const snn_train_io *io, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SNN_HIP_H */
