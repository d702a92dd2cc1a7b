// snn_train_step.hip -- host-only: one training step up to the flat gradient as ONE call of the C ABI
// (include/snn_hip.h: snn_train_step_grad; utils/training_utils.py:508-523).  Composition of the separate entry points,
// enqueued back to back on one stream: same launches, same order, same results -- what goes away is the host round
// trip per launch, which at the reference's batch size of 128 curves cost more than the GPU work.
#include <stdint.h>

#include "../../include/snn_hip.h"
#include "snn_host.h"

int snn_ce_loss_grad_total(int32_t n, int32_t n_classes, const float* logits, const int64_t* target, float grad_scale,
                           float* loss, float* dlogits, double* partials, const float* kl, float kl_scale, void* stream);   // snn_train.hip

extern "C" int snn_train_step_grad(const snn_cfg* cfg, const snn_train_io* io, void* stream)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if (!io) { snn_set_error("snn_train_step_grad: io is NULL"); return SNN_ERR_ARG; }
    const bool bayes = io->rho != nullptr;
    if (!io->x || !io->target || !io->params || !io->xp || !io->wk || !io->logits || !io->tape || !io->dlogits || !io->loss ||
        !io->dw || (bayes && !io->grad) || !io->fwd_ws || !io->bwd_ws || (bayes && (!io->kl || !io->kl_terms || !io->prior))) {
        snn_set_error("snn_train_step_grad: null pointer");
        return SNN_ERR_ARG;
    }
    if (io->step_off)
        rc = snn_pack_x_steps(cfg, io->x, io->step_off, io->stride_n, io->curve_off, io->order, io->len_sorted, io->tile_T,
                              io->tile_off, io->n_tiles, io->xp_floats, io->xp, stream);
    else
        rc = snn_pack_x(cfg, io->x, io->stride_t, io->stride_n, io->curve_off, io->order, io->len_sorted, io->tile_T,
                        io->tile_off, io->n_tiles, io->xp_floats, io->xp, stream);
    if (rc) return rc;
    if (bayes)
        rc = snn_bbb_sample(cfg, io->params, io->rho, io->eps, io->bbb_mode, io->seed, io->offset, 1, io->prior, io->wk, io->kl,
                            io->kl_terms, stream);
    else
        rc = snn_prepare_weights(cfg, io->params, io->wk, stream);
    if (rc) return rc;
    rc = snn_lstm_forward_train(cfg, io->xp, io->len_sorted, io->order, io->tile_T, io->tile_off, io->n_curves, io->n_tiles,
                                io->max_T, io->wk, io->masks, io->logits, io->tape, io->tape_floats, io->fwd_ws,
                                io->fwd_ws_bytes, stream);
    if (rc) return rc;
    rc = snn_ce_loss_grad_total(io->n_curves, cfg->nb_classes, io->logits, io->target, io->ce_grad_scale, io->loss, io->dlogits,
                                io->ce_partials, bayes ? io->kl : nullptr, io->loss_kl_scale, stream);
    if (rc) return rc;
    rc = snn_lstm_backward(cfg, io->xp, io->len_sorted, io->order, io->tile_T, io->tile_off, io->n_curves, io->n_tiles,
                           io->max_T, io->wk, io->masks, io->tape, io->tape_floats, io->dlogits, io->dw, 0, io->bwd_ws,
                           io->bwd_ws_bytes, stream);
    if (rc) return rc;
    if (bayes)
        rc = snn_bbb_backward(cfg, io->params, io->rho, io->eps, io->bbb_mode, io->seed, io->offset, io->prior, io->kl_terms,
                              io->kl_scale, nullptr, io->dw, io->grad, io->grad + y.P, 0, stream);
    return rc;
}
