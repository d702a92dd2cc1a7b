"""Hot-path subset of the reference's utils/training_utils.py, same names and contracts:
  get_model        :320-351   (vanilla | variational | bayesian)
  get_optimizer    :354-371
  get_data_batch   :374-481   host batching -> (packed, X_tensor, target, idxs_rev_sort)
  eval_step        :528-544
  get_evaluation_metrics :609-707, get_loss_string :710-736, save_training_results :739-769, StopOnPlateau :847-911
  unnormalize_arr  :89-124    (needed by the 'sin' OOD transform)
Data loading (HDF5) and plotting stay out of scope.
"""
import json
import os

import numpy as np
import torch
from torch import nn

from ..modules import VanillaRNN, VariationalRNN, BayesianRNN

_MODELS = {"vanilla": VanillaRNN, "variational": VariationalRNN, "bayesian": BayesianRNN}


def get_model(settings, input_size):
    """Create the RNN model `settings.model` names (training_utils.py:320-351)."""
    try:
        cls = VanillaRNN if settings.model == "swag" else _MODELS[settings.model]
    except KeyError:
        raise ValueError(f"supernnova_amd implements models {sorted(_MODELS) + ['swag']}, got {settings.model!r}") from None
    rnn = cls(input_size, settings)
    if settings.model == "swag":          # SWAG statistics around a VanillaRNN (training_utils.py:337-345)
        from .swag_utils import SwagModel
        rnn = SwagModel(rnn)
    if not getattr(settings, "no_dump", True):
        print(rnn)
    return rnn


def get_optimizer(settings, model):
    """Adam(lr, weight_decay) as training_utils.py:365-369 -- the fused HIP Adam over the flat buffer."""
    from ..optim import FusedAdam
    return FusedAdam(model, lr=settings.learning_rate, weight_decay=settings.weight_decay)


def train_step(settings, rnn, packed_tensor, target_tensor, criterion, optimizer, batch_size, num_batches):
    """Full training step, the reference's sequence (training_utils.py:484-525): train mode, zero grads,
    forward, CE (+ KL/(num_batches*batch_size) for BayesianRNN), backward, optimizer step.
    Under torch.distributed each rank passes its shard of the mini-batch and batch_size = the GLOBAL batch
    size; the optimizer all-reduces (averages) the flat gradient."""
    rnn.train()
    optimizer.zero_grad()
    output = rnn(packed_tensor)
    loss = criterion(output.squeeze(), target_tensor)
    if isinstance(rnn, BayesianRNN):
        loss = loss + rnn.kl / (num_batches * batch_size)
    loss.backward()
    optimizer.step()
    return loss


def train_step_fused(settings, rnn, packed_tensor, target_tensor, optimizer, batch_size, num_batches):
    """Same step without autograd: CE and its gradient in snn_ce_loss_grad, BPTT, BBB backward with the
    KL scale known on the host, one Adam launch.  Returns the loss as a device scalar."""
    from .. import engine as E
    rnn.train()
    return rnn.fused_train_step(packed_tensor, target_tensor, optimizer, 1.0 / (num_batches * batch_size))


def unnormalize_arr(arr, settings):
    """Inverse of the log-standardisation (training_utils.py:89-124)."""
    if settings.norm == "none":
        return arr
    lo, mean, std = settings.arr_norm[:, 0], settings.arr_norm[:, 1], settings.arr_norm[:, 2]
    if "cosmo" in settings.norm:
        col = settings.idx_features_to_normalize[-1]
        arr[:, col] = np.exp(arr[:, col] * std[-1] + mean[-1]) + lo[-1] - 1e-5
    else:
        cols = settings.idx_features_to_normalize
        arr[:, cols] = np.exp(arr[:, cols] * std + mean) + lo - 1e-5
    return arr


def _ood(X, kind, settings):
    """Out-of-distribution variants of one light curve (training_utils.py:397-433); consumes
    numpy's global RNG in the reference's order."""
    X = X.copy()
    if kind == "reverse":
        return np.ascontiguousarray(X[::-1])
    if kind == "shuffle":
        return X[np.random.permutation(X.shape[0])]
    flux, err = X[:, settings.idx_flux], X[:, settings.idx_fluxerr]
    if kind == "sin":
        dt = unnormalize_arr(X.copy(), settings)[:, settings.idx_delta_time]
        mjd = np.cumsum(dt, axis=0)
        X[:, settings.idx_flux] = np.sin(mjd * 2 * np.pi / 30) * np.max(flux, axis=0, keepdims=True)
        X[:, settings.idx_fluxerr] = np.random.uniform(err.min(), err.max(), size=err.shape)
    elif kind == "random":
        X[:, settings.idx_flux] = np.random.uniform(flux.min(), flux.max(), size=flux.shape)
        X[:, settings.idx_fluxerr] = np.random.uniform(err.min(), err.max(), size=err.shape)
    return X


def get_data_batch(list_data, idxs, settings, max_lengths=None, OOD=None):
    """Build one batch: optional OOD transform / truncation, sort by length (descending), zero-pad
    to (T, B, D), pack.  Returns (packed_tensor, X_tensor, target_tensor, idxs_rev_sort) exactly as
    training_utils.py:374-481 (tensors on the GPU when settings.use_cuda)."""
    rows, lens = [], []
    for pos, i in enumerate(idxs):
        X, target, *_ = list_data[i]
        if OOD is not None:
            X = _ood(X, OOD, settings)
        if max_lengths is not None:
            assert settings.random_length is False
            assert settings.random_redshift is False
            X = X[: max_lengths[pos]]
        if settings.random_length:
            X = X[: np.random.randint(3, X.shape[0] + 1)]
        if settings.redshift == "zspe" and getattr(settings, "random_redshift", False):
            if np.random.binomial(1, 0.5) == 0:
                X[:, settings.idx_specz] = -1
        rows.append((X, target))
        lens.append(X.shape[0])
    idx_sort = np.argsort(lens)[::-1]
    idxs_rev_sort = np.argsort(idx_sort)
    T, D = lens[idx_sort[0]], rows[0][0].shape[1]
    X_np = np.zeros((T, len(rows), D), np.float32)
    targets, lengths = [], []
    for j, src in enumerate(idx_sort):
        X, target = rows[src]
        X_np[: X.shape[0], j] = X
        targets.append(target)
        lengths.append(lens[src])
    X_tensor = torch.from_numpy(X_np)
    target_tensor = torch.LongTensor(targets)
    if settings.use_cuda:
        X_tensor, target_tensor = X_tensor.cuda(), target_tensor.cuda()
    packed_tensor = nn.utils.rnn.pack_padded_sequence(X_tensor, lengths)
    return packed_tensor, X_tensor, target_tensor, idxs_rev_sort


def eval_step(rnn, packed_tensor):
    """Forward only, eval mode (training_utils.py:528-544)."""
    rnn.eval()
    return rnn(packed_tensor)


def eval_step_swag(model, scale, cov, packed_tensor):
    """One SWAG draw evaluated on a batch (training_utils.py:547-577): the draw goes through the per-sample-weights
    kernel (SwagModel.forward_mc with one sample) instead of a deep-copied module; returns log-probabilities' logits
    equivalent for the softmax that follows -- log of the class probabilities."""
    model.eval()
    return torch.log(model.forward_mc(packed_tensor, 1, scale=scale, cov=cov)[:, 0])


def get_evaluation_metrics(settings, list_data, model, sample_size=None, swag_sampling=False):
    """AUC / accuracy / log-loss (/ mean KL) of `model` over `list_data`, batch by batch in eval mode -- one
    stochastic forward per batch for the variational and Bayesian models (and one SWAG draw per batch with
    `swag_sampling`), as the reference (training_utils.py:609-707).  Consumes numpy's global RNG for the batch
    subsample exactly where the reference does."""
    from sklearn import metrics
    n = len(list_data)
    batches = np.array_split(np.arange(n), n // min(n // 2, settings.batch_size))
    if sample_size:
        pick = np.random.permutation(len(batches))[: sample_size // min(sample_size // 2, settings.batch_size)]
        batches = [batches[i] for i in pick]
    preds, targets, kls = [], [], []
    keep = settings.random_length
    for idxs in batches:
        settings.random_length = False                     # full light curves for evaluation (:641-646)
        try:
            packed, _, target, rev = get_data_batch(list_data, idxs, settings)
        finally:
            settings.random_length = keep
        with torch.no_grad():
            if swag_sampling:
                out = eval_step_swag(model, settings.swag_scale, not getattr(settings, "swag_no_lr_cov", False), packed)
            else:
                out = eval_step(model, packed)
        if isinstance(model, BayesianRNN):
            kls.append(float(model.kl.detach()))
        preds.append(torch.softmax(out, dim=1).cpu().numpy()[rev])
        targets.append(target.cpu().numpy()[rev])
    preds, targets = np.concatenate(preds), np.concatenate(targets)
    if not np.isfinite(preds).all():
        raise FloatingPointError("non-finite class probabilities during evaluation")
    auc = None
    if settings.nb_classes == 2 and len(set(targets.tolist())) > 1:
        auc = metrics.roc_auc_score(targets, preds[:, 1])
    one_hot = np.zeros((len(targets), settings.nb_classes))
    one_hot[np.arange(len(targets)), targets] = 1
    d = {"AUC": auc, "Acc": metrics.accuracy_score(targets, preds.argmax(1)), "loss": metrics.log_loss(one_hot, preds)}
    if kls:
        d["KL"] = float(np.mean(kls))
    return d


def get_loss_string(d_losses_train, d_losses_val):
    """Progress string `AUC/Acc/loss[/KL] [T]: ... [V]: ...` (training_utils.py:710-736)."""
    fmt = lambda d: "/".join("NA" if (v is None or k == "epoch") else f"{v:.3g}" for k, v in d.items())
    return "/".join(d_losses_train.keys()) + " [T]: " + fmt(d_losses_train) + " [V]: " + fmt(d_losses_val)


def save_training_results(settings, d_monitor, training_time):
    """Best validation metrics + wall time into rnn_dir/training_log.json, keyed by the model name
    (training_utils.py:739-769)."""
    res = {"training_time": training_time}
    for key in ("AUC", "Acc"):
        if key == "AUC" and settings.nb_classes > 2:
            res[key] = -1
        else:
            res[key] = max(d_monitor[key]) if None not in d_monitor[key] else None
    res["loss"] = min(d_monitor["loss"])
    res["data_types_training"] = np.array2string(np.asarray(getattr(settings, "data_types_training", [])))
    path = os.path.join(str(settings.rnn_dir), "training_log.json")
    try:
        with open(path) as f:
            log = json.load(f)
    except (OSError, ValueError):
        log = {}
    log[settings.pytorch_model_name] = res
    with open(path, "w") as f:
        json.dump(log, f)


class StopOnPlateau:
    """Plateau detector on a metric (training_utils.py:847-911): once more than 10 values are in, a standard
    deviation of the last 10 below 5e-4 is a plateau; with `reduce_lr_on_plateau` the optimizer's learning rate is
    divided by 10 and training stops at the `max_learning_rate_reduction`-th reduction, without it training
    stops at once.  `step` returns True to stop, False inside the window, None before it is full -- as the
    reference, whose callers test `is True`."""

    def __init__(self, patience=10, reduce_lr_on_plateau=False, max_learning_rate_reduction=3):
        self.patience = patience
        self.last_epoch = -1
        self.list_metric = []
        self.reduce_lr_on_plateau = reduce_lr_on_plateau
        self.max_learning_rate_reduction = max_learning_rate_reduction
        self.learning_rate_reduction = 0

    def step(self, metric_value, optimizer=None, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        self.list_metric.append(metric_value)
        if len(self.list_metric) <= 10:
            return None
        self.list_metric = self.list_metric[-10:]
        if np.std(self.list_metric) >= 0.0005:
            return False
        if optimizer is None or not self.reduce_lr_on_plateau:
            return True
        for group in optimizer.param_groups:
            group["lr"] = group["lr"] / 10.0
        self.learning_rate_reduction += 1
        return True if self.learning_rate_reduction == self.max_learning_rate_reduction else None
