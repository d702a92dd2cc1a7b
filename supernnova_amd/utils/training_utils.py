"""Hot-path subset of the reference's utils/training_utils.py, same names and contracts:
  get_model        :320-351   (vanilla | variational | bayesian)
  get_optimizer    :354-371
  get_data_batch   :374-481   host batching -> (packed, X_tensor, target, idxs_rev_sort)
  eval_step        :528-544
  get_evaluation_metrics :609-707, get_loss_string :710-736, save_training_results :739-769, StopOnPlateau :847-911
  unnormalize_arr  :89-124    (needed by the 'sin' OOD transform)
  normalize_arr    :23-86, fill_data_list :127-191, load_HDF5 :194-317 (vectorised, file access through
                   utils/data_utils.open_database)
Plotting stays out of scope.
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
    from ..distributed import world_size
    world = world_size(getattr(optimizer, "process_group", None))
    if world > 1 and int(target_tensor.size(0)) * world != int(batch_size):
        # the optimizer averages the ranks' gradients, each a mean over that rank's own curves: with unequal shards
        # that is not the gradient of the global mini-batch.  train_step_fused weights curves exactly for any split.
        raise ValueError(f"train_step under torch.distributed needs equal shards (this rank: {int(target_tensor.size(0))} "
                         f"curves, world {world}, global batch {int(batch_size)}): use train_step_fused, or pad the shards")
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
    if not rnn.training:
        rnn.train()
    return rnn.fused_train_step(packed_tensor, target_tensor, optimizer, 1.0 / (num_batches * batch_size),
                                global_batch=batch_size)


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


def normalize_arr(arr, settings):
    """Clip + log-standardise the normalised columns in place (training_utils.py:23-86): float64 arithmetic on the
    float32 values, stored back as float32.  norm "none" | "global" | "perfilter" (same formula, different arr_norm)
    | "cosmo" | "cosmo_quantile" (per-light-curve flux scale)."""
    if settings.norm == "none":
        return arr
    lo, mean, std = settings.arr_norm[:, 0], settings.arr_norm[:, 1], settings.arr_norm[:, 2]
    cols = settings.idx_features_to_normalize
    a = np.clip(arr[:, cols], lo, np.inf)
    if "cosmo" not in settings.norm:
        a = (np.log(a - lo + 1e-5) - mean) / std
    else:
        scale = a[:, :-1].max() if settings.norm == "cosmo" else np.quantile(a[:, :-1], 0.99)
        a[:, :-1] = a[:, :-1] / scale
        a[:, -1] = (np.log(a[:, -1] - lo[-1] + 1e-5) - mean[-1]) / std[-1]
    arr[:, cols] = a
    return arr


class RaggedList(list):
    """list_data as the reference's load_HDF5 returns it -- tuples (X_normed, target, SNID[, X_all, X_ori]) -- whose
    X_normed entries are VIEWS of one concatenated (sum(len), D) float32 array: `.rows`, `.lengths`, `.curve_off`
    (float offsets) feed snn_pack_x's ragged form directly (validate_rnn.ragged_to_device), no per-curve copy."""
    rows = None
    lengths = None
    curve_off = None


def fill_data_list(idxs, arr_data, arr_target, arr_SNID, settings, n_features, desc="", test=False):
    """The data tuples of training_utils.py:127-191 for the light curves `idxs`, built for all curves at once: one
    concatenated (rows, n_features) array is clipped, checked (normalise -> unnormalise round trip within 0.1, :168-172)
    and normalised in one pass instead of a Python loop over light curves; element-wise arithmetic is the
    reference's (float64 on float32 values, stored as float32), so every tuple equals the reference's bit for bit.
    arr_data[i]: flat float32 array of light curve i (rows x n_features)."""
    idxs = np.asarray(idxs, dtype=np.int64)
    out = RaggedList()
    if idxs.size == 0:
        return out
    n_features = int(n_features)
    flat = [np.asarray(arr_data[i], dtype=np.float32) for i in idxs]
    lengths = np.array([a.size // n_features for a in flat], np.int64)
    X_all = np.concatenate(flat).reshape(-1, n_features)
    cols, feat = settings.idx_features_to_normalize, settings.idx_features
    X_ori = X_all[:, feat].copy()
    X_all[:, cols] = np.clip(X_all[:, cols], settings.arr_norm[:, 0], np.inf)
    bounds = np.concatenate([[0], np.cumsum(lengths)])
    if "cosmo" in settings.norm:          # per-light-curve flux scale: curve by curve
        X_normed = X_all.copy()
        for a, b in zip(bounds[:-1], bounds[1:]):
            X_normed[a:b] = normalize_arr(X_normed[a:b].copy(), settings)
    else:
        X_normed = normalize_arr(X_all.copy(), settings)
        back = unnormalize_arr(X_normed.copy(), settings)
        if not np.all(np.isclose(np.ravel(X_all), np.ravel(back), atol=1e-1)):
            raise AssertionError("normalisation does not round-trip within 0.1 (training_utils.py:168-172)")
    X_normed = np.ascontiguousarray(X_normed[:, feat])
    for k, i in enumerate(idxs):
        a, b = bounds[k], bounds[k + 1]
        tup = (X_normed[a:b], int(arr_target[i]), str(arr_SNID[i]))
        out.append(tup + (X_all[a:b], X_ori[a:b]) if test else tup)
    out.rows, out.lengths = X_normed, lengths.astype(np.int32)
    out.curve_off = bounds[:-1] * X_normed.shape[1]
    return out


def load_HDF5(settings, test=False):
    """Load `settings.processed_dir/database.h5` (training_utils.py:194-317): the test list, or (train list, val
    list) with the reference's shuffles of numpy's global RNG and `data_fraction` cut.  The file is opened through
    data_utils.open_database (h5py when installed, the package's native reader otherwise)."""
    from . import data_utils as du
    with du.open_database(f"{settings.processed_dir}/database.h5") as hf:
        config_name = f"{settings.source_data}_{settings.nb_classes}classes"
        split_key, target_key = f"dataset_{config_name}", f"target_{settings.nb_classes}classes"
        if test:
            try:        # "failsafe in case we have different classes in dataset/model" (:218-223)
                idxs_test = np.where(hf[split_key][:] == 2)[0]
            except Exception:
                idxs_test = np.where(hf["dataset_photometry_2classes"][:] != 100)[0]
        else:
            split = hf[split_key][:]
            idxs_train, idxs_val, idxs_test = (np.where(split == k)[0] for k in (0, 1, 2))
            np.random.shuffle(idxs_train)
            np.random.shuffle(idxs_val)
            np.random.shuffle(idxs_test)
            idxs_train = idxs_train[: int(settings.data_fraction * len(idxs_train))]
        n_features = int(hf["data"].attrs["n_features"])
        feats_db = list(du._as_str(hf["features"][:]))
        feats = settings.training_features
        if not set(feats) <= set(feats_db) and any("DES-" in k for k in feats_db):
            feats_db = [k.replace("DES-", "") for k in feats_db]
        assert set(feats) <= set(feats_db)
        if not set(settings.training_features_to_normalize) <= set(feats):
            raise ValueError("Database does not contain all training features (check --list_filters): "
                             f"database {feats_db}, to normalize {settings.training_features_to_normalize}")
        try:
            settings.data_types_training = du._as_str(hf["data_types_training"][:])
        except Exception:
            pass
        arr_data = hf["data"][:]
        if test:
            try:
                arr_target = hf[target_key][:]
            except Exception:
                arr_target = hf["target_2classes"][:]
        else:
            arr_target = hf[target_key][:]
        arr_SNID = du._as_str(hf["SNID"][:])
        if test:
            return fill_data_list(idxs_test, arr_data, arr_target, arr_SNID, settings, n_features, test=True)
        return (fill_data_list(idxs_train, arr_data, arr_target, arr_SNID, settings, n_features),
                fill_data_list(idxs_val, arr_data, arr_target, arr_SNID, settings, n_features))


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
