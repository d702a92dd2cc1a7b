"""Hot-path subset of the reference's utils/training_utils.py, same names and contracts:
  get_model        :320-351   (vanilla | variational | bayesian)
  get_optimizer    :354-371
  get_data_batch   :374-481   host batching -> (packed, X_tensor, target, idxs_rev_sort)
  eval_step        :528-544
  unnormalize_arr  :89-124    (needed by the 'sin' OOD transform)
Data loading (HDF5), plotting and metrics stay out of scope.
"""
import numpy as np
import torch
from torch import nn

from ..modules import VanillaRNN, VariationalRNN, BayesianRNN

_MODELS = {"vanilla": VanillaRNN, "variational": VariationalRNN, "bayesian": BayesianRNN}


def get_model(settings, input_size):
    """Create the RNN model `settings.model` names (training_utils.py:320-351)."""
    try:
        cls = _MODELS[settings.model]
    except KeyError:
        raise ValueError(f"supernnova_amd implements models {sorted(_MODELS)}, got {settings.model!r}") from None
    rnn = cls(input_size, settings)
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
