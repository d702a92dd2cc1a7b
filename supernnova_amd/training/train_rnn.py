"""Epoch loops of the reference's training/train_rnn.py around the HIP train step, same names and policies:
  get_lr              :13-95    learning-rate range test (exponential sweep, smoothed loss, stop at 4x the best)
  train_cyclic        :98-222   1-cycle policy
  save_normalizations :225-242  data_norm.json
  train / train_      :245-372  decay-on-plateau policy, best-validation-loss checkpoint, training_log.json
  train_swag          :375-516  SWAG statistics from swag_start_epoch on
  train_swag_finetune :516-663  SGD(momentum 0.9, weight decay 1e-4) from a pre-trained state dict, SWAG statistics from epoch 0
The loops are host glue; every step runs tu.train_step (forward tape, BPTT, BBB backward, fused Adam / SGD on the GPU).
Like the reference they read their data with tu.load_HDF5(settings, test=False) (utils/data_utils.open_database: h5py
when importable, the package's own reader otherwise); a caller that already holds the two lists load_HDF5 returns --
`(X_normed[L, D] float32, target, ...)` tuples -- may pass them instead.  Loss plots are not drawn."""
import json
import os
from time import time

import numpy as np
import torch
from torch import nn

from ..utils import training_utils as tu


def _data(settings, list_data_train, list_data_val):
    if list_data_train is None or list_data_val is None:
        return tu.load_HDF5(settings, test=False)                    # train_rnn.py:271, :387, :527
    return list_data_train, list_data_val


def _epoch_batches(n, batch_size):
    """Shuffled index batches of one epoch (train_rnn.py:302-305); consumes numpy's global RNG like the reference."""
    batches = np.array_split(np.arange(n), n // min(n // 2, batch_size))
    np.random.shuffle(batches)
    return batches


def _setup(settings, input_size):
    rnn = tu.get_model(settings, input_size)
    if settings.use_cuda:
        rnn.cuda()
    criterion = nn.CrossEntropyLoss()
    return rnn, criterion, tu.get_optimizer(settings, rnn)


def _monitors(settings):
    keys = ["loss", "AUC", "Acc", "epoch"] + (["KL"] if settings.model == "bayesian" else [])
    return {k: [] for k in keys}, {k: [] for k in keys}


def _run_epoch(settings, rnn, criterion, optimizer, list_data_train):
    batches = _epoch_batches(len(list_data_train), settings.batch_size)
    for idxs in batches:
        packed, _, target, _ = tu.get_data_batch(list_data_train, idxs, settings)
        if settings.nb_classes <= int(target.max()):
            raise ValueError("a target class index is >= nb_classes (all sntypes must be defined when the database "
                             "is built)")
        tu.train_step(settings, rnn, packed, target, criterion, optimizer, target.size(0), len(batches))


def _monitor(settings, rnn, list_data_train, list_data_val, d_train, d_val, epoch, best_loss):
    """Evaluate, log, and save the state dict when the validation log-loss improves (train_rnn.py:336-366)."""
    m_train = tu.get_evaluation_metrics(settings, list_data_train, rnn, sample_size=len(list_data_val))
    m_val = tu.get_evaluation_metrics(settings, list_data_val, rnn, sample_size=None)
    for k in m_train:
        d_train[k].append(m_train[k])
        d_val[k].append(m_val[k])
    d_train["epoch"].append(epoch + 1)
    d_val["epoch"].append(epoch + 1)
    if m_val["loss"] < best_loss:
        best_loss = m_val["loss"]
        torch.save(rnn.state_dict(), os.path.join(str(settings.rnn_dir), f"{settings.pytorch_model_name}.pt"))
    return m_train, m_val, best_loss


def save_normalizations(settings):
    """rnn_dir/data_norm.json: min / mean / std of every normalised feature (train_rnn.py:225-242)."""
    norm = {f: {w: float(settings.arr_norm[i, j]) for j, w in enumerate(("min", "mean", "std"))}
            for i, f in enumerate(settings.training_features_to_normalize)}
    with open(os.path.join(str(settings.rnn_dir), "data_norm.json"), "w") as fh:
        json.dump(norm, fh, indent=4, sort_keys=True)


def get_lr(settings, list_data_train=None, list_data_val=None):
    """Learning-rate range test (train_rnn.py:13-95): one pass over shuffled batches with the rate swept from 1e-8
    to 10; returns the rate at the minimum of the bias-corrected exponentially smoothed loss."""
    list_data_train, _ = _data(settings, list_data_train, list_data_val)
    batches = _epoch_batches(len(list_data_train), settings.batch_size)
    lr0, lr1, beta = 1e-8, 10.0, 0.98
    mult = (lr1 / lr0) ** (1 / len(batches))
    settings.learning_rate = lr0
    rnn, criterion, optimizer = _setup(settings, list_data_train[0][0].shape[1])
    lr, avg, best, losses, lrs = lr0, 0.0, 0.0, [], []
    for k, idxs in enumerate(batches, start=1):
        packed, _, target, _ = tu.get_data_batch(list_data_train, idxs, settings)
        loss = float(tu.train_step(settings, rnn, packed, target, criterion, optimizer, target.size(0), len(batches)).detach())
        avg = beta * avg + (1 - beta) * loss
        smoothed = avg / (1 - beta ** k)
        if k > 1 and smoothed > 4 * best:
            break
        if k == 1 or smoothed < best:
            best = smoothed
        losses.append(smoothed)
        lrs.append(lr)
        lr *= mult
        for group in optimizer.param_groups:
            group["lr"] = lr
    return lrs[int(np.argmin(losses))]


def one_cycle_sched(epoch, minv, maxv, phases):
    """Piecewise-linear 1-cycle rate: up to maxv at phases[0], back to minv at phases[1], down to minv/100 at
    phases[2] (train_rnn.py:112-122)."""
    if epoch <= phases[0]:
        return minv + (maxv - minv) / phases[0] * epoch
    if epoch <= phases[1]:
        return maxv + (minv - maxv) / (phases[1] - phases[0]) * (epoch - phases[0])
    return minv + (minv / 100 - minv) / (phases[2] - phases[1]) * (epoch - phases[1])


def train_cyclic(settings, list_data_train=None, list_data_val=None):
    """1-cycle policy (train_rnn.py:98-222): max rate = get_lr / 10, min = max / 10."""
    list_data_train, list_data_val = _data(settings, list_data_train, list_data_val)
    save_normalizations(settings)
    max_lr = get_lr(settings, list_data_train, list_data_val) / 10
    min_lr = max_lr / 10
    settings.learning_rate = min_lr
    rnn, criterion, optimizer = _setup(settings, list_data_train[0][0].shape[1])
    d_train, d_val = _monitors(settings)
    best_loss, t_start = float("inf"), time()
    for epoch in range(settings.cyclic_phases[-1]):
        _run_epoch(settings, rnn, criterion, optimizer, list_data_train)
        for group in optimizer.param_groups:
            group["lr"] = one_cycle_sched(epoch, min_lr, max_lr, settings.cyclic_phases)
        if (epoch + 1) % settings.monitor_interval == 0:
            _, _, best_loss = _monitor(settings, rnn, list_data_train, list_data_val, d_train, d_val, epoch, best_loss)
    tu.save_training_results(settings, d_val, time() - t_start)
    return rnn


def train_(settings, list_data_train=None, list_data_val=None):
    """Decay-on-plateau policy (train_rnn.py:259-372)."""
    list_data_train, list_data_val = _data(settings, list_data_train, list_data_val)
    save_normalizations(settings)
    rnn, criterion, optimizer = _setup(settings, list_data_train[0][0].shape[1])
    d_train, d_val = _monitors(settings)
    plateau = tu.StopOnPlateau(reduce_lr_on_plateau=True)
    best_loss, t_start = float("inf"), time()
    for epoch in range(settings.nb_epoch):
        _run_epoch(settings, rnn, criterion, optimizer, list_data_train)
        if (epoch + 1) % settings.monitor_interval == 0:
            # (the reference tests the plateau between evaluating and logging: a stopping epoch is not logged)
            m_train = tu.get_evaluation_metrics(settings, list_data_train, rnn, sample_size=len(list_data_val))
            m_val = tu.get_evaluation_metrics(settings, list_data_val, rnn, sample_size=None)
            if plateau.step(m_val["Acc"], optimizer) is True:
                break
            for k in m_train:
                d_train[k].append(m_train[k])
                d_val[k].append(m_val[k])
            d_train["epoch"].append(epoch + 1)
            d_val["epoch"].append(epoch + 1)
            if not getattr(settings, "no_dump", True):
                print(f"Epoch: {epoch} -- {tu.get_loss_string(m_train, m_val)}")
            if m_val["loss"] < best_loss:
                best_loss = m_val["loss"]
                torch.save(rnn.state_dict(), os.path.join(str(settings.rnn_dir), f"{settings.pytorch_model_name}.pt"))
    tu.save_training_results(settings, d_val, time() - t_start)
    return rnn


def train_swag(settings, list_data_train=None, list_data_val=None):
    """SWAG training (train_rnn.py:375-516): the plain loop at a constant learning rate; from
    `settings.swag_start_epoch` on, the parameters of every epoch enter the SwagModel statistics, and once three
    models are averaged SWAG draws are evaluated and the SwagModel is saved.  The reference pickles the whole module
    into `<name>_swag.pt`; here the file holds its state dict (same buffer names), loadable with weights_only=True.
    Returns (rnn, swag_rnn)."""
    from ..utils.swag_utils import SwagModel
    list_data_train, list_data_val = _data(settings, list_data_train, list_data_val)
    save_normalizations(settings)
    rnn, criterion, optimizer = _setup(settings, list_data_train[0][0].shape[1])
    swag_rnn = SwagModel(rnn)
    d_train, d_val = _monitors(settings)
    best_loss, t_start = float("inf"), time()
    for epoch in range(settings.nb_epoch):
        _run_epoch(settings, rnn, criterion, optimizer, list_data_train)
        if (epoch + 1) % settings.monitor_interval == 0:
            _, _, best_loss = _monitor(settings, rnn, list_data_train, list_data_val, d_train, d_val, epoch, best_loss)
        if epoch >= settings.swag_start_epoch:
            swag_rnn.update_parameters(rnn)
            if epoch >= settings.swag_start_epoch + 2:      # K > 1 deviations: the covariance draw is defined
                swag_rnn.last_metrics = tu.get_evaluation_metrics(settings, list_data_train, swag_rnn,
                                                                  sample_size=len(list_data_val), swag_sampling=True)
                torch.save(swag_rnn.state_dict(),
                           os.path.join(str(settings.rnn_dir), f"{settings.pytorch_model_name}_swag.pt"))
    tu.save_training_results(settings, d_val, time() - t_start)
    return rnn, swag_rnn


def train_swag_finetune(settings, list_data_train=None, list_data_val=None):
    """SWAG fine-tuning (train_rnn.py:516-663): load `settings.swag_pretrained_path` into a fresh model, train it with
    SGD (momentum 0.9, weight decay 1e-4 -- hard-coded in the reference, :541-546) for 30 epochs (:548-549), enter every
    epoch's parameters into the SwagModel statistics and evaluate / save SWAG draws from the third epoch on.
    The reference unpickles the pre-trained file (torch.load(..., weights_only=False), :537); here it must be a plain
    state dict, read with weights_only=True.  Returns (rnn, swag_rnn)."""
    from ..optim import FusedSGD
    from ..utils.swag_utils import SwagModel
    save_normalizations(settings)
    list_data_train, list_data_val = _data(settings, list_data_train, list_data_val)
    rnn = tu.get_model(settings, list_data_train[0][0].shape[1])
    rnn.load_state_dict(torch.load(settings.swag_pretrained_path, map_location="cpu", weights_only=True))
    criterion = nn.CrossEntropyLoss()
    settings.swag_start_epoch = 0
    settings.nb_epoch = 30
    swag_rnn = SwagModel(rnn)
    if settings.use_cuda:
        rnn.cuda()
        swag_rnn.cuda()
    optimizer = FusedSGD(rnn, lr=settings.learning_rate, momentum=0.9, weight_decay=1e-4)
    d_train, d_val = _monitors(settings)
    best_loss, t_start = float("inf"), time()
    for epoch in range(settings.nb_epoch):
        _run_epoch(settings, rnn, criterion, optimizer, list_data_train)
        if (epoch + 1) % settings.monitor_interval == 0:
            _, _, best_loss = _monitor(settings, rnn, list_data_train, list_data_val, d_train, d_val, epoch, best_loss)
        swag_rnn.update_parameters(rnn)
        if epoch >= 2:
            swag_rnn.last_metrics = tu.get_evaluation_metrics(settings, list_data_train, swag_rnn,
                                                              sample_size=len(list_data_val), swag_sampling=True)
            torch.save(swag_rnn.state_dict(),
                       os.path.join(str(settings.rnn_dir), f"{settings.pytorch_model_name}_swag.pt"))
    tu.save_training_results(settings, d_val, time() - t_start)
    return rnn, swag_rnn


def train(settings, list_data_train=None, list_data_val=None):
    """Entry point of `snn train_rnn` (train_rnn.py:245-256; the cyclic policy is chosen in cli/run.py:94-102)."""
    if getattr(settings, "swag", False):
        return train_swag(settings, list_data_train, list_data_val)
    if getattr(settings, "swag_pretrained_path", None):
        return train_swag_finetune(settings, list_data_train, list_data_val)
    if getattr(settings, "cyclic", False):
        return train_cyclic(settings, list_data_train, list_data_val)
    return train_(settings, list_data_train, list_data_val)
