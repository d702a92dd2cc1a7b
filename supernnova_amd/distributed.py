"""Multi-GPU glue, one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI;
"gloo" on CPU for tests).

Inference: light curves are independent, so ranks take contiguous shards and run the same persistent
launch; there is NO collective on the data path, only an optional final gather of (N_local, S, C)
probabilities or of the per-curve median/std.
Training: data parallel.  Every rank draws the SAME eps / W_hh row masks (same Philox seed: seed the ranks
identically) and distinct per-curve masks (`rnn.dp_curve_base`), runs forward/backward on its slice of the
mini-batch with batch_size = the GLOBAL batch in the KL term, and the optimizer all-reduces ONE flat fp32
gradient buffer (160 KB vanilla ... 320 KB Bayesian at H=32) and averages it: mean CE over equal shards and
the rank-identical KL gradient both come out exactly as in a single-process step.  Adam then runs
redundantly on every rank (no parameter broadcast after step 0).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """Initialise the default group from torchrun's env (RANK / WORLD_SIZE / MASTER_*). Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def get_rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def shard_range(n, rank, world):
    """Contiguous, balanced shard [start, end) of n items for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def allreduce_mean_scale(flat, group=None):
    """Sum-all-reduce `flat` in place (one call) and return the factor that turns the sum into the mean."""
    w = world_size(group)
    if w > 1:
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            # rehearsal only (ranks sharing a GPU): gloo's own device-tensor path stalled here with two processes on
            # one card, so the buffer is staged through the host explicitly; RCCL ("nccl") reduces in place
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / w


def broadcast_parameters(model, src=0, group=None):
    """Step-0 synchronisation of the flat parameter buffer."""
    if world_size(group) > 1:
        dist.broadcast(model.flat_parameters(), src=src, group=group)


def gather_predictions(local, n_total, dst=0, group=None):
    """Gather contiguous shards (N_local, ...) -> (n_total, ...) on rank `dst` (None elsewhere)."""
    w, r = world_size(group), get_rank(group)
    if w == 1:
        return local
    sizes = [shard_range(n_total, k, w) for k in range(w)]
    nmax = max(e - s for s, e in sizes)
    pad = local.new_zeros((nmax,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    if r == dst:
        bufs = [torch.empty_like(pad) for _ in range(w)]
        dist.gather(pad, bufs, dst=dst, group=group)
        return torch.cat([b[: e - s] for b, (s, e) in zip(bufs, sizes)], dim=0)
    dist.gather(pad, None, dst=dst, group=group)
    return None
