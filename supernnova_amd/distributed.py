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


def self_launch(script, argv, n_ranks, timeout=None):
    """Run `script argv` as n_ranks child processes, one per GPU, with torchrun's environment (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), and return the worst exit code.  The parent must not have
    touched the GPU (it only counts devices, which does not initialise HIP); children inherit stdout / stderr, so
    rank 0's JSON line is the parent's output.  With fewer GPUs than ranks the children would collide on RCCL
    (one communicator rank per device), so that case needs SNN_BENCH_BACKEND=gloo (rehearsal: ranks share devices)."""
    import socket
    import subprocess
    import sys
    n_dev = torch.cuda.device_count()
    if n_dev < n_ranks and os.environ.get("SNN_BENCH_BACKEND", "nccl") == "nccl":
        print(f"{os.path.basename(script)}: --gpus {n_ranks} but {n_dev} GPU(s) visible; for a rehearsal with ranks "
              "sharing a GPU set SNN_BENCH_BACKEND=gloo", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", script, *argv], env=env))
    rc = 0
    try:
        pending = list(procs)
        import time
        t0 = time.monotonic()
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is not None:
                    pending.remove(p)
                    if code != 0:
                        rc = rc or code
            if rc or (timeout is not None and time.monotonic() - t0 > timeout):
                rc = rc or 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:          # a failed or timed-out rank: stop exactly the children started here
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def get_rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def shard_range(n, rank, world):
    """Contiguous, balanced shard [start, end) of n items for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def allreduce_mean_scale(flat, group=None, producer_stream=None, mean=True):
    """Sum-all-reduce `flat` in place (one call) and return the factor that turns the sum into the mean (1.0 with
    mean=False: the caller has already weighted its contribution, see _HipRNN.fused_train_step).

    The MEAN of per-rank gradients equals the single-process gradient only for EQUAL shards (each rank's CE is a mean
    over its own curves): train_step refuses unequal shards; the fused step weights every curve by 1 / global batch
    and every rank's (identical) KL gradient by 1 / world instead, sums, and is exact for any split.

    Ordering: the kernels of this package are launched on torch's CURRENT stream (engine._stream), and
    torch.distributed's all_reduce orders itself against the current stream on both sides (RCCL: its internal stream
    waits on an event recorded on the current stream, and the current stream waits on the collective's end event
    before the call returns).  So a gradient produced by the preceding launches is complete when the collective reads
    it, and the Adam launch that follows sees the reduced values -- provided producer and consumer ARE the current
    stream.  `producer_stream` makes that explicit for a caller that produced `flat` elsewhere: the current stream
    waits on it first."""
    w = world_size(group)
    if w > 1:
        if flat.is_cuda and producer_stream is not None and producer_stream != torch.cuda.current_stream(flat.device):
            torch.cuda.current_stream(flat.device).wait_stream(producer_stream)
        if flat.is_cuda and dist.get_backend(group) == "gloo" and not os.environ.get("SNN_DP_GLOO_DEVICE"):
            # gloo (rehearsal of the multi-rank path with ranks sharing a GPU).  ProcessGroupGloo's device-tensor path
            # (a private stream that waits on an event of the current stream, pinned staging, a worker thread) stalls
            # when two processes share one GPU and the producer of the tensor is still in flight at the call: a few
            # rounds pass, then one takes seconds, then both ranks sit in all_reduce for good.  It does so WITHOUT any
            # kernel of this library (torch matmul + pinned upload + all_reduce: microbench/gloo_stall_variants.py
            # `heavy`), and it does not when the current stream is synchronised before the call (`step_sync`); records in
            # profiles/r2/gloo_stall_variants.txt.  .cpu() below is that synchronisation plus the staging copy gloo
            # would do itself; copy_ is enqueued on the current stream.  The RCCL path (else branch) is not involved.
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / w if mean else 1.0


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
    dev = local.device
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()      # gloo has no device-tensor gather (rehearsal with ranks sharing a GPU)
    if local.shape[0] == nmax:
        pad = local.contiguous()
    else:
        pad = local.new_zeros((nmax,) + tuple(local.shape[1:]))
        pad[: local.shape[0]] = local
    if r == dst:
        out = local.new_empty((w, nmax) + tuple(local.shape[1:]))
        dist.gather(pad, list(out.unbind(0)), dst=dst, group=group)
        if all(e - s == nmax for s, e in sizes):
            return out.view((w * nmax,) + tuple(local.shape[1:])).to(dev)
        return torch.cat([out[k, : e - s] for k, (s, e) in enumerate(sizes)], dim=0).to(dev)
    dist.gather(pad, None, dst=dst, group=group)
    return None
