"""CPU, world_size 2 over gloo: the N>1 path's host logic -- shard/gather of predictions and the
data-parallel gradient formulation (equal shards, global batch size in the KL term, mean all-reduce of one
flat buffer == single-process gradients).  Gradients come from the stock-PyTorch port (the oracle)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.nn.utils.rnn import pack_padded_sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from supernnova_amd import distributed as D
    from oracle import torch_port as tp
    from util import load_golden
    r, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # --- inference: shard, "classify", gather
    n = 37
    s, e = D.shard_range(n, rank, world)
    local = torch.arange(s, e, dtype=torch.float32).view(-1, 1, 1).repeat(1, 3, 2)
    full = D.gather_predictions(local, n, dst=0)
    if rank == 0:
        assert full.shape == (n, 3, 2) and torch.equal(full[:, 0, 0], torch.arange(n, dtype=torch.float32))
    else:
        assert full is None
    # --- training: DP gradient == single-process gradient
    g = load_golden("g4_train_bayesian")
    B, nb = int(g["batch_size"]), int(g["num_batches"])
    sizes = []
    for l in range(2):
        I = 24 if l == 0 else 64
        sizes += [128 * I, 128 * 32, 128, 128] * 2
    sizes += [2 * 64]

    def eps_iter():
        off = 0
        for k in sizes:
            yield torch.from_numpy(g["eps"][0][off:off + k].copy())
            off += k

    def grads(cols):
        sd = {k: torch.from_numpy(np.array(v)).requires_grad_(True) for k, v in g["sd0"].items()}
        lens = g["lengths"][cols].astype(np.int64)
        packed = pack_padded_sequence(torch.from_numpy(g["x"][: lens.max(), cols]), torch.from_numpy(lens))
        loss, _ = tp.train_loss("bayesian", sd, packed, torch.from_numpy(g["target"][cols]), B, nb, eps=eps_iter())
        loss.backward()
        return torch.cat([sd[k].grad.reshape(-1) for k in g["sd0"].keys()])

    cols = np.arange(B)
    s, e = D.shard_range(B, rank, world)          # equal halves of the length-sorted batch (lengths stay sorted)
    flat = grads(cols[s:e])
    scale = D.allreduce_mean_scale(flat)
    flat *= scale
    ref = grads(cols)
    err = float((flat - ref).abs().max() / ref.abs().max())
    q.put((rank, err))
    dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert set(res) == {0, 1}
    assert max(res.values()) < 1e-5, res


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    from supernnova_amd.distributed import shard_range
    for n in (1, 7, 8, 10_000_001):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
