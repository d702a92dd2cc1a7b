"""Worker of tests/test_gpu_dp_rehearsal.py: one rank of a 2-rank data-parallel training run whose ranks share
cuda:0 (gloo rendezvous; the one-GPU box has no second device for RCCL).  Runs `steps` fused train steps on this
rank's half of the batch and writes the final flat parameters (and the gathered predictions on rank 0)."""
import faulthandler
import os
import sys

import numpy as np
import torch
from torch.nn.utils.rnn import pack_padded_sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def batch(B=96, T=25, D=24, seed=12):
    g = np.random.default_rng(seed)
    lengths = np.sort(g.integers(2, T + 1, size=B))[::-1].copy()
    x = g.standard_normal((int(lengths[0]), B, D)).astype(np.float32)
    for b in range(B):
        x[lengths[b]:, b] = 0
    return x, lengths, g.integers(0, 2, size=B)


def run(model, steps, rank, world, dev, process_group_ready):
    from test_host_logic import settings
    from supernnova_amd import distributed as D
    from supernnova_amd.utils import training_utils as tu
    st = settings(model=model, dropout=0.0)
    torch.manual_seed(3)                      # same init and the same Philox seeds on every rank
    rnn = tu.get_model(st, 24).to(dev)
    if process_group_ready:
        D.broadcast_parameters(rnn)
    opt = tu.get_optimizer(st, rnn)
    x, lengths, target = batch()
    B = len(lengths)
    s, e = D.shard_range(B, rank, world)
    if world == 2 and os.environ.get("SNN_DP_SPLIT"):       # unequal shards: [0, split) and [split, B)
        cut = int(os.environ["SNN_DP_SPLIT"])
        s, e = (0, cut) if rank == 0 else (cut, B)
    pk = pack_padded_sequence(torch.from_numpy(x[:int(lengths[s]), s:e]).to(dev), torch.from_numpy(lengths[s:e].astype(np.int64)))
    tg = torch.from_numpy(target[s:e]).to(dev)
    losses = []
    for i in range(steps):
        faulthandler.dump_traceback_later(60, exit=True)     # inactivity limit per step
        losses.append(float(tu.train_step_fused(st, rnn, pk, tg, opt, B, 100)))
    faulthandler.cancel_dump_traceback_later()
    return rnn.flat_parameters().detach().cpu().numpy(), np.array(losses), e - s


def main():
    model, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from supernnova_amd import distributed as D
    D.init_from_env("gloo")
    flat, losses, n_local = run(model, steps, rank, world, dev, True)
    # inference exchange: every rank contributes a (n_local, 4) block of "predictions"
    n = 37
    s, e = D.shard_range(n, rank, world)
    local = torch.arange(s, e, dtype=torch.float32, device=dev).view(-1, 1).repeat(1, 4)
    full = D.gather_predictions(local, n, dst=0)
    if rank == 0:
        assert full.is_cuda and torch.equal(full[:, 0].cpu(), torch.arange(n, dtype=torch.float32))
    np.savez(f"{out}.rank{rank}.npz", flat=flat, losses=losses, n_local=n_local)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
