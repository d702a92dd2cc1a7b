"""Forward kernel time for the wider layer-0 buckets at H = 32 (run with and without SNN_NO_PAIR=1)."""
import sys
import numpy as np
import torch
from supernnova_amd import engine as E

dev = "cuda:0"
N, S, T = 200000, 50, 40
for D in (24, 44, 78):
    cfg = E.Config(D, 32, 2, True, 2, "mean")
    g = np.random.default_rng(0)
    lengths = g.integers(3, T + 1, size=N).astype(np.int32)
    rows = torch.randn(int(lengths.sum()), D, device=dev)
    off = np.zeros(N, np.int64)
    np.cumsum(lengths[:-1].astype(np.int64) * D, out=off[1:])
    plan = E.Plan(cfg, lengths, dev)
    xp = E.pack_ragged(cfg, plan, rows, torch.from_numpy(off).to(dev))
    wk = E.prepare_weights_multi(cfg, (0.2 * torch.randn(S, cfg.P, device=dev)))
    for _ in range(2):
        E.lstm_forward(cfg, plan, xp, wk, S, want_logits=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        E.lstm_forward(cfg, plan, xp, wk, S, want_logits=False)
    b.record()
    torch.cuda.synchronize()
    print(f"D={D}: {a.elapsed_time(b) / 3:.1f} ms per launch ({N} curves x {S} samples)", flush=True)
