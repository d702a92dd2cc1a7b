"""PCIe-inclusive rate of the headline workload: the caller's ragged rows start in pinned host memory and
the class probabilities end there.  H2D copy -> plan + pack_ragged -> BBB sample -> forward -> D2H, one stream, no
overlap between steps; prints light curves/s next to the resident-input rate of the same launches."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from supernnova_amd import engine as E  # noqa: E402

N, S, T = int(os.environ.get("CURVES", 1_250_000)), 50, 40
prec = os.environ.get("PRECISION", "f32")
dev = torch.device("cuda", 0)
cfg = E.Config(24, 32, 2, True, 2, "mean").with_precision(prec)
mu, rho = bench.bayes_init(cfg, 0, dev)
x, lengths = bench.synth_curves(N, T, 4, dev, seed=1000)
keep = (torch.arange(T, device=dev).unsqueeze(0) < lengths.unsqueeze(1))          # (N, T)
rows_h = x.permute(1, 0, 2)[keep].cpu().pin_memory()                              # (sum len, D), curve-major
del x, keep
lengths_host = lengths.cpu().numpy()
off_h = torch.from_numpy(np.concatenate([[0], np.cumsum(lengths_host.astype(np.int64))[:-1]]) * 24).pin_memory()
probs_h = torch.empty((N, S, 2), dtype=torch.float32).pin_memory()
prior = E.Prior()


def step(i, resident=None):
    rows, off = resident if resident else (rows_h.to(dev, non_blocking=True), off_h.to(dev, non_blocking=True))
    plan = E.Plan(cfg, lengths_host, dev)
    xp = E.pack_ragged(cfg, plan, rows, off)
    wk, _ = E.bbb_sample(cfg, mu, rho, S, prior, seed=1234, offset=i * S)
    _, probs = E.lstm_forward(cfg, plan, xp, wk, S, want_logits=False, want_probs=True, sample_major=False)
    if resident is None:
        probs_h.copy_(probs.view(N, S, 2), non_blocking=True)
    return probs


for name, res in (("pcie-inclusive", None), ("resident", (rows_h.to(dev), off_h.to(dev)))):
    step(0, res)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 3
    for i in range(K):
        step(1 + i, res)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{prec} {name}: {dt * 1e3:.1f} ms per {N} curves = {N / dt / 1e6:.3f} M light curves/s "
          f"(H2D {rows_h.numel() * 4 / 1e9:.2f} GB, D2H {probs_h.numel() * 4 / 1e9:.2f} GB)", flush=True)

# The same work with the copies on streams of their own (two device row buffers): the upload of step i+1 and the
# download of step i-1 run beside the forward kernel of step i.
up, down = torch.cuda.Stream(), torch.cuda.Stream()
main = torch.cuda.current_stream()
bufs = [(torch.empty(rows_h.shape, dtype=torch.float32, device=dev), torch.empty(off_h.shape, dtype=torch.int64, device=dev))
        for _ in range(2)]
uploaded = [torch.cuda.Event() for _ in range(2)]
consumed = [torch.cuda.Event() for _ in range(2)]


def upload(i):
    b = i % 2
    with torch.cuda.stream(up):
        up.wait_event(consumed[b])              # the pack of step i-2 has read this buffer
        bufs[b][0].copy_(rows_h, non_blocking=True)
        bufs[b][1].copy_(off_h, non_blocking=True)
        uploaded[b].record(up)


def run(K):
    upload(0)
    for i in range(K):
        b = i % 2
        main.wait_event(uploaded[b])
        plan = E.Plan(cfg, lengths_host, dev)
        xp = E.pack_ragged(cfg, plan, bufs[b][0], bufs[b][1])
        consumed[b].record(main)
        upload(i + 1)
        wk, _ = E.bbb_sample(cfg, mu, rho, S, prior, seed=1234, offset=i * S)
        _, probs = E.lstm_forward(cfg, plan, xp, wk, S, want_logits=False, want_probs=True, sample_major=False)
        done = torch.cuda.Event()
        done.record(main)
        with torch.cuda.stream(down):
            down.wait_event(done)
            probs_h.copy_(probs.view(N, S, 2), non_blocking=True)
            probs.record_stream(down)
    torch.cuda.synchronize()


for e in consumed:
    e.record(main)
run(1)
K = 4
t0 = time.perf_counter()
run(K)
dt = (time.perf_counter() - t0) / K       # includes one upload that nothing consumes (step K)
print(f"{prec} pcie-inclusive, copies on their own streams: {dt * 1e3:.1f} ms per {N} curves = {N / dt / 1e6:.3f} M light curves/s",
      flush=True)
