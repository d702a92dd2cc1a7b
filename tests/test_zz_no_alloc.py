"""snn_lstm_forward_noalloc (the one entry point that allocates nothing, include/snn_hip.h) against
snn_lstm_forward through the C ABI: bit-identical logits and probabilities on two calls in a row, the second on a
smaller input (the library's fixed scratch keeps the first call's contents), for the fp32 paired-tile kernel, the
one-tile kernel, the masked kernel and the split-precision kernels; plus the size limit's error."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(E, cfg, n, T, S, seed, dev, masks=None):
    g = np.random.default_rng(seed)
    lengths = g.integers(1, T + 1, n).astype(np.int32)
    lengths[0] = T
    x = g.standard_normal((T, n, cfg.input_size)).astype(np.float32)
    for b in range(n):
        x[lengths[b]:, b] = 0
    plan = E.Plan(cfg, lengths, dev)
    xp = E.pack_x(cfg, plan, torch.from_numpy(x).to(dev))
    flats = torch.from_numpy(g.uniform(-0.3, 0.3, (S, cfg.P)).astype(np.float32)).to(dev)
    wk = E.prepare_weights_multi(cfg, flats)
    ref_logits, ref_probs = E.lstm_forward(cfg, plan, xp, wk, S, masks=masks)
    # outputs are the caller's: allocated BEFORE the call under test and poisoned
    logits = torch.full_like(ref_logits, float("nan"))
    probs = torch.full_like(ref_probs, float("nan"))
    torch.cuda.synchronize()
    E.lstm_forward_noalloc(cfg, plan, xp, wk, S, logits=logits, probs=probs, masks=masks)
    torch.cuda.synchronize()
    assert torch.equal(logits, ref_logits), f"logits differ (n={n}, T={T}, {cfg.precision})"
    assert torch.equal(probs, ref_probs), f"probabilities differ (n={n}, T={T}, {cfg.precision})"
    assert bool(torch.isfinite(probs).all())


@pytest.mark.parametrize("precision", ["f32", "f16x2", "bf16x3"])
def test_noalloc_matches_forward_twice(precision):
    from supernnova_amd import engine as E
    dev = torch.device("cuda", 0)
    cfg = E.Config(24, 32, 2, True, 2, "mean").with_precision(precision)
    _case(E, cfg, 5000, 40, 4, 1, dev)      # more work items than slabs fit: the grid is cut to the scratch
    _case(E, cfg, 70, 17, 3, 2, dev)        # second call, smaller input, stale scratch from the first
    _case(E, cfg, 20, 9, 1, 3, dev)         # one tile: the one-tile kernel


def test_noalloc_masked_and_unidirectional():
    from supernnova_amd import engine as E
    dev = torch.device("cuda", 0)
    cfg = E.Config(24, 32, 2, True, 2, "mean")
    m = E.Masks(mode=2, which=0xF, dropout=0.05, seed=99, offset=0)
    _case(E, cfg, 900, 30, 3, 4, dev, masks=m)
    _case(E, cfg, 100, 12, 2, 5, dev, masks=m)
    _case(E, E.Config(26, 64, 1, False, 3, "standard"), 300, 25, 2, 6, dev)
    _case(E, E.Config(78, 128, 2, True, 14, "mean"), 64, 20, 2, 7, dev)


def test_noalloc_rejects_oversize():
    from supernnova_amd import engine as E, _lib as L
    dev = torch.device("cuda", 0)
    cfg = E.Config(78, 128, 2, True, 14, "mean")
    assert L.load().snn_lstm_forward_noalloc_scratch_bytes() == 64 << 20
    # H=128: one workgroup's slab is max_T * (2*128*32 + 16*1024) * 4 bytes -> beyond 64 MiB from max_T ~ 680
    T, n = 1200, 3
    lengths = np.full(n, T, np.int32)
    plan = E.Plan(cfg, lengths, dev)
    xp = E.pack_x(cfg, plan, torch.zeros((T, n, 78), device=dev))
    wk = E.prepare_weights(cfg, torch.zeros(cfg.P, device=dev))
    probs = torch.empty((1, n, 14), device=dev)
    with pytest.raises(L.SnnError, match="scratch"):
        E.lstm_forward_noalloc(cfg, plan, xp, wk, 1, probs=probs)
    torch.cuda.synchronize()
