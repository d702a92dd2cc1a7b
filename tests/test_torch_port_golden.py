"""CPU: the stock-PyTorch port (oracle/torch_port.py: gradient oracle + cpu_baseline
harness) reproduces the reference's golden vectors, including losses and gradients
of the train step (utils/training_utils.py:484-525)."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence

from oracle import torch_port as tp
from util import load_golden, LOGIT_ATOL


def _sd(d, grad=False):
    return {k: torch.from_numpy(np.array(v)).requires_grad_(grad) for k, v in d.items()}


def _packed(g):
    return pack_padded_sequence(torch.from_numpy(g["x"]), torch.from_numpy(g["lengths"].astype(np.int64)))


def _eps_iter(cfg_sizes, flat_eps):
    off = 0
    for n in cfg_sizes:
        yield torch.from_numpy(flat_eps[off:off + n].copy())
        off += n


def _bayes_sizes(D, H, C, F):
    s = []
    for l in range(2):
        I = D if l == 0 else 2 * H
        for d in range(2):
            s += [4 * H * I, 4 * H * H, 4 * H, 4 * H]
    return s + [C * F]


def test_vanilla():
    g = load_golden("g1_vanilla_mean")
    with torch.no_grad():
        lo = tp.vanilla_forward(_sd(g["sd"]), _packed(g))
    np.testing.assert_allclose(lo.numpy(), g["logits"], atol=1e-6)


def test_variational_seed_replay():
    g = load_golden("g2_variational_mean")
    sd = _sd(g["sd"])
    for s in range(g["logits"].shape[0]):
        torch.manual_seed(1000 + s)
        with torch.no_grad():
            lo = tp.variational_forward(sd, _packed(g), dropout=float(g["dropout"]))
        np.testing.assert_allclose(lo.numpy(), g["logits"][s], atol=1e-6)


def test_bayesian_eps():
    g = load_golden("g3_bayesian_mean")
    sd = _sd(g["sd"])
    for s in range(g["logits"].shape[0]):
        with torch.no_grad():
            lo, kl = tp.bayesian_forward(sd, _packed(g), eps=_eps_iter(_bayes_sizes(24, 32, 2, 64), g["eps"][s]))
        np.testing.assert_allclose(lo.numpy(), g["logits"][s], atol=1e-6)
        assert abs(float(kl) - g["kl"][s]) / g["kl"][s] < 1e-6


@pytest.mark.parametrize("model", ["bayesian", "variational", "vanilla"])
def test_train_step_grads(model):
    g = load_golden(f"g4_train_{model}")
    sd = _sd(g["sd0"], grad=True)
    target = torch.from_numpy(g["target"])
    B, nb = int(g["batch_size"]), int(g["num_batches"])
    kw = {}
    torch.manual_seed(4000)
    if model == "bayesian":
        kw["eps"] = _eps_iter(_bayes_sizes(24, 32, 2, 64), g["eps"][0])
    else:
        kw["dropout"] = float(g["dropout"])
    loss, logits = tp.train_loss(model, sd, _packed(g), target, B, nb, **kw)
    loss.backward()
    assert abs(float(loss) - g["loss"][0]) < 1e-6 * max(1, abs(g["loss"][0]))
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"][0], atol=1e-6)
    for k, ref in g["grads0"].items():
        got = sd[k].grad.numpy()
        scale = max(np.abs(ref).max(), 1e-12)
        assert np.abs(got - ref).max() / scale < 1e-5, k
