"""The sampled-mode gate (tests/util.sampled_gate) applied to the REFERENCE against itself: fixture
g5_gate_<model> holds two independent reference runs (A, B) of S = 50 draws for 10 000 curves in mini-batches of 5
(tools/make_golden.py: g5_gate).  B must pass the gate against A -- otherwise the gate would not be a statement
about the implementation under test.  (CPU only; the GPU side is tests/test_gpu_modules.py::test_sampled_mode_gate.)"""
import numpy as np
import pytest

from util import load_golden, sampled_gate, assert_sampled_gate


@pytest.mark.parametrize("model", ["bayesian", "variational"])
def test_reference_passes_its_own_gate(model):
    g = load_golden(f"g5_gate_{model}")
    assert g["mean"].shape == (2, 10000, 2) and int(g["S"]) == 50
    res = sampled_gate(g["mean"][1], g["std"][1], g["mean"][0], g["std"][0], int(g["S"]), int(g["group"]))
    assert_sampled_gate(res)
    # the standardised differences are not degenerate: the predictive std is resolved for most curves
    assert np.median(g["std"][0][:, 1]) > 1e-4
