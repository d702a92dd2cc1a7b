"""The multi-rank training path on the GPU: two processes, launched the way `bench.py --gpus 2` launches
its ranks (supernnova_amd.distributed.self_launch), share cuda:0 and run FusedAdam.step_flat with the all-reduce in
it (gloo: the box has one GPU; the RCCL call sits in the same place).  After 3 steps both ranks hold the parameters a
single process reaches on the whole batch."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("model", ["bayesian"])
def test_two_ranks_on_one_gpu_match_single_process(model, tmp_path):
    from supernnova_amd.distributed import self_launch
    import dp_rehearsal_worker as W
    out = str(tmp_path / "dp")
    env_before = os.environ.get("SNN_BENCH_BACKEND")
    os.environ["SNN_BENCH_BACKEND"] = "gloo"          # self_launch: ranks may share a device
    try:
        rc = self_launch(os.path.join(ROOT, "tests", "dp_rehearsal_worker.py"), [model, "3", out], 2, timeout=240)
    finally:
        if env_before is None:
            os.environ.pop("SNN_BENCH_BACKEND")
        else:
            os.environ["SNN_BENCH_BACKEND"] = env_before
    assert rc == 0, f"2-rank rehearsal exited with {rc}"
    r0, r1 = (np.load(f"{out}.rank{r}.npz") for r in range(2))
    assert np.array_equal(r0["flat"], r1["flat"]), "ranks diverged (Adam runs redundantly on identical reduced gradients)"
    ref, ref_losses = W.run(model, 3, 0, 1, torch.device("cuda", 0), False)
    # half-batch gradients averaged vs full-batch gradient: equal to summation rounding (2e-6 of max |grad|, see
    # test_data_parallel_gradient_equals_full_batch_gradient); Adam's normalised update turns that into at most a few
    # 1e-7 per step on the parameters
    assert float(np.abs(r0["flat"] - ref).max()) < 5e-6
    # each rank reports CE of its own half + the shared KL term; their mean is the full-batch loss
    np.testing.assert_allclose(0.5 * (r0["losses"] + r1["losses"]), ref_losses, rtol=2e-6)
