"""Is the batch-128 training step bound by the host (enqueue time) or by the GPU?"""
import os, sys, time
import numpy as np
import torch
from torch.nn.utils.rnn import pack_padded_sequence
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_host_logic import settings
from supernnova_amd.utils import training_utils as tu
dev = "cuda:0"
st = settings(model="bayesian")
torch.manual_seed(0)
rnn = tu.get_model(st, 24).to(dev)
opt = tu.get_optimizer(st, rnn)
g = np.random.default_rng(0)
B, T = 128, 40
lengths = np.sort(g.integers(3, T + 1, size=B))[::-1].copy()
x = g.standard_normal((int(lengths[0]), B, 24)).astype(np.float32)
packed = pack_padded_sequence(torch.from_numpy(x).to(dev), torch.from_numpy(lengths.astype(np.int64)))
target = torch.from_numpy(g.integers(0, 2, size=B)).to(dev)
for _ in range(20):
    tu.train_step_fused(st, rnn, packed, target, opt, B, 100)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    tu.train_step_fused(st, rnn, packed, target, opt, B, 100)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / n:.0f} us/step, total {1e6 * (t2 - t0) / n:.0f} us/step")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    tu.train_step_fused(st, rnn, packed, target, opt, B, 100)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
