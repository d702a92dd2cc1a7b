"""Host-side profile of the fused training step at the reference's batch size (cProfile over 400 steps):
where the Python / ctypes time of one step goes.  python microbench/train_host_profile.py [batch]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch
from torch.nn.utils.rnn import pack_padded_sequence

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_train import settings                      # noqa: E402
from supernnova_amd.utils import training_utils as tu  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
st = settings()
torch.manual_seed(0)
rnn = tu.get_model(st, 24).to(dev)
opt = tu.get_optimizer(st, rnn)
g = np.random.default_rng(0)
lengths = np.sort(g.integers(3, 41, size=B))[::-1].copy()
x = g.standard_normal((int(lengths[0]), B, 24)).astype(np.float32)
packed = pack_padded_sequence(torch.from_numpy(x).to(dev), torch.from_numpy(lengths.astype(np.int64)))
target = torch.from_numpy(g.integers(0, 2, size=B)).to(dev)
for _ in range(30):
    tu.train_step_fused(st, rnn, packed, target, opt, B, 100)
torch.cuda.synchronize()
# host-only time: enqueue N steps without waiting for the GPU, then wait
N = 400
t0 = time.perf_counter()
for _ in range(N):
    tu.train_step_fused(st, rnn, packed, target, opt, B, 100)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, drained after another {1e3 * (t2 - t1):.2f} ms -> {1e3 * (t2 - t0) / N:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tu.train_step_fused(st, rnn, packed, target, opt, B, 100)
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr).sort_stats("tottime")
ps.print_stats(28)
