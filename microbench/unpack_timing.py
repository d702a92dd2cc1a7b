"""Timing of PackedSequence unpacking variants on the GPU (modules.unpack_packed_sequence vs torch's pad_packed_sequence)."""
import time
import numpy as np
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
from supernnova_amd.modules import unpack_packed_sequence

dev = "cuda:0"
for B in (128, 8192, 65536):
    g = np.random.default_rng(0)
    T, D = 40, 24
    lengths = np.sort(g.integers(3, T + 1, size=B))[::-1].copy()
    x = torch.randn(int(lengths[0]), B, D, device=dev)
    pk = pack_padded_sequence(x, torch.from_numpy(lengths.astype(np.int64)))
    for name, fn in (("pad_packed_sequence", lambda: pad_packed_sequence(pk)), ("unpack_packed_sequence", lambda: unpack_packed_sequence(pk))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print(B, name, f"{(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
