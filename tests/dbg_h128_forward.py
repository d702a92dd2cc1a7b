"""Debug aid, not collected by pytest (GPU): the H = 128 forward against the oracle on a 32-curve, 2-step case with parts
of the weights zeroed (W_hh, the biases, all of W_hh but one gate) -- isolates which product a wrong logit comes from.
   python tests/dbg_h128_forward.py"""
import os, sys, numpy as np, torch
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here)); sys.path.insert(0, _here)
import oracle
from supernnova_amd import engine as E
from util import make_curves
def t(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda:0")
D, H, T, N = 24, 128, 2, 32
cfg = E.Config(D, H, 1, False, 3, "mean"); ocfg = oracle.make_cfg(D, H, 1, False, 3, "mean")
for mode in ("full", "whh0", "bias0", "whh_only_gate"):
    w = np.random.default_rng(1).uniform(-0.2, 0.2, cfg.P).astype(np.float32)
    a, b = 4 * H * D, 4 * H * D + 4 * H * H
    if mode == "whh0": w[a:b] = 0
    if mode == "bias0": w[b:b + 8 * H] = 0
    x, lengths = make_curves(3, N, T, D, min_len=T)
    wk = E.prepare_weights(cfg, t(w))
    plan = E.Plan(cfg, lengths, "cuda:0"); xp = E.pack_x(cfg, plan, t(x))
    lo, pr = E.lstm_forward(cfg, plan, xp, wk, 1)
    o = oracle.forward(ocfg, x, lengths, w)
    if mode == "whh_only_gate":
        for g in range(4):
            w2 = w.copy(); W = w2[a:b].reshape(4 * H, H); keep = W[g * H:(g + 1) * H].copy(); W[:] = 0; W[g * H:(g + 1) * H] = keep
            wk = E.prepare_weights(cfg, t(w2)); lo, pr = E.lstm_forward(cfg, plan, xp, wk, 1); o = oracle.forward(ocfg, x, lengths, w2)
            print("only W_hh gate", g, "max abs diff", float(np.abs(lo[0].cpu().numpy() - o["logits"]).max()))
    else:
        print(mode, "max abs diff", float(np.abs(lo[0].cpu().numpy() - o["logits"]).max()))
