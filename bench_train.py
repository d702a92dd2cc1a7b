#!/usr/bin/env python
"""bench_train.py -- secondary benchmark (BASELINE configs[3]): BayesianRNN training step
(forward-train + CE + BPTT + BBB/KL backward + fused Adam), data-parallel over RCCL.

  python bench_train.py [--batch 128] [--steps 50] [--warmup 10] [--path fused|autograd]
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 ... bench_train.py --gpus N

Weak scaling: every rank trains on `--batch` curves per step (global batch = N x batch); one all-reduce of
the flat fp32 gradient (320 KB) per step.  Prints one JSON line (rank 0).  bench.py remains the driver's
contract benchmark (inference metric of BASELINE.json)."""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
from torch.nn.utils.rnn import pack_padded_sequence

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def settings(**kw):
    d = dict(model="bayesian", layer_type="lstm", nb_classes=2, hidden_dim=32, num_layers=2, dropout=0.05,
             bidirectional=True, use_cuda=True, rnn_output_option="mean", pi=0.75, log_sigma1=-1.0,
             log_sigma2=-7.0, rho_scale_lower=4.0, rho_scale_upper=3.0, log_sigma1_output=-1.0,
             log_sigma2_output=-7.0, rho_scale_lower_output=4.0, rho_scale_upper_output=3.0,
             random_length=False, random_redshift=False, redshift="none", no_dump=True,
             learning_rate=1e-3, weight_decay=0.0)
    d.update(kw)
    return SimpleNamespace(**d)


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--warmup-ms", type=float, default=300.0,
                    help="single rank: after the warm-up steps keep stepping (untimed) until the warm-up has lasted this long")
    ap.add_argument("--batch", type=int, default=128, help="curves per GPU per step")
    ap.add_argument("--model", default="bayesian", choices=["bayesian", "variational", "vanilla"])
    ap.add_argument("--path", default="fused", choices=["fused", "autograd"])
    ap.add_argument("--cpu-steps", type=int, default=20, help="train steps of the CPU baseline (0 = skip)")
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--features", type=int, default=24, help="input width D (24: 4 filters; 78: 6 filters + redshift)")
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--max-len", type=int, default=40)
    ap.add_argument("--layer-type", default="lstm", choices=["lstm", "gru", "rnn"], help="VanillaRNN only for gru / rnn")
    ap.add_argument("--full-length", action="store_true", help="every curve has max-len steps (configs[4]: 300)")
    return ap


def main():
    args = parser().parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench_train.py --gpus N`: one fresh process per GPU, started before anything touches the GPU
        from supernnova_amd.distributed import self_launch
        raise SystemExit(self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SNN_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices;
    # correctness only -- the gradient all-reduce is staged through the host, see distributed.allreduce_mean_scale)
    backend = os.environ.get("SNN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from supernnova_amd import distributed as D
    D.init_from_env(backend, dev)
    out = measure(args, rank, world, dev, backend)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def measure(args, rank, world, dev, backend):
    """Warm-up + timed training steps of one configuration on an initialised process group (bench.py calls this for
    the `train` / `train_dp` objects of its line); returns the result dict on rank 0, None elsewhere."""
    # SNN_BENCH_WATCHDOG=<seconds>: INACTIVITY limit, re-armed after init and after every step, so a stuck rank prints
    # where it is and exits while slow progress does not trip it; SNN_BENCH_TRACE=1 prints a time stamp per step
    wd = int(os.environ.get("SNN_BENCH_WATCHDOG", "0"))
    trace = bool(os.environ.get("SNN_BENCH_TRACE"))
    import faulthandler

    def alive(what):
        if wd:
            faulthandler.dump_traceback_later(wd, exit=True)
        if trace:
            print(f"[r{os.environ.get('RANK', '0')} {time.strftime('%H:%M:%S')} +{time.perf_counter() - t_start:7.3f}s] {what}",
                  file=sys.stderr, flush=True)
    t_start = time.perf_counter()
    alive("start")
    from supernnova_amd import distributed as D
    from supernnova_amd.utils import training_utils as tu
    alive("process group up")

    st = settings(model=args.model, weight_decay=1e-7 if args.model == "variational" else 0.0,
                  hidden_dim=args.hidden, nb_classes=args.classes, layer_type=getattr(args, "layer_type", "lstm"))
    Dw = args.features
    torch.manual_seed(0)                       # identical init and identical Philox seeds on every rank
    rnn = tu.get_model(st, Dw).to(dev)
    D.broadcast_parameters(rnn)
    rnn.dp_curve_base = rank * args.batch
    opt = tu.get_optimizer(st, rnn)
    g = np.random.default_rng(100 + rank)
    B, T = args.batch, args.max_len
    lengths = np.sort(g.integers(3, T + 1, size=B))[::-1].copy()
    if args.full_length:
        lengths[:] = T
    x = g.standard_normal((int(lengths[0]), B, Dw)).astype(np.float32)
    for b in range(B):
        x[lengths[b]:, b] = 0
    packed = pack_padded_sequence(torch.from_numpy(x).to(dev), torch.from_numpy(lengths.astype(np.int64)))
    target = torch.from_numpy(g.integers(0, args.classes, size=B)).to(dev)
    crit = torch.nn.CrossEntropyLoss()
    nb, gB = 100, B * world

    def step():
        if args.path == "fused":
            return tu.train_step_fused(st, rnn, packed, target, opt, gB, nb)
        return tu.train_step(st, rnn, packed, target, crit, opt, gB, nb)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    t_w = time.perf_counter()
    for i in range(args.warmup):
        loss = step()
        alive(f"warm-up step {i} enqueued")
    barrier()
    # a short step leaves the device at its idle clocks after a handful of warm-up steps (B=128: 30 steps are 15 ms and
    # the timed loop then reads 0.5 ms per step where a sustained loop reads 0.32): keep stepping, untimed, until the
    # warm-up has lasted --warmup-ms.  Single rank only (under DP every rank must take the same number of steps).
    n_extra = 0
    while world == 1 and 1e3 * (time.perf_counter() - t_w) < args.warmup_ms:
        for _ in range(10):
            loss = step()
        n_extra += 10
        torch.cuda.synchronize()
    alive("warm-up done")
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step()
        alive(f"step {i} enqueued")
    barrier()
    dt = time.perf_counter() - t0
    alive("timed steps done")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(loss)) or bool(os.environ.get("SNN_BENCH_NOCHECK"))      # (timing-ablation builds produce garbage)
    # phase times of the fused step (HIP events on the launch stream between the C-ABI calls), a few extra steps
    phases = None
    if args.path == "fused":
        acc = {}
        n_ph = 5
        for _ in range(n_ph):
            rnn._timeline = []
            step()
            torch.cuda.synchronize()
            tl = rnn._timeline
            for (_, a), (name, b) in zip(tl[:-1], tl[1:]):
                acc[name] = acc.get(name, 0.0) + a.elapsed_time(b) / n_ph
        rnn._timeline = None
        phases = acc
    out = None
    if rank == 0:
        out = {"metric": f"light curves/s trained, {args.model} RNN (fwd + ELBO/KL bwd + Adam), data-parallel",
               "value": gB * args.steps / dt, "unit": "light curves/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup + n_extra, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{args.model} {getattr(args, 'layer_type', 'lstm').upper()} {args.hidden}x2 bidirectional D={Dw} C={args.classes}, batch {B} per GPU, "
                                      + (f"{T}-step sequences, " if args.full_length else f"lengths U{{3..{T}}}, ") +
                                      f"path={args.path}", "global_batch": gB, "parallelism": f"dp{world}"}}
        # algorithmic FLOPs: 3 x F_fwd per curve per step, F_fwd = l * 16H(D+4H) + 4HC (L=2, bidirectional)
        H, C = args.hidden, args.classes
        NG = {"lstm": 4, "gru": 3, "rnn": 1}[getattr(args, "layer_type", "lstm")]      # gate blocks per hidden unit
        f_fwd = float(lengths.astype(np.int64).sum()) * 4 * NG * H * (Dw + 4 * H) + B * 4 * H * C
        PEAK = 157.3      # fp32 MFMA TFLOP/s (MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32)
        if phases is not None:
            # achieved = this rank's algorithmic FLOPs per step / the step time of the TIMED loop (back-to-back steps: the
            # host prepares step i + 1 while the GPU runs step i).  The phase times below come from separate steps with a
            # synchronisation after each, so their sum is the latency of ONE isolated step, host gaps included.
            ach = 3 * f_fwd / (dt / args.steps) / 1e12
            out["roofline"] = {
                "bound": "mfma", "kernel": "whole fused step, timed loop (phases below: isolated steps; per-kernel times: profiles/)",
                "forward_kernel": __import__("supernnova_amd.engine", fromlist=["x"]).last_forward_kernel(),
                "backward_path": (__import__("supernnova_amd._lib", fromlist=["x"]).load().snn_last_backward_path() or b"").decode(),
                "achieved": ach, "peak": PEAK, "unit": "TFLOP/s", "frac": ach / PEAK, "traffic": None,
                "flop_per_step": 3 * f_fwd, "isolated_step_ms": sum(phases.values()),
                "phases_ms": phases,
                "forward": {"flop": f_fwd, "ms": phases.get("forward"),
                            "frac": f_fwd / (phases["forward"] * 1e-3) / 1e12 / PEAK},
                "backward": {"flop": 2 * f_fwd, "ms": phases.get("backward"),
                             "frac": 2 * f_fwd / (phases["backward"] * 1e-3) / 1e12 / PEAK}}
        if args.cpu_steps > 0 and world == 1:
            from oracle import torch_port as tp
            sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in rnn.state_dict().items()}
            pk = pack_padded_sequence(torch.from_numpy(x), torch.from_numpy(lengths.astype(np.int64)))
            tg = target.cpu()
            copt = torch.optim.Adam(list(sd.values()), lr=1e-3)
            t0 = time.perf_counter()
            for _ in range(args.cpu_steps):
                copt.zero_grad()
                l, _ = tp.train_loss(args.model, sd, pk, tg, B, nb, hidden=args.hidden)
                l.backward()
                copt.step()
            cdt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": B * args.cpu_steps / cdt, "unit": "light curves/s",
                                   "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"{args.cpu_steps} train steps of the same batch on the stock-PyTorch port"}
    faulthandler.cancel_dump_traceback_later()
    return out


if __name__ == "__main__":
    main()
