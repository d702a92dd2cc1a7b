#!/usr/bin/env python
"""bench.py -- light curves/s classified, BayesianRNN, 50 MC samples (BASELINE.json metric).

One "step" = one pass of the inference hot path over this rank's shard of
synthetic light curves, all on the GPU with inputs already resident in HBM:
  host plan (length sort + tiling)  -> snn_pack_x -> snn_bbb_sample (Philox, S
  draws + KL) -> snn_lstm_forward (S x N forwards + head + softmax -> (N,S,C)).
Inference shards curves across ranks with no collective on the data path
(SURVEY 8e): scaling is weak (fixed curves per GPU); value = all ranks' curves /
max-over-ranks time.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2516.6  # dense bf16 MFMA: 16 x the fp32 rate (v_mfma_f32_32x32x16_bf16, 1024 FLOP/clk/SIMD)


def synth_curves(n, T, F_filters, device, seed, redshift=False, chunk=250_000):
    """SURVEY 8d generator: lengths ~ U{3..T}; per step one filter combination (one-hot over
    2^F-1), normalised flux / flux-error ~ N(0,1) for the filters present and the normalised-zero
    constant otherwise; delta_time ~ N(0,1) (first step = normalised 0). (T, n, D) time-major."""
    D = 2 * F_filters + 1 + (2 ** F_filters - 1) + (2 if redshift else 0)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lengths = torch.randint(3, T + 1, (n,), generator=g, device=device, dtype=torch.int32)
    x = torch.zeros((T, n, D), dtype=torch.float32, device=device)
    FLUX0, ERR0, DT0 = -6.48, -1.31, -2.0  # log-standardised zeros (tests/onthefly_model/data_norm.json scales)
    bits = torch.arange(F_filters, device=device)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        m = b - a
        combo = torch.randint(1, 2 ** F_filters, (T, m), generator=g, device=device)
        present = ((combo.unsqueeze(-1) >> bits) & 1).bool()
        flux = torch.randn((T, m, F_filters), generator=g, device=device)
        err = torch.randn((T, m, F_filters), generator=g, device=device)
        xs = x[:, a:b]
        xs[..., :F_filters] = torch.where(present, flux, torch.full_like(flux, FLUX0))
        xs[..., F_filters:2 * F_filters] = torch.where(present, err, torch.full_like(err, ERR0))
        dt = torch.randn((T, m), generator=g, device=device)
        dt[0] = DT0
        xs[..., 2 * F_filters] = dt
        xs[..., 2 * F_filters + 1:2 * F_filters + 1 + 2 ** F_filters - 1].scatter_(
            -1, (combo - 1).unsqueeze(-1), 1.0)
        if redshift:
            xs[..., -2] = torch.rand((1, m), generator=g, device=device) * 1.15 + 0.05
            xs[..., -1] = torch.rand((1, m), generator=g, device=device) * 0.049 + 0.001
        t = torch.arange(T, device=device).unsqueeze(-1)
        xs.mul_((t < lengths[a:b].unsqueeze(0)).unsqueeze(-1))
    return x, lengths


def bayes_init(cfg, seed, device):
    """Reference init (bayesian_rnn.py:55-62, 766-775): mu ~ U(-0.05, 0.05),
    rho ~ U(log(exp(sigma_mix/4)-1), log(exp(sigma_mix/3)-1)), head bias 0."""
    import math
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    pi, s1, s2 = 0.75, math.exp(-1.0), math.exp(-7.0)
    mix = math.sqrt(pi * s1 * s1 + (1 - pi) * s2 * s2)
    lo, hi = math.log(math.exp(mix / 4.0) - 1.0), math.log(math.exp(mix / 3.0) - 1.0)
    P, C = cfg.P, cfg.nb_classes
    mu = torch.rand(P, generator=g) * 0.1 - 0.05
    mu[P - C:] = 0.0
    rho = torch.rand(P - C, generator=g) * (hi - lo) + lo
    return mu.to(device), rho.to(device)


def flops_per_pass(cfg, lengths_sum, n_curves, S):
    """per curve per sample l*16H(D+4H) + 4HC (L=2, bidirectional); a GRU has 3 gates' rows and a tanh RNN 1
    where the LSTM has 4 (algorithmic FLOPs: rows the kernels pad to a full MFMA tile are not counted)."""
    H, D, C, nd = cfg.hidden_size, cfg.input_size, cfg.nb_classes, cfg.dirs
    ng = {"lstm": 4, "gru": 3, "rnn": 1}[cfg.cell]
    # per direction and step: 2 ng H (D + H) in layer 0, 2 ng H (dirs H + H) in layer 1; head: 2 (dirs H) C
    return S * (lengths_sum * nd * 2 * ng * H * (D + (2 + nd) * H) + n_curves * 2 * nd * H * C)


def cpu_baseline(cfg, x_host, lengths_host, mu, rho, S, n_curves, model="bayesian"):
    x_host = np.ascontiguousarray(x_host)
    """Reference call pattern (validate_rnn.py:539-563) on the stock-PyTorch CPU port."""
    from oracle import torch_port as tp
    from supernnova_amd import layout
    sd, off = {}, 0
    H, D, C, F = cfg.hidden_size, cfg.input_size, cfg.nb_classes, cfg.F
    shapes = []
    for l in range(cfg.num_layers):
        I = D if l == 0 else 2 * H
        shapes += [(4 * H, I), (4 * H, H), (4 * H,), (4 * H,)] * 2
    shapes += [(C, F), (C,)]
    mu_c, rho_c = mu.cpu(), None if rho is None else rho.cpu()
    if model == "variational":      # mu = the flat weight vector, no rho
        key_pairs = [(k, None) for k in layout.variational_keys(2, 2)]
    else:
        key_pairs = zip(layout.bayesian_keys(2, 2, "mu"), layout.bayesian_keys(2, 2, "rho") + [None])
    for (k_mu, k_rho), shp in zip(key_pairs, shapes):
        n = int(np.prod(shp))
        sd[k_mu] = mu_c[off:off + n].view(shp).clone()
        if k_rho is not None:
            sd[k_rho] = rho_c[off:off + n].view(shp).clone()
        off += n
    curves = [x_host[:lengths_host[i], i].copy() for i in range(n_curves)]
    t0 = time.perf_counter()
    tp.mc_predict(model, sd, curves, S, batch_size=128, hidden=H)
    dt = time.perf_counter() - t0
    return n_curves / dt, dt


def end_to_end(E, cfg, x, lengths, lengths_host, mu, rho, prior, S, steps, dev):
    """the PCIe-inclusive rate -- ragged rows in pinned HOST memory -> H2D -> plan / pack / sample /
    forward -> D2H of the (N, S, C) probabilities into pinned host memory.  Three streams (upload of step i+1 and
    download of step i-1 beside the forward of step i; two device row buffers).  Reported beside `value`, never as it."""
    N, T, C = x.size(1), x.size(0), cfg.nb_classes
    keep = torch.arange(T, device=dev).unsqueeze(0) < lengths.unsqueeze(1)            # (N, T)
    rows_h = x.permute(1, 0, 2)[keep].cpu().pin_memory()                              # (sum len, D), curve-major
    del keep
    off_h = torch.from_numpy(np.concatenate([[0], np.cumsum(lengths_host.astype(np.int64))[:-1]])
                             * cfg.input_size).pin_memory()
    probs_h = torch.empty((N, S, C), dtype=torch.float32).pin_memory()
    up, down, main_s = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()
    bufs = [(torch.empty(rows_h.shape, dtype=torch.float32, device=dev),
             torch.empty(off_h.shape, dtype=torch.int64, device=dev)) for _ in range(2)]
    uploaded = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]

    def upload(i):
        b = i % 2
        with torch.cuda.stream(up):
            up.wait_event(consumed[b])              # the pack of step i-2 has read this buffer
            bufs[b][0].copy_(rows_h, non_blocking=True)
            bufs[b][1].copy_(off_h, non_blocking=True)
            uploaded[b].record(up)

    def run(K, first):
        upload(first)
        for i in range(first, first + K):
            b = i % 2
            main_s.wait_event(uploaded[b])
            plan = E.Plan(cfg, lengths_host, dev)
            xp = E.pack_ragged(cfg, plan, bufs[b][0], bufs[b][1])
            consumed[b].record(main_s)
            if i + 1 < first + K:
                upload(i + 1)
            wk, _ = E.bbb_sample(cfg, mu, rho, S, prior, seed=1234, offset=i * S)
            _, probs = E.lstm_forward(cfg, plan, xp, wk, S, want_logits=False, want_probs=True, sample_major=False)
            done = torch.cuda.Event()
            done.record(main_s)
            with torch.cuda.stream(down):
                down.wait_event(done)
                probs_h.copy_(probs.view(N, S, C), non_blocking=True)
                probs.record_stream(down)
        torch.cuda.synchronize()
        return probs

    for e in consumed:
        e.record(main_s)
    run(1, 0)
    t0 = time.perf_counter()
    probs = run(steps, 2)
    dt = (time.perf_counter() - t0) / steps
    assert bool(torch.equal(probs_h[:1000], probs[:1000].cpu())) and bool(torch.isfinite(probs_h[-1000:]).all())
    return {"value": N / dt, "unit": "light curves/s", "ms_per_step": 1e3 * dt,
            "h2d_bytes": rows_h.numel() * 4 + off_h.numel() * 8, "d2h_bytes": probs_h.numel() * 4,
            "what": "pinned host rows -> H2D -> plan/pack/sample/forward -> D2H of (N,S,C) into pinned host memory; copies "
                    "on their own streams beside the forward kernel (microbench/pcie_inclusive.py)"}


def cpu_baseline_full(cfg, x, lengths_host, mu, rho, S):
    """BASELINE.md 3 protocol (minutes of CPU work; --cpu-full only): 12 800 curves = 100 batches of 128, median of 3
    runs on all host threads this process was given, plus the same protocol on 1 thread."""
    out = {"unit": "light curves/s", "kind": "port"}
    n = 12800
    xh, lh = x[:, :n].cpu().numpy(), lengths_host[:n]
    threads = torch.get_num_threads()
    runs = []
    for r in range(3):
        v, secs = cpu_baseline(cfg, xh, lh, mu, rho, S, n)
        runs.append(v)
        print(f"cpu baseline (full): run {r}: {v:.1f} curves/s in {secs:.0f} s on {threads} threads", file=sys.stderr, flush=True)
    out.update(value=float(np.median(runs)), runs=runs, cores=threads, sample=f"{n} curves, median of 3")
    torch.set_num_threads(1)
    runs1 = []
    for r in range(3):
        v1, secs = cpu_baseline(cfg, xh, lh, mu, rho, S, n)
        runs1.append(v1)
        print(f"cpu baseline (full): run {r}: {v1:.1f} curves/s in {secs:.0f} s on 1 thread", file=sys.stderr, flush=True)
    torch.set_num_threads(threads)
    out["one_thread"] = {"value": float(np.median(runs1)), "runs": runs1, "cores": 1, "sample": f"{n} curves, median of 3"}
    return out


def run(args, rank, world, dev, backend, dist):
    """One workload: warm-up + timed steps + the derived objects of the JSON line (rank 0 returns the dict)."""
    out = None
    args = __import__("copy").copy(args)          # (the workload defaults below edit curves / max_len)
    from supernnova_amd import engine as E

    big = args.workload == "elasticc"
    var = args.workload == "variational"
    plain = args.workload in ("gru", "rnn")      # VanillaRNN with settings.layer_type gru / rnn: one deterministic forward
    if var and args.precision == "bf16x3":
        raise SystemExit("the masked (variational) kernels are f32 and f16x2 only")
    if plain and args.precision != "f32":
        raise SystemExit("the gru / rnn cells are f32 only")
    if big:
        cfg = E.Config(78, 128, 2, True, 14, "mean")
        if args.curves == 1_250_000:
            args.curves = 8192
        if args.max_len == 40:
            args.max_len = 300
    elif plain:
        cfg = E.Config(24, 32, 2, True, 2, "mean", cell=args.workload)
        if args.samples == 50:
            args.samples = 1          # (validate_rnn.py:528-529: a VanillaRNN is classified with one forward)
    else:
        cfg = E.Config(24, 64 if args.workload == "h64" else 32, 2, args.workload != "uni", 2, "mean")
    cfg = cfg.with_precision(args.precision)
    S, N = args.samples, args.curves
    mu, rho = bayes_init(cfg, 0, dev)
    x, lengths = synth_curves(N, args.max_len, 6 if big else 4, dev, seed=1000 + rank, redshift=big)
    if big:
        lengths.fill_(args.max_len)      # exactly 300 steps for config 5
    lengths_host = lengths.cpu().numpy()
    prior = E.Prior()
    if var or plain:   # torch.nn.LSTM / GRU / RNN default init U(+-1/sqrt(H)); one weight image shared by all samples
        gw = torch.Generator(device="cpu")
        gw.manual_seed(0)
        k = 1.0 / np.sqrt(cfg.hidden_size)
        flat = ((torch.rand(cfg.P, generator=gw) * 2 - 1) * k).to(dev)
        flat_unscaled = flat.clone()
        if args.precision == "f16x2":   # the f16x2 kernel applies the mask bits; 1/(1-p) goes into the weights (snn_hip.h)
            H, off, sc = cfg.hidden_size, 0, float(np.float32(1.0) / np.float32(0.95))
            for l in range(2):
                I = cfg.input_size if l == 0 else 2 * H
                for d in range(2):
                    flat[off:off + 4 * H * I] *= sc
                    off += 4 * H * I
                    if d == 0:
                        flat[off:off + 4 * H * H] *= sc
                    off += 4 * H * H + 8 * H
        w_var = E.prepare_weights(cfg, flat)
    torch.cuda.synchronize()

    # one event pair per timed step, read after the closing barrier: the host is then free to build the plan of step
    # i + 1 (a few ms of host work per 1.25 M curves) while the GPU runs step i, as a serving loop would
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    kern_ms = []
    dispatched = [""]          # what snn_lstm_forward launched in the timed steps (snn_last_forward_kernel)
    from supernnova_amd import distributed as Dist

    def step(i, timed):
        plan = E.Plan(cfg, lengths_host, dev)
        xp = E.pack_x(cfg, plan, x)
        if plain:
            wk, kl, masks = w_var, torch.zeros(1, device=dev), None
        elif var:
            wk, kl = w_var, torch.zeros(1, device=dev)
            masks = E.Masks(mode=2, which=0xF, dropout=0.05, seed=1234, offset=i * S)
        else:
            wk, kl = E.bbb_sample(cfg, mu, rho, S, prior, seed=1234, offset=i * S)
            masks = None
        if timed:
            evs[i - args.warmup][0].record()
        _, probs = E.lstm_forward(cfg, plan, xp, wk, S, masks=masks, want_logits=False, want_probs=True,
                                  sample_major=False)
        if timed:
            evs[i - args.warmup][1].record()
            dispatched[0] = E.last_forward_kernel()
        # the only exchange of the inference path: per-curve mean / std over the S draws, (N_local, 2C),
        # reduced on the device and gathered to rank 0 -- inside the timed region
        if S > 1:
            std, mean = torch.std_mean(probs, dim=1, unbiased=True)
        else:
            mean = probs[:, 0]
            std = torch.zeros_like(mean)
        gathered = Dist.gather_predictions(torch.cat([mean, std], dim=1), world * N)
        if timed:
            evs[i - args.warmup][2].record()
        return probs, kl, gathered

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        probs, kl, gathered = step(args.warmup + i, True)
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b, _ in evs]
    gather_ms = [b.elapsed_time(c) for _, b, c in evs]
    if rank == 0:
        assert tuple(gathered.shape) == (world * N, 2 * cfg.nb_classes) and bool(torch.isfinite(gathered).all())
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    selfcheck = None
    if not os.environ.get("SNN_BENCH_NOCHECK"):      # timing-ablation builds produce garbage on purpose
        assert bool(torch.isfinite(probs).all()) and bool(torch.isfinite(kl).all())
        # Size-independent property at the full bench size: a random subset of the curves, run again as a launch of
        # its own with the same weight draws, must reproduce its rows of the big launch bit for bit (results do not
        # depend on tiling, work-queue order or the number of curves in flight).
    if not os.environ.get("SNN_BENCH_NOCHECK") and not args.no_selfcheck:
        gsub = torch.Generator(device="cpu")
        gsub.manual_seed(7 + rank)
        sub = torch.randperm(N, generator=gsub)[:min(N, 3000)].sort().values
        sub_d = sub.to(dev)
        splan = E.Plan(cfg, lengths_host[sub.numpy()], dev)
        sxp = E.pack_x(cfg, splan, x[:, sub_d].contiguous())
        i_last = args.warmup + args.steps - 1
        if plain:
            swk, smasks = w_var, None
        elif var:
            swk, smasks = w_var, E.Masks(mode=2, which=0xF, dropout=0.05, seed=1234, offset=i_last * S)
            # (per-curve Philox masks are keyed by the curve index, which changes in the subset: compare shapes only)
        else:
            swk, _ = E.bbb_sample(cfg, mu, rho, S, prior, seed=1234, offset=i_last * S)
            smasks = None
        _, sprobs = E.lstm_forward(cfg, splan, sxp, swk, S, masks=smasks, want_logits=False, want_probs=True,
                                   sample_major=False)
        torch.cuda.synchronize()
        if var:
            same = sprobs.shape == probs[sub_d].shape and bool(torch.isfinite(sprobs).all())
            selfcheck = f"{len(sub)} curves re-run in a separate launch: finite (masks are keyed by curve index)"
        else:
            same = bool(torch.equal(sprobs, probs[sub_d]))
            selfcheck = f"{len(sub)} random curves re-run in a separate launch with the same draws: bit-identical"
        assert same, "subset launch does not reproduce the rows of the full-size launch"

    if rank == 0:
        value = world * N * args.steps / dt
        fl = flops_per_pass(cfg, int(lengths_host.astype(np.int64).sum()), N, S)
        k_ms = float(np.mean(kern_ms))
        achieved = fl / (k_ms * 1e-3) / 1e12
        b3 = args.precision != "f32"
        npr = 6 if args.precision == "bf16x3" else 3     # piece products per fp32 product
        peak, kname = PEAK_FP32_MFMA_TFLOPS, dispatched[0]
        if b3:
            # executed 16-bit MFMA FLOPs: 6 (bf16x3) / 3 (f16x2) piece products per fp32 product, K padded to blocks of 16
            H, Dp16 = cfg.hidden_size, -(-cfg.input_size // 16) * 16
            per_step = 2 * 4 * H * ((Dp16 + H) + (2 * H + H)) * 2 * npr
            fl_exec = S * int(lengths_host.astype(np.int64).sum()) * per_step
            achieved_f32eq, achieved = achieved, fl_exec / (k_ms * 1e-3) / 1e12
            peak = PEAK_BF16_MFMA_TFLOPS
        # PMC bytes per launch of this workload: NOT measured in this run -- read from the committed rocprofv3 passes
        # (profiles/r3/traffic.json: one entry per workload, written by microbench/traffic_json.py from the passes of
        # microbench/profile_round3.sh; an entry counts only for the kernel and the sizes it was measured on)
        traffic = traffic_src = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r3", "traffic.json")))["workloads"][args.workload]
            if (tj["workload"] == {"curves_per_gpu": N, "samples": S, "max_len": args.max_len} and
                    tj["precision"] == args.precision and tj["kernel"].split(" grid=")[0] == kname.split(" grid=")[0]):
                traffic = tj["traffic_bytes_per_launch"]
                traffic_src = "profiles/r3/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, not this run)"
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": ("light curves/s classified, VariationalRNN 50 MC samples" if var else
                       f"light curves/s classified, VanillaRNN ({args.workload.upper()} cells)" if plain else
                       "light curves/s classified, BayesianRNN 50 MC samples"),
            "value": value, "unit": "light curves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "selfcheck": selfcheck,
            "dtype": (f"f32 (each product = {npr} {'bf16' if npr == 6 else 'fp16'} piece products, fp32 accumulate)"
                      if b3 else "f32"), "data": "synthetic",
            "config": {"workload": (f"BayesianRNN (BBB) LSTM 128x2 bidirectional, mean pooling, D=78 (6 filters + redshift), "
                                    f"C=14, S={S} MC samples, {args.max_len}-step sequences, {N} light curves per GPU "
                                    "(configs[4], ELAsTiCC-like)") if big else
                                   ("VariationalRNN (MC dropout p=0.05, Philox masks) LSTM 32x2 bidirectional, mean "
                                    f"pooling, D=24 (4 filters), C=2, S={S} MC samples, lengths U{{3..{args.max_len}}}, "
                                    f"{N} light curves per GPU (configs[1])") if var else
                                   (f"VanillaRNN, layer_type={args.workload} (training/vanilla_rnn.py:26-32), 32x2 bidirectional, "
                                    f"mean pooling, D=24 (4 filters), C=2, {S} forward per curve, lengths U{{3..{args.max_len}}}, "
                                    f"{N} light curves per GPU") if plain else
                                   (f"BayesianRNN (BBB) LSTM {cfg.hidden_size}x2 {'bidirectional' if cfg.bidirectional else 'unidirectional'}, mean pooling, D=24 (4 filters), "
                                    f"C=2, S={S} MC samples, lengths U{{3..{args.max_len}}}, "
                                    f"{N} light curves per GPU (configs[2]: 10M over 8 GPUs)"),
                       "curves_per_gpu": N, "samples": S, "max_len": args.max_len, "parallelism": f"shard{world}"},
            "gather": {"what": "per-curve mean and std over the S draws, (N_local, 2C) fp32, reduced on the device and "
                               "gathered to rank 0 (the path's only exchange; inside the timed region)",
                       "bytes_per_rank": N * 2 * cfg.nb_classes * 4, "ms_per_step": float(np.mean(gather_ms)),
                       "backend": (backend if world > 1 else None)},
            "roofline": {"bound": "mfma",
                         "kernel": kname, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": k_ms, "flop_per_launch": fl_exec if b3 else fl},
        }
        if b3:
            out["roofline"]["fp32_equivalent_tflops"] = achieved_f32eq   # algorithmic FLOPs / kernel time
        des = args.workload == "des"
        if world == 1 and not b3 and des and not args.no_fast_path:
            # the opt-in split-precision path on the same inputs (same parity bar): one warm-up + `steps` passes
            fcfg = cfg.with_precision("f16x2")
            fevs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps + 1)]
            for i in range(args.steps + 1):
                if i == 1:
                    torch.cuda.synchronize()
                    ft0 = time.perf_counter()
                fev = fevs[i]
                fplan = E.Plan(fcfg, lengths_host, dev)
                fxp = E.pack_x(fcfg, fplan, x)
                fwk, fkl = E.bbb_sample(fcfg, mu, rho, S, prior, seed=1234, offset=i * S)
                fev[0].record()
                _, fprobs = E.lstm_forward(fcfg, fplan, fxp, fwk, S, want_logits=False, want_probs=True, sample_major=False)
                fev[1].record()
            torch.cuda.synchronize()
            fdt = time.perf_counter() - ft0
            fk = [a.elapsed_time(b) for a, b in fevs[1:]]
            assert bool(torch.isfinite(fprobs).all())
            out["fast_path"] = {"precision": "f16x2 (each fp32 product = 3 fp16 piece products, fp32 accumulate; "
                                             "opt-in, same 1e-5 parity bar)",
                                "value": N * args.steps / fdt, "unit": "light curves/s", "ms_per_step": 1e3 * fdt / args.steps,
                                "kernel": E.last_forward_kernel(), "kernel_ms": float(np.mean(fk)),
                                "fp32_equivalent_tflops": fl / (float(np.mean(fk)) * 1e-3) / 1e12}
            del fxp, fwk, fprobs
        if world == 1 and not b3 and des and not args.no_end_to_end:
            out["end_to_end"] = end_to_end(E, cfg, x, lengths, lengths_host, mu, rho, prior, S, args.steps, dev)
        if args.cpu_full and world == 1:
            out["cpu_baseline_full"] = cpu_baseline_full(cfg, x, lengths_host, mu, rho, S)
        if args.cpu_curves > 0 and world == 1 and not plain:     # (the stock-PyTorch port under oracle/ times the LSTM models)
            n_cpu = min(args.cpu_curves, N)
            if big and args.cpu_curves == 3840:
                n_cpu = 128       # one reference batch: 300 steps at H=128 x 50 forwards is already tens of seconds of CPU work
            run_cpu = (lambda n: cpu_baseline(cfg, x[:, :n].cpu().numpy(), lengths_host[:n], flat_unscaled, None, S, n,
                                              model="variational")) if var else \
                      (lambda n: cpu_baseline(cfg, x[:, :n].cpu().numpy(), lengths_host[:n], mu, rho, S, n))
            # The reference runs with torch's default thread count (all cores); on many-core hosts the small per-step
            # matmuls of a 128-curve batch run FASTER on one thread, so both settings are probed on a few batches and
            # the sample is timed with the faster one: the baseline is the port at its best, not at its default.
            all_threads = torch.get_num_threads()
            probe, n_probe = {}, min(n_cpu, 128 if big else 640)
            for th in sorted({1, all_threads}):
                torch.set_num_threads(th)
                probe[th] = run_cpu(n_probe)[0]
                print(f"cpu baseline probe: {probe[th]:.1f} curves/s on {th} thread(s) ({n_probe} curves)", file=sys.stderr, flush=True)
            best = max(probe, key=probe.get)
            torch.set_num_threads(best)
            v, secs = run_cpu(n_cpu)
            torch.set_num_threads(all_threads)
            out["cpu_baseline"] = {"value": v, "unit": "light curves/s", "cores": best, "kind": "port",
                                   "sample": f"first {n_cpu} curves of the same workload, batches of 128 x {S} "
                                             f"sequential forwards (reference call pattern), {secs:.1f} s; thread probe on "
                                             f"{n_probe} curves: " + ", ".join(f"{t} thread(s) {r:.0f}/s" for t, r in sorted(probe.items()))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--curves", type=int, default=1_250_000, help="light curves per GPU (10M / 8)")
    ap.add_argument("--samples", type=int, default=50)
    ap.add_argument("--max-len", type=int, default=40)
    ap.add_argument("--workload", choices=["des", "elasticc", "variational", "h64", "uni", "gru", "rnn"], default="des",
                    help="des: BASELINE metric config (H=32, 4 filters, lengths U{3..40}); elasticc: configs[4] "
                         "(H=128, 6 filters + redshift, 14 classes, 300-step sequences); variational: configs[1] "
                         "(VariationalRNN MC dropout p=0.05, Philox masks in-kernel, same shapes as des); h64: des with "
                         "hidden_dim 64; uni: des with bidirectional=False; gru / rnn: VanillaRNN with settings.layer_type gru / rnn, one forward per curve "
                         "-- all but des are secondary measurements")
    ap.add_argument("--cpu-curves", type=int, default=3840, help="curves of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-selfcheck", action="store_true",
                    help="skip the subset re-run after the timed region (profiling runs: keeps the kernel statistics to "
                         "the timed launches)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the PCIe-inclusive measurement the default (f32, des, 1 GPU) run appends as `end_to_end`")
    ap.add_argument("--cpu-full", action="store_true",
                    help="also time the CPU port with the BASELINE.md 3 protocol (12 800 curves, median of 3, and 1 thread): "
                         "several minutes of CPU work")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workloads (variational, elasticc, h64, uni, gru, rnn) the default run appends as `secondary`")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the training measurements (BASELINE configs[3]) the default run appends as secondary.train "
                         "and a --gpus N run as train_dp")
    ap.add_argument("--no-fast-path", action="store_true",
                    help="skip the extra f16x2 measurement the default (f32, des, 1 GPU) run appends as `fast_path`")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "f16x2"], default="f32",
                    help="f32: every gate product in fp32 (default); bf16x3 / f16x2: each fp32 product rebuilt from six "
                         "bf16 / three fp16 piece products on the 16-bit matrix pipe (include/snn_hip.h)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start one fresh process per GPU BEFORE anything here touches the GPU
        from supernnova_amd.distributed import self_launch
        raise SystemExit(self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SNN_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices)
    backend = os.environ.get("SNN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    out = run(args, rank, world, dev, backend, dist if world > 1 else None)   # (`dist` is bound above only when world > 1)
    default_run = (args.workload == "des" and args.precision == "f32" and world == 1 and args.curves == 1_250_000 and
                   not os.environ.get("SNN_BENCH_NOCHECK") and not args.no_secondary)
    if default_run:
        # the other single-GPU configurations of BASELINE.json (configs[1]: VariationalRNN MC dropout; configs[4]: the
        # ELAsTiCC-like H = 128 / 300-step model), two timed steps each, so that the line the driver records carries them
        import copy
        out["secondary"] = {}
        for wl in ("variational", "elasticc", "h64", "uni", "gru", "rnn"):
            a2 = copy.copy(args)
            a2.workload, a2.steps, a2.warmup, a2.cpu_curves = wl, 2, 1, 0
            if wl == "h64":
                a2.curves = args.curves // 4      # (an H=64 pass over 1.25 M curves is ~3 s; a quarter keeps the default run short)
            a2.no_fast_path = a2.no_end_to_end = a2.no_selfcheck = True
            a2.cpu_full = False
            sub = run(a2, rank, world, dev, backend, None)
            out["secondary"][wl] = {k: sub[k] for k in ("value", "unit", "ms_per_step", "steps", "roofline")}
            out["secondary"][wl]["workload"] = sub["config"]["workload"]
    # BASELINE configs[3], BayesianRNN training (fwd with tape + CE + BPTT + BBB/KL backward + fused Adam), in the line
    # the driver records: single rank -> secondary.train at the reference's own batch size (128, conf.py:517) and at a
    # batch that fills the chip; N ranks -> train_dp, data parallel over RCCL with ONE all-reduce of the flat gradient
    # per step, at a global batch of 128 (the reference's step split N ways: 128 / N curves per rank) and of 128 per rank.
    if not args.no_train and not os.environ.get("SNN_BENCH_NOCHECK") and args.workload == "des" and args.precision == "f32":
        import bench_train
        torch.cuda.empty_cache()

        def train_obj(per_rank, steps, warmup, **over):
            a = bench_train.parser().parse_args([])
            a.gpus, a.batch, a.steps, a.warmup, a.cpu_steps = world, per_rank, steps, warmup, 0
            for k, v in over.items():
                setattr(a, k, v)
            r = bench_train.measure(a, rank, world, dev, backend)
            if r is None:
                return None
            o = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "n_gpus")}
            o["global_batch"] = r["config"]["global_batch"]
            o["workload"] = r["config"]["workload"]
            rf = r.get("roofline")
            if rf:
                o["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "flop_per_step",
                                                    "isolated_step_ms", "phases_ms", "forward", "backward",
                                                    "forward_kernel", "backward_path")}
            return o

        if world == 1 and default_run:
            tr = {"b128": train_obj(128, 200, 30), "b65536": train_obj(65536, 10, 3)}
            # the wider models of BASELINE.json (configs[2]: H = 64; configs[4]: H = 128, D = 78, C = 14, 300-step curves)
            # at chip-filling batches: split backward (snn_train_big.hip, snn_bwd_dx.hip), two-pass forward at H = 128
            torch.cuda.empty_cache()
            tr["h64_b65536"] = train_obj(65536, 6, 2, hidden=64)
            torch.cuda.empty_cache()
            tr["h128_t300_b8192"] = train_obj(8192, 4, 2, hidden=128, features=78, classes=14, max_len=300, full_length=True)
            out.setdefault("secondary", {})["train"] = tr
        elif world > 1:
            per = max(1, 128 // world)
            tr = {f"global_b{per * world}": train_obj(per, 200, 30), f"global_b{128 * world}": train_obj(128, 200, 30)}
            if rank == 0:
                for o in tr.values():
                    o["backend"] = "rccl (torch.distributed nccl)" if backend == "nccl" else backend
                    o["allreduce_bytes"] = 4 * (2 * E_param_count() - 2)
                    o["allreduce_plus_adam_ms"] = (o.get("roofline") or {}).get("phases_ms", {}).get("allreduce+adam")
                out["train_dp"] = tr
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def E_param_count():
    from supernnova_amd import engine as E
    return E.Config(24, 32, 2, True, 2, "mean").P


if __name__ == "__main__":
    main()
