"""nn.Module mirror of the reference's three RNN classes over the HIP kernels.

Same constructor (`cls(input_size, settings)`), same `forward(x,
mean_field_inference=False)` contract (x a PackedSequence or a dense (T,B,D)
tensor, returns logits (B,C)), same state_dict key names/shapes and the same
parameter initialisation (same torch RNG calls in the same order), so reference
checkpoints load and a seeded construction gives identical weights:
  VanillaRNN      training/vanilla_rnn.py:4-90
  VariationalRNN  training/variational_rnn.py:4-244
  BayesianRNN     training/bayesian_rnn.py:21-139, 146-270, 574-616, 753-775
The modules hold parameters only; every FLOP of forward runs in libsnn_hip.so.
CPU tensors are rejected (no fallback).

Random draws.  `draws = "philox"` (default): masks / eps come from the in-kernel
Philox4x32-10 stream keyed by a seed taken from torch's CPU generator at each
forward (so torch.manual_seed makes runs reproducible).  `draws = "host"` (parity
mode): the draws are made on the host with exactly the reference's torch calls in
the reference's order and shipped to the kernels, so a run seeded like
a reference CPU run reproduces its outputs to fp32 rounding.
"""
import math

import numpy as np
import torch
from torch import nn
from torch.nn.utils.rnn import PackedSequence, pad_packed_sequence

from . import _lib as L
from . import engine as E


def _dirs(bidirectional):
    return 2 if bidirectional else 1


_GATES = {"lstm": 4, "gru": 3, "rnn": 1}     # gate blocks of torch.nn.LSTM / GRU / RNN parameters


class _LSTMParams(nn.Module):
    """Parameter container with torch.nn.LSTM's (layer_type gru / rnn: torch.nn.GRU's / torch.nn.RNN's) names,
    shapes, registration order and reset_parameters() (U(-1/sqrt(H), 1/sqrt(H)) over parameters in order)."""

    def __init__(self, input_size, hidden_size, num_layers, bidirectional, layer_type="lstm"):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.bidirectional = num_layers, bidirectional
        dirs = _dirs(bidirectional)
        G = _GATES[layer_type] * hidden_size
        for l in range(num_layers):
            I = input_size if l == 0 else hidden_size * dirs
            for d in range(dirs):
                s = "_reverse" if d else ""
                for name, shape in ((f"weight_ih_l{l}{s}", (G, I)), (f"weight_hh_l{l}{s}", (G, hidden_size)),
                                    (f"bias_ih_l{l}{s}", (G,)), (f"bias_hh_l{l}{s}", (G,))):
                    self.register_parameter(name, nn.Parameter(torch.empty(shape)))
        stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)


class _WeightDropParams(nn.Module):
    """`rnn_layer{i}` of VariationalRNN: holds `.module` whose weight_hh_l0 was re-registered
    as weight_hh_l0_raw (variational_rnn.py:216-222), keeping the reference's key order."""

    def __init__(self, input_size, hidden_size, bidirectional):
        super().__init__()
        m = _LSTMParams(input_size, hidden_size, 1, bidirectional)
        w = m._parameters.pop("weight_hh_l0")
        m.register_parameter("weight_hh_l0_raw", nn.Parameter(w.data))
        self.module = m


def unpack_packed_sequence(x):
    """What pad_packed_sequence (vanilla_rnn.py:75-76) is used for, without the padded copy: a PackedSequence's rows
    stay where they are (engine.PackedRows: the data tensor, the offset of every time step, the column of every curve)
    and snn_pack_x_steps reads them straight into the tile image.  Returns (PackedRows, lengths int32 numpy (B,) in the
    original batch order -- the lengths pad_packed_sequence returns)."""
    bs = x.batch_sizes.numpy().astype(np.int64)
    T, B, D = bs.size, int(bs[0]), x.data.size(-1)
    lengths = np.searchsorted(-bs, -np.arange(B, dtype=np.int64), side="left").astype(np.int32)   # steps with bs > b
    dev = x.data.device
    col_off = None
    if x.sorted_indices is not None:
        si = x.sorted_indices.cpu().numpy()
        lens = np.empty_like(lengths)
        lens[si] = lengths
        lengths = lens
        col_off = x.unsorted_indices.to(device=dev, dtype=torch.int64) * D
    step_h = torch.empty(T, dtype=torch.int64, pin_memory=x.data.is_cuda)      # pinned: the upload does not block the host
    so = step_h.numpy()
    so[0] = 0
    np.cumsum(bs[:-1] * D, out=so[1:])
    return E.PackedRows(x.data.float().contiguous(), step_h.to(dev, non_blocking=True), col_off, B), lengths


def _check_settings(settings, layer_types=("lstm",)):
    if settings.layer_type not in layer_types:
        raise ValueError(f"supernnova_amd implements layer_type {' / '.join(layer_types)} for this model, "
                         f"got {settings.layer_type!r}")
    if settings.rnn_output_option not in ("mean", "standard"):
        raise ValueError(f"unknown rnn_output_option {settings.rnn_output_option!r}")


class _HipRNN(nn.Module):
    """Shared host logic: input unpacking, plan/pack, launch."""

    draws = "philox"
    # arithmetic of the gate products in eval-mode forwards without dropout masks (VanillaRNN, BayesianRNN):
    # "f32" (exact fp32, default), "bf16x3" (six bf16 piece products per fp32 product) or "f16x2" (three fp16 piece
    # products; include/snn_hip.h)
    inference_precision = "f32"

    def infer_cfg(self, masked=False):
        """Engine config of an inference launch: the module's cfg, in `inference_precision` when that mode has a
        kernel for this shape and the launch carries no masks."""
        if masked or self.inference_precision == "f32":
            return self.cfg      # (VariationalRNN overrides this: its f16x2 kernel takes masks)
        cfg = self.cfg.with_precision(self.inference_precision)
        return cfg if cfg.supported() else self.cfg

    def _init_common(self, input_size, settings, layer_types=("lstm",)):
        _check_settings(settings, layer_types)
        self.input_size = input_size
        self.layer_type = settings.layer_type
        self.output_size = settings.nb_classes
        self.hidden_size = settings.hidden_dim
        self.num_layers = settings.num_layers
        self.bidirectional = bool(settings.bidirectional)
        self.use_cuda = settings.use_cuda
        self.rnn_output_option = settings.rnn_output_option
        self.cfg = E.Config(input_size, self.hidden_size, self.num_layers, self.bidirectional,
                            self.output_size, self.rnn_output_option, cell=self.layer_type)
        self._pflat = None
        self.dp_curve_base = 0   # data-parallel: global index of this rank's first curve (per-curve Philox masks)

    @property
    def last_input_size(self):
        f = self.hidden_size * _dirs(self.bidirectional)
        return f if self.rnn_output_option == "mean" else f * self.num_layers

    # -- parameters live as views of ONE flat fp32 buffer in kernel-canonical order, so the kernels,
    #    the fused Adam and the data-parallel all-reduce all see a single contiguous array.
    def _flat_groups(self):
        """[(tag, [state_dict keys in canonical order])]; concatenated they define the flat buffer."""
        raise NotImplementedError

    def _flat_params(self):
        # the Parameter objects are fixed after construction (.to() / load_state_dict re-point or fill their .data,
        # which flat_parameters() checks); walking named_parameters() on every call cost 65 us per training step
        cache = self.__dict__.get("_flat_params_cache")
        if cache is None:
            params = dict(self.named_parameters())
            cache = [params[k] for _, keys in self._flat_groups() for k in keys]
            self.__dict__["_flat_params_cache"] = cache
        return cache

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_flat_params_cache", None)   # conversions may replace Parameter objects
        self.__dict__.pop("_flat_checked", None)
        self.__dict__.pop("_train_state", None)
        return super()._apply(fn, *args, **kwargs)

    def _flatten(self):
        ps = self._flat_params()
        dev = ps[0].device
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1).float() for p in ps]).contiguous()
            off = 0
            for p in ps:
                n = p.numel()
                p.data = flat[off:off + n].view(p.shape)
                off += n
        self._pflat = flat
        self._slices, off = {}, 0
        params = dict(self.named_parameters())
        for tag, keys in self._flat_groups():
            n = sum(params[k].numel() for k in keys)
            self._slices[tag] = (off, off + n)
            off += n
        assert off == flat.numel() and dev == flat.device

    def flat_parameters(self):
        """The flat buffer all parameters are views of (re-flattened if something re-pointed them)."""
        flat = self.__dict__.get("_pflat")
        ok = flat is not None
        if ok:
            # every parameter must still be the view of `flat` that _flatten made it.  Conversions (.to(), .cuda(), ...)
            # go through _apply, which drops `_flat_checked`; between two of them the first and the last parameter are
            # compared (a training step calls this several times: walking all 35 parameters cost 17 us per call)
            ps = self._flat_params()
            if self.__dict__.get("_flat_checked"):
                ok = ps[0].data_ptr() == flat.data_ptr() and \
                    ps[-1].data_ptr() == flat.data_ptr() + (flat.numel() - ps[-1].numel()) * flat.element_size()
            else:
                ptr, esz = flat.data_ptr(), flat.element_size()
                for p in ps:
                    if p.data_ptr() != ptr or p.device != flat.device:
                        ok = False
                        break
                    ptr += p.numel() * esz
                if ok:
                    self.__dict__["_flat_checked"] = True
        if not ok:
            self._flatten()
            self.__dict__["_flat_checked"] = True
        return self._pflat

    def _flat(self, tag):
        flat = self.flat_parameters()
        a, b = self._slices[tag]
        return flat[a:b]

    # -- x -> (padded (T,B,D) on device, host lengths)
    def _unpack(self, x):
        if isinstance(x, PackedSequence):
            if x.data.dim() != 2:
                raise RuntimeError(f"input must have 2 dimensions, got {x.data.dim()}")
            if x.data.size(-1) != self.input_size:
                raise RuntimeError("input.size(-1) must be equal to input_size. Expected {}, got {}".format(
                    self.input_size, x.data.size(-1)))
            if not x.data.is_cuda:
                raise L.SnnError("supernnova_amd modules run on the GPU only: move the module and its input "
                                 "to a HIP device (there is no CPU fallback)")
            return unpack_packed_sequence(x)
        else:
            if x.dim() != 3:
                raise RuntimeError(f"input must have 3 dimensions, got {x.dim()}")
            if x.size(-1) != self.input_size:
                raise RuntimeError("input.size(-1) must be equal to input_size. Expected {}, got {}".format(
                    self.input_size, x.size(-1)))
            padded = x
            lengths = np.full(x.size(1), x.size(0), np.int32)  # dense input runs over padding (vanilla_rnn.py:82-83)
        if not padded.is_cuda:
            raise L.SnnError("supernnova_amd modules run on the GPU only: move the module and its input "
                             "to a HIP device (there is no CPU fallback)")
        return padded.float().contiguous(), lengths

    def _seed(self):
        return int(torch.empty((), dtype=torch.int64).random_().item())

    def _use_train_path(self):
        return self.training and torch.is_grad_enabled()

    def _train_forward(self, x, masks_fn, eps=None, mean_field_inference=False):
        """Differentiable forward (autograd.Function over the HIP train/backward kernels)."""
        padded, lengths = self._unpack(x)
        if not self.cfg.supported():
            raise L.SnnError(f"no HIP kernel for this configuration: {self.cfg}")
        self.flat_parameters()
        return _TrainFn.apply(self, padded, lengths, masks_fn, eps, mean_field_inference, *self._flat_params())

    def fused_train_step(self, x, target, optimizer, kl_scale=0.0, eps=None, masks=None, global_batch=None):
        """One training step without autograd (utils/training_utils.py:508-525 in five launches):
        forward-train, CE + dlogits, BPTT, (BBB backward), fused Adam.  Returns the loss (device scalar).
        Data parallel (torch.distributed initialised, `global_batch` = curves of ALL ranks in this step): every curve's
        CE gradient is weighted 1 / global_batch and the rank-identical KL gradient 1 / world, and the ranks' flat
        gradients are SUMMED -- the single-process gradient for any split of the batch, equal shards or not."""
        tl = self.__dict__.get("_timeline")      # measurement hook (bench_train.py): [(phase name, event)] per step
        if tl is None and masks is None:
            return self._train_step_one_call(x, target, optimizer, kl_scale, eps, global_batch)
        padded, lengths = self._unpack(x)
        if not self.cfg.supported():
            raise L.SnnError(f"no HIP kernel for this configuration: {self.cfg}")
        cfg, dev = self.cfg, padded.device
        self.flat_parameters()

        def mark(name):
            if tl is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                tl.append((name, ev))
        mark("start")
        plan = E.Plan(cfg, lengths, dev)
        xp = E.pack_x(cfg, plan, padded)
        mark("plan+pack")
        bayes = isinstance(self, BayesianRNN)
        if masks is None and not bayes:
            masks = self._masks(plan) if isinstance(self, VanillaRNN) else \
                self._masks(1, plan.n_curves, dev, False)
        if bayes:
            mu, rho = self._mu_rho()
            if eps is None and self.draws == "host":
                eps = self.host_eps(1, dev)
            seed = 0 if eps is not None else self._seed()
            if eps is not None:
                eps = eps.reshape(-1).contiguous()
            wk, kl, terms, mode = E.bbb_sample_train(cfg, mu, rho, self._prior, eps=eps, seed=seed)
        else:
            wk = E.prepare_weights(cfg, self._flat("w"))
        mark("sample")
        logits, tape = E.lstm_forward_train(cfg, plan, xp, wk, masks)
        mark("forward")
        from .distributed import world_size
        world = world_size(getattr(optimizer, "process_group", None))
        dp = world > 1 and global_batch is not None
        if dp and not (plan.n_curves <= global_batch):
            raise ValueError(f"fused_train_step: this rank holds {plan.n_curves} curves but global_batch is {global_batch}")
        ce, dlogits = E.ce_loss_grad(logits, target, grad_scale=1.0 / global_batch if dp else None)
        mark("ce")
        dw = E.lstm_backward(cfg, plan, xp, wk, tape, dlogits, masks)
        mark("backward")
        if bayes:
            P, Ps = cfg.P, cfg.P - cfg.nb_classes
            gflat = torch.empty(P + Ps, dtype=torch.float32, device=dev)
            E.bbb_backward(cfg, mu, rho, self._prior, terms, dw, mode, eps=eps, seed=seed,
                           kl_scale=kl_scale / world if dp else kl_scale, dmu=gflat[:P], drho=gflat[P:])
            self.kl = kl
            loss = ce + kl * kl_scale
        else:
            gflat, loss = dw, ce
        mark("bbb_backward")
        optimizer.step_flat(gflat, mean=not dp)
        mark("allreduce+adam")
        self.last_logits = logits
        return loss

    def _train_step_one_call(self, x, target, optimizer, kl_scale, eps, global_batch):
        """fused_train_step through snn_train_step_grad: the same launches as the call-by-call sequence above, enqueued
        by ONE call of the C ABI into buffers that are kept between steps (grown on demand; every consumer is ordered
        behind its producer on the current stream).  What is allocated per step is what the caller may keep: the
        logits, and one small tensor holding the loss and the KL."""
        import ctypes
        import os
        from .distributed import world_size
        padded, lengths = self._unpack(x)
        cfg = self.cfg
        dev = padded.device
        st = self.__dict__.get("_train_state")
        if st is None or st["dev"] != dev:
            if not cfg.supported():
                raise L.SnnError(f"no HIP kernel for this configuration: {cfg}")
            st = {"dev": dev, "cfgc": cfg.c, "P": cfg.P, "PK": cfg.PK, "C": cfg.nb_classes, "nt": cfg.n_tensors,
                  "xsf": cfg.x_step_floats, "io": L.SnnTrainIO(), "sizes": {}, "bufs": {},
                  "ktf": cfg.kl_terms_floats(1) if isinstance(self, BayesianRNN) else 0,
                  "prior": self._prior.c if isinstance(self, BayesianRNN) else None}
            self.__dict__["_train_state"] = st
        flat = self.flat_parameters()
        bayes = st["prior"] is not None
        lib, io, P, C = L.load(), st["io"], st["P"], st["C"]
        plan = E.Plan(cfg, lengths, dev)
        N = plan.n_curves
        if target.dtype != torch.int64 or not target.is_cuda or target.numel() != N:
            raise L.SnnError("target must be a device int64 tensor with one entry per curve")
        target = target.contiguous()
        key = (plan.n_tiles, plan.max_T, os.environ.get("SNN_BWD_PATH"), os.environ.get("SNN_FWD_SPLIT"))
        sz = st["sizes"].get(key)
        if sz is None:
            cref = ctypes.byref(st["cfgc"])
            sz = (L.check(lib.snn_train_tape_floats(cref, plan.n_tiles, plan.max_T), "snn_train_tape_floats"),
                  L.check(lib.snn_lstm_forward_workspace_bytes(cref, plan.max_T), "snn_lstm_forward_workspace_bytes"),
                  L.check(lib.snn_lstm_backward_workspace_bytes(cref, plan.n_tiles, plan.max_T), "snn_lstm_backward_workspace_bytes"))
            st["sizes"][key] = sz
        tape_floats, fwd_bytes, bwd_bytes = sz

        def buf(name, n, dtype=torch.float32):
            t = st["bufs"].get(name)
            if t is None or t.numel() < n:
                t = torch.empty(int(n * 1.25) + 64, dtype=dtype, device=dev)
                st["bufs"][name] = t
            return t
        xp_floats = plan.total_steps * st["xsf"]
        n_part = lib.snn_ce_loss_partials(N)
        poff = (4 + st["ktf"] + 1) & ~1            # fp64 partials of the CE sum: 8-byte aligned
        small = torch.empty(poff + 2 * n_part, dtype=torch.float32, device=dev)   # [CE, loss, KL, -, kl_terms..., partials...]
        logits = torch.empty((N, C), dtype=torch.float32, device=dev)
        xp, wk, tape = buf("xp", xp_floats), buf("wk", st["PK"]), buf("tape", tape_floats)
        dlogits, dw = buf("dlogits", N * C), buf("dw", P)
        fwd_ws, bwd_ws = buf("fwd_ws", fwd_bytes, torch.uint8), buf("bwd_ws", bwd_bytes, torch.uint8)
        world = world_size(getattr(optimizer, "process_group", None))
        dp = world > 1 and global_batch is not None
        if dp and not (N <= global_batch):
            raise ValueError(f"fused_train_step: this rank holds {N} curves but global_batch is {global_batch}")
        io.x, io.step_off, io.stride_t, io.stride_n = padded.data.data_ptr(), None, 0, cfg.input_size
        if isinstance(padded, E.PackedRows):
            io.step_off = padded.step_off.data_ptr()
            io.curve_off = None if padded.col_off is None else padded.col_off.data_ptr()
        else:      # dense (T, B, D)
            io.x, io.curve_off = padded.data_ptr(), None
            io.stride_t, io.stride_n = N * cfg.input_size, cfg.input_size
        io.order, io.len_sorted = plan.d_order.data_ptr(), plan.d_len_sorted.data_ptr()
        io.tile_T, io.tile_off = plan.d_tile_T.data_ptr(), plan.d_tile_off.data_ptr()
        io.n_curves, io.n_tiles, io.max_T, io.xp_floats = N, plan.n_tiles, plan.max_T, xp_floats
        io.target = target.data_ptr()
        mc = None
        if bayes:
            (a0, a1), (b0, b1) = self._slices["mu"], self._slices["rho"]
            mu, rho = flat[a0:a1], flat[b0:b1]
            if eps is None and self.draws == "host":
                eps = self.host_eps(1, dev)
            if eps is not None:
                eps = eps.reshape(-1).contiguous()
                if eps.numel() != P - C or not eps.is_cuda:
                    raise L.SnnError(f"eps must hold {P - C} floats on the GPU")
            io.params, io.rho = mu.data_ptr(), rho.data_ptr()
            io.eps = None if eps is None else eps.data_ptr()
            io.bbb_mode = L.BBB_EPS if eps is not None else L.BBB_PHILOX
            io.seed, io.offset = (0 if eps is not None else self._seed()), 0
            io.prior = ctypes.pointer(st["prior"])
            io.kl, io.kl_terms = small[2:].data_ptr(), small[4:].data_ptr()
            gflat = buf("grad", 2 * P - C)
            io.grad = gflat.data_ptr()
            io.masks = None
        else:
            io.params, io.rho, io.eps = flat.data_ptr(), None, None
            masks = self._masks(plan) if isinstance(self, VanillaRNN) else self._masks(1, N, dev, False)
            mc = masks.c() if masks is not None and masks.mode != L.SNN_MASK_NONE else None
            io.masks = ctypes.pointer(mc) if mc is not None else None
            gflat, io.grad = dw, None
        io.ce_grad_scale = 1.0 / global_batch if dp else 1.0 / N
        io.kl_scale = kl_scale / world if dp else kl_scale
        io.loss_kl_scale = kl_scale
        io.xp, io.wk, io.logits, io.tape, io.tape_floats = xp.data_ptr(), wk.data_ptr(), logits.data_ptr(), tape.data_ptr(), tape.numel()
        io.dlogits, io.loss, io.dw = dlogits.data_ptr(), small.data_ptr(), dw.data_ptr()
        io.ce_partials = small[poff:].data_ptr() if n_part else None
        io.fwd_ws, io.fwd_ws_bytes, io.bwd_ws, io.bwd_ws_bytes = fwd_ws.data_ptr(), fwd_ws.numel(), bwd_ws.data_ptr(), bwd_ws.numel()
        L.check(lib.snn_train_step_grad(ctypes.byref(st["cfgc"]), ctypes.byref(io), E._stream()), "snn_train_step_grad")
        n_grad = (2 * P - C) if bayes else P
        optimizer.step_flat(gflat[:n_grad], mean=not dp)
        self.__dict__["last_logits"] = logits          # (plain attributes: nn.Module.__setattr__ costs 4 us apiece)
        if bayes:
            self.__dict__["kl"] = small[2]
            return small[1]
        return small[0]

    def _launch(self, x, wk, n_samples, masks=None, want_probs=False):
        padded, lengths = self._unpack(x)
        if not self.cfg.supported():
            raise L.SnnError(f"no HIP kernel for this configuration: {self.cfg}")
        cfg = self.infer_cfg(masks is not None)
        plan = E.Plan(cfg, lengths, padded.device)
        if callable(masks):          # masks whose shape depends on the plan (per-step between-layer dropout)
            masks = masks(plan)
        xp = E.pack_x(cfg, plan, padded)
        return E.lstm_forward(cfg, plan, xp, wk, n_samples, masks=masks, want_logits=True,
                              want_probs=want_probs, sample_major=True), plan


class VanillaRNN(_HipRNN):
    def __init__(self, input_size, settings):
        super().__init__()
        # layer_type gru / rnn (vanilla_rnn.py:26-32: getattr(torch.nn, layer_type.upper())): fp32 kernels, H = 32 / 64
        self._init_common(input_size, settings, ("lstm", "gru", "rnn"))
        self.dropout = settings.dropout
        self.rnn_layer = _LSTMParams(input_size, self.hidden_size, self.num_layers, self.bidirectional, self.layer_type)
        self.output_dropout_layer = nn.Dropout(self.dropout)
        self.output_layer = nn.Linear(self.last_input_size, self.output_size)

    def _flat_groups(self):
        from .layout import vanilla_keys
        return [("w", vanilla_keys(self.num_layers, _dirs(self.bidirectional)))]

    def _masks(self, plan=None, *_):
        """train mode only: torch.nn.LSTM's per-step inter-layer dropout (vanilla_rnn.py:26-32) and
        nn.Dropout on the pooled features (:33, :86), drawn in-kernel (Philox)."""
        if not self.training or self.dropout <= 0:
            return None
        which = (4 if self.num_layers > 1 else 0) | 8
        return E.Masks(mode=L.SNN_MASK_PHILOX, which=which, dropout=self.dropout, seed=self._seed(),
                       mid_T=plan.max_T if plan is not None else 1, curve_base=self.dp_curve_base)

    def forward(self, x, mean_field_inference=False, masks=None):
        if self._use_train_path():
            fn = (lambda plan: masks) if masks is not None else self._masks
            logits, _ = self._train_forward(x, fn)
            return logits
        if masks is None and self.training and self.dropout > 0:
            # train mode under torch.no_grad(): the reference still applies nn.LSTM's and nn.Dropout's dropout
            # (vanilla_rnn.py:26-33) -- grad mode only decides whether a tape is kept, not whether masks are drawn
            masks = self._masks
            if self.layer_type != "lstm":      # GRU / tanh RNN take masks in their training kernel only: run it, drop the tape
                with torch.no_grad():
                    logits, _ = self._train_forward(x, masks)
                return logits
        wk = E.prepare_weights(self.infer_cfg(masks is not None), self._flat("w"))
        (logits, _), _ = self._launch(x, wk, 1, masks)
        return logits[0]

    def mc_weights(self, n_samples, n_curves, device, mean_field_inference=False):
        """(kernel-layout weights, masks) for n_samples forwards over n_curves curves."""
        return E.prepare_weights(self.infer_cfg(), self._flat("w")), None

    def forward_mc(self, x, n_samples=1, mean_field_inference=False):
        """probs (B, n_samples, C); a vanilla model is deterministic in eval mode."""
        wk = E.prepare_weights(self.infer_cfg(), self._flat("w"))
        (_, probs), _ = self._launch(x, wk, n_samples, None, want_probs=True)
        return probs.permute(1, 0, 2)


class VariationalRNN(_HipRNN):
    def __init__(self, input_size, settings):
        super().__init__()
        self._init_common(input_size, settings)
        self.dropout = settings.dropout
        dirs = _dirs(self.bidirectional)
        for i in range(self.num_layers):
            I = input_size if i == 0 else self.hidden_size * dirs
            setattr(self, f"rnn_layer{i}", _WeightDropParams(I, self.hidden_size, self.bidirectional))
        self.output_layer = nn.Linear(self.last_input_size, self.output_size)

    def _flat_groups(self):
        from .layout import variational_keys
        return [("w", variational_keys(self.num_layers, _dirs(self.bidirectional)))]

    def host_masks(self, n_samples, n_curves, device):
        """Draw the masks on the host with the reference's calls in the reference's order
        (input (1,B,D) -> per layer [W_hh rows (4H,1) -> locked (1,B,2H)] -> output (B,F))."""
        p, H, Ln, D = self.dropout, self.hidden_size, self.num_layers, self.input_size
        dH, F = _dirs(self.bidirectional) * H, self.last_input_size

        def bern(*shape):
            return torch.empty(shape).bernoulli_(1 - p).div_(1 - p)

        m_in, m_whh, m_mid, m_out = [], [], [], []
        for _ in range(n_samples):
            m_in.append(bern(1, n_curves, D)[0])
            whh, mid = [], []
            for l in range(Ln):
                whh.append(bern(4 * H, 1)[:, 0])
                if l != Ln - 1:
                    mid.append(bern(1, n_curves, dH))
            m_whh.append(torch.stack(whh))
            m_mid.append(torch.stack(mid) if mid else torch.empty(0, 1, n_curves, dH))
            m_out.append(bern(n_curves, F))
        to = lambda ts: torch.stack(ts).to(device).contiguous()
        return E.Masks(mode=L.SNN_MASK_ARRAYS, m_in=to(m_in), m_whh=to(m_whh),
                       m_mid=to(m_mid) if Ln > 1 else None, m_out=to(m_out), mid_T=1)

    def _masks(self, n_samples, n_curves, device, mean_field_inference):
        if mean_field_inference or self.dropout <= 0:
            return None
        if self.draws == "host":
            return self.host_masks(n_samples, n_curves, device)
        return E.Masks(mode=L.SNN_MASK_PHILOX, which=0xF, dropout=self.dropout, seed=self._seed(),
                       curve_base=self.dp_curve_base)

    def _n_curves(self, x):
        return int(x.batch_sizes[0]) if isinstance(x, PackedSequence) else x.size(1)

    def infer_cfg(self, masked=False):
        """fp32 unless `inference_precision == "f16x2"` and that mode has a kernel for this launch (masked launches:
        input width <= 48, i.e. up to 5 filters; see snn_forward_h2.hip)."""
        if self.inference_precision != "f16x2":
            return self.cfg
        cfg = self.cfg.with_precision("f16x2")
        if not cfg.supported() or (masked and self.input_size > 48):
            return self.cfg
        return cfg

    def _infer_weights(self, masks):
        """Kernel-layout weights of an inference launch.  The f16x2 kernel applies the dropout BITS exactly (ANDs on
        the fp16 pieces) and expects the masks' 1/(1-p) factor inside the weights that multiply a masked operand
        (include/snn_hip.h): W_ih of layer 0 (input mask), W_hh of the forward directions (W_hh row mask), W_ih of
        the layers above (between-layer mask)."""
        cfg = self.infer_cfg(masks is not None)
        flat = self._flat("w")
        if masks is not None and cfg.precision == "f16x2":
            arrays = masks.mode == L.SNN_MASK_ARRAYS
            on = [(masks.m_in is not None) if arrays else bool(masks.which & 1),
                  (masks.m_whh is not None) if arrays else bool(masks.which & 2),
                  (masks.m_mid is not None) if arrays else bool(masks.which & 4)]
            scale = float(np.float32(1.0) / np.float32(1.0 - masks.dropout if not arrays else 1.0 - self.dropout))
            flat = flat.clone()
            H, dirs, off = self.hidden_size, _dirs(self.bidirectional), 0
            for l in range(self.num_layers):
                I = self.input_size if l == 0 else H * dirs
                for d in range(dirs):
                    if on[0] if l == 0 else on[2]:
                        flat[off:off + 4 * H * I] *= scale
                    off += 4 * H * I
                    if d == 0 and on[1]:
                        flat[off:off + 4 * H * H] *= scale
                    off += 4 * H * H + 8 * H
        return E.prepare_weights(cfg, flat)

    def mc_weights(self, n_samples, n_curves, device, mean_field_inference=False):
        """(kernel-layout weights, masks) for n_samples MC-dropout forwards over n_curves curves."""
        masks = self._masks(n_samples, n_curves, device, mean_field_inference)
        return self._infer_weights(masks), masks

    def forward(self, x, mean_field_inference=False, masks=None):
        dev = x.data.device if isinstance(x, PackedSequence) else x.device
        if masks is None:
            masks = self._masks(1, self._n_curves(x), dev, mean_field_inference)
        if self._use_train_path():
            logits, _ = self._train_forward(x, lambda plan: masks)
            return logits
        (logits, _), _ = self._launch(x, self._infer_weights(masks), 1, masks)
        return logits[0]

    def forward_mc(self, x, n_samples, mean_field_inference=False, masks=None):
        """All n_samples MC-dropout forwards in one launch -> probs (B, n_samples, C)."""
        dev = x.data.device if isinstance(x, PackedSequence) else x.device
        if masks is None:
            masks = self._masks(n_samples, self._n_curves(x), dev, mean_field_inference)
        (_, probs), _ = self._launch(x, self._infer_weights(masks), n_samples, masks, want_probs=True)
        return probs.permute(1, 0, 2)


class Prior(object):
    """training/bayesian_rnn.py:753-763."""

    def __init__(self, pi=0.25, log_sigma1=-1.0, log_sigma2=-7.0):
        self.pi_mixture = pi
        self.log_sigma1, self.log_sigma2 = log_sigma1, log_sigma2
        self.sigma1, self.sigma2 = math.exp(log_sigma1), math.exp(log_sigma2)
        self.sigma_mix = math.sqrt(pi * self.sigma1 ** 2 + (1.0 - pi) * self.sigma2 ** 2)


def _bbb_pair(shape, mu_lower, mu_upper, rho_lower, rho_upper):
    """get_bbb_variable (bayesian_rnn.py:766-775): mu then rho, uniform."""
    mu, rho = nn.Parameter(torch.empty(*shape)), nn.Parameter(torch.empty(*shape))
    mu.data.uniform_(mu_lower, mu_upper)
    rho.data.uniform_(rho_lower, rho_upper)
    return mu, rho


class _BayesLSTMParams(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, bidirectional, rho_lower, rho_upper):
        super().__init__()
        dirs = _dirs(bidirectional)
        G = 4 * hidden_size
        for l in range(num_layers):
            for d in range(dirs):
                I = input_size if l == 0 else hidden_size * dirs
                s = "_reverse" if d else ""
                pairs = [_bbb_pair(shp, -0.05, 0.05, rho_lower, rho_upper)
                         for shp in ((G, I), (G, hidden_size), (G,), (G,))]
                for n, (mu, rho) in zip(("weight_ih", "weight_hh", "bias_ih", "bias_hh"), pairs):
                    self.register_parameter(f"{n}_mu_l{l}{s}", mu)
                    self.register_parameter(f"{n}_rho_l{l}{s}", rho)


class _BayesLinearParams(nn.Module):
    def __init__(self, in_features, out_features, rho_lower, rho_upper):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.mu, self.rho = _bbb_pair((out_features, in_features), -0.05, 0.05, rho_lower, rho_upper)
        self.bias = nn.Parameter(torch.zeros(out_features))


class BayesianRNN(_HipRNN):
    def __init__(self, input_size, settings):
        super().__init__()
        self._init_common(input_size, settings)
        self.prior = Prior(settings.pi, settings.log_sigma1, settings.log_sigma2)
        self.prior_output = Prior(settings.pi, settings.log_sigma1_output, settings.log_sigma2_output)
        rl = lambda prior, scale: math.log(math.exp(prior.sigma_mix / scale) - 1.0)
        self.rnn_layer = _BayesLSTMParams(input_size, self.hidden_size, self.num_layers, self.bidirectional,
                                          rl(self.prior, settings.rho_scale_lower),
                                          rl(self.prior, settings.rho_scale_upper))
        self.output_layer = _BayesLinearParams(self.last_input_size, self.output_size,
                                               rl(self.prior_output, settings.rho_scale_lower_output),
                                               rl(self.prior_output, settings.rho_scale_upper_output))
        self._prior = E.Prior(settings.pi, settings.log_sigma1, settings.log_sigma2,
                              settings.log_sigma1_output, settings.log_sigma2_output)
        self.kl = None
        self._draw_count = 0

    def _flat_groups(self):
        from .layout import bayesian_keys
        dirs = _dirs(self.bidirectional)
        return [("mu", bayesian_keys(self.num_layers, dirs, "mu")), ("rho", bayesian_keys(self.num_layers, dirs, "rho"))]

    def _mu_rho(self):
        return self._flat("mu"), self._flat("rho")

    def host_eps(self, n_samples, device):
        """eps drawn like the reference (bayesian_rnn.py:492-523, 608): one CPU normal_ per tensor,
        in the order w_ih, w_hh, b_ih, b_hh per (layer, direction), then the head."""
        from .layout import bayesian_keys
        params = dict(self.named_parameters())
        keys = bayesian_keys(self.num_layers, _dirs(self.bidirectional), "rho")
        rows = [torch.cat([torch.empty(params[k].shape).normal_(0.0, 1.0).reshape(-1) for k in keys])
                for _ in range(n_samples)]
        return torch.stack(rows).to(device).contiguous()

    def _sample(self, n_samples, mean_field_inference, eps, device):
        mu, rho = self._mu_rho()
        if not mean_field_inference and eps is None and self.draws == "host":
            eps = self.host_eps(n_samples, device)
        seed = offset = 0
        if not mean_field_inference and eps is None:
            seed = self._seed()
        return E.bbb_sample(self.infer_cfg(), mu, rho, n_samples, self._prior, eps=eps,
                            mean_field=mean_field_inference, seed=seed, offset=offset)

    def mc_weights(self, n_samples, n_curves, device, mean_field_inference=False, eps=None):
        """(n_samples weight draws in kernel layout, None); sets self.kl to the (S,) KL vector."""
        wk, self.kl = self._sample(n_samples, mean_field_inference, eps, device)
        return wk, None

    def forward(self, x, mean_field_inference=False, eps=None):
        dev = x.data.device if isinstance(x, PackedSequence) else x.device
        if self._use_train_path():
            if eps is None and not mean_field_inference and self.draws == "host":
                eps = self.host_eps(1, dev)
            logits, self.kl = self._train_forward(x, lambda plan: None, eps, mean_field_inference)
            return logits
        wk, kl = self._sample(1, mean_field_inference, eps, dev)
        (logits, _), _ = self._launch(x, wk, 1)
        self.kl = kl[0]
        return logits[0]

    def forward_mc(self, x, n_samples, mean_field_inference=False, eps=None):
        """n_samples weight draws x all curves in one launch -> probs (B, n_samples, C); self.kl (S,)."""
        dev = x.data.device if isinstance(x, PackedSequence) else x.device
        wk, kl = self._sample(n_samples, mean_field_inference, eps, dev)
        (_, probs), _ = self._launch(x, wk, n_samples, want_probs=True)
        self.kl = kl
        return probs.permute(1, 0, 2)


class _TrainFn(torch.autograd.Function):
    """logits, kl = f(parameters): forward = snn_bbb_sample/snn_prepare_weights + snn_lstm_forward_train,
    backward = snn_lstm_backward (+ snn_bbb_backward).  Gradients come back as views of one flat buffer
    laid out like module.flat_parameters()."""

    @staticmethod
    def forward(ctx, mod, padded, lengths, masks_fn, eps, mean_field, *params):
        cfg = mod.cfg
        dev = padded.device
        plan = E.Plan(cfg, lengths, dev)
        xp = E.pack_x(cfg, plan, padded)
        masks = masks_fn(plan) if masks_fn is not None else None
        bayes = isinstance(mod, BayesianRNN)
        ctx.bayes = bayes
        if bayes:
            mu, rho = mod._mu_rho()
            seed = 0 if (mean_field or eps is not None) else mod._seed()
            if eps is not None:
                eps = eps.reshape(-1).contiguous()
            wk, kl, terms, mode = E.bbb_sample_train(cfg, mu, rho, mod._prior, eps=eps, mean_field=mean_field,
                                                     seed=seed)
            ctx.bbb = (mu, rho, terms, mode, eps, seed)
        else:
            wk = E.prepare_weights(cfg, mod._flat("w"))
            kl = torch.zeros((), dtype=torch.float32, device=dev)
        logits, tape = E.lstm_forward_train(cfg, plan, xp, wk, masks)
        ctx.mod, ctx.plan, ctx.xp, ctx.wk, ctx.tape, ctx.masks = mod, plan, xp, wk, tape, masks
        ctx.shapes = [p.shape for p in params]
        return logits, kl

    @staticmethod
    def backward(ctx, dlogits, dkl):
        mod, cfg = ctx.mod, ctx.mod.cfg
        dev = ctx.xp.device
        if dlogits is None:
            dlogits = torch.zeros((ctx.plan.n_curves, cfg.nb_classes), dtype=torch.float32, device=dev)
        dlogits = dlogits.float().contiguous()
        dw = E.lstm_backward(cfg, ctx.plan, ctx.xp, ctx.wk, ctx.tape, dlogits, ctx.masks)
        if ctx.bayes:
            mu, rho, terms, mode, eps, seed = ctx.bbb
            P, Ps = cfg.P, cfg.P - cfg.nb_classes
            gflat = torch.empty(P + Ps, dtype=torch.float32, device=dev)
            if dkl is None:
                dkl = torch.zeros((), dtype=torch.float32, device=dev)
            E.bbb_backward(cfg, mu, rho, mod._prior, terms, dw, mode, eps=eps, seed=seed,
                           kl_scale_dev=dkl.float().contiguous(), dmu=gflat[:P], drho=gflat[P:])
        else:
            gflat = dw
        grads, off = [], 0
        for shp in ctx.shapes:
            n = int(np.prod(shp))
            grads.append(gflat[off:off + n].view(shp))
            off += n
        return (None, None, None, None, None, None, *grads)
