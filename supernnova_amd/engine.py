"""Thin host layer over the C ABI: owns device buffers (torch tensors), builds the
batching plan, and launches the kernels on torch's current HIP stream.

PyTorch is plumbing here (allocator, streams); all arithmetic of the hot path
runs in libsnn_hip.so.  Nothing in this module computes on the CPU.
"""
import ctypes
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib as L


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(t, name):
    if not t.is_cuda:
        raise L.SnnError(
            f"{name} must live on a HIP device: supernnova_amd runs only on the GPU (no CPU fallback)")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise L.SnnError(f"{name} must be contiguous float32")


@dataclass(frozen=True)
class Config:
    """The settings the modules read (training/vanilla_rnn.py:9-16)."""
    input_size: int
    hidden_size: int
    num_layers: int = 2
    bidirectional: bool = True
    nb_classes: int = 2
    pool: str = "mean"
    precision: str = "f32"      # "f32" | "bf16x3" | "f16x2" (include/snn_hip.h: SNN_PREC_*; the last two inference only)
    cell: str = "lstm"          # settings.layer_type: "lstm" | "gru" | "rnn" (SNN_CELL_*; gru / rnn: f32, H 32 / 64, no Bayesian path)

    @property
    def c(self):
        return L.SnnCfg(self.input_size, self.hidden_size, self.num_layers, int(bool(self.bidirectional)),
                        self.nb_classes, L.SNN_POOL_MEAN if self.pool == "mean" else L.SNN_POOL_STANDARD,
                        {"f32": L.SNN_PREC_F32, "bf16x3": L.SNN_PREC_BF16X3, "f16x2": L.SNN_PREC_F16X2}[self.precision],
                        {"lstm": L.SNN_CELL_LSTM, "gru": L.SNN_CELL_GRU, "rnn": L.SNN_CELL_RNN}[self.cell])

    def with_precision(self, precision):
        from dataclasses import replace
        return replace(self, precision=precision)

    @property
    def x_step_floats(self):
        return L.check(L.load().snn_x_step_floats(ctypes.byref(self.c)), "snn_x_step_floats")

    @property
    def dirs(self):
        return 2 if self.bidirectional else 1

    @property
    def P(self):
        return L.check(L.load().snn_param_count(ctypes.byref(self.c)), "snn_param_count")

    @property
    def PK(self):
        return L.check(L.load().snn_kparam_count(ctypes.byref(self.c)), "snn_kparam_count")

    @property
    def F(self):
        return L.check(L.load().snn_feat(ctypes.byref(self.c)), "snn_feat")

    @property
    def n_tensors(self):
        return 4 * self.num_layers * self.dirs + 1

    def kl_terms_floats(self, n_samples):
        """floats of snn_bbb_sample's kl_terms scratch (include/snn_hip.h)."""
        return L.check(L.load().snn_kl_terms_floats(ctypes.byref(self.c), n_samples), "snn_kl_terms_floats")

    def supported(self):
        return bool(L.load().snn_supported(ctypes.byref(self.c)))


class Plan:
    """Length-sorted 32-curve tiling of a batch (snn_plan_build), host + device copies."""

    def __init__(self, cfg, lengths, device=None):
        lengths = np.ascontiguousarray(np.asarray(lengths), dtype=np.int32)
        if lengths.ndim != 1 or lengths.size < 1:
            raise L.SnnError("lengths must be a non-empty 1-D array")
        self.cfg = cfg
        self.n_curves = int(lengths.size)
        self.n_tiles = (self.n_curves + L.TILE - 1) // L.TILE
        npad = self.n_tiles * L.TILE
        # the four plan arrays are views of one host buffer, so the device copy is ONE transfer (to())
        a, b = 8 * (self.n_tiles + 1), 4 * npad
        # pinned when a GPU is present: the upload is then asynchronous (a pageable source would block the host until the
        # stream reaches the copy, i.e. serialise host and GPU once per launch)
        self._host_t = torch.empty(a + 2 * b + 4 * self.n_tiles, dtype=torch.uint8,
                                   pin_memory=device is not None and torch.device(device).type == "cuda")
        self._host = self._host_t.numpy()
        self.tile_off = self._host[:a].view(np.int64)
        self.order = self._host[a:a + b].view(np.int32)
        self.len_sorted = self._host[a + b:a + 2 * b].view(np.int32)
        self.tile_T = self._host[a + 2 * b:].view(np.int32)
        i32 = ctypes.POINTER(ctypes.c_int32)
        L.check(L.load().snn_plan_build(
            ctypes.byref(cfg.c), lengths.ctypes.data_as(i32), self.n_curves,
            self.order.ctypes.data_as(i32), self.len_sorted.ctypes.data_as(i32),
            self.tile_T.ctypes.data_as(i32), self.tile_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))),
            "snn_plan_build")
        self.max_T = int(self.tile_T[0])
        self.total_steps = int(self.tile_T.astype(np.int64).sum())
        self.xp_floats = int(self.tile_off[-1])      # of the fp32 image; see packed_floats()
        self.lengths = lengths
        self.device = None
        if device is not None:
            self.to(device)

    def packed_floats(self, cfg):
        """4-byte words of the packed x image for cfg's precision."""
        return self.total_steps * cfg.x_step_floats

    def to(self, device):
        self.device = torch.device(device)
        d = self._host_t.to(self.device, non_blocking=True)
        a, b = self.tile_off.nbytes, self.order.nbytes
        self.d_tile_off = d[:a].view(torch.int64)
        self.d_order = d[a:a + b].view(torch.int32)
        self.d_len_sorted = d[a + b:a + 2 * b].view(torch.int32)
        self.d_tile_T = d[a + 2 * b:].view(torch.int32)
        return self


class PackedRows:
    """The rows of a torch PackedSequence as snn_pack_x_steps reads them -- no padded copy:
    data (sum(batch_sizes), D) fp32 on the GPU, step_off (T,) device int64 float offset of each step's first row,
    col_off (B,) device int64 float offset of each curve's column inside a step (None: curve n is column n)."""

    def __init__(self, data, step_off, col_off, n_curves):
        self.data, self.step_off, self.col_off, self.n_curves = data, step_off, col_off, n_curves

    @property
    def device(self):
        return self.data.device

    @property
    def is_cuda(self):
        return self.data.is_cuda


def pack_x(cfg, plan, x, time_major=True):
    """x: (T,B,D) (time_major) or (B,T,D) padded fp32 on the GPU, or PackedRows -> packed tile buffer."""
    if isinstance(x, PackedRows):
        _require_cuda(x.data, "x.data")
        D = cfg.input_size
        if x.data.dim() != 2 or x.data.size(1) != D or x.n_curves != plan.n_curves or x.step_off.numel() < plan.max_T:
            raise L.SnnError(f"packed rows {tuple(x.data.shape)} / {x.step_off.numel()} steps do not match the plan "
                             f"(n_curves={plan.n_curves}, max_T={plan.max_T}, D={D})")
        if x.data.size(0) != int(plan.lengths.astype(np.int64).sum()):
            raise L.SnnError("packed rows do not hold sum(lengths) rows")
        n_xp = plan.packed_floats(cfg)
        xp = torch.empty(n_xp, dtype=torch.float32, device=x.data.device)
        L.check(L.load().snn_pack_x_steps(ctypes.byref(cfg.c), _ptr(x.data), _ptr(x.step_off), D,
                                          None if x.col_off is None else _ptr(x.col_off), _ptr(plan.d_order),
                                          _ptr(plan.d_len_sorted), _ptr(plan.d_tile_T), _ptr(plan.d_tile_off), plan.n_tiles,
                                          n_xp, _ptr(xp), _stream()), "snn_pack_x_steps")
        return xp
    _require_cuda(x, "x")
    if x.dim() != 3 or x.size(-1) != cfg.input_size:
        raise L.SnnError(f"x must be 3-D with last dim {cfg.input_size}, got {tuple(x.shape)}")
    T, B = (x.size(0), x.size(1)) if time_major else (x.size(1), x.size(0))
    if B != plan.n_curves or T < plan.max_T:
        raise L.SnnError(f"x {tuple(x.shape)} does not match plan (n_curves={plan.n_curves}, max_T={plan.max_T})")
    D = cfg.input_size
    st, sn = (B * D, D) if time_major else (D, T * D)
    n_xp = plan.packed_floats(cfg)
    xp = torch.empty(n_xp, dtype=torch.float32, device=x.device)
    L.check(L.load().snn_pack_x(ctypes.byref(cfg.c), _ptr(x), st, sn, None, _ptr(plan.d_order),
                                _ptr(plan.d_len_sorted), _ptr(plan.d_tile_T), _ptr(plan.d_tile_off), plan.n_tiles,
                                n_xp, _ptr(xp), _stream()), "snn_pack_x")
    return xp


def pack_ragged(cfg, plan, rows, curve_off, shared_rows=False):
    """rows: concatenated (sum(len), D) fp32 on the GPU; curve_off: int64 (N,) float offset of each
    curve's first row (device).  No padded copy is ever materialised.
    shared_rows: the plan's curves are VIEWS into `rows` that may overlap (prefixes of the same light curve for
    early classification): rows then need not hold sum(lengths) rows; the caller guarantees that every view
    [curve_off, curve_off + length * D) lies inside `rows`."""
    _require_cuda(rows, "rows")
    D = cfg.input_size
    if rows.dim() != 2 or rows.size(1) != D:
        raise L.SnnError(f"rows must be (sum(len), {D}), got {tuple(rows.shape)}")
    if curve_off.dtype != torch.int64 or curve_off.numel() != plan.n_curves or not curve_off.is_cuda:
        raise L.SnnError("curve_off must be a device int64 tensor with one entry per curve")
    if not shared_rows and rows.size(0) != int(plan.lengths.astype(np.int64).sum()):
        raise L.SnnError("rows does not hold sum(lengths) rows")
    n_xp = plan.packed_floats(cfg)
    xp = torch.empty(n_xp, dtype=torch.float32, device=rows.device)
    L.check(L.load().snn_pack_x(ctypes.byref(cfg.c), _ptr(rows), D, 0, _ptr(curve_off), _ptr(plan.d_order),
                                _ptr(plan.d_len_sorted), _ptr(plan.d_tile_T), _ptr(plan.d_tile_off), plan.n_tiles,
                                n_xp, _ptr(xp), _stream()), "snn_pack_x")
    return xp


def prepare_weights(cfg, flat):
    """Canonical flat (P,) -> kernel layout (PK,)."""
    _require_cuda(flat, "flat")
    if flat.numel() != cfg.P:
        raise L.SnnError(f"flat has {flat.numel()} floats, expected {cfg.P}")
    wk = torch.empty(cfg.PK, dtype=torch.float32, device=flat.device)
    L.check(L.load().snn_prepare_weights(ctypes.byref(cfg.c), _ptr(flat), _ptr(wk), _stream()), "snn_prepare_weights")
    return wk


def prepare_weights_multi(cfg, flats):
    """(S, P) canonical weight sets -> (S, PK) kernel layout, one launch."""
    _require_cuda(flats, "flats")
    if flats.dim() != 2 or flats.size(1) != cfg.P or not flats.is_contiguous() or flats.dtype != torch.float32:
        raise L.SnnError(f"flats must be a contiguous fp32 (S, {cfg.P}) tensor, got {tuple(flats.shape)} {flats.dtype}")
    wk = torch.empty((flats.size(0), cfg.PK), dtype=torch.float32, device=flats.device)
    L.check(L.load().snn_prepare_weights_multi(ctypes.byref(cfg.c), _ptr(flats), cfg.P, flats.size(0), _ptr(wk), _stream()),
            "snn_prepare_weights_multi")
    return wk


@dataclass(frozen=True)
class Prior:
    """training/bayesian_rnn.py:753-763 for the LSTM and the output layer."""
    pi: float = 0.75
    log_sigma1: float = -1.0
    log_sigma2: float = -7.0
    log_sigma1_output: float = -1.0
    log_sigma2_output: float = -7.0

    @property
    def c(self):
        return L.SnnPrior(self.pi, math.exp(self.log_sigma1), math.exp(self.log_sigma2),
                          math.exp(self.log_sigma1_output), math.exp(self.log_sigma2_output))


def bbb_sample(cfg, mu, rho, n_samples, prior, eps=None, mean_field=False, seed=0, offset=0, want_kl=True):
    """Draw n_samples weight sets. Returns (wk (S,PK), kl (S,) or None)."""
    _require_cuda(mu, "mu")
    _require_cuda(rho, "rho")
    Ps = cfg.P - cfg.nb_classes
    if mu.numel() != cfg.P or rho.numel() != Ps:
        raise L.SnnError(f"mu/rho sizes {mu.numel()}/{rho.numel()} != {cfg.P}/{Ps}")
    if mean_field:
        mode = L.BBB_MFE
    elif eps is not None:
        _require_cuda(eps, "eps")
        if eps.numel() != n_samples * Ps:
            raise L.SnnError(f"eps has {eps.numel()} floats, expected {n_samples}x{Ps}")
        mode = L.BBB_EPS
    else:
        mode = L.BBB_PHILOX
    wk = torch.empty((n_samples, cfg.PK), dtype=torch.float32, device=mu.device)
    kl = terms = None
    if want_kl:
        kl = torch.empty(n_samples, dtype=torch.float32, device=mu.device)
        terms = torch.empty(cfg.kl_terms_floats(n_samples), dtype=torch.float32, device=mu.device)
    pr = prior.c
    L.check(L.load().snn_bbb_sample(ctypes.byref(cfg.c), _ptr(mu), _ptr(rho), _ptr(eps), mode, seed, offset,
                                    n_samples, ctypes.byref(pr), _ptr(wk), _ptr(kl), _ptr(terms), _stream()),
            "snn_bbb_sample")
    return wk, kl


@dataclass
class Masks:
    """Variational dropout masks (host-drawn arrays) or a Philox request."""
    mode: int = L.SNN_MASK_NONE
    m_in: torch.Tensor = None    # (S,N,D)
    m_whh: torch.Tensor = None   # (S,L,4H)
    m_mid: torch.Tensor = None   # (S,L-1,mid_T,N,dirs*H)
    m_out: torch.Tensor = None   # (S,N,F)
    mid_T: int = 1
    which: int = 0xF
    dropout: float = 0.0
    seed: int = 0
    offset: int = 0
    curve_base: int = 0

    def c(self):
        keep = 1.0 - self.dropout
        scale = float(np.float32(1.0) / np.float32(keep)) if keep > 0 else 0.0
        return L.SnnMasks(self.mode, self.which, self.mid_T, scale, keep, self.seed, self.offset,
                          self.curve_base, 0,
                          *(None if m is None else m.data_ptr() for m in (self.m_in, self.m_whh, self.m_mid, self.m_out)))


# Forward / backward workspaces, one per (device, stream) in use, grown on demand and kept for re-use: the forward's is
# 256 B + grid x layer-output slab (H=32, T=40: 768 workgroups x 655 KB = 503 MB; H=128, T=300: 256 x 29 MB = 7.5 GB),
# the split backward's 64 KB per tile-step (4 GB at 65 536 curves).  release_workspaces() hands them back to torch's
# allocator (e.g. between a T=300 job and a T=40 one, or when a side stream is retired: a recycled stream handle
# would otherwise inherit the retired stream's buffer, which is harmless but keeps the memory).
_workspaces = {}


def release_workspaces(device=None):
    """Drop the cached forward / backward workspaces (of `device`, or of every device)."""
    if device is not None:
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:      # "cuda" means the current device; keys carry the index
            device = torch.device("cuda", torch.cuda.current_device())
    for cache in (_workspaces, _bwd_workspaces):
        for key in [k for k in cache if device is None or k[0] == device]:
            del cache[key]


def _dev_key(device):
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def _workspace(cfg, max_T, device):
    need = L.check(L.load().snn_lstm_forward_workspace_bytes(ctypes.byref(cfg.c), max_T),
                   "snn_lstm_forward_workspace_bytes")
    device = _dev_key(device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def lstm_forward(cfg, plan, xp, wk, n_samples, masks=None, want_logits=True, want_probs=True,
                 sample_major=True):
    """Run the LSTM stack + head for n_samples draws of every curve.

    wk: (PK,) shared by all samples or (S,PK) one set per sample.
    Returns (logits, probs), each (S,N,C) if sample_major else (N,S,C), in ORIGINAL curve order.
    """
    _require_cuda(xp, "xp")
    _require_cuda(wk, "wk")
    dev = xp.device
    N, C, S = plan.n_curves, cfg.nb_classes, n_samples
    if wk.dim() == 1:
        stride = 0
        if wk.numel() != cfg.PK:
            raise L.SnnError("shared wk must have PK floats")
    else:
        if wk.shape != (S, cfg.PK):
            raise L.SnnError(f"wk shape {tuple(wk.shape)} != ({S},{cfg.PK})")
        stride = cfg.PK
    if xp.numel() != plan.packed_floats(cfg):
        raise L.SnnError("xp does not match the plan")
    shape = (S, N, C) if sample_major else (N, S, C)
    os_s, os_n = (N * C, C) if sample_major else (C, S * C)
    logits = torch.empty(shape, dtype=torch.float32, device=dev) if want_logits else None
    probs = torch.empty(shape, dtype=torch.float32, device=dev) if want_probs else None
    mc = None
    if masks is not None and masks.mode != L.SNN_MASK_NONE:
        if masks.mode == L.SNN_MASK_ARRAYS:
            F, H, D, dirs, Ln = cfg.F, cfg.hidden_size, cfg.input_size, cfg.dirs, cfg.num_layers
            exp = {"m_in": (S, N, D), "m_whh": (S, Ln, 4 * H),
                   "m_mid": (S, Ln - 1, masks.mid_T, N, dirs * H), "m_out": (S, N, F)}
            for k, shp in exp.items():
                m = getattr(masks, k)
                if m is not None:
                    _require_cuda(m, k)
                    if tuple(m.shape) != shp:
                        raise L.SnnError(f"mask {k} shape {tuple(m.shape)} != {shp}")
        mc = masks.c()
    ws = _workspace(cfg, plan.max_T, dev)
    L.check(L.load().snn_lstm_forward(
        ctypes.byref(cfg.c), _ptr(xp), _ptr(plan.d_len_sorted), _ptr(plan.d_order), _ptr(plan.d_tile_T),
        _ptr(plan.d_tile_off), N, plan.n_tiles, plan.max_T, _ptr(wk), stride, S,
        ctypes.byref(mc) if mc is not None else None, _ptr(logits), _ptr(probs), os_s, os_n,
        _ptr(ws), ws.numel(), _stream()), "snn_lstm_forward")
    return logits, probs


def last_forward_kernel():
    """Name + launch geometry of the forward kernel this thread dispatched last (snn_last_forward_kernel)."""
    return (L.load().snn_last_forward_kernel() or b"").decode()


def lstm_forward_noalloc(cfg, plan, xp, wk, n_samples, logits=None, probs=None, masks=None, sample_major=True):
    """snn_lstm_forward_noalloc: as lstm_forward, but nothing is allocated -- the caller passes the output tensors
    (either may be None) and the library's own fixed device scratch replaces the workspace (include/snn_hip.h)."""
    _require_cuda(xp, "xp")
    _require_cuda(wk, "wk")
    N, C, S = plan.n_curves, cfg.nb_classes, n_samples
    stride = 0 if wk.dim() == 1 else cfg.PK
    if (wk.dim() == 1 and wk.numel() != cfg.PK) or (wk.dim() != 1 and wk.shape != (S, cfg.PK)):
        raise L.SnnError(f"wk shape {tuple(wk.shape)} matches neither ({cfg.PK},) nor ({S},{cfg.PK})")
    if xp.numel() != plan.packed_floats(cfg):
        raise L.SnnError("xp does not match the plan")
    shape = (S, N, C) if sample_major else (N, S, C)
    os_s, os_n = (N * C, C) if sample_major else (C, S * C)
    for t, n in ((logits, "logits"), (probs, "probs")):
        if t is not None:
            _require_cuda(t, n)
            if tuple(t.shape) != shape:
                raise L.SnnError(f"{n} shape {tuple(t.shape)} != {shape}")
    mc = masks.c() if masks is not None and masks.mode != L.SNN_MASK_NONE else None
    L.check(L.load().snn_lstm_forward_noalloc(
        ctypes.byref(cfg.c), _ptr(xp), _ptr(plan.d_len_sorted), _ptr(plan.d_order), _ptr(plan.d_tile_T),
        _ptr(plan.d_tile_off), N, plan.n_tiles, plan.max_T, _ptr(wk), stride, S,
        ctypes.byref(mc) if mc is not None else None, _ptr(logits), _ptr(probs), os_s, os_n, _stream()),
        "snn_lstm_forward_noalloc")
    return logits, probs


# ----------------------------------------------------------------------------- training
def lstm_forward_train(cfg, plan, xp, wk, masks=None):
    """Training forward of one weight set. Returns (logits (N,C) original order, tape)."""
    _require_cuda(xp, "xp")
    _require_cuda(wk, "wk")
    dev = xp.device
    if wk.numel() != cfg.PK:
        raise L.SnnError("training forward takes exactly one weight set (PK floats)")
    n_tape = L.check(L.load().snn_train_tape_floats(ctypes.byref(cfg.c), plan.n_tiles, plan.max_T),
                     "snn_train_tape_floats")
    tape = torch.empty(n_tape, dtype=torch.float32, device=dev)
    logits = torch.empty((plan.n_curves, cfg.nb_classes), dtype=torch.float32, device=dev)
    mc = masks.c() if masks is not None and masks.mode != L.SNN_MASK_NONE else None
    ws = _workspace(cfg, plan.max_T, dev)
    L.check(L.load().snn_lstm_forward_train(
        ctypes.byref(cfg.c), _ptr(xp), _ptr(plan.d_len_sorted), _ptr(plan.d_order), _ptr(plan.d_tile_T),
        _ptr(plan.d_tile_off), plan.n_curves, plan.n_tiles, plan.max_T, _ptr(wk),
        ctypes.byref(mc) if mc is not None else None, _ptr(logits), _ptr(tape), tape.numel(),
        _ptr(ws), ws.numel(), _stream()), "snn_lstm_forward_train")
    return logits, tape


_bwd_workspaces = {}


def lstm_backward(cfg, plan, xp, wk, tape, dlogits, masks=None, out=None, accumulate=False):
    """BPTT: dlogits (N,C) -> gradient w.r.t. the weight set, canonical flat (P,)."""
    _require_cuda(dlogits, "dlogits")
    dev = xp.device
    if tuple(dlogits.shape) != (plan.n_curves, cfg.nb_classes):
        raise L.SnnError(f"dlogits shape {tuple(dlogits.shape)} != ({plan.n_curves},{cfg.nb_classes})")
    need = L.check(L.load().snn_lstm_backward_workspace_bytes(ctypes.byref(cfg.c), plan.n_tiles, plan.max_T),
                   "snn_lstm_backward_workspace_bytes")
    key = (_dev_key(dev), torch.cuda.current_stream(dev).cuda_stream)
    ws = _bwd_workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _bwd_workspaces[key] = ws
    if out is None:
        out = torch.empty(cfg.P, dtype=torch.float32, device=dev)
        accumulate = False
    mc = masks.c() if masks is not None and masks.mode != L.SNN_MASK_NONE else None
    L.check(L.load().snn_lstm_backward(
        ctypes.byref(cfg.c), _ptr(xp), _ptr(plan.d_len_sorted), _ptr(plan.d_order), _ptr(plan.d_tile_T),
        _ptr(plan.d_tile_off), plan.n_curves, plan.n_tiles, plan.max_T, _ptr(wk),
        ctypes.byref(mc) if mc is not None else None, _ptr(tape), tape.numel(), _ptr(dlogits), _ptr(out),
        int(bool(accumulate)), _ptr(ws), ws.numel(), _stream()), "snn_lstm_backward")
    return out


def bbb_sample_train(cfg, mu, rho, prior, eps=None, mean_field=False, seed=0, offset=0):
    """One draw for training: (wk (PK,), kl 0-dim tensor, kl_terms) -- kl_terms feeds bbb_backward."""
    Ps = cfg.P - cfg.nb_classes
    mode = L.BBB_MFE if mean_field else (L.BBB_EPS if eps is not None else L.BBB_PHILOX)
    if eps is not None:
        _require_cuda(eps, "eps")
        if eps.numel() != Ps:
            raise L.SnnError(f"eps has {eps.numel()} floats, expected {Ps}")
    wk = torch.empty(cfg.PK, dtype=torch.float32, device=mu.device)
    kl = torch.empty(1, dtype=torch.float32, device=mu.device)
    terms = torch.empty(cfg.kl_terms_floats(1), dtype=torch.float32, device=mu.device)
    pr = prior.c
    L.check(L.load().snn_bbb_sample(ctypes.byref(cfg.c), _ptr(mu), _ptr(rho), _ptr(eps), mode, seed, offset, 1,
                                    ctypes.byref(pr), _ptr(wk), _ptr(kl), _ptr(terms), _stream()), "snn_bbb_sample")
    return wk, kl[0], terms, mode


def bbb_backward(cfg, mu, rho, prior, terms, dw, mode, eps=None, seed=0, offset=0, kl_scale=0.0, kl_scale_dev=None,
                 dmu=None, drho=None, accumulate=False):
    """dCE/dw (P) -> (dmu (P), drho (P-C)) of CE + kl_scale*KL."""
    dev = mu.device
    if dmu is None:
        dmu = torch.empty(cfg.P, dtype=torch.float32, device=dev)
        drho = torch.empty(cfg.P - cfg.nb_classes, dtype=torch.float32, device=dev)
        accumulate = False
    pr = prior.c
    L.check(L.load().snn_bbb_backward(ctypes.byref(cfg.c), _ptr(mu), _ptr(rho), _ptr(eps), mode, seed, offset,
                                      ctypes.byref(pr), _ptr(terms), float(kl_scale), _ptr(kl_scale_dev), _ptr(dw),
                                      _ptr(dmu), _ptr(drho), int(bool(accumulate)), _stream()), "snn_bbb_backward")
    return dmu, drho


def ce_loss_grad(logits, target, grad_scale=None):
    """Mean CE over the batch + dlogits = grad_scale*(softmax - onehot) (default 1/N)."""
    _require_cuda(logits, "logits")
    N, C = logits.shape
    if target.dtype != torch.int64 or not target.is_cuda or target.numel() != N:
        raise L.SnnError("target must be a device int64 tensor with one entry per curve")
    n_part = L.load().snn_ce_loss_partials(N)
    # one allocation: the loss scalar, then (8-byte aligned) the fp64 per-workgroup partials large batches need
    buf = torch.empty(2 + 2 * n_part, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    L.check(L.load().snn_ce_loss_grad(N, C, _ptr(logits), _ptr(target), 1.0 / N if grad_scale is None else grad_scale,
                                      _ptr(buf), _ptr(dlogits), _ptr(buf[2:]) if n_part else None, _stream()),
            "snn_ce_loss_grad")
    return buf[0], dlogits


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
              grad_scale=1.0):
    for t, n in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _require_cuda(t, n)
        if t.numel() != params.numel():
            raise L.SnnError("adam_step: size mismatch")
    L.check(L.load().snn_adam_step(params.numel(), _ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), step,
                                   lr, betas[0], betas[1], eps, weight_decay, grad_scale, _stream()), "snn_adam_step")


def sgd_step(params, grads, momentum_buf, step, lr, momentum=0.0, weight_decay=0.0, grad_scale=1.0):
    for t, n in ((params, "params"), (grads, "grads"), (momentum_buf, "momentum_buf")):
        _require_cuda(t, n)
        if t.numel() != params.numel():
            raise L.SnnError("sgd_step: size mismatch")
    L.check(L.load().snn_sgd_step(params.numel(), _ptr(params), _ptr(grads), _ptr(momentum_buf), step, lr, momentum,
                                  weight_decay, grad_scale, _stream()), "snn_sgd_step")
