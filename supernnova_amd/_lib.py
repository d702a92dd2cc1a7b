"""ctypes binding of libsnn_hip.so (the C ABI declared in include/snn_hip.h).

The library is built in-tree by `make -C supernnova_amd/csrc` (or
__graft_entry__.build()).  There is NO fallback: if the shared object is
missing or fails to load, importing the kernels raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNN_HIP_LIB") or os.path.join(_HERE, "libsnn_hip.so")  # env override: A/B builds

SNN_POOL_MEAN, SNN_POOL_STANDARD = 0, 1
SNN_PREC_F32, SNN_PREC_BF16X3, SNN_PREC_F16X2 = 0, 1, 2
SNN_CELL_LSTM, SNN_CELL_GRU, SNN_CELL_RNN = 0, 1, 2
SNN_MASK_NONE, SNN_MASK_ARRAYS, SNN_MASK_PHILOX = 0, 1, 2
BBB_EPS, BBB_MFE, BBB_PHILOX = 0, 1, 2
TILE = 32


class SnnCfg(ctypes.Structure):
    _fields_ = [("input_size", ctypes.c_int32), ("hidden_size", ctypes.c_int32),
                ("num_layers", ctypes.c_int32), ("bidirectional", ctypes.c_int32),
                ("nb_classes", ctypes.c_int32), ("pool", ctypes.c_int32), ("precision", ctypes.c_int32),
                ("cell", ctypes.c_int32)]


class SnnPrior(ctypes.Structure):
    _fields_ = [("pi", ctypes.c_float), ("sigma1", ctypes.c_float), ("sigma2", ctypes.c_float),
                ("sigma1_out", ctypes.c_float), ("sigma2_out", ctypes.c_float)]


class SnnMasks(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int32), ("which", ctypes.c_int32), ("mid_T", ctypes.c_int32),
                ("keep_scale", ctypes.c_float), ("keep_prob", ctypes.c_float),
                ("seed", ctypes.c_uint64), ("offset", ctypes.c_uint64),
                ("curve_base", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("m_in", ctypes.c_void_p), ("m_whh", ctypes.c_void_p),
                ("m_mid", ctypes.c_void_p), ("m_out", ctypes.c_void_p)]


class SnnTrainIO(ctypes.Structure):
    """snn_train_io of include/snn_hip.h (snn_train_step_grad)."""
    _fields_ = [("x", ctypes.c_void_p), ("step_off", ctypes.c_void_p), ("stride_t", ctypes.c_int64), ("stride_n", ctypes.c_int64),
                ("curve_off", ctypes.c_void_p),
                ("order", ctypes.c_void_p), ("len_sorted", ctypes.c_void_p), ("tile_T", ctypes.c_void_p), ("tile_off", ctypes.c_void_p),
                ("n_curves", ctypes.c_int32), ("n_tiles", ctypes.c_int32), ("max_T", ctypes.c_int32), ("bbb_mode", ctypes.c_int32),
                ("xp_floats", ctypes.c_int64),
                ("target", ctypes.c_void_p),
                ("params", ctypes.c_void_p), ("rho", ctypes.c_void_p), ("eps", ctypes.c_void_p),
                ("seed", ctypes.c_uint64), ("offset", ctypes.c_uint64),
                ("prior", ctypes.POINTER(SnnPrior)), ("masks", ctypes.POINTER(SnnMasks)),
                ("ce_grad_scale", ctypes.c_float), ("kl_scale", ctypes.c_float), ("loss_kl_scale", ctypes.c_float),
                ("reserved", ctypes.c_float),
                ("xp", ctypes.c_void_p), ("wk", ctypes.c_void_p), ("kl", ctypes.c_void_p), ("kl_terms", ctypes.c_void_p),
                ("logits", ctypes.c_void_p), ("tape", ctypes.c_void_p), ("tape_floats", ctypes.c_int64),
                ("dlogits", ctypes.c_void_p), ("loss", ctypes.c_void_p), ("ce_partials", ctypes.c_void_p),
                ("dw", ctypes.c_void_p), ("grad", ctypes.c_void_p),
                ("fwd_ws", ctypes.c_void_p), ("fwd_ws_bytes", ctypes.c_int64),
                ("bwd_ws", ctypes.c_void_p), ("bwd_ws_bytes", ctypes.c_int64)]


class SnnError(RuntimeError):
    pass


_P = ctypes.c_void_p
_CFG = ctypes.POINTER(SnnCfg)
_I32P = ctypes.POINTER(ctypes.c_int32)
_I64P = ctypes.POINTER(ctypes.c_int64)

# name -> (restype, argtypes); must list every symbol include/snn_hip.h declares
SIGNATURES = {
    "snn_last_error": (ctypes.c_char_p, []),
    "snn_version": (ctypes.c_int, []),
    "snn_last_forward_kernel": (ctypes.c_char_p, []),
    "snn_param_count": (ctypes.c_int64, [_CFG]),
    "snn_kparam_count": (ctypes.c_int64, [_CFG]),
    "snn_feat": (ctypes.c_int32, [_CFG]),
    "snn_supported": (ctypes.c_int, [_CFG]),
    "snn_x_step_floats": (ctypes.c_int64, [_CFG]),
    "snn_featurize_plan": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int64, _P, _P, ctypes.c_double, _P, _P, _P, _P]),
    "snn_featurize_fill": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32] + [_P] * 10),
    "snn_h5_vlen_gather": (ctypes.c_int, [_P, ctypes.c_int64, _P, ctypes.c_int64, ctypes.c_int32, _P, ctypes.c_int64]),
    "snn_plan_build": (ctypes.c_int, [_CFG, _I32P, ctypes.c_int32, _I32P, _I32P, _I32P, _I64P]),
    "snn_pack_x": (ctypes.c_int, [_CFG, _P, ctypes.c_int64, ctypes.c_int64, _P, _P, _P, _P, _P,
                                  ctypes.c_int32, ctypes.c_int64, _P, _P]),
    "snn_pack_x_steps": (ctypes.c_int, [_CFG, _P, _P, ctypes.c_int64, _P, _P, _P, _P, _P,
                                        ctypes.c_int32, ctypes.c_int64, _P, _P]),
    "snn_prepare_weights": (ctypes.c_int, [_CFG, _P, _P, _P]),
    "snn_prepare_weights_multi": (ctypes.c_int, [_CFG, _P, ctypes.c_int64, ctypes.c_int32, _P, _P]),
    "snn_bbb_sample": (ctypes.c_int, [_CFG, _P, _P, _P, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64,
                                      ctypes.c_int32, ctypes.POINTER(SnnPrior), _P, _P, _P, _P]),
    "snn_kl_terms_floats": (ctypes.c_int64, [_CFG, ctypes.c_int32]),
    "snn_lstm_forward_workspace_bytes": (ctypes.c_int64, [_CFG, ctypes.c_int32]),
    "snn_lstm_forward": (ctypes.c_int, [_CFG, _P, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_int32, _P, ctypes.c_int64, ctypes.c_int32,
                                        ctypes.POINTER(SnnMasks), _P, _P, ctypes.c_int64, ctypes.c_int64,
                                        _P, ctypes.c_int64, _P]),
    "snn_lstm_forward_noalloc_scratch_bytes": (ctypes.c_int64, []),
    "snn_lstm_forward_noalloc": (ctypes.c_int, [_CFG, _P, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_int32, _P, ctypes.c_int64, ctypes.c_int32,
                                                ctypes.POINTER(SnnMasks), _P, _P, ctypes.c_int64, ctypes.c_int64, _P]),
    "snn_train_tape_floats": (ctypes.c_int64, [_CFG, ctypes.c_int32, ctypes.c_int32]),
    "snn_lstm_forward_train": (ctypes.c_int, [_CFG, _P, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_int32, _P, ctypes.POINTER(SnnMasks), _P, _P, ctypes.c_int64,
                                              _P, ctypes.c_int64, _P]),
    "snn_lstm_backward_workspace_bytes": (ctypes.c_int64, [_CFG, ctypes.c_int32, ctypes.c_int32]),
    "snn_last_backward_path": (ctypes.c_char_p, []),
    "snn_lstm_backward": (ctypes.c_int, [_CFG, _P, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                         _P, ctypes.POINTER(SnnMasks), _P, ctypes.c_int64, _P, _P, ctypes.c_int32,
                                         _P, ctypes.c_int64, _P]),
    "snn_bbb_backward": (ctypes.c_int, [_CFG, _P, _P, _P, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64,
                                        ctypes.POINTER(SnnPrior), _P, ctypes.c_float, _P, _P, _P, _P, ctypes.c_int32, _P]),
    "snn_ce_loss_partials": (ctypes.c_int64, [ctypes.c_int32]),
    "snn_ce_loss_grad": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, _P, _P, ctypes.c_float, _P, _P, _P, _P]),
    "snn_adam_step": (ctypes.c_int, [ctypes.c_int64, _P, _P, _P, _P, ctypes.c_int64] + [ctypes.c_float] * 6 + [_P]),
    "snn_train_step_grad": (ctypes.c_int, [_CFG, ctypes.POINTER(SnnTrainIO), _P]),
    "snn_sgd_step": (ctypes.c_int, [ctypes.c_int64, _P, _P, _P, ctypes.c_int64] + [ctypes.c_float] * 4 + [_P]),
}

_lib = None


def load():
    """Load libsnn_hip.so; raises (never falls back) when it is unavailable."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(`make -C supernnova_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "supernnova_amd has no CPU fallback.")
        # The library shares device pointers and streams with PyTorch, so both must bind to the
        # SAME HIP runtime: torch's wheel ships its own libamdhip64.so.7; loading torch first makes
        # the dynamic loader resolve our NEEDED libamdhip64.so.7 to that already-loaded copy.
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc < 0:
        msg = load().snn_last_error()
        raise SnnError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
    return rc
