"""supernnova_amd -- MI355X-native implementation of SuperNNova's RNN hot path.

Host-side mirror of the reference's module API over hand-written gfx950 HIP
kernels (libsnn_hip.so, C ABI in include/snn_hip.h).  GPU only: there is no CPU
fallback, and importing the kernels without the built library raises.
"""
__version__ = "0.1.0"
