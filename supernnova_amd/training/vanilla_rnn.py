"""Drop-in for supernnova/training/vanilla_rnn.py (VanillaRNN over the HIP kernels)."""
from ..modules import VanillaRNN  # noqa: F401
