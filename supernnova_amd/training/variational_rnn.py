"""Drop-in for supernnova/training/variational_rnn.py (VariationalRNN over the HIP kernels)."""
from ..modules import VariationalRNN  # noqa: F401
