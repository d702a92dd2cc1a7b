"""Drop-in for supernnova/training/bayesian_rnn.py (BayesianRNN over the HIP kernels)."""
from ..modules import BayesianRNN, Prior  # noqa: F401
