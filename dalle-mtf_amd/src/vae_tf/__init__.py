"""Discrete VAE on libdalle_hip (implicit-im2col MFMA convolutions, Gumbel-softmax, MSE)."""
from .models import DiscreteVAE

__all__ = ["DiscreteVAE"]
