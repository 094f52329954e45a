"""GaussianActivation (API of reference models/activation.py:5-24)."""
import torch
import torch.nn as nn

from .. import hip_ops
from .._grad import needs_grad


class GaussianActivation(nn.Module):
    """y = exp(-x^2 / (2 alpha^2)), optionally normalised to zero mean / unit std for x ~ N(0, 1)."""

    def __init__(self, normalized: bool = True):
        super().__init__()
        self.alpha = nn.Parameter(torch.tensor(1.0))
        self.normalized = normalized

    def forward(self, x):
        if needs_grad(self, x):
            from ..autograd import GaussActFn
            return GaussActFn.apply(x, self.alpha, self.normalized)
        return hip_ops.gaussian_act(x.contiguous(), self.alpha, self.normalized)
