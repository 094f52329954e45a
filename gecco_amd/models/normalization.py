"""AdaGN (API of reference models/normalization.py:14-44): GroupNorm over (points x channels-in-group), then a
noise-level dependent scale and shift."""
import torch
from torch import Tensor, nn

from .. import hip_ops
from .._grad import needs_grad


class AdaNorm(nn.Module):
    def forward(self, x: Tensor, ctx: Tensor) -> Tensor:
        raise NotImplementedError()


class AdaGN(AdaNorm):
    def __init__(self, num_channels: int, ctx_dim: int, num_groups: int = 32):
        super().__init__()
        # kept for state-dict / repr compatibility (affine=False: no parameters); statistics run in HIP
        self.gn = nn.GroupNorm(num_groups=num_groups, num_channels=num_channels, affine=False)
        self.bias = nn.Linear(ctx_dim, num_channels)
        self.scale = nn.Linear(ctx_dim, num_channels)
        with torch.no_grad():  # starts as a plain GroupNorm
            self.bias.weight.fill_(0.0)
            self.bias.bias.fill_(0.0)
            self.scale.weight.fill_(0.0)
            self.scale.bias.fill_(1.0)

    def forward(self, x: Tensor, ctx: Tensor) -> Tensor:
        """x (B, n, C) [or (B, ..., C)], ctx (B, 1, ctx_dim)."""
        B, Cc = x.shape[0], x.shape[-1]
        if needs_grad(self, x, ctx):
            from .. import autograd as ag
            return ag.adagn(self, x.reshape(B, -1, Cc), ctx.reshape(B, 1, -1).float()).reshape(x.shape)
        y = hip_ops.adagn(x.reshape(B, -1, Cc).contiguous(), ctx.reshape(B, 1, -1).float(),
                          (self.scale.weight, self.scale.bias, self.bias.weight, self.bias.bias),
                          self.gn.num_groups, self.gn.eps)
        return y.reshape(x.shape)
