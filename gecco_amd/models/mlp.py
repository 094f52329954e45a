"""MLP (API of reference models/mlp.py:5-39): Linear -> act -> [Linear -> act]* -> Linear as an nn.Sequential so
the state-dict keys stay `0.weight, 0.bias, 1.alpha, 2.weight, 2.bias`; the forward runs the fused HIP linear
(bias + activation in the GEMM epilogue)."""
from typing import Callable

import torch.nn as nn
from torch import Tensor

from .. import hip_ops
from .._grad import needs_grad
from .activation import GaussianActivation


class MLP(nn.Sequential):
    def __init__(self, in_features: int, out_features: int, width_size: int, depth: int = 1,
                 activation: Callable = nn.ReLU):
        layers = [nn.Linear(in_features, width_size), activation()]
        for _ in range(depth - 1):
            layers += [nn.Linear(width_size, width_size), activation()]
        layers.append(nn.Linear(width_size, out_features))
        super().__init__(*layers)

    def forward(self, x: Tensor) -> Tensor:
        shape = x.shape
        if needs_grad(self, x):
            from .. import autograd as ag
            h3 = x.reshape(-1, shape[-2], shape[-1]) if x.dim() >= 3 else x.reshape(1, -1, shape[-1])
            return ag.mlp(self, h3).reshape(*shape[:-1], -1)
        h = x.reshape(-1, shape[-2], shape[-1]) if x.dim() >= 3 else x.reshape(1, -1, shape[-1])
        h = h.contiguous()
        mods = list(self)
        i = 0
        while i < len(mods):
            lin = mods[i]
            act = mods[i + 1] if i + 1 < len(mods) else None
            if act is None:
                h = hip_ops.linear(h, lin.weight, lin.bias)
                i += 1
            else:   # bias + activation in the GEMM epilogue: GaussianActivation, nn.ReLU (the reference's default), nn.Identity
                code, alpha = hip_ops.module_act(act)
                h = hip_ops.linear(h, lin.weight, lin.bias, act_alpha=alpha, act=code)
                i += 2
        return h.reshape(*shape[:-1], h.shape[-1])
