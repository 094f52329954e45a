"""Data-space <-> diffusion-space bijections (API of reference reparam.py:14-201), computed by the HIP
kernels of csrc/sampler.hip in the input's precision (fp32, or fp64 for the sampler state)."""
from __future__ import annotations

import math

import torch
from torch import Tensor

from . import hip_ops
from .structs import Context3d


def _c(x: Tensor) -> Tensor:
    return x if x.is_contiguous() else x.contiguous()


class Reparam(torch.nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def data_to_diffusion(self, data: Tensor, ctx: Context3d) -> Tensor:
        raise NotImplementedError()

    def diffusion_to_data(self, diff: Tensor, ctx: Context3d) -> Tensor:
        raise NotImplementedError()

    # which projection the ray lookup kernel must undo: (kind, mean, std, logit_scale)
    def lookup_spec(self):
        raise NotImplementedError(f"{type(self).__name__} is not supported by the HIP ray lookup")


class NoReparam(Reparam):
    def data_to_diffusion(self, data: Tensor, ctx: Context3d) -> Tensor:
        return data

    def diffusion_to_data(self, diff: Tensor, ctx: Context3d) -> Tensor:
        return diff

    def lookup_spec(self):
        return 0, None, None, 1.1

    def ladj_data_to_diffusion(self, data: Tensor, ctx: Context3d) -> Tensor:
        del ctx
        return torch.zeros(data.shape[0], dtype=torch.float64, device=data.device)


class GaussianReparam(Reparam):
    """(data - mean) / sigma (reference reparam.py:43-66)."""

    def __init__(self, mean: Tensor, sigma: Tensor):
        assert mean.ndim == 1
        assert mean.shape == sigma.shape
        super().__init__(mean.shape[0])
        self.register_buffer("mean", mean)
        self.register_buffer("sigma", sigma)

    def data_to_diffusion(self, data: Tensor, ctx: Context3d) -> Tensor:
        del ctx
        return hip_ops.gaussian_reparam(_c(data), self.mean, self.sigma, inverse=False)

    def diffusion_to_data(self, diff: Tensor, ctx: Context3d) -> Tensor:
        del ctx
        return hip_ops.gaussian_reparam(_c(diff), self.mean, self.sigma, inverse=True)

    def lookup_spec(self):
        return 1, self.mean, self.sigma, 1.1

    def ladj_data_to_diffusion(self, data: Tensor, ctx: Context3d) -> Tensor:
        """log |det| of (x - mean) / sigma over a cloud's points: -N sum_d log sigma_d, (B,) fp64."""
        del ctx
        return (-float(data.shape[1]) * torch.log(self.sigma.double()).sum()).expand(data.shape[0]).clone()

    def extra_repr(self) -> str:
        return f"mean={self.mean.flatten().tolist()}, sigma={self.sigma.flatten().tolist()}"


class UVLReparam(Reparam):
    """Image-plane (u, v) through atanh and log-range l, then normalised (reference reparam.py:69-201)."""

    def __init__(self, mean: Tensor, sigma: Tensor, logit_scale: float = 1.1):
        assert mean.shape == (3,)
        assert sigma.shape == (3,)
        super().__init__(dim=3)
        self.register_buffer("uvl_mean", mean)
        self.register_buffer("uvl_std", sigma)
        self.logit_scale = logit_scale

    def data_to_diffusion(self, data: Tensor, ctx: Context3d) -> Tensor:
        assert isinstance(ctx, Context3d)
        return hip_ops.uvl_reparam(_c(data), _c(ctx.K.float()), self.uvl_mean, self.uvl_std, self.logit_scale, inverse=False)

    def diffusion_to_data(self, diff: Tensor, ctx: Context3d) -> Tensor:
        assert isinstance(ctx, Context3d)
        return hip_ops.uvl_reparam(_c(diff), _c(ctx.K.float()), self.uvl_mean, self.uvl_std, self.logit_scale, inverse=True)

    def lookup_spec(self):
        return 2, self.uvl_mean, self.uvl_std, self.logit_scale

    def ladj_data_to_diffusion(self, data: Tensor, ctx: Context3d) -> Tensor:
        """log |det d(diffusion) / d(data)| summed over the points of each cloud, (B,) fp64 — gecco-jax obtains it point by point from
        `jax.jacrev` + `slogdet` (models/reparam.py:27-37); here in closed form.  With h = fx x / z + cx, w = fy y / z + cy, d = |xyz|:
        det d(h, w, d) / d(x, y, z) = fx fy d / z^3; r = atanh((2 hw - 1) / s) contributes (2 / s) / (1 - t^2) per image coordinate,
        l = log d contributes 1 / d (which cancels the d above), the normalisation 1 / std each."""
        K = ctx.K.to(device=data.device, dtype=torch.float64)
        x = data.double()
        z = x[..., 2]
        fx, fy, cx, cy = K[:, 0, 0, None], K[:, 1, 1, None], K[:, 0, 2, None], K[:, 1, 2, None]
        th = (2 * (fx * x[..., 0] / z + cx) - 1) / self.logit_scale
        tw = (2 * (fy * x[..., 1] / z + cy) - 1) / self.logit_scale
        per_point = (torch.log((fx * fy).abs()) - 3 * torch.log(z.abs()) + 2 * math.log(2.0 / self.logit_scale)
                     - torch.log1p(-th * th) - torch.log1p(-tw * tw) - torch.log(self.uvl_std.double()).sum())
        return per_point.sum(-1)

    def extra_repr(self) -> str:
        return (f"uvl_mean={self.uvl_mean.flatten().tolist()}, uvl_std={self.uvl_std.flatten().tolist()}, "
                f"logit_scale={self.logit_scale}")
