"""Import the real reference (`/root/reference/gecco-torch`) with import-time stubs.

BUILD-CONTAINER ONLY: `/root/reference` does not exist on the GPU box; nothing under tests/ (gpu
marker), bench.py or __graft_entry__ imports this module.  Used by tools/make_golden.py, tools/make_golden_optim.py and
tools/jax_standin_check.py (the golden generators; the tests read only the vectors they wrote).

Stubs:
* `gecco_torch` registered as a bare namespace so `gecco_torch/__init__.py` (which pulls the
  data modules -> imageio/h5py/lightning) is skipped;
* `lightning.pytorch.LightningModule` -> a plain `nn.Module` subclass (base class only);
* `kornia.geometry.camera.perspective.{project_points,unproject_points}` -> the oracle's
  definitions (kornia is absent and unpinned: "parity unpinned" for those two functions);
* `torchvision.models` -> empty module (ConvNeXtExtractor is then not constructible; feature
  pyramids are fed directly as FeaturePyramidContext).
"""
from __future__ import annotations

import os
import sys
import types

import torch

REF_SRC = "/root/reference/gecco-torch/src/gecco_torch"


def available() -> bool:
    return os.path.isdir(REF_SRC)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """Returns a namespace with the reference classes."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference not present (expected only in the build container)")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import cpu_ref

    pkg = types.ModuleType("gecco_torch")
    pkg.__path__ = [REF_SRC]
    sys.modules["gecco_torch"] = pkg

    class _LM(torch.nn.Module):
        def log(self, *a, **k):
            pass

    if "lightning" not in sys.modules:
        pl = _stub("lightning.pytorch", LightningModule=_LM, LightningDataModule=object, Callback=object)
        _stub("lightning", pytorch=pl)
    if "kornia" not in sys.modules:
        _stub("kornia")
        _stub("kornia.geometry")
        _stub("kornia.geometry.camera")

        def project_points(point_3d, camera_matrix):
            return cpu_ref.project_points(point_3d, camera_matrix)

        def unproject_points(point_2d, depth, camera_matrix, normalize=False):
            return cpu_ref.unproject_points(point_2d, depth, camera_matrix, normalize=normalize)

        _stub("kornia.geometry.camera.perspective", project_points=project_points,
              unproject_points=unproject_points)
    if "torchvision" not in sys.modules:
        tvm = _stub("torchvision.models")
        _stub("torchvision", models=tvm)

    from gecco_torch.models.set_transformer import SetTransformer, BroadcastingLayer, Broadcast, AttentionPool
    from gecco_torch.models.normalization import AdaGN
    from gecco_torch.models.mlp import MLP
    from gecco_torch.models.activation import GaussianActivation
    from gecco_torch.models.linear_lift import LinearLift
    from gecco_torch.models.ray import RayNetwork, GroupNormBNC
    from gecco_torch.models.feature_pyramid import FeaturePyramidContext
    from gecco_torch import reparam as reparam_mod
    from gecco_torch import diffusion as diffusion_mod
    from gecco_torch.structs import Context3d, Example

    ns = types.SimpleNamespace(**{k: v for k, v in locals().items() if not k.startswith("_")})
    _loaded = ns
    return ns


class TorchRandnProxy:
    """Stands in for the `torch` global of the reference's diffusion module so that the
    reference's own sampler loops run on *injected* noise (a list consumed in call order)."""

    def __init__(self, draws):
        self._draws = list(draws)
        self._i = 0

    def randn(self, shape, *a, **k):
        t = self._draws[self._i]
        self._i += 1
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.to(k.get("dtype", torch.float32))

    def __getattr__(self, name):
        return getattr(torch, name)


def build_uncond(ns, d, L, I, H, sigma_max=165.0, mean=(0.0, 0.01, 0.05), sigma=(0.11, 0.04, 0.17)):
    net = ns.LinearLift(
        inner=ns.SetTransformer(n_layers=L, num_inducers=I, feature_dim=d, t_embed_dim=1, num_heads=H,
                                activation=ns.GaussianActivation),
        feature_dim=d)
    D = ns.diffusion_mod
    return D.Diffusion(backbone=D.EDMPrecond(model=net), conditioner=D.IdleConditioner(),
                       reparam=ns.reparam_mod.GaussianReparam(torch.tensor(mean), torch.tensor(sigma)),
                       loss=D.EDMLoss(schedule=D.LogUniformSchedule(max=sigma_max)))


def build_cond(ns, d, L, I, H, features, context_dims=(96, 192, 384), sigma_max=180.0,
               uvl_mean=(0.0, 0.0, 1.38), uvl_std=(0.56, 0.60, 0.49)):
    D = ns.diffusion_mod
    rp = ns.reparam_mod.UVLReparam(torch.tensor(uvl_mean), torch.tensor(uvl_std))
    net = ns.RayNetwork(
        backbone=ns.SetTransformer(n_layers=L, num_inducers=I, feature_dim=d, t_embed_dim=1, num_heads=H,
                                   activation=ns.GaussianActivation),
        reparam=rp, context_dims=context_dims)

    class FixedPyramid(D.Conditioner):
        def forward(self, raw_ctx):
            return ns.FeaturePyramidContext(features=features, K=raw_ctx.K)

    return D.Diffusion(backbone=D.EDMPrecond(model=net), conditioner=FixedPyramid(), reparam=rp,
                       loss=D.EDMLoss(schedule=D.LogUniformSchedule(max=sigma_max)))
