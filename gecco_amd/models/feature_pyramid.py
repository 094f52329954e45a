"""Feature pyramid conditioner (API of reference models/feature_pyramid.py:17-73).

The ConvNeXt backbone is the boundary of the hot path: it runs once per batch through
torchvision/MIOpen and hands its maps to the HIP lookup.  torchvision is an optional dependency exactly as in the
reference; without it ConvNeXtExtractor raises at construction and feature pyramids can be supplied through any
`Conditioner` returning a FeaturePyramidContext."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Literal

import torch
from torch import Tensor, nn

from ..structs import Context3d


@dataclass
class FeaturePyramidContext:
    features: list[Tensor]
    K: Tensor
    # channels-last copies for the HIP lookup, filled lazily by RayNetwork (once per conditioner call)
    _nhwc: list[Tensor] | None = field(default=None, repr=False, compare=False)


class FeaturePyramidExtractor(nn.Module):
    def forward(self, ctx_raw: Context3d) -> FeaturePyramidContext:
        raise NotImplementedError()


class ConvNeXtExtractor(FeaturePyramidExtractor):
    def __init__(self, n_stages: int = 3, model: Literal["tiny", "small"] = "tiny", pretrained: bool = True):
        super().__init__()
        try:
            import torchvision.models as tvm
        except ImportError as e:  # same hard dependency as the reference
            raise ImportError("ConvNeXtExtractor needs torchvision (as in gecco_torch)") from e
        if model == "tiny":
            convnext = tvm.convnext_tiny(weights=tvm.ConvNeXt_Tiny_Weights.DEFAULT if pretrained else None)
        elif model == "small":
            convnext = tvm.convnext_small(weights=tvm.ConvNeXt_Small_Weights.DEFAULT if pretrained else None)
        else:
            raise ValueError(f"Unknown model {model}")
        stages = [nn.Sequential(convnext.features[i], convnext.features[i + 1])
                  for i in range(0, len(convnext.features), 2)]
        self.stages = nn.ModuleList(stages[:n_stages])
        for m in self.modules():  # stochastic depth harms generative quality (reference :56-60)
            if isinstance(m, tvm.convnext.CNBlock):
                m.stochastic_depth = torch.nn.Identity()

    def forward(self, raw_ctx: Context3d) -> FeaturePyramidContext:
        x = raw_ctx.image
        feats = []
        for stage in self.stages:
            x = stage(x)
            feats.append(x)
        return FeaturePyramidContext(features=feats, K=raw_ctx.K)
