"""RayNetwork (API of reference models/ray.py:20-120): a SetTransformer whose per-point inputs are augmented with
CNN features looked up at the projection of each noisy 3-D point."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .. import hip_ops
from .._grad import needs_grad, result_into as _into
from ..reparam import Reparam
from ..structs import Context3d
from .feature_pyramid import FeaturePyramidContext
from .set_transformer import SetTransformer, _PlanCache


class GroupNormBNC(nn.GroupNorm):
    """GroupNorm over a channels-last (B, N, C) tensor."""

    def forward(self, tensor_bnc: Tensor) -> Tensor:
        assert tensor_bnc.ndim == 3
        if self.affine:
            raise NotImplementedError("GroupNormBNC on HIP supports affine=False (as RayNetwork uses it)")
        return hip_ops.adagn(tensor_bnc.contiguous(), None, None, self.num_groups, self.eps)


def _levels(post_context: FeaturePyramidContext) -> list[Tensor]:
    if post_context._nhwc is None:
        post_context._nhwc = hip_ops.to_channels_last_levels(post_context.features)
    return post_context._nhwc


class RayNetwork(nn.Module):
    def __init__(self, backbone: SetTransformer, reparam: Reparam, context_dims: list[int]):
        super().__init__()
        self.backbone = backbone
        self.reparam = reparam
        self.context_dims = context_dims
        self.xyz_embed = nn.Linear(reparam.dim, backbone.feature_dim)
        self.img_feature_proj = nn.Sequential(GroupNormBNC(16, sum(context_dims), affine=False),
                                              nn.Linear(sum(context_dims), backbone.feature_dim))
        self.output_proj = nn.Sequential(GroupNormBNC(16, backbone.feature_dim, affine=False),
                                         nn.Linear(backbone.feature_dim, reparam.dim))
        self._cache = _PlanCache()

    def extra_repr(self) -> str:
        return f"context_dims={self.context_dims}"

    def _reparam_struct(self):
        kind, mean, std, ls = self.reparam.lookup_spec()
        return hip_ops.make_reparam(kind, mean, std, ls)

    def extract_image_features(self, geometry_diffusion: Tensor, features: list[Tensor], ctx: Context3d) -> Tensor:
        """features: NCHW maps (converted to channels-last for the kernel)."""
        levels = hip_ops.to_channels_last_levels(features)
        return hip_ops.ray_lookup(geometry_diffusion.float().contiguous(), ctx.K.float().contiguous(), levels,
                                  self._reparam_struct())

    def forward(self, geometry: Tensor, t: Tensor, raw_ctx: Context3d, post_context: FeaturePyramidContext,
                do_cache: bool = False, cache: list[Tensor] | None = None):
        if needs_grad(self, geometry, t, *post_context.features):
            from .. import autograd as ag
            return ag.ray_network(self, geometry, t.float(), raw_ctx.K, post_context.features, do_cache, cache)
        g = geometry.float().contiguous()
        xyz = hip_ops.lift(g, None, self.xyz_embed.weight, self.xyz_embed.bias)
        raw, st_raw = hip_ops.ray_lookup(g, raw_ctx.K.float().contiguous(), _levels(post_context),
                                         self._reparam_struct(), want_stats=True)
        a, o = hip_ops.adagn_coeffs(st_raw, g.shape[1], None, None, 16, self.img_feature_proj[0].eps)
        lin = self.img_feature_proj[1]
        feats, stats = hip_ops.linear(raw, lin.weight, lin.bias, pro=(a, o), residual=xyz, want_stats=True, out=xyz)
        feats, out_cache, so = self.backbone.plan().forward_(feats, t.float(), stats=stats, hs=cache, return_h=do_cache,
                                                             want_stats_out=True)
        ga, go = hip_ops.adagn_coeffs(so, g.shape[1], None, None, 16, self.output_proj[0].eps)
        out = hip_ops.lower_edm(feats, None, None, self.output_proj[1].weight, self.output_proj[1].bias, gn=(ga, go))
        return out, out_cache

    def fused_edm(self, x: Tensor, sigma: Tensor, raw_ctx: Context3d, post_context: FeaturePyramidContext,
                  do_cache: bool, cache, sigma_data: float, out: Tensor | None = None):
        if needs_grad(self, x, sigma, *post_context.features):
            from .. import autograd as ag
            return _into(out, ag.ray_network_edm(self, x.float(), sigma, sigma_data, raw_ctx.K, post_context.features, do_cache,
                                                 cache), do_cache)

        def build():
            st = self.backbone.plan()
            kind, mean, std, ls = self.reparam.lookup_spec()
            p = {k: v for k, v in self.named_parameters()}
            return hip_ops.RayNetworkPlan(p, st.H, st.I, reparam_kind=kind, rp_mean=mean, rp_std=std, logit_scale=ls,
                                          sigma_data=sigma_data, act=st.act, precision=self.backbone.precision, options=self.backbone.options)
        plan = self._cache.get(self, build)
        return plan.forward(x.float().contiguous(), sigma.float().contiguous(), raw_ctx.K.float().contiguous(),
                            _levels(post_context), cache=cache, do_cache=do_cache, out=out)
