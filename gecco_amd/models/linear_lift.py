"""LinearLift (API of reference models/linear_lift.py:7-46): Linear(3 -> d), SetTransformer, LayerNorm, Linear(d -> 3)."""
from __future__ import annotations

from typing import Any

from torch import Tensor, nn

from .. import hip_ops
from .._grad import needs_grad, result_into as _into
from .set_transformer import SetTransformer, _PlanCache


class LinearLift(nn.Module):
    def __init__(self, inner: SetTransformer, feature_dim: int, geometry_dim: int = 3, do_norm: bool = True):
        super().__init__()
        self.lift = nn.Linear(geometry_dim, feature_dim)
        self.inner = inner
        if do_norm:
            self.lower = nn.Sequential(nn.LayerNorm(feature_dim, elementwise_affine=False),
                                       nn.Linear(feature_dim, geometry_dim))
        else:
            self.lower = nn.Linear(feature_dim, geometry_dim)
        self._geometry_dim, self._do_norm = geometry_dim, do_norm
        self._cache = _PlanCache()

    def _check(self):
        if self._geometry_dim != 3 or not self._do_norm:
            raise NotImplementedError("the HIP LinearLift supports geometry_dim=3, do_norm=True (the reference defaults)")

    def forward(self, geometry: Tensor, embed: Tensor, raw_context: Any, post_context: Any, do_cache: bool = False,
                cache: list[Tensor] | None = None):
        del raw_context, post_context
        self._check()
        if needs_grad(self, geometry, embed):
            from .. import autograd as ag
            feats = ag.LiftFn.apply(geometry.float(), self.lift.weight, self.lift.bias)
            feats, out_cache = ag.set_transformer(self.inner, feats, embed.float(), do_cache, cache)
            return ag.LowerFn.apply(feats, self.lower[1].weight, self.lower[1].bias, self.lower[0].eps), out_cache
        feats, stats = hip_ops.lift(geometry.float().contiguous(), None, self.lift.weight, self.lift.bias, want_stats=True)
        feats, out_cache, _ = self.inner.plan().forward_(feats, embed.float(), stats=stats, hs=cache, return_h=do_cache)
        out = hip_ops.lower_edm(feats, None, None, self.lower[1].weight, self.lower[1].bias, eps=self.lower[0].eps)
        return out, out_cache

    # EDMPrecond's fused path: preconditioning, lift, set transformer, lower and the EDM combine in one C call
    def fused_edm(self, x: Tensor, sigma: Tensor, raw_context, post_context, do_cache: bool, cache, sigma_data: float,
                  out: Tensor | None = None):
        del raw_context, post_context
        self._check()
        if needs_grad(self, x, sigma):
            from .. import autograd as ag
            return _into(out, ag.linear_lift_edm(self, x.float(), sigma, sigma_data, do_cache, cache), do_cache)

        def build():
            st = self.inner.plan()
            p = dict(self.named_parameters())
            return hip_ops.LinearLiftPlan(p, st.H, st.I, sigma_data=sigma_data, act=st.act, precision=self.inner.precision,
                                          options=self.inner.options)
        plan = self._cache.get(self, build)
        return plan.forward(x.float().contiguous(), sigma.float().contiguous(), cache=cache, do_cache=do_cache, out=out)
