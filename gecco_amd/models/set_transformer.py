"""Set Transformer with inducing points and noise-level conditioning (API of reference
models/set_transformer.py:14-216).  Parameters live in ordinary nn.Modules with the reference's names, so state
dicts are interchangeable; the forward passes run on the HIP kernels (csrc/gemm_f32.hip, attention_f32.hip)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .. import hip_ops
from .._grad import needs_grad
from .activation import GaussianActivation
from .mlp import MLP
from .normalization import AdaGN


def _own_settings(module: nn.Module):
    """(precision | None, ((option, value), ...)) pinned on the SetTransformer inside `module` (`SetTransformer.set_precision` /
    `set_option`; `Diffusion.set_precision` walks the model) — None / empty: follow the process-wide defaults."""
    for m in module.modules():
        if isinstance(m, SetTransformer):
            return m.precision, tuple(sorted(m.options.items()))
    return None, ()


def _param_sig(module: nn.Module):
    return (hip_ops.default_precision(), _own_settings(module)) + tuple(p.data_ptr() for p in module.parameters()) + \
        tuple(b.data_ptr() for b in module.buffers())


class _PlanCache:
    """A SetTransformerPlan holds raw device pointers; rebuild it when any parameter storage moved
    (.to(), .cuda(), load of a differently-placed state dict).  In-place updates keep it valid."""

    def __init__(self):
        self.sig = None
        self.plan = None

    # plans hold ctypes structures with raw pointers: a copied / pickled module starts with an empty cache and
    # rebuilds its plan on first use (copy.deepcopy(model), torch.save(model), spawn-based DDP, swa_utils.AveragedModel)
    def __deepcopy__(self, memo):
        return _PlanCache()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.sig = None
        self.plan = None

    def get(self, owner: nn.Module, build):
        sig = _param_sig(owner)
        if sig != self.sig:
            self.plan = build()
            self.sig = sig
        return self.plan


def _act_of(mlp: MLP) -> int:
    """Epilogue code of the MLPs' activation: GaussianActivation (every shipped config), nn.ReLU (the reference's
    default, set_transformer.py:81,133) or nn.Identity; anything else has no HIP epilogue and raises."""
    return hip_ops.module_act(mlp[1])[0]


class AttentionPool(nn.Module):
    """num_inducers learned queries attend over the N input tokens (softmax over N)."""

    def __init__(self, feature_dim: int, num_heads: int, num_inducers: int):
        super().__init__()
        assert feature_dim % num_heads == 0, (feature_dim, num_heads)
        dims_per_head = feature_dim // num_heads
        self.inducers = nn.Parameter(torch.randn(1, num_heads, num_inducers, dims_per_head))
        self.kv_proj = nn.Linear(feature_dim, feature_dim * 2, bias=False)
        self.out_proj = nn.Linear(feature_dim, feature_dim, bias=False)
        self.num_heads = num_heads
        self.feature_dim = feature_dim
        self.dims_per_head = dims_per_head

    def forward(self, kv: Tensor) -> Tensor:
        if needs_grad(self, kv):
            from .. import autograd as ag
            KV = ag.LinearFn.apply(kv, self.kv_proj.weight, None)
            return ag.LinearFn.apply(ag.PoolAttnFn.apply(KV, self.inducers, self.num_heads), self.out_proj.weight, None)
        KV = hip_ops.linear(kv.contiguous(), self.kv_proj.weight)
        merged = hip_ops.pool_attn(KV, self.inducers, self.num_heads)
        return hip_ops.linear(merged, self.out_proj.weight)


class Broadcast(nn.Module):
    """pool -> norm -> mlp -> norm -> unpool: returns the update of the input tokens (and the inducer states)."""

    def __init__(self, feature_dim: int, num_inducers: int, t_embed_dim: int, num_heads: int = 8,
                 mlp_blowup: int = 2, activation: nn.Module = nn.ReLU):
        super().__init__()
        self.pool = AttentionPool(feature_dim, num_heads, num_inducers)
        self.norm_1 = AdaGN(feature_dim, t_embed_dim)
        self.mlp = MLP(feature_dim, feature_dim, mlp_blowup * feature_dim, activation=activation)
        self.norm_2 = AdaGN(feature_dim, t_embed_dim)
        # parameter container with nn.MultiheadAttention's names/initialisation (in_proj_weight, in_proj_bias,
        # out_proj.{weight,bias}); its own forward is never called
        self.unpool = nn.MultiheadAttention(feature_dim, num_heads, batch_first=True)

    def forward(self, x: Tensor, t_embed: Tensor, return_h: bool = False, h: Tensor | None = None):
        x = x.contiguous()
        Cc = x.shape[-1]
        H = self.pool.num_heads
        if needs_grad(self, x, t_embed):
            from .. import autograd as ag
            if h is None:
                h = self.norm_2(self.mlp(self.norm_1(self.pool(x), t_embed)), t_embed)
            W, b = self.unpool.in_proj_weight, self.unpool.in_proj_bias
            attn = ag.UnpoolAttnFn.apply(ag.LinearFn.apply(x, W[:Cc], b[:Cc]), ag.LinearFn.apply(h, W[Cc:], b[Cc:]), H)
            out = ag.LinearFn.apply(attn, self.unpool.out_proj.weight, self.unpool.out_proj.bias)
            return (out, h) if return_h else (out, None)
        if h is None:
            h = self.pool(x)
            h = self.norm_1(h, t_embed)
            h = self.mlp(h)
            h = self.norm_2(h, t_embed)
        W, b = self.unpool.in_proj_weight, self.unpool.in_proj_bias
        q = hip_ops.linear(x, W[:Cc], b[:Cc])
        kvh = hip_ops.linear(h.contiguous(), W[Cc:], b[Cc:])
        attn = hip_ops.unpool_attn(q, kvh, H)
        out = hip_ops.linear(attn, self.unpool.out_proj.weight, self.unpool.out_proj.bias)
        return (out, h) if return_h else (out, None)


class BroadcastingLayer(nn.Module):
    """Pre-norm residual block: x += Broadcast(AdaGN(x)); x += MLP(AdaGN(x))."""

    def __init__(self, feature_dim: int, num_inducers: int, embed_dim: int, num_heads: int = 8,
                 mlp_blowup: int = 2, activation: nn.Module = nn.ReLU):
        super().__init__()
        self.broadcast_norm = AdaGN(feature_dim, embed_dim)
        self.broadcast = Broadcast(feature_dim, num_inducers, embed_dim, num_heads, mlp_blowup=mlp_blowup,
                                   activation=activation)
        self.mlp_norm = AdaGN(feature_dim, embed_dim)
        self.mlp = MLP(feature_dim, feature_dim, mlp_blowup * feature_dim, activation=activation)
        with torch.no_grad():  # scale down the residual branches at init
            self.broadcast.unpool.out_proj.weight *= 0.1
            self.mlp[-1].weight *= 0.1
        self._cache = _PlanCache()

    def _plan(self):
        def build():
            p = {"layers.0." + k: v for k, v in self.named_parameters()}
            return hip_ops.SetTransformerPlan(p, "", self.broadcast.pool.num_heads,
                                              self.broadcast.pool.inducers.shape[2], self.broadcast_norm.gn.num_groups,
                                              act=_act_of(self.mlp))
        return self._cache.get(self, build)

    def forward(self, x: Tensor, t_embed: Tensor, return_h: bool = False, h: Tensor | None = None):
        if needs_grad(self, x, t_embed):
            from .. import autograd as ag
            y, h_out = ag.broadcasting_layer(self, x, t_embed.float(), h)
            return y, (h_out if return_h else None)
        y, hs, _ = self._plan().forward_(x.contiguous().clone(), t_embed.float(), hs=None if h is None else [h.contiguous()],
                                         return_h=return_h)
        return y, (hs[0] if return_h else None)


class SetTransformer(nn.Module):
    """A sequence of broadcasting layers; `hs` / `return_h` thread the per-layer inducer states used by the
    upsampler."""

    def __init__(self, n_layers: int, feature_dim: int, num_inducers: int, t_embed_dim: int, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([
            BroadcastingLayer(feature_dim=feature_dim, num_inducers=num_inducers, embed_dim=t_embed_dim, **kwargs)
            for _ in range(n_layers)])
        self.feature_dim = feature_dim
        self._cache = _PlanCache()
        # this model's own arithmetic mode / path switches (None / {}: the process-wide defaults, hip_ops.set_default_precision /
        # set_option / the environment).  The reference's modules carry no global state (models/set_transformer.py:176-216); with
        # these two attributes neither do ours: two models of different precision live in one process, on any host threads.
        self.precision: str | None = None
        self.options: dict[str, int] = {}

    def set_precision(self, name: str | None) -> "SetTransformer":
        if name is not None and name not in hip_ops.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(hip_ops.PRECISIONS)}")
        self.precision = name
        return self

    def set_option(self, name: str, value: int) -> "SetTransformer":
        hip_ops._option_bit(name)   # (raises on an unknown name)
        if value < 0:
            self.options.pop(name, None)
        else:
            self.options[name] = int(bool(value))
        return self

    def plan(self) -> hip_ops.SetTransformerPlan:
        def build():
            l0 = self.layers[0]
            return hip_ops.SetTransformerPlan(dict(self.named_parameters()), "", l0.broadcast.pool.num_heads,
                                              l0.broadcast.pool.inducers.shape[2], l0.broadcast_norm.gn.num_groups,
                                              act=_act_of(l0.mlp), precision=self.precision, options=self.options)
        return self._cache.get(self, build)

    def forward(self, features: Tensor, t_embed: Tensor, return_h: bool = False, hs: list[Tensor] | None = None):
        if needs_grad(self, features, t_embed):
            from .. import autograd as ag
            return ag.set_transformer(self, features, t_embed.float(), return_h, hs)
        y, stored, _ = self.plan().forward_(features.contiguous().clone(), t_embed.float(), hs=hs, return_h=return_h)
        return (y, stored) if return_h else (y, None)
