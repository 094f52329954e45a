"""Closed-form / seeded weight and input generators shared by the golden-vector
script and the tests (so fixtures only need to store *outputs*).

State-dict key names and shapes follow the reference exactly
(gecco-torch/src/gecco_torch/models/set_transformer.py:20-153,
normalization.py:15-34, mlp.py:6-39, activation.py:12-15, linear_lift.py:14-31,
models/ray.py:34-59).  numpy's legacy RandomState is used because its stream is
stable across numpy versions and platforms.

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
from __future__ import annotations

import numpy as np
import torch


def _t(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _linear(rs, out_f, in_f, bias=True, prefix="", gain=1.0):
    bound = gain / np.sqrt(in_f)
    d = {prefix + "weight": _t(rs.uniform(-bound, bound, size=(out_f, in_f)))}
    if bias:
        d[prefix + "bias"] = _t(rs.uniform(-bound, bound, size=(out_f,)))
    return d


def _adagn(rs, C, ctx_dim, prefix):
    # deliberately NON-default (reference zero/one-inits these, normalization.py:30-34)
    return {
        prefix + "bias.weight": _t(0.2 * rs.randn(C, ctx_dim)),
        prefix + "bias.bias": _t(0.1 * rs.randn(C)),
        prefix + "scale.weight": _t(0.2 * rs.randn(C, ctx_dim)),
        prefix + "scale.bias": _t(1.0 + 0.1 * rs.randn(C)),
    }


def _mlp(rs, d, blowup, prefix):
    out = {}
    out.update(_linear(rs, blowup * d, d, prefix=prefix + "0."))
    out[prefix + "1.alpha"] = _t(np.array(rs.uniform(0.8, 1.2)))
    out.update(_linear(rs, d, blowup * d, prefix=prefix + "2."))
    return out


def layer_state_dict(rs, d, I, H, t_dim=1, blowup=2, prefix=""):
    """One BroadcastingLayer (set_transformer.py:120-153)."""
    hd = d // H
    sd = {}
    sd.update(_adagn(rs, d, t_dim, prefix + "broadcast_norm."))
    sd[prefix + "broadcast.pool.inducers"] = _t(rs.randn(1, H, I, hd))
    sd.update(_linear(rs, 2 * d, d, bias=False, prefix=prefix + "broadcast.pool.kv_proj."))
    sd.update(_linear(rs, d, d, bias=False, prefix=prefix + "broadcast.pool.out_proj."))
    sd.update(_adagn(rs, d, t_dim, prefix + "broadcast.norm_1."))
    sd.update(_mlp(rs, d, blowup, prefix + "broadcast.mlp."))
    sd.update(_adagn(rs, d, t_dim, prefix + "broadcast.norm_2."))
    b = 1.0 / np.sqrt(d)
    sd[prefix + "broadcast.unpool.in_proj_weight"] = _t(rs.uniform(-b, b, size=(3 * d, d)))
    sd[prefix + "broadcast.unpool.in_proj_bias"] = _t(rs.uniform(-b, b, size=(3 * d,)))
    sd.update(_linear(rs, d, d, prefix=prefix + "broadcast.unpool.out_proj."))
    sd.update(_adagn(rs, d, t_dim, prefix + "mlp_norm."))
    sd.update(_mlp(rs, d, blowup, prefix + "mlp."))
    return sd


def set_transformer_state_dict(seed, d, L, I, H, t_dim=1, blowup=2, prefix=""):
    rs = np.random.RandomState(seed)
    sd = {}
    for i in range(L):
        sd.update(layer_state_dict(rs, d, I, H, t_dim, blowup, prefix=f"{prefix}layers.{i}."))
    return sd


def linear_lift_state_dict(seed, d, L, I, H, t_dim=1, blowup=2, prefix="", geometry_dim=3):
    """LinearLift(SetTransformer) (linear_lift.py:22-31): lift, inner.*, lower.1."""
    rs = np.random.RandomState(seed + 1000)
    sd = {}
    sd.update(_linear(rs, d, geometry_dim, prefix=prefix + "lift.", gain=1.0))
    sd.update(set_transformer_state_dict(seed, d, L, I, H, t_dim, blowup, prefix=prefix + "inner."))
    sd.update(_linear(rs, geometry_dim, d, prefix=prefix + "lower.1."))
    return sd


def ray_network_state_dict(seed, d, L, I, H, context_dims=(96, 192, 384), t_dim=1, blowup=2,
                           prefix="", uvl_mean=(0.0, 0.0, 1.38), uvl_std=(0.56, 0.60, 0.49)):
    """RayNetwork(SetTransformer, UVLReparam) (models/ray.py:46-59)."""
    rs = np.random.RandomState(seed + 2000)
    sd = {}
    sd.update(set_transformer_state_dict(seed, d, L, I, H, t_dim, blowup, prefix=prefix + "backbone."))
    sd[prefix + "reparam.uvl_mean"] = _t(np.array(uvl_mean))
    sd[prefix + "reparam.uvl_std"] = _t(np.array(uvl_std))
    sd.update(_linear(rs, d, 3, prefix=prefix + "xyz_embed."))
    sd.update(_linear(rs, d, int(sum(context_dims)), prefix=prefix + "img_feature_proj.1."))
    sd.update(_linear(rs, 3, d, prefix=prefix + "output_proj.1."))
    return sd


def synthetic_cloud(seed, B, N, sigma_min=0.002, sigma_max=165.0):
    """SURVEY.md 8(d) synthetic inputs: unit-variance data, stratified log-uniform sigma
    (diffusion.py:104-115), x = data + sigma*noise."""
    rs = np.random.RandomState(seed)
    data = rs.randn(B, N, 3)
    u = (np.arange(B) + rs.uniform(size=B)) / B
    sigma = np.exp(np.log(sigma_min) + u * (np.log(sigma_max) - np.log(sigma_min)))
    x = data + sigma[:, None, None] * rs.randn(B, N, 3)
    return _t(x), _t(sigma)


def synthetic_context(seed, B, hw=224, context_dims=(96, 192, 384), strides=(4, 8, 16)):
    """Random feature pyramids (NCHW fp32, like ConvNeXt stages 0-2) + normalised intrinsics K
    (data/shapenet_cond.py:58: K maps to [0,1]^2 image coordinates)."""
    rs = np.random.RandomState(seed)
    feats = [_t(rs.randn(B, c, hw // s, hw // s)) for c, s in zip(context_dims, strides)]
    f = rs.uniform(0.8, 1.5, size=B)
    K = np.zeros((B, 3, 3))
    K[:, 0, 0] = f
    K[:, 1, 1] = f * rs.uniform(0.9, 1.1, size=B)
    K[:, 0, 2] = 0.5
    K[:, 1, 2] = 0.5
    K[:, 2, 2] = 1.0
    return feats, _t(K)


def frustum_points(seed, B, N, K, outside_frac=0.05):
    """xyz points in the camera frustum (z in [1,5]) with a fraction pushed outside the image."""
    rs = np.random.RandomState(seed)
    uv = rs.uniform(0.02, 0.98, size=(B, N, 2))
    out = rs.uniform(size=(B, N)) < outside_frac
    uv[out] = rs.uniform(-0.3, 1.3, size=(int(out.sum()), 2))
    z = rs.uniform(1.0, 5.0, size=(B, N, 1))
    Kn = K.numpy().astype(np.float64)
    x = (uv[..., 0:1] - Kn[:, None, 0, 2:3]) / Kn[:, None, 0, 0:1] * z
    y = (uv[..., 1:2] - Kn[:, None, 1, 2:3]) / Kn[:, None, 1, 1:2] * z
    return _t(np.concatenate([x, y, z], axis=-1))
