"""Plain-PyTorch CPU restatement of the reference `gecco_torch` hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py): the checker and the `cpu_baseline`
of bench.py — never the product path.

Every function is functional (`p` is a dict of tensors keyed exactly like the
reference state dict, `pre` the key prefix) and cites the reference lines it
restates; paths are relative to /root/reference/gecco-torch/src/gecco_torch/.
Pinned against the imported reference by tools/make_golden.py (<=5e-6 fp32) and by
tests/golden/*.npz (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
from typing import Callable, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor

GN_EPS = 1e-5  # nn.GroupNorm / nn.LayerNorm default eps


# --------------------------------------------------------------------------- building blocks
def group_norm_bnc(x: Tensor, G: int, eps: float = GN_EPS) -> Tensor:
    """GroupNorm(G, affine=False) over a channels-last (B, n, C) tensor: statistics per
    (sample, group) over the n*C/G elements, biased variance.
    models/normalization.py:37-39 (AdaGN) and models/ray.py:20-30 (GroupNormBNC)."""
    B, n, C = x.shape
    xg = x.reshape(B, n, G, C // G)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = xg.var(dim=(1, 3), unbiased=False, keepdim=True)
    return ((xg - mean) / torch.sqrt(var + eps)).reshape(B, n, C)


def adagn(x: Tensor, t: Tensor, p: dict, pre: str, G: int = 32) -> Tensor:
    """AdaGN.forward, models/normalization.py:36-44.  x (B,n,C); t (B,1,ctx_dim)."""
    normed = group_norm_bnc(x, G)
    bias = F.linear(t, p[pre + "bias.weight"], p[pre + "bias.bias"])  # (B,1,C)
    scale = F.linear(t, p[pre + "scale.weight"], p[pre + "scale.bias"])
    return scale * normed + bias


def gaussian_activation(x: Tensor, alpha: Tensor, normalized: bool = True) -> Tensor:
    """GaussianActivation.forward, models/activation.py:17-24."""
    y = (-(x ** 2) / (2 * alpha ** 2)).exp()
    if normalized:
        y = (y - 0.7) / 0.28
    return y


def mlp(x: Tensor, p: dict, pre: str) -> Tensor:
    """MLP(depth=1, activation=GaussianActivation), models/mlp.py:5-39:
    Linear -> act -> Linear, children indexed 0,1,2."""
    h = F.linear(x, p[pre + "0.weight"], p[pre + "0.bias"])
    # GaussianActivation carries a parameter ("1.alpha"); without it the MLP has the reference's default nn.ReLU
    h = gaussian_activation(h, p[pre + "1.alpha"]) if (pre + "1.alpha") in p else F.relu(h)
    return F.linear(h, p[pre + "2.weight"], p[pre + "2.bias"])


def attention_pool(y: Tensor, p: dict, pre: str, H: int) -> Tensor:
    """AttentionPool.forward, models/set_transformer.py:47-65.
    kv_proj(no bias) -> split "b n (t h d) -> t b h n d" -> SDPA(inducers, K, V) (scale
    1/sqrt(hd), softmax over the N keys) -> "b h i d -> b i (h d)" -> out_proj(no bias)."""
    B, N, C = y.shape
    hd = C // H
    kv = F.linear(y, p[pre + "kv_proj.weight"])  # (B,N,2C)
    k = kv[..., :C].reshape(B, N, H, hd).permute(0, 2, 1, 3)  # (B,H,N,hd)
    v = kv[..., C:].reshape(B, N, H, hd).permute(0, 2, 1, 3)
    q = p[pre + "inducers"]  # (1,H,I,hd), no projection
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)  # (B,H,I,N)
    a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v)  # (B,H,I,hd)
    o = o.permute(0, 2, 1, 3).reshape(B, -1, C)  # (B,I,C)
    return F.linear(o, p[pre + "out_proj.weight"])


def mha_unpool(y: Tensor, h: Tensor, p: dict, pre: str, H: int) -> Tensor:
    """nn.MultiheadAttention(C, H, batch_first=True)(y, h, h, need_weights=False) as used at
    models/set_transformer.py:90,112: packed in_proj (rows [0:C]=Wq,[C:2C]=Wk,[2C:3C]=Wv) with
    bias, per-head softmax(q k^T / sqrt(hd)) v over the I inducers, out_proj with bias."""
    B, N, C = y.shape
    I = h.shape[1]
    hd = C // H
    W, b = p[pre + "in_proj_weight"], p[pre + "in_proj_bias"]
    q = F.linear(y, W[:C], b[:C]).reshape(B, N, H, hd).permute(0, 2, 1, 3)  # (B,H,N,hd)
    k = F.linear(h, W[C:2 * C], b[C:2 * C]).reshape(B, I, H, hd).permute(0, 2, 1, 3)
    v = F.linear(h, W[2 * C:], b[2 * C:]).reshape(B, I, H, hd).permute(0, 2, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)  # (B,H,N,I)
    a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v).permute(0, 2, 1, 3).reshape(B, N, C)
    return F.linear(o, p[pre + "out_proj.weight"], p[pre + "out_proj.bias"])


def inducer_states(y: Tensor, t: Tensor, p: dict, pre: str, H: int) -> Tensor:
    """h = norm_2(mlp(norm_1(pool(y)))), models/set_transformer.py:106-110."""
    h = attention_pool(y, p, pre + "pool.", H)
    h = adagn(h, t, p, pre + "norm_1.")
    h = mlp(h, p, pre + "mlp.")
    return adagn(h, t, p, pre + "norm_2.")


def broadcast(y: Tensor, t: Tensor, p: dict, pre: str, H: int, h: Tensor | None = None):
    """Broadcast.forward, models/set_transformer.py:92-117."""
    if h is None:
        h = inducer_states(y, t, p, pre, H)
    return mha_unpool(y, h, p, pre + "unpool.", H), h


def broadcasting_layer(x: Tensor, t: Tensor, p: dict, pre: str, H: int, h: Tensor | None = None):
    """BroadcastingLayer.forward, models/set_transformer.py:155-168."""
    y = adagn(x, t, p, pre + "broadcast_norm.")
    xb, h = broadcast(y, t, p, pre + "broadcast.", H, h)
    x = x + xb
    y = adagn(x, t, p, pre + "mlp_norm.")
    x = x + mlp(y, p, pre + "mlp.")
    return x, h


def n_layers_of(p: dict, pre: str) -> int:
    L = 0
    while f"{pre}layers.{L}.mlp.0.weight" in p:
        L += 1
    return L


def set_transformer(x: Tensor, t: Tensor, p: dict, pre: str, H: int,
                    return_h: bool = False, hs: Sequence[Tensor | None] | None = None):
    """SetTransformer.forward, models/set_transformer.py:198-216."""
    L = n_layers_of(p, pre)
    if hs is None:
        hs = [None] * L
    stored = []
    for i in range(L):
        x, h = broadcasting_layer(x, t, p, f"{pre}layers.{i}.", H, hs[i])
        stored.append(h)
    return x, (stored if return_h else None)


# --------------------------------------------------------------------------- wrappers
def linear_lift(geometry: Tensor, t: Tensor, p: dict, pre: str, H: int,
                do_cache: bool = False, cache=None):
    """LinearLift.forward (do_norm=True), models/linear_lift.py:33-46."""
    f = F.linear(geometry, p[pre + "lift.weight"], p[pre + "lift.bias"])
    f, out_cache = set_transformer(f, t, p, pre + "inner.", H, do_cache, cache)
    f = F.layer_norm(f, (f.shape[-1],), eps=GN_EPS)
    return F.linear(f, p[pre + "lower.1.weight"], p[pre + "lower.1.bias"]), out_cache


def project_points(xyz: Tensor, K: Tensor) -> Tensor:
    """kornia.geometry.camera.perspective.project_points (third party, not in
    /root/reference; SURVEY.md Appendix A.5 — PARITY UNPINNED, this is the definition).
    xyz (B,N,3); K (B,1,3,3) or (B,3,3).  Returns (B,N,2) = (u, v)."""
    if K.ndim == 3:
        K = K.unsqueeze(1)
    z = xyz[..., 2:3]
    scale = torch.where(z.abs() > 1e-8, 1.0 / (z + 1e-8), torch.ones_like(z))
    xy = scale * xyz[..., :2]
    u = xy[..., 0] * K[..., 0, 0] + K[..., 0, 2]
    v = xy[..., 1] * K[..., 1, 1] + K[..., 1, 2]
    return torch.stack([u, v], dim=-1)


def unproject_points(uv: Tensor, depth: Tensor, K: Tensor, normalize: bool = True) -> Tensor:
    """kornia unproject_points (Appendix A.5, PARITY UNPINNED).  uv (B,N,2); depth (B,N,1)."""
    if K.ndim == 3:
        K = K.unsqueeze(1)
    x = (uv[..., 0] - K[..., 0, 2]) / K[..., 0, 0]
    y = (uv[..., 1] - K[..., 1, 2]) / K[..., 1, 1]
    xyz = torch.stack([x, y, torch.ones_like(x)], dim=-1)
    if normalize:
        xyz = xyz / xyz.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return xyz * depth


# reparam.py ------------------------------------------------------------------------------
def gaussian_data_to_diffusion(data, mean, sigma):
    """GaussianReparam.data_to_diffusion, reparam.py:57-59."""
    return (data - mean) / sigma


def gaussian_diffusion_to_data(diff, mean, sigma):
    """GaussianReparam.diffusion_to_data, reparam.py:61-63."""
    return diff * sigma + mean


def uvl_diffusion_to_data(diff: Tensor, K: Tensor, uvl_mean: Tensor, uvl_std: Tensor,
                          logit_scale: float = 1.1) -> Tensor:
    """UVLReparam.diffusion_to_data = hwd_to_xyz(uvl_to_hwd(.)), reparam.py:102-110,159-177,
    131-137,191-201."""
    uvl = diff * uvl_std + uvl_mean
    u, v, l = uvl.unbind(-1)
    s_u = (torch.tanh(u) * logit_scale + 1.0) / 2
    s_v = (torch.tanh(v) * logit_scale + 1.0) / 2
    d = torch.exp(l)
    return unproject_points(torch.stack([s_u, s_v], -1), d.unsqueeze(-1), K, normalize=True)


def uvl_data_to_diffusion(xyz: Tensor, K: Tensor, uvl_mean: Tensor, uvl_std: Tensor,
                          logit_scale: float = 1.1) -> Tensor:
    """UVLReparam.data_to_diffusion = hwd_to_uvl(xyz_to_hwd(.)), reparam.py:112-129,139-157,
    179-189."""
    hw = project_points(xyz, K)
    d = torch.linalg.norm(xyz, dim=-1, keepdim=True)
    r = torch.arctanh((2 * hw - 1.0) / logit_scale)
    uvl = torch.cat([r, torch.log(d)], dim=-1)
    return (uvl - uvl_mean) / uvl_std


def bilinear_taps(uv: Tensor, Hh: int, Ww: int):
    """Integer tap indices + weights of F.grid_sample(bilinear, zeros, align_corners=False) for
    normalised uv in [0,1] (grid = uv*2-1), following torch's op order.
    Returns x0,y0 (int32), wx1=(ix-x0), wy1=(iy-y0) in fp32."""
    g = uv * 2 - 1
    ix = ((g[..., 0] + 1) * Ww - 1) / 2
    iy = ((g[..., 1] + 1) * Hh - 1) / 2
    x0f = torch.floor(ix)
    y0f = torch.floor(iy)
    return x0f.to(torch.int32), y0f.to(torch.int32), ix - x0f, iy - y0f


def grid_sample_bilinear_zeros(feat: Tensor, uv: Tensor) -> Tensor:
    """F.grid_sample(feat (B,C,H,W), (uv*2-1)[:, :, None], align_corners=False)
    -> (B,N,C), models/ray.py:79-84, restated tap by tap."""
    B, C, Hh, Ww = feat.shape
    x0, y0, wx1, wy1 = bilinear_taps(uv, Hh, Ww)
    x0 = x0.long()
    y0 = y0.long()
    x1, y1 = x0 + 1, y0 + 1
    # torch's weights (GridSampler): nw = (x1 - ix)(y1 - iy), ne = (ix - x0)(y1 - iy), ...
    g = uv * 2 - 1
    ix = ((g[..., 0] + 1) * Ww - 1) / 2
    iy = ((g[..., 1] + 1) * Hh - 1) / 2
    wx0, wy0 = x1.to(ix.dtype) - ix, y1.to(iy.dtype) - iy
    flat = feat.reshape(B, C, Hh * Ww)
    out = torch.zeros(B, uv.shape[1], C, dtype=feat.dtype)

    def tap(xi, yi, w):
        ok = (xi >= 0) & (xi <= Ww - 1) & (yi >= 0) & (yi <= Hh - 1)
        idx = (yi.clamp(0, Hh - 1) * Ww + xi.clamp(0, Ww - 1))  # (B,N)
        val = torch.gather(flat, 2, idx[:, None, :].expand(B, C, -1)).permute(0, 2, 1)  # (B,N,C)
        return val * (w * ok.to(w.dtype))[..., None]

    out = tap(x0, y0, wx0 * wy0) + tap(x1, y0, wx1 * wy0) + tap(x0, y1, wx0 * wy1) + tap(x1, y1, wx1 * wy1)
    return out


def extract_image_features(geometry_diffusion: Tensor, features: Sequence[Tensor], K: Tensor,
                           uvl_mean: Tensor, uvl_std: Tensor, use_torch_grid_sample: bool = False):
    """RayNetwork.extract_image_features, models/ray.py:64-87 (reparam = UVLReparam)."""
    xyz = uvl_diffusion_to_data(geometry_diffusion, K, uvl_mean, uvl_std)
    uv = project_points(xyz, K)
    outs = []
    for f in features:
        if use_torch_grid_sample:
            o = F.grid_sample(f, (uv * 2 - 1)[:, :, None, :], align_corners=False)
            outs.append(o[..., 0].permute(0, 2, 1))
        else:
            outs.append(grid_sample_bilinear_zeros(f, uv))
    return torch.cat(outs, dim=-1)


def ray_network(geometry: Tensor, t: Tensor, K: Tensor, features: Sequence[Tensor], p: dict,
                pre: str, H: int, do_cache: bool = False, cache=None):
    """RayNetwork.forward, models/ray.py:89-120 (lookup branch always fp32, :103-109)."""
    xyz_f = F.linear(geometry, p[pre + "xyz_embed.weight"], p[pre + "xyz_embed.bias"])
    raw = extract_image_features(geometry.float(), [f.float() for f in features], K.float(),
                                 p[pre + "reparam.uvl_mean"], p[pre + "reparam.uvl_std"])
    img = F.linear(group_norm_bnc(raw, 16), p[pre + "img_feature_proj.1.weight"],
                   p[pre + "img_feature_proj.1.bias"])
    f, out_cache = set_transformer(xyz_f + img, t, p, pre + "backbone.", H, do_cache, cache)
    out = F.linear(group_norm_bnc(f, 16), p[pre + "output_proj.1.weight"], p[pre + "output_proj.1.bias"])
    return out, out_cache


# --------------------------------------------------------------------------- diffusion.py
def edm_coeffs(sigma: Tensor, sigma_data: float = 1.0):
    """EDMPrecond coefficients, diffusion.py:46-51.  sigma (B,) -> four (B,1,1) tensors."""
    sigma = sigma.reshape(-1, 1, 1)
    c_skip = sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2)
    c_out = sigma * sigma_data / (sigma ** 2 + sigma_data ** 2).sqrt()
    c_in = 1 / (sigma_data ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.log() / 4
    return c_skip, c_out, c_in, c_noise


def edm_precond(model: Callable, x: Tensor, sigma: Tensor, sigma_data: float = 1.0,
                do_cache: bool = False, cache=None, return_raw: bool = False):
    """EDMPrecond.forward, diffusion.py:37-62.  `model(x_in, c_noise, do_cache, cache)` returns
    (F_x, cache)."""
    c_skip, c_out, c_in, c_noise = edm_coeffs(sigma, sigma_data)
    F_x, out_cache = model(c_in * x, c_noise, do_cache, cache)
    denoised = c_skip * x + c_out * F_x
    res = (denoised, F_x) if return_raw else denoised
    return (res, out_cache) if do_cache else res


def uncond_denoiser(p: dict, pre: str, H: int):
    """Diffusion.forward for backbone=EDMPrecond(LinearLift(...)), diffusion.py:233-247."""
    def model(x_in, c_noise, do_cache=False, cache=None):
        return linear_lift(x_in, c_noise, p, pre, H, do_cache, cache)

    def D(x, sigma, do_cache=False, cache=None, return_raw=False):
        return edm_precond(model, x, sigma, 1.0, do_cache, cache, return_raw)
    return D


def cond_denoiser(p: dict, pre: str, H: int, K: Tensor, features: Sequence[Tensor]):
    """Diffusion.forward for backbone=EDMPrecond(RayNetwork(...)) with a precomputed
    FeaturePyramidContext (post_context), diffusion.py:233-247 + models/ray.py:89-120."""
    def model(x_in, c_noise, do_cache=False, cache=None):
        return ray_network(x_in, c_noise, K, features, p, pre, H, do_cache, cache)

    def D(x, sigma, do_cache=False, cache=None, return_raw=False):
        return edm_precond(model, x, sigma, 1.0, do_cache, cache, return_raw)
    return D


def t_steps(num_steps: int, sigma_max: float, sigma_min: float, rho: float) -> Tensor:
    """Diffusion.t_steps, diffusion.py:253-269 (fp64, Karras schedule, t_N = 0)."""
    i = torch.arange(num_steps, dtype=torch.float64)
    t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros_like(t[:1])])


def churn_gamma(t_cur: float, num_steps: int, S_churn: float, S_min: float, S_max: float) -> float:
    """diffusion.py:318-322."""
    return min(S_churn / num_steps, math.sqrt(2.0) - 1) if S_min <= t_cur <= S_max else 0.0


def sample_stochastic(D: Callable, latents: Tensor, noises: Sequence[Tensor], num_steps: int,
                      sigma_max: float, sigma_min: float = 0.002, rho: float = 7, S_churn: float = 0.5,
                      S_min: float = 0.0, S_max: float = float("inf"), S_noise: float = 1.0,
                      dtype=torch.float32) -> Tensor:
    """Diffusion.sample_stochastic, diffusion.py:271-352, with the drawn noises INJECTED
    (`latents` = the first randn, `noises[i]` = the randn of step i) so CPU and GPU agree.
    Returns x_next in diffusion space (fp64); caller applies reparam.diffusion_to_data."""
    B = latents.shape[0]
    ts = t_steps(num_steps, sigma_max, sigma_min, rho)
    x_next = latents.to(torch.float64) * ts[0]
    for i in range(num_steps):
        t_cur, t_next = ts[i], ts[i + 1]
        x_cur = x_next
        gamma = churn_gamma(float(t_cur), num_steps, S_churn, S_min, S_max)
        t_hat = t_cur + gamma * t_cur
        x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).sqrt() * S_noise * noises[i].to(dtype)
        den = D(x_hat.to(dtype), t_hat.repeat(B).to(dtype)).to(torch.float64)
        d_cur = (x_hat - den) / t_hat
        x_next = x_hat + (t_next - t_hat) * d_cur
        if i < num_steps - 1:
            den = D(x_next.to(dtype), t_next.repeat(B).to(dtype)).to(torch.float64)
            d_prime = (x_next - den) / t_next
            x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
    return x_next


def upsample(D: Callable, data_diff: Tensor, new_latents: Tensor, randn: Callable, num_steps: int,
             sigma_max: float, num_substeps: int = 5, sigma_min: float = 0.002, rho: float = 7,
             S_churn: float = 0.5, S_min: float = 0.0, S_max: float = float("inf"), S_noise: float = 1.0,
             dtype=torch.float32) -> Tensor:
    """Diffusion.upsample, diffusion.py:354-470.  `data_diff` is already in diffusion space,
    `randn(shape)` supplies the noise draws in the reference's call order."""
    ts = t_steps(num_steps, sigma_max, sigma_min, rho)
    x_next = new_latents.to(torch.float64) * ts[0]
    for i in range(num_steps):
        t_cur, t_next = ts[i], ts[i + 1]
        data_ctx = data_diff + randn(data_diff.shape) * t_cur
        _, cache = D(data_ctx.to(dtype), t_cur.to(dtype).expand(data_ctx.shape[0]), do_cache=True, cache=None)
        for u in range(num_substeps):
            x_cur = x_next
            gamma = churn_gamma(float(t_cur), num_steps, S_churn, S_min, S_max)
            t_hat = t_cur + gamma * t_cur
            x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).sqrt() * S_noise * randn(x_cur.shape)
            den = D(x_hat.to(dtype), t_hat.to(dtype).expand(x_hat.shape[0]), cache=cache).to(torch.float64)
            d_cur = (x_hat - den) / t_hat
            x_next = x_hat + (t_next - t_hat) * d_cur
            if i < num_steps - 1:
                den = D(x_next.to(dtype), t_next.to(dtype).expand(x_next.shape[0]), cache=cache).to(torch.float64)
                d_prime = (x_next - den) / t_next
                x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
            if u < num_substeps - 1 and i < num_steps - 1:
                x_next = x_next + (t_cur ** 2 - t_next ** 2).sqrt() * randn(x_next.shape)
    return x_next


def evaluate_logp(D: Callable, x0: Tensor, probes: Tensor, ts: Tensor, sigma_max: float, ladj: Tensor):
    """Restatement of gecco-jax `Diffusion.evaluate_logp` (models/diffusion.py:446-540) for the EDM schedule (sigma(t) = t, scale 1):
    Heun (diffrax `Heun` with `StepTo(ts)`: every step takes the second-order correction) on the augmented state (x, delta) from ts[0]
    (sigma_min) to ts[-1] (sigma_max) with dx/dt = (x - D(x, t)) / t (models/diffusion.py:311-331) and d delta / dt = the Hutchinson
    estimate of its divergence (trace_jac_estimator, :175-192: mean over the probes of eps . grad_x (f(x) . eps), the same probes at
    every evaluation); logp = log N(latent; 0, sigma_max^2) + delta + ladj.  D(x, sigma) -> denoised, differentiable (torch autograd);
    x0 (B, N, 3) in diffusion space; probes (S, B, N, 3) of +-1; ts fp64 ascending; ladj (B,).  Test infrastructure; parity unpinned
    (jax absent from the image)."""
    B = x0.shape[0]

    def field(t, xs):
        xg = xs.float().clone().requires_grad_(True)
        f = (xg - D(xg, torch.full((B,), float(t)))) / float(t)
        div = torch.zeros(B, dtype=torch.float64)
        for k in range(probes.shape[0]):
            g, = torch.autograd.grad((f * probes[k]).sum(), xg, retain_graph=True)
            div += (g.double() * probes[k].double()).sum((1, 2))
        return f.detach().double(), div / probes.shape[0]

    x = x0.double()
    delta = torch.zeros(B, dtype=torch.float64)
    for i in range(len(ts) - 1):
        t0, t1 = float(ts[i]), float(ts[i + 1])
        h = t1 - t0
        k1, d1 = field(t0, x)
        k2, d2 = field(t1, x + h * k1)
        x = x + 0.5 * h * (k1 + k2)
        delta = delta + 0.5 * h * (d1 + d2)
    n = x0[0].numel()
    prior = -0.5 * ((x / sigma_max) ** 2).sum((1, 2)) - n * math.log(sigma_max * math.sqrt(2.0 * math.pi))
    return prior + delta + ladj.double(), prior, delta, x


def log_uniform_sigma(u: Tensor, sigma_max: float, sigma_min: float = 0.002) -> Tensor:
    """LogUniformSchedule.forward (low_discrepancy=True) with the uniform draws `u` (B,)
    injected, diffusion.py:104-115."""
    B = u.shape[0]
    u = u / B + torch.arange(B, dtype=u.dtype) / B
    return (u * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min)).exp().reshape(-1, 1, 1)


def edm_loss(D: Callable, ex_diff: Tensor, sigma: Tensor, noise: Tensor, sigma_data: float = 1.0,
             loss_scale: float = 100.0) -> Tensor:
    """EDMLoss.forward with sigma (B,1,1) and the unit noise injected, diffusion.py:136-143."""
    weight = (sigma ** 2 + sigma_data ** 2) / ((sigma * sigma_data) ** 2)
    n = noise * sigma
    D_yn = D(ex_diff + n, sigma.reshape(-1))
    return (loss_scale * weight * ((D_yn - ex_diff) ** 2)).mean()


def rel_err(y: Tensor, ref: Tensor):
    """The two norms used everywhere: max|y-ref|/max|ref| and relative L2."""
    y, ref = y.double(), ref.double()
    return ((y - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item(), \
           ((y - ref).norm() / ref.norm().clamp_min(1e-30)).item()


# --------------------------------------------------------------------------- gecco-jax samplers / metrics (f4 rows)
def sample_inpaint(D: Callable, known_diff: Tensor, m: int, draws: Sequence[Tensor], num_steps: int, num_substeps: int,
                   sigma_max: float, sigma_min: float = 0.002, rho: float = 7, S_churn: float = 0.5, S_noise: float = 1.0):
    """gecco-jax models/stochastic.py:101-199 (`_sample_inpaint`) restated on the torch schedule (`t_steps`, fp64 state like
    the torch samplers): x = [m new points | known points]; per step i, sub-step j: re-draw the known part at sigma_i,
    churn, Euler to sigma_{i+1}, Heun correction when i < steps - 1, noise back up to sigma_i between sub-steps.
    `draws` in call order: initial (B, m + n, 3); per (i, j): known noise (B, n, 3), churn noise (B, m + n, 3) [, redo noise].
    gecco-jax itself cannot be imported here (jax is not installed): parity unpinned for this function."""
    ts = t_steps(num_steps, sigma_max, sigma_min, rho)
    it = iter(draws)
    B, n, _ = known_diff.shape
    x = torch.zeros(B, m + n, 3, dtype=torch.float64)
    x[:, m:] = known_diff.double()
    x = x + (next(it) * float(ts[0])).double()
    for i in range(num_steps):
        s_cur, s_next = ts[i], ts[i + 1]
        for j in range(num_substeps):
            x = x.clone()
            x[:, m:] = known_diff.double() + (next(it) * s_cur.float()).double()
            gamma = min(S_churn / num_steps, math.sqrt(2.0) - 1)
            s_hat = s_cur + gamma * s_cur
            x_hat = x + (((s_hat ** 2 - s_cur ** 2).sqrt() * S_noise).float() * next(it)).double()
            den = D(x_hat.float(), s_hat.repeat(B).float()).double()
            d_cur = (x_hat - den) / s_hat
            x_next = x_hat + (s_next - s_hat) * d_cur
            if i < num_steps - 1:
                den2 = D(x_next.float(), s_next.repeat(B).float()).double()
                d_prime = (x_next - den2) / s_next
                x_next = x_hat + (s_next - s_hat) * (0.5 * d_cur + 0.5 * d_prime)
            if j < num_substeps - 1:
                x_next = x_next + ((s_cur ** 2 - s_next ** 2).sqrt().float() * next(it)).double()
            x = x_next
    return x[:, :m]


def distance_matrix(a: Tensor, b: Tensor, squared: bool = False) -> Tensor:
    """gecco-jax geometry.py:8-24."""
    aa, bb = (a * a).sum(-1), (b * b).sum(-1)
    d2 = (aa[..., :, None] + bb[..., None, :] - 2 * a @ b.transpose(-1, -2)).clamp_min(0.0)
    return d2 if squared else d2.sqrt()


def chamfer_distance(a: Tensor, b: Tensor, squared: bool = False) -> Tensor:
    """gecco-jax metrics.py:92-103, batched over the leading dim."""
    d = distance_matrix(a, b, squared)
    return (d.min(dim=-2).values.mean(-1) + d.min(dim=-1).values.mean(-1)) / 2


def set_pairwise_distance(a: Tensor, b: Tensor, squared: bool = False) -> Tensor:
    """gecco-jax benchmark.py:21-39 (`batched_pairwise_distance` over `chamfer_distance`): out[s, t] = Chamfer(a[s], b[t]) for
    every pair of the sets a (S, N, 3), b (T, M, 3)."""
    return torch.stack([chamfer_distance(a[s][None].expand(b.shape[0], -1, -1), b, squared) for s in range(a.shape[0])])


def set_metrics(ss, sd, dd) -> dict:
    """1-NN accuracy, MMD, coverage from the (n, n) distance matrices sample-sample, sample-data, data-data: gecco-jax
    benchmark.py:128-156 (`_assemble_dist_m`, `_one_nn_acc`, `_mmd`, `_cov`) restated in numpy — with the reference's own
    conventions: the nearest neighbour of every COLUMN of [[ss, sd], [sd^T, dd]] (infinite diagonal; numpy's first-of-equals argmin),
    `<= n` for the samples' half (sic), `> n` for the data's."""
    import numpy as np
    ss, sd, dd = (np.asarray(m, dtype=np.float64) for m in (ss, sd, dd))
    n = ss.shape[0]
    m = np.concatenate([np.concatenate([ss, sd], axis=1), np.concatenate([sd.T, dd], axis=1)], axis=0)
    np.fill_diagonal(m, float("inf"))
    amin = m.argmin(axis=0)
    one_nn = np.concatenate([amin[:n] <= n, amin[n:] > n]).mean()
    return {"1-nn": float(one_nn), "mmd": float(sd.min(axis=0).min()), "cov": float(np.unique(sd.argmin(axis=1)).size / sd.shape[1])}


def sinkhorn_cost(Cm: Tensor, epsilon: float, iterations: int) -> Tensor:
    """Log-domain Sinkhorn between uniform marginals on cost matrices (B, N, M), then <P, C>: the arithmetic of
    gecco_sinkhorn_f32 (what ott's Sinkhorn solver iterates, gecco-jax metrics.py:141-156), in fp64."""
    Cm = Cm.double()
    B, N, M = Cm.shape
    f = torch.zeros(B, N, dtype=torch.float64)
    g = torch.zeros(B, M, dtype=torch.float64)
    for _ in range(iterations):
        f = -epsilon * torch.logsumexp((g[:, None, :] - Cm) / epsilon - math.log(M), dim=2)
        g = -epsilon * torch.logsumexp((f[:, :, None] - Cm) / epsilon - math.log(N), dim=1)
    P = torch.exp((f[:, :, None] + g[:, None, :] - Cm) / epsilon - math.log(N) - math.log(M))
    return (P * Cm).sum((1, 2))


# --------------------------------------------------------------------------- ConvNeXt conditioner (a15 / f2)
def convnext_features(image: Tensor, p: dict, pre: str = "stages.", n_stages: int = 3):
    """ConvNeXtExtractor.forward (models/feature_pyramid.py:62-73) on torchvision's ConvNeXt stages restated with plain
    torch ops (torchvision itself is absent from the image: parity unpinned for this function).  torchvision
    `convnext.py`: stem Conv2d(3, C, k4, s4) + LayerNorm2d(eps 1e-6); CNBlock: x + layer_scale * (Linear(GELU(Linear(
    LayerNorm(dwconv7x7(x), eps 1e-6))))) in NHWC; downsample LayerNorm2d + Conv2d(C, 2C, k2, s2).  `p` is keyed like the
    reference's extractor: stages.{s}.0.* the stem / downsample, stages.{s}.1.{i}.block.{0,2,3,5}.*, .layer_scale."""
    def ln2d(x, w, b):
        return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, 1e-6).permute(0, 3, 1, 2)

    x, feats = image, []
    for s in range(n_stages):
        h = f"{pre}{s}.0."
        if s == 0:
            x = ln2d(F.conv2d(x, p[h + "0.weight"], p[h + "0.bias"], stride=4), p[h + "1.weight"], p[h + "1.bias"])
        else:
            x = F.conv2d(ln2d(x, p[h + "0.weight"], p[h + "0.bias"]), p[h + "1.weight"], p[h + "1.bias"], stride=2)
        i = 0
        while f"{pre}{s}.1.{i}.layer_scale" in p:
            b = f"{pre}{s}.1.{i}."
            C = x.shape[1]
            y = F.conv2d(x, p[b + "block.0.weight"], p[b + "block.0.bias"], padding=3, groups=C).permute(0, 2, 3, 1)
            y = F.layer_norm(y, (C,), p[b + "block.2.weight"], p[b + "block.2.bias"], 1e-6)
            y = F.linear(F.gelu(F.linear(y, p[b + "block.3.weight"], p[b + "block.3.bias"])), p[b + "block.5.weight"], p[b + "block.5.bias"])
            x = x + p[b + "layer_scale"] * y.permute(0, 3, 1, 2)
            i += 1
        feats.append(x)
    return feats
