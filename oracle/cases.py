"""Seeded parity cases shared by tools/make_golden.py (reference side) and tests/ (oracle and
HIP side).  A case is fully determined by its name: weights, inputs and noise draws come from
numpy RandomState streams, so tests/golden/*.npz store only the reference's OUTPUTS.

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import weights as W

H = 8       # num_heads in every shipped config (example_configs/*.py)
I = 64      # num_inducers in every shipped config

LAYER_CASES = {  # name: (d, N, B, seed)
    "layer_d64_N64": (64, 64, 2, 11),
    "layer_d128_N256": (128, 256, 2, 12),
    "layer_d384_N64": (384, 64, 2, 13),
}

UNCOND_CASES = {  # name: (d, L, N, seed)
    "uncond_d128_L4_N256": (128, 4, 256, 21),
    "uncond_d384_L6_N128": (384, 6, 128, 22),
}
SIGMAS5 = (0.002, 0.1, 1.0, 20.0, 165.0)

COND_CASES = {  # name: (d, L, N, hw, context_dims, seed)
    "cond_d128_L2_N96": (128, 2, 96, 64, (96, 192, 384), 31),
}
SIGMAS3 = (0.01, 1.0, 50.0)

LOOKUP_CASES = {  # name: (B, N, hw, context_dims, seed)
    "lookup_small": (2, 200, 64, (8, 16, 24), 41),
    "lookup_convnext": (2, 24, 224, (96, 192, 384), 42),
}


def _randn(seed, *shape, dtype=np.float32):
    return torch.from_numpy(np.random.RandomState(seed).randn(*shape).astype(dtype))


def layer_inputs(name):
    d, N, B, seed = LAYER_CASES[name]
    p = W.layer_state_dict(np.random.RandomState(seed), d, I, H)
    x = _randn(seed + 1, B, N, d) * 1.5 + 0.3
    sigma = torch.tensor([0.05, 30.0][:B], dtype=torch.float32)
    t = (sigma.log() / 4).reshape(B, 1, 1)
    return p, x, t


def uncond_inputs(name, sigmas=SIGMAS5):
    d, L, N, seed = UNCOND_CASES[name]
    p = W.linear_lift_state_dict(seed, d, L, I, H)
    B = len(sigmas)
    sigma = torch.tensor(sigmas, dtype=torch.float32)
    data = _randn(seed + 1, B, N, 3)
    x = data + sigma.reshape(-1, 1, 1) * _randn(seed + 2, B, N, 3)
    return p, x, sigma


def cached_inputs(name="uncond_d128_L4_N256", n_new=96):
    p, x, sigma = uncond_inputs(name)
    d, L, N, seed = UNCOND_CASES[name]
    x_new = _randn(seed + 3, x.shape[0], n_new, 3) * (1 + sigma.reshape(-1, 1, 1))
    return p, x, sigma, x_new


def lookup_inputs(name):
    B, N, hw, cdims, seed = LOOKUP_CASES[name]
    feats, K = W.synthetic_context(seed, B, hw=hw, context_dims=cdims)
    uvl_mean = torch.tensor([0.0, 0.0, 1.38])
    uvl_std = torch.tensor([0.56, 0.60, 0.49])
    # diffusion-space geometry: tanh(u*std)*1.1 spreads over (and up to 5% beyond) the image,
    # so border taps and fully out-of-bounds taps are both exercised (zeros padding)
    geom_diff = _randn(seed + 1, B, N, 3) * torch.tensor([2.0, 2.0, 0.7])
    geom_diff[:, :3, :2] *= 4.0  # saturated tanh: far outside the frustum
    return feats, K, geom_diff, uvl_mean, uvl_std


def cond_inputs(name, sigmas=SIGMAS3):
    d, L, N, hw, cdims, seed = COND_CASES[name]
    B = len(sigmas)
    p = W.ray_network_state_dict(seed, d, L, I, H, context_dims=cdims)
    feats, K = W.synthetic_context(seed + 5, B, hw=hw, context_dims=cdims)
    sigma = torch.tensor(sigmas, dtype=torch.float32)
    data = _randn(seed + 1, B, N, 3)
    x = data + sigma.reshape(-1, 1, 1) * _randn(seed + 2, B, N, 3)
    return p, x, sigma, K, feats


SAMPLER_CASE = dict(d=64, L=2, N=64, B=2, seed=51, num_steps=6, sigma_max=165.0)
UPSAMPLE_CASE = dict(d=64, L=2, N=64, B=2, seed=52, n_new=32, num_steps=3, num_substeps=2, sigma_max=165.0)
LOSS_CASE = dict(d=64, L=2, N=64, B=4, seed=53, sigma_max=165.0)
GAUSS_MEAN = (0.0, 0.01, 0.05)
GAUSS_SIGMA = (0.11, 0.04, 0.17)


def sampler_inputs():
    c = SAMPLER_CASE
    p = W.linear_lift_state_dict(c["seed"], c["d"], c["L"], I, H)
    latents = _randn(c["seed"] + 1, c["B"], c["N"], 3)
    noises = [_randn(c["seed"] + 10 + i, c["B"], c["N"], 3) for i in range(c["num_steps"])]
    return p, latents, noises


def upsample_inputs():
    c = UPSAMPLE_CASE
    p = W.linear_lift_state_dict(c["seed"], c["d"], c["L"], I, H)
    data = _randn(c["seed"] + 1, c["B"], c["N"], 3) * torch.tensor(GAUSS_SIGMA) + torch.tensor(GAUSS_MEAN)
    return p, data


def upsample_draw_list():
    """All randn draws of the upsample case, in the reference's call order."""
    c = UPSAMPLE_CASE
    rs = np.random.RandomState(c["seed"] + 2)
    B, N, n_new, S, U = c["B"], c["N"], c["n_new"], c["num_steps"], c["num_substeps"]
    out = [torch.from_numpy(rs.randn(B, n_new, 3).astype(np.float32))]  # new_latents
    for i in range(S):
        out.append(torch.from_numpy(rs.randn(B, N, 3).astype(np.float32)))  # data_ctx noise
        for u in range(U):
            out.append(torch.from_numpy(rs.randn(B, n_new, 3).astype(np.float32)))  # churn noise
            if u < U - 1 and i < S - 1:
                out.append(torch.from_numpy(rs.randn(B, n_new, 3).astype(np.float32)))  # redo noise
    return out


COND_LOSS_CASE = dict(name="cond_d128_L2_N96", seed=61, sigma_max=165.0)


def cond_loss_inputs():
    """Conditional training step: the conditional case's network and pyramid, examples given in DIFFUSION space
    (the reference's loss converts data -> diffusion; the generator feeds it diffusion_to_data of these)."""
    c = COND_LOSS_CASE
    d, L, N, hw, cdims, seed = COND_CASES[c["name"]]
    B = len(SIGMAS3)
    p = W.ray_network_state_dict(seed, d, L, I, H, context_dims=cdims)
    feats, K = W.synthetic_context(seed + 5, B, hw=hw, context_dims=cdims)
    ex_diff = _randn(c["seed"] + 1, B, N, 3) * 0.7
    u = torch.from_numpy(np.random.RandomState(c["seed"] + 2).uniform(size=B).astype(np.float32))
    noise = _randn(c["seed"] + 3, B, N, 3)
    return p, ex_diff, u, noise, K, feats


def loss_inputs():
    c = LOSS_CASE
    p = W.linear_lift_state_dict(c["seed"], c["d"], c["L"], I, H)
    ex = _randn(c["seed"] + 1, c["B"], c["N"], 3)
    u = torch.from_numpy(np.random.RandomState(c["seed"] + 2).uniform(size=c["B"]).astype(np.float32))
    noise = _randn(c["seed"] + 3, c["B"], c["N"], 3)
    return p, ex, u, noise


OPTIM_CASE = dict(d=64, L=1, seed=71, decay=0.99, steps_before=3, steps_after=2)


def optim_grads(step: int, shapes):
    """Injected gradients of optimizer step `step` (seeded; the same on the reference and the HIP side): magnitudes
    spread over four decades across parameters so that Adam's normalisation and eps both matter."""
    rs = np.random.RandomState(OPTIM_CASE["seed"] * 100 + step)
    out = []
    for j, shp in enumerate(shapes):
        scale = 10.0 ** (-(j % 5))
        out.append(torch.from_numpy(np.asarray(rs.randn(*shp) * scale, dtype=np.float32)).reshape(shp))
    return out


def ema_update_ref(ema, params, decay):
    """ema_update, ema.py:187-194: ema = ema * decay + (1 - decay) * param (fp32, per tensor)."""
    return [e * decay + p * (1.0 - decay) for e, p in zip(ema, params)]


# the reference's DEFAULT activation (nn.ReLU: models/mlp.py:12, set_transformer.py:81,133) — no shipped config uses it
RELU_CASE = dict(d=128, L=2, N=256, B=2, seed=81)


def drop_alpha(p):
    """The state dict of the same network built with activation=nn.ReLU: no `*.1.alpha` entries."""
    return {k: v for k, v in p.items() if not k.endswith(".1.alpha")}


def relu_inputs():
    c = RELU_CASE
    p = drop_alpha(W.linear_lift_state_dict(c["seed"], c["d"], c["L"], I, H))
    sigma = torch.tensor([0.05, 30.0][: c["B"]], dtype=torch.float32)
    data = _randn(c["seed"] + 1, c["B"], c["N"], 3)
    x = data + sigma.reshape(-1, 1, 1) * _randn(c["seed"] + 2, c["B"], c["N"], 3)
    return p, x, sigma


# set-vs-set evaluation metrics (gecco-jax benchmark.py:21-39, 128-156): n generated clouds against n reference clouds of N points.
# `spread`: the generated set is the reference distribution (a blob per cloud: random centre + anisotropic scale) perturbed by that much —
# small: 1-NNA near 0.5, large: near 1.
SETMETRIC_CASES = {  # name: (n, N, seed, spread)
    "sets_n24_N64_close": (24, 64, 91, 0.05),
    "sets_n24_N64_far": (24, 64, 92, 0.6),
    "sets_n16_N200": (16, 200, 93, 0.2),
}


def setmetric_inputs(name):
    """(samples, data): two sets of n clouds of N points, fp32 (n, N, 3)."""
    import torch
    n, N, seed, spread = SETMETRIC_CASES[name]
    rs = np.random.RandomState(seed)

    def blobs(shift):
        centre = rs.randn(n, 1, 3) * 0.5 + shift
        scale = 0.3 + 0.4 * rs.rand(n, 1, 3)
        return (centre + scale * rs.randn(n, N, 3)).astype(np.float32)
    data = blobs(0.0)
    samples = blobs(spread)
    return torch.from_numpy(samples), torch.from_numpy(data)
