"""Timing of the image-conditional evaluation (config C3: 224x224 ConvNeXt-T-shaped pyramids, N=2048, d=384) and of
the cached (upsampling-mode) evaluation (config C5 shape: n_new=16384 points against cached inducer states).
Synthetic pyramids stand in for the ConvNeXt conditioner.  python tools/cond_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gecco_amd import hip_ops as ops  # noqa: E402

D, L, I, H = bench.D, bench.L, bench.I, bench.H


def ray_state_dict(p_ll, cdims=(96, 192, 384)):
    g = torch.Generator().manual_seed(11)
    u = lambda o, i_: (torch.rand(o, i_, generator=g) * 2 - 1) / i_ ** 0.5
    p = {k.replace("inner.", "backbone."): v for k, v in p_ll.items() if k.startswith("inner.")}
    p["xyz_embed.weight"], p["xyz_embed.bias"] = u(D, 3), u(1, D)[0]
    p["img_feature_proj.1.weight"], p["img_feature_proj.1.bias"] = u(D, sum(cdims)), u(1, D)[0]
    p["output_proj.1.weight"], p["output_proj.1.bias"] = u(3, D), u(1, 3)[0]
    p["reparam.uvl_mean"], p["reparam.uvl_std"] = torch.tensor([0.0, 0.0, 1.38]), torch.tensor([0.56, 0.60, 0.49])
    return p


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ops.set_default_precision(os.environ.get("GECCO_PRECISION", "fp16"))
    print("precision", ops.default_precision())
    N = 2048
    dev = torch.device("cuda", 0)
    p_ll = bench.random_state_dict(3)
    p = {k: v.to(dev).contiguous() for k, v in ray_state_dict(p_ll).items()}
    net = ops.RayNetworkPlan(p, H, I)
    g = torch.Generator().manual_seed(1)
    feats = [torch.randn(B, c, 224 // s, 224 // s, generator=g).to(dev) for c, s in ((96, 4), (192, 8), (384, 16))]
    levels = ops.to_channels_last_levels(feats)
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = 1.1
    K[:, 0, 2] = K[:, 1, 2] = 0.5
    K[:, 2, 2] = 1.0
    K = K.to(dev)
    x = (torch.randn(B, N, 3, generator=g) * 1.5).to(dev)
    sigma = torch.exp(torch.linspace(-6, 5, B)).to(dev)
    out = torch.empty_like(x)
    t_fwd = bench.time_events(lambda: net.forward(x, sigma, K, levels, out=out), 10)
    print(f"C3 conditional evaluation  B={B} N={N}: {t_fwd:.2f} ms = {B * N / t_fwd * 1e3:.3e} points/s")
    rp = ops.make_reparam(2, p["reparam.uvl_mean"], p["reparam.uvl_std"], 1.1)
    coef = ops.edm_coeffs(sigma)
    t_lk = bench.time_events(lambda: ops.ray_lookup(x, K, levels, rp, coef=coef, want_stats=True), 20)
    gb = B * N * (4 * 672 * 4 + 672 * 4) / 1e9
    print(f"   ray_lookup kernel: {t_lk * 1e3:.1f} us, {gb / (t_lk * 1e-3):.0f} GB/s algorithmic (12 taps x 672 ch read + 672 written per point = {gb:.2f} GB)")

    # cached (upsampling) evaluation: inducer states from one full evaluation, then n_new points
    pl = {k: v.to(dev) for k, v in p_ll.items()}
    ll = ops.LinearLiftPlan(pl, H, I)
    Bc, n_new = 8, 16384
    xk, sk = x[:Bc].contiguous(), sigma[:Bc].contiguous()
    _, cache = ll.forward(xk, sk, do_cache=True)
    xn = (torch.randn(Bc, n_new, 3, generator=g) * 1.5).to(dev)
    outn = torch.empty_like(xn)
    t_c = bench.time_events(lambda: ll.forward(xn, sk, cache=cache, out=outn), 10)
    fl = Bc * L * (12 * n_new * D * D + 4 * n_new * I * D)
    print(f"C5-shape cached evaluation B={Bc} n_new={n_new}: {t_c:.2f} ms = {Bc * n_new / t_c * 1e3:.3e} points/s "
          f"({fl / (t_c * 1e-3) / 1e12:.1f} TFLOP/s algorithmic, 12 n d^2 + 4 n I d per layer)")


if __name__ == "__main__":
    main()
