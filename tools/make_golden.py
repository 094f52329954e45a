"""Generate tests/golden/*.npz from the REAL reference (build container only).

For each case in oracle/cases.py this script
  1. builds the reference module, loads the seeded state dict with strict=True
     (which also pins the state-dict key/shape contract of SURVEY.md section 8(b)),
  2. runs the reference's own code on the seeded inputs,
  3. checks the oracle (oracle/cpu_ref.py) against it (fp32 noise floor), and
  4. stores the reference's outputs.

Run:  python tools/make_golden.py        (needs /root/reference; never runs on the GPU box)
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, cpu_ref  # noqa: E402
from oracle import weights as W  # noqa: E402
from tools import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TOL = 2e-5   # oracle-vs-reference, max-abs/max-abs, fp32 (different op order => ~1e-6 typical)


def check(name, got, ref, tol=TOL):
    e_max, e_l2 = cpu_ref.rel_err(got, ref)
    status = "ok" if e_max <= tol else "FAIL"
    print(f"  {name:44s} oracle-vs-reference max-rel {e_max:.2e}  rel-L2 {e_l2:.2e}  {status}")
    assert e_max <= tol, (name, e_max)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  -> {os.path.relpath(path, ROOT)} ({os.path.getsize(path) / 1024:.0f} KiB)")


@torch.no_grad()
def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    ns = ref_import.load()
    D = ns.diffusion_mod
    H, I = cases.H, cases.I

    # ---- (1) BroadcastingLayer
    for name, (d, N, B, seed) in cases.LAYER_CASES.items():
        print(name)
        p, x, t = cases.layer_inputs(name)
        layer = ns.BroadcastingLayer(feature_dim=d, num_inducers=I, embed_dim=1, num_heads=H,
                                     activation=ns.GaussianActivation)
        layer.load_state_dict(p, strict=True)
        y_ref, h_ref = layer(x, t, return_h=True)
        y, h = cpu_ref.broadcasting_layer(x, t, p, "", H)
        check("x_out", y, y_ref)
        check("h", h, h_ref)
        # cached-h path on the same layer
        y2_ref, _ = layer(x[:, : N // 2] * 0.5, t, return_h=False, h=h_ref)
        y2, _ = cpu_ref.broadcasting_layer(x[:, : N // 2] * 0.5, t, p, "", H, h=h_ref)
        check("x_out_cached", y2, y2_ref)
        save(name, x_out=y_ref, h=h_ref, x_out_cached=y2_ref)

    # ---- (2) EDMPrecond(LinearLift): D and raw F_x at five noise levels
    for name, (d, L, N, seed) in cases.UNCOND_CASES.items():
        print(name)
        p, x, sigma = cases.uncond_inputs(name)
        model = ref_import.build_uncond(ns, d, L, I, H)
        sd = {"backbone.model." + k: v for k, v in p.items()}
        sd["reparam.mean"] = torch.tensor(cases.GAUSS_MEAN)
        sd["reparam.sigma"] = torch.tensor(cases.GAUSS_SIGMA)
        model.load_state_dict(sd, strict=True)
        n_params = sum(v.numel() for v in model.parameters())
        print(f"  params {n_params}, keys {len(sd)}")
        den_ref = model(x, sigma, None)
        c_skip, c_out, c_in, c_noise = cpu_ref.edm_coeffs(sigma)
        F_ref, _ = model.backbone.model(c_in * x, c_noise, None, None)
        Dor = cpu_ref.uncond_denoiser(p, "", H)
        den, F_x = Dor(x, sigma, return_raw=True)
        check("denoised", den, den_ref)
        check("F_x", F_x, F_ref)
        save(name, denoised=den_ref, F_x=F_ref)
        if name == "uncond_d128_L4_N256":
            # ---- (3) cached mode
            p, x, sigma, x_new = cases.cached_inputs(name)
            (den_c_ref, cache_ref) = model(x, sigma, None, do_cache=True)
            out2_ref = model(x_new, sigma, None, cache=cache_ref)
            den_c, cache = Dor(x, sigma, do_cache=True)
            out2 = Dor(x_new, sigma, cache=cache_ref)
            check("denoised(do_cache)", den_c, den_c_ref)
            for li, (a, b) in enumerate(zip(cache, cache_ref)):
                check(f"cache[{li}]", a, b)
            check("cached eval", out2, out2_ref)
            save("cached_d128_L4", cache=torch.stack(cache_ref), out_new=out2_ref)

    # ---- (4) projective lookup (torch grid_sample on the reference side)
    for name in cases.LOOKUP_CASES:
        print(name)
        feats, K, geom, um, us = cases.lookup_inputs(name)
        rp = ns.reparam_mod.UVLReparam(um, us)
        net = ns.RayNetwork(backbone=ns.SetTransformer(n_layers=1, num_inducers=I, feature_dim=64, t_embed_dim=1,
                                                       num_heads=H, activation=ns.GaussianActivation),
                            reparam=rp, context_dims=[f.shape[1] for f in feats])
        ctx = ns.Context3d(image=torch.zeros(1), K=K)
        ref = net.extract_image_features(geom, feats, ctx)
        got = cpu_ref.extract_image_features(geom, feats, K, um, us)
        check("extract_image_features", got, ref, tol=5e-5)
        save(name, lookup=ref)

    # ---- (5) EDMPrecond(RayNetwork)
    for name, (d, L, N, hw, cdims, seed) in cases.COND_CASES.items():
        print(name)
        p, x, sigma, K, feats = cases.cond_inputs(name)
        model = ref_import.build_cond(ns, d, L, I, H, feats, context_dims=cdims)
        sd = {"backbone.model." + k: v for k, v in p.items()}
        sd["reparam.uvl_mean"] = p["reparam.uvl_mean"]
        sd["reparam.uvl_std"] = p["reparam.uvl_std"]
        model.load_state_dict(sd, strict=True)
        ctx = ns.Context3d(image=torch.zeros(len(sigma), 3, hw, hw), K=K)
        den_ref = model(x, sigma, ctx)
        c_skip, c_out, c_in, c_noise = cpu_ref.edm_coeffs(sigma)
        F_ref, _ = model.backbone.model(c_in * x, c_noise, ctx, model.conditioner(ctx))
        den, F_x = cpu_ref.cond_denoiser(p, "", H, K, feats)(x, sigma, return_raw=True)
        check("denoised", den, den_ref, tol=5e-5)
        check("F_x", F_x, F_ref, tol=5e-5)
        save(name, denoised=den_ref, F_x=F_ref)

    # ---- (6) reparam round trips
    print("reparam")
    feats, K, geom, um, us = cases.lookup_inputs("lookup_small")
    rp = ns.reparam_mod.UVLReparam(um, us)
    ctx = ns.Context3d(image=torch.zeros(1), K=K)
    xyz_ref = rp.diffusion_to_data(geom, ctx)
    back_ref = rp.data_to_diffusion(xyz_ref, ctx)
    check("uvl diffusion_to_data", cpu_ref.uvl_diffusion_to_data(geom, K, um, us), xyz_ref)
    check("uvl data_to_diffusion", cpu_ref.uvl_data_to_diffusion(xyz_ref, K, um, us), back_ref)
    gr = ns.reparam_mod.GaussianReparam(torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA))
    g_ref = gr.diffusion_to_data(geom, None)
    check("gaussian", cpu_ref.gaussian_diffusion_to_data(geom, gr.mean, gr.sigma), g_ref)
    save("reparam", uvl_xyz=xyz_ref, uvl_back=back_ref, gauss=g_ref)

    # ---- (7) t_steps + (8) sampler trajectory with injected noise
    print("sampler")
    c = cases.SAMPLER_CASE
    p, latents, noises = cases.sampler_inputs()
    model = ref_import.build_uncond(ns, c["d"], c["L"], I, H, sigma_max=c["sigma_max"])
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.mean"] = torch.tensor(cases.GAUSS_MEAN)
    sd["reparam.sigma"] = torch.tensor(cases.GAUSS_SIGMA)
    model.load_state_dict(sd, strict=True)
    ts64 = model.t_steps(64, 165.0, 0.002, 7)
    ts128 = model.t_steps(128, 165.0, 0.002, 7)
    check("t_steps(64)", cpu_ref.t_steps(64, 165.0, 0.002, 7), ts64, tol=1e-14)
    check("t_steps(128)", cpu_ref.t_steps(128, 165.0, 0.002, 7), ts128, tol=1e-14)
    D.torch = ref_import.TorchRandnProxy([latents] + noises)
    try:
        samp_ref = model.sample_stochastic((c["B"], c["N"], 3), None, num_steps=c["num_steps"])
    finally:
        D.torch = torch
    Dor = cpu_ref.uncond_denoiser(p, "", H)
    x_next = cpu_ref.sample_stochastic(Dor, latents, noises, c["num_steps"], c["sigma_max"])
    samp = cpu_ref.gaussian_diffusion_to_data(x_next, torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA))
    check("sample_stochastic (6 steps, 11 evals)", samp, samp_ref, tol=1e-4)
    save("sampler", t_steps_64=ts64, t_steps_128=ts128, sample=samp_ref)

    # ---- (9) upsample with injected noise
    print("upsample")
    c = cases.UPSAMPLE_CASE
    p, data = cases.upsample_inputs()
    model = ref_import.build_uncond(ns, c["d"], c["L"], I, H, sigma_max=c["sigma_max"])
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.mean"] = torch.tensor(cases.GAUSS_MEAN)
    sd["reparam.sigma"] = torch.tensor(cases.GAUSS_SIGMA)
    model.load_state_dict(sd, strict=True)
    draws = cases.upsample_draw_list()
    D.torch = ref_import.TorchRandnProxy(draws)
    try:
        up_ref = model.upsample(data, n_new=c["n_new"], num_steps=c["num_steps"], num_substeps=c["num_substeps"])
        assert D.torch._i == len(draws), (D.torch._i, len(draws))
    finally:
        D.torch = torch
    it = iter(draws)
    new_latents = next(it)
    gm, gs = torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA)
    Dor = cpu_ref.uncond_denoiser(p, "", H)
    up = cpu_ref.upsample(Dor, cpu_ref.gaussian_data_to_diffusion(data, gm, gs), new_latents,
                          lambda shape: next(it), c["num_steps"], c["sigma_max"], c["num_substeps"])
    up = cpu_ref.gaussian_diffusion_to_data(up, gm, gs)
    check("upsample", up, up_ref, tol=1e-4)
    save("upsample", upsampled=up_ref)

    # ---- (10) EDMLoss value + a few gradients (for the backward kernels)
    print("loss")
    torch.set_grad_enabled(True)
    c = cases.LOSS_CASE
    p, ex, u, noise = cases.loss_inputs()
    model = ref_import.build_uncond(ns, c["d"], c["L"], I, H, sigma_max=c["sigma_max"], mean=(0., 0., 0.), sigma=(1., 1., 1.))
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.mean"] = torch.zeros(3)
    sd["reparam.sigma"] = torch.ones(3)
    model.load_state_dict(sd, strict=True)
    B = c["B"]
    D.torch = ref_import.TorchRandnProxy([])
    D.torch.rand = lambda *a, **k: u.clone()
    D.torch.randn_like = lambda t: noise.clone()
    try:
        loss_ref = model.loss(model, ex, None)
    finally:
        D.torch = torch
    loss_ref.backward()
    grads = {k: v.grad.clone() for k, v in model.named_parameters()}
    pg = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    sigma = cpu_ref.log_uniform_sigma(u, c["sigma_max"])
    loss = cpu_ref.edm_loss(cpu_ref.uncond_denoiser(pg, "", H), ex, sigma, noise)
    loss.backward()
    check("loss", loss.detach(), loss_ref.detach())
    sel = ["lift.weight", "inner.layers.0.mlp.2.bias", "inner.layers.1.broadcast.pool.inducers",
           "inner.layers.0.broadcast_norm.scale.weight", "inner.layers.1.broadcast.unpool.in_proj_weight",
           "inner.layers.0.mlp.1.alpha", "lower.1.weight"]
    out = {"loss": loss_ref.detach()}
    for k in sel:
        check("grad " + k, pg[k].grad, grads["backbone.model." + k], tol=2e-4)
        out["grad." + k] = grads["backbone.model." + k]
    save("loss", **out)

    # ---- (10) conditional training step: EDMLoss through RayNetwork, gradients into parameters AND the pyramid
    print("cond_loss")
    c = cases.COND_LOSS_CASE
    d, L, N, hw, cdims, seed = cases.COND_CASES[c["name"]]
    p, ex_diff, u, noise, K, feats = cases.cond_loss_inputs()
    feats_ref = [f.clone().requires_grad_(True) for f in feats]
    model = ref_import.build_cond(ns, d, L, I, H, feats_ref, context_dims=cdims, sigma_max=c["sigma_max"])
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.uvl_mean"], sd["reparam.uvl_std"] = p["reparam.uvl_mean"], p["reparam.uvl_std"]
    model.load_state_dict(sd, strict=True)
    ctx = ns.Context3d(image=torch.zeros(len(u), 3, hw, hw), K=K)
    with torch.no_grad():
        ex_data = model.reparam.diffusion_to_data(ex_diff, ctx)
    D.torch = ref_import.TorchRandnProxy([])
    D.torch.rand = lambda *a, **k: u.clone()
    D.torch.randn_like = lambda t: noise.clone()
    try:
        loss_ref = model.loss(model, ex_data, ctx)
    finally:
        D.torch = torch
    loss_ref.backward()
    grads = {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}
    pg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "reparam" not in k else v) for k, v in p.items()}
    fg = [f.clone().requires_grad_(True) for f in feats]
    sigma = cpu_ref.log_uniform_sigma(u, c["sigma_max"])
    with torch.no_grad():
        ex_back = cpu_ref.uvl_data_to_diffusion(ex_data, K, p["reparam.uvl_mean"], p["reparam.uvl_std"])
    loss = cpu_ref.edm_loss(cpu_ref.cond_denoiser(pg, "", H, K, fg), ex_back, sigma, noise)
    loss.backward()
    check("cond loss", loss.detach(), loss_ref.detach(), tol=5e-5)
    sel = ["xyz_embed.weight", "img_feature_proj.1.weight", "img_feature_proj.1.bias", "output_proj.1.weight",
           "output_proj.1.bias", "backbone.layers.0.mlp.2.bias", "backbone.layers.1.broadcast.pool.inducers"]
    out = {"loss": loss_ref.detach(), "ex_data": ex_data}
    for k in sel:
        check("grad " + k, pg[k].grad, grads["backbone.model." + k], tol=5e-4)
        out["grad." + k] = grads["backbone.model." + k]
    for l, (a, b) in enumerate(zip(fg, feats_ref)):
        check(f"grad features[{l}]", a.grad, b.grad, tol=5e-4)
        out[f"grad.features.{l}"] = b.grad
    save("cond_loss", **out)

    # ---- the reference's default activation, nn.ReLU
    print("relu")
    with torch.no_grad():
        c = cases.RELU_CASE
        p, x, sigma = cases.relu_inputs()
        net = ns.LinearLift(inner=ns.SetTransformer(n_layers=c["L"], num_inducers=I, feature_dim=c["d"], t_embed_dim=1,
                                                    num_heads=H),   # activation left at its default
                            feature_dim=c["d"])
        model = D.Diffusion(backbone=D.EDMPrecond(model=net), conditioner=D.IdleConditioner(),
                            reparam=ns.reparam_mod.GaussianReparam(torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA)),
                            loss=D.EDMLoss(schedule=D.LogUniformSchedule(max=165.0)))
        assert isinstance(net.inner.layers[0].mlp[1], torch.nn.ReLU)
        sd = {"backbone.model." + k: v for k, v in p.items()}
        sd["reparam.mean"], sd["reparam.sigma"] = torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA)
        model.load_state_dict(sd, strict=True)
        den_ref = model(x, sigma, None)
        den, raw = cpu_ref.uncond_denoiser(p, "", H)(x, sigma, return_raw=True)
        check("relu denoised", den, den_ref)
        save("relu_d128_L2_N256", denoised=den_ref)
    print("all golden vectors written")


if __name__ == "__main__":
    main()
