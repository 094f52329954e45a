"""The oracle (oracle/cpu_ref.py) against the golden vectors captured from the real reference by
tools/make_golden.py.  CPU only.  Tolerance: fp32 noise floor of a re-ordered op sequence."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, cpu_ref

TOL = 2e-5


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


def _close(got, ref, tol=TOL):
    e_max, e_l2 = cpu_ref.rel_err(got, ref)
    assert e_max <= tol, (e_max, e_l2)


@pytest.mark.parametrize("name", list(cases.LAYER_CASES))
def test_layer(golden_dir, name):
    g = _load(golden_dir, name)
    p, x, t = cases.layer_inputs(name)
    with torch.no_grad():
        y, h = cpu_ref.broadcasting_layer(x, t, p, "", cases.H)
        N = x.shape[1]
        y2, _ = cpu_ref.broadcasting_layer(x[:, : N // 2] * 0.5, t, p, "", cases.H, h=g["h"])
    _close(y, g["x_out"])
    _close(h, g["h"])
    _close(y2, g["x_out_cached"])


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_uncond(golden_dir, name):
    g = _load(golden_dir, name)
    p, x, sigma = cases.uncond_inputs(name)
    with torch.no_grad():
        den, F_x = cpu_ref.uncond_denoiser(p, "", cases.H)(x, sigma, return_raw=True)
    _close(den, g["denoised"])
    _close(F_x, g["F_x"])


def test_cached(golden_dir):
    g = _load(golden_dir, "cached_d128_L4")
    p, x, sigma, x_new = cases.cached_inputs()
    D = cpu_ref.uncond_denoiser(p, "", cases.H)
    with torch.no_grad():
        _, cache = D(x, sigma, do_cache=True)
        out = D(x_new, sigma, cache=list(g["cache"]))
    _close(torch.stack(cache), g["cache"])
    _close(out, g["out_new"])


@pytest.mark.parametrize("name", list(cases.LOOKUP_CASES))
def test_lookup(golden_dir, name):
    g = _load(golden_dir, name)
    feats, K, geom, um, us = cases.lookup_inputs(name)
    got = cpu_ref.extract_image_features(geom, feats, K, um, us)
    _close(got, g["lookup"], 5e-5)
    # the tap-by-tap restatement against torch's own grid_sample kernel
    got2 = cpu_ref.extract_image_features(geom, feats, K, um, us, use_torch_grid_sample=True)
    _close(got, got2, 5e-5)
    # out-of-bounds taps really are exercised
    xyz = cpu_ref.uvl_diffusion_to_data(geom, K, um, us)
    uv = cpu_ref.project_points(xyz, K)
    Hh, Ww = feats[0].shape[-2:]
    x0, y0, _, _ = cpu_ref.bilinear_taps(uv, Hh, Ww)
    oob = (x0 < 0) | (x0 + 1 > Ww - 1) | (y0 < 0) | (y0 + 1 > Hh - 1)
    assert oob.any() and (~oob).any()


@pytest.mark.parametrize("name", list(cases.COND_CASES))
def test_cond(golden_dir, name):
    g = _load(golden_dir, name)
    p, x, sigma, K, feats = cases.cond_inputs(name)
    with torch.no_grad():
        den, F_x = cpu_ref.cond_denoiser(p, "", cases.H, K, feats)(x, sigma, return_raw=True)
    _close(den, g["denoised"], 5e-5)
    _close(F_x, g["F_x"], 5e-5)


def test_reparam(golden_dir):
    g = _load(golden_dir, "reparam")
    feats, K, geom, um, us = cases.lookup_inputs("lookup_small")
    xyz = cpu_ref.uvl_diffusion_to_data(geom, K, um, us)
    _close(xyz, g["uvl_xyz"], 1e-6)
    _close(cpu_ref.uvl_data_to_diffusion(g["uvl_xyz"], K, um, us), g["uvl_back"], 1e-6)
    gm, gs = torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA)
    _close(cpu_ref.gaussian_diffusion_to_data(geom, gm, gs), g["gauss"], 1e-7)
    # round trip (find_hyperparameters.ipynb sanity check): data -> diffusion -> data
    _close(cpu_ref.gaussian_data_to_diffusion(g["gauss"], gm, gs), geom, 1e-5)


def test_t_steps_and_sampler(golden_dir):
    g = _load(golden_dir, "sampler")
    assert torch.equal(cpu_ref.t_steps(64, 165.0, 0.002, 7), g["t_steps_64"])
    assert torch.equal(cpu_ref.t_steps(128, 165.0, 0.002, 7), g["t_steps_128"])
    c = cases.SAMPLER_CASE
    p, latents, noises = cases.sampler_inputs()
    with torch.no_grad():
        x = cpu_ref.sample_stochastic(cpu_ref.uncond_denoiser(p, "", cases.H), latents, noises,
                                      c["num_steps"], c["sigma_max"])
    x = cpu_ref.gaussian_diffusion_to_data(x, torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA))
    assert x.dtype == torch.float64
    _close(x, g["sample"], 1e-4)


def test_upsample(golden_dir):
    g = _load(golden_dir, "upsample")
    c = cases.UPSAMPLE_CASE
    p, data = cases.upsample_inputs()
    it = iter(cases.upsample_draw_list())
    gm, gs = torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA)
    with torch.no_grad():
        up = cpu_ref.upsample(cpu_ref.uncond_denoiser(p, "", cases.H),
                              cpu_ref.gaussian_data_to_diffusion(data, gm, gs), next(it),
                              lambda shape: next(it), c["num_steps"], c["sigma_max"], c["num_substeps"])
    _close(cpu_ref.gaussian_diffusion_to_data(up, gm, gs), g["upsampled"], 1e-4)


def test_loss_and_grads(golden_dir):
    g = _load(golden_dir, "loss")
    c = cases.LOSS_CASE
    p, ex, u, noise = cases.loss_inputs()
    pg = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    sigma = cpu_ref.log_uniform_sigma(u, c["sigma_max"])
    loss = cpu_ref.edm_loss(cpu_ref.uncond_denoiser(pg, "", cases.H), ex, sigma, noise)
    loss.backward()
    _close(loss.detach(), g["loss"], 1e-6)
    for k, v in g.items():
        if k.startswith("grad."):
            _close(pg[k[5:]].grad, v, 2e-4)


@pytest.mark.parametrize("name", list(cases.SETMETRIC_CASES))
def test_set_metrics(golden_dir, name):
    """Set-vs-set distances and 1-NNA / MMD / COV (gecco-jax benchmark.py:21-39, 128-156): the oracle's restatement against the
    values the reference's OWN numpy methods produced (tools/make_golden_setmetrics.py executed them from /root/reference)."""
    g = np.load(os.path.join(golden_dir, "setmetrics.npz"))
    samples, data = cases.setmetric_inputs(name)
    for kind, sq in (("chamfer", False), ("chamfer_squared", True)):
        tag = f"{name}/{kind}"
        ss = cpu_ref.set_pairwise_distance(samples.double(), samples.double(), sq).numpy()
        sd = cpu_ref.set_pairwise_distance(samples.double(), data.double(), sq).numpy()
        dd = cpu_ref.set_pairwise_distance(data.double(), data.double(), sq).numpy()
        for m, key in ((ss, "ss"), (sd, "sd"), (dd, "dd")):
            assert np.abs(m - g[f"{tag}/{key}"]).max() <= 1e-6 * g[f"{tag}/{key}"].max()
        got = cpu_ref.set_metrics(g[f"{tag}/ss"], g[f"{tag}/sd"], g[f"{tag}/dd"])
        assert [got["1-nn"], got["mmd"], got["cov"]] == pytest.approx(list(g[f"{tag}/metrics"]), rel=1e-6)
