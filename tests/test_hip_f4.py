"""SURVEY.md 8(f) row 4 on the GPU: the deterministic ODE sampler, the inpainting sampler and the Chamfer / EMD metrics of
gecco-jax, on the HIP path against oracle/cpu_ref.py restatements.  gecco-jax cannot be imported (jax is absent from the
image), so these are "parity unpinned" against the JAX code itself; the ODE sampler is additionally pinned through the
torch reference: it is the stochastic sampler with S_churn = 0, whose golden trajectory test lives in test_hip_modules."""
import numpy as np
import pytest
import torch

from oracle import cases, cpu_ref
from tests.test_modules_cpu import build_uncond, uncond_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    import __graft_entry__ as ge
    ge.build()
    c = cases.SAMPLER_CASE
    p, _, _ = cases.sampler_inputs()
    m = build_uncond(c["d"], c["L"], sigma_max=c["sigma_max"])
    m.load_state_dict(uncond_state_dict(p), strict=True)
    return m.cuda().eval(), p


def _rn(seed, *shape):
    return torch.from_numpy(np.random.RandomState(seed).randn(*shape).astype(np.float32))


def test_ode_sampler_is_the_churn_free_heun_loop(model):
    m, p = model
    c = cases.SAMPLER_CASE
    B, N, steps = c["B"], c["N"], c["num_steps"]
    latents = _rn(5, B, N, 3)
    D = cpu_ref.uncond_denoiser(p, "", cases.H)
    with torch.no_grad():
        ref = cpu_ref.sample_stochastic(D, latents, [torch.zeros(B, N, 3)] * steps, steps, c["sigma_max"], S_churn=0.0)
    ref = cpu_ref.gaussian_diffusion_to_data(ref, torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA))
    for use_graph in (True, False):
        out = m.sample_ode((B, N, 3), None, latents=latents.cuda(), num_steps=steps, use_graph=use_graph)
        e = cpu_ref.rel_err(out.cpu(), ref)
        print("ODE sampler vs oracle", "graph" if use_graph else "eager", e)
        assert e[0] < 1e-4, e
    # deterministic: no generator state is consumed by the steps
    a = m.sample_ode((B, N, 3), None, latents=latents.cuda(), num_steps=steps)
    assert torch.equal(a, out)


@pytest.mark.parametrize("num_substeps", [1, 2])
def test_inpainting_sampler_vs_oracle(model, num_substeps):
    m, p = model
    c = cases.SAMPLER_CASE
    B, n, mm, steps = 2, 40, 24, 4
    known = _rn(11, B, n, 3) * torch.tensor(cases.GAUSS_SIGMA) + torch.tensor(cases.GAUSS_MEAN)
    draws, seed = [_rn(100, B, mm + n, 3)], 101
    for i in range(steps):
        for j in range(num_substeps):
            draws.append(_rn(seed, B, n, 3)); seed += 1
            draws.append(_rn(seed, B, mm + n, 3)); seed += 1
            if j < num_substeps - 1:
                draws.append(_rn(seed, B, mm + n, 3)); seed += 1
    known_diff = cpu_ref.gaussian_data_to_diffusion(known, torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA))
    D = cpu_ref.uncond_denoiser(p, "", cases.H)
    with torch.no_grad():
        ref = cpu_ref.sample_inpaint(D, known_diff, mm, draws, steps, num_substeps, c["sigma_max"])
    ref = cpu_ref.gaussian_diffusion_to_data(ref, torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA))
    out = m.sample_inpaint(known.cuda(), mm, None, num_substeps=num_substeps, noise=draws, num_steps=steps)
    assert out.shape == (B, mm, 3)
    e = cpu_ref.rel_err(out.cpu(), ref)
    print("inpainting sampler vs oracle, substeps", num_substeps, e)
    assert e[0] < 2e-4, e
    # the generator path runs and keeps the known points out of the result
    out2 = m.sample_inpaint(known.cuda(), mm, None, num_substeps=num_substeps, num_steps=steps, seed=3)
    assert out2.shape == (B, mm, 3) and torch.isfinite(out2).all()


@pytest.mark.parametrize("N,M", [(2048, 2048), (300, 517)])
def test_chamfer_and_distance_matrix(N, M):
    from gecco_amd import metrics
    a, b = _rn(1, 3, N, 3), _rn(2, 3, M, 3) * 1.1 + 0.05
    for squared in (False, True):
        d = metrics.distance_matrix(a.cuda(), b.cuda(), squared=squared)
        ref = cpu_ref.distance_matrix(a.double(), b.double(), squared)
        # |a|^2 + |b|^2 - 2 a.b in fp32: absolute error ~1e-6 of the squared norms (and its root near zero distance)
        assert (d.cpu().double() - ref).abs().max() < (2e-5 if squared else 2e-3)
        cd = metrics.chamfer_distance(a.cuda(), b.cuda(), squared=squared)
        cref = cpu_ref.chamfer_distance(a.double(), b.double(), squared)
        e = ((cd.cpu().double() - cref).abs() / cref).max()
        print("chamfer", N, M, squared, float(e))
        assert e < 1e-4
    # a cloud against itself: zero, and symmetric in its arguments
    assert metrics.chamfer_distance(a.cuda(), a.cuda()).abs().max() < 2e-3
    assert torch.allclose(metrics.chamfer_distance(a[:, :M].cuda(), b[:, :N].cuda()), metrics.chamfer_distance(b[:, :N].cuda(), a[:, :M].cuda()))
    # single-cloud form
    assert metrics.chamfer_distance(a[0].cuda(), b[0].cuda()).shape == ()


def test_emd_exact_and_sinkhorn():
    from scipy.optimize import linear_sum_assignment
    from gecco_amd import metrics
    N = 256
    a, b = _rn(3, 2, N, 3), _rn(4, 2, N, 3) * 0.9
    emd = metrics.scipy_emd(a.cuda(), b.cuda())
    for i in range(2):
        d = cpu_ref.distance_matrix(a[i].double(), b[i].double()).numpy()
        r, c = linear_sum_assignment(d)
        assert abs(float(emd[i]) - d[r, c].mean()) < 1e-4
    # a permuted copy: the assignment finds it
    perm = torch.randperm(N)
    assert metrics.scipy_emd(a.cuda(), a[:, perm].cuda()).abs().max() < 2e-3
    for eps in (0.05, 0.01):
        sk = metrics.sinkhorn_emd(a.cuda(), b.cuda(), epsilon=eps, iterations=100)
        ref = cpu_ref.sinkhorn_cost(cpu_ref.distance_matrix(a.double(), b.double(), squared=True), eps, 100)
        e = ((sk.cpu().double() - ref).abs() / ref).max()
        print("sinkhorn eps", eps, sk.tolist(), ref.tolist(), float(e))
        assert e < 2e-3
    # more entropy, more blur: the transport cost grows with epsilon
    assert (metrics.sinkhorn_emd(a.cuda(), b.cuda(), epsilon=0.05, iterations=100) >
            metrics.sinkhorn_emd(a.cuda(), b.cuda(), epsilon=0.01, iterations=100)).all()


@pytest.mark.parametrize("name", list(cases.SETMETRIC_CASES))
def test_set_metrics_golden(name):
    """Set-vs-set Chamfer distances (every pair of two sets of clouds, one HIP kernel per direction, no N x M matrix) and 1-NN
    accuracy / MMD / coverage on the device against tests/golden/setmetrics.npz — values the reference's own numpy methods produced
    (gecco-jax benchmark.py:21-39, 128-156; tools/make_golden_setmetrics.py)."""
    import os
    from gecco_amd import metrics
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "setmetrics.npz"))
    samples, data = (t.cuda() for t in cases.setmetric_inputs(name))
    for kind in ("chamfer", "chamfer_squared"):
        tag = f"{name}/{kind}"
        ss = metrics.pairwise_set_distance(samples, samples, kind)
        sd = metrics.pairwise_set_distance(samples, data, kind)
        dd = metrics.pairwise_set_distance(data, data, kind)
        for m, key in ((ss, "ss"), (sd, "sd"), (dd, "dd")):
            ref = torch.from_numpy(g[f"{tag}/{key}"])
            diff = (m.cpu() - ref).abs()
            if key != "sd":   # a cloud against itself: fp32 cancellation noise of |a|^2 + |b|^2 - 2 a.b under the root (the golden matrix is
                assert diff.diagonal().max().item() <= 1e-3   # float64: exactly 0); `_one_nn_acc` overwrites the diagonal anyway
                diff.fill_diagonal_(0.0)
            assert diff.max().item() <= 2e-5 * ref.max().item(), (tag, key)
        # the metrics on the GOLDEN matrices (bit-identical inputs: the integer-valued outputs must be equal), then end to end
        dev = {k: torch.from_numpy(g[f"{tag}/{k}"]).cuda() for k in ("ss", "sd", "dd")}
        got = metrics.set_metrics(dev["ss"], dev["sd"], dev["dd"])
        want = g[f"{tag}/metrics"]
        assert float(got["1-nn"]) == pytest.approx(want[0], abs=1e-7) and float(got["cov"]) == pytest.approx(want[2], abs=1e-7)
        assert float(got["mmd"]) == pytest.approx(want[1], rel=1e-6)
        e2e = metrics.evaluate_sets(samples, data, kind)
        assert float(e2e["mmd"]) == pytest.approx(want[1], rel=1e-4)


def test_set_distance_shapes_and_symmetry():
    """Ragged shapes (S != T, N != M, clouds that do not fill a thread's 8 points or a 2048-point LDS tile), the pairwise entry of
    the matrix against the per-pair kernel, symmetry of a set against itself."""
    from gecco_amd import metrics
    rs = np.random.RandomState(5)
    a = torch.from_numpy(rs.randn(5, 300, 3).astype(np.float32)).cuda()
    b = torch.from_numpy(rs.randn(9, 2500, 3).astype(np.float32)).cuda()
    for kind in ("chamfer", "chamfer_squared"):
        D = metrics.pairwise_set_distance(a, b, kind)
        assert D.shape == (5, 9)
        ref = cpu_ref.set_pairwise_distance(a.cpu().double(), b.cpu().double(), kind == "chamfer_squared")
        assert (D.cpu().double() - ref).abs().max().item() <= 2e-5 * ref.max().item()
        for s, t in ((0, 0), (4, 8), (2, 5)):
            one = metrics.chamfer_distance(a[s], b[t], squared=kind == "chamfer_squared")
            assert abs(float(one) - float(D[s, t])) <= 2e-6 * float(one)
        Daa = metrics.pairwise_set_distance(a, a, kind)
        assert torch.equal(Daa, Daa.t().contiguous()) or (Daa - Daa.t()).abs().max().item() <= 1e-6 * Daa.max().item()
        assert Daa.diagonal().abs().max().item() <= 1e-3   # a cloud against itself: the clamp hides the cancellation noise of |a|^2 + |b|^2 - 2ab
    E = metrics.pairwise_set_distance(a[:3, :64], b[:2, :64], "emd", block_size=2)
    assert E.shape == (3, 2) and torch.isfinite(E).all()


def test_evaluate_logp_vs_oracle(model):
    """`Diffusion.evaluate_logp` (gecco-jax models/diffusion.py:446-540: Heun on (x, delta) from sigma_min to sigma_max with Hutchinson's
    divergence estimate, + the prior's log-density at the latent + the reparametrisation's log-determinant) on the HIP path — the
    denoiser's vector-Jacobian products through the autograd Functions — against the oracle's restatement on torch autograd, with the
    same Rademacher probes; each part separately.  Deterministic for a seed; the model's parameters keep requires_grad."""
    from gecco_amd.diffusion import karras_t_steps
    m, p = model
    c = cases.SAMPLER_CASE
    B, N, steps = c["B"], c["N"], 8
    mean, sig = torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA)
    data = _rn(21, B, N, 3) * sig * 0.8 + mean
    probes = torch.from_numpy(np.random.RandomState(22).randint(0, 2, size=(2, B, N, 3)).astype(np.float32)) * 2 - 1
    ts = karras_t_steps(steps, c["sigma_max"], 0.002, 7.0)[:steps].flip(0)
    D = cpu_ref.uncond_denoiser(p, "", cases.H)
    ladj = torch.full((B,), -float(N) * float(torch.log(sig.double()).sum()), dtype=torch.float64)
    x0 = (data - mean) / sig
    ref, prior_ref, delta_ref, lat_ref = cpu_ref.evaluate_logp(D, x0, probes, ts, c["sigma_max"], ladj)
    out = m.evaluate_logp(data.cuda(), None, probes=probes.cuda(), num_steps=steps, sigma_min=0.002, rho=7.0, return_details=True)
    print("logp", out["logp"].cpu().tolist(), "oracle", ref.tolist(), "| delta", out["delta_jacobian"].cpu().tolist(), delta_ref.tolist())
    assert cpu_ref.rel_err(out["latent"].cpu(), lat_ref)[0] < 1e-4
    assert torch.allclose(out["prior_logp"].cpu(), prior_ref, rtol=1e-4, atol=1e-3)
    assert torch.allclose(out["delta_reparam"].cpu(), ladj, rtol=1e-6)
    assert torch.allclose(out["delta_jacobian"].cpu(), delta_ref, rtol=1e-4, atol=1e-2), (out["delta_jacobian"], delta_ref)   # measured: 3e-8
    assert torch.allclose(out["logp"].cpu(), ref, rtol=1e-4, atol=1e-2)
    assert out["trajectory_diff"].shape == (steps, B, N, 3)
    a = m.evaluate_logp(data.cuda(), None, n_trace_samples=2, seed=7, num_steps=steps)
    assert torch.equal(a, m.evaluate_logp(data.cuda(), None, n_trace_samples=2, seed=7, num_steps=steps)) and a.shape == (B,)
    assert all(q.requires_grad for q in m.parameters())


def test_evaluate_logp_conditional_uvl_vs_oracle():
    """evaluate_logp on the image-conditional model (RayNetwork: the vector-Jacobian product runs through the projective lookup's geometry
    gradient, `ray_lookup_dgeom_kernel`) with the UVL reparametrisation's closed-form log-determinant, against the oracle's restatement
    (cond_denoiser under torch autograd, the Jacobian-by-autograd log-determinant)."""
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd.diffusion import Conditioner, karras_t_steps
    from gecco_amd.models.feature_pyramid import FeaturePyramidContext
    from gecco_amd.structs import Context3d
    from tests.test_modules_cpu import build_cond
    name = "cond_d128_L2_N96"
    d, L, N, hw, cdims, seed = cases.COND_CASES[name]
    p, x, sigma, K, feats = cases.cond_inputs(name)
    B = 2
    K, feats = K[:B], [f[:B] for f in feats]

    class FixedPyramid(Conditioner):
        def forward(self, raw_ctx):
            return FeaturePyramidContext(features=[f.cuda() for f in feats], K=raw_ctx.K)

    m = build_cond(d, L, cdims, conditioner=FixedPyramid())
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.uvl_mean"], sd["reparam.uvl_std"] = p["reparam.uvl_mean"], p["reparam.uvl_std"]
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    mean, std = p["reparam.uvl_mean"], p["reparam.uvl_std"]
    # data in front of the camera: the images of moderate diffusion-space points
    z0 = _rn(31, B, N, 3) * 0.6
    data = cpu_ref.uvl_diffusion_to_data(z0, K, mean, std)
    steps = 5
    probes = torch.from_numpy(np.random.RandomState(32).randint(0, 2, size=(1, B, N, 3)).astype(np.float32)) * 2 - 1
    ts = karras_t_steps(steps, 165.0, 0.002, 7.0)[:steps].flip(0)
    D = cpu_ref.cond_denoiser(p, "", cases.H, K, feats)
    x0 = cpu_ref.uvl_data_to_diffusion(data, K, mean, std)
    ladj = torch.zeros(B, dtype=torch.float64)
    for b in range(B):
        for n in range(N):
            f = lambda q: cpu_ref.uvl_data_to_diffusion(q[None, None], K[b:b + 1].double(), mean.double(), std.double())[0, 0]
            ladj[b] += torch.linalg.slogdet(torch.autograd.functional.jacobian(f, data[b, n].double()))[1]
    ref, prior_ref, delta_ref, lat_ref = cpu_ref.evaluate_logp(D, x0, probes, ts, 165.0, ladj)
    ctx = Context3d(image=torch.zeros(B, 3, hw, hw).cuda(), K=K.cuda())
    out = m.evaluate_logp(data.cuda(), ctx, probes=probes.cuda(), num_steps=steps, sigma_max=165.0, sigma_min=0.002, rho=7.0, return_details=True)
    print("conditional logp", out["logp"].cpu().tolist(), "oracle", ref.tolist(), "| reparam", out["delta_reparam"].cpu().tolist(), ladj.tolist())
    assert torch.allclose(out["delta_reparam"].cpu(), ladj, rtol=1e-4, atol=1e-2)
    assert cpu_ref.rel_err(out["latent"].cpu(), lat_ref)[0] < 1e-3
    assert torch.allclose(out["delta_jacobian"].cpu(), delta_ref, rtol=2e-3, atol=5e-2), (out["delta_jacobian"], delta_ref)
    assert torch.allclose(out["logp"].cpu(), ref, rtol=2e-3, atol=1e-1)
