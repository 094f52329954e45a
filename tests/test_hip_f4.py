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
