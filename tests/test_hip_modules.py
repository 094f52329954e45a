"""GPU parity of the drop-in module API (gecco_amd.*) — the tests read like uses of the reference's modules:
construct, load_state_dict, call — against the golden vectors from the real reference and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, cpu_ref
from oracle import weights as W
from tests.test_modules_cpu import build_cond, build_uncond, uncond_state_dict

pytestmark = pytest.mark.gpu
TOL = 5e-5


@pytest.fixture(scope="module", autouse=True)
def _build():
    import __graft_entry__ as ge
    ge.build()


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


def _close(got, ref, tol=TOL):
    e = cpu_ref.rel_err(got.cpu(), ref)
    assert e[0] <= tol, e
    return e


# ------------------------------------------------------------------------------------------- unit modules
def test_unit_modules_vs_oracle():
    from gecco_amd.models.activation import GaussianActivation
    from gecco_amd.models.mlp import MLP
    from gecco_amd.models.normalization import AdaGN
    from gecco_amd.models.set_transformer import AttentionPool, Broadcast
    rs = np.random.RandomState(0)
    d, N, B = 128, 200, 2
    p = W.layer_state_dict(rs, d, cases.I, cases.H)
    x = torch.from_numpy(rs.randn(B, N, d).astype(np.float32))
    t = torch.tensor([[[-0.7]], [[0.9]]])
    with torch.no_grad():
        act = GaussianActivation().cuda()
        act.alpha.fill_(0.8)
        _close(act(x.cuda()), cpu_ref.gaussian_activation(x, torch.tensor(0.8)))

        norm = AdaGN(d, 1).cuda()
        norm.load_state_dict({k[len("mlp_norm."):]: v for k, v in p.items() if k.startswith("mlp_norm.")})
        _close(norm(x.cuda(), t.cuda()), cpu_ref.adagn(x, t, p, "mlp_norm."))

        mlp = MLP(d, d, 2 * d, activation=GaussianActivation).cuda()
        mlp.load_state_dict({k[len("mlp."):]: v for k, v in p.items() if k.startswith("mlp.")})
        _close(mlp(x.cuda()), cpu_ref.mlp(x, p, "mlp."))

        pool = AttentionPool(d, cases.H, cases.I).cuda()
        pool.load_state_dict({k[len("broadcast.pool."):]: v for k, v in p.items() if k.startswith("broadcast.pool.")})
        _close(pool(x.cuda()), cpu_ref.attention_pool(x, p, "broadcast.pool.", cases.H))

        bc = Broadcast(d, cases.I, 1, cases.H, activation=GaussianActivation).cuda()
        bc.load_state_dict({k[len("broadcast."):]: v for k, v in p.items() if k.startswith("broadcast.")})
        out, h = bc(x.cuda(), t.cuda(), return_h=True)
        ref_out, ref_h = cpu_ref.broadcast(x, t, p, "broadcast.", cases.H)
        _close(out, ref_out)
        _close(h, ref_h)


@pytest.mark.parametrize("name", list(cases.LAYER_CASES))
def test_broadcasting_layer_module(golden_dir, name):
    from gecco_amd.models.activation import GaussianActivation
    from gecco_amd.models.set_transformer import BroadcastingLayer
    g = _load(golden_dir, name)
    p, x, t = cases.layer_inputs(name)
    d = x.shape[-1]
    layer = BroadcastingLayer(feature_dim=d, num_inducers=cases.I, embed_dim=1, num_heads=cases.H,
                              activation=GaussianActivation)
    layer.load_state_dict(p, strict=True)
    layer = layer.cuda()
    with torch.no_grad():
        y, h = layer(x.cuda(), t.cuda(), return_h=True)
    _close(y, g["x_out"])
    _close(h, g["h"])


# ------------------------------------------------------------------------------------------- Diffusion
@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_diffusion_forward_golden(golden_dir, name):
    g = _load(golden_dir, name)
    d, L, N, seed = cases.UNCOND_CASES[name]
    p, x, sigma = cases.uncond_inputs(name)
    m = build_uncond(d, L)
    m.load_state_dict(uncond_state_dict(p), strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        den = m(x.cuda(), sigma.cuda(), None)
    _close(den, g["denoised"])
    if name == "uncond_d128_L4_N256":
        gc = _load(golden_dir, "cached_d128_L4")
        _, _, _, x_new = cases.cached_inputs(name)
        with torch.no_grad():
            den2, cache = m(x.cuda(), sigma.cuda(), None, do_cache=True)
            out_new = m(x_new.cuda(), sigma.cuda(), None, cache=cache)
        assert torch.equal(den2, den)
        _close(torch.stack(cache), gc["cache"])
        _close(out_new, gc["out_new"])


def test_module_api_in_the_w2_mode(golden_dir):
    """The default arithmetic of bench.py ("w2": the point MLP of a layer as one launch) through the MODULE API — `Diffusion.forward`, the
    hipGraph forward inside and outside `frozen_weights`, a short captured sampler run and a cached evaluation — against the reference's golden vector of the d = 384 network (5e-4, the
    mode's bar) and against the plan-level call (same bits)."""
    from gecco_amd import hip_ops
    name = "uncond_d384_L6_N128"
    g = _load(golden_dir, name)
    d, L, N, seed = cases.UNCOND_CASES[name]
    p, x, sigma = cases.uncond_inputs(name)
    m = build_uncond(d, L)
    m.load_state_dict(uncond_state_dict(p), strict=True)
    m = m.cuda().eval()
    old = hip_ops.default_precision()
    hip_ops.set_default_precision("w2")
    try:
        with torch.no_grad():
            den = m(x.cuda(), sigma.cuda(), None)
            _close(den, g["denoised"], 5e-4)
            plan = hip_ops.LinearLiftPlan({k: v.cuda() for k, v in p.items()}, cases.H, cases.I, precision="w2")
            assert torch.equal(den, plan.forward(x.cuda(), sigma.cuda()))
            strict = hip_ops.LinearLiftPlan({k: v.cuda() for k, v in p.items()}, cases.H, cases.I, precision="mixed").forward(x.cuda(), sigma.cuda())
            assert not torch.equal(den, strict), "the one-launch point MLP did not run"
            for frozen in (False, True):
                run = m.graphed_forward(x.cuda(), sigma.cuda(), None, frozen_weights=frozen)
                assert torch.equal(run(), den) and torch.equal(run(), den)
            den2, cache = m(x.cuda(), sigma.cuda(), None, do_cache=True)
            assert torch.equal(den2, den)
            assert torch.isfinite(m(x[:, :64].contiguous().cuda(), sigma.cuda(), None, cache=cache)).all()
            out = m.sample_stochastic((2, N, 3), None, num_steps=4)
            assert out.shape == (2, N, 3) and torch.isfinite(out).all()
    finally:
        hip_ops.set_default_precision(old)


def test_forward_and_sampler_under_autocast_and_half_inputs(golden_dir):
    """Callers of the reference wrap sampling in `torch.autocast` (its notebooks) and train under fp16 autocast:
    the HIP path computes in its own arithmetic whatever the autocast state or the input dtype — same bits as the plain call."""
    name = "uncond_d128_L4_N256"
    d, L, N, seed = cases.UNCOND_CASES[name]
    p, x, sigma = cases.uncond_inputs(name)
    m = build_uncond(d, L)
    m.load_state_dict(uncond_state_dict(p), strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        ref = m(x.cuda(), sigma.cuda(), None)
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            got = m(x.cuda(), sigma.cuda(), None)
        assert got.dtype == torch.float32 and torch.equal(got, ref)
        got16 = m(x.cuda().half(), sigma.cuda(), None)      # a half input is widened, not computed on in fp16
        _close(got16, m(x.half().float().cuda(), sigma.cuda(), None).cpu(), 1e-6)
        s0 = m.sample_stochastic((2, 64, 3), None, num_steps=4, rng=torch.Generator("cuda").manual_seed(7))
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            s1 = m.sample_stochastic((2, 64, 3), None, num_steps=4, rng=torch.Generator("cuda").manual_seed(7))
        assert torch.equal(s0, s1)


def test_plan_follows_parameter_moves_and_updates():
    d, L = 64, 2
    p, x, sigma = W.linear_lift_state_dict(5, d, L, cases.I, cases.H), *W.synthetic_cloud(5, 2, 96)
    m = build_uncond(d, L)
    m.load_state_dict(uncond_state_dict(p))
    m = m.cuda()
    ref = cpu_ref.uncond_denoiser(p, "", cases.H)
    with torch.no_grad():
        _close(m(x.cuda(), sigma.cuda(), None), ref(x, sigma))
        # in-place parameter update (what an optimizer / load_state_dict does) is seen without a rebuild
        p2 = W.linear_lift_state_dict(6, d, L, cases.I, cases.H)
        m.load_state_dict(uncond_state_dict(p2))
        _close(m(x.cuda(), sigma.cuda(), None), cpu_ref.uncond_denoiser(p2, "", cases.H)(x, sigma))
        # re-allocated parameters (round trip through the CPU) rebuild the pointer tables
        m = m.cpu().cuda()
        _close(m(x.cuda(), sigma.cuda(), None), cpu_ref.uncond_denoiser(p2, "", cases.H)(x, sigma))


@pytest.mark.parametrize("use_graph", [True, False])
def test_sample_stochastic_golden(golden_dir, use_graph):
    g = _load(golden_dir, "sampler")
    c = cases.SAMPLER_CASE
    p, latents, noises = cases.sampler_inputs()
    m = build_uncond(c["d"], c["L"], sigma_max=c["sigma_max"])
    m.load_state_dict(uncond_state_dict(p))
    m = m.cuda().eval()
    out = m.sample_stochastic((c["B"], c["N"], 3), None, noise=[latents] + noises, num_steps=c["num_steps"],
                              use_graph=use_graph)
    assert out.dtype == torch.float64 and out.shape == (c["B"], c["N"], 3)
    _close(out, g["sample"], 2e-4)
    assert torch.equal(m.t_steps(64, 165.0, 0.002, 7).cpu(), g["t_steps_64"])


def test_sample_stochastic_graph_equals_eager_and_is_deterministic():
    m = build_uncond(64, 2)
    m.load_state_dict(uncond_state_dict(W.linear_lift_state_dict(8, 64, 2, cases.I, cases.H)))
    m = m.cuda().eval()
    a = m.sample_stochastic((3, 128, 3), None, num_steps=8)
    b = m.sample_stochastic((3, 128, 3), None, num_steps=8)                     # default generator seed 42
    c = m.sample_stochastic((3, 128, 3), None, num_steps=8, use_graph=False)
    assert torch.equal(a, b) and torch.equal(a, c)
    d = m.sample_stochastic((3, 128, 3), None, num_steps=8, rng=torch.Generator("cuda").manual_seed(7))
    assert not torch.equal(a, d) and torch.isfinite(d).all()


def test_upsample_golden(golden_dir):
    g = _load(golden_dir, "upsample")
    c = cases.UPSAMPLE_CASE
    p, data = cases.upsample_inputs()
    m = build_uncond(c["d"], c["L"], sigma_max=c["sigma_max"])
    m.load_state_dict(uncond_state_dict(p))
    m = m.cuda().eval()
    out = m.upsample(data.cuda(), n_new=c["n_new"], num_steps=c["num_steps"], num_substeps=c["num_substeps"],
                     noise=cases.upsample_draw_list(), use_graph=True)
    assert out.dtype == torch.float64
    _close(out, g["upsampled"], 2e-4)
    # the captured outer-step graph and the eager loop (default) are the same launches: bit-identical clouds
    eager = m.upsample(data.cuda(), n_new=c["n_new"], num_steps=c["num_steps"], num_substeps=c["num_substeps"],
                       noise=cases.upsample_draw_list(), use_graph=False)
    assert torch.equal(out, eager)
    out2 = m.upsample(data.cuda(), n_new=500, num_steps=4, num_substeps=2, use_graph=True)  # generator path, many new points
    assert out2.shape == (c["B"], 500, 3) and torch.isfinite(out2).all()
    out3 = m.upsample(data.cuda(), n_new=500, num_steps=4, num_substeps=2, use_graph=False)   # same seed, same draw order
    assert torch.equal(out2, out3)


# ------------------------------------------------------------------------------------------- conditional
def test_reparam_golden(golden_dir):
    from gecco_amd.reparam import GaussianReparam, UVLReparam
    from gecco_amd.structs import Context3d
    g = _load(golden_dir, "reparam")
    feats, K, geom, um, us = cases.lookup_inputs("lookup_small")
    ctx = Context3d(image=torch.zeros(1), K=K.cuda())
    rp = UVLReparam(um, us).cuda()
    xyz = rp.diffusion_to_data(geom.cuda(), ctx)
    _close(xyz, g["uvl_xyz"], 2e-5)
    _close(rp.data_to_diffusion(g["uvl_xyz"].cuda(), ctx), g["uvl_back"], 2e-4)  # atanh near +-1 amplifies ulps
    xyz64 = rp.diffusion_to_data(geom.double().cuda(), ctx)                      # sampler state is fp64
    assert xyz64.dtype == torch.float64
    _close(xyz64, cpu_ref.uvl_diffusion_to_data(geom.double(), K.double(), um.double(), us.double()), 1e-12)
    gr = GaussianReparam(torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA)).cuda()
    _close(gr.diffusion_to_data(geom.cuda(), None), g["gauss"], 1e-6)
    _close(gr.data_to_diffusion(g["gauss"].cuda(), None), geom, 1e-5)


@pytest.mark.parametrize("hw", [(14, 14), (56, 56), (7, 9)])
def test_bilinear_tap_indices_bit_exact(hw):
    """North-star bar: bit-exact projection index math.  Same uv bits in -> identical integer taps and identical
    fractional weights as the oracle's torch-order fp32 arithmetic, including out-of-range coordinates."""
    from gecco_amd import hip_ops
    rs = np.random.RandomState(hw[0])
    uv = np.concatenate([rs.uniform(-0.2, 1.2, size=(20000, 2)), rs.uniform(0, 1, size=(20000, 2)),
                         np.array([[0.0, 0.0], [1.0, 1.0], [0.5, 0.5], [1e-8, 1 - 1e-8]])]).astype(np.float32)
    # texel-boundary coordinates, where a 1-ulp difference would flip floor()
    k = np.arange(0, hw[1] + 1, dtype=np.float32)
    edge = np.stack([(k + 0.5) / hw[1], (k[::-1] + 0.5) / hw[1]], -1).astype(np.float32)
    uv = torch.from_numpy(np.concatenate([uv, edge, np.nextafter(edge, np.float32(2)), np.nextafter(edge, np.float32(-2))]))
    x0, y0, wx, wy = cpu_ref.bilinear_taps(uv, hw[0], hw[1])
    gx0, gy0, gwx, gwy = hip_ops.bilinear_taps(uv.cuda(), hw[0], hw[1])
    assert torch.equal(gx0.cpu(), x0) and torch.equal(gy0.cpu(), y0)
    assert torch.equal(gwx.cpu(), wx) and torch.equal(gwy.cpu(), wy)


def _lookup_chain_inputs(seed, B=3, N=60000):
    """Geometry that exercises the projection's corners: in-frustum points, points behind the camera, |z| at / below / just above the
    1e-8 guard of kornia's divide, far outside the image; per-sample intrinsics."""
    rs = np.random.RandomState(seed)
    g = rs.randn(B, N, 3).astype(np.float32)
    g[..., 2] = rs.uniform(0.5, 5.0, size=(B, N)).astype(np.float32)
    g[:, :2000, 2] *= -1.0                                              # behind the camera
    g[:, 2000:2200, 2] = np.float32(1e-8)                               # exactly the guard (|z| > eps is false)
    g[:, 2200:2400, 2] = np.nextafter(np.float32(1e-8), np.float32(1))  # one ulp above it
    g[:, 2400:2600, 2] = 0.0
    g[:, 2600:2800, 2] = np.float32(-1e-8)                              # z + eps == 0 would divide by zero: the guard picks 1
    g[:, 2800:4000, :2] *= 30.0                                         # far outside the image
    f = rs.uniform(0.8, 1.5, size=(B,)).astype(np.float32)
    K = np.zeros((B, 3, 3), dtype=np.float32)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = f, f * 1.1, 0.5, 0.45, 1.0
    return torch.from_numpy(g), torch.from_numpy(K)


@pytest.mark.parametrize("kind", ["none", "gaussian"])
def test_fused_lookup_index_chain_bit_exact(kind):
    """North-star bar, through the fused lookup: geometry -> reparametrisation -> project_points -> taps as ray_lookup_kernel
    computes them (gecco_ray_lookup_taps_f32 runs the kernel's own device functions, csrc/lookup.hip project_uv / bilinear_taps)
    against the oracle's chain from the SAME geometry bits (models/ray.py:64-87, reparam.py:57-63, kornia project_points): the
    projected uv, every level's integer taps and the fractional weights must be identical to the bit — with and without the
    EDM input scale c_in, including z ~ 0, points behind the camera and far outside the image."""
    from gecco_amd import hip_ops
    geom, K = _lookup_chain_inputs(11 if kind == "none" else 12)
    B, N, _ = geom.shape
    hw = [(56, 56), (28, 28), (14, 14), (7, 9)]
    levels = [torch.zeros(B, h, w, 4, device="cuda") for h, w in hw]
    if kind == "gaussian":
        mean, sigma = torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA)
        mean_d, sigma_d = mean.cuda(), sigma.cuda()          # (the table holds raw pointers: the tensors must outlive it)
        rp = hip_ops.make_reparam(1, mean_d, sigma_d)
    else:
        mean = sigma = None
        rp = hip_ops.make_reparam(0)
    sig = torch.tensor([0.002, 1.0, 80.0])
    coef = torch.stack([torch.zeros(3), torch.zeros(3), 1.0 / torch.sqrt(sig * sig + 1.0), torch.zeros(3)], dim=1).contiguous()
    for use_cin in (False, True):
        gin = geom * coef[:, 2].reshape(B, 1, 1) if use_cin else geom      # EDMPrecond's c_in * x (diffusion.py:52-56), one fp32 product
        xyz = cpu_ref.gaussian_diffusion_to_data(gin, mean, sigma) if kind == "gaussian" else gin
        uv = cpu_ref.project_points(xyz, K)
        guv, gx0, gy0, gwx, gwy = hip_ops.ray_lookup_taps(geom.cuda(), K.cuda(), levels, rp, coef=coef.cuda() if use_cin else None)
        assert torch.equal(guv.cpu(), uv), (kind, use_cin, (guv.cpu() != uv).sum().item())
        # a coordinate beyond int32 (z ~ 0: uv ~ 1e8) converts to INT_MIN on the host and saturates on the device — either way every tap of
        # such a point lies outside the image (zero padding): identical bits are required wherever the conversion is defined
        ok = (uv.abs() < 1e6).all(-1)
        assert ok.float().mean().item() > 0.9
        for l, (h, w) in enumerate(hw):
            x0, y0, wx, wy = cpu_ref.bilinear_taps(uv, h, w)
            assert torch.equal(gx0[l].cpu()[ok], x0[ok]) and torch.equal(gy0[l].cpu()[ok], y0[ok]), (kind, use_cin, l)
            assert torch.equal(gwx[l].cpu()[ok], wx[ok]) and torch.equal(gwy[l].cpu()[ok], wy[ok]), (kind, use_cin, l)
            far = ~ok
            gx, gy = gx0[l].cpu()[far], gy0[l].cpu()[far]
            assert (((gx < -1) | (gx > w)) | ((gy < -1) | (gy > h))).all(), (kind, use_cin, l)


def test_fused_lookup_index_chain_uvl():
    """The UVL chain goes through tanh / exp, where the device's functions and libm differ by an ulp or two: the projected uv is
    compared in ulps (histogram printed) and the INTEGER taps must not flip on the lookup fixtures' geometry nor on a dense sample
    — wherever a flip would need uv within an ulp of a texel edge, it is reported with its distance to the edge."""
    from gecco_amd import hip_ops
    for name in list(cases.LOOKUP_CASES):
        feats, K, geom, um, us = cases.lookup_inputs(name)
        um_d, us_d = um.cuda(), us.cuda()
        rp = hip_ops.make_reparam(2, um_d, us_d, 1.1)
        levels = [f.permute(0, 2, 3, 1).contiguous().cuda() for f in feats]
        guv, gx0, gy0, gwx, gwy = hip_ops.ray_lookup_taps(geom.cuda(), K.cuda(), levels, rp)
        uv = cpu_ref.project_points(cpu_ref.uvl_diffusion_to_data(geom, K, um, us), K)
        ulps = (guv.cpu().view(torch.int32) - uv.view(torch.int32)).abs()
        hist = torch.bincount(ulps.flatten().clamp(max=8), minlength=9).tolist()
        flips = 0
        for l, f in enumerate(feats):
            x0, y0, wx, wy = cpu_ref.bilinear_taps(uv, f.shape[2], f.shape[3])
            bad = (gx0[l].cpu() != x0) | (gy0[l].cpu() != y0)
            flips += int(bad.sum())
            assert (gwx[l].cpu() - wx)[~bad].abs().max().item() <= 1e-4 and (gwy[l].cpu() - wy)[~bad].abs().max().item() <= 1e-4
        print(f"{name}: uv ulp histogram (0..7, >=8) {hist}, integer tap flips {flips} of {uv.numel() // 2 * len(feats)}")
        # (ulps of uv are large only where uv itself is small: the chain subtracts cx from a value near it; the absolute bar below is the
        # meaningful one: 1e-6 of a coordinate that spans [0, 1] is 1e-4 of a texel on the finest level)
        assert (guv.cpu() - uv).abs().max().item() <= 1e-6 and flips == 0, (name, hist, flips, (guv.cpu() - uv).abs().max().item())


@pytest.mark.parametrize("name", list(cases.LOOKUP_CASES))
def test_lookup_golden(golden_dir, name):
    from gecco_amd.models.activation import GaussianActivation
    from gecco_amd.models.ray import RayNetwork
    from gecco_amd.models.set_transformer import SetTransformer
    from gecco_amd.reparam import UVLReparam
    from gecco_amd.structs import Context3d
    g = _load(golden_dir, name)
    feats, K, geom, um, us = cases.lookup_inputs(name)
    net = RayNetwork(backbone=SetTransformer(n_layers=1, num_inducers=cases.I, feature_dim=64, t_embed_dim=1,
                                             num_heads=cases.H, activation=GaussianActivation),
                     reparam=UVLReparam(um, us), context_dims=[f.shape[1] for f in feats]).cuda()
    ctx = Context3d(image=torch.zeros(1), K=K.cuda())
    with torch.no_grad():
        got = net.extract_image_features(geom.cuda(), [f.cuda() for f in feats], ctx)
    _close(got, g["lookup"], 1e-4)
    # channels-last inputs are consumed without a copy and give identical bits
    with torch.no_grad():
        got2 = net.extract_image_features(geom.cuda(), [f.cuda().contiguous(memory_format=torch.channels_last) for f in feats], ctx)
    assert torch.equal(got, got2)


@pytest.mark.parametrize("name", list(cases.LOOKUP_CASES))
def test_lookup_on_fp16_texels(name):
    """The "w2" mode's texel image (GeccoPyramid.texel_f16, gecco_cast_f16): the lookup on fp16 texels equals — to the bit, outputs and
    GroupNorm partials — the lookup on fp32 texels holding the rounded values (only the gathered bytes differ: coordinates, taps, weights
    and the interpolation are the same fp32 expressions); against the golden lookup it is the fp16 rounding of the features away
    (models/ray.py:64-87); the gradient entry points refuse it."""
    from gecco_amd import _lib, hip_ops
    feats, K, geom, um, us = cases.lookup_inputs(name)
    umc, usc = um.cuda(), us.cuda()                       # (the struct holds raw pointers: keep the tensors)
    rp = hip_ops.make_reparam(2, umc, usc, 1.1)
    lv32 = hip_ops.to_channels_last_levels([f.cuda() for f in feats])
    lv16 = hip_ops.half_levels(lv32)
    assert all(h.dtype == torch.float16 and torch.equal(h, f.half()) for h, f in zip(lv16, lv32))   # round to nearest even
    got16, st16 = hip_ops.ray_lookup(geom.cuda(), K.cuda(), lv16, rp, want_stats=True)
    got32r, st32r = hip_ops.ray_lookup(geom.cuda(), K.cuda(), [h.float() for h in lv16], rp, want_stats=True)
    assert torch.equal(got16, got32r) and torch.equal(st16, st32r)
    got32 = hip_ops.ray_lookup(geom.cuda(), K.cuda(), lv32, rp)
    e = cpu_ref.rel_err(got16.cpu(), got32.cpu())
    print(f"{name}: fp16 texels vs fp32 texels: max-norm {e[0]:.2e}, rel-L2 {e[1]:.2e}")
    assert 0 < e[0] <= 1e-3 and e[1] <= 4e-4, e
    uv16 = hip_ops.ray_lookup_taps(geom.cuda(), K.cuda(), lv16, rp)
    uv32 = hip_ops.ray_lookup_taps(geom.cuda(), K.cuda(), lv32, rp)
    assert all(torch.equal(a, b) for a, b in zip(uv16, uv32))                                        # the index chain never sees the texels
    pyr = hip_ops.make_pyramid(lv16)
    import ctypes as C
    dout = torch.zeros_like(got16)
    dg = torch.zeros(geom.shape, device="cuda")
    rc = _lib.load().gecco_ray_lookup_dgeom_f32(hip_ops._ptr(geom.cuda().contiguous()), hip_ops._ptr(K.cuda().float().contiguous()), C.byref(rp), C.byref(pyr),
                                                hip_ops._ptr(dout), hip_ops._ptr(dg), None, geom.shape[0], geom.shape[1], None)
    assert rc != 0


def test_graphed_forward_replays_the_eager_bits():
    """Diffusion.graphed_forward: one captured evaluation replayed on new inputs equals the eager call bit for bit."""
    name = "uncond_d128_L4_N256"
    d, L, N, seed = cases.UNCOND_CASES[name]
    p, x, sigma = cases.uncond_inputs(name)
    m = build_uncond(d, L)
    m.load_state_dict(uncond_state_dict(p))
    m = m.cuda().eval()
    run = m.graphed_forward(x.cuda(), sigma.cuda(), None)
    with torch.no_grad():
        ref = m(x.cuda(), sigma.cuda(), None)
        assert torch.equal(run(), ref)
        x2 = (x * 0.5 + 0.1).cuda()
        s2 = (sigma * 1.7).cuda()
        assert torch.equal(run(x2, s2), m(x2, s2, None))


@pytest.mark.parametrize("precision,tol", [("mixed", 2e-4), ("w2", 5e-4)])
@pytest.mark.parametrize("B,N,d", [(2, 333, 128), (2, 130, 384), (1, 1000, 256), (3, 128, 384)])
def test_conditional_network_ragged_point_counts(B, N, d, precision, tol):
    """RayNetwork (models/ray.py:89-123) at point counts that are not whole 128-row tiles, against the oracle: in the "w2" mode the point MLP
    runs as one launch only on whole tiles, the lookup on fp16 texels always; with option "imgproj16" img_feature_proj runs on fp16 operands
    with per-sample folded weight images from N >= 128 on (its last row tile is ragged) — both settings."""
    from gecco_amd import hip_ops
    L, hw, cdims = 1, 64, (96, 192, 384)
    p = W.ray_network_state_dict(91 + N, d, L, cases.I, cases.H, context_dims=cdims)
    feats, K = W.synthetic_context(92 + N, B, hw=hw, context_dims=cdims)
    g = torch.Generator().manual_seed(93 + N)
    x = torch.randn(B, N, 3, generator=g)
    sigma = torch.tensor([0.05, 3.0, 80.0][:B])
    with torch.no_grad():
        ref, raw_ref = cpu_ref.cond_denoiser(p, "", cases.H, K, feats)(x, sigma, return_raw=True)
    levels = hip_ops.to_channels_last_levels([f.cuda() for f in feats])
    for ip in ((0, 1) if precision == "w2" else (0,)):
        net = hip_ops.RayNetworkPlan({k: v.cuda() for k, v in p.items()}, cases.H, cases.I, precision=precision, options={"imgproj16": ip})
        den, raw = net.forward(x.cuda(), sigma.cuda(), K.cuda(), levels, return_raw=True)
        e = (_close(den, ref, tol), _close(raw, raw_ref, tol))
        print(f"conditional B={B} N={N} d={d} {precision} imgproj16={ip}: D {e[0]}, F_x {e[1]}")


@pytest.mark.parametrize("d", [128, 384])
def test_conditional_module_api_in_the_w2_mode(d):
    """The image-conditional model through the MODULE API in the headline arithmetic at a shape where every fused piece of the mode runs
    (N = 256: whole 128-row tiles; 672 pyramid channels): the one-launch point MLP (d = 128 and 384), the fp16 texel image and (opt-in, option "imgproj16") img_feature_proj
    on fp16 operands with GroupNorm folded into per-sample weight images — `Diffusion.forward` against the oracle (cpu_ref.cond_denoiser:
    models/ray.py:89-123 under diffusion.py:37-57) at the mode's 5e-4, the captured forward inside and outside `frozen_weights`, the
    model-level option switch, and a short captured sampler run."""
    from gecco_amd.diffusion import Conditioner
    from gecco_amd.models.feature_pyramid import FeaturePyramidContext
    from gecco_amd.structs import Context3d
    L, N, B, hw, cdims = 2, 256, 2, 64, (96, 192, 384)
    p = W.ray_network_state_dict(61, d, L, cases.I, cases.H, context_dims=cdims)
    feats, K = W.synthetic_context(66, B, hw=hw, context_dims=cdims)
    g = torch.Generator().manual_seed(62)
    x = torch.randn(B, N, 3, generator=g)
    sigma = torch.tensor([0.1, 20.0])
    with torch.no_grad():
        ref = cpu_ref.cond_denoiser(p, "", cases.H, K, feats)(x, sigma)

    class FixedPyramid(Conditioner):
        def forward(self, raw_ctx):
            return FeaturePyramidContext(features=[f.cuda() for f in feats], K=raw_ctx.K)

    m = build_cond(d, L, cdims, conditioner=FixedPyramid())
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.uvl_mean"], sd["reparam.uvl_std"] = p["reparam.uvl_mean"], p["reparam.uvl_std"]
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval().set_precision("w2")
    ctx = Context3d(image=torch.zeros(B, 3, hw, hw).cuda(), K=K.cuda())
    with torch.no_grad():
        den = m(x.cuda(), sigma.cuda(), ctx)
        e = _close(den, ref, 5e-4)
        for frozen in (False, True):
            run = m.graphed_forward(x.cuda(), sigma.cuda(), ctx, frozen_weights=frozen)
            assert torch.equal(run(), den) and torch.equal(run(), den)
        m.set_option("imgproj16", 1)                     # (opt-in) img_feature_proj on fp16 operands, GroupNorm folded into per-sample weight images
        den0 = m(x.cuda(), sigma.cuda(), ctx)
        e0 = _close(den0, ref, 5e-4)
        assert not torch.equal(den0, den)
        for frozen in (False, True):
            run = m.graphed_forward(x.cuda(), sigma.cuda(), ctx, frozen_weights=frozen)
            assert torch.equal(run(), den0) and torch.equal(run(), den0)
        m.set_option("imgproj16", -1)
        assert torch.equal(m(x.cuda(), sigma.cuda(), ctx), den)
        m.set_option("mlpw", 0)                          # the mixed mode's two launches instead of the one-launch point MLP
        assert not torch.equal(m(x.cuda(), sigma.cuda(), ctx), den)
        m.set_option("mlpw", 1)
        print(f"conditional module, w2, d={d}: D vs oracle {e} (imgproj16 on: {e0})")
        out = m.sample_stochastic((B, N, 3), ctx, num_steps=4)
        assert out.shape == (B, N, 3) and torch.isfinite(out).all()


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16x3", 2e-4), ("mixed", 2e-4), ("fp16", 1e-3)])
@pytest.mark.parametrize("name", list(cases.COND_CASES))
def test_conditional_diffusion_golden(golden_dir, name, precision, tol):
    from gecco_amd import hip_ops
    from gecco_amd.diffusion import Conditioner
    from gecco_amd.models.feature_pyramid import FeaturePyramidContext
    from gecco_amd.structs import Context3d
    g = _load(golden_dir, name)
    d, L, N, hw, cdims, seed = cases.COND_CASES[name]
    p, x, sigma, K, feats = cases.cond_inputs(name)

    class FixedPyramid(Conditioner):
        def forward(self, raw_ctx):
            return FeaturePyramidContext(features=[f.cuda() for f in feats], K=raw_ctx.K)

    m = build_cond(d, L, cdims, conditioner=FixedPyramid())
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.uvl_mean"], sd["reparam.uvl_std"] = p["reparam.uvl_mean"], p["reparam.uvl_std"]
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    ctx = Context3d(image=torch.zeros(len(sigma), 3, hw, hw).cuda(), K=K.cuda())
    old = hip_ops.default_precision()
    hip_ops.set_default_precision(precision)   # the arithmetic mode of the module mirror (fp32 unless told otherwise)
    try:
        with torch.no_grad():
            den = m(x.cuda(), sigma.cuda(), ctx)
            # unfused module path (RayNetwork.forward called directly, as a user of the class would)
            c_skip, c_out, c_in, c_noise = cpu_ref.edm_coeffs(sigma)
            F_x, _ = m.backbone.model((c_in * x).cuda(), c_noise.cuda(), ctx, m.conditioner(ctx))
        e1 = _close(den, g["denoised"], tol)
        e2 = _close(F_x, g["F_x"], tol)
        print(name, precision, "denoised", e1, "F_x", e2)
        # a few sampler steps run end to end on the conditional model (graph-captured)
        out = m.sample_stochastic((len(sigma), N, 3), ctx, num_steps=4)
        assert out.shape == (len(sigma), N, 3) and torch.isfinite(out).all()
    finally:
        hip_ops.set_default_precision(old)


# ------------------------------------------------------------------------------------------- nn.ReLU (reference default)
@pytest.mark.parametrize("precision,tol", [("fp32", TOL), ("bf16x3", 2e-4), ("mixed", 2e-4), ("fp16", 1e-3)])
def test_default_relu_activation_golden(golden_dir, precision, tol):
    """The reference's DEFAULT `activation=nn.ReLU` (models/mlp.py:12, set_transformer.py:81,133): module API with the
    argument left out, state dict without alpha entries, fused path (epilogue code 3) in every arithmetic mode, against
    the reference's own output; and the training path's gradient through ReLU against torch autograd on the oracle."""
    from gecco_amd import hip_ops
    from gecco_amd.diffusion import Diffusion, EDMLoss, EDMPrecond, IdleConditioner, LogUniformSchedule
    from gecco_amd.models.linear_lift import LinearLift
    from gecco_amd.models.set_transformer import SetTransformer
    from gecco_amd.reparam import GaussianReparam
    g = _load(golden_dir, "relu_d128_L2_N256")
    c = cases.RELU_CASE
    p, x, sigma = cases.relu_inputs()
    net = LinearLift(inner=SetTransformer(n_layers=c["L"], num_inducers=cases.I, feature_dim=c["d"], t_embed_dim=1,
                                          num_heads=cases.H), feature_dim=c["d"])
    assert isinstance(net.inner.layers[0].mlp[1], torch.nn.ReLU)
    m = Diffusion(backbone=EDMPrecond(model=net), conditioner=IdleConditioner(),
                  reparam=GaussianReparam(torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA)),
                  loss=EDMLoss(schedule=LogUniformSchedule(max=165.0)))
    m.load_state_dict(uncond_state_dict(p), strict=True)
    m = m.cuda().eval()
    old = hip_ops.default_precision()
    hip_ops.set_default_precision(precision)
    try:
        with torch.no_grad():
            den = m(x.cuda(), sigma.cuda(), None)
        e = _close(den, g["denoised"], tol)
        print("relu", precision, e)
        if precision == "fp32":   # gradient of a scalar of the output w.r.t. two weights that sit before / after a ReLU
            pg = {k: v.clone().requires_grad_(True) for k, v in p.items()}
            ref = cpu_ref.uncond_denoiser(pg, "", cases.H)(x, sigma)
            (ref ** 2).mean().backward()
            m.train()
            out = m(x.cuda(), sigma.cuda(), None)
            (out ** 2).mean().backward()
            for k in ("inner.layers.0.mlp.0.weight", "inner.layers.1.broadcast.mlp.2.weight", "lift.weight"):
                got = dict(m.backbone.model.named_parameters())[k].grad
                _close(got, pg[k].grad, 5e-4)
    finally:
        hip_ops.set_default_precision(old)


@pytest.mark.parametrize("I,d", [(32, 128), (96, 128), (64, 96), (64, 640)])
def test_num_inducers_other_than_64_runs_the_general_path(I, d):
    """The reference's `SetTransformer(num_inducers=...)` takes any count; the fused kernels are built around 64.  Other
    counts run the general composition (every op still in libgecco_hip.so) instead of raising: forward and cached
    evaluation against the oracle, the sampler, and the training gradients against the oracle's autograd.  Likewise head
    dimensions outside the fused attention kernels' set (d = 96: 12, d = 640: 80 with the 8 heads of every config)."""
    L, N, B = 2, 256, 3
    p = W.linear_lift_state_dict(23, d, L, I, cases.H)
    m = build_uncond(d, L, num_inducers=I)
    m.load_state_dict(uncond_state_dict(p), strict=True)
    m = m.cuda().eval()
    rs = np.random.RandomState(I)
    x = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
    sigma = torch.tensor([0.05, 1.0, 40.0])
    D = cpu_ref.uncond_denoiser(p, "", cases.H)
    with torch.no_grad():
        ref, ref_cache = D(x, sigma, do_cache=True)
        out, cache = m(x.cuda(), sigma.cuda(), None, do_cache=True)
        _close(out, ref, 2e-5)
        assert len(cache) == L and cache[0].shape == (B, I, d)
        x2 = torch.from_numpy(rs.randn(B, 2 * N, 3).astype(np.float32))
        _close(m(x2.cuda(), sigma.cuda(), None, cache=cache), D(x2, sigma, cache=ref_cache), 2e-5)
        smp = m.sample_stochastic((B, N, 3), None, num_steps=4)
        assert smp.shape == (B, N, 3) and torch.isfinite(smp).all()
    # training: loss and every gradient
    m.train()
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    data = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
    noise = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
    s3 = sigma.reshape(-1, 1, 1)
    ref_loss = (100.0 * (s3 ** 2 + 1) / s3 ** 2 * (cpu_ref.uncond_denoiser(pr, "", cases.H)(data + noise * s3, sigma) - data) ** 2).mean()
    ref_loss.backward()
    s3c = s3.cuda()
    loss = (100.0 * (s3c ** 2 + 1) / s3c ** 2 * (m(data.cuda() + noise.cuda() * s3c, sigma.cuda(), None) - data.cuda()) ** 2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) / abs(float(ref_loss.detach())) < 1e-5
    for k, q in m.named_parameters():
        if k.startswith("backbone.model."):
            _close(q.grad, pr[k[len("backbone.model."):]].grad, 2e-4)


def test_integration_md_ctypes_binding_runs():
    """INTEGRATION.md section 2 shows the ctypes binding a maintainer of gecco_torch would add around `libgecco_hip.so`.  The code block is
    EXECUTED here as it stands in the document (struct layouts incl. the ABI-14 fields, argtypes, the table builder, the forward) on a
    network with the reference's parameter names, and must give the bits of the package's own plan in the same mode."""
    import re
    from gecco_amd import hip_ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    code = next(b for b in blocks if "def edm_precond_forward" in b and "class SetTransformer(C.Structure)" in b)
    code = code.replace('C.CDLL("gecco_amd/libgecco_hip.so")', f'C.CDLL("{os.path.join(root, "gecco_amd", "libgecco_hip.so")}")')
    ns: dict = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    name = "uncond_d384_L6_N128"
    d, L, N, seed = cases.UNCOND_CASES[name]
    p, x, sigma = cases.uncond_inputs(name)
    m = build_uncond(d, L)
    m.load_state_dict(uncond_state_dict(p))
    m = m.cuda().eval()
    net = m.backbone.model                                  # LinearLift: the reference's parameter names
    tbl, keep = ns["table"](net)
    got = ns["edm_precond_forward"](tbl, x.cuda().contiguous(), sigma.cuda().contiguous())
    torch.cuda.synchronize()
    ref = hip_ops.LinearLiftPlan(dict(net.named_parameters()), cases.H, cases.I, precision="w2").forward(x.cuda(), sigma.cuda())
    assert torch.equal(got, ref)
    del keep
