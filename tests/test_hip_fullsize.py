"""Full-size parity of the configurations BASELINE.json quotes (C2 .. C5 shapes), HIP path through the C ABI against
oracle/cpu_ref.py on the same seeded inputs, in every arithmetic mode, on BOTH the preconditioned output D and the raw
network output F_x (at low sigma c_out ~ sigma hides network error in D by ~100x, so F_x is the real bar).

Bar (BASELINE.json north_star): 1e-3 relative (max-norm) against the fp32 reference.  The oracle runs on the host cores
of the GPU box (a few seconds per case); sizes are the headline sizes with a batch the CPU finishes quickly.
"""
import numpy as np
import pytest
import torch

from oracle import cases, cpu_ref
from oracle import weights as W

pytestmark = pytest.mark.gpu

# Per-mode bars (max-norm).  The north-star bar is 1e-3; "fp32" and "bf16x3" (the headline mode of bench.py) are held
# to much tighter numbers on BOTH outputs — what they actually deliver, so regressions show.
# The "fp16" mode (11-bit operands, opt-in fast mode) meets 1e-3 on the forward's output D at every size, but at the
# headline depth (L=6, N=2048) its raw network output F_x sits AT the bar: 0.9e-3 .. 1.2e-3 depending on sigma (measured
# here and reproduced by the CPU emulation tools/experiments/fp16_site_sensitivity.py: ~70 % of that variance is the
# static rounding of the WEIGHTS to fp16, which does not average out over the points of a cloud).  That is why fp16 is
# not the headline mode; its F_x is asserted against 2e-3 and printed.
# "mixed" = fp16 where the sensitivity analysis (tools/experiments/fp16_site_sensitivity.py) shows operand rounding does not
# reach the output (kv_proj | q_proj activations, K | V, q, both attention products — with two-term fp16 weights for
# kv_proj | q_proj), split-bf16 everywhere else: held to the split-bf16 bars.
# "w2" = the mixed mode with the point MLP of every layer as ONE launch (csrc/mlp_fused_w.hip): the 768-wide hidden layer stays in
# registers as fp16 (no second term; AdaGN(x) and both weights keep theirs).  Bar 5e-4 on BOTH outputs, half the north star's 1e-3;
# measured 2.5e-4 .. 4e-4 on F_x (the per-site emulation, profiles/r03_precision_search.txt, predicts 3.6e-4 at C2).  feature_dim 128 .. 512
# in steps of 128 (512, C4: two passes over the hidden width); shapes off the kernel's reach run the mixed mode's launches in this mode.
BARS = {"fp32": 5e-5, "bf16x3": 2e-4, "fp16": 1e-3, "mixed": 2e-4, "w2": 5e-4}
BARS_FX = {"fp32": 5e-5, "bf16x3": 2e-4, "fp16": 2e-3, "mixed": 2e-4, "w2": 5e-4}
MODES = ["fp32", "bf16x3", "mixed", "w2", "fp16"]


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import hip_ops
    return hip_ops


def _cuda(p):
    return {k: v.cuda() for k, v in p.items()}


def _report(tag, got, ref, bar):
    e = cpu_ref.rel_err(got.cpu(), ref)
    print(f"{tag}: max-rel {e[0]:.2e} rel-L2 {e[1]:.2e} (bar {bar:.0e}, margin {bar / max(e[0], 1e-30):.1f}x)")
    assert e[0] <= bar, (tag, e)
    assert e[1] <= bar, (tag, e)     # ... and in the relative L2 norm (the other reading of "relative error"): the same bar
    return e


def _noisy(seed, B, N, sigmas):
    rs = np.random.RandomState(seed)
    sigma = torch.tensor(sigmas, dtype=torch.float32)
    data = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
    x = data + sigma.reshape(-1, 1, 1) * torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
    return x.contiguous(), sigma


# ------------------------------------------------------------------------------------------------- C2
@pytest.fixture(scope="module")
def c2_case():
    """C2 network (N=2048, d=384, L=6, I=64, H=8) on five clouds at sigma = 0.002, 0.1, 1, 20, 165."""
    d, L, N = 384, 6, 2048
    p = W.linear_lift_state_dict(3, d, L, cases.I, cases.H)
    x, sigma = _noisy(5, len(cases.SIGMAS5), N, cases.SIGMAS5)
    with torch.no_grad():
        ref, raw_ref = cpu_ref.uncond_denoiser(p, "", cases.H)(x, sigma, return_raw=True)
    return p, x, sigma, ref, raw_ref


@pytest.mark.parametrize("precision", MODES)
def test_c2_full_size_vs_oracle(ops, c2_case, precision):
    p, x, sigma, ref, raw_ref = c2_case
    pc = _cuda(p)
    den, raw = ops.LinearLiftPlan(pc, cases.H, cases.I, precision=precision).forward(x.cuda(), sigma.cuda(), return_raw=True)
    _report(f"C2 {precision} D   vs oracle", den, ref, BARS[precision])
    _report(f"C2 {precision} F_x vs oracle", raw, raw_ref, BARS_FX[precision])
    # per-sigma: every noise level on its own (a max-norm over the batch is carried by the sigma = 165 cloud for D)
    for b, s in enumerate(cases.SIGMAS5):
        _report(f"C2 {precision} F_x sigma={s}", raw[b], raw_ref[b], BARS_FX[precision])
        _report(f"C2 {precision} D   sigma={s}", den[b], ref[b], BARS[precision])
    if precision != "fp32":   # and against the exact-fp32 HIP mode (what DESIGN.md calls "the mode's own error")
        den32, raw32 = ops.LinearLiftPlan(pc, cases.H, cases.I, precision="fp32").forward(x.cuda(), sigma.cuda(), return_raw=True)
        _report(f"C2 {precision} F_x vs exact-fp32 HIP mode", raw, raw32.cpu(), BARS_FX[precision])
        _report(f"C2 {precision} D   vs exact-fp32 HIP mode", den, den32.cpu(), BARS[precision])


@pytest.mark.parametrize("precision", ["mixed", "w2", "bf16x3"])
def test_c2_headline_batch_vs_oracle(ops, precision):
    """The headline's exact launch shapes (bench.py: B = 64 clouds of N = 2048, d = 384, L = 6 in ONE evaluation) against the
    oracle directly: the samples of a batch are independent (SetTransformer has no cross-sample op), so the oracle runs on eight
    clouds of the batch — every ninth: sigma spans the stratified range 0.002 .. 165 — and the HIP path on all 64 at once."""
    d, L, N, B = 384, 6, 2048, 64
    p = W.linear_lift_state_dict(3, d, L, cases.I, cases.H)
    x, sigma = W.synthetic_cloud(1, B, N)
    pick = [0, 9, 18, 27, 36, 45, 54, 63]
    with torch.no_grad():
        ref, raw_ref = cpu_ref.uncond_denoiser(p, "", cases.H)(x[pick].contiguous(), sigma[pick].contiguous(), return_raw=True)
    den, raw = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision=precision).forward(x.cuda(), sigma.cuda(), return_raw=True)
    for i, b in enumerate(pick):
        _report(f"C2 B=64 {precision} cloud {b} (sigma {float(sigma[b]):.3g}) F_x", raw[b], raw_ref[i], BARS_FX[precision])
        _report(f"C2 B=64 {precision} cloud {b} (sigma {float(sigma[b]):.3g}) D  ", den[b], ref[i], BARS[precision])


# ------------------------------------------------------------------------------------------------- C3 / C4
def _cond_case(seed, d, L, N, hw, B, sigmas):
    cdims = (96, 192, 384)
    p = W.ray_network_state_dict(seed, d, L, cases.I, cases.H, context_dims=cdims)
    feats, K = W.synthetic_context(seed + 5, B, hw=hw, context_dims=cdims)
    x, sigma = _noisy(seed + 1, B, N, sigmas)
    with torch.no_grad():
        ref, raw_ref = cpu_ref.cond_denoiser(p, "", cases.H, K, feats)(x, sigma, return_raw=True)
    return p, x, sigma, K, feats, ref, raw_ref


@pytest.fixture(scope="module")
def c3_case():
    """C3: image-conditional, 224 x 224 ConvNeXt-T-shaped pyramids (96 x 56^2, 192 x 28^2, 384 x 14^2), N=2048, d=384, L=6."""
    return _cond_case(71, 384, 6, 2048, 224, 2, (0.05, 30.0))


@pytest.fixture(scope="module")
def c4_case():
    """C4: image-conditional Taskonomy shape, 256 x 256 -> 64 / 32 / 16 pyramids, N=4096, d=512, L=6."""
    return _cond_case(81, 512, 6, 4096, 256, 2, (0.01, 50.0))


def _run_cond(ops, case, precision, tag):
    p, x, sigma, K, feats, ref, raw_ref = case
    net = ops.RayNetworkPlan(_cuda(p), cases.H, cases.I, precision=precision)
    levels = ops.to_channels_last_levels([f.cuda() for f in feats])
    den, raw = net.forward(x.cuda(), sigma.cuda(), K.cuda(), levels, return_raw=True)
    _report(f"{tag} {precision} D   vs oracle", den, ref, BARS[precision])
    _report(f"{tag} {precision} F_x vs oracle", raw, raw_ref, BARS_FX[precision])
    assert torch.equal(den, net.forward(x.cuda(), sigma.cuda(), K.cuda(), levels))   # deterministic
    with ops.frozen_weights():   # the scope's evaluations share one build of the weight images (a sampler call): same bits
        for _ in range(2):
            assert torch.equal(den, net.forward(x.cuda(), sigma.cuda(), K.cuda(), levels))
        assert net.images.tokens and all(t == net.images.token(False) for t in net.images.tokens.values())   # built once, reused
    # the "w2" mode gathers the pyramid's fp16 texel image (cast once, kept while the fp32 levels are the same tensors); the others fp32 texels
    assert (net._tex16 is not None) == (precision == "w2")
    if precision == "w2":
        img16 = net._tex16[1]
        net.forward(x.cuda(), sigma.cuda(), K.cuda(), levels)
        assert net._tex16[1] is img16
        # img_feature_proj on fp16 operands (option "imgproj16", opt-in): fp16(lookup) times per-sample fp16 images of W with GN16's scale folded
        # in, against the split-bf16 launch the mode runs by default: a different rounding of ONE linear, inside the mode's bar at these shapes
        alt = ops.RayNetworkPlan(_cuda(p), cases.H, cases.I, precision="w2", options={"imgproj16": 1})
        den1, raw1 = alt.forward(x.cuda(), sigma.cuda(), K.cuda(), levels, return_raw=True)
        assert not torch.equal(raw1, raw)
        _report(f"{tag} w2 F_x, imgproj16 = 1, vs oracle", raw1, raw_ref, BARS_FX["w2"])
        e = cpu_ref.rel_err(raw1.cpu(), raw.cpu())
        print(f"{tag} w2: imgproj16 on vs off: max-rel {e[0]:.2e}")
        assert e[0] < 4e-4


@pytest.mark.parametrize("precision", MODES)
def test_c3_full_size_vs_oracle(ops, c3_case, precision):
    _run_cond(ops, c3_case, precision, "C3")


@pytest.mark.parametrize("precision", MODES)
def test_c4_shape_vs_oracle(ops, c4_case, precision):
    _run_cond(ops, c4_case, precision, "C4")


# ------------------------------------------------------------------------------------------------- C5
@pytest.fixture(scope="module")
def c5_case():
    """C5 shape: inducer cache built by one full evaluation at N=2048 (do_cache), then the cached evaluation of
    n_new = 16384 other points against it (diffusion.py:415-447, set_transformer.py:106-117)."""
    d, L, N, n_new, B = 384, 6, 2048, 16384, 2
    p = W.linear_lift_state_dict(9, d, L, cases.I, cases.H)
    x, sigma = _noisy(91, B, N, (0.3, 8.0))
    rs = np.random.RandomState(92)
    x_new = torch.from_numpy(rs.randn(B, n_new, 3).astype(np.float32)) * (1 + sigma.reshape(-1, 1, 1))
    D = cpu_ref.uncond_denoiser(p, "", cases.H)
    with torch.no_grad():
        (ref, raw_ref), cache = D(x, sigma, do_cache=True, return_raw=True)
        new_ref, new_raw_ref = D(x_new, sigma, cache=cache, return_raw=True)
    return p, x, sigma, x_new, ref, raw_ref, cache, new_ref, new_raw_ref


@pytest.mark.parametrize("precision", MODES)
def test_c5_cached_upsampling_shape_vs_oracle(ops, c5_case, precision):
    p, x, sigma, x_new, ref, raw_ref, cache_ref, new_ref, new_raw_ref = c5_case
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision=precision)
    (den, raw), cache = net.forward(x.cuda(), sigma.cuda(), do_cache=True, return_raw=True)
    _report(f"C5 {precision} full evaluation F_x", raw, raw_ref, BARS_FX[precision])
    # the inducer states are internal tensors: reported, with 5x slack in the reduced-precision modes (as
    # tests/test_hip_network.py::test_cached_mode_golden)
    _report(f"C5 {precision} inducer cache", torch.stack(cache), torch.stack(cache_ref),
            BARS[precision] * (1 if precision == "fp32" else 5))
    # (1) cached evaluation on the HIP path's OWN cache: the end-to-end upsampling data flow
    new, new_raw = net.forward(x_new.cuda(), sigma.cuda(), cache=cache, return_raw=True)
    _report(f"C5 {precision} cached n_new=16384 D   (own cache)", new, new_ref, BARS[precision])
    _report(f"C5 {precision} cached n_new=16384 F_x (own cache)", new_raw, new_raw_ref, BARS_FX[precision])
    # (2) cached evaluation on the oracle's cache: the cached layer path in isolation
    new2, new_raw2 = net.forward(x_new.cuda(), sigma.cuda(), cache=[c.cuda() for c in cache_ref], return_raw=True)
    _report(f"C5 {precision} cached n_new=16384 F_x (oracle cache)", new_raw2, new_raw_ref, BARS_FX[precision])


# ------------------------------------------------------------------------------------------------- deep networks
@pytest.mark.parametrize("d,L", [(384, 8), (256, 10), (128, 14)])
def test_fp16_deep_network_weight_staging(ops, d, L):
    """Networks deeper than the shipped L=6: the per-forward weight-image staging table (96 split jobs per launch,
    csrc/api.hip st_forward) is flushed and refilled mid-network; output must still match the oracle."""
    N, B = 256, 2
    p = W.linear_lift_state_dict(100 + L, d, L, cases.I, cases.H)
    x, sigma = _noisy(7, B, N, (0.2, 4.0))
    with torch.no_grad():
        ref, raw_ref = cpu_ref.uncond_denoiser(p, "", cases.H)(x, sigma, return_raw=True)
    for precision in ("fp16", "bf16x3", "mixed", "w2"):   # ("w2": the one-launch point MLP at d = 384, 256 and 128)
        den, raw = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision=precision).forward(x.cuda(), sigma.cuda(), return_raw=True)
        # error grows with depth (each layer adds its own rounding); the bar stays the north star's
        # ("w2": the hidden layer's one-term rounding grows with depth — the mode's own 5e-4 bar belongs to the shipped L = 6 .. 8; deeper
        # networks are held to the north star's 1e-3, as in test_w2_deep_networks_at_the_fused_width)
        w2bar = 5e-4 if L <= 8 else 1e-3
        _report(f"d={d} L={L} {precision} D", den, ref, {"fp16": 1e-3, "w2": w2bar}.get(precision, 3e-4))
        _report(f"d={d} L={L} {precision} F_x", raw, raw_ref, {"fp16": 2e-3, "w2": w2bar}.get(precision, 3e-4))


@pytest.mark.parametrize("seed", [11, 23, 37])
def test_w2_margin_over_weight_seeds(ops, seed):
    """The headline mode's 5e-4 bar at the headline shape on OTHER draws of the weights (the full-size tests above all use seed 3): the
    one-term hidden layer's error depends on the activation statistics of the weights at hand, so the margin is measured over draws, not
    taken from one (the strict `mixed` mode beside it for scale)."""
    d, L, N = 384, 6, 2048
    p = W.linear_lift_state_dict(seed, d, L, cases.I, cases.H)
    x, sigma = _noisy(seed + 1, 2, N, (0.05, 5.0))
    with torch.no_grad():
        ref, raw_ref = cpu_ref.uncond_denoiser(p, "", cases.H)(x, sigma, return_raw=True)
    for precision in ("w2", "mixed"):
        den, raw = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision=precision).forward(x.cuda(), sigma.cuda(), return_raw=True)
        _report(f"seed {seed} {precision} D", den, ref, BARS[precision])
        _report(f"seed {seed} {precision} F_x", raw, raw_ref, BARS_FX[precision])


@pytest.mark.parametrize("L", [10, 14])
def test_w2_deep_networks_at_the_fused_width(ops, L):
    """d = 384 (the width the one-launch point MLP exists for) at depths the shipped configs do not reach: every layer adds its own
    one-term rounding, so F_x grows with depth — measured and printed; asserted against the north star's 1e-3 (the mode's own 5e-4 bar is
    a statement about the shipped depth L = 6, held by the tests above), the strict mode against its own bar."""
    d, N = 384, 1024
    p = W.linear_lift_state_dict(200 + L, d, L, cases.I, cases.H)
    x, sigma = _noisy(9, 2, N, (0.2, 4.0))
    with torch.no_grad():
        ref, raw_ref = cpu_ref.uncond_denoiser(p, "", cases.H)(x, sigma, return_raw=True)
    den, raw = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="w2").forward(x.cuda(), sigma.cuda(), return_raw=True)
    _report(f"d=384 L={L} w2 D", den, ref, 1e-3)
    _report(f"d=384 L={L} w2 F_x", raw, raw_ref, 1e-3)
    denm, rawm = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="mixed").forward(x.cuda(), sigma.cuda(), return_raw=True)
    assert not torch.equal(raw, rawm)   # the fused launch ran
    _report(f"d=384 L={L} mixed F_x", rawm, raw_ref, 3e-4)


# ------------------------------------------------------------------------------------------------- training path at full size
def _edm_loss(D, data, noise, sigma):
    """EDMLoss with injected draws (reference diffusion.py:136-143): 100 (s^2 + 1) / s^2 |D(x + s n) - x|^2, mean."""
    s = sigma.reshape(-1, 1, 1)
    return (100.0 * (s ** 2 + 1.0) / s ** 2 * (D(data + noise * s, sigma) - data) ** 2).mean()


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_c2_full_size_gradients_vs_oracle(ops, precision):
    """The TRAINING path at the headline size (N = 2048, d = 384, L = 6): EDM loss and the gradient of every parameter,
    HIP autograd Functions (gecco_amd/autograd.py) against torch autograd through the oracle on the host cores.  The mixed /
    fp16 modes train in split-bf16 outside an autocast region (no loss scaling), so "bf16x3" is what `bench.py --train` runs without
    `--amp`; the 16-mixed setting: tests/test_hip_amp.py."""
    from tests.test_modules_cpu import build_uncond, uncond_state_dict
    d, L, N, B = 384, 6, 2048, 2
    p = W.linear_lift_state_dict(3, d, L, cases.I, cases.H)
    rs = np.random.RandomState(11)
    data = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
    noise = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
    sigma = torch.tensor([0.1, 5.0])
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref_loss = _edm_loss(cpu_ref.uncond_denoiser(pr, "", cases.H), data, noise, sigma)
    ref_loss.backward()
    m = build_uncond(d, L)
    m.load_state_dict(uncond_state_dict(p), strict=True)
    m = m.cuda().train()
    old = ops.default_precision()
    ops.set_default_precision(precision)
    try:
        loss = _edm_loss(lambda x, s: m(x, s, None), data.cuda(), noise.cuda(), sigma.cuda())
        loss.backward()
    finally:
        ops.set_default_precision(old)
    lv, rv = float(loss.detach()), float(ref_loss.detach())
    print(f"C2 training [{precision}]: loss {lv:.6f} (oracle {rv:.6f})")
    assert abs(lv - rv) / abs(rv) < (1e-5 if precision == "fp32" else 1e-4)
    bar = 2e-4 if precision == "fp32" else 2e-3
    worst = ("", 0.0)
    grads = {k[len("backbone.model."):]: q.grad for k, q in m.named_parameters() if k.startswith("backbone.model.")}
    assert set(grads) == set(p)
    for k in p:
        assert grads[k] is not None and pr[k].grad is not None, k
        e = cpu_ref.rel_err(grads[k].cpu(), pr[k].grad)[0]
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < bar, (precision, k, e)
    print(f"C2 training [{precision}]: worst parameter gradient {worst[0]} {worst[1]:.2e} (bar {bar:.0e})")
    assert worst[1] > 0.0   # gradients were really compared


@pytest.mark.parametrize("cfg", ["C3", "C4", "C3-amp", "C4-amp"])
def test_conditional_full_size_training_step_vs_oracle(ops, cfg):
    """"C3-amp" / "C4-amp" (d = 512: the K = 512 forms of the A-stationary fp16 kernels): the same step in the reference's trainer setting (torch.autocast(float16), scaled loss: tests/test_hip_amp.py) — the
    denoiser's AND the conditioner's linears with fp16 operands; bars from the reference's own deviation in that setting (2.1e-3
    overall, 4e-3 per tensor at d = 384: profiles/r04h_autocast_grad_deviation.txt): 6e-3 per tensor here, where the conditioner's
    small first-stage matrices sit behind three more stages of rounding.
    The image-conditional training step at the C3 size (224 x 224 images -> ConvNeXt-T pyramids 96 / 192 / 384 channels ->
    projective lookup -> RayNetwork, N = 2048, d = 384, L = 6) and at the C4 size — BASELINE's data-parallel training
    configuration: 256 x 256 images, N = 4096, d = 512 — conditioner TRAINED as in the reference: the loss and the gradient of
    every parameter — conditioner and denoiser — against torch autograd through the oracle chain convnext_features ->
    cond_denoiser -> EDM loss (split-bf16, what `bench.py --train --config C3 | C4` runs)."""
    from gecco_amd.models.feature_pyramid import ConvNeXtExtractor
    from gecco_amd.structs import Context3d
    from tests.test_hip_convnext import _seeded_state
    from tests.test_modules_cpu import build_cond
    amp = cfg.endswith("-amp")
    cfg = cfg[:2]
    d, L, N, hw, B = (384, 6, 2048, 224, 2) if cfg == "C3" else (512, 6, 4096, 256, 2)
    cn = ConvNeXtExtractor(n_stages=3, model="tiny", pretrained=False)
    csd = _seeded_state(cn, 9)
    cn.load_state_dict(csd, strict=True)
    m = build_cond(d, L, conditioner=cn)
    p = W.ray_network_state_dict(17, d, L, cases.I, cases.H)
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.uvl_mean"], sd["reparam.uvl_std"] = p["reparam.uvl_mean"], p["reparam.uvl_std"]
    sd.update({"conditioner." + k: v for k, v in csd.items()})
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    rs = np.random.RandomState(3)
    img = torch.from_numpy(rs.rand(B, 3, hw, hw).astype(np.float32))
    _, K = W.synthetic_context(4, B, hw=hw)
    data = torch.from_numpy((0.5 * rs.randn(B, N, 3)).astype(np.float32))
    noise = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
    sigma = torch.tensor([0.3, 2.0])
    cp = {k: v.clone().requires_grad_(True) for k, v in csd.items()}
    pr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.startswith("reparam.") else v) for k, v in p.items()}
    feats = cpu_ref.convnext_features(img, cp)
    ref_loss = _edm_loss(cpu_ref.cond_denoiser(pr, "", cases.H, K, feats), data, noise, sigma)
    ref_loss.backward()
    old = ops.default_precision()
    ops.set_default_precision("bf16x3")
    try:
        ctx = Context3d(image=img.cuda(), K=K.cuda())
        scale = 2.0 ** 9 if amp else 1.0
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            loss = _edm_loss(lambda x, s: m(x, s, ctx), data.cuda(), noise.cuda(), sigma.cuda())
        (loss * scale).backward()
        if amp:
            for q in m.parameters():
                if q.grad is not None:
                    q.grad /= scale
    finally:
        ops.set_default_precision(old)
    lv, rv = float(loss.detach()), float(ref_loss.detach())
    print(f"{cfg} training{' [autocast fp16]' if amp else ''}: loss {lv:.6f} (oracle {rv:.6f})")
    assert abs(lv - rv) / abs(rv) < (5e-4 if amp else 1e-4)
    bar = 6e-3 if amp else 3e-3
    worst = {"conditioner": ("", 0.0), "denoiser": ("", 0.0), "denoiser matrices": ("", 0.0)}
    for k, q in m.named_parameters():
        if k.startswith("conditioner."):
            r, part = cp[k[len("conditioner."):]].grad, "conditioner"
        elif k.startswith("backbone.model."):
            r, part = pr[k[len("backbone.model."):]].grad, "denoiser"
            if q.dim() == 2 and min(q.shape) > 1:
                e2 = cpu_ref.rel_err(q.grad.cpu(), r)[0]
                worst["denoiser matrices"] = max(worst["denoiser matrices"], (k, e2), key=lambda t: t[1])
        else:
            continue
        assert q.grad is not None and r is not None, k
        e = cpu_ref.rel_err(q.grad.cpu(), r)[0]
        if amp and k.endswith(".alpha"):   # cancelling scalar sums: 2.4e-2 in the reference's own autocast run
            assert e < 5e-2, (k, e)
            continue
        worst[part] = max(worst[part], (k, e), key=lambda t: t[1])
        assert e < bar, (k, e)
    for part, (k, e) in worst.items():
        print(f"{cfg} training{' [autocast fp16]' if amp else ''}: worst {part} gradient {k} {e:.2e} (bar {bar:.0e})")
        assert e > 0.0
