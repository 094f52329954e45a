"""GPU parity of the layer / network level entry points against the golden vectors captured from
the real reference (tests/golden, tools/make_golden.py) and against the oracle on the same seeded
inputs.  Bar (BASELINE.json north_star): 1e-3 relative fp32; the fp32-MFMA path sits ~1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, cpu_ref

pytestmark = pytest.mark.gpu
TOL = 5e-5


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import hip_ops
    return hip_ops


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


def _close(got, ref, tol=TOL):
    e = cpu_ref.rel_err(got.cpu(), ref)
    assert e[0] <= tol, e
    return e


def _cuda(p):
    return {k: v.cuda() for k, v in p.items()}


@pytest.mark.parametrize("name", list(cases.LAYER_CASES))
def test_layer_golden(ops, golden_dir, name):
    g = _load(golden_dir, name)
    p, x, t = cases.layer_inputs(name)
    plan = ops.SetTransformerPlan(_cuda({"layers.0." + k: v for k, v in p.items()}), "", cases.H, cases.I)
    y, hs, _ = plan.forward_(x.cuda().clone(), t.cuda(), return_h=True)
    _close(y, g["x_out"])
    _close(hs[0], g["h"])
    N = x.shape[1]
    y2, _, _ = plan.forward_((x[:, : N // 2] * 0.5).contiguous().cuda(), t.cuda(), hs=[g["h"].cuda()])
    _close(y2, g["x_out_cached"])


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_uncond_golden(ops, golden_dir, name):
    g = _load(golden_dir, name)
    p, x, sigma = cases.uncond_inputs(name)
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I)
    den, raw = net.forward(x.cuda(), sigma.cuda(), return_raw=True)
    e1 = _close(den, g["denoised"])
    e2 = _close(raw, g["F_x"])
    print(name, "denoised", e1, "F_x", e2)
    # determinism: same inputs -> bit-identical outputs (no atomics anywhere on the path)
    den2 = net.forward(x.cuda(), sigma.cuda())
    assert torch.equal(den, den2)


@pytest.mark.parametrize("precision,tol", [("fp32", TOL), ("bf16x3", 2e-4), ("fp16", 1e-3)])
def test_cached_mode_golden(ops, golden_dir, precision, tol):
    g = _load(golden_dir, "cached_d128_L4")
    p, x, sigma, x_new = cases.cached_inputs()
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision=precision)
    den, cache = net.forward(x.cuda(), sigma.cuda(), do_cache=True)
    # the cached inducer states are internal tensors (the 1e-3 bar is on the forward's output): 5x slack in the
    # reduced-precision modes, where the max-norm is carried by a few inducers
    e = _close(torch.stack(cache), g["cache"], tol if precision == "fp32" else 5 * tol)
    out = net.forward(x_new.cuda(), sigma.cuda(), cache=[c.cuda() for c in g["cache"]])
    e2 = _close(out, g["out_new"], tol)
    print("cached", precision, "cache", e, "out_new", e2)


@pytest.mark.parametrize("precision,tol", [("fp32", TOL), ("bf16x3", 2e-4), ("mixed", 2e-4), ("w2", 5e-4), ("fp16", 1e-3)])
@pytest.mark.parametrize("B,N,d,L", [(3, 333, 128, 2), (2, 2048, 384, 1), (1, 4096, 512, 1), (2, 100, 64, 3), (2, 130, 256, 1),
                                     (2, 384, 384, 2), (1, 640, 512, 1), (3, 128, 256, 2), (2, 256, 64, 2), (2, 256, 192, 2), (1, 200, 320, 1),
                                     (2, 2048, 128, 4)])   # the last: BASELINE config C1 (airplane: d = 128, 4 layers) at its true N
def test_uncond_vs_oracle_ragged(ops, B, N, d, L, precision, tol):
    """Sizes the golden set does not hold (ragged N, d=512, N=4096, head dim 8 that stays on the fp32 attention
    kernels, rows < 128; N = 128, 384, 640: an odd number of 128-row tiles under the 256-row tiles and the activation
    images; d = 64 at N = 256: head dim 8 on the fp32 attention kernels with the hidden-layer image at K = 128; d = 192, 320: head
    dims 24, 40), oracle computed on the fly, every arithmetic mode ("w2": the one-launch point MLP at d = 128, 256, 384, 512 with whole 128-row tiles, the
    mixed mode's launches everywhere else)."""
    from oracle import weights as W
    p = W.linear_lift_state_dict(77 + N, d, L, cases.I, cases.H)
    x, sigma = W.synthetic_cloud(N, B, N)
    with torch.no_grad():
        ref, raw_ref = cpu_ref.uncond_denoiser(p, "", cases.H)(x, sigma, return_raw=True)
    den, raw = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision=precision).forward(x.cuda(), sigma.cuda(), return_raw=True)
    _close(den, ref, tol)
    _close(raw, raw_ref, tol)


def _outlier_network(d, L, smax, wmax):
    from oracle import weights as W
    p = W.linear_lift_state_dict(131 + d, d, L, cases.I, cases.H)
    rs = np.random.RandomState(d + L)
    for li in range(L):
        pre = f"inner.layers.{li}."
        for key in ("broadcast.unpool.out_proj.weight", "mlp.0.weight", "mlp.2.weight"):
            w = p[pre + key]
            for _ in range(6):
                w[rs.randint(w.shape[0]), rs.randint(w.shape[1])] = float(rs.choice([-1.0, 1.0, 0.6, -0.4])) * wmax
        if smax:   # y = (1 + s) * GroupNorm(x) + b: single channels of the h8 operand at ~smax x unit scale
            p[pre + "mlp_norm.scale.bias"][rs.randint(d)] = smax
            p[pre + "mlp_norm.scale.bias"][rs.randint(d)] = -0.75 * smax
            p[pre + "mlp_norm.bias.bias"][rs.randint(d)] = 0.5 * smax
    return p


@pytest.mark.parametrize("N,d,L", [(256, 128, 3), (384, 384, 2)])
def test_mixed_mode_on_outlier_weights_and_channels(ops, N, d, L):
    """Weights a trained checkpoint may hold, against the oracle, at the sites that run in h8 arithmetic (out_proj, mlp.0, mlp.2;
    models/set_transformer.py:112, 164-166, normalization.py:36-44), on a batch that spans sigma = 0.002 .. 165.
    (1) Isolated entries up to |w| = 8 in those matrices: the mode's bar (2e-4) holds — the fp8 operands' scales cover |w| <= 14
    (csrc/h8_scales.h; round 3's covered 1.75).
    (2) Outlier CHANNELS of the h8 operand y = AdaGN(x): the AdaGN scale / bias of mlp_norm puts single channels at |y| ~ 100 .. 500.
    Such a channel feeds the GaussianActivation with pre-activations ~100 x larger, so every finite-precision arithmetic pays its
    relative error on a ~100 x larger magnitude (tools/debug/outlier_dbg.py: the exact-fp32 HIP mode goes 6e-7 -> 9e-6, split-bf16
    3e-5 -> 4e-4): the claim that can hold, and is asserted, is that the mixed mode degrades IN PROPORTION — finite, within 6 x the
    split-bf16 mode on the same network (it is 2.4 x on well-conditioned ones) — where round 3's scales saturated the lo term
    above |y| = 56 and produced NaN above 448."""
    from oracle import weights as W
    x, sigma = W.synthetic_cloud(N, 4, N)

    def errors(p, modes):
        with torch.no_grad():
            raw_ref = cpu_ref.uncond_denoiser(p, "", cases.H)(x, sigma, return_raw=True)[1]
        pc = _cuda(p)
        out = {}
        for pr in modes:
            raw = ops.LinearLiftPlan(pc, cases.H, cases.I, precision=pr).forward(x.cuda(), sigma.cuda(), return_raw=True)[1]
            assert torch.isfinite(raw).all()
            out[pr] = cpu_ref.rel_err(raw.cpu(), raw_ref)[0]
        return out
    e = errors(_outlier_network(d, L, 0.0, 8.0), ("mixed",))
    print(f"outlier network d={d} L={L}, |w| <= 8: mixed F_x {e['mixed']:.2e}")
    assert e["mixed"] <= 2e-4, e
    e = errors(_outlier_network(d, L, 120.0, 8.0), ("bf16x3", "mixed"))
    print(f"outlier network d={d} L={L}, |w| <= 8, |y| ~ 500 channels: split-bf16 F_x {e['bf16x3']:.2e}, mixed F_x {e['mixed']:.2e}")
    assert e["mixed"] <= 6 * e["bf16x3"] and e["mixed"] <= 5e-3, e


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "mixed", "fp16"])
def test_full_size_properties(ops, precision):
    """BASELINE config C2 (B=64, N=2048, d=384, L=6) is too slow for the CPU oracle inside a test, so
    check size-independent properties: (1) samples are independent — evaluating a batch equals
    evaluating its samples in two half batches, bit for bit; (2) permutation equivariance over the
    points of a cloud (the set transformer has no positional input) within fp32 re-association."""
    from oracle import weights as W
    B, N, d, L = 64, 2048, 384, 6
    p = _cuda(W.linear_lift_state_dict(3, d, L, cases.I, cases.H))
    x, sigma = W.synthetic_cloud(0, B, N)
    x, sigma = x.cuda(), sigma.cuda()
    net = ops.LinearLiftPlan(p, cases.H, cases.I, precision=precision)
    full = net.forward(x, sigma)
    assert torch.isfinite(full).all()
    lo = net.forward(x[:32].contiguous(), sigma[:32].contiguous())
    hi = net.forward(x[32:].contiguous(), sigma[32:].contiguous())
    assert torch.equal(full, torch.cat([lo, hi]))
    perm = torch.randperm(N, device="cuda")
    full_p = net.forward(x[:4, perm].contiguous(), sigma[:4].contiguous())
    _close(full_p, full[:4, perm].cpu(), {"fp32": 1e-4, "bf16x3": 2e-4, "mixed": 4e-4, "fp16": 2e-3}[precision])
    if precision != "fp32":   # and the mode stays inside the parity bar at full size (against the exact-fp32 mode)
        exact = ops.LinearLiftPlan(p, cases.H, cases.I, precision="fp32").forward(x[:8].contiguous(), sigma[:8].contiguous())
        e = cpu_ref.rel_err(full[:8].cpu(), exact.cpu())
        print("C2", precision, "vs exact fp32 mode", e)
        assert e[0] < 1e-3, e


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_split_bf16_mode_golden(ops, golden_dir, name):
    """precision="bf16x3": hi + lo bf16 operands, three bf16 MFMAs per product, fp32 accumulate.  Bar: the
    north-star 1e-3 relative against the fp32 reference; measured ~1e-5 (plain bf16 operands would be ~6e-3)."""
    g = _load(golden_dir, name)
    p, x, sigma = cases.uncond_inputs(name)
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="bf16x3")
    den, raw = net.forward(x.cuda(), sigma.cuda(), return_raw=True)
    e1 = cpu_ref.rel_err(den.cpu(), g["denoised"])
    e2 = cpu_ref.rel_err(raw.cpu(), g["F_x"])
    print(name, "bf16x3 denoised", e1, "F_x", e2)
    assert e1[0] <= 2e-4 and e2[0] <= 2e-4, (e1, e2)
    exact = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="fp32").forward(x.cuda(), sigma.cuda())
    assert not torch.equal(den, exact)          # the mode really is a different arithmetic ...
    assert torch.equal(den, net.forward(x.cuda(), sigma.cuda()))   # ... and still deterministic


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_fp16_mode_golden(ops, golden_dir, name):
    """precision="fp16": operands rounded to fp16 (11 significant bits), one MFMA per product, fp32 accumulate.
    Bar: the north-star 1e-3 relative against the fp32 reference's golden output; measured ~3e-4."""
    g = _load(golden_dir, name)
    p, x, sigma = cases.uncond_inputs(name)
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="fp16")
    den, raw = net.forward(x.cuda(), sigma.cuda(), return_raw=True)
    e1 = cpu_ref.rel_err(den.cpu(), g["denoised"])
    e2 = cpu_ref.rel_err(raw.cpu(), g["F_x"])
    print(name, "fp16 denoised", e1, "F_x", e2)
    assert e1[0] <= 1e-3 and e2[0] <= 1e-3, (e1, e2)
    assert torch.equal(den, net.forward(x.cuda(), sigma.cuda()))   # deterministic
    half = net.forward(x[:2].contiguous().cuda(), sigma[:2].contiguous().cuda())
    assert torch.equal(half, den[:2])                               # bits do not depend on the batch


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_fp16_inducer_chain_matches_standalone_kernels(ops, golden_dir, name):
    """The one-launch inducer chain (pool merge .. unpool k|v, inducer_chain_f16.hip) against the eight stand-alone
    launches it replaces: same rounding points, GroupNorm column sums added in another order -> agreement far inside
    the mode's own error; the cached inducer states (fp32) agree to fp32 rounding of a 64-row GroupNorm."""
    g = _load(golden_dir, name)
    p, x, sigma = cases.uncond_inputs(name)
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="fp16")
    out = {}
    try:
        for chain in (0, 1):
            ops.set_option("chain", chain)
            den, hs = net.forward(x.cuda(), sigma.cuda(), do_cache=True)
            out[chain] = (den.cpu(), [c.cpu() for c in hs])
    finally:
        ops.set_option("chain", -1)
    e_fused = cpu_ref.rel_err(out[1][0], g["denoised"])
    e_ab = cpu_ref.rel_err(out[1][0], out[0][0])
    print(name, "chain vs golden", e_fused, "chain vs stand-alone", e_ab)
    assert e_fused[0] <= 1e-3, e_fused
    # Layer 0 sees identical inputs: only the GroupNorm sums differ (order), which moves an fp16 rounding in ~1e-4 of
    # the elements.  From layer 1 on every fp16 rounding downstream amplifies such a perturbation towards the mode's
    # own noise floor (delta -> sqrt(delta * 2^-11) per rounding stage), so later states agree like two fp16 runs do.
    e0 = cpu_ref.rel_err(out[1][1][0], out[0][1][0])
    assert out[1][1][0].abs().max() > 0 and e0[1] <= 5e-5 and e0[0] <= 1e-3, e0
    for a, b_ in zip(out[1][1][1:], out[0][1][1:]):
        e = cpu_ref.rel_err(a, b_)
        assert e[0] <= 5e-3 and e[1] <= 2e-3, e
    assert e_ab[0] <= 1e-3, e_ab


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_fp16_head_major_is_bit_identical(ops, golden_dir, name):
    """K | V and q stored head-major (the default) against row-major: a layout, not an arithmetic — same bits."""
    p, x, sigma = cases.uncond_inputs(name)
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="fp16")
    out = {}
    try:
        for hm in (0, 1):
            ops.set_option("headmajor", hm)
            out[hm] = net.forward(x.cuda(), sigma.cuda()).cpu()
    finally:
        ops.set_option("headmajor", -1)
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_fp16_fused_mlp_matches_two_launch_form(ops, golden_dir, name):
    """The one-launch point MLP (mlp_fused_f16.hip) against AdaGN + mlp.0 + activation (A-stationary kernel) followed by
    mlp.2 + residual + statistics: every rounding point, accumulation order and partial-sum order is the same, so the
    whole network output is — bit for bit."""
    p, x, sigma = cases.uncond_inputs(name)
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="fp16")
    out = {}
    try:
        for on in (0, 1):
            ops.set_option("mlpfused", on)
            out[on] = net.forward(x.cuda(), sigma.cuda()).cpu()
    finally:
        ops.set_option("mlpfused", -1)
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_fp16_fused_unpool_matches_two_launch_form(ops, golden_dir, name):
    """The one-launch unpool attention + out_proj (unpool_outproj_f16.hip) against the two kernels it replaces: the
    whole network output is the same bit for bit."""
    p, x, sigma = cases.uncond_inputs(name)
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="fp16")
    out = {}
    try:
        for on in (0, 1):
            ops.set_option("unpoolfused", on)
            out[on] = net.forward(x.cuda(), sigma.cuda()).cpu()
    finally:
        ops.set_option("unpoolfused", -1)
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_mixed_fp8_lo_term_matches_fp16_lo_term(ops, golden_dir, name):
    """Mixed mode: the second term of the V projection's two-term weights on the fp8 matrix instruction (option "lo8",
    v_mfma_scale_f32_32x32x64_f8f6f4 with scale 2^-19; d = 256 / 384) against the same term as fp16 — the lo term is 2^-12 of
    the product, so 3 mantissa bits of it move the output by ~2^-16 — and both against the golden reference output."""
    p, x, sigma = cases.uncond_inputs(name)
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="mixed")
    out = {}
    try:
        ops.set_option("kvq64", 0)   # "lo8" chooses the form of the lo term of the 128-column-tile kernel
        for on in (0, 1):
            ops.set_option("lo8", on)
            out[on] = net.forward(x.cuda(), sigma.cuda()).cpu()
        ops.set_option("kvq64", 1)   # the 64-column-tile kernel (default): fp8 lo term as well
        out[2] = net.forward(x.cuda(), sigma.cuda()).cpu()
    finally:
        ops.set_option("lo8", -1)
        ops.set_option("kvq64", -1)
    e = cpu_ref.rel_err(out[1], out[0])
    assert e[0] <= 6e-5, e
    e2 = cpu_ref.rel_err(out[2], out[0])
    assert e2[0] <= 6e-5, e2   # the fp8 second term moves the output by ~2^-15; through the fp16-activation chain ~4e-5
    assert cpu_ref.rel_err(out[2], torch.from_numpy(g["denoised"]))[0] <= 2e-4
    for on in (0, 1):
        eg = cpu_ref.rel_err(out[on], torch.from_numpy(g["denoised"]))
        assert eg[0] <= 2e-4, (on, eg)
    d = p["lift.weight"].shape[0]
    if d in (256, 384):
        assert not torch.equal(out[0], out[1]), "the fp8 form did not run"


@pytest.mark.parametrize("precision", ["bf16x3", "mixed"])
@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_activation_images_are_bit_identical(ops, golden_dir, name, precision):
    """Split-bf16 / mixed modes: the MLP hidden layer and the unpool attention output handed to the next GEMM as tiled split
    images and loaded global -> registers there (option "actimg", gemm_x3_areg.hip) against the fp32 hand-over through the LDS
    ring: the same hi / lo values reach the same matrix instructions in the same order — same bits."""
    p, x, sigma = cases.uncond_inputs(name)
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision=precision)
    out = {}
    try:
        ops.set_option("h8", 0)   # the h8 products exist only with the image hand-over: compare like with like
        ops.set_option("h8areg", 0)
        for on in (0, 1):
            ops.set_option("actimg", on)
            out[on] = net.forward(x.cuda(), sigma.cuda()).cpu()
    finally:
        ops.set_option("actimg", -1)
        ops.set_option("h8", -1)
        ops.set_option("h8areg", -1)
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_mixed_h8_mlp0_matches_split_bf16_mlp0(ops, golden_dir, name):
    """Mixed mode: mlp.0 as fp16 main product + two fp8 cross terms on the A-stationary kernel (option "h8", gemm_h8_astat.hip)
    against the split-bf16 product it replaces: both within the bar of the reference's golden output, and close to each other
    (the two arithmetics differ at ~2^-16 per product); the h8 form must actually have run."""
    p, x, sigma = cases.uncond_inputs(name)
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="mixed")
    out, raw = {}, {}
    try:
        for on in (0, 1):
            ops.set_option("h8", on)
            ops.set_option("h8areg", 0)
            d, r = net.forward(x.cuda(), sigma.cuda(), return_raw=True)
            out[on], raw[on] = d.cpu(), r.cpu()
        ops.set_option("h8areg", 1)   # + mlp.2 and out_proj as h8 products on h8 activation images (gemm_h8_areg.hip)
        d, r = net.forward(x.cuda(), sigma.cuda(), return_raw=True)
        out[2], raw[2] = d.cpu(), r.cpu()
    finally:
        ops.set_option("h8", -1)
        ops.set_option("h8areg", -1)
    assert cpu_ref.rel_err(out[2], torch.from_numpy(g["denoised"]))[0] <= 2e-4
    assert cpu_ref.rel_err(raw[2], raw[0])[0] <= 2e-4
    assert not torch.equal(raw[2], raw[1]), "the h8 mlp.2 / out_proj did not run"
    for on in (0, 1):
        eg = cpu_ref.rel_err(out[on], torch.from_numpy(g["denoised"]))
        assert eg[0] <= 2e-4, (on, eg)
    e = cpu_ref.rel_err(raw[1], raw[0])
    assert e[0] <= 2e-4, e
    if x.shape[1] % 128 == 0 and p["lift.weight"].shape[0] in (128, 256, 384):
        assert not torch.equal(raw[0], raw[1]), "the h8 form did not run"


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_mixed_unpool_outproj_fused_matches_the_two_launch_form(ops, golden_dir, name):
    """Mixed mode: unpool attention + out_proj (h8) + residual + statistics as ONE launch (option "unpoolh8",
    unpool_outproj_h8.hip; models/set_transformer.py:70-75, 112, 164) against the attention writing an h8 activation image for
    gemm_h8_areg.hip: same attention bits and operand split, so the whole network agrees to fp32 summation order; both within the
    bar of the reference's golden output; the fused form must actually have run; a cached evaluation takes the same path."""
    p, x, sigma = cases.uncond_inputs(name)
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="mixed")
    out, raw = {}, {}
    try:
        for on in (0, 1):
            ops.set_option("unpoolh8", on)
            d, r = net.forward(x.cuda(), sigma.cuda(), return_raw=True)
            out[on], raw[on] = d.cpu(), r.cpu()
    finally:
        ops.set_option("unpoolh8", -1)
    for on in (0, 1):
        eg = cpu_ref.rel_err(out[on], torch.from_numpy(g["denoised"]))
        assert eg[0] <= 2e-4, (on, eg)
    e = cpu_ref.rel_err(raw[1], raw[0])
    assert e[0] <= 1e-4, e   # per layer the two forms differ by <= 4e-6 of the stream's scale (tests/test_hip_ops.py); the network amplifies it
    assert not torch.equal(raw[0], raw[1]), "the fused unpool + out_proj did not run"
    d2 = net.forward(x.cuda(), sigma.cuda())
    assert torch.equal(d2.cpu(), out[1])   # reproducible run to run


def test_w2_mode_golden_and_against_the_mixed_mode(ops, golden_dir):
    """"w2" mode (precision 4): the mixed mode with the point MLP of every layer as ONE launch, the hidden layer kept in registers as
    fp16 (mlp_fused_w.hip; models/set_transformer.py:164-166).  Golden vector of the reference at d = 384 inside the mode's 5e-4 bar;
    against the mixed mode only the MLP sites differ; option "mlpw" = 0 runs the mixed mode's launches (its bits)."""
    name = "uncond_d384_L6_N128"
    p, x, sigma = cases.uncond_inputs(name)
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    w2 = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="w2")
    mixed = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="mixed")
    d, r = w2.forward(x.cuda(), sigma.cuda(), return_raw=True)
    dm, rm = mixed.forward(x.cuda(), sigma.cuda(), return_raw=True)
    eg = cpu_ref.rel_err(d.cpu(), torch.from_numpy(g["denoised"]))
    er = cpu_ref.rel_err(r.cpu(), torch.from_numpy(g["F_x"]))
    e = cpu_ref.rel_err(r.cpu(), rm.cpu())
    print(f"w2 mode {name}: D vs golden {eg[0]:.2e}, F_x vs golden {er[0]:.2e}, F_x vs the mixed mode {e[0]:.2e}")
    assert eg[0] <= 5e-4 and er[0] <= 5e-4, (eg, er)
    assert e[0] <= 5e-4, e
    assert not torch.equal(r, rm), "the one-launch point MLP did not run"
    assert torch.equal(w2.forward(x.cuda(), sigma.cuda()), d)   # reproducible run to run
    try:
        ops.set_option("mlpw", 0)
        assert torch.equal(w2.forward(x.cuda(), sigma.cuda(), return_raw=True)[1], rm)
    finally:
        ops.set_option("mlpw", -1)


@pytest.mark.parametrize("precision", ["w2", "mixed"])
def test_head_aligned_kvq_tiles_are_bit_identical(ops, precision):
    """feature_dim 384 (head dim 48): the kvq stream deals K, V and q columns to the 64-column tiles head-aligned and the projection
    kernel writes a head's (32 rows, 48) slab as three contiguous 1 KiB stores (option "kvqperm", default on) instead of 32-byte pieces
    — a store pattern, not an arithmetic: full and cached evaluations, outputs and inducer states, to the bit."""
    name = "uncond_d384_L6_N128"
    p, x, sigma = cases.uncond_inputs(name)
    p = _cuda(p)
    x, sigma = x.cuda(), sigma.cuda()
    xn = x[:, :128].contiguous()
    out = {}
    for on in (1, 0):
        net = ops.LinearLiftPlan(p, cases.H, cases.I, precision=precision, options={"kvqperm": on})
        (d, r), cache = net.forward(x, sigma, return_raw=True, do_cache=True)
        out[on] = (d, r, cache, net.forward(xn, sigma, cache=cache))
    assert torch.equal(out[1][0], out[0][0]) and torch.equal(out[1][1], out[0][1]) and torch.equal(out[1][3], out[0][3])
    assert all(torch.equal(a, b) for a, b in zip(out[1][2], out[0][2]))


@pytest.mark.parametrize("precision", ["w2", "mixed", "bf16x3", "fp16"])
def test_frozen_weights_scope_reuses_the_weight_images(ops, precision):
    """hip_ops.frozen_weights(): the evaluations of a scope share one build of the weight images per workspace
    (GeccoSetTransformer.images_ready; a sampler's 255 evaluations) — same bits as rebuilding every time, full and cached
    evaluations side by side; a new scope, `weights_changed()` and `set_option` rebuild; and the contract is real: a weight changed
    behind the scope's back is NOT seen (which is what shows that the reuse happens)."""
    name = "uncond_d384_L6_N128"
    p, x, sigma = cases.uncond_inputs(name)
    p = _cuda(p)
    x, sigma = x.cuda(), sigma.cuda()
    net = ops.LinearLiftPlan(p, cases.H, cases.I, precision=precision)
    (d0, r0), cache = net.forward(x, sigma, return_raw=True, do_cache=True)
    xn = x[:, :64].contiguous()
    c0 = net.forward(xn, sigma, cache=cache)
    with ops.frozen_weights():
        for _ in range(3):
            d, r = net.forward(x, sigma, return_raw=True)
            assert torch.equal(d, d0) and torch.equal(r, r0)
            assert torch.equal(net.forward(xn, sigma, cache=cache), c0)
        assert net.images.ready((x.shape[0], x.shape[1], 0), net.images.token(False)) == 1
        w = p["inner.layers.3.mlp.0.weight"]
        w_saved = w.clone()
        w.mul_(1.25)
        stale = net.forward(x, sigma)
        assert torch.equal(stale, d0), "the scope rebuilt its images (no reuse happened)"
        ops.weights_changed()
        fresh = net.forward(x, sigma)
        assert not torch.equal(fresh, d0)
        with ops.frozen_weights():   # nested: same generation
            assert torch.equal(net.forward(x, sigma), fresh)
    assert torch.equal(net.forward(x, sigma), fresh) and net.images.token(False) is None
    w.copy_(w_saved)
    with ops.frozen_weights():       # a new scope trusts nothing built before it
        assert torch.equal(net.forward(x, sigma), d0)


@pytest.mark.parametrize("name", list(cases.UNCOND_CASES))
def test_mixed_two_term_chain_matches_split_bf16_chain(ops, golden_dir, name):
    """Mixed mode: the 64-inducer chain of a layer as ONE launch with two-term fp16 weights (option "chain2",
    inducer_chain_f16_kernel<.., TWO>: pool merge, pool.out_proj, norm_1, broadcast.mlp, norm_2, unpool k|v —
    models/set_transformer.py:99-117) against five 64-row split-bf16 GEMMs: both within the bar of the reference's golden
    output, close to each other (the chain's activations are rounded to fp16 once: ~1e-4 on F_x), and a cached evaluation
    (which skips the chain and reads the split-bf16 k|v image) still works beside it."""
    p, x, sigma = cases.uncond_inputs(name)
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    net = ops.LinearLiftPlan(_cuda(p), cases.H, cases.I, precision="mixed")
    out, raw = {}, {}
    try:
        for on in (0, 1):
            ops.set_option("chain2", on)
            d, r = net.forward(x.cuda(), sigma.cuda(), return_raw=True)
            out[on], raw[on] = d.cpu(), r.cpu()
        _, cache = net.forward(x.cuda(), sigma.cuda(), do_cache=True)          # chain2 on: the cache it produces ...
        again = net.forward(x.cuda(), sigma.cuda(), cache=cache).cpu()         # ... and a cached evaluation of the same points
    finally:
        ops.set_option("chain2", -1)
    for on in (0, 1):
        assert cpu_ref.rel_err(out[on], torch.from_numpy(g["denoised"]))[0] <= 2e-4, on
    e = cpu_ref.rel_err(raw[1], raw[0])
    assert 0 < e[0] <= 3e-4, e
    assert cpu_ref.rel_err(again, out[1])[0] <= 2e-4


def test_split_bf16_linear_accuracy(ops):
    """The split itself: products with operands spanning 8 orders of magnitude keep ~2^-16 relative accuracy."""
    import ctypes as C
    from gecco_amd import _lib
    rs = np.random.RandomState(3)
    B, R, K, Nout = 2, 256, 384, 256
    A = torch.from_numpy((rs.randn(B, R, K) * np.exp(rs.uniform(-8, 8, size=(B, R, 1)))).astype(np.float32))
    W = torch.from_numpy((rs.randn(Nout, K) / 20).astype(np.float32))
    ref = (A.double() @ W.double().T)
    st = {"layers.0." + k: v for k, v in __import__("oracle.weights", fromlist=["x"]).layer_state_dict(rs, K, cases.I, cases.H).items()}
    # route through the network entry so the precision flag applies: a 1-layer plan whose kv_proj is W would do, but
    # the unit entry point is fp32-only by design; instead compare the two modes on the full layer
    plan32 = ops.SetTransformerPlan(_cuda(st), "", cases.H, cases.I, precision="fp32")
    plan16 = ops.SetTransformerPlan(_cuda(st), "", cases.H, cases.I, precision="bf16x3")
    x = torch.from_numpy(rs.randn(2, 256, K).astype(np.float32)).cuda()
    t = torch.tensor([[-0.5], [0.7]]).cuda()
    y32, _, _ = plan32.forward_(x.clone(), t)
    y16, _, _ = plan16.forward_(x.clone(), t)
    e = cpu_ref.rel_err(y16.cpu(), y32.cpu())
    assert 0 < e[0] < 1e-4, e


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["mixed", "fp32"])
def test_two_stream_evaluation_is_bit_identical(precision, monkeypatch):
    """hip_ops._two_stream_halves: an evaluation as two half batches on two HIP streams (the default for even batches of
    >= 32 K points) returns, to the bit, what one launch sequence over the whole batch returns — denoised cloud, raw output,
    the inducer cache, a cached evaluation of other points, and a hipGraph capture of the forked / joined sequence."""
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import hip_ops
    from oracle import weights as W
    d, L, N, B = 128, 2, 2048, 16
    p = {k: v.cuda() for k, v in W.linear_lift_state_dict(31, d, L, cases.I, cases.H).items()}
    rs = np.random.RandomState(8)
    x = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rs.randn(B, 2 * N, 3).astype(np.float32)).cuda()
    sigma = torch.from_numpy(np.exp(rs.uniform(np.log(0.002), np.log(165.0), size=B)).astype(np.float32)).cuda()

    def run(streams):
        monkeypatch.setenv("GECCO_FWD_STREAMS", streams)
        plan = hip_ops.LinearLiftPlan(p, cases.H, cases.I, precision=precision)
        (den, raw), cache = plan.forward(x, sigma, return_raw=True, do_cache=True)
        den2 = plan.forward(x2, sigma, cache=cache)
        out = torch.empty_like(x)
        plan.forward(x, sigma, out=out)           # warm-up outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            plan.forward(x, sigma, out=out)
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        return den, raw, cache, den2, out.clone()
    a, b = run("2"), run("1")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert all(torch.equal(u, v) for u, v in zip(a[2], b[2]))
    assert torch.equal(a[4], a[0])


@pytest.mark.gpu
def test_empty_batch_returns_empty_results():
    """B = 0 (a data-parallel rank that owns no cloud; the reference's torch modules return empty tensors for it): empty
    denoised / raw clouds and an empty inducer cache, nothing launched, no error."""
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import hip_ops
    from oracle import weights as W
    d, L, N = 128, 2, 256
    p = {k: v.cuda() for k, v in W.linear_lift_state_dict(31, d, L, cases.I, cases.H).items()}
    plan = hip_ops.LinearLiftPlan(p, cases.H, cases.I, precision="mixed")
    x = torch.empty(0, N, 3, device="cuda")
    sigma = torch.empty(0, device="cuda")
    (den, raw), cache = plan.forward(x, sigma, return_raw=True, do_cache=True)
    assert den.shape == (0, N, 3) and raw.shape == (0, N, 3)
    assert len(cache) == L and all(h.shape == (0, cases.I, d) for h in cache)
    torch.cuda.synchronize()


@pytest.mark.parametrize("precision,d", [("mixed", 384), ("mixed", 256), ("mixed", 512), ("fp16", 384), ("fp16", 512), ("fp16", 256)])
def test_inducer_chain_cluster_matches_one_block_chain_bitwise(ops, precision, d):
    """The cluster form of the one-launch inducer chain (option "chaincl": d / 128 blocks per sample, each streaming a third of the weights,
    handing each other h0, u and h2 as register images through L2 — inducer_chain_f16_kernel<.., CL>; models/set_transformer.py:99-117)
    computes every value from the same operands in the same order as the one-block-per-sample chain: the network output and the cached
    inducer states (which come straight out of the chain) are bit-identical; repeated runs too (the hand-offs carry no race), at a batch
    larger than the CUs can hold at once (5 * 60 blocks: later clusters start while earlier ones wait)."""
    from oracle import weights as OW
    L, N, B = 3, 256, 100
    p = _cuda(OW.linear_lift_state_dict(11, d, L, cases.I, 8))
    rs = np.random.RandomState(d)
    x = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32)).cuda()
    sigma = torch.from_numpy(np.exp(rs.uniform(-4, 4, size=B)).astype(np.float32)).cuda()
    net = ops.LinearLiftPlan(p, 8, cases.I, precision=precision)
    out = {}
    try:
        for on in (0, 1):
            ops.set_option("chaincl", on)
            den, hs = net.forward(x, sigma, do_cache=True)
            out[on] = (den.clone(), [c.clone() for c in hs])
        for _ in range(3):   # cluster form again: same bits every time
            den, hs = net.forward(x, sigma, do_cache=True)
            assert torch.equal(den, out[1][0]) and all(torch.equal(a, b) for a, b in zip(hs, out[1][1]))
    finally:
        ops.set_option("chaincl", -1)
    assert torch.isfinite(out[1][0]).all()
    # a single cloud (one cluster on the whole chip) gives the bits it has inside the batch
    ops.set_option("chaincl", 1)
    try:
        one = net.forward(x[:1].contiguous(), sigma[:1].contiguous())
    finally:
        ops.set_option("chaincl", -1)
    assert torch.equal(one[0], out[1][0][0])
    if precision == "mixed" and d == 512:
        # d = 512 runs the one-launch chain ONLY as a cluster (one block per sample loses to the five split-bf16 launches there, api.hip):
        # "chaincl" = 0 is that split-bf16 chain — the two agree like "chain2" on / off do (test_mixed_two_term_chain_...)
        e = cpu_ref.rel_err(out[1][0].cpu(), out[0][0].cpu())
        assert 0 < e[0] <= 3e-4, e
        return
    for li, (a, b) in enumerate(zip(out[1][1], out[0][1])):
        assert torch.equal(a, b), (li, float((a - b).abs().max()))
    assert torch.equal(out[1][0], out[0][0])


@pytest.mark.parametrize("d,N", [(384, 2048), (256, 640), (128, 1024)])
def test_chain_writes_the_unpool_kv_image_itself_bitwise(ops, d, N):
    """Mixed mode, option "kvfold" (default on): the inducer chain's last epilogue writes k | v of the 64 inducer states straight as the fp16
    image the fused unpool + out_proj kernel streams (per (sample, head) K rows padded, V transposed and key-permuted:
    unpool_outproj_h8.hip) instead of fp32 kvh for a reformatting pass — same sums, one rounding to fp16 either way: bit-identical network
    output; the cluster form (d = 256, 384) and the one-block form (d = 128); a cached evaluation (which does not run the chain) beside it."""
    from oracle import weights as OW
    L, B = 3, 7
    p = _cuda(OW.linear_lift_state_dict(31, d, L, cases.I, 8))
    rs = np.random.RandomState(d + N)
    x = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32)).cuda()
    sigma = torch.from_numpy(np.exp(rs.uniform(-4, 4, size=B)).astype(np.float32)).cuda()
    net = ops.LinearLiftPlan(p, 8, cases.I, precision="mixed")
    out = {}
    try:
        for on in (0, 1):
            ops.set_option("kvfold", on)
            den, hs = net.forward(x, sigma, do_cache=True)
            again = net.forward(x, sigma, cache=hs)
            out[on] = (den.clone(), [c.clone() for c in hs], again.clone())
    finally:
        ops.set_option("kvfold", -1)
    assert torch.isfinite(out[1][0]).all()
    assert torch.equal(out[1][0], out[0][0]) and torch.equal(out[1][2], out[0][2])
    for a, b in zip(out[1][1], out[0][1]):
        assert torch.equal(a, b)
