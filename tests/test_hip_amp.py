"""The training path under the reference's own trainer setting: Lightning's precision="16-mixed" of both shipped configs
(example_configs/shapenet_airplane_unconditional.py:74, taskonomy_conditional.py:102) = `training_step` (diffusion.py:213-222)
under torch.autocast(float16) with a GradScaler around the optimizer.

Under that autocast the reference's nn.Linear / in_proj products run with fp16 operands; so do the HIP path's linears then
(gecco_amd/autograd.py `_lin_precision`): forward, dX and dW products with fp16 operands, ONE MFMA per product, fp32 accumulation,
fp32 tensors between the kernels (the reference rounds those to fp16 too).  The bar is the reference's own deviation in that
setting — its algorithm differentiated under torch.autocast(float16) on the host cores is 2.1e-3 (all parameters) / 2.8e-3 (worst
weight matrix) from its fp32 gradients (tools/experiments/autocast_grad_deviation.py, profiles/r04h_autocast_grad_deviation.txt).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import cases, cpu_ref
from oracle import weights as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _build():
    from gecco_amd import _lib
    _lib.load()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _leaf(t, dev="cuda"):
    return t.clone().to(dev).requires_grad_(True)


def _rel(got, ref):
    return float((got.double().cpu() - ref.double().cpu()).norm() / ref.double().cpu().norm().clamp_min(1e-30))


def _r16(t):
    return t.half().double()


def _check_fp16_dw(N, K):
    from gecco_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(7)
    Z, R = 5, 256
    dy = _t(rs.randn(Z, R, N) * np.exp(rs.uniform(-3, 2, size=(Z, R, 1))))
    x = _t(rs.randn(Z, R, K))
    pa, po = _t(1.0 + 0.3 * rs.randn(Z, K)), _t(0.2 * rs.randn(Z, K))
    dyc, xc, pac, poc = dy.cuda(), x.cuda(), pa.cuda(), po.cuda()
    for pro in (False, True):
        xe = x * pa[:, None, :] + po[:, None, :] if pro else x
        ref16 = torch.einsum("zrn,zrk->nk", _r16(dy), _r16(xe))
        ref = torch.einsum("zrn,zrk->nk", dy.double(), xe.double())
        for group in (1, 2, 5):
            G = -(-Z // group)
            parts = torch.full((G, N, K), float("nan"), device="cuda")
            cs = torch.full((G, N), float("nan"), device="cuda")
            _lib.check(lib.gecco_gemm_tn_f16_f32(C.c_void_p(dyc.data_ptr()), C.c_void_p(xc.data_ptr()),
                                                 C.c_void_p(pac.data_ptr()) if pro else None, C.c_void_p(poc.data_ptr()) if pro else None,
                                                 C.c_void_p(parts.data_ptr()), C.c_void_p(cs.data_ptr()), Z, R, N, K, group, None),
                       "gemm_tn_f16")
            got = parts.double().sum(0).cpu()
            # (with the AdaGN apply the kernel's a x + o is one fma, the host's a mul and an add: a few values round to the other fp16)
            assert _rel(got, ref16) <= (2e-5 if pro else 2e-6), (pro, group, _rel(got, ref16))
            assert _rel(got, ref) <= 1e-3, (pro, group, _rel(got, ref))
            assert _rel(cs.double().sum(0), dy.double().sum((0, 1))) <= 1e-6   # column sums are fp32 sums of the fp32 values
            # (round 6) the general entry with the group partials summed inside the launch, by the last block to finish a tile, in group
            # order: the bits of the separate fixed-order reduction, and the counters back at zero
            from gecco_amd import autograd as ag
            tiles = lib.gecco_gemm_tn_f16_tiles(N, K)
            ctr = torch.zeros(tiles, dtype=torch.int32, device="cuda")
            pe = torch.full((G, N, K), float("nan"), device="cuda")
            ce = torch.full((G, N), float("nan"), device="cuda")
            oe, coe = torch.full((N, K), float("nan"), device="cuda"), torch.full((N,), float("nan"), device="cuda")
            for _ in range(2):   # (twice on the same counters: the kernel leaves them zero)
                _lib.check(lib.gecco_gemm_tn_f16_ex_f32(C.c_void_p(dyc.data_ptr()), 0, C.c_void_p(xc.data_ptr()), 0,
                                                        C.c_void_p(pac.data_ptr()) if pro else None, C.c_void_p(poc.data_ptr()) if pro else None,
                                                        C.c_void_p(pe.data_ptr()), C.c_void_p(ce.data_ptr()), C.c_void_p(oe.data_ptr()),
                                                        C.c_void_p(coe.data_ptr()), C.c_void_p(ctr.data_ptr()), Z, R, N, K, group, None), "gemm_tn_f16_ex")
                torch.cuda.synchronize()
                assert torch.equal(pe, parts) and torch.equal(ce, cs) and int(ctr.abs().sum()) == 0
                assert torch.equal(oe, ag._reduce(parts, N * K, G, N * K).reshape(N, K))
                assert torch.equal(coe, ag._reduce(cs, N, G, N))
            if not pro and N % 128 == 0 and K % 128 == 0:
                # (round 6) BOTH operands fp16 tensors: slabs global -> LDS by DMA (gemm_tn_f16_dma_kernel) — the same halves, the same
                # matrix instructions in the same order: the register-staged form's bits; the column sums by a ones-fragment product
                pd = torch.full((G, N, K), float("nan"), device="cuda")
                cd = torch.full((G, N), float("nan"), device="cuda")
                dy16b, x16b = dyc.half(), xc.half()
                _lib.check(lib.gecco_gemm_tn_f16_ex_f32(C.c_void_p(dy16b.data_ptr()), 1, C.c_void_p(x16b.data_ptr()), 1, None, None,
                                                        C.c_void_p(pd.data_ptr()), C.c_void_p(cd.data_ptr()), None, None, None, Z, R, N, K, group, None),
                           "gemm_tn_f16_ex(dma)")
                assert torch.equal(pd, parts)
                assert _rel(cd.double().sum(0), dy16b.double().sum((0, 1))) <= 1e-6
            if N % 8 == 0:   # dY as an fp16 tensor already (round 6: the MLP backward's du): the same product of the same halves ...
                p16a = torch.full((G, N, K), float("nan"), device="cuda")
                cs16 = torch.full((G, N), float("nan"), device="cuda")
                dy16 = dyc.half()
                _lib.check(lib.gecco_gemm_tn_f16_a16_f32(C.c_void_p(dy16.data_ptr()), C.c_void_p(xc.data_ptr()),
                                                         C.c_void_p(pac.data_ptr()) if pro else None, C.c_void_p(poc.data_ptr()) if pro else None,
                                                         C.c_void_p(p16a.data_ptr()), C.c_void_p(cs16.data_ptr()), Z, R, N, K, group, None),
                           "gemm_tn_f16_a16")
                assert torch.equal(p16a, parts)
                # ... and the bias gradient's column sums are those of the halves
                assert _rel(cs16.double().sum(0), dy16.double().sum((0, 1))) <= 1e-6
            if not pro and K % 8 == 0:   # X as an fp16 tensor already: the same product of the same halves
                p16 = torch.full((G, N, K), float("nan"), device="cuda")
                x16 = xc.half()
                _lib.check(lib.gecco_gemm_tn_f16_b16_f32(C.c_void_p(dyc.data_ptr()), C.c_void_p(x16.data_ptr()), C.c_void_p(p16.data_ptr()), None,
                                                         Z, R, N, K, group, None), "gemm_tn_f16_b16")
                assert torch.equal(p16, parts)


@pytest.mark.parametrize("N,K", [(256, 384), (768, 384), (384, 768), (96, 384), (384, 96), (96, 48), (384, 672), (4, 132)])
@pytest.mark.parametrize("wide", ["1", "0"])
def test_fp16_weight_gradient_kernel(N, K, wide):
    """gecco_gemm_tn_f16_f32 (gemm_tn_f16.hip): dW = dY^T X with both operands rounded to fp16 and one MFMA per product — against fp64 on
    the fp16-rounded operands (only the fp32 accumulation order separates them) and against fp64 on the operands themselves (the fp16
    rounding: ~3e-4); grouped partials; the AdaGN apply on X and the column sums of dY (the bias gradient) out of the same pass; X as
    an fp16 tensor (gecco_gemm_tn_f16_b16_f32) gives the same bits.  wide = 1: the 256 x 128 / 128 x 256 block tiles (the default since
    round 6); 0: 128 x 128 everywhere (GECCO_TN_F16_WIDE=0, read once per process: a child process)."""
    if wide == "1":
        _check_fp16_dw(N, K)
        return
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", f"from tests.test_hip_amp import _check_fp16_dw; _check_fp16_dw({N}, {K})"], cwd=root,
                       env={**os.environ, "GECCO_TN_F16_WIDE": "0"}, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_transposed_fp16_weight_image_equals_the_image_of_the_transposed_copy():
    """gecco_split_f16_images_f32: a transposed job writes, from W itself, the bytes the plain job writes from W.t().contiguous()
    (what the dX product of a linear streams); a ready image gives the same bits as the per-call image."""
    from gecco_amd import _lib, hip_ops
    lib = _lib.load()
    rs = np.random.RandomState(4)
    big = _t(rs.randn(3 * 384, 384)).cuda()
    for Wm in (_t(rs.randn(768, 384)).cuda(), _t(rs.randn(96, 384)).cuda(), _t(rs.randn(384, 96)).cuda(), _t(rs.randn(672, 64)).cuda(), big[384:]):
        K_img, N_img = Wm.shape                                     # image of W^T: (N_img, K_img)
        nb = lib.gecco_split_f16_image_bytes(N_img, K_img)
        a = torch.full((nb,), 7, dtype=torch.uint8, device="cuda")
        b = torch.full((nb,), 9, dtype=torch.uint8, device="cuda")
        Wt = Wm.t().contiguous()
        jobs = (_lib.GeccoSplitJob * 2)(_lib.GeccoSplitJob(Wm.data_ptr(), a.data_ptr(), N_img, K_img, Wm.stride(0), 1),
                                        _lib.GeccoSplitJob(Wt.data_ptr(), b.data_ptr(), N_img, K_img, Wt.stride(0), 0))
        _lib.check(lib.gecco_split_f16_images_f32(jobs, 2, None), "split images")
        assert torch.equal(a, b), tuple(Wm.shape)
        dy = _t(rs.randn(2, 256, K_img)).cuda()
        if lib.gecco_linear_image_ok_f16(256, K_img, N_img, 0):
            y_img = hip_ops.linear(dy, None, precision="fp16", w_image=a, w_shape=(N_img, K_img))
            y_cal = hip_ops.linear(dy, Wt, precision="fp16")
            assert torch.equal(y_img, y_cal)
            assert _rel(y_img, dy.double().cpu() @ Wt.double().cpu().t()) < 1e-3


@pytest.mark.parametrize("kind", [1, 4])   # GaussianActivation (the denoiser's), GELU (the conditioner's): smooth — a ReLU kink would flip with u's rounding
def test_mlp_function_under_autocast_runs_fp16_linears(kind):
    """LinearActLinearFn / ActLinearFn / LinearFn under torch.autocast(float16): outputs and every gradient against torch autograd
    in fp64 at fp16-operand accuracy, the forward equal to the fp16 kernel's bits (the arithmetic really is the one-MFMA one, not the
    split-bf16 default), the activation's backward as the epilogue of the fp16 dX product (GECCO_TRAIN_ACTBWD=0: same values from
    two kernels)."""
    from gecco_amd import autograd as ag
    from gecco_amd import hip_ops
    rs = np.random.RandomState(3 + kind)
    prev = hip_ops.default_precision()
    hip_ops.set_default_precision("mixed")
    try:
        B, R, K, Wd = 2, 384, 384, 768
        x = _t(rs.randn(B, R, K))
        W0, b0 = _t(rs.randn(Wd, K) / np.sqrt(K)), _t(rs.randn(Wd) * 0.1)
        W2, b2 = _t(rs.randn(K, Wd) / np.sqrt(Wd)), _t(rs.randn(K) * 0.1)
        res, dy = _t(rs.randn(B, R, K)), _t(rs.randn(B, R, K))
        alpha = torch.tensor(0.8)

        def act64(u, a):
            if kind == 1:
                return (torch.exp(-u ** 2 / (2 * a ** 2)) - 0.7) / 0.28
            return torch.nn.functional.gelu(u)
        x64, a64, W064, W264, r64 = (t.double().requires_grad_(True) for t in (x, alpha, W0, W2, res))
        y64 = act64(x64 @ W064.t() + b0.double(), a64) @ W264.t() + b2.double() + r64
        (y64 * dy.double()).sum().backward()

        def run(amp):
            xg, ag_, W0g, b0g, W2g, b2g, rg = (_leaf(t) for t in (x, alpha, W0, b0, W2, b2, res))
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                y = ag.LinearActLinearFn.apply(xg, W0g, b0g, ag_ if kind == 1 else None, W2g, b2g, rg, kind)
            assert y.dtype == torch.float32
            y.backward(dy.cuda())
            return y.detach(), xg.grad, W0g.grad, b0g.grad, W2g.grad, b2g.grad, rg.grad, (ag_.grad if kind == 1 else None)
        amp, plain = run(True), run(False)
        assert not torch.equal(amp[0], plain[0])                    # another arithmetic ran
        refs = (y64.detach(), x64.grad, W064.grad, None, W264.grad, None, r64.grad, a64.grad if kind == 1 else None)
        for i, (g, r) in enumerate(zip(amp, refs)):
            if r is not None:
                bar = 2e-3 if i != 7 else 2e-2
                assert _rel(g, r) < bar, (i, _rel(g, r))
                assert _rel(plain[i], r) < _rel(g, r) + 1e-7        # and split-bf16 is the tighter one
        assert torch.equal(amp[6], dy.cuda())                        # the skip connection's gradient is dy itself
        # the hidden layer's forward in the fp16 kernel's own bits
        with torch.autocast("cuda", dtype=torch.float16):
            y1 = ag.LinearFn.apply(x.cuda(), W0.cuda(), b0.cuda())
        assert torch.equal(y1, hip_ops.linear(x.cuda(), W0.cuda(), b0.cuda(), precision="fp16"))
    finally:
        hip_ops.set_default_precision(prev)


def test_mlp_backward_gradient_as_halves_is_the_same_arithmetic(monkeypatch):
    """Round 6: inside AdaGNMlpFn's backward du = (dout W2) act'(u) leaves the dX product's epilogue as an fp16 tensor
    (gecco_linear_astat16_actbwd_h16) and its two consumers read those halves (gecco_gemm_tn_f16_a16_f32, gecco_linear_dotstats_a16_f32) —
    the operand bits they would round the fp32 tensor to: every gradient of the block is BIT-identical to the fp32-du path
    (GECCO_TRAIN_DU16=0), except mlp.0's bias gradient, which is now the column sum of the halves (as in the reference's autocast backward,
    where this gradient is an fp16 tensor)."""
    from gecco_amd import autograd as ag
    from gecco_amd import hip_ops
    rs = np.random.RandomState(11)
    prev = hip_ops.default_precision()
    hip_ops.set_default_precision("mixed")
    try:
        B, R, K, Wd, G = 3, 256, 384, 768, 32
        x, t = _t(rs.randn(B, R, K) * 2 + 0.3), _t(rs.randn(B, 1))
        sw, sb, bw, bb = _t(rs.randn(K, 1) * 0.3), _t(1 + 0.1 * rs.randn(K)), _t(rs.randn(K, 1) * 0.3), _t(0.1 * rs.randn(K))
        W0, b0 = _t(rs.randn(Wd, K) / np.sqrt(K)), _t(rs.randn(Wd) * 0.1)
        W2, b2 = _t(rs.randn(K, Wd) / np.sqrt(Wd)), _t(rs.randn(K) * 0.1)
        alpha, dy = torch.tensor(0.8), _t(rs.randn(B, R, K) * 64.0)     # (a loss-scaled gradient)

        def run(du16):
            monkeypatch.setenv("GECCO_TRAIN_DU16", du16)
            leaves = [_leaf(v) for v in (x, t, sw, sb, bw, bb, W0, b0, alpha, W2, b2)]
            xg, tg, swg, sbg, bwg, bbg, W0g, b0g, ag_, W2g, b2g = leaves
            seen = {}
            orig = ag._act_linear_dx

            def spy(*a, **k):
                r = orig(*a, **k)
                seen["du"] = r[0].dtype
                return r
            monkeypatch.setattr(ag, "_act_linear_dx", spy)
            with torch.autocast("cuda", dtype=torch.float16):
                y = ag.AdaGNMlpFn.apply(xg, tg, swg, sbg, bwg, bbg, G, 1e-5, None, W0g, b0g, ag_, W2g, b2g, 1, False)
            y.backward(dy.cuda())
            monkeypatch.setattr(ag, "_act_linear_dx", orig)
            return seen["du"], y.detach(), [v.grad for v in leaves]
        d16, y16, g16 = run("1")
        d32, y32, g32 = run("0")
        assert d16 == torch.float16 and d32 == torch.float32, (d16, d32)
        assert torch.equal(y16, y32)
        names = ["x", "t", "scale.w", "scale.b", "bias.w", "bias.b", "W0", "b0", "alpha", "W2", "b2"]
        for n, a, b in zip(names, g16, g32):
            assert torch.isfinite(a).all(), n
            if n == "b0":
                assert _rel(a, b) <= 1e-3 and not torch.equal(a, b), (n, _rel(a, b))   # (the fp16 rounding of its 768 x B R summands)
            else:
                assert torch.equal(a, b), (n, _rel(a, b))
    finally:
        hip_ops.set_default_precision(prev)


@pytest.mark.parametrize("d,N", [(384, 256), (128, 384), (512, 128)])
def test_fp16_tensors_between_the_training_kernels_are_the_same_arithmetic(d, N, monkeypatch):
    """Round 6 (`autograd._io16_ok`): under autocast(float16) K | V, q, the unpool attention's output and their gradients are fp16
    TENSORS between the kernels (as in the reference's autocast run) instead of fp32 tensors every consumer rounds on the way in.
    A whole training step's loss and parameter gradients against the fp32-tensor path (GECCO_TRAIN_IO16=0): the same operand bits
    everywhere, so the two agree far inside fp16 rounding (most gradients to the bit; the in_proj / out_proj bias gradients and the
    attention's row statistics see the halves)."""
    from gecco_amd import autograd as ag
    from gecco_amd import hip_ops
    from tests.test_modules_cpu import build_uncond, uncond_state_dict
    prev = hip_ops.default_precision()
    hip_ops.set_default_precision("mixed")
    try:
        L = 2
        sd = uncond_state_dict(W.linear_lift_state_dict(21, d, L, cases.I, cases.H))
        g = torch.Generator().manual_seed(9)
        data = torch.randn(3, N, 3, generator=g).cuda()
        noise = torch.randn(3, N, 3, generator=g).cuda()
        sigma = torch.tensor([0.05, 0.7, 9.0]).cuda()

        def run(io16):
            monkeypatch.setenv("GECCO_TRAIN_IO16", io16)
            ag.WEIGHT_IMAGES.__init__()
            m = build_uncond(d, L)
            m.load_state_dict(sd, strict=True)
            m = m.cuda().train()
            seen = []
            orig = ag.PoolAttnFn.forward

            def spy(ctx, KV, ind, H):
                seen.append(KV.dtype)
                return orig(ctx, KV, ind, H)
            monkeypatch.setattr(ag.PoolAttnFn, "forward", staticmethod(spy))
            s_ = sigma.reshape(-1, 1, 1)
            with torch.autocast("cuda", dtype=torch.float16):
                den = m(data + noise * s_, sigma, None)
                loss = (100.0 * (s_ ** 2 + 1.0) / s_ ** 2 * (den.float() - data) ** 2).mean()
            (loss * 1024.0).backward()
            monkeypatch.setattr(ag.PoolAttnFn, "forward", staticmethod(orig))
            return seen, float(loss.detach()), {n: p.grad.detach().clone() / 1024.0 for n, p in m.named_parameters()}
        s16, l16, g16 = run("1")
        s32, l32, g32 = run("0")
        assert s16 and all(t == torch.float16 for t in s16) and all(t == torch.float32 for t in s32), (s16, s32)
        assert abs(l16 - l32) <= 2e-4 * abs(l32), (l16, l32)
        same, worst = 0, (0.0, "")
        for n in g32:
            assert torch.isfinite(g16[n]).all(), n
            e = _rel(g16[n], g32[n])
            same += int(torch.equal(g16[n], g32[n]))
            worst = max(worst, (e, n))
            assert e <= 2e-3, (n, e)
        print(f"d={d} N={N}: loss {l16:.6f} vs {l32:.6f}; {same} of {len(g32)} gradients bit-identical; worst {worst[1]} {worst[0]:.2e}")
    finally:
        hip_ops.set_default_precision(prev)


def test_in_proj_gradients_on_the_side_stream_are_the_same_bits(monkeypatch):
    """Round 6: the weight gradients with respect to nn.MultiheadAttention's packed in_proj thirds (q_proj's, the inducers' k | v
    projection's) and the join of the thirds run on the weight-gradient side stream like every other weight gradient
    (`InProjSplitFn`, `_inproj_side_ok`) instead of on the main stream's critical path: a scheduling change — every gradient of a step
    to the bit, two steps in a row (the second accumulates nothing: zero_grad(set_to_none=True))."""
    from gecco_amd import autograd as ag
    from gecco_amd import hip_ops
    from tests.test_modules_cpu import build_uncond, uncond_state_dict
    prev = hip_ops.default_precision()
    hip_ops.set_default_precision("mixed")
    try:
        d, L, N = 384, 2, 256
        sd = uncond_state_dict(W.linear_lift_state_dict(23, d, L, cases.I, cases.H))
        g = torch.Generator().manual_seed(10)
        data = torch.randn(3, N, 3, generator=g).cuda()
        noise = torch.randn(3, N, 3, generator=g).cuda()
        sigma = torch.tensor([0.05, 0.7, 9.0]).cuda()

        def run(side):
            monkeypatch.setenv("GECCO_TRAIN_INPROJ_SIDE", side)
            ag.WEIGHT_IMAGES.__init__()
            m = build_uncond(d, L)
            m.load_state_dict(sd, strict=True)
            m = m.cuda().train()
            out = []
            for _ in range(2):
                m.zero_grad(set_to_none=True)
                s_ = sigma.reshape(-1, 1, 1)
                with torch.autocast("cuda", dtype=torch.float16):
                    den = m(data + noise * s_, sigma, None)
                    loss = (100.0 * (s_ ** 2 + 1.0) / s_ ** 2 * (den.float() - data) ** 2).mean()
                (loss * 256.0).backward()
                torch.cuda.synchronize()
                out.append({n: p.grad.detach().clone() for n, p in m.named_parameters()})
            return out
        a, b = run("1"), run("0")
        for ga, gb in zip(a, b):
            for n in gb:
                assert torch.isfinite(ga[n]).all() and torch.equal(ga[n], gb[n]), n
    finally:
        hip_ops.set_default_precision(prev)


def test_fused_adam_in_the_grad_scaler_protocol_matches_torch():
    """scaler.step(FusedAdamEMA(amp_on_device=True)): the optimizer then declares `_step_supports_amp_scaling` (opt-in, per instance: with
    it Lightning's MixedPrecision plugin would skip unscale_ and refuse gradient clipping), so torch.amp.GradScaler hands it the scale and
    found_inf tensors and never reads found_inf back on the host.  Against torch.optim.Adam driven by a GradScaler the ordinary way
    (unscale_, host decision): same parameters (1e-6) over steps that include an overflow (skipped: nothing moves, Adam's step
    count does not advance, the scale backs off), and the optimizer state that is saved afterwards carries torch's step count."""
    from gecco_amd.optim import FusedAdamEMA
    rs = np.random.RandomState(1)
    shapes = [(64, 48), (48,), (7, 5), (1,)]
    init = [_t(rs.randn(*s)) for s in shapes]
    grads = [[_t(rs.randn(*s) * 0.1) for s in shapes] for _ in range(6)]
    bad = 2                                                            # the step whose gradients overflow
    # the reference side: torch.optim.Adam's single-tensor path on the host (what FusedAdamEMA follows op for op, tests/test_optim_ckpt.py)
    # under GradScaler's rules (torch/amp/grad_scaler.py: unscale by 1 / scale, skip on inf and back off by 0.5, grow by 2 after
    # `growth_interval` clean steps)
    ps = [torch.nn.Parameter(t.clone()) for t in init]
    ref_opt = torch.optim.Adam(ps, lr=1e-2, foreach=False)
    scale, tracker, ref_scales = 2.0 ** 10, 0, []
    for it, gs in enumerate(grads):
        if it == bad:
            scale, tracker = scale * 0.5, 0
        else:
            for p, g in zip(ps, gs):
                p.grad = (g * scale) * (1.0 / scale)
            ref_opt.step()
            tracker += 1
            if tracker == 2:
                scale, tracker = scale * 2.0, 0
        ref_scales.append(scale)
    ref_params = [p.detach().clone() for p in ps]
    # the HIP side: a real torch.amp.GradScaler around FusedAdamEMA
    ps = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    fused = FusedAdamEMA(ps, lr=1e-2, ema_decay=0.9, amp_on_device=True)
    assert not hasattr(FusedAdamEMA(ps[:1], lr=1e-2), "_step_supports_amp_scaling")   # the default leaves the scaler's host path
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10, growth_interval=2)
    scales = []
    for it, gs in enumerate(grads):
        fused.zero_grad(set_to_none=True)
        scaler.scale(torch.zeros(1, device="cuda"))                # (what scaler.scale(loss) does first: the scale tensor exists)
        sc = scaler.get_scale()
        for p, g in zip(ps, gs):
            p.grad = (g.cuda() * sc).clone()
        if it == bad:
            ps[0].grad.view(-1)[3] = float("inf")
        scaler.step(fused)
        scaler.update()
        scales.append(scaler.get_scale())
    torch.cuda.synchronize()
    assert scales == ref_scales                                        # the scale's history, incl. the back-off
    for a, b in zip(ref_params, ps):   # 1e-6: host libm / ISA differences, as in tests/test_optim_ckpt.py (a step count off by one after the
        assert _rel(b.detach(), a) <= 1e-6   # skip would move the bias corrections, and the parameters, by percents)
    assert fused.adam_steps_taken == len(grads) - 1
    st = fused.state_dict()
    steps = {float(v["step"]) for v in (st["opt"]["state"] if "opt" in st else st["state"]).values()}
    assert steps == {float(len(grads) - 1)} == {float(v["step"]) for v in ref_opt.state_dict()["state"].values()}


def test_grad_scaler_unscale_then_clip_then_step():
    """Lightning's order when the trainer clips gradients (both shipped configs do: precision="16-mixed" with gradient_clip_val = 1,
    SURVEY section 3): scaler.unscale_(optimizer) -> clip_grad_norm_ -> scaler.step(optimizer).  That order exists only for an optimizer
    WITHOUT `_step_supports_amp_scaling` (Lightning's plugin skips unscale_ and refuses to clip otherwise) — FusedAdamEMA's default: the
    scaler unscales, reads found_inf on the host and calls the plain step.  With amp_on_device=True a hand-written loop may still unscale
    first: the scaler then hands over `grad_scale = None` with the found_inf of the unscale pass and the kernel must not divide again.
    Both against torch.optim.Adam's single-tensor path on the host with the same clipping."""
    from gecco_amd.optim import FusedAdamEMA
    rs = np.random.RandomState(2)
    shapes = [(32, 24), (24,), (5, 3)]
    init = [_t(rs.randn(*s)) for s in shapes]
    grads = [[_t(rs.randn(*s)) for s in shapes] for _ in range(3)]
    ps = [torch.nn.Parameter(t.clone()) for t in init]
    ref_opt = torch.optim.Adam(ps, lr=1e-2, foreach=False)
    for gs in grads:
        for p, g in zip(ps, gs):
            p.grad = (g * 512.0) * (1.0 / 512.0)
        torch.nn.utils.clip_grad_norm_(ps, 1.0, foreach=False)
        ref_opt.step()
    for on_device in (False, True):
        ps2 = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        fused = FusedAdamEMA(ps2, lr=1e-2, ema_decay=None, amp_on_device=on_device)
        scaler = torch.amp.GradScaler("cuda", init_scale=512.0, growth_interval=1000)
        for gs in grads:
            fused.zero_grad(set_to_none=True)
            scaler.scale(torch.zeros(1, device="cuda"))
            for p, g in zip(ps2, gs):
                p.grad = (g.cuda() * scaler.get_scale()).clone()
            scaler.unscale_(fused)
            torch.nn.utils.clip_grad_norm_(ps2, 1.0)
            scaler.step(fused)
            scaler.update()
        torch.cuda.synchronize()
        for a, b in zip(ps, ps2):
            assert _rel(b.detach(), a.detach()) <= 2e-6, on_device
        assert fused.adam_steps_taken == 3


def test_c2_full_size_gradients_under_autocast_vs_oracle():
    """The training path at the headline size (N = 2048, d = 384, L = 6) in the reference's trainer setting: EDM loss under
    torch.autocast(float16), scaled loss, every parameter gradient against torch autograd through the oracle in fp32.  Bars: the
    reference's own algorithm under torch.autocast(float16) is 2.1e-3 (all parameters) and up to 2.8e-3 per weight matrix, 4e-3 per
    tensor, from its fp32 gradients (profiles/r04h_autocast_grad_deviation.txt); the HIP path keeps fp32 tensors between its kernels
    and fp32 weight gradients and must stay below that: 2e-3 overall, 3e-3 per matrix, 4e-3 per tensor (the GaussianActivation alpha
    scalars — cancelling sums over the whole batch, 2.4e-2 in the reference's own setting — 5e-2)."""
    from gecco_amd import hip_ops as ops
    from tests.test_hip_fullsize import _edm_loss
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
    ops.set_default_precision("mixed")
    scale = 2.0 ** 9
    try:
        with torch.autocast("cuda", dtype=torch.float16):
            loss = _edm_loss(lambda x, s: m(x, s, None), data.cuda(), noise.cuda(), sigma.cuda())
        (loss * scale).backward()
    finally:
        ops.set_default_precision(old)
    lv, rv = float(loss.detach()), float(ref_loss.detach())
    print(f"C2 training [autocast fp16]: loss {lv:.6f} (oracle {rv:.6f})")
    assert abs(lv - rv) / abs(rv) < 3e-4
    grads = {k[len("backbone.model."):]: q.grad / scale for k, q in m.named_parameters() if k.startswith("backbone.model.")}
    assert set(grads) == set(p)
    num = den = 0.0
    worst, worst_m = ("", 0.0), ("", 0.0)
    for k in p:
        assert bool(torch.isfinite(grads[k]).all()), k
        g, r = grads[k].double().cpu(), pr[k].grad.double()
        e = float((g - r).norm() / r.norm())
        num, den = num + float(((g - r) ** 2).sum()), den + float((r ** 2).sum())
        if k.endswith(".alpha"):
            assert e < 5e-2, (k, e)
            continue
        worst = max(worst, (k, e), key=lambda t: t[1])
        if r.dim() == 2 and min(r.shape) > 1:
            worst_m = max(worst_m, (k, e), key=lambda t: t[1])
        assert e < 4e-3, (k, e)
    tot = (num / den) ** 0.5
    print(f"C2 training [autocast fp16]: gradient rel-L2 over all parameters {tot:.2e} (bar 2e-3), worst matrix {worst_m[0]} {worst_m[1]:.2e} "
          f"(bar 3e-3), worst tensor {worst[0]} {worst[1]:.2e} (bar 4e-3)")
    assert tot < 2e-3 and worst_m[1] < 3e-3


@pytest.mark.parametrize("K,B,R", [(384, 3, 256), (512, 2, 128), (256, 1, 384), (128, 2, 256)])
def test_astat16_training_linears(K, B, R):
    """The A-stationary fp16 forms of the training path's products (gecco_linear_astat16_f32 / _keep / _actbwd, gemm_h8_astat.hip): against
    fp64 on the operands (fp16-operand accuracy) and against the LDS-DMA fp16 GEMM of the same arithmetic (only the fp32 accumulation
    order differs); ready streams (gecco_astat16_images_f32, incl. the stream of W^T from W) give the per-call bits."""
    from gecco_amd import _lib, hip_ops
    lib = _lib.load()
    rs = np.random.RandomState(K + R)
    N1, N2, Wd = 2 * K, K, 2 * K
    x = _t(rs.randn(B, R, K)).cuda()
    pa, po = _t(1.0 + 0.3 * rs.randn(B, K)).cuda(), _t(0.2 * rs.randn(B, K)).cuda()
    W1, W2, b2 = _t(rs.randn(N1, K) / np.sqrt(K)).cuda(), _t(rs.randn(N2, K) / np.sqrt(K)).cuda(), _t(rs.randn(N2) * 0.1).cuda()
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    ws = lambda n: torch.empty(n, dtype=torch.uint8, device="cuda")     # noqa: E731
    # (1) AdaGN(x) -> K | V, q
    c1, c2 = torch.full((B, R, N1), float("nan"), device="cuda"), torch.full((B, R, N2), float("nan"), device="cuda")
    w = ws(lib.gecco_astat16_image_bytes(N1, K) + lib.gecco_astat16_image_bytes(N2, K))
    _lib.check(lib.gecco_linear_astat16_f32(p(x), p(pa), p(po), p(W1), None, N1, p(c1), p(W2), p(b2), N2, p(c2), None, 0, B, R, K, p(w), None), "astat16")
    xe = (x * pa[:, None] + po[:, None]).double()
    assert _rel(c1, xe @ W1.double().t()) < 1e-3 and _rel(c2, xe @ W2.double().t() + b2.double()) < 1e-3
    r1, r2 = hip_ops.linear_pair(x, W1, None, W2, b2, pro=(pa, po), precision="fp16")
    assert _rel(c1, r1) < 3e-6 and _rel(c2, r2) < 3e-6
    # ready streams equal the per-call ones
    w2 = ws(w.numel())
    o2 = lib.gecco_astat16_image_bytes(N1, K)
    jobs = (_lib.GeccoSplitJob * 2)(_lib.GeccoSplitJob(W1.data_ptr(), w2.data_ptr(), N1, K, K, 0),
                                    _lib.GeccoSplitJob(W2.data_ptr(), w2.data_ptr() + o2, N2, K, K, 0))
    _lib.check(lib.gecco_astat16_images_f32(jobs, 2, None), "astat16 images")
    assert torch.equal(w, w2)
    d1, d2 = torch.empty_like(c1), torch.empty_like(c2)
    _lib.check(lib.gecco_linear_astat16_f32(p(x), p(pa), p(po), None, None, N1, p(d1), None, p(b2), N2, p(d2), None, 0, B, R, K, p(w2), None), "astat16 ready")
    assert torch.equal(c1, d1) and torch.equal(c2, d2)
    # (2) a dX product: dy (B, R, K) through a (K, Nout) weight, its W^T stream straight from W
    Wo = _t(rs.randn(K, N2) / np.sqrt(K)).cuda()
    dy = _t(rs.randn(B, R, K)).cuda()
    dx = torch.full((B, R, N2), float("nan"), device="cuda")
    wt = ws(lib.gecco_astat16_image_bytes(N2, K))
    _lib.check(lib.gecco_linear_astat16_f32(p(dy), None, None, p(Wo), None, N2, p(dx), None, None, 0, None, None, 1, B, R, K, p(wt), None), "astat16 T")
    assert _rel(dx, dy.double() @ Wo.double()) < 1e-3
    other = _t(rs.randn(B, R, N2)).cuda()                   # ... the "added onto another gradient" form is an experiment outside the shipped surface
    dx2 = torch.full((B, R, N2), float("nan"), device="cuda")
    assert lib.gecco_linear_astat16_f32(p(dy), None, None, p(Wo), None, N2, p(dx2), None, None, 0, None, p(other), 1, B, R, K, p(wt), None) != 0
    assert b"GECCO_EXPERIMENTAL" in lib.gecco_last_error()
    wt2 = ws(wt.numel())
    Wot = Wo.t().contiguous()
    jobs = (_lib.GeccoSplitJob * 1)(_lib.GeccoSplitJob(Wot.data_ptr(), wt2.data_ptr(), N2, K, K, 0))
    _lib.check(lib.gecco_astat16_images_f32(jobs, 1, None), "astat16 images")
    assert torch.equal(wt, wt2)
    # (3) the first linear of an MLP: u (fp32) and act(u) (fp16), (4) the dX product through the activation
    W0, b0 = _t(rs.randn(Wd, K) / np.sqrt(K)).cuda(), _t(rs.randn(Wd) * 0.1).cuda()
    Wm = _t(rs.randn(K, Wd) / np.sqrt(Wd)).cuda()          # the second linear's weight (K outputs, Wd inputs)
    alpha = torch.tensor([0.8], device="cuda")
    for kind in (1, 2, 3):
        u = torch.full((B, R, Wd), float("nan"), device="cuda")
        h = torch.full((B, R, Wd), float("nan"), device="cuda", dtype=torch.float16)
        w0 = ws(lib.gecco_astat16_image_bytes(Wd, K))
        _lib.check(lib.gecco_linear_astat16_keep(p(x), p(pa), p(po), p(W0), p(b0), p(alpha) if kind < 3 else None, kind, p(u), p(h), B, R, K, Wd,
                                                 p(w0), None), "astat16 keep")
        u64 = xe @ W0.double().t() + b0.double()

        def act64(t):
            if kind == 3:
                return torch.relu(t)
            y = torch.exp(-t ** 2 / (2 * 0.8 ** 2))
            return (y - 0.7) / 0.28 if kind == 1 else y
        assert _rel(u, u64) < 1e-3
        assert torch.equal(h, {1: (torch.exp(-u ** 2 / (2 * 0.8 ** 2)) - 0.7) / 0.28, 2: torch.exp(-u ** 2 / (2 * 0.8 ** 2)), 3: torch.relu(u)}[kind].half()) \
            or _rel(h.float(), act64(u.double())) < 1e-3
        # backward: du = (dy Wm) * act'(u), d alpha = sum (dy Wm) * d act / d alpha
        du = torch.full((B, R, Wd), float("nan"), device="cuda")
        ag = torch.full((B * R // 128,), float("nan"), device="cuda")
        wb = ws(lib.gecco_astat16_image_bytes(Wd, K))
        _lib.check(lib.gecco_linear_astat16_actbwd(p(dy), p(Wm), p(u), p(alpha) if kind < 3 else None, kind, p(du), p(ag) if kind < 3 else None, B, R,
                                                   K, Wd, p(wb), None), "astat16 actbwd")
        ud = u.double().requires_grad_(True)
        ad = torch.tensor(0.8, dtype=torch.float64, requires_grad=True)
        if kind == 3:
            yv = torch.relu(ud)
        else:
            yv = torch.exp(-ud ** 2 / (2 * ad ** 2))
            yv = (yv - 0.7) / 0.28 if kind == 1 else yv
        (yv * (dy.double() @ Wm.double())).sum().backward()
        assert _rel(du, ud.grad) < (2e-3 if kind < 3 else 1e-6 + 2e-3), (kind, _rel(du, ud.grad))
        if kind < 3:
            assert abs(float(ag.double().sum()) - float(ad.grad)) < 3e-3 * abs(float(ad.grad)) + 1e-2


@pytest.mark.parametrize("B,N,C,H", [(2, 1024, 384, 8), (3, 2048, 128, 8), (2, 333, 256, 8), (1, 4096, 512, 8), (5, 1500, 384, 8)])
def test_attention_fn_grads_under_autocast(B, N, C, H, monkeypatch):
    """Under torch.autocast(float16) the fused attention kernels take fp16 operands — forward and the backward that recomputes P from
    the same operands (attention_x3.hip / attention_bwd_x3.hip, template flag F16: one plane per operand, one MFMA per product) —
    as torch's SDPA / nn.MultiheadAttention do in that context: outputs and all four gradients against the exact-fp32 kernels at
    fp16-operand accuracy, and different from the split-bf16 results (the other arithmetic really ran); head dims 16 / 32 / 48 / 64,
    ragged N, several key splits / query chunks."""
    from gecco_amd import hip_ops
    from gecco_amd.autograd import PoolAttnFn, UnpoolAttnFn
    monkeypatch.delenv("GECCO_TRAIN_ATTN16", raising=False)
    monkeypatch.delenv("GECCO_TRAIN_AMP", raising=False)
    hd = C // H
    rs = np.random.RandomState(N + C)
    KV, ind, g = _t(rs.randn(B, N, 2 * C)), _t(rs.randn(1, H, 64, hd)), _t(rs.randn(B, 64, C))
    q, kvh, g2 = _t(rs.randn(B, N, C)), _t(rs.randn(B, 64, 2 * C)), _t(rs.randn(B, N, C))
    out = {}
    for mode in ("fp32", "bf16x3", "amp"):
        hip_ops.set_default_precision("fp32" if mode == "fp32" else "bf16x3")
        try:
            KVg, indg, qg, kvg = _leaf(KV), _leaf(ind), _leaf(q), _leaf(kvh)
            with torch.autocast("cuda", dtype=torch.float16, enabled=mode == "amp"):
                o1 = PoolAttnFn.apply(KVg, indg, H)
                o2 = UnpoolAttnFn.apply(qg, kvg, H)
            assert o1.dtype == torch.float32 and o2.dtype == torch.float32
            o1.backward(g.cuda())
            o2.backward(g2.cuda())
            out[mode] = [o1.detach(), o2.detach()] + [t.grad for t in (KVg, indg, qg, kvg)]
        finally:
            hip_ops.set_default_precision("fp32")
    for i, (a, b, c) in enumerate(zip(out["amp"], out["fp32"], out["bf16x3"])):
        assert _rel(a, b) < 2e-3, (i, _rel(a, b))
        assert _rel(c, b) < 1e-4
        assert not torch.equal(a, c)


def test_scaler_dynamics_on_a_real_run():
    """120 steps of the 16-mixed setting with a fast-growing loss scale (growth_interval 8): the scale climbs until the fp16 operands
    of a backward product overflow, the infinities reach the weight gradients, GradScaler sees them, FusedAdamEMA skips that step on
    the device and the scale backs off — repeatedly.  The loss of the fixed batch goes down, every parameter stays finite, Adam's step
    count = launches - skips, and no step reads found_inf back on the host (the loop issues all steps before the one synchronisation)."""
    from gecco_amd import autograd as ag
    from gecco_amd.optim import FusedAdamEMA
    from gecco_amd.structs import Example
    from tests.test_hip_training import _small_training_setup
    ag.WEIGHT_IMAGES.__init__()
    m, batch = _small_training_setup(13)
    opt = FusedAdamEMA(m.parameters(), lr=2e-4, ema_decay=0.99, amp_on_device=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 20, growth_interval=8)
    losses, scales = [], []
    for it in range(120):
        torch.manual_seed(100)     # the same sigma / noise draws each step
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = m.training_step(batch, it)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.detach())
        scales.append(scaler._scale.clone())
    torch.cuda.synchronize()
    losses = [float(v) for v in losses]
    scales = [float(v) for v in scales]
    skipped = 120 - opt.adam_steps_taken
    print(f"16-mixed run: loss {losses[0]:.3f} -> {losses[-1]:.3f}, scale 2^20 -> 2^{int(np.log2(scales[-1]))} (max 2^{int(np.log2(max(scales)))}), "
          f"{skipped} skipped steps")
    assert all(np.isfinite(losses)) and losses[-1] < 0.9 * losses[0]
    assert skipped >= 2 and skipped < 60                       # the scale really hit the ceiling, and training went on
    assert min(scales) < max(scales)
    assert all(bool(torch.isfinite(p).all()) for p in m.parameters())
    assert all(bool(torch.isfinite(e).all()) for e in opt.ema_params)
    ag.WEIGHT_IMAGES.__init__()


@pytest.mark.parametrize("N", [1000, 333, 96, 2000])
def test_ragged_clouds_under_autocast(N):
    """Cloud sizes that leave the fast shapes (not a multiple of 128 / 32; 96 points = two 64-row tiles per sample): the 16-mixed step
    falls back form by form (A-stationary -> LDS-DMA fp16 -> exact kernels) and stays at fp16-operand distance from the plain step,
    every gradient finite, the same bits run to run."""
    from gecco_amd import autograd as ag
    from gecco_amd.structs import Example
    from tests.test_modules_cpu import build_uncond, uncond_state_dict

    def run(amp):
        ag.WEIGHT_IMAGES.__init__()
        torch.manual_seed(0)
        m = build_uncond(128, 2)
        m.load_state_dict(uncond_state_dict(W.linear_lift_state_dict(9, 128, 2, cases.I, cases.H)))
        m = m.cuda().train()
        x = torch.from_numpy(np.random.RandomState(4).randn(3, N, 3).astype(np.float32))
        data = (x * torch.tensor(cases.GAUSS_SIGMA) + torch.tensor(cases.GAUSS_MEAN)).cuda()
        torch.manual_seed(5)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            loss = m.training_step(Example(data, None), 0)
        (loss * 256.0).backward()
        torch.cuda.synchronize()
        return float(loss), torch.cat([p.grad.flatten() for p in m.parameters()]) / 256.0
    lp, gp = run(False)
    la, ga = run(True)
    lb, gb = run(True)
    assert bool(torch.isfinite(ga).all()) and abs(la - lp) / abs(lp) < 5e-4
    assert float((ga - gp).norm() / gp.norm()) < 3e-3
    assert la == lb and torch.equal(ga, gb)
    ag.WEIGHT_IMAGES.__init__()
