"""The channels-last ConvNeXt conditioner on the HIP path against the oracle's restatement of
torchvision's ConvNeXt stages (oracle/cpu_ref.py::convnext_features; torchvision is absent: parity unpinned against
torchvision itself).  Weights are seeded and deliberately non-degenerate (torchvision's layer_scale init of 1e-6 would hide
the blocks)."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref


def _seeded_state(model, seed):
    rs = np.random.RandomState(seed)
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith("layer_scale"):
            a = rs.uniform(0.2, 1.0, size=v.shape)
        elif k.endswith("weight") and v.dim() == 1:     # LayerNorm weights
            a = 1.0 + 0.2 * rs.randn(*v.shape)
        elif k.endswith("bias"):
            a = 0.1 * rs.randn(*v.shape)
        else:
            fan_in = int(np.prod(v.shape[1:]))
            a = rs.randn(*v.shape) / np.sqrt(fan_in)
        sd[k] = torch.from_numpy(np.asarray(a, dtype=np.float32))
    return sd


def test_state_dict_keys_match_torchvision_layout():
    """Key names / shapes of torchvision's convnext_tiny features[0:6] re-indexed as the reference's `stages` (CPU)."""
    from gecco_amd.models.feature_pyramid import ConvNeXtExtractor
    m = ConvNeXtExtractor(n_stages=3, model="tiny", pretrained=False)
    sd = m.state_dict()
    assert sd["stages.0.0.0.weight"].shape == (96, 3, 4, 4) and sd["stages.0.0.1.weight"].shape == (96,)
    assert sd["stages.1.0.0.weight"].shape == (96,) and sd["stages.1.0.1.weight"].shape == (192, 96, 2, 2)
    assert sd["stages.2.0.1.weight"].shape == (384, 192, 2, 2)
    assert sd["stages.0.1.2.block.0.weight"].shape == (96, 1, 7, 7) and sd["stages.0.1.2.block.3.weight"].shape == (384, 96)
    assert sd["stages.2.1.8.block.5.weight"].shape == (384, 1536) and sd["stages.2.1.8.layer_scale"].shape == (384, 1, 1)
    assert "stages.2.1.9.layer_scale" not in sd and "stages.3.0.0.weight" not in sd
    n = sum(v.numel() for v in sd.values())
    assert n == 12_348_000, n   # stem (4 896) + 3 x 79 296 + 74 112 + 3 x 306 048 + 295 680 + 9 x 1 201 920


@pytest.mark.gpu
@pytest.mark.parametrize("hw,B", [(224, 2), (256, 1), (64, 3)])
def test_convnext_pyramid_vs_oracle(hw, B):
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import hip_ops
    from gecco_amd.models.feature_pyramid import ConvNeXtExtractor
    from gecco_amd.structs import Context3d
    m = ConvNeXtExtractor(n_stages=3, model="tiny", pretrained=False)
    sd = _seeded_state(m, 5)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    img = torch.from_numpy(np.random.RandomState(6).rand(B, 3, hw, hw).astype(np.float32))
    with torch.no_grad():
        ref = cpu_ref.convnext_features(img, sd)
    K = torch.eye(3).repeat(B, 1, 1)
    old = hip_ops.default_precision()
    try:
        for precision, tol in (("fp32", 2e-5), ("bf16x3", 2e-4)):
            hip_ops.set_default_precision(precision)
            out = m(Context3d(image=img.cuda(), K=K.cuda()))
            assert len(out.features) == 3
            for lvl, (f, r) in enumerate(zip(out.features, ref)):
                assert f.shape == r.shape                                           # NCHW shape ...
                assert f.is_contiguous(memory_format=torch.channels_last) or f.shape[1] == 1   # ... channels-last memory
                e = cpu_ref.rel_err(f.cpu(), r)
                print(f"convnext {hw}x{hw} B={B} [{precision}] level {lvl} {tuple(r.shape)}: {e}")
                assert e[0] < tol, (precision, lvl, e)
            levels = hip_ops.to_channels_last_levels(out.features)
            assert all(l.data_ptr() == f.data_ptr() for l, f in zip(levels, out.features))   # consumed without a copy
    finally:
        hip_ops.set_default_precision(old)


@pytest.mark.gpu
def test_conditional_diffusion_with_the_device_conditioner():
    """The whole image-conditional evaluation as the reference's configs build it (taskonomy_conditional.py): image ->
    ConvNeXtExtractor -> projective lookup -> RayNetwork, against the oracle chain convnext_features -> cond_denoiser."""
    import __graft_entry__ as ge
    ge.build()
    from oracle import cases
    from oracle import weights as Wt
    from gecco_amd.models.feature_pyramid import ConvNeXtExtractor
    from gecco_amd.structs import Context3d
    from tests.test_modules_cpu import build_cond
    d, L, N, hw, B = 128, 2, 96, 64, 2
    cn = ConvNeXtExtractor(n_stages=3, model="tiny", pretrained=False)
    csd = _seeded_state(cn, 9)
    cn.load_state_dict(csd, strict=True)
    m = build_cond(d, L, conditioner=cn)
    p = Wt.ray_network_state_dict(17, d, L, cases.I, cases.H)
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.uvl_mean"], sd["reparam.uvl_std"] = p["reparam.uvl_mean"], p["reparam.uvl_std"]
    sd.update({"conditioner." + k: v for k, v in csd.items()})
    m.load_state_dict(sd, strict=True)     # the released checkpoints' layout: conditioner.stages.* beside backbone.*
    m = m.cuda().eval()
    rs = np.random.RandomState(3)
    img = torch.from_numpy(rs.rand(B, 3, hw, hw).astype(np.float32))
    _, K = Wt.synthetic_context(4, B, hw=hw)
    x = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
    sigma = torch.tensor([0.05, 5.0])
    with torch.no_grad():
        feats = cpu_ref.convnext_features(img, csd)
        ref = cpu_ref.cond_denoiser(p, "", cases.H, K, feats)(x, sigma)
        out = m(x.cuda(), sigma.cuda(), Context3d(image=img.cuda(), K=K.cuda()))
        smp = m.sample_stochastic((B, N, 3), Context3d(image=img.cuda(), K=K.cuda()), num_steps=4)
    e = cpu_ref.rel_err(out.cpu(), ref)
    print("image -> ConvNeXt -> lookup -> RayNetwork vs oracle:", e)
    assert e[0] < 1e-4, e
    assert torch.isfinite(smp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("hw,B", [(64, 2), (96, 3), (224, 11)])
def test_convnext_parameter_gradients_vs_oracle(hw, B):
    """The conditioner's TRAINING path (the reference optimises it with the denoiser: diffusion.py:210-211 over
    self.parameters()): gradients of sum_levels <features, R> with respect to every parameter, HIP autograd Functions
    (autograd.convnext_pyramid) against torch autograd through the oracle's restatement.  224 x 224 x 11 images exercises the
    multi-batch-per-block loops of the reduction kernels and ragged tails."""
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import hip_ops
    from gecco_amd.models.feature_pyramid import ConvNeXtExtractor
    from gecco_amd.structs import Context3d
    m = ConvNeXtExtractor(n_stages=3, model="tiny", pretrained=False)
    sd = _seeded_state(m, 21)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    rs = np.random.RandomState(22)
    img = torch.from_numpy(rs.rand(B, 3, hw, hw).astype(np.float32))
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    img_c = img.clone().requires_grad_(True)
    ref = cpu_ref.convnext_features(img_c, p)
    R = [torch.from_numpy(rs.randn(*f.shape).astype(np.float32)) for f in ref]
    sum((f * r).sum() for f, r in zip(ref, R)).backward()
    K = torch.eye(3).repeat(B, 1, 1)
    old = hip_ops.default_precision()
    try:
        for precision, tol in (("fp32", 1e-4), ("bf16x3", 1e-3)):
            hip_ops.set_default_precision(precision)
            m.zero_grad(set_to_none=True)
            img_g = img.clone().cuda().requires_grad_(True)   # ... and with respect to the image itself (the stem's dX: CnxStemFn)
            out = m(Context3d(image=img_g, K=K.cuda()))
            for f, r in zip(out.features, ref):
                assert f.shape == r.shape and f.requires_grad
                assert cpu_ref.rel_err(f.detach().cpu(), r.detach())[0] < (2e-5 if precision == "fp32" else 2e-4)
            sum((f * r.cuda()).sum() for f, r in zip(out.features, R)).backward()
            worst = ("", 0.0)
            for k, prm in m.named_parameters():
                assert prm.grad is not None, k
                e = cpu_ref.rel_err(prm.grad.cpu(), p[k].grad)[0]
                if e > worst[1]:
                    worst = (k, e)
                assert e < tol, (precision, k, e)
            ei = cpu_ref.rel_err(img_g.grad.cpu(), img_c.grad)[0]
            assert ei < tol, (precision, "image", ei)
            print(f"convnext gradients {hw}x{hw} B={B} [{precision}]: worst {worst[0]} {worst[1]:.2e}, image {ei:.2e}")
    finally:
        hip_ops.set_default_precision(old)


@pytest.mark.gpu
def test_conditional_training_step_reaches_the_conditioner():
    """training_step of the image-conditional model with a trainable conditioner: the loss and the conditioner's parameter
    gradients against torch autograd through the oracle chain convnext_features -> cond_denoiser -> EDM loss."""
    import __graft_entry__ as ge
    ge.build()
    from oracle import cases
    from oracle import weights as Wt
    from gecco_amd import hip_ops
    from gecco_amd.models.feature_pyramid import ConvNeXtExtractor
    from gecco_amd.structs import Context3d
    from tests.test_modules_cpu import build_cond
    d, L, N, hw, B = 128, 2, 128, 64, 2
    cn = ConvNeXtExtractor(n_stages=3, model="tiny", pretrained=False)
    csd = _seeded_state(cn, 9)
    cn.load_state_dict(csd, strict=True)
    m = build_cond(d, L, conditioner=cn)
    p = Wt.ray_network_state_dict(17, d, L, cases.I, cases.H)
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.uvl_mean"], sd["reparam.uvl_std"] = p["reparam.uvl_mean"], p["reparam.uvl_std"]
    sd.update({"conditioner." + k: v for k, v in csd.items()})
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    rs = np.random.RandomState(3)
    img = torch.from_numpy(rs.rand(B, 3, hw, hw).astype(np.float32))
    _, K = Wt.synthetic_context(4, B, hw=hw)
    data = torch.from_numpy((0.5 * rs.randn(B, N, 3)).astype(np.float32))
    noise = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
    sigma = torch.tensor([0.3, 2.0])
    cp = {k: v.clone().requires_grad_(True) for k, v in csd.items()}
    feats = cpu_ref.convnext_features(img, cp)
    D = cpu_ref.cond_denoiser(p, "", cases.H, K, feats)
    s3 = sigma.reshape(-1, 1, 1)
    ref_loss = (100.0 * (s3 ** 2 + 1.0) / s3 ** 2 * (D(data + noise * s3, sigma) - data) ** 2).mean()
    ref_loss.backward()
    old = hip_ops.default_precision()
    try:
        for precision, tol in (("fp32", 2e-4), ("bf16x3", 2e-3)):
            hip_ops.set_default_precision(precision)
            m.zero_grad(set_to_none=True)
            s3c = s3.cuda()
            den = m(data.cuda() + noise.cuda() * s3c, sigma.cuda(), Context3d(image=img.cuda(), K=K.cuda()))
            loss = (100.0 * (s3c ** 2 + 1.0) / s3c ** 2 * (den - data.cuda()) ** 2).mean()
            loss.backward()
            lv, rv = float(loss.detach()), float(ref_loss.detach())
            assert abs(lv - rv) / abs(rv) < 1e-4, (lv, rv)
            worst = ("", 0.0)
            for k, prm in m.conditioner.named_parameters():
                e = cpu_ref.rel_err(prm.grad.cpu(), cp[k].grad)[0]
                worst = max(worst, (k, e), key=lambda t: t[1])
                assert e < tol, (precision, k, e)
            print(f"conditional training step [{precision}]: loss {lv:.6f} (oracle {rv:.6f}), worst conditioner gradient {worst[0]} {worst[1]:.2e}")
    finally:
        hip_ops.set_default_precision(old)
