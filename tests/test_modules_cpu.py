"""CPU-side checks of the drop-in module API: constructor signatures, state-dict contract (the seeded state dicts
of oracle/weights.py were loaded strict=True into the real reference by tools/make_golden.py, so loading them
strict=True here pins key names and shapes), the sampler's host-side schedule, refusal to run without HIP."""
import math

import pytest
import torch

from oracle import cases, cpu_ref
from oracle import weights as W


def build_uncond(d, L, sigma_max=165.0, num_inducers=None):
    from gecco_amd.diffusion import Diffusion, EDMLoss, EDMPrecond, IdleConditioner, LogUniformSchedule
    from gecco_amd.models.activation import GaussianActivation
    from gecco_amd.models.linear_lift import LinearLift
    from gecco_amd.models.set_transformer import SetTransformer
    from gecco_amd.reparam import GaussianReparam
    net = LinearLift(inner=SetTransformer(n_layers=L, num_inducers=num_inducers or cases.I, feature_dim=d, t_embed_dim=1,
                                          num_heads=cases.H, activation=GaussianActivation), feature_dim=d)
    return Diffusion(backbone=EDMPrecond(model=net), conditioner=IdleConditioner(),
                     reparam=GaussianReparam(torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA)),
                     loss=EDMLoss(schedule=LogUniformSchedule(max=sigma_max)))


def build_cond(d, L, context_dims=(96, 192, 384), conditioner=None):
    from gecco_amd.diffusion import Conditioner, Diffusion, EDMLoss, EDMPrecond, LogUniformSchedule
    from gecco_amd.models.activation import GaussianActivation
    from gecco_amd.models.ray import RayNetwork
    from gecco_amd.models.set_transformer import SetTransformer
    from gecco_amd.reparam import UVLReparam
    rp = UVLReparam(torch.tensor([0.0, 0.0, 1.38]), torch.tensor([0.56, 0.60, 0.49]))
    net = RayNetwork(backbone=SetTransformer(n_layers=L, num_inducers=cases.I, feature_dim=d, t_embed_dim=1,
                                             num_heads=cases.H, activation=GaussianActivation),
                     reparam=rp, context_dims=context_dims)
    return Diffusion(backbone=EDMPrecond(model=net), conditioner=conditioner or Conditioner(), reparam=rp,
                     loss=EDMLoss(schedule=LogUniformSchedule(max=180.0)))


def uncond_state_dict(p):
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.mean"] = torch.tensor(cases.GAUSS_MEAN)
    sd["reparam.sigma"] = torch.tensor(cases.GAUSS_SIGMA)
    return sd


def test_state_dict_contract_unconditional():
    d, L = 384, 6
    m = build_uncond(d, L)
    sd = uncond_state_dict(W.linear_lift_state_dict(1, d, L, cases.I, cases.H))
    assert len(sd) == 204
    m.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in m.parameters()) == 13_481_103
    assert sorted(m.state_dict()) == sorted(sd)


def test_state_dict_contract_conditional():
    d, L = 128, 2
    m = build_cond(d, L)
    p = W.ray_network_state_dict(1, d, L, cases.I, cases.H)
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.uvl_mean"], sd["reparam.uvl_std"] = p["reparam.uvl_mean"], p["reparam.uvl_std"]
    m.load_state_dict(sd, strict=True)
    assert sorted(m.state_dict()) == sorted(sd)


def test_init_matches_reference_conventions():
    from gecco_amd.models.activation import GaussianActivation
    from gecco_amd.models.set_transformer import BroadcastingLayer
    torch.manual_seed(0)
    layer = BroadcastingLayer(64, 64, 1, num_heads=8, activation=GaussianActivation)
    n = layer.broadcast_norm
    assert n.scale.weight.abs().sum() == 0 and n.bias.weight.abs().sum() == 0 and n.bias.bias.abs().sum() == 0
    assert torch.all(n.scale.bias == 1)
    assert layer.mlp[1].alpha.item() == 1.0
    assert layer.broadcast.pool.inducers.shape == (1, 8, 64, 8)
    # residual branches are scaled by 0.1 at init: out_proj std << default in_proj std
    assert layer.broadcast.unpool.out_proj.weight.std() < 0.3 * layer.broadcast.unpool.in_proj_weight.std()


def test_schedule_table_matches_oracle():
    from gecco_amd.diffusion import build_schedule_table, karras_t_steps
    ts = karras_t_steps(128, 165.0, 0.002, 7)
    assert torch.equal(ts, cpu_ref.t_steps(128, 165.0, 0.002, 7))
    tab = build_schedule_table(ts, 128, 0.5, 0.0, float("inf"), 1.0)
    for i in (0, 5, 127):
        g = cpu_ref.churn_gamma(float(ts[i]), 128, 0.5, 0.0, float("inf"))
        t_hat = ts[i] + g * ts[i]
        assert tab[i, 0] == ts[i] and tab[i, 1] == t_hat and tab[i, 2] == ts[i + 1]
        assert tab[i, 3] == (t_hat ** 2 - ts[i] ** 2).sqrt()
    tab2 = build_schedule_table(ts, 128, 0.5, 1.0, 10.0, 1.0)  # churn only inside [S_min, S_max]
    assert tab2[0, 3] == 0 and tab2[127, 3] == 0 and (tab2[:, 3] > 0).any()


def test_modules_refuse_cpu_and_autograd():
    from gecco_amd import _lib
    m = build_uncond(64, 1)
    x, s = torch.randn(2, 64, 3), torch.tensor([1.0, 2.0])
    with torch.no_grad(), pytest.raises(_lib.GeccoHipError):
        m(x, s, None)
    with pytest.raises(_lib.GeccoHipError):
        m(x, s, None)  # grad enabled: the HIP training path — still no CPU fallback
    mc = build_cond(64, 1, (8, 16, 24))
    from gecco_amd.models.feature_pyramid import FeaturePyramidContext
    from gecco_amd.structs import Context3d
    ctx = Context3d(image=torch.zeros(2, 3, 8, 8), K=torch.eye(3).repeat(2, 1, 1))
    pyr = FeaturePyramidContext(features=[torch.zeros(2, c, 4, 4) for c in (8, 16, 24)], K=ctx.K)
    with pytest.raises(_lib.GeccoHipError):
        mc(x, s, ctx, pyr)  # grad enabled: RayNetwork's HIP training path (lookup backward included) — no CPU fallback
    with pytest.raises(ValueError):
        m.upsample(x, new_latents=None, n_new=None)


def test_structs_and_config(tmp_path):
    from gecco_amd import load_config
    from gecco_amd.structs import Context3d, Example
    ctx = Context3d(image=torch.zeros(2, 3, 8, 8), K=torch.eye(3).repeat(2, 1, 1))
    ex = Example(data=torch.zeros(2, 5, 3), ctx=ctx)
    ex64 = ex.apply_to_tensors(lambda t: t.double())
    assert ex64.data.dtype == torch.float64 and ex64.ctx.K.dtype == torch.float64 and ex.data.dtype == torch.float32
    assert "Context3d" in repr(ex)
    cfg = tmp_path / "cfg.py"
    cfg.write_text("model = 41 + 1\n")
    assert load_config(str(cfg)).model == 42
    with pytest.raises(ValueError):
        load_config(str(tmp_path / "cfg.txt"))


def test_log_uniform_schedule_is_stratified():
    from gecco_amd.diffusion import LogUniformSchedule
    s = LogUniformSchedule(max=165.0)(torch.zeros(64, 10, 3)).reshape(-1)
    assert s.shape == (64,) and torch.all(s[1:] > s[:-1]) and s.min() >= 0.002 and s.max() <= 165.0
    lo = math.log(0.002) + torch.arange(64) / 64 * (math.log(165.0) - math.log(0.002))
    assert torch.all(s.log() >= lo - 1e-5)


def test_frozen_weights_scope_bookkeeping():
    """hip_ops.frozen_weights(): the token a workspace's weight images are built under (host logic only).  Outside a scope there is no
    token (every forward rebuilds); inside, full and cached evaluations have different tokens; nesting keeps the generation; a new
    outermost scope, `weights_changed()` (the optimizer's step calls it) and `set_option` start a new one."""
    from gecco_amd import hip_ops as ops
    st = ops._ImageState()
    assert st.token(False) is None
    with ops.frozen_weights():
        t_full, t_cached = st.token(False), st.token(True)
        assert t_full is not None and t_full != t_cached and t_full == st.token(False)
        with ops.frozen_weights():
            assert st.token(False) == t_full
        assert st.token(False) == t_full
        ops.weights_changed()
        t2 = st.token(False)
        assert t2 != t_full
    assert st.token(False) is None
    with ops.frozen_weights():
        assert st.token(False) not in (t_full, t2)
    import gecco_amd
    assert gecco_amd.frozen_weights is ops.frozen_weights and gecco_amd.weights_changed is ops.weights_changed


def test_frozen_scope_and_image_tokens_are_per_plan_and_per_thread():
    """The frozen scope's depth is per host THREAD, the built-image record per PLAN (`_ImageState`): a scope held by one thread is not a
    scope for another; a plan frozen by name leaves the others rebuilding; one plan's `changed()` does not touch another's tokens; a
    forward that failed, or was only captured into a graph, records nothing."""
    import threading
    from gecco_amd import hip_ops as ops
    a, b = ops._ImageState(), ops._ImageState()
    seen = {}
    with ops.frozen_weights():
        t = threading.Thread(target=lambda: seen.update(other=a.token(False)))
        t.start(); t.join()
        assert a.token(False) is not None and seen["other"] is None           # the other thread holds no scope
    with ops.frozen_weights(a):                                                # only plan a
        ta = a.token(False)
        assert ta is not None and b.token(False) is None
        key = (2, 128, 0)
        assert a.ready(key, ta) == 0
        a.built(key, ta, 0)
        assert a.ready(key, ta) == 1 and b.ready(key, ta) == 0                 # b has built nothing
        b.changed()
        assert a.ready(key, ta) == 1                                           # b's change is b's
        a.failed(key)
        assert a.ready(key, ta) == 0                                           # a failed forward leaves no "ready" behind
        a.built(key, ta, 0)
        a.changed()
        assert a.ready(key, a.token(False)) == 0
    assert a.token(False) is None


def test_frozen_weights_accepts_plans_states_and_modules():
    from gecco_amd import hip_ops as ops
    from gecco_amd.models.set_transformer import SetTransformer
    st = ops._ImageState()
    m = SetTransformer(n_layers=1, feature_dim=64, num_inducers=64, t_embed_dim=1, num_heads=8)
    assert ops._image_states_of(st) == [st] and ops._image_states_of(m) == []     # (no plan built yet: nothing to freeze)

    class FakePlan:
        images = st
    m._cache.plan = FakePlan()
    assert ops._image_states_of(m) == [st] and ops._image_states_of(FakePlan()) == [st]
    with ops.frozen_weights(m):
        assert st.depth == 1 and st.token(False) is not None
    assert st.depth == 0
    with pytest.raises(TypeError):
        with ops.frozen_weights(3):
            pass


def test_two_plans_with_different_modes_and_options_do_not_alias(monkeypatch):
    """Two plans in one process: each carries its own precision and its own pinned path switches in its table (GeccoSetTransformer.opt_mask
    / opt_vals, ABI 14); pinning on one leaves the other — and the process-wide defaults — untouched.  (Plans hold raw pointers only:
    built here on CPU tensors with the device check lifted; nothing is launched.)"""
    import ctypes as C
    from gecco_amd import _lib, hip_ops as ops
    from oracle import cases
    lib = _lib.load()
    monkeypatch.setattr(ops, "_ptr", lambda t: C.c_void_p(0 if t is None else t.data_ptr()))
    p, _, _ = cases.uncond_inputs("uncond_d128_L4_N256")
    w2 = ops.LinearLiftPlan(p, cases.H, cases.I, precision="w2", options={"chain2": 0})
    mx = ops.LinearLiftPlan(p, cases.H, cases.I, precision="mixed")
    bit = lambda n: 1 << lib.gecco_option_index(n.encode())
    assert lib.gecco_option_index(b"no-such-option") == -1 and bit("astat") == 1
    assert w2.table.inner.precision == ops.PRECISIONS["w2"] and mx.table.inner.precision == ops.PRECISIONS["mixed"]
    assert w2.table.inner.opt_mask == bit("chain2") and w2.table.inner.opt_vals == 0 and mx.table.inner.opt_mask == 0
    mx.set_option("mlpw", 1)
    mx.set_option("kvfold", 0)
    assert mx.table.inner.opt_mask == bit("mlpw") | bit("kvfold") and mx.table.inner.opt_vals == bit("mlpw")
    assert mx.st.table.opt_mask == mx.table.inner.opt_mask
    assert w2.table.inner.opt_mask == bit("chain2")                            # untouched
    mx.set_option("mlpw", -1)
    assert mx.table.inner.opt_mask == bit("kvfold") and mx.table.inner.opt_vals == 0
    assert w2.images is not mx.images
    # a call's table is a COPY: what it pins (images_ready, "mlpwshare" of a two-stream evaluation) never reaches the plan's own
    tbl = _lib.GeccoLinearLift.from_buffer_copy(w2.table)
    tbl.inner.images_ready = 1
    ops._pin_option(tbl.inner, "mlpwshare", 1)
    assert w2.table.inner.images_ready == 0 and w2.table.inner.opt_mask == bit("chain2")
    assert tbl.inner.layers and C.addressof(tbl.inner.layers.contents) == C.addressof(w2.table.inner.layers.contents)
    with pytest.raises(ValueError):
        w2.set_option("no-such-option", 1)


def test_modules_carry_their_own_precision_and_options():
    """`Diffusion.set_precision` / `set_option`: the model's SetTransformer holds them, the plan cache's signature includes them (a change
    rebuilds the plan), and a second model is untouched."""
    from gecco_amd import hip_ops as ops
    from gecco_amd.models.set_transformer import SetTransformer, _own_settings, _param_sig
    a = SetTransformer(n_layers=1, feature_dim=64, num_inducers=64, t_embed_dim=1, num_heads=8)
    b = SetTransformer(n_layers=1, feature_dim=64, num_inducers=64, t_embed_dim=1, num_heads=8)
    assert _own_settings(a) == (None, ())
    sig0 = _param_sig(a)
    a.set_precision("w2").set_option("chain2", 0)
    assert _own_settings(a) == ("w2", (("chain2", 0),)) and _own_settings(b) == (None, ())
    assert _param_sig(a) != sig0
    a.set_option("chain2", -1)
    assert _own_settings(a) == ("w2", ())
    with pytest.raises(ValueError):
        a.set_precision("fp4")
    with pytest.raises(ValueError):
        a.set_option("no-such-option", 1)
    assert ops.default_precision() in ops.PRECISIONS


def test_reparam_log_determinants_vs_autograd_jacobian():
    """`Reparam.ladj_data_to_diffusion` (what `Diffusion.evaluate_logp` adds for the change of variables; gecco-jax models/reparam.py:27-37
    obtains it from `jax.jacrev` + `slogdet` per point): the closed forms against torch's Jacobian of the oracle's restatement of the
    same map, point by point — UVL (pinhole projection, atanh, log-range, normalisation) and Gaussian."""
    from gecco_amd.reparam import GaussianReparam, NoReparam, UVLReparam
    from gecco_amd.structs import Context3d
    rs = torch.Generator().manual_seed(3)
    B, N = 2, 17
    xyz = torch.rand(B, N, 3, generator=rs, dtype=torch.float64) * torch.tensor([0.8, 0.6, 2.0]) + torch.tensor([-0.4, -0.3, 1.0])
    K = torch.tensor([[[0.9, 0.0, 0.5], [0.0, 1.1, 0.5], [0.0, 0.0, 1.0]], [[0.7, 0.0, 0.45], [0.0, 0.8, 0.55], [0.0, 0.0, 1.0]]], dtype=torch.float64)
    mean, std = torch.tensor([0.0, 0.0, 1.38], dtype=torch.float64), torch.tensor([0.56, 0.60, 0.49], dtype=torch.float64)
    rp = UVLReparam(mean.float(), std.float())
    got = rp.ladj_data_to_diffusion(xyz.float(), Context3d(image=None, K=K.float()))
    ref = torch.zeros(B, dtype=torch.float64)
    for b in range(B):
        for n in range(N):
            f = lambda p: cpu_ref.uvl_data_to_diffusion(p[None, None], K[b:b + 1], mean, std)[0, 0]
            J = torch.autograd.functional.jacobian(f, xyz[b, n])
            ref[b] += torch.linalg.slogdet(J)[1]
    assert got.dtype == torch.float64 and torch.allclose(got, ref, rtol=1e-5, atol=1e-4), (got, ref)
    g = GaussianReparam(torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA))
    assert torch.allclose(g.ladj_data_to_diffusion(xyz.float(), None),
                          torch.full((B,), -N * float(torch.log(torch.tensor(cases.GAUSS_SIGMA).double()).sum()), dtype=torch.float64))
    assert torch.equal(NoReparam(3).ladj_data_to_diffusion(xyz.float(), None), torch.zeros(B, dtype=torch.float64))
