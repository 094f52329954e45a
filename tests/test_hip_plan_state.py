"""Precision, path switches and the frozen-weights record are PER PLAN (and the scope per host thread): the reference's modules
carry no process-wide state (models/set_transformer.py:176-216, diffusion.py:160-178) — two models of different arithmetic in one
serving process must not see each other.  Plus the hand-overs that used to leave `images_ready = 1` behind images that were never
(re)built: an EMA swap inside a scope, a first forward that was only captured into a graph."""
import threading

import pytest
import torch

from oracle import cases
from oracle import weights as W
from tests.test_modules_cpu import build_uncond, uncond_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import hip_ops
    return hip_ops


def _cuda(p):
    return {k: v.cuda() for k, v in p.items()}


def test_two_plans_interleaved_inside_one_frozen_scope(ops):
    """Two plans of different modes and options, evaluated alternately inside ONE `frozen_weights()` scope (full and cached
    evaluations), then concurrently from two host threads on two streams: every result is bit-identical to the plan run alone."""
    name = "uncond_d384_L6_N128"
    p, x, sigma = cases.uncond_inputs(name)
    p = _cuda(p)
    x, sigma = x.cuda(), sigma.cuda()
    xn = x[:, :64].contiguous()
    a = ops.LinearLiftPlan(p, cases.H, cases.I, precision="w2", options={"chain2": 1, "kvfold": 1})   # pinned to the defaults
    b = ops.LinearLiftPlan(p, cases.H, cases.I, precision="mixed", options={"chain2": 0, "kvfold": 0})
    alone = {}
    for k, net in (("a", a), ("b", b)):
        (d, r), cache = net.forward(x, sigma, return_raw=True, do_cache=True)
        alone[k] = (d, r, net.forward(xn, sigma, cache=cache), cache)
    assert not torch.equal(alone["a"][1], alone["b"][1])          # they ARE different arithmetics
    # b's pinned switches are b's: an unpinned mixed plan under the process-wide switches gives the same bits, a default one does not
    c = ops.LinearLiftPlan(p, cases.H, cases.I, precision="mixed")
    dflt = c.forward(x, sigma, return_raw=True)[1]
    try:
        ops.set_option("chain2", 0)
        ops.set_option("kvfold", 0)
        assert torch.equal(c.forward(x, sigma, return_raw=True)[1], alone["b"][1])
        assert torch.equal(a.forward(x, sigma, return_raw=True)[1], alone["a"][1])   # a pinned its own: the process-wide change is not a's
    finally:
        ops.set_option("chain2", -1)
        ops.set_option("kvfold", -1)
    assert not torch.equal(dflt, alone["b"][1])
    with ops.frozen_weights():
        for _ in range(3):
            for k, net in (("a", a), ("b", b)):
                d, r = net.forward(x, sigma, return_raw=True)
                assert torch.equal(d, alone[k][0]) and torch.equal(r, alone[k][1]), k
                assert torch.equal(net.forward(xn, sigma, cache=alone[k][3]), alone[k][2]), k
        for net in (a, b):
            assert net.images.tokens and all(t == net.images.token(t[-1]) for t in net.images.tokens.values())
        assert a.images is not b.images
    # two host threads, each with its own plan, stream and scope
    errs = []

    def worker(k, net):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s), ops.frozen_weights():
                xs, ss = x.clone(), sigma.clone()
                for _ in range(8):
                    d, r = net.forward(xs, ss, return_raw=True)
                    if not (torch.equal(d, alone[k][0]) and torch.equal(r, alone[k][1])):
                        errs.append(k)
            s.synchronize()
        except Exception as e:   # noqa: BLE001
            errs.append((k, repr(e)))
    torch.cuda.synchronize()
    ts = [threading.Thread(target=worker, args=(k, net)) for k, net in (("a", a), ("b", b))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_scope_by_plan_freezes_only_that_plan(ops):
    name = "uncond_d384_L6_N128"
    p, x, sigma = cases.uncond_inputs(name)
    p = _cuda(p)
    x, sigma = x.cuda(), sigma.cuda()
    a = ops.LinearLiftPlan(p, cases.H, cases.I, precision="mixed")
    b = ops.LinearLiftPlan(p, cases.H, cases.I, precision="mixed")
    d0 = a.forward(x, sigma)
    w = p["inner.layers.2.mlp.2.weight"]
    saved = w.clone()
    try:
        with ops.frozen_weights(a):
            assert torch.equal(a.forward(x, sigma), d0)
            w.mul_(1.5)
            assert torch.equal(a.forward(x, sigma), d0)          # a is frozen: its images are the scope's
            fresh = b.forward(x, sigma)                          # b is not: it rebuilds and sees the new weight
            assert not torch.equal(fresh, d0)
            a.images.changed()
            assert torch.equal(a.forward(x, sigma), fresh)
    finally:
        w.copy_(saved)
    assert torch.equal(a.forward(x, sigma), d0)


def test_first_forward_of_a_scope_only_captured_leaves_nothing_ready(ops):
    """a first forward of a scope that is only CAPTURED (no eager warm-up) executes no image build; the eager call that
    follows must rebuild instead of trusting `images_ready`, and the graph — which carries its own builds — replays correctly."""
    name = "uncond_d384_L6_N128"
    p, x, sigma = cases.uncond_inputs(name)
    p = _cuda(p)
    x, sigma = x.cuda(), sigma.cuda()
    net = ops.LinearLiftPlan(p, cases.H, cases.I, precision="w2")
    d0 = net.forward(x, sigma).clone()
    w = p["inner.layers.1.mlp.0.weight"]
    saved = w.clone()
    try:
        w.mul_(1.25)                                             # the workspace's images are now stale
        out = torch.empty_like(x)
        torch.cuda.synchronize()
        with ops.frozen_weights():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                net.forward(x, sigma, out=out)                   # captured, never executed
            assert not net.images.tokens                         # nothing recorded as built
            fresh = net.forward(x, sigma).clone()                # eager: rebuilds
            assert not torch.equal(fresh, d0)
            assert net.images.tokens
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, fresh)
        ref = ops.LinearLiftPlan(p, cases.H, cases.I, precision="w2").forward(x, sigma)
        assert torch.equal(ref, fresh)
    finally:
        w.copy_(saved)


def test_ema_swap_inside_a_frozen_scope_rebuilds_the_images():
    """`FusedAdamEMA.switch_main_parameter_weights` rewrites every parameter through the flat buffer; a plan inside
    `frozen_weights()` must not keep streaming the pre-swap weight images beside the post-swap biases."""
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import hip_ops
    from gecco_amd.optim import FusedAdamEMA
    d, L = 128, 2
    m = build_uncond(d, L)
    m.load_state_dict(uncond_state_dict(W.linear_lift_state_dict(5, d, L, cases.I, cases.H)), strict=True)
    m = m.cuda().set_precision("mixed")
    opt = FusedAdamEMA(m.parameters(), lr=1e-2, ema_decay=0.5)
    params = list(m.parameters())
    m.zero_grad(set_to_none=True)
    for q, gr in zip(params, cases.optim_grads(0, [tuple(q.shape) for q in params])):
        q.grad = gr.cuda()
    opt.step()                                                   # the EMA shadows now differ from the weights
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 256, 3, generator=g).cuda()
    sigma = torch.tensor([0.5, 2.0]).cuda()
    with torch.no_grad():
        raw_out = m(x, sigma, None).clone()
        with opt.swap_ema_weights():
            ema_out = m(x, sigma, None).clone()                  # outside a scope: every forward rebuilds
        assert not torch.equal(raw_out, ema_out)
        with hip_ops.frozen_weights():
            assert torch.equal(m(x, sigma, None), raw_out)
            with opt.swap_ema_weights():
                assert torch.equal(m(x, sigma, None), ema_out)   # the swap started a new generation
            assert torch.equal(m(x, sigma, None), raw_out)       # and so did the swap back
        # a graph captured with frozen weights serves the values it was captured with; capturing again after a swap serves the new ones
        with opt.swap_ema_weights():
            run = m.graphed_forward(x, sigma, None, frozen_weights=True)
            assert torch.equal(run(), ema_out)


def test_model_level_precision_is_the_models_own(ops):
    """`Diffusion.set_precision`: two models in one process, different arithmetic, interleaved; the process default untouched."""
    d, L = 128, 2
    sd = uncond_state_dict(W.linear_lift_state_dict(5, d, L, cases.I, cases.H))
    ma, mb = build_uncond(d, L), build_uncond(d, L)
    for m in (ma, mb):
        m.load_state_dict(sd, strict=True)
    ma, mb = ma.cuda().set_precision("fp32"), mb.cuda().set_precision("mixed")
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 256, 3, generator=g).cuda()
    sigma = torch.tensor([0.3, 5.0]).cuda()
    before = ops.default_precision()
    with torch.no_grad():
        ra = ops.LinearLiftPlan(dict(ma.backbone.model.named_parameters()), cases.H, cases.I, precision="fp32").forward(x, sigma)
        rb = ops.LinearLiftPlan(dict(mb.backbone.model.named_parameters()), cases.H, cases.I, precision="mixed").forward(x, sigma)
        assert not torch.equal(ra, rb)
        for _ in range(2):
            assert torch.equal(ma(x, sigma, None), ra)
            assert torch.equal(mb(x, sigma, None), rb)
        mb.set_precision("fp32")                                 # a change rebuilds that model's plan
        assert torch.equal(mb(x, sigma, None), ra)
    assert ops.default_precision() == before
