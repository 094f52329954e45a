"""Optimizer and checkpoint rows: the fused Adam + EMA HIP kernel and the Lightning checkpoint
wire format, against golden data written by the REAL reference (tools/make_golden_optim.py: the reference's own
EMAOptimizer(torch.optim.Adam) stepped on injected gradients, its checkpoint assembled as Lightning + EMACallback do).

CPU part: the golden file's structure, and that the CPU restatement (torch.optim.Adam + ema_update restated in
oracle/cases.py) resumes from the reference checkpoint to the reference's own later state — pinning the restatement.
GPU part: FusedAdamEMA (one HIP launch per step) does the same, resumed from the checkpoint and from scratch; a
checkpoint written here has the reference file's structure and loads back; the EMA weights reproduce the reference's
forward."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, cpu_ref
from oracle import weights as W
from tests.test_modules_cpu import build_uncond, uncond_state_dict

CASE = cases.OPTIM_CASE


def _ckpt(golden_dir):
    return torch.load(os.path.join(golden_dir, "ref_ckpt_d64_L1.ckpt"), map_location="cpu", weights_only=False)


def _golden(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "optim.npz")))


def _shape_tree(obj):
    if torch.is_tensor(obj):
        return ("T", tuple(obj.shape), str(obj.dtype))
    if isinstance(obj, dict):
        return {k: _shape_tree(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_shape_tree(v) for v in obj]
    return type(obj).__name__


# ------------------------------------------------------------------------------------------------- CPU
def test_reference_checkpoint_structure(golden_dir):
    ck = _ckpt(golden_dir)
    assert {"epoch", "global_step", "pytorch-lightning_version", "state_dict", "optimizer_states", "lr_schedulers",
            "ema_state_dict"} <= set(ck)
    st = ck["optimizer_states"][0]
    assert sorted(st) == ["current_step", "decay", "ema", "every_n_steps", "opt"]      # ema.py:384-390
    assert sorted(st["opt"]) == ["param_groups", "state"]
    assert st["current_step"] == CASE["steps_before"] and st["decay"] == CASE["decay"]
    m = build_uncond(CASE["d"], CASE["L"])
    assert sorted(ck["state_dict"]) == sorted(m.state_dict()) == sorted(ck["ema_state_dict"])
    m.load_state_dict(ck["ema_state_dict"], strict=True)      # gecco-torch/README.md:35-39
    n = len(list(m.parameters()))
    assert len(st["ema"]) == n == len(st["opt"]["state"])
    # the EMA weights are the optimizer's shadows, in parameter order
    for (k, p), e in zip(m.named_parameters(), st["ema"]):
        assert torch.equal(ck["ema_state_dict"][k], e), k


def test_cpu_restatement_resumes_like_the_reference(golden_dir):
    """torch.optim.Adam + ema_update_ref, resumed from the reference checkpoint, reach the reference's later state."""
    ck, g = _ckpt(golden_dir), _golden(golden_dir)
    m = build_uncond(CASE["d"], CASE["L"])
    m.load_state_dict(ck["state_dict"], strict=True)
    params = list(m.parameters())
    opt = torch.optim.Adam(params, lr=1e-4)
    st = ck["optimizer_states"][0]
    opt.load_state_dict(st["opt"])
    ema = [t.clone() for t in st["ema"]]
    for i in range(CASE["steps_before"], CASE["steps_before"] + CASE["steps_after"]):
        for q, gr in zip(params, cases.optim_grads(i, [tuple(q.shape) for q in params])):
            q.grad = gr
        opt.step()
        ema = cases.ema_update_ref(ema, [q.detach() for q in params], st["decay"])
    for j, q in enumerate(params):
        assert torch.allclose(q.detach(), torch.from_numpy(g[f"p{j}"]), rtol=1e-6, atol=1e-8), j   # host libm / ISA differences
        assert torch.allclose(ema[j], torch.from_numpy(g[f"ema{j}"]), rtol=1e-6, atol=1e-9), j


def test_checkpoint_dict_without_optimizer_has_the_reference_keys(golden_dir):
    from gecco_amd import checkpoint
    ck = _ckpt(golden_dir)
    m = build_uncond(CASE["d"], CASE["L"])
    mine = checkpoint.build_checkpoint(m, None, epoch=0, global_step=3)
    assert set(mine) == set(ck) - {"ema_state_dict"}
    assert _shape_tree(mine["state_dict"]) == _shape_tree(ck["state_dict"])


# ------------------------------------------------------------------------------------------------- GPU
def _model_on_gpu(sd):
    import __graft_entry__ as ge
    ge.build()
    m = build_uncond(CASE["d"], CASE["L"])
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def _inject(params, i):
    for q, gr in zip(params, cases.optim_grads(i, [tuple(q.shape) for q in params])):
        q.grad.copy_(gr.cuda()) if q.grad is not None else setattr(q, "grad", gr.cuda())


def _compare_state(opt, params, g, tag):
    worst = {}
    for j, q in enumerate(params):
        for key, got in (("p", q.detach()), ("ema", opt.ema_params[j]), ("m", opt.view_of("m", j)), ("v", opt.view_of("v", j))):
            ref = torch.from_numpy(g[f"{key}{j}"])
            e = cpu_ref.rel_err(got.cpu().reshape(ref.shape), ref)[0]
            worst[key] = max(worst.get(key, 0.0), e)
    print(tag, {k: f"{v:.2e}" for k, v in worst.items()})
    # fp32 rounding of a handful of operations per element (the kernel follows torch's op order; divisions and the
    # square root may differ in the last place)
    assert worst["p"] <= 1e-6 and worst["ema"] <= 1e-6 and worst["m"] <= 2e-6 and worst["v"] <= 2e-6, worst


@pytest.mark.gpu
def test_fused_adam_ema_resumes_from_the_reference_checkpoint(golden_dir):
    from gecco_amd import checkpoint
    from gecco_amd.optim import FusedAdamEMA
    ck, g = _ckpt(golden_dir), _golden(golden_dir)
    m = _model_on_gpu(uncond_state_dict(W.linear_lift_state_dict(999, CASE["d"], CASE["L"], cases.I, cases.H)))  # other weights
    opt = FusedAdamEMA(m.parameters(), lr=1e-4, ema_decay=0.5)       # other decay: both come from the file
    checkpoint.load_checkpoint(ck, m, opt, weights="raw")
    assert opt.decay == CASE["decay"] and opt.current_step == CASE["steps_before"]
    params = list(m.parameters())
    for i in range(CASE["steps_before"], CASE["steps_before"] + CASE["steps_after"]):
        opt.zero_grad()
        _inject(params, i)
        opt.step()
    torch.cuda.synchronize()
    assert opt.current_step == int(g["current_step"]) and opt._adam_step == int(g["adam_step"])
    _compare_state(opt, params, g, "resumed from the reference checkpoint")


@pytest.mark.gpu
def test_fused_adam_ema_from_scratch_and_checkpoint_round_trip(golden_dir, tmp_path):
    from gecco_amd import checkpoint
    from gecco_amd.optim import FusedAdamEMA
    ck, g = _ckpt(golden_dir), _golden(golden_dir)
    sd0 = uncond_state_dict(W.linear_lift_state_dict(CASE["seed"], CASE["d"], CASE["L"], cases.I, cases.H))
    m = _model_on_gpu(sd0)
    opt = FusedAdamEMA(m.parameters(), lr=1e-4, ema_decay=CASE["decay"])
    params = list(m.parameters())
    for i in range(CASE["steps_before"]):
        opt.zero_grad()
        _inject(params, i)
        opt.step()
    # --- the checkpoint written here against the file the reference wrote at the same point
    path = str(tmp_path / "mine.ckpt")
    checkpoint.save_checkpoint(path, m, opt, epoch=0, global_step=CASE["steps_before"])
    mine = torch.load(path, map_location="cpu", weights_only=False)
    assert set(mine) == set(ck)
    assert _shape_tree(mine["state_dict"]) == _shape_tree(ck["state_dict"])
    assert _shape_tree(mine["ema_state_dict"]) == _shape_tree(ck["ema_state_dict"])
    a, b = mine["optimizer_states"][0], ck["optimizer_states"][0]
    assert sorted(a) == sorted(b) and a["current_step"] == b["current_step"] and a["decay"] == b["decay"]
    assert _shape_tree(a["opt"]["state"]) == _shape_tree(b["opt"]["state"])
    assert [sorted(x) for x in a["opt"]["param_groups"]] == [sorted(x) for x in b["opt"]["param_groups"]]
    for k in ck["state_dict"]:
        assert cpu_ref.rel_err(mine["state_dict"][k], ck["state_dict"][k])[0] <= 1e-6, k
        assert cpu_ref.rel_err(mine["ema_state_dict"][k], ck["ema_state_dict"][k])[0] <= 1e-6, k
    for j in range(len(params)):
        assert cpu_ref.rel_err(a["ema"][j], b["ema"][j])[0] <= 1e-6
        assert cpu_ref.rel_err(a["opt"]["state"][j]["exp_avg"], b["opt"]["state"][j]["exp_avg"])[0] <= 2e-6
        assert float(a["opt"]["state"][j]["step"]) == float(b["opt"]["state"][j]["step"])
    # the reference's own optimizer classes accept the file written here (torch.optim.Adam.load_state_dict)
    ref_opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros_like(q, device="cpu")) for q in params], lr=1e-4)
    ref_opt.load_state_dict(a["opt"])
    # --- continue to the golden end state
    for i in range(CASE["steps_before"], CASE["steps_before"] + CASE["steps_after"]):
        opt.zero_grad()
        _inject(params, i)
        opt.step()
    torch.cuda.synchronize()
    _compare_state(opt, params, g, "from scratch")
    # --- a fresh model + optimizer resume from the file written here and agree bit for bit with the run that wrote it
    m2 = _model_on_gpu(sd0)
    opt2 = FusedAdamEMA(m2.parameters(), lr=1e-4, ema_decay=CASE["decay"])
    checkpoint.load_checkpoint(path, m2, opt2, weights="raw")
    p2 = list(m2.parameters())
    for i in range(CASE["steps_before"], CASE["steps_before"] + CASE["steps_after"]):
        opt2.zero_grad()
        _inject(p2, i)
        opt2.step()
    for q, r in zip(params, p2):
        assert torch.equal(q, r)
    assert all(torch.equal(x, y) for x, y in zip(opt.ema_params, opt2.ema_params))


@pytest.mark.gpu
def test_ema_weights_reproduce_the_reference_forward(golden_dir):
    """Inference as gecco-torch/README.md:35-39: load ckpt["ema_state_dict"], evaluate."""
    from gecco_amd import checkpoint
    ck, g = _ckpt(golden_dir), _golden(golden_dir)
    m = _model_on_gpu(uncond_state_dict(W.linear_lift_state_dict(5, CASE["d"], CASE["L"], cases.I, cases.H)))
    checkpoint.load_checkpoint(ck, m, weights="ema")
    x, sigma = W.synthetic_cloud(CASE["seed"] + 7, 2, 128)
    with torch.no_grad():
        out = m.eval()(x.cuda(), sigma.cuda(), None)
    e = cpu_ref.rel_err(out.cpu(), torch.from_numpy(g["ema_forward"]))
    print("EMA-weights forward vs the reference:", e)
    assert e[0] <= 5e-5, e


@pytest.mark.gpu
def test_swap_ema_weights_and_foreign_grads():
    from gecco_amd.optim import FusedAdamEMA
    m = _model_on_gpu(uncond_state_dict(W.linear_lift_state_dict(CASE["seed"], CASE["d"], CASE["L"], cases.I, cases.H)))
    opt = FusedAdamEMA(m.parameters(), lr=1e-3, ema_decay=0.9)
    params = list(m.parameters())
    ref = torch.optim.Adam([torch.nn.Parameter(q.detach().cpu().clone()) for q in params], lr=1e-3)
    rp = [q for gq in ref.param_groups for q in gq["params"]]
    for i in range(2):
        # the Lightning pattern: module.zero_grad(set_to_none=True), backward leaves fresh gradient tensors behind
        m.zero_grad(set_to_none=True)
        grads = cases.optim_grads(i, [tuple(q.shape) for q in params])
        for q, gr, r in zip(params, grads, rp):
            q.grad = gr.cuda()
            r.grad = gr.clone()
        opt.step()
        ref.step()
    for q, r in zip(params, rp):
        assert cpu_ref.rel_err(q.detach().cpu(), r.detach())[0] <= 1e-6
    raw = [q.detach().clone() for q in params]
    shadows = [e.clone() for e in opt.ema_params]
    with opt.swap_ema_weights():
        swapped = [q.detach().clone() for q in params]
        assert all(torch.equal(q, e) for q, e in zip(swapped, shadows))   # the module now holds the EMA weights
    assert all(torch.equal(q, r) for q, r in zip(params, raw))            # swapped back
    assert any(not torch.equal(s, r) for s, r in zip(swapped, raw))       # and the EMA differs from the raw weights
