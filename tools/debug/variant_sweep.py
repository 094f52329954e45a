"""Training-path sweep over model VARIANTS the shipped configs do not use: the reference's default activation (nn.ReLU), other inducer
counts, head counts and MLP widths — exact-fp32 HIP kernels against split-bf16 and the 16-mixed setting.
    python tools/debug/variant_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import cases  # noqa: E402
from gecco_amd import autograd as ag, hip_ops  # noqa: E402
from gecco_amd.diffusion import Diffusion, EDMLoss, EDMPrecond, IdleConditioner, LogUniformSchedule  # noqa: E402
from gecco_amd.models.activation import GaussianActivation  # noqa: E402
from gecco_amd.models.linear_lift import LinearLift  # noqa: E402
from gecco_amd.models.set_transformer import SetTransformer  # noqa: E402
from gecco_amd.reparam import GaussianReparam  # noqa: E402
from gecco_amd.structs import Example  # noqa: E402


def build(d, act, I, H, mult):
    torch.manual_seed(1)
    net = LinearLift(inner=SetTransformer(n_layers=2, num_inducers=I, feature_dim=d, t_embed_dim=1, num_heads=H, activation=act,
                                          mlp_blowup=mult), feature_dim=d)
    return Diffusion(backbone=EDMPrecond(model=net), conditioner=IdleConditioner(),
                     reparam=GaussianReparam(torch.tensor(cases.GAUSS_MEAN), torch.tensor(cases.GAUSS_SIGMA)),
                     loss=EDMLoss(schedule=LogUniformSchedule(max=165.0)))


def run(cfg, N, mode, state):
    ag.WEIGHT_IMAGES.__init__()
    hip_ops.set_default_precision("fp32" if mode == "fp32" else "bf16x3")
    m = build(*cfg)
    if state is not None:
        m.load_state_dict(state)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda().train()
    x = torch.from_numpy(np.random.RandomState(4).randn(3, N, 3).astype(np.float32))
    data = (x * torch.tensor(cases.GAUSS_SIGMA) + torch.tensor(cases.GAUSS_MEAN)).cuda()
    torch.manual_seed(5)
    with torch.autocast("cuda", dtype=torch.float16, enabled=mode == "amp"):
        loss = m.training_step(Example(data, None), 0)
    (loss * 64.0).backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().clone() / 64.0 for n, p in m.named_parameters() if p.grad is not None}, state


bad = 0
for cfg in ((128, torch.nn.ReLU, 64, 8, 2), (384, torch.nn.ReLU, 64, 8, 2), (256, GaussianActivation, 32, 8, 2), (256, GaussianActivation, 96, 4, 2),
            (128, GaussianActivation, 64, 4, 4), (384, GaussianActivation, 64, 12, 2), (256, torch.nn.ReLU, 64, 16, 1)):
    for N in (96, 256, 1000):
        name = f"d={cfg[0]} act={cfg[1].__name__[:5]} I={cfg[2]} H={cfg[3]} mult={cfg[4]} N={N}"
        try:
            l0, g0, st = run(cfg, N, "fp32", None)
        except Exception as e:  # noqa: BLE001
            print(f"{name} fp32 FAILED: {str(e)[:300]}")
            bad += 1
            continue
        for mode, bar in (("bf16x3", 5e-4), ("amp", 1e-2)):
            try:
                l, g, _ = run(cfg, N, mode, st)
            except Exception as e:  # noqa: BLE001
                print(f"{name} {mode} FAILED: {str(e)[:300]}")
                bad += 1
                continue
            tot = float(torch.cat([(g[n] - g0[n]).flatten() for n in g0]).norm() / torch.cat([g0[n].flatten() for n in g0]).norm())
            fin = all(bool(torch.isfinite(v).all()) for v in g.values())
            flag = "" if (tot < bar and fin) else "   <-- OUTLIER"
            bad += bool(flag)
            print(f"{name:52s} {mode:6s} loss rel {abs(l - l0) / abs(l0):.1e} grads {tot:.1e}{flag}", flush=True)
print("outliers / failures:", bad)
