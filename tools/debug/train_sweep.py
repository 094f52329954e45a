"""Robustness sweep of the training path: loss and every parameter gradient of small unconditional models over odd cloud sizes and
widths, exact-fp32 HIP kernels against split-bf16 and against the 16-mixed setting (looking for outliers, not for timing).
    python tools/debug/train_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import cases, weights as W  # noqa: E402
from tests.test_modules_cpu import build_uncond, uncond_state_dict  # noqa: E402
from gecco_amd import autograd as ag, hip_ops  # noqa: E402
from gecco_amd.structs import Example  # noqa: E402


def run(d, L, N, B, mode):
    ag.WEIGHT_IMAGES.__init__()
    hip_ops.set_default_precision("fp32" if mode == "fp32" else "bf16x3")
    torch.manual_seed(0)
    m = build_uncond(d, L)
    m.load_state_dict(uncond_state_dict(W.linear_lift_state_dict(9, d, L, cases.I, cases.H)))
    m = m.cuda().train()
    x = torch.from_numpy(np.random.RandomState(4).randn(B, N, 3).astype(np.float32))
    data = (x * torch.tensor(cases.GAUSS_SIGMA) + torch.tensor(cases.GAUSS_MEAN)).cuda()
    torch.manual_seed(5)
    with torch.autocast("cuda", dtype=torch.float16, enabled=mode == "amp"):
        loss = m.training_step(Example(data, None), 0)
    (loss * 64.0).backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().clone() / 64.0 for n, p in m.named_parameters()}


bad = 0
for d in (64, 128, 256, 384, 512):
    for N in (40, 64, 65, 100, 127, 128, 129, 192, 200, 256, 1000, 1024):
        B = 2 if d >= 384 else 3
        try:
            l0, g0 = run(d, 2, N, B, "fp32")
        except Exception as e:  # noqa: BLE001
            print(f"d={d} N={N} fp32 FAILED: {str(e)[:200]}")
            bad += 1
            continue
        for mode, bar in (("bf16x3", 3e-4), ("amp", 4e-3)):
            try:
                l, g = run(d, 2, N, B, mode)
            except Exception as e:  # noqa: BLE001
                print(f"d={d} N={N} {mode} FAILED: {str(e)[:200]}")
                bad += 1
                continue
            tot = float(torch.cat([(g[n] - g0[n]).flatten() for n in g0]).norm() / torch.cat([g0[n].flatten() for n in g0]).norm())
            worst = max(((float((g[n] - g0[n]).norm() / g0[n].norm().clamp_min(1e-20)), n) for n in g0 if not n.endswith(".alpha")), key=lambda t: t[0])
            walpha = max(((float((g[n] - g0[n]).norm() / g0[n].norm().clamp_min(1e-20)), n) for n in g0 if n.endswith(".alpha")), key=lambda t: t[0])
            flag = "" if (tot < bar and worst[0] < 10 * bar and np.isfinite(tot)) else "   <-- OUTLIER"
            if flag:
                bad += 1
            print(f"d={d:3d} N={N:4d} {mode:6s} loss rel {abs(l - l0) / abs(l0):.1e} grads {tot:.1e} worst {worst[0]:.1e} ({worst[1][-34:]}) alpha {walpha[0]:.1e}{flag}", flush=True)
print("outliers / failures:", bad)
