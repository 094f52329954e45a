"""Which switch moves the 16-mixed gradients of a 96-point cloud (rows < 128: the 64-row tile forms)?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import cases, weights as W  # noqa: E402
from tests.test_modules_cpu import build_uncond, uncond_state_dict  # noqa: E402
from gecco_amd.structs import Example  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96


def run(amp):
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
    return float(loss), {n: p.grad.detach().clone() / 256.0 for n, p in m.named_parameters()}


lp, gp = run(False)
for rep in range(3):
    la, ga = run(True)
    rel = {n: float((ga[n] - gp[n]).norm() / gp[n].norm().clamp_min(1e-30)) for n in gp}
    tot = float(torch.cat([(ga[n] - gp[n]).flatten() for n in gp]).norm() / torch.cat([gp[n].flatten() for n in gp]).norm())
    worst = sorted(((v, n) for n, v in rel.items()), reverse=True)[:4]
    print(f"rep {rep}: loss {la:.5f} (plain {lp:.5f}) total {tot:.3e} worst {[(f'{v:.1e}', n[-40:]) for v, n in worst]}", flush=True)
