"""How far the gradients of the reference's algorithm move when it runs in its own shipped trainer precision.

Both shipped configs train with Lightning's precision="16-mixed" (example_configs/shapenet_airplane_unconditional.py:74,
taskonomy_conditional.py:102): `training_step` under torch.autocast(float16).  This script takes the oracle's restatement of the
unconditional denoiser + EDM loss (oracle/cpu_ref.py, torch ops on the host cores) and differentiates it twice — plain fp32 and
under torch.autocast("cpu", float16), where every nn.functional.linear / matmul / bmm runs with fp16 operands AND fp16 results as
on the GPU — and prints the deviation of the loss and of every parameter gradient.  It is the yardstick for the HIP training
path's autocast arithmetic (fp16 operands, fp32 accumulation, fp32 tensors between kernels: gecco_amd/autograd.py
`_lin_precision`), which must not be further from the fp32 gradients than the reference's own setting is.
    python tools/experiments/autocast_grad_deviation.py [N] [d] [L]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import cases, cpu_ref  # noqa: E402
from oracle import weights as W  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
L = int(sys.argv[3]) if len(sys.argv) > 3 else 6
B = 2
p = W.linear_lift_state_dict(3, d, L, cases.I, cases.H)
rs = np.random.RandomState(11)
data = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
noise = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32))
sigma = torch.tensor([0.1, 5.0])


def edm_loss(D):
    s = sigma.reshape(-1, 1, 1)
    return (100.0 * (s ** 2 + 1.0) / s ** 2 * (D(data + noise * s, sigma).float() - data) ** 2).mean()


def run(dtype):
    if dtype is None:
        pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        loss = edm_loss(cpu_ref.uncond_denoiser(pr, "", cases.H))
        loss.backward()
        return float(loss.detach()), {k: v.grad for k, v in pr.items()}, 1.0
    scale = 2.0 ** 16   # GradScaler's start; halved until the gradients are finite, as the scaler does
    while True:
        pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        with torch.autocast("cpu", dtype=dtype):
            loss = edm_loss(cpu_ref.uncond_denoiser(pr, "", cases.H))
        (loss * scale).backward()
        if all(bool(torch.isfinite(v.grad).all()) for v in pr.values() if v.grad is not None):
            return float(loss.detach()), {k: (v.grad / scale if v.grad is not None else None) for k, v in pr.items()}, scale
        scale /= 2.0


l32, g32, _ = run(None)
for name, dt in (("autocast(float16)", torch.float16), ("autocast(bfloat16)", torch.bfloat16)):
    l, g, sc = run(dt)
    rel = {k: float((g[k] - g32[k]).norm() / g32[k].norm().clamp_min(1e-30)) for k in g32 if g32[k] is not None and g[k] is not None}
    tot = float(torch.cat([(g[k] - g32[k]).flatten() for k in rel]).norm() / torch.cat([g32[k].flatten() for k in rel]).norm())
    mats = [v for k, v in rel.items() if g32[k].dim() == 2 and min(g32[k].shape) > 1]
    worst = sorted(((v, k) for k, v in rel.items()), reverse=True)[:4]
    print(f"N={N} d={d} L={L}  {name:20s} (loss scale 2^{int(np.log2(sc))}) loss rel {abs(l - l32) / abs(l32):.2e}; gradient rel-L2: all parameters {tot:.2e}, "
          f"median tensor {sorted(rel.values())[len(rel) // 2]:.2e}, worst matrix {max(mats):.2e}, worst {[(f'{v:.1e}', k) for v, k in worst]}")
