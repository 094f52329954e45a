"""Soak of the one-launch point MLP at every width: random (B, rows, activation, seed), out of place and in place, with and without the
statistics, against float64 on the first clouds and for run-to-run reproducibility; launches larger than the grid (persistent blocks, the
weight stream wraps), GECCO_MLPW_CUS forcing odd grids.
    python tools/debug/mlpw_soak.py [cases]
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from gecco_amd import hip_ops as ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(os.environ.get("SOAK_SEED", "1")))
worst = {}
for it in range(n):
    K = int(rs.choice([128, 256, 384, 512]))
    Wd = 2 * K
    rows = 128 * int(rs.randint(1, 9))
    B = int(rs.choice([1, 2, 3, 7, 40, 150, 333]))
    act = str(rs.choice(["gauss", "relu", "none"]))
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    x, W0, b0 = t(rs.randn(B, rows, K)), t(rs.randn(Wd, K) / math.sqrt(K)), t(rs.randn(Wd) / math.sqrt(K))
    W2, b2 = t(rs.randn(K, Wd) / math.sqrt(Wd)), t(rs.randn(K) / math.sqrt(Wd))
    pa, po = t(1 + 0.3 * rs.randn(B, K)), t(0.3 * rs.randn(B, K))
    alpha = t(np.array(0.9))
    kw = dict(act_alpha=alpha.cuda()) if act == "gauss" else dict(act="relu") if act == "relu" else {}
    xc, pro = x.cuda(), (pa.cuda(), po.cuda())
    got, st = ops.mlp_fused_w(xc, pro, W0.cuda(), b0.cuda(), W2.cuda(), b2.cuda(), want_stats=True, out=torch.empty_like(xc), **kw)
    again = ops.mlp_fused_w(xc.clone(), pro, W0.cuda(), b0.cuda(), W2.cuda(), b2.cuda(), **kw)[0]     # in place, no statistics
    assert torch.equal(got, again), (it, K, rows, B, act)
    actf = (lambda u: (torch.exp(-u * u / (2 * 0.9 ** 2)) - 0.7) / 0.28) if act == "gauss" else torch.relu if act == "relu" else (lambda u: u)
    nb = min(B, 3)
    y = torch.addcmul(po[:nb, None], x[:nb], pa[:nb, None]).double()
    ref = F.linear(actf(F.linear(y, W0.double(), b0.double())), W2.double(), b2.double())
    e = ((got[:nb].cpu().double() - x[:nb].double()) - ref).abs().max().item() / ref.abs().max().item()
    g4 = got.double().reshape(B, rows // 128, 128, K)
    es = (st[:, :, 0].double() - g4.sum(2)).abs().max().item() / max(1.0, got.abs().max().item())
    worst[K] = max(worst.get(K, 0.0), e)
    assert e <= 5e-4 and es <= 1e-3 and torch.isfinite(got).all(), (it, K, rows, B, act, e, es)
    print(f"{it:3d}: d={K} B={B} rows={rows} {act}: whole MLP {e:.2e} of its scale, statistics {es:.1e}", flush=True)
print("worst per width:", {k: f"{v:.2e}" for k, v in sorted(worst.items())})
