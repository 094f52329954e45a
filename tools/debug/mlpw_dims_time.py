"""The one-launch point MLP of the "w2" mode against the mixed mode's two launches at feature_dim 128 / 256 / 384 (option "mlpw" on / off):
ms per evaluation of the unconditional denoiser, B = 64, N = 2048, L = 6 (L = 4 at d = 128: BASELINE.json's C1), one stream, graph replay.
    python tools/debug/mlpw_dims_time.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from gecco_amd import hip_ops as ops  # noqa: E402
from oracle import weights as W  # noqa: E402  (seeded weights only)

B, N = 64, 2048
for d, L in ((128, 4), (256, 6), (384, 6), (512, 6)):
    p = {k: v.cuda() for k, v in W.linear_lift_state_dict(3, d, L, 64, 8).items()}
    x, sigma = W.synthetic_cloud(1, B, N)
    x, sigma = x.cuda(), sigma.cuda()
    row = []
    for on in (1, 0):
        net = ops.LinearLiftPlan(p, 8, 64, precision="w2", options={"mlpw": on})
        out = torch.empty_like(x)
        with ops.frozen_weights(net):
            net.forward(x, sigma, out=out)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                net.forward(x, sigma, out=out)
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                g.replay()
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / 30 * 1e3)
    print(f"d={d} L={L}: one launch {row[0]:.3f} ms, two launches {row[1]:.3f} ms per evaluation (B={B}, N={N})", flush=True)
