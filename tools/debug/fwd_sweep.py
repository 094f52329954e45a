"""Robustness sweep of the forward (no-grad) path: small unconditional models over odd cloud sizes, widths and batch sizes, the exact-fp32
HIP kernels against every other arithmetic mode (looking for outliers: the modes route to different kernels by shape).
    python tools/debug/fwd_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import cases, weights as W  # noqa: E402
from tests.test_modules_cpu import build_uncond, uncond_state_dict  # noqa: E402
from gecco_amd import hip_ops  # noqa: E402

BARS = {"bf16x3": 2e-4, "mixed": 4e-4, "fp16": 3e-3}
bad = 0
for d in (128, 256, 384, 512):
    m = build_uncond(d, 3)
    m.load_state_dict(uncond_state_dict(W.linear_lift_state_dict(9, d, 3, cases.I, cases.H)))
    m = m.cuda().eval()
    for N in (40, 64, 100, 128, 129, 200, 256, 1000, 2048, 4096):
        for B in (1, 3):
            rs = np.random.RandomState(N + B)
            x = torch.from_numpy(rs.randn(B, N, 3).astype(np.float32)).cuda()
            sigma = torch.from_numpy(np.exp(rs.uniform(np.log(0.002), np.log(80.0), size=B)).astype(np.float32)).cuda()
            outs = {}
            for mode in ("fp32", "bf16x3", "mixed", "fp16"):
                hip_ops.set_default_precision(mode)
                try:
                    with torch.no_grad():
                        outs[mode] = m(x * (1 + sigma.view(-1, 1, 1)), sigma, None).float()
                except Exception as e:  # noqa: BLE001
                    print(f"d={d} N={N} B={B} {mode} FAILED: {str(e)[:200]}")
                    bad += 1
            ref = outs.get("fp32")
            line = f"d={d:3d} N={N:4d} B={B}"
            for mode in ("bf16x3", "mixed", "fp16"):
                if mode not in outs or ref is None:
                    continue
                e = float((outs[mode] - ref).abs().max() / ref.abs().max())
                flag = "" if (e < BARS[mode] and np.isfinite(e)) else "<--OUTLIER"
                bad += bool(flag)
                line += f"  {mode} {e:.1e}{flag}"
            print(line, flush=True)
hip_ops.set_default_precision("mixed")
print("outliers / failures:", bad)
