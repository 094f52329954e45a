"""Does it pay to start the second half batch of a two-stream evaluation LATE, so that one stream's point MLP (matrix / LDS bound) runs beside
the other's kv | q, pool and unpool (HBM bound) instead of beside the other's point MLP?  A spin kernel of d microseconds in front of the side
stream's evaluation; C2 (B = 64, N = 2048, d = 384, L = 6, w2), hipGraph replay, ms per evaluation.
    python tools/debug/stagger_sweep.py
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

B, N, d, L = 64, 2048, 384, 6
p = {k: v.cuda() for k, v in W.linear_lift_state_dict(3, d, L, 64, 8).items()}
x, sigma = W.synthetic_cloud(1, B, N)
x, sigma = x.cuda(), sigma.cuda()

# calibrate the spin kernel
torch.cuda._sleep(1000)
torch.cuda.synchronize()
t0 = time.perf_counter()
torch.cuda._sleep(10_000_000)
torch.cuda.synchronize()
cyc_per_us = 10_000_000 / ((time.perf_counter() - t0) * 1e6)
print(f"spin kernel: {cyc_per_us:.1f} cycles per us", flush=True)

orig = ops._two_stream_halves
delay_us = [0.0]


def patched(Bb, call, tensors, parts=2):
    def call2(lo, hi, idx):
        if idx == 1 and delay_us[0] > 0:
            torch.cuda._sleep(int(delay_us[0] * cyc_per_us))
        call(lo, hi, idx)
    return orig(Bb, call2, tensors, parts)


ops._two_stream_halves = patched
net = ops.LinearLiftPlan(p, 8, 64, precision="w2")
out = torch.empty_like(x)
ref = None
for frozen in (False, True):
    for dl in (0, 40, 80, 120, 160, 200, 240, 300, 360, 0):
        delay_us[0] = float(dl)
        ctx = ops.frozen_weights(net) if frozen else None
        if ctx:
            ctx.__enter__()
        net.forward(x, sigma, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            net.forward(x, sigma, out=out)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            g.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
        if ctx:
            ctx.__exit__(None, None, None)
        if ref is None:
            ref = out.clone()
        assert torch.equal(out, ref)
        print(f"frozen={int(frozen)} side stream starts {dl:4d} us late: {ms:.3f} ms per evaluation", flush=True)
