"""Set-vs-set Chamfer distances at the evaluation protocol's size (gecco-jax benchmark.py:21-39: every generated cloud against every
reference cloud): S = T = 256 clouds of N = 2048 points by default — 2 x 2.7e11 point pairs — timed with HIP events, against the vector-ALU
roofline (the inner dimension is 3: no matrix-core shape).  Algorithmic work per point pair and direction: 3 FMA + 1 min = 7 flops.

  python tools/metrics_bench.py [S] [N]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import metrics
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    g = torch.Generator().manual_seed(0)
    a = torch.randn(S, N, 3, generator=g).cuda()
    b = torch.randn(S, N, 3, generator=g).cuda()
    metrics.pairwise_set_distance(a[:8], b[:8])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        D = metrics.pairwise_set_distance(a, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    pairs = 2.0 * S * S * N * N
    flops = 7.0 * pairs
    rec = {"workload": f"set-vs-set Chamfer, S = T = {S} clouds x N = M = {N} points (gecco_set_chamfer_f32, two launches)", "ms": ms,
           "point_pairs_per_s": pairs / ms * 1e3,
           "roofline": {"bound": "valu", "achieved": flops / ms / 1e9, "peak": 157.3, "unit": "TFLOP/s (fp32 vector)", "frac": flops / ms / 1e9 / 157.3,
                        "flops_per_point_pair": 7},
           "checksum": float(D.double().sum())}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
