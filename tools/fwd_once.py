"""A few C2 evaluations and nothing else: the target of the rocprofv3 --pmc passes that need every kernel of the
forward (attention, inducer chain) and not only the GEMM call sites.  Usage: python tools/fwd_once.py [evaluations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gecco_amd import hip_ops as ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ops.set_default_precision(os.environ.get("GECCO_PRECISION", "w2"))
    dev = torch.device("cuda", 0)
    model = bench.build_model(bench.random_state_dict(0)).to(dev).eval()
    x, sigma = (t.to(dev) for t in bench.synthetic_cloud(1))
    with torch.no_grad():
        for _ in range(n):
            out = model(x, sigma, None)
    torch.cuda.synchronize()
    print("ok", float(out.abs().mean()))


if __name__ == "__main__":
    main()
