"""Focused micro-benchmark of the dominant kernel (fp32-MFMA fused GEMM) at the C2 call-site shapes.
Usage:  python tools/gemm_bench.py [iters] [site ...]     (also the target of the rocprofv3 --pmc passes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gecco_amd import hip_ops as ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    only = set(sys.argv[2:])
    dev = torch.device("cuda", 0)
    tot_f = tot_ms = 0.0
    precision = os.environ.get("GECCO_PRECISION", "bf16x3")
    print("precision", precision)
    for name, fl, _bytes, fn in bench.gemm_call_sites(ops, dev, precision):
        if only and name not in only:
            continue
        t = bench.time_events(fn, iters)
        tot_f += fl
        tot_ms += t
        print(f"{name:22s} {t:8.4f} ms  {fl / (t * 1e-3) / 1e12:7.2f} TFLOP/s")
    print(f"{'all':22s} {tot_ms:8.4f} ms  {tot_f / (tot_ms * 1e-3) / 1e12:7.2f} TFLOP/s")


if __name__ == "__main__":
    main()
