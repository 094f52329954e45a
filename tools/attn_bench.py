"""Micro-benchmark of the two attention kernels at C2 (B=64, N=2048, d=384, H=8) in the fp16 mode with fp16 tensors.
Usage: python tools/attn_bench.py [iters] [d]      (also the target of rocprofv3 --pmc passes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gecco_amd import hip_ops as ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    B, N, D, H = bench.B, bench.N, (int(sys.argv[2]) if len(sys.argv) > 2 else bench.D), bench.H
    g = torch.Generator().manual_seed(0)
    kv16 = torch.randn(B, N, 2 * D, generator=g).half().to(dev)
    q16 = torch.randn(B, N, D, generator=g).half().to(dev)
    ind = torch.randn(1, H, 64, D // H, generator=g).to(dev)
    kvh = torch.randn(B, 64, 2 * D, generator=g).to(dev)
    out16 = torch.empty_like(q16)
    for name, by, fn in (("pool (fp16 K|V in)", kv16.numel() * 2, lambda: ops.pool_attn_f16in(kv16, ind, H)),
                         ("unpool (fp16 q in, out)", q16.numel() * 4, lambda: ops.unpool_attn_f16io(q16, kvh, H, out=out16))):
        t = bench.time_events(fn, iters)
        print(f"{name:26s} {t * 1e3:8.1f} us  {by / (t * 1e-3) / 1e12:6.2f} TB/s of algorithmic bytes ({by / 1e6:.0f} MB)")


if __name__ == "__main__":
    main()
