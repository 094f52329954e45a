"""Does the C2 evaluation run faster as sub-batches whose intermediates fit the 256 MB Infinity Cache?
python tools/batch_split_bench.py  ->  ms per evaluation of 64 clouds as 1 x 64, 2 x 32, 4 x 16 (hipGraph replay each)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gecco_amd import hip_ops as ops  # noqa: E402


def main():
    ops.set_default_precision("fp16")
    dev = torch.device("cuda", 0)
    model = bench.build_model(bench.random_state_dict(3)).to(dev).eval()
    x, sigma = (t.to(dev) for t in bench.synthetic_cloud(0))
    for parts in (1, 2, 4):
        bs = bench.B // parts
        runs = [model.graphed_forward(x[i * bs:(i + 1) * bs].contiguous(), sigma[i * bs:(i + 1) * bs].contiguous(), None)
                for i in range(parts)]

        def step():
            for r in runs:
                r()
        t = bench.time_events(step, 20, warmup=3)
        print(f"{parts} x {bs:2d} clouds: {t:.3f} ms per 64 clouds = {bench.B * bench.N / t * 1e3:.3e} points/s")


if __name__ == "__main__":
    main()
