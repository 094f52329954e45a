"""Training-step timing on the HIP training path: forward + backward + Adam at the shipped config
(example_configs/shapenet_airplane_unconditional.py: batch 48, N=2048, d=384, L=6).  python tools/train_bench.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gecco_amd.structs import Example  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda", 0)
    m = bench.build_model(bench.random_state_dict(3)).to(dev).train()
    opt = m.configure_optimizers()
    g = torch.Generator().manual_seed(0)
    data = (torch.randn(B, bench.N, 3, generator=g) * m.reparam.sigma.cpu() + m.reparam.mean.cpu()).to(dev)

    def step(i):
        opt.zero_grad(set_to_none=True)
        loss = m.training_step(Example(data, None), i)
        loss.backward()
        opt.step()
        return loss

    step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(i + 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fl = 3 * bench.flops_per_sample() * B  # fwd + ~2x for bwd
    print(f"train step B={B}: {dt * 1e3:.1f} ms  ({B * bench.N / dt:.3e} points/s, ~{fl / dt / 1e12:.1f} TFLOP/s algorithmic), "
          f"loss {float(loss.detach()):.3f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
