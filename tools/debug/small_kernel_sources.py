"""Which torch ops launch the small kernels (fills, copies, adds) inside one 16-mixed training step: torch.profiler with shapes and stacks.
    python tools/debug/small_kernel_sources.py [batch]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from gecco_amd import hip_ops as ops  # noqa: E402
from gecco_amd.optim import FusedAdamEMA  # noqa: E402
from gecco_amd.structs import Example  # noqa: E402

Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda:0")
model = bench.build_model(bench.random_state_dict(seed=3)).to(dev).train()
g = torch.Generator().manual_seed(100)
data = (torch.randn(Bt, bench.N, 3, generator=g) * model.reparam.sigma.cpu() + model.reparam.mean.cpu()).to(dev)
ex = Example(data, None)
ops.set_default_precision("mixed")
opt = FusedAdamEMA(list(model.parameters()), lr=1e-4, ema_decay=0.99)
scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 14)


def step(i):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        loss = model.training_step(ex, i)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()


for i in range(3):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step(3)
torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::add", "aten::add_", "aten::clone", "aten::contiguous",
                  "aten::cat", "aten::mul", "aten::zeros_like", "aten::new_zeros", "aten::slice_backward", "aten::select_backward"):
        st = [s for s in (e.stack or []) if "gecco_amd" in s or "bench.py" in s]
        c[(e.name, str(e.input_shapes)[:60], st[0][-70:] if st else "(autograd engine / torch)")] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:60]:
    print(v, k)
