"""Which Python lines issue the device-to-device copies of a training step?  (tools/probe: diagnostic)"""
import collections, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from gecco_amd import hip_ops as ops
from gecco_amd.optim import FusedAdamEMA
from gecco_amd.structs import Example
import __graft_entry__ as ge
ge.build()
ops.set_default_precision("mixed")
dev = torch.device("cuda:0")
model = bench.build_model(bench.random_state_dict(seed=3)).to(dev).train()
g = torch.Generator().manual_seed(1)
data = (torch.randn(8, bench.N, 3, generator=g) * model.reparam.sigma.cpu() + model.reparam.mean.cpu()).to(dev)
opt = FusedAdamEMA(list(model.parameters()), lr=1e-4, ema_decay=0.99)
def step(i):
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(Example(data, None), i)
    loss.backward()
    opt.step()
for i in range(3):
    step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(3)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::zero_", "aten::fill_"):
        st = [s for s in ev.stack if "gecco_amd" in s or "bench.py" in s]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"))] += 1
for k, v in cnt.most_common(30):
    print(v, k)
