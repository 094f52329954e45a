"""Host-side issue time of a C2 training step against its device time: is the eager step launch-bound?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from gecco_amd import hip_ops as ops
from gecco_amd.optim import FusedAdamEMA
from gecco_amd.structs import Example
ops.set_default_precision("mixed")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(100)
model = bench.build_model(bench.random_state_dict(seed=3)).to(dev).train()
data = (torch.randn(48, bench.N, 3, generator=g) * model.reparam.sigma.cpu() + model.reparam.mean.cpu()).to(dev)
ex = Example(data, None)
opt = FusedAdamEMA(list(model.parameters()), lr=1e-4, ema_decay=0.99)
def step(i):
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(ex, i)
    loss.backward()
    opt.step()
for i in range(5): step(i)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for i in range(20):
    a = time.perf_counter(); step(i); host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue ms/step (median)", sorted(host)[10] * 1e3, "issue total", (t1 - t0) * 1e3 / 20, "wall ms/step", (t2 - t0) * 1e3 / 20, "drain after last issue ms", (t2 - t1) * 1e3)
