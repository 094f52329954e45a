"""Is the 16-mixed training step host-bound?  The same step eager and as ONE captured hipGraph (forward + backward + fused Adam/EMA), C2 shape.
    python tools/debug/graph_train.py [batch] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from gecco_amd import autograd as ag  # noqa: E402
from gecco_amd import hip_ops as ops  # noqa: E402
from gecco_amd.optim import FusedAdamEMA  # noqa: E402
from gecco_amd.structs import Example  # noqa: E402

Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 48
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
ops.set_default_precision("mixed")
model = bench.build_model(bench.random_state_dict(seed=3)).to(dev).train()
g = torch.Generator().manual_seed(100)
data = (torch.randn(Bt, bench.N, 3, generator=g) * model.reparam.sigma.cpu() + model.reparam.mean.cpu()).to(dev)
ex = Example(data, None)
opt = FusedAdamEMA(list(model.parameters()), lr=1e-4, ema_decay=0.99, amp_on_device=True)
scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 14)


def step(i):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        loss = model.training_step(ex, i)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    return loss


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, host


for i in range(4):
    loss = step(i)
ms, host = timed(step, steps)
print(f"eager: {ms:.2f} ms per step (host issue {host:.2f}), loss {float(loss):.4f}")

mode = os.environ.get("GRAPH_MODE", "fwdbwd")
try:
    # capture on a side stream as torch requires; the weight-gradient side stream joins the capture through its event waits
    os.environ.setdefault("GECCO_TRAIN_DW_STREAM_CAPTURE", "1")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3):
            step(i)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        with torch.autocast("cuda", dtype=torch.float16):
            gloss = model.training_step(ex, 0)
        scaler.scale(gloss).backward()
        if mode == "all":
            scaler.step(opt)
            scaler.update()

    def gstep(i):
        graph.replay()
        if mode != "all":
            scaler.step(opt)
            scaler.update()
    for i in range(3):
        gstep(i)
    ms, host = timed(gstep, steps)
    print(f"graph ({mode}): {ms:.2f} ms per step (host issue {host:.2f}), loss {float(gloss):.4f}, scale {scaler.get_scale()}")
except Exception as e:   # noqa: BLE001
    import traceback
    traceback.print_exc()
    print("graph capture failed:", repr(e)[:500])
