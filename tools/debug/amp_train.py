"""Training step under the reference's trainer setting (torch.autocast(float16) + GradScaler) against the plain step: time per
step and the deviation of every parameter gradient from the exact-fp32 run of the same batch and noise.
    python tools/debug/amp_train.py [batch] [steps]"""
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")


def setup():
    ag.WEIGHT_IMAGES.__init__()
    model = bench.build_model(bench.random_state_dict(seed=3)).to(dev).train()
    g = torch.Generator().manual_seed(100)
    data = (torch.randn(Bt, bench.N, 3, generator=g) * model.reparam.sigma.cpu() + model.reparam.mean.cpu()).to(dev)
    return model, Example(data, None)


def grads(mode):
    model, ex = setup()
    ops.set_default_precision("fp32" if mode == "fp32" else "mixed")
    torch.manual_seed(7)
    if mode == "amp":
        with torch.autocast("cuda", dtype=torch.float16):
            loss = model.training_step(ex, 0)
        (loss * 2.0 ** 14).backward()
        gs = [p.grad.detach().clone() / 2.0 ** 14 for p in model.parameters()]
    else:
        loss = model.training_step(ex, 0)
        loss.backward()
        gs = [p.grad.detach().clone() for p in model.parameters()]
    torch.cuda.synchronize()
    return float(loss.detach()), gs, [n for n, _ in model.named_parameters()]


def timing(mode):
    model, ex = setup()
    ops.set_default_precision("mixed")
    opt = FusedAdamEMA(list(model.parameters()), lr=1e-4, ema_decay=0.99)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 14) if mode == "amp" else None

    def step(i):
        opt.zero_grad(set_to_none=True)
        if scaler is None:
            loss = model.training_step(ex, i)
            loss.backward()
            opt.step()
        else:
            with torch.autocast("cuda", dtype=torch.float16):
                loss = model.training_step(ex, i)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
        return loss
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(i)
    host = (time.perf_counter() - t0) / steps * 1e3   # host issue time (the device may still be running)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"   host issue {host:.2f} ms per step")
    return ms, float(loss.detach()), (scaler.get_scale() if scaler else None)


if os.environ.get("AMP_SKIP_GRADS", "0") != "1":
    l32, g32, names = grads("fp32")
    for mode in ("plain", "amp"):
        l, g, _ = grads(mode)
        rel = [float((a - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(g, g32)]
        worst = sorted(zip(rel, names), reverse=True)[:6]
        tot = float(torch.cat([(a - b).flatten() for a, b in zip(g, g32)]).norm() / torch.cat([b.flatten() for b in g32]).norm())
        print(f"{mode:6s} loss {l:.6f} (fp32 {l32:.6f}, rel {abs(l - l32) / abs(l32):.2e})  gradient rel-L2: all params {tot:.2e}, "
              f"median tensor {sorted(rel)[len(rel) // 2]:.2e}, worst {[(f'{r:.2e}', n) for r, n in worst]}", flush=True)
for mode in os.environ.get("AMP_MODES", "plain,amp").split(","):
    ms, l, sc = timing(mode)
    print(f"{mode:6s} {ms:.2f} ms per step, loss {l:.4f}, scale {sc}", flush=True)
