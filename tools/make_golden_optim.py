"""Golden vectors for the optimizer / checkpoint rows, from the REAL reference
(build container only; never runs on the GPU box).

What runs here is the reference's own code: `gecco_torch.ema.EMAOptimizer` (ema.py:200-400) wrapped around
`torch.optim.Adam(lr=1e-4)` (what `Diffusion.configure_optimizers` returns, diffusion.py:210-211) on the reference's
`Diffusion` module (d=64, L=1), stepped on injected gradients (seeded, so the HIP side can replay them); the checkpoint
dict is assembled the way Lightning's `dump_checkpoint` + `EMACallback.on_save_checkpoint` (ema.py:174-184) do — with
the reference's `save_ema_model` weight swap — and written with `torch.save`.

Outputs (data only — tensors and scalars, no reference source):
  tests/golden/ref_ckpt_d64_L1.ckpt   the checkpoint after 3 steps
  tests/golden/optim.npz              parameters / EMA / Adam moments after 2 MORE steps, and the EMA-weights forward
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases  # noqa: E402
from oracle import weights as W  # noqa: E402
from tools import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CASE = cases.OPTIM_CASE


def main():
    torch.manual_seed(0)
    ns = ref_import.load()
    # ema.py's import-time needs beyond ref_import's stubs (names only; none of them runs in this script)
    pl = sys.modules["lightning.pytorch"]
    ex = types.ModuleType("lightning.pytorch.utilities.exceptions")
    ex.MisconfigurationException = type("MisconfigurationException", (Exception,), {})
    rz = types.ModuleType("lightning.pytorch.utilities.rank_zero")
    rz.rank_zero_info = print
    ut = types.ModuleType("lightning.pytorch.utilities")
    ut.exceptions, ut.rank_zero = ex, rz
    pl.utilities = ut
    pl.Trainer = object
    sys.modules.update({"lightning.pytorch.utilities": ut, "lightning.pytorch.utilities.exceptions": ex,
                        "lightning.pytorch.utilities.rank_zero": rz})
    from gecco_torch import ema as ema_mod

    d, L = CASE["d"], CASE["L"]
    model = ref_import.build_uncond(ns, d, L, cases.I, cases.H)
    p = W.linear_lift_state_dict(CASE["seed"], d, L, cases.I, cases.H)
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.mean"], sd["reparam.sigma"] = model.reparam.mean, model.reparam.sigma
    model.load_state_dict(sd, strict=True)

    inner = model.configure_optimizers()                        # torch.optim.Adam(lr=1e-4), diffusion.py:210-211
    opt = ema_mod.EMAOptimizer(inner, device=torch.device("cpu"), decay=CASE["decay"], every_n_steps=1, current_step=0)
    params = [q for g in opt.param_groups for q in g["params"]]

    def step(i):
        for q, g in zip(params, cases.optim_grads(i, [tuple(q.shape) for q in params])):
            q.grad = g
        opt.step()
        opt.join()

    for i in range(CASE["steps_before"]):
        step(i)

    # the checkpoint as Lightning + EMACallback write it
    opt.switch_main_parameter_weights(saving_ema_model=True)    # EMACallback.save_ema_model (ema.py:115-125)
    try:
        ema_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    finally:
        opt.switch_main_parameter_weights(saving_ema_model=False)
    ckpt = {
        "epoch": 0, "global_step": CASE["steps_before"], "pytorch-lightning_version": "2.0.0",
        "state_dict": {k: v.detach().clone() for k, v in model.state_dict().items()},
        "loops": {}, "callbacks": {},
        "optimizer_states": [opt.state_dict()], "lr_schedulers": [],
        "ema_state_dict": ema_sd,
    }
    path = os.path.join(OUT, "ref_ckpt_d64_L1.ckpt")
    torch.save(ckpt, path)
    print(f"-> {os.path.relpath(path, ROOT)} ({os.path.getsize(path) / 1024:.0f} KiB); optimizer state keys "
          f"{sorted(ckpt['optimizer_states'][0])}")

    for i in range(CASE["steps_before"], CASE["steps_before"] + CASE["steps_after"]):
        step(i)
    out = {}
    st = opt.optimizer.state_dict()["state"]
    for j, q in enumerate(params):
        out[f"p{j}"] = q.detach().numpy().copy()
        out[f"ema{j}"] = opt.ema_params[j].numpy().copy()
        out[f"m{j}"] = st[j]["exp_avg"].numpy().copy()
        out[f"v{j}"] = st[j]["exp_avg_sq"].numpy().copy()
    out["adam_step"] = np.asarray(float(st[0]["step"]))
    out["current_step"] = np.asarray(opt.current_step)
    # the forward of the EMA weights (what inference does with ckpt["ema_state_dict"], README.md:35-39)
    model.load_state_dict(ema_sd, strict=True)
    x, sigma = W.synthetic_cloud(CASE["seed"] + 7, 2, 128)
    with torch.no_grad():
        out["ema_forward"] = model(x, sigma, None).numpy()
    np.savez(os.path.join(OUT, "optim.npz"), **out)
    print(f"-> tests/golden/optim.npz ({os.path.getsize(os.path.join(OUT, 'optim.npz')) / 1024:.0f} KiB), "
          f"{len(params)} parameters, adam step {float(st[0]['step'])}, current_step {opt.current_step}")


if __name__ == "__main__":
    main()
