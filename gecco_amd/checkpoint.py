"""Checkpoint wire format of the reference's training runs.

The reference trains under Lightning with `EMACallback` (ema.py:32-184): a `.ckpt` file is `torch.save` of the dict
Lightning's `dump_checkpoint` builds — "epoch", "global_step", "pytorch-lightning_version", "state_dict" (the raw
weights), "loops", "callbacks", "optimizer_states" (one `EMAOptimizer.state_dict()` per optimizer: {"opt", "ema",
"current_step", "decay", "every_n_steps"}, ema.py:369-388), "lr_schedulers" — plus "ema_state_dict", which
`EMACallback.on_save_checkpoint` adds with the EMA weights swapped into the module (ema.py:174-184).  Inference loads
`ckpt["ema_state_dict"]` (gecco-torch/README.md:35-39).

`save_checkpoint` writes that dict from a `gecco_amd.Diffusion` and a `FusedAdamEMA`; `load_checkpoint` reads one —
written here or by the reference — into them.  Tensors are moved to the CPU on save (Lightning does the same through
`torch.save` of CPU-mapped state when asked; the layout does not depend on it) and to the module's device on load.
"""
from __future__ import annotations

from typing import Any

import torch

from .optim import FusedAdamEMA

LIGHTNING_VERSION = "2.0.0"   # the reference's pinned major (gecco-torch/pyproject.toml: lightning >= 2.0)


def _cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu().clone()
    if isinstance(obj, dict):
        return {k: _cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu(v) for v in obj)
    return obj


def ema_state_dict(model: torch.nn.Module, optimizer: FusedAdamEMA) -> dict[str, torch.Tensor]:
    """`model.state_dict()` with the EMA weights in place of the raw ones (EMACallback.on_save_checkpoint,
    ema.py:174-184): parameters registered in the optimizer are replaced by their shadows, buffers kept."""
    shadows = {id(p): e for p, e in zip(optimizer.all_parameters(), optimizer.ema_params)}
    by_name = dict(model.named_parameters())
    out = {}
    for k, v in model.state_dict().items():
        p = by_name.get(k)
        out[k] = (shadows[id(p)] if (p is not None and id(p) in shadows) else v).detach().clone()
    return out


def build_checkpoint(model: torch.nn.Module, optimizer: FusedAdamEMA | None = None, epoch: int = 0, global_step: int = 0,
                     extra: dict[str, Any] | None = None) -> dict[str, Any]:
    ckpt: dict[str, Any] = {
        "epoch": epoch,
        "global_step": global_step,
        "pytorch-lightning_version": LIGHTNING_VERSION,
        "state_dict": _cpu(dict(model.state_dict())),
        "loops": {},
        "callbacks": {},
        "optimizer_states": [],
        "lr_schedulers": [],
    }
    if optimizer is not None:
        ckpt["optimizer_states"] = [_cpu(optimizer.state_dict())]
        if optimizer.decay is not None:
            ckpt["ema_state_dict"] = _cpu(ema_state_dict(model, optimizer))
    if extra:
        ckpt.update(extra)
    return ckpt


def save_checkpoint(path: str, model: torch.nn.Module, optimizer: FusedAdamEMA | None = None, epoch: int = 0,
                    global_step: int = 0, extra: dict[str, Any] | None = None) -> dict[str, Any]:
    ckpt = build_checkpoint(model, optimizer, epoch, global_step, extra)
    torch.save(ckpt, path)
    return ckpt


def load_checkpoint(path_or_dict, model: torch.nn.Module, optimizer: FusedAdamEMA | None = None,
                    weights: str = "raw", strict: bool = True) -> dict[str, Any]:
    """Load a Lightning checkpoint (a path or an already loaded dict).

    weights = "raw": `state_dict` into the module (resuming training: the optimizer then receives
    `optimizer_states[0]`, EMA shadows included); weights = "ema": `ema_state_dict` into the module (inference, as
    gecco-torch/README.md:35-39 does).  Returns the checkpoint dict."""
    ckpt = torch.load(path_or_dict, map_location="cpu", weights_only=False) if isinstance(path_or_dict, (str, bytes)) \
        else path_or_dict
    if weights not in ("raw", "ema"):
        raise ValueError("weights must be 'raw' or 'ema'")
    key = "ema_state_dict" if weights == "ema" else "state_dict"
    if key not in ckpt:
        raise KeyError(f"checkpoint has no '{key}' (keys: {sorted(ckpt)})")
    model.load_state_dict(ckpt[key], strict=strict)
    if optimizer is not None:
        states = ckpt.get("optimizer_states") or []
        if not states:
            raise KeyError("checkpoint has no optimizer state")
        optimizer.load_state_dict(states[0])
    return ckpt
