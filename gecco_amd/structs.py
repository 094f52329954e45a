"""Generation context and training example containers (API of reference structs.py:61-91)."""
from __future__ import annotations

from typing import Callable, NamedTuple

import torch
from torch import Tensor


def _fields(obj):
    return obj._asdict().items()


def _apply(obj, f: Callable[[Tensor], Tensor]):
    """Out-of-place map over every tensor field (recursing into nested containers)."""
    new = {}
    for name, value in _fields(obj):
        if hasattr(value, "apply_to_tensors"):
            new[name] = value.apply_to_tensors(f)
        elif torch.is_tensor(value):
            new[name] = f(value)
        else:
            new[name] = value
    return type(obj)(**new)


def _describe(obj, indent: int = 0) -> str:
    pad = " " * indent
    lines = [f"{type(obj).__name__}("]
    for name, value in _fields(obj):
        if hasattr(value, "_describe"):
            lines.append(f"{pad} {name}={value._describe(indent + 1)}")
        elif torch.is_tensor(value):
            lines.append(f"{pad} {name}={tuple(value.shape)},")
        else:
            lines.append(f"{pad} {name}={value},")
    lines.append(f"{pad})")
    return "\n".join(lines)


class DataError(RuntimeError):
    pass


class Context3d(NamedTuple):
    """Conditioning image (B, 3, H, W) and 3x3 camera intrinsics K (B, 3, 3), normalised so that
    projected points land in [0, 1]^2."""
    image: Tensor
    K: Tensor

    apply_to_tensors = _apply
    _describe = _describe

    def __repr__(self) -> str:
        return _describe(self)


class Example(NamedTuple):
    """A point cloud (B, N, 3) and its (optional) context."""
    data: Tensor
    ctx: Context3d | None

    apply_to_tensors = _apply
    _describe = _describe

    def __repr__(self) -> str:
        return _describe(self)
