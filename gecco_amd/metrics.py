"""Evaluation metrics on generated clouds, on the HIP device.

API of gecco-jax/src/gecco_jax/metrics.py:92-156 and geometry.py:8-24, batched: every function takes (B, N, 3) / (B, M, 3)
fp32 HIP tensors (or single (N, 3) clouds) and returns one value per sample; the JAX package vmaps single clouds.
`scipy_emd` solves the assignment on the host with scipy, exactly as the reference does (its `_scipy_lsa` is a
`jax.pure_callback` into `scipy.optimize.linear_sum_assignment`, metrics.py:108-121) on the distance matrix the device
computed.  There is no CPU fallback for the device parts."""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor

from . import _lib
from .hip_ops import _ptr, _stream


def _batched(a: Tensor, b: Tensor):
    single = a.dim() == 2
    if single:
        a, b = a[None], b[None]
    if a.dim() != 3 or b.dim() != 3 or a.shape[0] != b.shape[0] or a.shape[2] != 3 or b.shape[2] != 3:
        raise ValueError("expected clouds of shape (B, N, 3) and (B, M, 3)")
    return a.float().contiguous(), b.float().contiguous(), single


def distance_matrix(a: Tensor, b: Tensor, squared: bool = False) -> Tensor:
    """(B, N, M) pairwise distances, formed like the reference: sqrt(max(|a|^2 + |b|^2 - 2 a.b, 0))."""
    a, b, single = _batched(a, b)
    B, N, _ = a.shape
    M = b.shape[1]
    D = torch.empty(B, N, M, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().gecco_distance_matrix_f32(_ptr(a), _ptr(b), _ptr(D), B, N, M, int(squared), _stream()),
               "gecco_distance_matrix_f32")
    return D[0] if single else D


def chamfer_distance(a: Tensor, b: Tensor, squared: bool = False) -> Tensor:
    """(mean_i min_j d(a_i, b_j) + mean_j min_i d(a_i, b_j)) / 2 per sample; the distance matrix is never materialised."""
    a, b, single = _batched(a, b)
    B, N, _ = a.shape
    M = b.shape[1]
    out = torch.empty(B, device=a.device, dtype=torch.float32)
    ws = torch.empty(B * (N + M), device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().gecco_chamfer_f32(_ptr(a), _ptr(b), _ptr(out), _ptr(ws), B, N, M, int(squared), _stream()),
               "gecco_chamfer_f32")
    return out[0] if single else out


def chamfer_distance_squared(a: Tensor, b: Tensor) -> Tensor:
    return chamfer_distance(a, b, squared=True)


def scipy_emd(p1: Tensor, p2: Tensor, match: str = "l1", average: str = "l1") -> Tensor:
    """Earth mover's distance through an exact assignment (host scipy on the device's distance matrix)."""
    from scipy.optimize import linear_sum_assignment
    sq = {"l1": False, "l2": True}
    a, b, single = _batched(p1, p2)
    if a.shape[1] != b.shape[1]:
        raise ValueError("scipy_emd needs clouds of equal size")
    match_d = distance_matrix(a, b, squared=sq[match])
    avg_d = match_d if sq[average] == sq[match] else distance_matrix(a, b, squared=sq[average])
    out = []
    md, ad = match_d.cpu().numpy(), avg_d.cpu().numpy()
    for i in range(a.shape[0]):
        rows, cols = linear_sum_assignment(md[i])
        out.append(float(ad[i][rows, cols].mean()))
    res = torch.tensor(out, dtype=torch.float32, device=a.device)
    return res[0] if single else res


def sinkhorn_emd(p1: Tensor, p2: Tensor, epsilon: float = 0.01, iterations: int = 200) -> Tensor:
    """Entropic OT cost <P, C> on the squared-Euclidean cost between uniform clouds (ott's PointCloud default cost), by
    `iterations` log-domain Sinkhorn sweeps on the device.  (ott stops on a marginal-error threshold; a fixed sweep count
    keeps the call free of host reads.)"""
    a, b, single = _batched(p1, p2)
    B, N, _ = a.shape
    M = b.shape[1]
    Cm = distance_matrix(a, b, squared=True)
    f = torch.empty(B, N, device=a.device, dtype=torch.float32)
    g = torch.empty(B, M, device=a.device, dtype=torch.float32)
    rowcost = torch.empty(B, N, device=a.device, dtype=torch.float32)
    out = torch.empty(B, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().gecco_sinkhorn_f32(_ptr(Cm), _ptr(f), _ptr(g), _ptr(rowcost), _ptr(out), B, N, M, float(epsilon),
                                              int(iterations), _stream()), "gecco_sinkhorn_f32")
    return out[0] if single else out


# ----------------------------------------------------------------------------------------------- set against set
def pairwise_set_distance(a: Tensor, b: Tensor, kind: str = "chamfer", block_size: int = 16, epsilon: float = 0.1) -> Tensor:
    """(S, T) distances between EVERY cloud of a (S, N, 3) and every cloud of b (T, M, 3): gecco-jax benchmark.py:21-39
    (`batched_pairwise_distance`).  kind "chamfer" / "chamfer_squared": one HIP kernel per direction, no N x M matrix per pair;
    "emd": the entropic `sinkhorn_emd(epsilon=0.1)` of BenchmarkCallback (:73-77) on blocks of `block_size` x `block_size` pairs."""
    if a.dim() != 3 or b.dim() != 3 or a.shape[2] != 3 or b.shape[2] != 3:
        raise ValueError("expected sets of clouds of shape (S, N, 3) and (T, M, 3)")
    a, b = a.float().contiguous(), b.float().contiguous()
    S, N, _ = a.shape
    T, M, _ = b.shape
    if kind in ("chamfer", "chamfer_squared"):
        out = torch.empty(S, T, device=a.device, dtype=torch.float32)
        _lib.check(_lib.load().gecco_set_chamfer_f32(_ptr(a), _ptr(b), _ptr(out), S, T, N, M, int(kind == "chamfer_squared"), _stream()),
                   "gecco_set_chamfer_f32")
        return out
    if kind != "emd":
        raise ValueError("kind must be 'chamfer', 'chamfer_squared' or 'emd'")
    out = torch.empty(S, T, device=a.device, dtype=torch.float32)
    for s0 in range(0, S, block_size):
        for t0 in range(0, T, block_size):
            ab, bb = a[s0:s0 + block_size], b[t0:t0 + block_size]
            pa = ab[:, None].expand(-1, bb.shape[0], -1, -1).reshape(-1, N, 3)
            pb = bb[None].expand(ab.shape[0], -1, -1, -1).reshape(-1, M, 3)
            out[s0:s0 + block_size, t0:t0 + block_size] = sinkhorn_emd(pa, pb, epsilon=epsilon).reshape(ab.shape[0], bb.shape[0])
    return out


def set_metrics(ss: Tensor, sd: Tensor, dd: Tensor) -> dict[str, Tensor]:
    """1-NN accuracy, MMD and coverage of a generated set against a reference set from their (n, n) distance matrices — sample-sample,
    sample (row) - data (column), data-data — as gecco-jax benchmark.py:128-156 computes them (`_one_nn_acc`, `_mmd`, `_cov`)."""
    n = ss.shape[0]
    for m in (ss, sd, dd):
        if m.shape != (n, n):
            raise ValueError("expected three (n, n) distance matrices")
    ss, sd, dd = ss.float().contiguous(), sd.float().contiguous(), dd.float().contiguous()
    out = torch.empty(3, device=ss.device, dtype=torch.float32)
    flags = torch.empty(n, device=ss.device, dtype=torch.int32)
    _lib.check(_lib.load().gecco_set_metrics_f32(_ptr(ss), _ptr(sd), _ptr(dd), n, _ptr(out), C.c_void_p(flags.data_ptr()), _stream()),
               "gecco_set_metrics_f32")
    return {"1-nn": out[0], "mmd": out[1], "cov": out[2]}


def one_nn_accuracy(ss: Tensor, sd: Tensor, dd: Tensor) -> Tensor:
    return set_metrics(ss, sd, dd)["1-nn"]


def mmd(ss: Tensor, sd: Tensor, dd: Tensor) -> Tensor:
    return set_metrics(ss, sd, dd)["mmd"]


def cov(ss: Tensor, sd: Tensor, dd: Tensor) -> Tensor:
    return set_metrics(ss, sd, dd)["cov"]


def evaluate_sets(samples: Tensor, data: Tensor, kind: str = "chamfer") -> dict[str, Tensor]:
    """What BenchmarkCallback.__call__ reports (benchmark.py:186-215): the three distance matrices and 1-NNA / MMD / COV from them."""
    dd = pairwise_set_distance(data, data, kind)
    ss = pairwise_set_distance(samples, samples, kind)
    sd = pairwise_set_distance(samples, data, kind)
    return set_metrics(ss, sd, dd)
