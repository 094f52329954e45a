"""Tensor-level wrappers over the C ABI (include/gecco_hip.h).

PyTorch is used for device memory and the current HIP stream only; every FLOP runs in
libgecco_hip.so.  All wrappers raise on non-HIP / non-fp32 / non-contiguous tensors — there is
no fallback path.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import itertools
import os
import threading
from typing import Mapping, Sequence

import torch
from torch import Tensor

from . import _lib
from ._lib import GeccoAdaGN, GeccoLayer, GeccoLinearLift, GeccoMLP, GeccoSetTransformer, check

GN_EPS = 1e-5

# Arithmetic of the N-token GEMMs: "fp32" = exact fp32 MFMA (~1e-6 against the fp32 reference), "bf16x3" = split-bf16
# (hi + lo operands, three bf16 MFMAs per product, fp32 accumulate: ~2e-5, inside the 1e-3 parity bar, ~2x faster).
PRECISIONS = {"fp32": 0, "bf16x3": 1, "fp16": 2, "mixed": 3, "w2": 4}
_default_precision = os.environ.get("GECCO_PRECISION", "fp32")


def set_default_precision(name: str) -> None:
    """Precision used by plans built afterwards (modules rebuild theirs when this changes)."""
    global _default_precision
    if name not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
    _default_precision = name


def set_option(name: str, value: int) -> None:
    """Process-wide DEFAULT of a path switch of the library ("astat", "chain"; include/gecco_hip.h gecco_set_option): 0 / 1, or a
    negative value to return to the environment / built-in default.  Plans that did not pin the option themselves
    (`plan.set_option`) follow it; the unit operators always do.  For A/B measurements and the tests that compare the fused
    launches with the stand-alone kernels they replace."""
    _lib.check(_lib.load().gecco_set_option(name.encode(), int(value)), "set_option")
    weights_changed()   # (the options decide which weight images a forward builds)


def _option_bit(name: str) -> int:
    idx = _lib.load().gecco_option_index(name.encode())
    if idx < 0:
        raise ValueError(f"unknown option {name!r}")
    return idx


def _pin_option(table: GeccoSetTransformer, name: str, value: int) -> None:
    """Pin (0 / 1) or release (negative) one switch in a plan's own table (GeccoSetTransformer.opt_mask / opt_vals)."""
    bit = 1 << _option_bit(name)
    if value < 0:
        table.opt_mask &= ~bit
        table.opt_vals &= ~bit
    else:
        table.opt_mask |= bit
        table.opt_vals = (table.opt_vals | bit) if value else (table.opt_vals & ~bit)


def default_precision() -> str:
    return _default_precision


ACT_NONE, ACT_GAUSS, ACT_GAUSS_RAW, ACT_RELU, ACT_GELU = 0, 1, 2, 3, 4


def act_code(act_alpha: Tensor | None, normalized: bool = True, act: str | int | None = None) -> int:
    """Epilogue activation code of the C ABI: GaussianActivation (alpha given) 1 / 2, "relu" 3, none 0."""
    if act in ("relu", ACT_RELU):
        return ACT_RELU
    if act in ("gelu", ACT_GELU):
        return ACT_GELU
    if act not in (None, "gauss", "none", ACT_NONE, ACT_GAUSS, ACT_GAUSS_RAW):
        raise ValueError(f"unknown activation {act!r}")
    if act_alpha is None:
        return ACT_NONE
    return ACT_GAUSS if normalized else ACT_GAUSS_RAW


def module_act(m) -> tuple[int, Tensor | None]:
    """(code, alpha) of an activation module: GaussianActivation, nn.ReLU (the reference's default) or nn.Identity."""
    from .models.activation import GaussianActivation
    if isinstance(m, GaussianActivation):
        return (ACT_GAUSS if m.normalized else ACT_GAUSS_RAW), m.alpha
    if isinstance(m, torch.nn.ReLU):
        return ACT_RELU, None
    if isinstance(m, torch.nn.Identity):
        return ACT_NONE, None
    if isinstance(m, torch.nn.GELU) and getattr(m, "approximate", "none") == "none":
        return ACT_GELU, None
    raise NotImplementedError(f"activation {type(m).__name__} has no HIP epilogue (GaussianActivation, nn.ReLU, nn.Identity do)")


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Tensor | None) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise _lib.GeccoHipError("gecco_amd operators need tensors on the HIP device (no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.GeccoHipError(f"expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.GeccoHipError("expected a contiguous tensor")
    return C.c_void_p(t.data_ptr())


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------- unit operators
def linear(A: Tensor, W: Tensor, bias: Tensor | None = None, pro: tuple[Tensor, Tensor] | None = None,
           act_alpha: Tensor | None = None, residual: Tensor | None = None, want_stats: bool = False,
           normalized: bool = True, out: Tensor | None = None, precision: str = "fp32", act: str | int | None = None,
           w_image: Tensor | None = None, w_shape: tuple[int, int] | None = None):
    """C = residual + act((A*pro_a + pro_o) @ W^T + bias) on (B, rows, K) x (Nout, K).
    act: GaussianActivation when act_alpha is given (normalized or raw), "relu", or none.
    w_image (precision "bf16x3" / "fp16"): the READY tiled image of W (autograd.WeightImages) — W may then be None, w_shape = (Nout, K)."""
    lib = _lib.load()
    B, rows, K = A.shape
    Nout = W.shape[0] if W is not None else w_shape[0]
    assert (W.shape[1] if W is not None else w_shape[1]) == K
    out = torch.empty(B, rows, Nout, device=A.device, dtype=torch.float32) if out is None else out
    stats = None
    if want_stats:
        stats = torch.empty(B, lib.gecco_linear_row_tiles(rows), 2, Nout, device=A.device, dtype=torch.float32)
    act = act_code(act_alpha, normalized, act)
    if w_image is not None:
        assert precision in ("bf16x3", "fp16")
        wsplit, W = w_image, None
    else:
        wsplit = _ws((Nout + 127) // 128 * 128 * K * 4, A.device) if precision != "fp32" else None
    check(lib.gecco_linear_ex_f32(_ptr(A), _ptr(W), _ptr(bias), _ptr(pro[0]) if pro else None,
                                  _ptr(pro[1]) if pro else None, _ptr(act_alpha), _ptr(residual), _ptr(out), _ptr(stats),
                                  B, rows, K, Nout, act, PRECISIONS[precision],
                                  C.c_void_p(wsplit.data_ptr()) if wsplit is not None else None, _stream()),
          "gecco_linear_ex_f32")
    return (out, stats) if want_stats else out


def col_stats(x: Tensor) -> Tensor:
    lib = _lib.load()
    B, rows, Cc = x.shape
    stats = torch.empty(B, lib.gecco_stats_row_tiles(rows), 2, Cc, device=x.device, dtype=torch.float32)
    check(lib.gecco_col_stats_f32(_ptr(x), _ptr(stats), B, rows, Cc, _stream()), "gecco_col_stats_f32")
    return stats


def _adagn_struct(scale_w, scale_b, bias_w, bias_b) -> GeccoAdaGN:
    return GeccoAdaGN(_ptr(scale_w), _ptr(scale_b), _ptr(bias_w), _ptr(bias_b))


def adagn_coeffs(stats: Tensor, rows: int, t: Tensor | None, params: Sequence[Tensor] | None, G: int,
                 eps: float = GN_EPS):
    """(a, o) with AdaGN(x) = a*x + o.  params = (scale.weight, scale.bias, bias.weight, bias.bias) or
    None for a plain GroupNorm."""
    lib = _lib.load()
    B, T, _, Cc = stats.shape
    a = torch.empty(B, Cc, device=stats.device, dtype=torch.float32)
    o = torch.empty_like(a)
    st = _adagn_struct(*params) if params is not None else None
    ctx = 0 if t is None else t.shape[-1]
    check(lib.gecco_adagn_coeffs_f32(_ptr(stats), T, rows, _ptr(t), ctx, C.byref(st) if st is not None else None,
                                     _ptr(a), _ptr(o), B, Cc, G, eps, _stream()), "gecco_adagn_coeffs_f32")
    return a, o


def affine_apply(x: Tensor, a: Tensor, o: Tensor) -> Tensor:
    lib = _lib.load()
    B, rows, Cc = x.shape
    y = torch.empty_like(x)
    check(lib.gecco_affine_apply_f32(_ptr(x), _ptr(a), _ptr(o), _ptr(y), B, rows, Cc, _stream()),
          "gecco_affine_apply_f32")
    return y


def adagn(x: Tensor, t: Tensor | None, params: Sequence[Tensor] | None, G: int, eps: float = GN_EPS) -> Tensor:
    """AdaGN.forward (or GroupNormBNC when params is None) on (B, rows, C)."""
    lib = _lib.load()
    B, rows, Cc = x.shape
    y = torch.empty_like(x)
    nb = lib.gecco_adagn_workspace_bytes(B, rows, Cc)
    ws = _ws(nb, x.device)
    st = _adagn_struct(*params) if params is not None else None
    t2 = None if t is None else t.reshape(B, -1).contiguous()
    ctx = 0 if t2 is None else t2.shape[-1]
    check(lib.gecco_adagn_f32(_ptr(x), _ptr(t2), ctx, C.byref(st) if st is not None else None, _ptr(y), B, rows, Cc,
                              G, eps, C.c_void_p(ws.data_ptr()), nb, _stream()), "gecco_adagn_f32")
    return y


def _ptr16(t: Tensor) -> C.c_void_p:
    if not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous():
        raise _lib.GeccoHipError("expected a contiguous float16 HIP tensor")
    return C.c_void_p(t.data_ptr())


def _ptr_io(t: Tensor) -> C.c_void_p:
    return _ptr16(t) if t.dtype == torch.float16 else _ptr(t)


def linear_f16io(A: Tensor, W: Tensor | None, bias: Tensor | None = None, act_alpha: Tensor | None = None,
                 residual: Tensor | None = None, want_stats: bool = False, out_f16: bool = False,
                 normalized: bool = True, out: Tensor | None = None, w_image: Tensor | None = None,
                 w_shape: tuple[int, int] | None = None):
    """fp16-mode linear whose A and / or C are fp16 TENSORS (the stored intermediates of precision "fp16").
    w_image: the READY fp16 image of W (autograd.WeightImages) — W may then be None, w_shape = (Nout, K)."""
    lib = _lib.load()
    B, rows, K = A.shape
    Nout = W.shape[0] if W is not None else w_shape[0]
    a16 = A.dtype == torch.float16
    if out is None:
        out = torch.empty(B, rows, Nout, device=A.device, dtype=torch.float16 if out_f16 else torch.float32)
    stats = torch.empty(B, lib.gecco_linear_row_tiles(rows), 2, Nout, device=A.device) if want_stats else None
    act = 0 if act_alpha is None else (1 if normalized else 2)
    if w_image is not None:
        wsplit, W = w_image, None
    else:
        wsplit = _ws((Nout + 127) // 128 * 128 * K * 4, A.device)
    check(lib.gecco_linear_f16io(_ptr_io(A), _ptr(W), _ptr(bias), _ptr(act_alpha), _ptr(residual), _ptr_io(out), _ptr(stats),
                                 B, rows, K, Nout, act, int(a16), int(out.dtype == torch.float16),
                                 C.c_void_p(wsplit.data_ptr()), _stream()), "gecco_linear_f16io")
    return (out, stats) if want_stats else out


def linear_pair_f16io(A16: Tensor, W1: Tensor, b1: Tensor | None, W2: Tensor, b2: Tensor | None,
                      out: tuple[Tensor, Tensor] | None = None) -> tuple[Tensor, Tensor]:
    lib = _lib.load()
    B, rows, K = A16.shape
    n1, n2 = W1.shape[0], W2.shape[0]
    c1, c2 = out if out is not None else (torch.empty(B, rows, n1, device=A16.device, dtype=torch.float16),
                                          torch.empty(B, rows, n2, device=A16.device, dtype=torch.float16))
    wsplit = _ws(((n1 + 127) // 128 + (n2 + 127) // 128) * 128 * K * 4, A16.device)
    check(lib.gecco_linear_pair_f16io(_ptr16(A16), _ptr(W1), _ptr(b1), n1, _ptr16(c1), _ptr(W2), _ptr(b2), n2, _ptr16(c2),
                                      B, rows, K, C.c_void_p(wsplit.data_ptr()), _stream()), "gecco_linear_pair_f16io")
    return c1, c2


def linear_astat_f16(x: Tensor, pro: tuple[Tensor, Tensor] | None, W1: Tensor, b1: Tensor | None, W2: Tensor | None = None,
                     b2: Tensor | None = None, act_alpha: Tensor | None = None, normalized: bool = True,
                     out: tuple[Tensor, Tensor | None] | None = None, head_dim: int = 0, wsplit: Tensor | None = None,
                     image_ready: bool = False):
    """fp16(act(fp16(x*pa + po) @ W^T + b)) for W = W1 (| W2), fp16 outputs, one pass over x (fp16 mode).
    head_dim > 0: head-major outputs (B, Nout / head_dim, rows, head_dim) — "b n (g d) -> b g n d".
    wsplit / image_ready: caller-owned scratch holding the weight image of a previous call (kernel launch only)."""
    lib = _lib.load()
    B, rows, K = x.shape
    n1 = W1.shape[0]
    n2 = W2.shape[0] if W2 is not None else 0
    if out is not None:
        c1, c2 = out
    else:
        shape = lambda n: (B, n // head_dim, rows, head_dim) if head_dim else (B, rows, n)
        c1 = torch.empty(*shape(n1), device=x.device, dtype=torch.float16)
        c2 = torch.empty(*shape(n2), device=x.device, dtype=torch.float16) if n2 else None
    if wsplit is None:
        wsplit = _ws(((n1 + 127) // 128 + (n2 + 127) // 128) * 128 * K * 4, x.device)
    act = 0 if act_alpha is None else (1 if normalized else 2)
    check(lib.gecco_linear_astat_f16(_ptr(x), _ptr(pro[0]) if pro else None, _ptr(pro[1]) if pro else None,
                                     None if image_ready else _ptr(W1), _ptr(b1),
                                     n1, _ptr16(c1), None if image_ready else _ptr(W2), _ptr(b2), n2, _ptr16(c2) if c2 is not None else None,
                                     _ptr(act_alpha), act, B, rows, K, head_dim, C.c_void_p(wsplit.data_ptr()), _stream()),
          "gecco_linear_astat_f16")
    return (c1, c2) if c2 is not None else c1


def linear_kvq_f16(x: Tensor, pro: tuple[Tensor, Tensor] | None, W1: Tensor, b1: Tensor | None, W2: Tensor | None = None,
                   b2: Tensor | None = None, lo: tuple[int, int] = (0, 0), head_dim: int = 0, wsplit: Tensor | None = None,
                   image_ready: bool = False, y16: Tensor | None = None):
    """fp16(fp16(x*pa + po) @ W^T + b) for W = W1 (| W2), fp16 outputs, the columns lo[0] .. lo[1] of W1 with two-term weights
    (fp16 + fp8 second term): kv_proj | q_proj of the mixed mode.  head_dim > 0: head-major outputs."""
    lib = _lib.load()
    B, rows, K = x.shape
    n1 = W1.shape[0]
    n2 = W2.shape[0] if W2 is not None else 0
    shape = lambda n: (B, n // head_dim, rows, head_dim) if head_dim else (B, rows, n)
    c1 = torch.empty(*shape(n1), device=x.device, dtype=torch.float16)
    c2 = torch.empty(*shape(n2), device=x.device, dtype=torch.float16) if n2 else None
    if wsplit is None:
        wsplit = _ws((n1 + n2) * K * 2 + (lo[1] - lo[0]) * K, x.device)
    # y16: also store fp16(x * pa + po) (B, rows, K), the operand the kernel forms (the training path's weight gradients read it back)
    check(lib.gecco_linear_kvq_y16_f16(_ptr(x), _ptr(pro[0]) if pro else None, _ptr(pro[1]) if pro else None,
                                       None if image_ready else _ptr(W1), _ptr(b1), n1, _ptr16(c1),
                                       None if image_ready else _ptr(W2), _ptr(b2), n2, _ptr16(c2) if c2 is not None else None,
                                       _ptr16(y16) if y16 is not None else None,
                                       B, rows, K, head_dim, lo[0], lo[1], C.c_void_p(wsplit.data_ptr()), _stream()), "gecco_linear_kvq_f16")
    return (c1, c2) if c2 is not None else c1


def linear_h8_img(x: Tensor, pro: tuple[Tensor, Tensor] | None, W: Tensor, b: Tensor | None, act_alpha: Tensor | None = None,
                  normalized: bool = True, act: str | int | None = None, wsplit: Tensor | None = None,
                  image_ready: bool = False, out: Tensor | None = None, kind: int = 1) -> Tensor:
    """act((x*pa + po) @ W^T + b) in fp16 + fp8-cross-term arithmetic (mixed mode's mlp.0), returned as an image the next linear
    loads into registers.  kind 1: tiled split image (B, rows / 128, Nout / 16, 2, 128, 16) of bf16 bit patterns (int16) — hi plane,
    lo plane (`decode_split_image`); kind 2: h8 activation image (B, rows / 128, Nout / 64, 24576) bytes (`decode_h8_image`)."""
    lib = _lib.load()
    B, rows, K = x.shape
    n = W.shape[0]
    if out is None:
        out = (torch.empty(B, rows // 128, n // 16, 2, 128, 16, device=x.device, dtype=torch.int16) if kind == 1 else
               torch.empty(B, rows // 128, n // 64, 24576, device=x.device, dtype=torch.uint8))
    if wsplit is None:
        wsplit = _ws(n * K * 4, x.device)
    check(lib.gecco_linear_h8_img_f32(_ptr(x), _ptr(pro[0]) if pro else None, _ptr(pro[1]) if pro else None,
                                      None if image_ready else _ptr(W), _ptr(b), _ptr(act_alpha),
                                      act_code(act_alpha, normalized, act), C.c_void_p(out.data_ptr()), kind, B, rows, K, n,
                                      C.c_void_p(wsplit.data_ptr()), _stream()), "gecco_linear_h8_img_f32")
    return out


def linear_h8_areg(a_img: Tensor, W: Tensor, b: Tensor | None = None, residual: Tensor | None = None, want_stats: bool = False,
                   out: Tensor | None = None, wsplit: Tensor | None = None, image_ready: bool = False):
    """residual + A @ W^T + b with A an h8 activation image (B, rows / 128, K / 64, 24576) bytes: mlp.2 / out_proj of the mixed mode."""
    lib = _lib.load()
    B, T, G = a_img.shape[:3]
    rows, K, n = T * 128, G * 64, W.shape[0]
    if out is None:
        out = torch.empty(B, rows, n, device=a_img.device, dtype=torch.float32)
    stats = torch.empty(B, rows // 128, 2, n, device=a_img.device, dtype=torch.float32) if want_stats else None
    if wsplit is None:
        wsplit = _ws((n + 127) // 128 * 128 * K * 4, a_img.device)
    check(lib.gecco_linear_h8_areg_f32(C.c_void_p(a_img.data_ptr()), None if image_ready else _ptr(W), _ptr(b), _ptr(residual), _ptr(out),
                                       _ptr(stats), B, rows, K, n, C.c_void_p(wsplit.data_ptr()), _stream()), "gecco_linear_h8_areg_f32")
    return (out, stats) if want_stats else out


def decode_split_image(img: Tensor) -> Tensor:
    """(B, rows / 128, Nout / 16, 2, 128, 16) int16 tiled split image -> the (B, rows, Nout) fp32 tensor it represents
    (hi + lo; GemmArgs::c_img layout: the 8-element half of a row is swapped when (row >> 3) & 1)."""
    Bn, T, KT = img.shape[:3]
    f = (img.to(torch.int32) << 16).view(torch.float32)             # bf16 bits -> fp32
    v = f[:, :, :, 0] + f[:, :, :, 1]                                # (B, T, KT, 128, 16)
    rows = torch.arange(128, device=img.device)
    swap = ((rows >> 3) & 1).bool()
    v = torch.where(swap[None, None, None, :, None], torch.cat([v[..., 8:], v[..., :8]], dim=-1), v)
    return v.permute(0, 1, 3, 2, 4).reshape(Bn, T * 128, KT * 16)


def decode_h8_image(img: Tensor) -> Tensor:
    """(B, rows / 128, K / 64, 24576) uint8 h8 activation image -> the (B, rows, K) float64 tensor hi + 2^-11 lo it represents
    (csrc/h8_scales.h).
    hi: [rt 4][sub 2][c 2][lane 64][8 fp16], column 32 sub + 16 (lane >> 5) + 8 c + e; lo: [rt 4][t 2][lane 64][16 fp8 e4m3],
    column 32 t + 16 (lane >> 5) + e; row 32 rt + (lane & 31)."""
    Bn, T, G = img.shape[:3]
    hi = img[..., :16384].contiguous().view(torch.float16).reshape(Bn, T, G, 4, 2, 2, 2, 32, 8).double()   # rt sub c h r e
    lo = img[..., 16384:].contiguous().view(torch.float8_e4m3fn).reshape(Bn, T, G, 4, 2, 2, 32, 16).double()   # rt t h r e
    hi = hi.permute(0, 1, 3, 7, 2, 4, 6, 5, 8)        # B T rt r G sub h c e
    lo = lo.permute(0, 1, 3, 6, 2, 4, 5, 7)           # B T rt r G t h e
    return hi.reshape(Bn, T * 128, G * 64) + lo.reshape(Bn, T * 128, G * 64) * 2.0 ** -11


def mlp_fused_f16(x: Tensor, pro: tuple[Tensor, Tensor], W0: Tensor, b0: Tensor | None, W2: Tensor, b2: Tensor | None,
                  act_alpha: Tensor | None = None, normalized: bool = True, want_stats: bool = False,
                  wsplit: Tensor | None = None, image_ready: bool = False, stats: Tensor | None = None):
    """x += mlp.2(act(mlp.0(x*pa + po))) in place (fp16 mode, one launch); returns (x, stats | None).
    wsplit / image_ready: caller-owned scratch holding the weight image of a previous call (kernel launch only)."""
    lib = _lib.load()
    B, rows, Cc = x.shape
    width = W0.shape[0]
    if stats is None and want_stats:
        stats = torch.empty(B, rows // 128, 2, Cc, device=x.device, dtype=torch.float32)
    if wsplit is None:
        wsplit = _ws(4 * Cc * width, x.device)
    act = 0 if act_alpha is None else (1 if normalized else 2)
    check(lib.gecco_mlp_fused_f16(_ptr(x), _ptr(pro[0]), _ptr(pro[1]), None if image_ready else _ptr(W0), _ptr(b0),
                                  None if image_ready else _ptr(W2), _ptr(b2), _ptr(act_alpha),
                                  act, _ptr(stats), B, rows, Cc, width, C.c_void_p(wsplit.data_ptr()), _stream()),
          "gecco_mlp_fused_f16")
    return x, stats


def unpool_outproj_f16(x: Tensor, q16: Tensor, kvh: Tensor, W: Tensor, bias: Tensor | None, H: int, want_stats: bool = False,
                       wsplit: Tensor | None = None, image_ready: bool = False, stats: Tensor | None = None):
    """x += MHA(q, inducer k | v) @ W^T + bias in place (fp16 mode, one launch); q16 head-major (B, H, N, hd).
    Returns (x, stats | None).  wsplit / image_ready: scratch holding the weight image of a previous call."""
    lib = _lib.load()
    B, rows, Cc = x.shape
    if stats is None and want_stats:
        stats = torch.empty(B, rows // 128, 2, Cc, device=x.device, dtype=torch.float32)
    if wsplit is None:
        wsplit = _ws(2 * Cc * Cc, x.device)
    check(lib.gecco_unpool_outproj_f16(_ptr(x), _ptr16(q16), _ptr(kvh), None if image_ready else _ptr(W), _ptr(bias), _ptr(stats), B, rows, Cc, H,
                                       C.c_void_p(wsplit.data_ptr()), _stream()), "gecco_unpool_outproj_f16")
    return x, stats


def mlp_fused_w(x: Tensor, pro: tuple[Tensor, Tensor], W0: Tensor, b0: Tensor | None, W2: Tensor, b2: Tensor | None,
                act_alpha: Tensor | None = None, normalized: bool = True, act: str | int | None = None, want_stats: bool = False,
                wsplit: Tensor | None = None, image_ready: bool = False, stats: Tensor | None = None, out: Tensor | None = None,
                dbg_u: Tensor | None = None):
    """out (default: x, in place) = x + mlp.2(act(mlp.0(x*pa + po))) ("w2" mode, one launch, the hidden layer kept in registers);
    returns (out, stats | None).  wsplit / image_ready: caller-owned scratch holding the weight stream of a previous call."""
    lib = _lib.load()
    B, rows, Cc = x.shape
    width = W0.shape[0]
    if out is None:
        out = x
    if stats is None and want_stats:
        stats = torch.empty(B, rows // 128, 2, Cc, device=x.device, dtype=torch.float32)
    if wsplit is None:
        wsplit = _ws(lib.gecco_mlp_fused_w_wsplit_bytes(Cc, width), x.device)
    check(lib.gecco_mlp_fused_w(_ptr(x), _ptr(out), _ptr(pro[0]), _ptr(pro[1]), None if image_ready else _ptr(W0), _ptr(b0),
                                None if image_ready else _ptr(W2), _ptr(b2), _ptr(act_alpha), act_code(act_alpha, normalized, act),
                                _ptr(stats), B, rows, Cc, width, C.c_void_p(wsplit.data_ptr()), _ptr(dbg_u), _stream()), "gecco_mlp_fused_w")
    return out, stats


def unpool_outproj_h8(x: Tensor, q16: Tensor, kvh: Tensor, W: Tensor, bias: Tensor | None, H: int, want_stats: bool = False,
                      wsplit: Tensor | None = None, image_ready: bool = False, stats: Tensor | None = None):
    """x += MHA(q, inducer k | v) @ W^T + bias in place (mixed mode, one launch: fp16 attention, h8 out_proj); q16 head-major
    (B, H, N, hd).  Returns (x, stats | None).  wsplit / image_ready: scratch holding the weight image of a previous call."""
    lib = _lib.load()
    B, rows, Cc = x.shape
    if stats is None and want_stats:
        stats = torch.empty(B, rows // 128, 2, Cc, device=x.device, dtype=torch.float32)
    if wsplit is None:
        wsplit = _ws(lib.gecco_unpool_outproj_h8_wsplit_bytes(B, Cc, H), x.device)
    check(lib.gecco_unpool_outproj_h8(_ptr(x), _ptr16(q16), _ptr(kvh), None if image_ready else _ptr(W), _ptr(bias), _ptr(stats), B, rows, Cc, H,
                                      C.c_void_p(wsplit.data_ptr()), _stream()), "gecco_unpool_outproj_h8")
    return x, stats


def unpool_attn_h8img(q16: Tensor, kvh: Tensor, H: int) -> Tensor:
    """The mixed mode's unpool attention: head-major fp16 q16 (B, H, N, hd) -> the h8 activation image (B, N / 128, C / 64, 24576) bytes
    that `linear_h8_areg` consumes (`decode_h8_image` turns it back into (B, N, C) fp32)."""
    lib = _lib.load()
    B, _, N, hd = q16.shape
    Cc = H * hd
    out = torch.empty(B, N // 128, Cc // 64, 24576, device=q16.device, dtype=torch.uint8)
    check(lib.gecco_unpool_attn_h8img(_ptr16(q16), _ptr(kvh), C.c_void_p(out.data_ptr()), B, N, Cc, H, _stream()), "gecco_unpool_attn_h8img")
    return out


def affine_cast_f16(x: Tensor, a: Tensor, o: Tensor, out: Tensor | None = None) -> Tensor:
    """fp16(a[b, c] * x[b, m, c] + o[b, c]): the AdaGN apply stored as the fp16 GEMM operand."""
    lib = _lib.load()
    B, rows, Cc = x.shape
    out = torch.empty(B, rows, Cc, device=x.device, dtype=torch.float16) if out is None else out
    check(lib.gecco_affine_cast_f16(_ptr(x), _ptr(a), _ptr(o), _ptr16(out), B, rows, Cc, _stream()), "gecco_affine_cast_f16")
    return out


def pool_attn_f16in(KV16: Tensor, inducers: Tensor, H: int, head_major: bool = False) -> Tensor:
    """KV16: (B, N, 2C) fp16, or head-major (B, 2H, N, hd) = K heads then V heads."""
    lib = _lib.load()
    if head_major:
        B, _, N, hd = KV16.shape
        Cc = H * hd
    else:
        B, N, C2 = KV16.shape
        Cc = C2 // 2
    I = inducers.shape[-2]
    merged = torch.empty(B, I, Cc, device=KV16.device, dtype=torch.float32)
    nb = lib.gecco_pool_attn_workspace_bytes(B, N, Cc, H, I)
    ws = _ws(nb, KV16.device)
    check(lib.gecco_pool_attn_f16in(_ptr16(KV16), _ptr(inducers), _ptr(merged), B, N, Cc, H, I, int(head_major),
                                    C.c_void_p(ws.data_ptr()), nb, _stream()), "gecco_pool_attn_f16in")
    return merged


def unpool_attn_f16io(q16: Tensor, kvh: Tensor, H: int, out: Tensor | None = None, head_major: bool = False) -> Tensor:
    """q16: (B, N, C) fp16, or head-major (B, H, N, hd); out is (B, N, C) fp16 either way."""
    lib = _lib.load()
    if head_major:
        B, _, N, hd = q16.shape
        Cc = H * hd
    else:
        B, N, Cc = q16.shape
    out = torch.empty(B, N, Cc, device=q16.device, dtype=torch.float16) if out is None else out
    check(lib.gecco_unpool_attn_f16io(_ptr16(q16), _ptr(kvh), _ptr16(out), B, N, Cc, H, kvh.shape[1], int(head_major),
                                      _stream()), "gecco_unpool_attn_f16io")
    return out


def linear_pair(A: Tensor, W1: Tensor, b1: Tensor | None, W2: Tensor, b2: Tensor | None,
                pro: tuple[Tensor, Tensor] | None = None, out: tuple[Tensor, Tensor] | None = None,
                precision: str = "fp32", w_image: Tensor | None = None) -> tuple[Tensor, Tensor]:
    """(A' @ W1^T + b1, A' @ W2^T + b2) with A' = A*pro_a + pro_o, one launch (A read once).
    w_image (precision "bf16x3" / "fp16"): the READY images of W1 | W2 (autograd.WeightImages): launch only."""
    lib = _lib.load()
    B, rows, K = A.shape
    n1, n2 = W1.shape[0], W2.shape[0]
    c1, c2 = out if out is not None else (torch.empty(B, rows, n1, device=A.device, dtype=torch.float32),
                                          torch.empty(B, rows, n2, device=A.device, dtype=torch.float32))
    if w_image is not None:
        assert precision in ("bf16x3", "fp16")
        wsplit, W1, W2 = w_image, None, None
    else:
        wsplit = _ws(((n1 + 127) // 128 + (n2 + 127) // 128) * 128 * K * 4, A.device) if precision != "fp32" else None
    check(lib.gecco_linear_pair_f32(_ptr(A), _ptr(W1), _ptr(b1), n1, _ptr(c1), _ptr(W2), _ptr(b2), n2, _ptr(c2),
                                    _ptr(pro[0]) if pro else None, _ptr(pro[1]) if pro else None, B, rows, K,
                                    PRECISIONS[precision], C.c_void_p(wsplit.data_ptr()) if wsplit is not None else None,
                                    _stream()), "gecco_linear_pair_f32")
    return c1, c2


def pool_attn(KV: Tensor, inducers: Tensor, H: int, precision: str = "fp32") -> Tensor:
    """AttentionPool core: KV (B, N, 2C), inducers (1, H, I, hd) -> (B, I, C) merged heads (before out_proj)."""
    lib = _lib.load()
    pr = PRECISIONS[precision]
    B, N, C2 = KV.shape
    Cc = C2 // 2
    I = inducers.shape[-2]
    merged = torch.empty(B, I, Cc, device=KV.device, dtype=torch.float32)
    nb = lib.gecco_pool_attn_workspace_bytes(B, N, Cc, H, I)
    ws = _ws(nb, KV.device)
    check(lib.gecco_pool_attn_ex_f32(_ptr(KV), _ptr(inducers), _ptr(merged), B, N, Cc, H, I, pr,
                                     C.c_void_p(ws.data_ptr()), nb, _stream()), "gecco_pool_attn_ex_f32")
    return merged


def unpool_attn(q: Tensor, kvh: Tensor, H: int, precision: str = "fp32") -> Tensor:
    lib = _lib.load()
    pr = PRECISIONS[precision]
    B, N, Cc = q.shape
    I = kvh.shape[1]
    out = torch.empty_like(q)
    check(lib.gecco_unpool_attn_ex_f32(_ptr(q), _ptr(kvh), _ptr(out), B, N, Cc, H, I, pr, _stream()),
          "gecco_unpool_attn_ex_f32")
    return out


def edm_coeffs(sigma: Tensor, sigma_data: float = 1.0) -> Tensor:
    lib = _lib.load()
    B = sigma.numel()
    coef = torch.empty(5 * B, device=sigma.device, dtype=torch.float32)
    check(lib.gecco_edm_coeffs_f32(_ptr(sigma), sigma_data, _ptr(coef), B, _stream()), "gecco_edm_coeffs_f32")
    return coef


def lift(x: Tensor, coef: Tensor | None, W: Tensor, bias: Tensor, want_stats: bool = False):
    lib = _lib.load()
    B, N, three = x.shape
    assert three == 3 and W.shape[1] == 3
    Cc = W.shape[0]
    out = torch.empty(B, N, Cc, device=x.device, dtype=torch.float32)
    stats = torch.empty(B, lib.gecco_stats_row_tiles(N), 2, Cc, device=x.device, dtype=torch.float32) if want_stats else None
    check(lib.gecco_lift_f32(_ptr(x), _ptr(coef), _ptr(W), _ptr(bias), _ptr(out), _ptr(stats), B, N, Cc, _stream()),
          "gecco_lift_f32")
    return (out, stats) if want_stats else out


def lower_edm(feat: Tensor, x: Tensor | None, coef: Tensor | None, W: Tensor, bias: Tensor,
              gn: tuple[Tensor, Tensor] | None = None, want_raw: bool = False, eps: float = GN_EPS):
    lib = _lib.load()
    B, N, Cc = feat.shape
    out = torch.empty(B, N, 3, device=feat.device, dtype=torch.float32)
    raw = torch.empty_like(out) if want_raw else None
    check(lib.gecco_lower_edm_f32(_ptr(feat), _ptr(x), _ptr(coef), _ptr(W), _ptr(bias), _ptr(gn[0]) if gn else None,
                                  _ptr(gn[1]) if gn else None, _ptr(out), _ptr(raw), B, N, Cc, eps, _stream()),
          "gecco_lower_edm_f32")
    return (out, raw) if want_raw else out


# ------------------------------------------------------------------------------- parameter tables
def _adagn_from(p: Mapping[str, Tensor], pre: str) -> GeccoAdaGN:
    return _adagn_struct(p[pre + "scale.weight"], p[pre + "scale.bias"], p[pre + "bias.weight"], p[pre + "bias.bias"])


def _mlp_from(p: Mapping[str, Tensor], pre: str) -> GeccoMLP:
    # "1.alpha" exists only with GaussianActivation (nn.ReLU / nn.Identity have no parameters)
    return GeccoMLP(_ptr(p[pre + "0.weight"]), _ptr(p[pre + "0.bias"]), _ptr(p.get(pre + "1.alpha")),
                    _ptr(p[pre + "2.weight"]), _ptr(p[pre + "2.bias"]))


def layer_table(p: Mapping[str, Tensor], pre: str) -> GeccoLayer:
    """One BroadcastingLayer's device pointers, keyed like the reference state dict."""
    return GeccoLayer(
        _adagn_from(p, pre + "broadcast_norm."), _ptr(p[pre + "broadcast.pool.inducers"]),
        _ptr(p[pre + "broadcast.pool.kv_proj.weight"]), _ptr(p[pre + "broadcast.pool.out_proj.weight"]),
        _adagn_from(p, pre + "broadcast.norm_1."), _mlp_from(p, pre + "broadcast.mlp."),
        _adagn_from(p, pre + "broadcast.norm_2."), _ptr(p[pre + "broadcast.unpool.in_proj_weight"]),
        _ptr(p[pre + "broadcast.unpool.in_proj_bias"]), _ptr(p[pre + "broadcast.unpool.out_proj.weight"]),
        _ptr(p[pre + "broadcast.unpool.out_proj.bias"]), _adagn_from(p, pre + "mlp_norm."), _mlp_from(p, pre + "mlp."))


class SetTransformerPlan:
    """Host-side parameter table + workspace cache for gecco_set_transformer_fwd_f32.  Holds references to the
    parameter tensors so the raw pointers stay valid; rebuild it if parameters are re-allocated."""

    def __init__(self, p: Mapping[str, Tensor], pre: str, H: int, I: int = 64, G: int = 32, normalized: bool = True,
                 precision: str | None = None, act: int | None = None, options: Mapping[str, int] | None = None):
        self.lib = _lib.load()
        self.precision = precision or _default_precision
        if self.precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        self.p = p  # keep tensors alive
        L = 0
        while f"{pre}layers.{L}.mlp.0.weight" in p:
            L += 1
        if L == 0:
            raise KeyError(f"no layers under prefix {pre!r}")
        w0 = p[f"{pre}layers.0.mlp.0.weight"]
        self.width, self.C = w0.shape
        self.L, self.H, self.I, self.G = L, H, I, G
        self.ctx_dim = p[f"{pre}layers.0.mlp_norm.scale.weight"].shape[1]
        self.device = w0.device
        self._layers = (GeccoLayer * L)(*[layer_table(p, f"{pre}layers.{i}.") for i in range(L)])
        if act is None:   # GaussianActivation when the state dict carries its alpha, else the reference's default ReLU
            act = (ACT_GAUSS if normalized else ACT_GAUSS_RAW) if f"{pre}layers.0.mlp.1.alpha" in p else ACT_RELU
        self.act = act
        self.table = GeccoSetTransformer(L, self.C, H, I, self.ctx_dim, G, self.width, act,
                                         PRECISIONS[self.precision], 0, 0, 0, self._layers)
        for name, value in (options or {}).items():
            _pin_option(self.table, name, value)
        self._ws: dict[tuple[int, int], Tensor] = {}
        self.images = _ImageState()

    def set_option(self, name: str, value: int) -> None:
        """Pin a path switch for THIS plan (0 / 1; negative: follow the process-wide default again)."""
        _pin_option(self.table, name, value)
        self.images.changed()

    def workspace(self, B: int, N: int) -> Tensor:
        key = (B, N)
        if key not in self._ws:
            self._ws[key] = _ws(self.lib.gecco_set_transformer_workspace_bytes(C.byref(self.table), B, N), self.device)
        return self._ws[key]

    @staticmethod
    def _ptr_array(ts: Sequence[Tensor | None] | None, L: int):
        if ts is None:
            return None
        assert len(ts) == L
        return (C.c_void_p * L)(*[(t.data_ptr() if t is not None else 0) for t in ts])

    def forward_(self, x: Tensor, t: Tensor, stats: Tensor | None = None, hs: Sequence[Tensor | None] | None = None,
                 return_h: bool = False, want_stats_out: bool = False):
        """In place on x (B, N, C).  Returns (x, hs_out | None, stats_out | None)."""
        B, N, Cc = x.shape
        assert Cc == self.C
        ws = self.workspace(B, N)
        t2 = t.reshape(B, -1).contiguous()
        h_out = [torch.empty(B, self.I, Cc, device=x.device, dtype=torch.float32) for _ in range(self.L)] if return_h else None
        if return_h and hs is not None:  # layers with a cached h just hand it back (reference semantics)
            h_out = [h if h is not None else o for h, o in zip(hs, h_out)]
        so = torch.empty(B, self.lib.gecco_linear_row_tiles(N), 2, Cc, device=x.device, dtype=torch.float32) if want_stats_out else None
        hin = self._ptr_array(hs, self.L)
        hout = self._ptr_array([None if (hs is not None and hs[i] is not None) else h_out[i] for i in range(self.L)], self.L) if return_h else None
        check(self.lib.gecco_set_transformer_fwd_f32(
            C.byref(self.table), _ptr(x), _ptr(t2), _ptr(stats), 0 if stats is None else stats.shape[1],
            hin, hout, _ptr(so), B, N, C.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "gecco_set_transformer_fwd_f32")
        return x, h_out, so


# ------------------------------------------------------------------------------- two streams per evaluation
# Every kernel of an evaluation keeps the matrix pipe 33 - 50 % busy on its own (profiles/r02za_forward_pmc_summary.txt): the
# latency of one barrier per K-step.  The samples of a batch are independent (the forward is bit-identical for a sample
# whatever batch it sits in), so an evaluation runs as TWO half batches on two HIP streams — two C calls, each with its own
# workspace — whose kernels share the CUs: one half's latency-bound stretches (the 64-inducer chain, a kernel's tail) sit
# under the other's GEMMs.  C2: 6.64 -> 6.38 ms per evaluation, outputs identical to the bit.  Inside a hipGraph capture the
# fork / join (event waits) is captured with it.  GECCO_FWD_STREAMS=1 keeps one stream.
_FWD_SIDE = {"streams": []}

# ------------------------------------------------------------------------------- weight images across evaluations
# A forward rebuilds the images of the weights it streams (fp16 / fp6 tiles in the kernels' consumption order: ~0.14 ms of a
# 4.4 ms C2 evaluation) because weights may change between calls.  Inside `frozen_weights()` the caller vouches that they do
# not — a sampler's 255 evaluations, a serving loop between two weight updates — and the fused plans build the images once
# per scope and workspace and hand `images_ready = 1` to the library afterwards.  Entering the outermost scope, `set_option`
# and every optimizer step (`weights_changed()`) start a new generation: nothing built earlier is trusted.
#
# State: the scope depth is per host THREAD (`threading.local`), the built-image tokens are per PLAN and workspace (`_ImageState`), and a
# plan has its own generation next to the process-wide one — two models, or two threads, cannot alias each other's scopes; the
# library reads `images_ready` from a per-call COPY of the plan's table, never from shared state.
_GENERATION = itertools.count(1)          # (next() on a count is atomic under the GIL)
_generation = [next(_GENERATION)]         # process-wide: bumped by weights_changed() / set_option() / an outermost frozen_weights()


class _Scope(threading.local):
    depth = 0


_SCOPE = _Scope()


def weights_changed() -> None:
    """Tell EVERY plan that weight values (or the process-wide path options) changed by a route no tensor version counter sees
    (an optimizer step, an EMA swap, a state-dict load into the same storage).  A plan's own `images.changed()` is the narrow form."""
    _generation[0] = next(_GENERATION)


@contextlib.contextmanager
def frozen_weights(*plans):
    """Inside the scope the caller vouches that weights do not change: plans build their weight images once per workspace.
    Without arguments the scope covers every plan used by THIS thread; with plans (LinearLiftPlan / RayNetworkPlan /
    SetTransformerPlan, or modules' `.plan`) only those."""
    states = [st for p in plans for st in _image_states_of(p)]
    if plans:
        for st in states:
            st.enter()
        try:
            yield
        finally:
            for st in states:
                st.leave()
        return
    if _SCOPE.depth == 0:
        _generation[0] = next(_GENERATION)
    _SCOPE.depth += 1
    try:
        yield
    finally:
        _SCOPE.depth -= 1


def _image_states_of(obj) -> list:
    """The built-image records behind `obj`: a plan's own, an `_ImageState`, or — for an nn.Module (a `Diffusion`, a `SetTransformer`) — those
    of every plan its sub-modules have built so far (a module builds its plan at its first evaluation: freeze after a warm-up call)."""
    if isinstance(obj, _ImageState):
        return [obj]
    if hasattr(obj, "images"):
        return [obj.images]
    if isinstance(obj, torch.nn.Module):
        found = []
        for m in obj.modules():
            plan = getattr(getattr(m, "_cache", None), "plan", None)
            if plan is not None and hasattr(plan, "images") and plan.images not in found:
                found.append(plan.images)
        return found
    raise TypeError(f"frozen_weights: a plan or a module expected, got {type(obj).__name__}")


class _ImageState:
    """One plan's record of which workspaces hold valid weight images (and its own frozen scope / generation)."""

    def __init__(self):
        self.depth = 0
        self.generation = 0
        self.tokens: dict[tuple, tuple] = {}

    def enter(self) -> None:
        if self.depth == 0:
            self.generation = next(_GENERATION)
        self.depth += 1

    def leave(self) -> None:
        self.depth -= 1

    def changed(self) -> None:
        self.generation = next(_GENERATION)
        self.tokens.clear()

    def token(self, cached: bool):
        """What a workspace's images are built under now, or None outside every frozen scope (then each forward rebuilds)."""
        if (self.depth == 0 and _SCOPE.depth == 0) or os.environ.get("GECCO_FROZEN_IMAGES", "1") == "0":
            return None
        return (_generation[0], self.generation, self.depth > 0, bool(cached))

    def ready(self, key, tok) -> int:
        """images_ready for the call about to be issued on workspace `key`."""
        return int(tok is not None and self.tokens.get(key) == tok)

    def built(self, key, tok, was_ready: int) -> None:
        """Record a SUCCESSFUL forward on `key` (call after check()).  A forward that was only captured into a graph built nothing
        yet: its images exist once the graph replays, so nothing is recorded and later eager calls rebuild (the graph carries its
        own build launches)."""
        if tok is None or (not was_ready and torch.cuda.is_initialized() and torch.cuda.is_current_stream_capturing()):
            self.tokens.pop(key, None)
        else:
            self.tokens[key] = tok

    def failed(self, key) -> None:
        self.tokens.pop(key, None)


def _fwd_parts(B: int, N: int) -> int:
    n = int(os.environ.get("GECCO_FWD_STREAMS", "2"))
    if n <= 1 or B < 2 * n or B * N < 32768:
        return 1
    return n


def _two_stream_halves(B: int, call, tensors, parts: int = 2) -> None:
    """call(lo, hi, idx) issues the evaluation of samples [lo, hi) on the current stream; part 0 runs on the caller's stream,
    the others on side streams.  `tensors`: what the side streams touch (allocated on the caller's stream)."""
    while len(_FWD_SIDE["streams"]) < parts - 1:
        _FWD_SIDE["streams"].append(torch.cuda.Stream())
    sides, main = _FWD_SIDE["streams"][:parts - 1], torch.cuda.current_stream()
    cuts = [B * i // parts for i in range(parts + 1)]
    # (the calls pin "mlpwshare" in their own copy of the table: launches that fill their CUs leave room for the other stream's kernels)
    for i, side in enumerate(sides, start=1):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            call(cuts[i], cuts[i + 1], i)
    call(cuts[0], cuts[1], 0)
    for side in sides:
        main.wait_stream(side)
    if not torch.cuda.is_current_stream_capturing():   # (a captured graph owns its memory: nothing to tell the allocator)
        for t in tensors:
            if t is not None:
                for side in sides:
                    t.record_stream(side)


class LinearLiftPlan:
    """EDMPrecond(LinearLift(SetTransformer)) = the unconditional Diffusion.forward, one C call."""

    def __init__(self, p: Mapping[str, Tensor], H: int, I: int = 64, pre: str = "", sigma_data: float = 1.0,
                 precision: str | None = None, act: int | None = None, options: Mapping[str, int] | None = None):
        self.st = SetTransformerPlan(p, pre + "inner.", H, I, precision=precision, act=act, options=options)
        self.p = p
        self.lib = self.st.lib
        self.table = GeccoLinearLift(self.st.table, _ptr(p[pre + "lift.weight"]), _ptr(p[pre + "lift.bias"]),
                                     _ptr(p[pre + "lower.1.weight"]), _ptr(p[pre + "lower.1.bias"]), sigma_data)
        self._ws: dict[tuple[int, int], Tensor] = {}
        self.images = self.st.images   # workspace key -> the token its weight images were built under (frozen_weights)

    def set_option(self, name: str, value: int) -> None:
        """Pin a path switch for THIS plan (0 / 1; negative: follow the process-wide default again)."""
        _pin_option(self.table.inner, name, value)
        self.st.set_option(name, value)

    def workspace(self, B: int, N: int, idx: int = 0) -> Tensor:
        key = (B, N, idx)
        if key not in self._ws:
            self._ws[key] = _ws(self.lib.gecco_linear_lift_workspace_bytes(C.byref(self.table), B, N), self.st.device)
        return self._ws[key]

    def forward(self, x: Tensor, sigma: Tensor, return_raw: bool = False, cache: Sequence[Tensor] | None = None,
                do_cache: bool = False, out: Tensor | None = None):
        B, N, _ = x.shape
        den = torch.empty_like(x) if out is None else out
        raw = torch.empty_like(x) if return_raw else None
        L = self.st.L
        h_out = [torch.empty(B, self.st.I, self.st.C, device=x.device, dtype=torch.float32) for _ in range(L)] if do_cache else None

        def call(lo, hi, idx):
            ws = self.workspace(hi - lo, N, idx)
            key, tok = (hi - lo, N, idx), self.images.token(cache is not None)
            tbl = GeccoLinearLift.from_buffer_copy(self.table)    # this call's own copy: nothing another call / thread can alias
            ready = tbl.inner.images_ready = self.images.ready(key, tok)
            if parts > 1:
                _pin_option(tbl.inner, "mlpwshare", 1)
            cut = (lambda ts: None if ts is None else [None if t is None else t[lo:hi] for t in ts])
            try:
                check(self.lib.gecco_linear_lift_fwd_f32(
                    C.byref(tbl), _ptr(x[lo:hi]), _ptr(sigma[lo:hi]), _ptr(den[lo:hi]), _ptr(None if raw is None else raw[lo:hi]),
                    self.st._ptr_array(cut(cache), L), self.st._ptr_array(cut(h_out), L), hi - lo, N, C.c_void_p(ws.data_ptr()),
                    ws.numel(), _stream()), "gecco_linear_lift_fwd_f32")
            except BaseException:
                self.images.failed(key)
                raise
            self.images.built(key, tok, ready)
        parts = _fwd_parts(B, N) if B else 1
        if B == 0:
            pass   # an empty batch (a rank that owns no cloud): empty results, nothing to launch
        elif parts > 1:
            _two_stream_halves(B, call, [x, sigma, den, raw, *(cache or []), *(h_out or [])], parts)
        else:
            call(0, B, 0)
        res = (den, raw) if return_raw else den
        return (res, h_out) if do_cache else res


# ------------------------------------------------------------------------------- conditional path
def nchw_to_nhwc(f: Tensor) -> Tensor:
    """(B, C, H, W) contiguous -> (B, H, W, C) contiguous."""
    lib = _lib.load()
    B, Cc, Hh, Ww = f.shape
    out = torch.empty(B, Hh, Ww, Cc, device=f.device, dtype=torch.float32)
    check(lib.gecco_nchw_to_nhwc_f32(_ptr(f), _ptr(out), B, Cc, Hh, Ww, _stream()), "gecco_nchw_to_nhwc_f32")
    return out


def to_channels_last_levels(features: Sequence[Tensor]) -> list[Tensor]:
    """Feature pyramid levels as (B, H, W, C) contiguous fp32.  A level that is NCHW-shaped but already stored
    channels-last (torch.channels_last) is re-viewed without a copy."""
    out = []
    for f in features:
        if f.dtype != torch.float32:
            f = f.float()
        if f.dim() != 4:
            raise _lib.GeccoHipError("feature maps must be (B, C, H, W)")
        if f.is_contiguous(memory_format=torch.channels_last) and not f.is_contiguous():
            out.append(f.permute(0, 2, 3, 1))  # a view: already (B, H, W, C) in memory
        else:
            out.append(nchw_to_nhwc(f.contiguous()))
    return out


def make_pyramid(levels_nhwc: Sequence[Tensor]) -> _lib.GeccoPyramid:
    """Channels-last (B, H, W, C) levels, all fp32 — or all fp16 (`half_levels`: the forward lookup's texel image)."""
    n = len(levels_nhwc)
    if not 1 <= n <= 4:
        raise _lib.GeccoHipError("1..4 pyramid levels supported")
    pyr = _lib.GeccoPyramid()
    pyr.n_levels = n
    dt = levels_nhwc[0].dtype
    for l, f in enumerate(levels_nhwc):
        assert f.is_contiguous() or f.permute(0, 3, 1, 2).is_contiguous(memory_format=torch.channels_last)
        pyr.C[l], pyr.H[l], pyr.W[l] = f.shape[3], f.shape[1], f.shape[2]
        if not f.is_cuda or f.dtype != dt or dt not in (torch.float32, torch.float16):
            raise _lib.GeccoHipError("pyramid levels must be HIP tensors, all fp32 or all fp16")
        pyr.feat[l] = f.data_ptr()
    pyr.texel_f16 = int(dt == torch.float16)
    return pyr


def half_levels(levels_nhwc: Sequence[Tensor]) -> list[Tensor]:
    """fp16 copies of channels-last fp32 levels (gecco_cast_f16): the texel image the "w2" forward lookup gathers — half the bytes;
    made once per conditioner call (`RayNetworkPlan.forward` keeps them while the fp32 levels are the same tensors, unchanged)."""
    lib = _lib.load()
    out = []
    for f in levels_nhwc:
        if f.dtype != torch.float32 or not f.is_cuda or not f.is_contiguous():
            raise _lib.GeccoHipError("half_levels: contiguous fp32 HIP levels expected")
        h = torch.empty(f.shape, dtype=torch.float16, device=f.device)
        check(lib.gecco_cast_f16(_ptr(f), C.c_void_p(h.data_ptr()), f.numel(), _stream()), "gecco_cast_f16")
        out.append(h)
    return out


def make_reparam(kind: int, mean: Tensor | None = None, std: Tensor | None = None, logit_scale: float = 1.1):
    return _lib.GeccoReparam(kind, _ptr(mean), _ptr(std), logit_scale)


def bilinear_taps(uv: Tensor, Hh: int, Ww: int):
    lib = _lib.load()
    n = uv.numel() // 2
    x0 = torch.empty(n, device=uv.device, dtype=torch.int32)
    y0 = torch.empty_like(x0)
    wx = torch.empty(n, device=uv.device, dtype=torch.float32)
    wy = torch.empty_like(wx)
    check(lib.gecco_bilinear_taps_f32(_ptr(uv), Hh, Ww, C.c_void_p(x0.data_ptr()), C.c_void_p(y0.data_ptr()), _ptr(wx),
                                      _ptr(wy), n, _stream()), "gecco_bilinear_taps_f32")
    shp = uv.shape[:-1]
    return x0.reshape(shp), y0.reshape(shp), wx.reshape(shp), wy.reshape(shp)


def ray_lookup(geom: Tensor, K: Tensor, levels_nhwc: Sequence[Tensor], reparam: _lib.GeccoReparam,
               coef: Tensor | None = None, want_stats: bool = False):
    lib = _lib.load()
    B, N, _ = geom.shape
    pyr = make_pyramid(levels_nhwc)
    Ct = sum(f.shape[3] for f in levels_nhwc)
    out = torch.empty(B, N, Ct, device=geom.device, dtype=torch.float32)
    stats = torch.empty(B, lib.gecco_lookup_row_tiles(N), 2, Ct, device=geom.device, dtype=torch.float32) if want_stats else None
    check(lib.gecco_ray_lookup_f32(_ptr(geom), _ptr(coef), _ptr(K), C.byref(reparam), C.byref(pyr), _ptr(out),
                                   _ptr(stats), B, N, _stream()), "gecco_ray_lookup_f32")
    return (out, stats) if want_stats else out


def ray_lookup_taps(geom: Tensor, K: Tensor, levels_nhwc: Sequence[Tensor], reparam: _lib.GeccoReparam, coef: Tensor | None = None):
    """The fused lookup's coordinate chain alone (same device functions as the lookup kernel): uv (B, N, 2) and, per pyramid level,
    integer taps x0 / y0 and fractional weights wx1 / wy1, each (L, B, N).  Diagnostics / index bit-exactness tests."""
    lib = _lib.load()
    B, N, _ = geom.shape
    pyr = make_pyramid(levels_nhwc)
    L = len(levels_nhwc)
    uv = torch.empty(B, N, 2, device=geom.device, dtype=torch.float32)
    x0 = torch.empty(L, B, N, device=geom.device, dtype=torch.int32)
    y0 = torch.empty_like(x0)
    wx = torch.empty(L, B, N, device=geom.device, dtype=torch.float32)
    wy = torch.empty_like(wx)
    check(lib.gecco_ray_lookup_taps_f32(_ptr(geom), _ptr(coef), _ptr(K), C.byref(reparam), C.byref(pyr), _ptr(uv), C.c_void_p(x0.data_ptr()),
                                        C.c_void_p(y0.data_ptr()), _ptr(wx), _ptr(wy), B, N, _stream()), "gecco_ray_lookup_taps_f32")
    return uv, x0, y0, wx, wy


class RayNetworkPlan:
    """EDMPrecond(RayNetwork(SetTransformer, reparam)) with a precomputed pyramid = the image-conditional
    Diffusion.forward, one C call."""

    def __init__(self, p: Mapping[str, Tensor], H: int, I: int = 64, pre: str = "", reparam_kind: int = 2,
                 rp_mean: Tensor | None = None, rp_std: Tensor | None = None, logit_scale: float = 1.1,
                 sigma_data: float = 1.0, precision: str | None = None, act: int | None = None,
                 options: Mapping[str, int] | None = None):
        self.st = SetTransformerPlan(p, pre + "backbone.", H, I, precision=precision, act=act, options=options)
        self.p = p
        self.lib = self.st.lib
        if reparam_kind == 2 and rp_mean is None:
            rp_mean, rp_std = p[pre + "reparam.uvl_mean"], p[pre + "reparam.uvl_std"]
        self._rp = (rp_mean, rp_std)
        self.table = _lib.GeccoRayNetwork(
            self.st.table, _ptr(p[pre + "xyz_embed.weight"]), _ptr(p[pre + "xyz_embed.bias"]),
            _ptr(p[pre + "img_feature_proj.1.weight"]), _ptr(p[pre + "img_feature_proj.1.bias"]),
            _ptr(p[pre + "output_proj.1.weight"]), _ptr(p[pre + "output_proj.1.bias"]),
            make_reparam(reparam_kind, rp_mean, rp_std, logit_scale), sigma_data)
        self._ws: dict[tuple, Tensor] = {}
        self.images = self.st.images
        self._tex16: tuple | None = None   # (signature of the fp32 levels, their fp16 texel images)

    def set_option(self, name: str, value: int) -> None:
        """Pin a path switch for THIS plan (0 / 1; negative: follow the process-wide default again)."""
        _pin_option(self.table.backbone, name, value)
        self.st.set_option(name, value)

    def _lookup_levels(self, levels_nhwc: Sequence[Tensor]) -> Sequence[Tensor]:
        """The levels the forward lookup gathers: in the "w2" mode the fp16 texel image of the pyramid (half the gathered bytes: the
        lookup sits at the Infinity-Cache gather ceiling on fp32 texels; coordinates, taps and weights stay fp32 and bit-exact) — cast
        once per conditioner call and kept while the fp32 levels are the same, unchanged tensors.  Every other mode, fp16 inputs,
        non-contiguous (re-viewed channels_last) levels and GECCO_LOOKUP16=0 gather the fp32 texels."""
        if (self.st.precision != "w2" or os.environ.get("GECCO_LOOKUP16", "1") == "0"
                or any(f.dtype != torch.float32 or not f.is_contiguous() for f in levels_nhwc)):
            return levels_nhwc
        sig = tuple((f.data_ptr(), f._version, tuple(f.shape)) for f in levels_nhwc)
        if self._tex16 is None or self._tex16[0] != sig:
            if torch.cuda.is_current_stream_capturing():
                return levels_nhwc   # (no allocation / stale-able cache inside a capture: the warm-up call makes the image)
            self._tex16 = (sig, half_levels(levels_nhwc), list(levels_nhwc))   # (the fp32 levels kept alive: a freed pointer could be re-used)
        return self._tex16[1]

    def forward(self, x: Tensor, sigma: Tensor, K: Tensor, levels_nhwc: Sequence[Tensor], return_raw: bool = False,
                cache: Sequence[Tensor] | None = None, do_cache: bool = False, out: Tensor | None = None):
        B, N, _ = x.shape
        den = torch.empty_like(x) if out is None else out
        raw = torch.empty_like(x) if return_raw else None
        L = self.st.L
        h_out = [torch.empty(B, self.st.I, self.st.C, device=x.device, dtype=torch.float32) for _ in range(L)] if do_cache else None

        levels_nhwc = self._lookup_levels(levels_nhwc)

        def call(lo, hi, idx):
            lv = [f[lo:hi] for f in levels_nhwc]               # a sample's pyramid: its slice of every level
            pyr = make_pyramid(lv)
            key = (hi - lo, N, tuple(f.shape[1:] for f in levels_nhwc), idx)
            if key not in self._ws:
                self._ws[key] = _ws(self.lib.gecco_ray_network_workspace_bytes(C.byref(self.table), C.byref(pyr), hi - lo, N),
                                    self.st.device)
            ws = self._ws[key]
            tok = self.images.token(cache is not None)
            tbl = _lib.GeccoRayNetwork.from_buffer_copy(self.table)   # this call's own copy (see LinearLiftPlan)
            ready = tbl.backbone.images_ready = self.images.ready(key, tok)
            if parts > 1:
                _pin_option(tbl.backbone, "mlpwshare", 1)
            cut = (lambda ts: None if ts is None else [None if t is None else t[lo:hi] for t in ts])
            try:
                check(self.lib.gecco_ray_network_fwd_f32(
                    C.byref(tbl), _ptr(x[lo:hi]), _ptr(sigma[lo:hi]), _ptr(K[lo:hi]), C.byref(pyr), _ptr(den[lo:hi]),
                    _ptr(None if raw is None else raw[lo:hi]), self.st._ptr_array(cut(cache), L), self.st._ptr_array(cut(h_out), L),
                    hi - lo, N, C.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "gecco_ray_network_fwd_f32")
            except BaseException:
                self.images.failed(key)
                raise
            self.images.built(key, tok, ready)
        parts = _fwd_parts(B, N) if B else 1
        if B == 0:
            pass   # an empty batch: empty results, nothing to launch
        elif parts > 1:
            _two_stream_halves(B, call, [x, sigma, K, den, raw, *levels_nhwc, *(cache or []), *(h_out or [])], parts)
        else:
            call(0, B, 0)
        res = (den, raw) if return_raw else den
        return (res, h_out) if do_cache else res


# ------------------------------------------------------------------------------- reparam / activation
def _ptr_any(t: Tensor) -> C.c_void_p:
    if not t.is_cuda or not t.is_contiguous() or t.dtype not in (torch.float32, torch.float64):
        raise _lib.GeccoHipError("expected a contiguous fp32/fp64 HIP tensor")
    return C.c_void_p(t.data_ptr())


def gaussian_reparam(x: Tensor, mean: Tensor, sigma: Tensor, inverse: bool) -> Tensor:
    lib = _lib.load()
    y = torch.empty_like(x)
    check(lib.gecco_gaussian_reparam(_ptr_any(x), _ptr(mean), _ptr(sigma), _ptr_any(y), x.numel(), x.shape[-1],
                                     int(inverse), int(x.dtype == torch.float64), _stream()), "gecco_gaussian_reparam")
    return y


def uvl_reparam(x: Tensor, K: Tensor, mean: Tensor, std: Tensor, logit_scale: float, inverse: bool) -> Tensor:
    lib = _lib.load()
    B, N, _ = x.shape
    y = torch.empty_like(x)
    check(lib.gecco_uvl_reparam(_ptr_any(x), _ptr(K), _ptr(mean), _ptr(std), logit_scale, _ptr_any(y), B, N,
                                int(inverse), int(x.dtype == torch.float64), _stream()), "gecco_uvl_reparam")
    return y


def relu(x: Tensor) -> Tensor:
    lib = _lib.load()
    y = torch.empty_like(x)
    check(lib.gecco_relu_f32(_ptr(x), _ptr(y), x.numel(), _stream()), "gecco_relu_f32")
    return y


def relu_bwd(y: Tensor, dy: Tensor) -> Tensor:
    lib = _lib.load()
    du = torch.empty_like(y)
    check(lib.gecco_relu_bwd_f32(_ptr(y), _ptr(dy), _ptr(du), y.numel(), _stream()), "gecco_relu_bwd_f32")
    return du


def gaussian_act(x: Tensor, alpha: Tensor, normalized: bool = True) -> Tensor:
    lib = _lib.load()
    y = torch.empty_like(x)
    check(lib.gecco_gaussian_act_f32(_ptr(x), _ptr(alpha), _ptr(y), x.numel(), int(normalized), _stream()),
          "gecco_gaussian_act_f32")
    return y
