"""CPU emulation of candidate GEMM arithmetic schemes on the oracle network: how far from the fp32 forward does the
denoised output land when every nn.Linear product is computed as ...
   bf16x3   : hi/lo bf16 split, a_lo*b_hi + a_hi*b_lo + a_hi*b_hi          (what the HIP kernels do)
   bf16     : plain bf16 operands
   f16f8    : fp16 main product + the two cross terms with fp8(e4m3) operands (a8*b_lo8 + a_lo8*b8, lo scaled 2^11)
   f16x2w   : (a_hi16 + a_lo16) * b_hi16   (weights rounded once to fp16)
   f16x2a   : a_hi16 * (b_hi16 + b_lo16)   (activations rounded once to fp16)
   f16      : plain fp16 operands
Usage: python tools/experiments/mixed_precision_emulation.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cases, cpu_ref, weights  # noqa: E402

MODE = "fp32"
_orig = F.linear


def _bf16_hi(x):
    return (x.view(torch.int32) & -65536).view(torch.float32)


def _f8(x):
    return x.to(torch.float8_e4m3fn).float()


def emu_linear(x, w, b=None):
    if MODE == "fp32" or w.shape[1] < 16:
        return _orig(x, w, b)
    x = x.float()
    if MODE == "bf16x3":
        xh, wh = _bf16_hi(x), _bf16_hi(w)
        xl, wl = (x - xh).bfloat16().float(), (w - wh).bfloat16().float()
        y = _orig(xl, wh) + _orig(xh, wl) + _orig(xh, wh)
    elif MODE == "bf16":
        y = _orig(x.bfloat16().float(), w.bfloat16().float())
    elif MODE == "f16f8":
        xh, wh = x.half().float(), w.half().float()
        s = 2.0 ** 11
        xl8, wl8 = _f8((x - xh) * s), _f8((w - wh) * s)
        x8, w8 = _f8(x), _f8(w)
        y = _orig(xh, wh) + (_orig(x8, wl8) + _orig(xl8, w8)) / s
    elif MODE == "f16":
        y = _orig(x.half().float(), w.half().float())
    elif MODE == "f16x2a":   # activations rounded once, weights hi + lo
        xh, wh = x.half().float(), w.half().float()
        wl = (w - wh).half().float()
        y = _orig(xh, wh) + _orig(xh, wl)
    elif MODE == "f16x2w":
        xh, wh = x.half().float(), w.half().float()
        xl = (x - xh).half().float()
        y = _orig(xh, wh) + _orig(xl, wh)
    else:
        raise ValueError(MODE)
    return y if b is None else y + b


def main():
    global MODE
    F.linear = emu_linear
    torch.manual_seed(0)
    for name in cases.UNCOND_CASES:
        p, data, sigma = cases.uncond_inputs(name)
        D = cpu_ref.uncond_denoiser(p, "", cases.H)
        MODE = "fp32"
        ref = D(data, sigma)
        for m in ("bf16x3", "f16f8", "f16x2w", "f16x2a", "f16", "bf16"):
            MODE = m
            out = D(data, sigma)
            print(f"{name:24s} {m:8s} rel err {cpu_ref.rel_err(out, ref)}")


if __name__ == "__main__":
    main()
