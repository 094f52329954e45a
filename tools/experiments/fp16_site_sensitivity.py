"""Where does the fp16 mode's error on the raw network output F_x come from?  CPU emulation on the oracle network at
the C2 size (N=2048, d=384, L=6): every matrix product of the path has its operands rounded to fp16 (fp32 accumulate),
as the HIP fp16 mode does; then one class of product sites at a time is computed exactly, or with one operand exact,
and the F_x error (max-norm / rel-L2 against the fp32 forward) is printed.  The variance a site class contributes is
err_all^2 - err_without^2.

Usage: python tools/experiments/fp16_site_sensitivity.py [sigma] [N] [scheme ...]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cases, cpu_ref  # noqa: E402
from oracle import weights as W  # noqa: E402

_linear, _matmul = F.linear, torch.matmul
SITE_OF = {}          # weight data_ptr -> site class
SCHEME = {}           # site class -> "exact" | "f16" | "f16A" (A rounded, W exact) | "f16W" | "x2a" | "x2w" | "x3"
CTX = [None]          # attention context for matmul sites
CALL = [0]


def r16(x):
    return x.half().float()


def r8(x):
    """fp8 e4m3 rounding behind a power-of-two scale that puts the tensor's max near 256 (what a per-launch scale does)"""
    m = float(x.abs().max())
    if m == 0:
        return x
    sc = 2.0 ** np.floor(np.log2(256.0 / m))
    return (x * sc).to(torch.float8_e4m3fn).float() / sc


def rmx(x, mant_bits, emax, block=32):
    """block-scaled low-precision rounding (OCP MX style): along the last dim in blocks of `block`, a power-of-two scale puts the
    block's max at the format's top binade; elements keep `mant_bits` mantissa bits (implicit one) over an exponent range of
    2^emax .. 2^0 (below: fixed-point steps of 2^-mant_bits).  mant_bits 1, emax 2 = fp4 e2m1; 3, 2 = fp6 e2m3; 2, 4 = fp6 e3m2."""
    shp = x.shape
    K = shp[-1]
    pad = (-K) % block
    xp = torch.nn.functional.pad(x, (0, pad)) if pad else x
    xb = xp.reshape(*xp.shape[:-1], -1, block)
    m = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    sc = 2.0 ** (torch.floor(torch.log2(m)) - emax)
    y = xb / sc
    e = torch.floor(torch.log2(y.abs().clamp_min(2.0 ** -20))).clamp_min(0.0)       # binade (>= 0: below 1 the step is fixed)
    step = 2.0 ** (e - mant_bits)
    y = torch.round(y / step) * step
    top = (2.0 - 2.0 ** -mant_bits) * 2.0 ** emax
    y = y.clamp(-top, top) * sc
    return y.reshape(xp.shape)[..., :K].reshape(shp)


def product(a, b_t, scheme, mm):
    """a @ b_t-ish product through `mm(a, b)` with operands per `scheme`."""
    if scheme == "exact":
        return mm(a, b_t)
    if scheme == "f16":
        return mm(r16(a), r16(b_t))
    if scheme == "f16A":
        return mm(r16(a), b_t)
    if scheme == "f16W":
        return mm(a, r16(b_t))
    if scheme == "x2w":    # A split in two fp16 terms, W rounded once: 2 MFMAs
        ah = r16(a)
        return mm(ah, r16(b_t)) + mm(r16(a - ah), r16(b_t))
    if scheme == "x2a":    # W split, A rounded once
        bh = r16(b_t)
        return mm(r16(a), bh) + mm(r16(a), r16(b_t - bh))
    if scheme in ("x2a_v", "x2a_k"):   # linear sites only: two-term weights for the second (V) / first (K) half of the outputs
        bh = r16(b_t)
        lo = r16(b_t - bh)
        half = b_t.shape[0] // 2
        lo = torch.cat([torch.zeros_like(lo[:half]), lo[half:]]) if scheme == "x2a_v" else torch.cat([lo[:half], torch.zeros_like(lo[half:])])
        return mm(r16(a), bh) + mm(r16(a), lo)
    if scheme == "x2a_v8":   # V half: fp16 a x (fp16 w_hi) + fp8 a x fp8 w_lo (the lo term on the fp8 matrix instruction)
        bh = r16(b_t)
        lo = b_t - bh
        half = b_t.shape[0] // 2
        lo = torch.cat([torch.zeros_like(lo[:half]), lo[half:]])
        return mm(r16(a), bh) + mm(r8(a), r8(lo))
    if scheme == "x3":
        ah, bh = r16(a), r16(b_t)
        return mm(ah, bh) + mm(r16(a - ah), bh) + mm(ah, r16(b_t - bh))
    if scheme == "h8":     # fp16 main product + both cross terms on fp8 (e4m3) operands, power-of-two tensor scales
        ah, bh = r16(a), r16(b_t)
        return mm(ah, bh) + mm(r8(a - ah), r8(b_t)) + mm(r8(a), r8(b_t - bh))
    if scheme in ("h4", "h6", "h6b"):   # fp16 main product + both cross terms on block-scaled fp4 (e2m1) / fp6 (e2m3 / e3m2) operands
        mb, em = {"h4": (1, 2), "h6": (3, 2), "h6b": (2, 4)}[scheme]
        ah, bh = r16(a), r16(b_t)
        q = lambda t: rmx(t, mb, em)
        return mm(ah, bh) + mm(q(a - ah), q(b_t)) + mm(q(a), q(b_t - bh))
    if scheme == "h8w":    # A rounded once (fp16), W = fp16 hi + fp8 lo: 1.5 MFMA units
        bh = r16(b_t)
        return mm(r16(a), bh) + mm(r8(a), r8(b_t - bh))
    if scheme == "f8":     # both operands fp8 e4m3 behind power-of-two tensor scales (what an "fp8 mode" would compute)
        return mm(r8(a), r8(b_t))
    if scheme == "bf16":
        return mm(a.bfloat16().float(), b_t.bfloat16().float())
    raise ValueError(scheme)


def emu_linear(x, w, b=None):
    site = SITE_OF.get(w.data_ptr())
    if site is None or w.shape[1] < 16:
        return _linear(x, w, b)
    y = product(x.float(), w, SCHEME.get(site, "exact"), lambda a, ww: _linear(a, ww))
    return y if b is None else y + b


def emu_matmul(a, b):
    if CTX[0] is None:
        return _matmul(a, b)
    CALL[0] += 1
    site = CTX[0] + (".qk" if CALL[0] % 2 == 1 else ".pv")
    return product(a, b, SCHEME.get(site, "exact"), _matmul)


def wrap_attn(fn, name):
    def inner(*a, **k):
        CTX[0], CALL[0] = name, 0
        try:
            return fn(*a, **k)
        finally:
            CTX[0] = None
    return inner


def main():
    sigma_v = float(sys.argv[1]) if len(sys.argv) > 1 else 0.002
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    d, L = 384, 6
    p = W.linear_lift_state_dict(3, d, L, cases.I, cases.H)
    for k, v in p.items():
        if k.endswith("in_proj_weight"):
            C = v.shape[1]
            SITE_OF[v[:C].data_ptr()] = "q_proj"
            SITE_OF[v[C:2 * C].data_ptr()] = "chain"
            SITE_OF[v[2 * C:].data_ptr()] = "chain"
        elif k.endswith("kv_proj.weight"):
            SITE_OF[v.data_ptr()] = "kv_proj"
        elif "broadcast.pool.out_proj" in k or "broadcast.mlp." in k and k.endswith("weight"):
            SITE_OF[v.data_ptr()] = "chain"
        elif k.endswith("unpool.out_proj.weight"):
            SITE_OF[v.data_ptr()] = "out_proj"
        elif k.endswith(".mlp.0.weight"):
            SITE_OF[v.data_ptr()] = "mlp0"
        elif k.endswith(".mlp.2.weight"):
            SITE_OF[v.data_ptr()] = "mlp2"
    sites = ["kv_proj", "q_proj", "pool.qk", "pool.pv", "chain", "unpool.qk", "unpool.pv", "out_proj", "mlp0", "mlp2"]
    rs = np.random.RandomState(5)
    data = torch.from_numpy(rs.randn(1, N, 3).astype(np.float32))
    sigma = torch.tensor([sigma_v])
    x = data + sigma_v * torch.from_numpy(rs.randn(1, N, 3).astype(np.float32))
    F.linear = emu_linear
    torch.matmul = emu_matmul
    cpu_ref.attention_pool = wrap_attn(cpu_ref.attention_pool, "pool")
    cpu_ref.mha_unpool = wrap_attn(cpu_ref.mha_unpool, "unpool")
    D = cpu_ref.uncond_denoiser(p, "", cases.H)

    def run(scheme):
        SCHEME.clear()
        SCHEME.update(scheme)
        with torch.no_grad():
            return D(x, sigma, return_raw=True)

    den0, raw0 = run({})
    allf16 = {s: "f16" for s in sites}
    den, raw = run(allf16)
    e_all = cpu_ref.rel_err(raw, raw0)
    print(f"sigma={sigma_v} N={N}: all sites fp16: F_x {e_all}  D {cpu_ref.rel_err(den, den0)}")
    extra = sys.argv[3:]
    if extra:   # named experiments: site=scheme,site=scheme ...
        for spec in extra:
            sch = dict(allf16)
            for kv in spec.split(","):
                k, v = kv.split("=")
                for s in sites:
                    if s == k or k == "all" or (k.endswith("*") and s.startswith(k[:-1])):
                        sch[s] = v
            den, raw = run(sch)
            print(f"  {spec:60s} F_x {cpu_ref.rel_err(raw, raw0)}")
        return
    for s in sites:
        for alt in ("exact", "f16A", "f16W"):
            sch = dict(allf16)
            sch[s] = alt
            den, raw = run(sch)
            e = cpu_ref.rel_err(raw, raw0)
            print(f"  {s:10s} -> {alt:6s}: F_x max {e[0]:.3e} L2 {e[1]:.3e}   variance share (L2) {1 - (e[1] / e_all[1]) ** 2:+.2f}")


if __name__ == "__main__":
    main()
