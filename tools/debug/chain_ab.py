"""A/B of a fused launch (option "chain", "mlpfused", ...) against the kernels it replaces: per-layer cached inducer
states and the denoised output.  python tools/debug/chain_ab.py [case] [option]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gecco_amd import hip_ops as ops
from oracle import cases

name = sys.argv[1] if len(sys.argv) > 1 else "uncond_d128_L4_N256"
opt = sys.argv[2] if len(sys.argv) > 2 else "chain"
p, x, sigma = cases.uncond_inputs(name)
pc = {k: v.cuda() for k, v in p.items()}
net = ops.LinearLiftPlan(pc, cases.H, cases.I, precision="fp16")
out = {}
for chain in (0, 1):
    ops.set_option(opt, chain)
    den, hs = net.forward(x.cuda(), sigma.cuda(), do_cache=True)
    out[chain] = (den.cpu(), [c.cpu() for c in hs])
for li, (a, b) in enumerate(zip(out[1][1], out[0][1])):
    d = (a - b).abs()
    print(f"layer {li}: max|d| {d.max():.3e} max|ref| {b.abs().max():.3e} frac differing {(d > 0).float().mean():.4f} "
          f"rel-L2 {(d.norm() / b.norm()):.3e} nan {torch.isnan(a).any().item()}")
    if li == 0:
        rows = d.amax(dim=(0, 2))
        cols = d.amax(dim=(0, 1))
        print("  worst rows", rows.topk(5).indices.tolist(), "worst cols", cols.topk(5).indices.tolist())
        print("  per-sample max", d.amax(dim=(1, 2)).tolist())
d = (out[1][0] - out[0][0]).abs()
print("den: max|d|", d.max().item(), "rel-L2", (d.norm() / out[0][0].norm()).item())
