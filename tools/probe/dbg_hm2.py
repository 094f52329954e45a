import sys, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
ge.build()
from oracle import cases
from gecco_amd import hip_ops as ops
for name in ("uncond_d128_L4_N256", "uncond_d384_L6_N128"):
    p, x, sigma = cases.uncond_inputs(name)
    pc = {k: v.cuda() for k, v in p.items()}
    net = ops.LinearLiftPlan(pc, cases.H, cases.I, precision="fp16")
    out = {}
    for hm in (0, 1):
        ops.set_option("headmajor", hm)
        out[hm] = net.forward(x.cuda(), sigma.cuda()).cpu()
    ops.set_option("headmajor", -1)
    d = (out[0] - out[1]).abs()
    print(name, "equal", torch.equal(out[0], out[1]), "max", d.max().item(), "n diff", int((d > 0).sum()), "nan", int(torch.isnan(out[0]).sum()), int(torch.isnan(out[1]).sum()))
