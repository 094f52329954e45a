import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import cases
from gecco_amd import hip_ops as ops
name = "uncond_d128_L4_N256"
p, x, sigma = cases.uncond_inputs(name)
pc = {k: v.cuda() for k, v in p.items()}
net = ops.LinearLiftPlan(pc, cases.H, cases.I, precision="fp16")
outs = []
for hm in (0, 1, 0, 1, 1):
    ops.set_option("headmajor", hm)
    outs.append(net.forward(x.cuda(), sigma.cuda()).cpu())
ops.set_option("headmajor", -1)
print("0 vs 0:", (outs[0] - outs[2]).abs().max().item(), "1 vs 1:", (outs[1] - outs[3]).abs().max().item(), (outs[3] - outs[4]).abs().max().item(),
      "0 vs 1:", (outs[0] - outs[1]).abs().max().item())
d = (outs[0] - outs[1]).abs()
print("n diff", int((d > 0).sum()), "of", d.numel(), "idx", (d > 0).nonzero()[:5].tolist())
