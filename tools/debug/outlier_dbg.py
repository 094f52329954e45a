import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
ge.build()
from gecco_amd import hip_ops as ops
from oracle import cases, cpu_ref, weights as W
def run(N, d, L, smax, wmax):
    p = W.linear_lift_state_dict(131 + d, d, L, cases.I, cases.H)
    rs = np.random.RandomState(d + L)
    for li in range(L):
        pre = f"inner.layers.{li}."
        for key in ("broadcast.unpool.out_proj.weight", "mlp.0.weight", "mlp.2.weight"):
            w = p[pre + key]
            for _ in range(6):
                w[rs.randint(w.shape[0]), rs.randint(w.shape[1])] = float(rs.choice([-1, 1, 0.6, -0.4])) * wmax
        if smax:
            p[pre + "mlp_norm.scale.bias"][rs.randint(d)] = smax
            p[pre + "mlp_norm.scale.bias"][rs.randint(d)] = -0.75 * smax
            p[pre + "mlp_norm.bias.bias"][rs.randint(d)] = 0.5 * smax
    x, sigma = W.synthetic_cloud(N, 4, N)
    with torch.no_grad():
        ref, raw_ref = cpu_ref.uncond_denoiser(p, "", cases.H)(x, sigma, return_raw=True)
    pc = {k: v.cuda() for k, v in p.items()}
    out = []
    for pr in ("fp32", "bf16x3", "mixed"):
        den, raw = ops.LinearLiftPlan(pc, cases.H, cases.I, precision=pr).forward(x.cuda(), sigma.cuda(), return_raw=True)
        out.append(f"{pr} F_x {cpu_ref.rel_err(raw.cpu(), raw_ref)[0]:.2e}")
    print(f"d={d} L={L} scale-outlier {smax} |w|<={wmax}: " + ", ".join(out))
for (smax, wmax) in ((0, 0.3), (0, 8.0), (10.0, 0.3), (120.0, 0.3), (120.0, 8.0), (30.0, 8.0)):
    run(256, 128, 3, smax, wmax)
    run(384, 384, 2, smax, wmax)
