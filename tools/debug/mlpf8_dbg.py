import sys, math, numpy as np, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
ge.build()
from gecco_amd import hip_ops as ops
import torch.nn.functional as F
rs = np.random.RandomState(3)
_t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
B, rows, K, Wd = 1, 128, 384, 768
x, W0, b0 = _t(rs.randn(B, rows, K)), _t(rs.randn(Wd, K) / math.sqrt(K)), _t(rs.randn(Wd) / math.sqrt(K))
W2, b2 = _t(rs.randn(K, Wd) / math.sqrt(Wd)), _t(rs.randn(K) / math.sqrt(Wd))
pa, po = _t(1 + 0.3 * rs.randn(B, K)), _t(0.3 * rs.randn(B, K))
for act in ("none", "relu"):
    kw = dict(act="relu") if act == "relu" else {}
    xc, pro = x.cuda(), (pa.cuda(), po.cuda())
    img = ops.linear_h8_img(xc, pro, W0.cuda(), b0.cuda(), kind=2, **kw)
    ref = ops.linear_h8_areg(img, W2.cuda(), b2.cuda(), residual=xc)
    got, st = ops.mlp_fused_h8(xc.clone(), pro, W0.cuda(), b0.cuda(), W2.cuda(), b2.cuda(), want_stats=True, **kw)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(got)
    print(act, "nonfinite:", int(bad.sum()), "of", got.numel())
    if bad.any():
        r = bad[0].any(1).nonzero().flatten().tolist(); c = bad[0].any(0).nonzero().flatten().tolist()
        print("  rows", r[:20], len(r), "cols", c[:20], len(c))
    d = (got - ref).abs()
    d[bad] = 0
    print("  max diff finite", float(d.max()), "ref max", float(ref.abs().max()))
    # per column block error
    for cb in range(6):
        print("   cb", cb, float(d[0, :, 64*cb:64*cb+64].max()), end=";")
    print()
    for rg in range(4):
        print("   rowgroup", rg, float(d[0, 32*rg:32*rg+32].max()), end=";")
    print()
