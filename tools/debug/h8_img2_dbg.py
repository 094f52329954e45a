import math, sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge; ge.build()
from gecco_amd import hip_ops as ops
rs = np.random.RandomState(1)
B, rows, K, Wd = 1, 128, 128, 256
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
x, W0 = t(rs.randn(B, rows, K)), t(rs.randn(Wd, K) / math.sqrt(K))
u = F.linear(x.double(), W0.double())
for kind in (1, 2):
    img = ops.linear_h8_img(x.cuda(), None, W0.cuda(), None, kind=kind)
    dec = (ops.decode_split_image(img) if kind == 1 else ops.decode_h8_image(img)).cpu().double()
    err = (dec - u).abs()
    print("kind", kind, "max", float(err.max()), "rel", float(err.max() / u.abs().max()), "L2", float(err.norm() / u.norm()))
    bad = (err > 20 * err.median()).nonzero()
    print(" n bad", len(bad), bad[:24].tolist())
    if kind == 2:
        i8 = img.cpu()
        hi = i8[..., :16384].contiguous().view(torch.float16).reshape(B, rows // 128, Wd // 64, 4, 2, 2, 2, 32, 8).double().permute(0, 1, 3, 7, 2, 4, 6, 5, 8).reshape(B, rows, Wd)
        lo = dec - hi
        ehi = (hi - u)
        print(" hi-only rel", float(ehi.abs().max() / u.abs().max()), " lo vs needed: max |(u-hi) - lo|", float(((u - hi) - lo).abs().max()))
        b = bad[0].tolist() if len(bad) else [0, 0, 0]
        print(" sample", b, float(u[tuple(b)]), float(hi[tuple(b)]), float(lo[tuple(b)]), float(u[tuple(b)] - hi[tuple(b)]))
print("--- with AdaGN prologue, bias, GaussianActivation, K=384")
B, rows, K, Wd = 2, 256, 384, 768
x, W0, b0 = t(rs.randn(B, rows, K)), t(rs.randn(Wd, K) / math.sqrt(K)), t(rs.randn(Wd) / math.sqrt(K))
pa, po = t(1 + 0.3 * rs.randn(B, K)), t(0.3 * rs.randn(B, K))
alpha = t(np.array(0.9))
u = F.linear((x.double() * pa[:, None].double() + po[:, None].double()), W0.double(), b0.double())
hid = (torch.exp(-u * u / (2 * 0.9 ** 2)) - 0.7) / 0.28
for kind in (1, 2):
    img = ops.linear_h8_img(x.cuda(), (pa.cuda(), po.cuda()), W0.cuda(), b0.cuda(), act_alpha=alpha.cuda(), kind=kind)
    dec = (ops.decode_split_image(img) if kind == 1 else ops.decode_h8_image(img)).cpu().double()
    err = (dec - hid).abs()
    print("kind", kind, "max", float(err.max()), "rel", float(err.max() / hid.abs().max()), "L2", float(err.norm() / hid.norm()))
    bad = (err > 0.3 * err.max()).nonzero()
    print(" n bad", len(bad), bad[:16].tolist())
    for bb in bad[:6].tolist():
        print("   ", bb, "ref", float(hid[tuple(bb)]), "got", float(dec[tuple(bb)]), "u", float(u[tuple(bb)]))
    if kind == 2:
        i8 = img.cpu()
        G = Wd // 64
        hi = i8[..., :16384].contiguous().view(torch.float16).reshape(B, rows // 128, G, 4, 2, 2, 2, 32, 8).double().permute(0, 1, 3, 7, 2, 4, 6, 5, 8).reshape(B, rows, Wd)
        lob = i8[..., 16384:].contiguous().reshape(B, rows // 128, G, 4, 2, 2, 32, 16).permute(0, 1, 3, 6, 2, 4, 5, 7).reshape(B, rows, Wd)
        for bb in bad[:9].tolist():
            tb = tuple(bb)
            # the kernel's own fp32 value is unknown; show hi, lo byte, neighbours' lo bytes
            print("   ", bb, "hi", float(hi[tb]), "lo byte", hex(int(lob[tb])), "ref-hi scaled", float((hid[tb] - hi[tb]) * 16384),
                  "row lo bytes around", [hex(int(v)) for v in lob[bb[0], bb[1], max(0, bb[2] - 2):bb[2] + 3]])
