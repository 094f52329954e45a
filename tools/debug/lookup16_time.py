"""Time the forward lookup on fp32 and fp16 texels at the C3 shape (B 64, N 2048, 224^2 pyramids): python tools/debug/lookup16_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
ge.build()
from gecco_amd import hip_ops as ops
g = torch.Generator().manual_seed(0)
B, N, hw = 64, 2048, 224
feats = [torch.randn(B, c, hw // s, hw // s, generator=g).cuda() for c, s in ((96, 4), (192, 8), (384, 16))]
lv32 = ops.to_channels_last_levels(feats)
lv16 = ops.half_levels(lv32)
K = torch.zeros(B, 3, 3); K[:, 0, 0] = K[:, 1, 1] = 1.1; K[:, 0, 2] = K[:, 1, 2] = 0.5; K[:, 2, 2] = 1.0
K = K.cuda()
x = (torch.randn(B, N, 3, generator=g) * torch.tensor([1.5, 1.5, 0.6])).cuda()
um, us = torch.tensor([0.0, 0.0, 1.38]).cuda(), torch.tensor([0.56, 0.60, 0.49]).cuda()
rp = ops.make_reparam(2, um, us, 1.1)
def t(lv):
    for _ in range(3): ops.ray_lookup(x, K, lv, rp, want_stats=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.ray_lookup(x, K, lv, rp, want_stats=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
print(f"lookup fp32 texels {t(lv32):.4f} ms, fp16 texels {t(lv16):.4f} ms")
