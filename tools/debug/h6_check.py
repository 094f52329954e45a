"""mlp.0's product in the h8 arithmetic with fp8 cross terms ("h6" = 0) and with fp6 block-scaled cross terms ("h6" = 1) against float64:
plain operands and the outlier cases of tests/test_hip_ops.py.   python tools/debug/h6_check.py"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from gecco_amd import hip_ops as ops  # noqa: E402
from oracle import cpu_ref  # noqa: E402
from tests.test_hip_ops import _h8_outlier_case, _rs, _t  # noqa: E402

for h6 in (0, 1):
    ops.set_option("h6", h6)
    for (B, rows, K, Nout) in ((2, 512, 384, 768), (1, 256, 256, 512), (3, 256, 128, 256)):
        rs = _rs(B * 7 + rows + K + Nout)
        x, W = _t(rs.randn(B, rows, K)), _t(rs.randn(Nout, K) / math.sqrt(K))
        img = ops.linear_h8_img(x.cuda(), None, W.cuda(), None, kind=2)
        e = cpu_ref.rel_err(ops.decode_h8_image(img).cpu().double(), F.linear(x.double(), W.double()))
        print(f"h6={h6} plain K={K}: max-rel {e[0]:.2e} rel-L2 {e[1]:.2e}")
    for wmax, ymax in ((1.0, 4.0), (8.0, 500.0), (14.0, 448.0), (100.0, 3000.0), (1000.0, 1e5)):
        rs = _rs(int(wmax) + int(ymax) % 1000)
        x, W0, b0, W2 = _h8_outlier_case(rs, 2, 256, 384, 768, wmax, ymax)
        img = ops.linear_h8_img(x.cuda(), None, W0.cuda(), b0.cuda(), kind=2)
        u_ref = F.linear(x.double().clamp(-3584, 3584), W0.double(), b0.double()).clamp(-3584, 3584)
        u = ops.decode_h8_image(img).cpu()
        e = cpu_ref.rel_err(u.double(), u_ref)
        print(f"h6={h6} outliers |w|<={wmax} |y|<={ymax}: finite {bool(torch.isfinite(u).all())} max-rel {e[0]:.2e} rel-L2 {e[1]:.2e}")
ops.set_option("h6", -1)
