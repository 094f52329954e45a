"""Robustness sweep of the image-conditional training path (ConvNeXt conditioner trained): loss and every parameter gradient over odd
image and cloud sizes, split-bf16 against the 16-mixed setting.    python tools/debug/cond_train_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import weights as W  # noqa: E402
from tests.test_hip_convnext import _seeded_state  # noqa: E402
from tests.test_modules_cpu import build_cond  # noqa: E402
from gecco_amd import autograd as ag, hip_ops  # noqa: E402
from gecco_amd.models.feature_pyramid import ConvNeXtExtractor  # noqa: E402
from gecco_amd.structs import Context3d, Example  # noqa: E402


def run(d, N, hw, B, amp):
    ag.WEIGHT_IMAGES.__init__()
    hip_ops.set_default_precision("bf16x3")
    cn = ConvNeXtExtractor(n_stages=3, model="tiny", pretrained=False)
    csd = _seeded_state(cn, 9)
    cn.load_state_dict(csd, strict=True)
    m = build_cond(d, 2, conditioner=cn)
    p = W.ray_network_state_dict(17, d, 2, 64, 8)
    sd = {"backbone.model." + k: v for k, v in p.items()}
    sd["reparam.uvl_mean"], sd["reparam.uvl_std"] = p["reparam.uvl_mean"], p["reparam.uvl_std"]
    sd.update({"conditioner." + k: v for k, v in csd.items()})
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    rs = np.random.RandomState(3)
    img = torch.from_numpy(rs.rand(B, 3, hw, hw).astype(np.float32)).cuda()
    _, K = W.synthetic_context(4, B, hw=hw)
    ctx = Context3d(image=img, K=K.cuda())
    data = m.reparam.diffusion_to_data(torch.from_numpy((0.5 * rs.randn(B, N, 3)).astype(np.float32)).cuda(), ctx)
    torch.manual_seed(5)
    with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        loss = m.training_step(Example(data, ctx), 0)
    (loss * 64.0).backward()
    torch.cuda.synchronize()
    return float(loss), {n: q.grad.detach().clone() / 64.0 for n, q in m.named_parameters() if q.grad is not None}


bad = 0
for d in (128, 384):
    for hw in (64, 96, 224):
        for N in (200, 256, 1000):
            B = 3
            try:
                l0, g0 = run(d, N, hw, B, False)
                l, g = run(d, N, hw, B, True)
            except Exception as e:  # noqa: BLE001
                print(f"d={d} hw={hw} N={N} FAILED: {str(e)[:300]}")
                bad += 1
                continue
            tot = float(torch.cat([(g[n] - g0[n]).flatten() for n in g0]).norm() / torch.cat([g0[n].flatten() for n in g0]).norm())
            worst = max(((float((g[n] - g0[n]).norm() / g0[n].norm().clamp_min(1e-20)), n) for n in g0 if not n.endswith(".alpha")), key=lambda t: t[0])
            fin = all(bool(torch.isfinite(v).all()) for v in g.values())
            flag = "" if (tot < 5e-3 and worst[0] < 3e-2 and fin) else "   <-- OUTLIER"
            bad += bool(flag)
            print(f"d={d:3d} hw={hw:3d} N={N:4d} loss rel {abs(l - l0) / abs(l0):.1e} grads {tot:.1e} worst {worst[0]:.1e} ({worst[1][-40:]}){flag}", flush=True)
print("outliers / failures:", bad)
