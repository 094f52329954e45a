"""Where the conditional path's error in the "w2" mode comes from: F_x against the oracle with the fp16 texel image and the fp16-operand
img_feature_proj switched off one at a time (small L: the image features reach the output almost directly).
    python tools/debug/imgproj_err.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from gecco_amd import hip_ops  # noqa: E402
from oracle import cases, cpu_ref  # noqa: E402
from oracle import weights as W  # noqa: E402

for (B, N, d, L) in ((2, 130, 384, 1), (3, 128, 384, 1), (2, 333, 128, 1), (2, 256, 384, 6)):
    hw, cdims = 64, (96, 192, 384)
    p = W.ray_network_state_dict(91 + N, d, L, cases.I, cases.H, context_dims=cdims)
    feats, K = W.synthetic_context(92 + N, B, hw=hw, context_dims=cdims)
    g = torch.Generator().manual_seed(93 + N)
    x = torch.randn(B, N, 3, generator=g)
    sigma = torch.tensor([0.05, 3.0, 80.0][:B])
    with torch.no_grad():
        ref, raw_ref = cpu_ref.cond_denoiser(p, "", cases.H, K, feats)(x, sigma, return_raw=True)
    levels = hip_ops.to_channels_last_levels([f.cuda() for f in feats])
    for tex16 in (1, 0):
        for ip in (1, 0):
            os.environ["GECCO_LOOKUP16"] = str(tex16)
            net = hip_ops.RayNetworkPlan({k: v.cuda() for k, v in p.items()}, cases.H, cases.I, precision="w2", options={"imgproj16": ip})
            den, raw = net.forward(x.cuda(), sigma.cuda(), K.cuda(), levels, return_raw=True)
            e = cpu_ref.rel_err(raw.cpu(), raw_ref)
            print(f"B={B} N={N} d={d} L={L}: fp16 texels {tex16}, imgproj16 {ip}: F_x max-rel {e[0]:.2e} rel-L2 {e[1]:.2e}", flush=True)
