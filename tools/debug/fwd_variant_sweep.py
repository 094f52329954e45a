"""Forward (no-grad) sweep over model VARIANTS: the reference's default activation (nn.ReLU), other inducer / head counts and MLP widths,
every arithmetic mode against the exact-fp32 HIP kernels on the raw network output.    python tools/debug/fwd_variant_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gecco_amd import hip_ops  # noqa: E402
from gecco_amd.models.activation import GaussianActivation  # noqa: E402
from gecco_amd.models.set_transformer import SetTransformer  # noqa: E402

BARS = {"bf16x3": 3e-4, "mixed": 1e-3, "fp16": 6e-3}
bad = 0
for (d, act, I, H, mult, L) in ((128, torch.nn.ReLU, 64, 8, 2, 3), (384, torch.nn.ReLU, 64, 8, 2, 3), (256, GaussianActivation, 32, 8, 2, 3),
                                (256, GaussianActivation, 96, 4, 2, 3), (128, GaussianActivation, 64, 4, 4, 3), (384, GaussianActivation, 64, 12, 2, 3),
                                (256, torch.nn.ReLU, 64, 16, 1, 3), (384, GaussianActivation, 64, 6, 2, 3), (512, GaussianActivation, 64, 8, 2, 3),
                                (384, GaussianActivation, 64, 8, 3, 2)):
    torch.manual_seed(2)
    st = SetTransformer(n_layers=L, num_inducers=I, feature_dim=d, t_embed_dim=1, num_heads=H, activation=act, mlp_blowup=mult)
    with torch.no_grad():   # undo the x 0.1 of the residual branches at init: the branches should matter
        for layer in st.layers:
            layer.broadcast.unpool.out_proj.weight *= 10.0
            layer.mlp[-1].weight *= 10.0
    st = st.cuda().eval()
    for N in (100, 256, 1000, 2048):
        for B in (1, 4):
            rs = np.random.RandomState(N + B)
            x = torch.from_numpy(rs.randn(B, N, d).astype(np.float32)).cuda()
            t = torch.from_numpy(rs.randn(B, 1, 1).astype(np.float32)).cuda()
            outs = {}
            for mode in ("fp32", "bf16x3", "mixed", "fp16"):
                hip_ops.set_default_precision(mode)
                try:
                    with torch.no_grad():
                        y = st(x, t)
                    outs[mode] = (y[0] if isinstance(y, tuple) else y).float()
                except Exception as e:  # noqa: BLE001
                    print(f"d={d} act={act.__name__[:5]} I={I} H={H} mult={mult} N={N} B={B} {mode} FAILED: {str(e)[:200]}")
                    bad += 1
            ref = outs.get("fp32")
            line = f"d={d:3d} act={act.__name__[:5]} I={I:2d} H={H:2d} mult={mult} N={N:4d} B={B}"
            for mode in ("bf16x3", "mixed", "fp16"):
                if mode not in outs or ref is None:
                    continue
                e = float((outs[mode] - ref).abs().max() / ref.abs().max())
                flag = "" if (e < BARS[mode] and np.isfinite(e)) else "<--OUTLIER"
                bad += bool(flag)
                line += f"  {mode} {e:.1e}{flag}"
            print(line, flush=True)
hip_ops.set_default_precision("mixed")
print("outliers / failures:", bad)
