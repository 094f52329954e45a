"""mlp.0 of the mixed mode at the C2 shape: the h8 A-stationary kernel (gemm_h8_astat.hip) against the split-bf16 LDS-DMA
kernel it replaces, both writing the tiled split image; hipGraph replays, HIP events.  python tools/h8_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from gecco_amd import hip_ops as ops  # noqa: E402

B, N, D = 64, 2048, 384
dev = "cuda"
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
x = rn(B, N, D)
pa, po = 1 + 0.1 * rn(B, D), 0.1 * rn(B, D)
W1, b1 = rn(2 * D, D) / 20, rn(2 * D) / 20
alpha = torch.tensor(1.0, device=dev)
ws = torch.empty(2 * D * D * 4, dtype=torch.uint8, device=dev)
img = ops.linear_h8_img(x, (pa, po), W1, b1, act_alpha=alpha, wsplit=ws)
ref = torch.exp(-(torch.nn.functional.linear(x.double() * pa[:, None] + po[:, None], W1.double(), b1.double())) ** 2 / 2)
ref = (ref - 0.7) / 0.28
got = ops.decode_split_image(img).double()
print("h8 max-rel err vs fp64:", float((got - ref).abs().max() / ref.abs().max()))
hid = torch.empty(B, N, 2 * D, device=dev)
y3 = ops.linear(x, W1, b1, (pa, po), act_alpha=alpha, out=hid, precision="bf16x3")
print("x3 max-rel err vs fp64:", float((y3.double() - ref).abs().max() / ref.abs().max()))


def timed(fn, reps=10, iters=5):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        gr.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / (reps * iters) * 1e3


t_h8 = timed(lambda: ops.linear_h8_img(x, (pa, po), W1, b1, act_alpha=alpha, wsplit=ws, image_ready=True, out=img))
t_x3 = timed(lambda: ops.linear(x, W1, b1, (pa, po), act_alpha=alpha, out=hid, precision="bf16x3"))
fl = 2 * B * N * D * 2 * D
print(f"h8  mlp.0: {t_h8:7.1f} us  {fl / t_h8 / 1e6:6.1f} TFLOP/s of 2MNK")
print(f"x3  mlp.0: {t_x3:7.1f} us  {fl / t_x3 / 1e6:6.1f} TFLOP/s of 2MNK (LDS-DMA kernel, fp32 output, weight split included)")

# kv_proj | q_proj: the 64-column-tile kernel against the 128-column-tile A-stationary kernel (one-term form; the mixed mode's
# lo image is built inside the network entry point only)
H = 8
Wkv, Wq, bq = rn(2 * D, D) / 20, rn(D, D) / 20, rn(D) / 20
ws2 = torch.empty(3 * D * D * 2 + D * D, dtype=torch.uint8, device=dev)
ops.linear_kvq_f16(x, (pa, po), Wkv, None, Wq, bq, lo=(D, 2 * D), head_dim=D // H, wsplit=ws2)
t_kvq = timed(lambda: ops.linear_kvq_f16(x, (pa, po), Wkv, None, Wq, bq, lo=(D, 2 * D), head_dim=D // H, wsplit=ws2, image_ready=True))
t_kvq1 = timed(lambda: ops.linear_kvq_f16(x, (pa, po), Wkv, None, Wq, bq, lo=(0, 0), head_dim=D // H, wsplit=ws2, image_ready=False))
ws3 = torch.empty(3 * D * D * 4, dtype=torch.uint8, device=dev)
kv16 = torch.empty(B, 2 * H, N, D // H, device=dev, dtype=torch.float16)
q16 = torch.empty(B, H, N, D // H, device=dev, dtype=torch.float16)
ops.linear_astat_f16(x, (pa, po), Wkv, None, Wq, bq, out=(kv16, q16), head_dim=D // H, wsplit=ws3)
t_as = timed(lambda: ops.linear_astat_f16(x, (pa, po), Wkv, None, Wq, bq, out=(kv16, q16), head_dim=D // H, wsplit=ws3, image_ready=True))
fl = 2 * B * N * D * 3 * D
print(f"kvq64 (V two-term): {t_kvq:7.1f} us   kvq64 one-term (+ image build): {t_kvq1:7.1f} us   astat128 one-term: {t_as:7.1f} us   ({fl / t_kvq / 1e6:.0f} TFLOP/s of 2MNK; 502 MB -> {502e6 / t_kvq / 1e6:.2f} TB/s)")

# mlp.2 and out_proj: the h8 register-fed kernel against the split-bf16 register-fed kernel is only reachable inside the network;
# here the h8 pair mlp.0 -> mlp.2 on shared buffers
W2, b2 = rn(D, 2 * D) / 28, rn(D) / 20
img2 = ops.linear_h8_img(x, (pa, po), W1, b1, act_alpha=alpha, wsplit=ws, kind=2)
wsb = torch.empty(D * 2 * D * 4, dtype=torch.uint8, device=dev)
xo = x.clone()
ops.linear_h8_areg(img2, W2, b2, residual=xo, out=xo, wsplit=wsb)
t_m0 = timed(lambda: ops.linear_h8_img(x, (pa, po), W1, b1, act_alpha=alpha, wsplit=ws, image_ready=True, out=img2, kind=2))
t_pair = timed(lambda: (ops.linear_h8_img(x, (pa, po), W1, b1, act_alpha=alpha, wsplit=ws, image_ready=True, out=img2, kind=2),
                        ops.linear_h8_areg(img2, W2, b2, residual=xo, out=xo, wsplit=wsb, image_ready=True)))
Wo = rn(D, D) / 20
wsc = torch.empty(D * D * 4, dtype=torch.uint8, device=dev)
att = img2[:, :, :D // 64].contiguous()
ops.linear_h8_areg(att, Wo, b2, residual=xo, out=xo, wsplit=wsc)
t_op = timed(lambda: ops.linear_h8_areg(att, Wo, b2, residual=xo, out=xo, wsplit=wsc, image_ready=True))
print(f"h8 mlp.0 (h8 image): {t_m0:7.1f} us   mlp.0 + mlp.2 pair: {t_pair:7.1f} us  -> mlp.2 {t_pair - t_m0:7.1f} us   out_proj (K = {D}): {t_op:7.1f} us")
