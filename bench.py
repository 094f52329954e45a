#!/usr/bin/env python3
"""Benchmark of the GECCO denoiser forward on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A "step" is one Diffusion.forward (EDMPrecond + LinearLift + 6-layer set transformer) over one batch
of synthetic clouds resident in HBM: config C2 = B=64, N=2048, d=384, L=6, I=64, H=8.  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank evaluates its own batch — samples
are independent, there is no data-path collective ("replicas only", weak scaling) — and the time is
the max over ranks.  Rank 0 prints ONE JSON line.  Started as plain `python bench.py --gpus N` (no
torch.distributed.run around it) the script spawns the N ranks itself, as fresh child processes, before it
touches the GPU.

Default arithmetic: `--precision mixed` — fp16 operands where their rounding does not reach the output (the
kv_proj | q_proj activations, K | V, q, both attention products; the V projection's weights carry an fp8 second term), and
two terms on BOTH operands for every product that feeds the residual stream or the shared inducer states: out_proj and the
point MLP as an fp16 main product plus two fp8 cross terms ("h8": 2 matrix-pipe units per product at split-bf16 accuracy,
round 3), the 64-inducer chain with two-term fp16 weights — the cheapest recipe of the per-site search (profiles/r03_precision_search.txt)
that holds BOTH outputs of the network (denoised D and raw F_x) within the 1e-3 parity bar with a >= 10x margin at every
BASELINE shape (tests/test_hip_fullsize.py: F_x 5e-5 .. 8e-5; <= 1.6e-4 on 14-layer networks).  `--precision bf16x3` is split-bf16 everywhere
(F_x 3e-5 .. 5e-5); `--precision fp16` is the faster opt-in mode: D within 1e-3 (2.5x margin) but F_x AT the bar
(0.9e-3 .. 1.2e-3 at L=6, N=2048), so it is not the headline.

`--train` times the data-parallel TRAINING step instead: per rank batch 48 (the shipped
config), forward + backward on the HIP training path, gradient all-reduce overlapped with the backward
(one RCCL all-reduce per ~8 MB bucket of the flat gradient buffer, issued as the bucket completes), fused
Adam + EMA step (one launch); it reports step ms, the stand-alone all-reduce time and bus bandwidth.  `--train --config
C3 | C4` is the image-conditional step (frozen channels-last ConvNeXt conditioner inside the step, projective lookup,
RayNetwork; C4 = BASELINE's data-parallel configuration: N = 4096, d = 512).  `--config C3 | C4 | C5` without `--train`
times one evaluation of the other BASELINE shapes (C5: the cached upsampling evaluation of 16 384 new points).

Extra objects on that line:
  roofline     — the dominant kernel of the measured mode.  mixed (default): mlp.0 on the A-stationary h8 kernel
                 (gemm_h8_astat_kernel: AdaGN apply + fp16 main product + two fp8 cross terms + GaussianActivation, 20 % of device
                 time), bound "mfma" against 2500 / 1.5 TFLOP/s of 2MNK (1.5 matrix-pipe units per product with the fp6 cross terms; 2 with GECCO_H6=0), its HBM side beside it
                 ("hbm": the kernel sits at the ridge), timed with HIP events inside hipGraph replays of the round out_proj ->
                 mlp.0 -> mlp.2 on shared buffers ("round - round without that launch"); "traffic" / "mfma_busy_pmc" = that
                 kernel's FETCH_SIZE x 2 + WRITE_SIZE and matrix-pipe busy fraction from the committed --pmc passes
                 (profiles/gemm_hbm_traffic.json <- profiles/r03p_forward_pmc_summary.txt); "whole" = the whole evaluation:
                 executed matrix-pipe work / time / 2500 TFLOP/s and counter bytes / time / 8 TB/s.  fp16 (opt-in): the fused
                 point MLP (mlp_fused_f16_kernel), same method.  bf16x3 / fp32: the LDS-DMA GEMM at its four call-site shapes.
                 The rocprofv3 --kernel-trace --stats summary of the same model lives in profiles/.
  train, configs, upsample — what else the tree does, measured by child processes of the default run (`--no-extras` skips
                 them): the C2 training step with its dominant kernel's own roofline fraction, the C3 / C4 / C5 forward shapes,
                 Diffusion.upsample per outer step (eager and captured).
  cpu_baseline — the oracle (plain PyTorch CPU restatement of the reference) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, N, D, L, I, H = 64, 2048, 384, 6, 64, 8
PEAK_F32_MFMA_TFLOPS = 157.3     # dense fp32 MFMA (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0            # HBM3E spec (6.3 TB/s measured achievable)


def flops_per_sample():
    per_layer = 16 * N * D * D + 8 * N * I * D + 14 * I * D * D
    return L * per_layer + 2 * (2 * N * 3 * D)


def random_state_dict(seed, D=D, L=L):
    """Random-init weights of the C2 architecture (or another width / depth), keyed like the reference state dict
    (LinearLift: lift / inner.layers.i.* / lower.1); AdaGN and alpha deliberately non-default."""
    g = torch.Generator().manual_seed(seed)
    u = lambda o, i_: (torch.rand(o, i_, generator=g) * 2 - 1) / i_ ** 0.5
    ub = lambda o, i_: (torch.rand(o, generator=g) * 2 - 1) / i_ ** 0.5
    sd = {"lift.weight": u(D, 3), "lift.bias": ub(D, 3), "lower.1.weight": u(3, D), "lower.1.bias": ub(3, D)}

    def adagn(pre):
        sd[pre + "scale.weight"] = 0.2 * torch.randn(D, 1, generator=g)
        sd[pre + "scale.bias"] = 1 + 0.1 * torch.randn(D, generator=g)
        sd[pre + "bias.weight"] = 0.2 * torch.randn(D, 1, generator=g)
        sd[pre + "bias.bias"] = 0.1 * torch.randn(D, generator=g)

    def mlp(pre):
        sd[pre + "0.weight"], sd[pre + "0.bias"] = u(2 * D, D), ub(2 * D, D)
        sd[pre + "1.alpha"] = torch.tensor(1.0) + 0.1 * torch.randn((), generator=g)
        sd[pre + "2.weight"], sd[pre + "2.bias"] = u(D, 2 * D), ub(D, 2 * D)

    for li in range(L):
        pre = f"inner.layers.{li}."
        adagn(pre + "broadcast_norm.")
        sd[pre + "broadcast.pool.inducers"] = torch.randn(1, H, I, D // H, generator=g)
        sd[pre + "broadcast.pool.kv_proj.weight"] = u(2 * D, D)
        sd[pre + "broadcast.pool.out_proj.weight"] = u(D, D)
        adagn(pre + "broadcast.norm_1.")
        mlp(pre + "broadcast.mlp.")
        adagn(pre + "broadcast.norm_2.")
        sd[pre + "broadcast.unpool.in_proj_weight"] = u(3 * D, D)
        sd[pre + "broadcast.unpool.in_proj_bias"] = ub(3 * D, D)
        sd[pre + "broadcast.unpool.out_proj.weight"] = u(D, D)
        sd[pre + "broadcast.unpool.out_proj.bias"] = ub(D, D)
        adagn(pre + "mlp_norm.")
        mlp(pre + "mlp.")
    return {k: v.float().contiguous() for k, v in sd.items()}


def synthetic_cloud(seed):
    """unit-variance data, stratified log-uniform sigma in [0.002, 165], x = data + sigma*noise."""
    g = torch.Generator().manual_seed(seed)
    import math
    data = torch.randn(B, N, 3, generator=g)
    u = (torch.arange(B) + torch.rand(B, generator=g)) / B
    sigma = torch.exp(math.log(0.002) + u * (math.log(165.0) - math.log(0.002)))
    return (data + sigma[:, None, None] * torch.randn(B, N, 3, generator=g)).contiguous(), sigma.float().contiguous()


def build_model(p_cpu):
    """The shipped unconditional architecture (example_configs/shapenet_airplane_unconditional.py:23-57) built from
    the drop-in modules, with the random-init state dict loaded strict=True."""
    from gecco_amd.diffusion import Diffusion, EDMLoss, EDMPrecond, IdleConditioner, LogUniformSchedule
    from gecco_amd.models.activation import GaussianActivation
    from gecco_amd.models.linear_lift import LinearLift
    from gecco_amd.models.set_transformer import SetTransformer
    from gecco_amd.reparam import GaussianReparam
    net = LinearLift(inner=SetTransformer(n_layers=L, num_inducers=I, feature_dim=D, t_embed_dim=1, num_heads=H,
                                          activation=GaussianActivation), feature_dim=D)
    m = Diffusion(backbone=EDMPrecond(model=net), conditioner=IdleConditioner(),
                  reparam=GaussianReparam(torch.tensor([0.0, 0.01, 0.05]), torch.tensor([0.11, 0.04, 0.17])),
                  loss=EDMLoss(schedule=LogUniformSchedule(max=165.0)))
    sd = {"backbone.model." + k: v for k, v in p_cpu.items()}
    sd["reparam.mean"], sd["reparam.sigma"] = m.reparam.mean, m.reparam.sigma
    m.load_state_dict(sd, strict=True)
    return m


def gemm_call_sites(ops, dev, precision="fp32"):
    """The four (B*N)-row GEMM launches of one layer (kv_proj and the unpool q projection share one), as closures
    launching the unit operator.  Each entry: (name, algorithmic FLOPs = 2 M N K, algorithmic HBM bytes, closure)."""
    g = torch.Generator(device="cpu").manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    x, big = rn(B, N, D), rn(B, N, 2 * D)
    pa, po = 1 + 0.1 * rn(B, D), 0.1 * rn(B, D)
    Wkv, Wq, Wo, W1, W2 = rn(2 * D, D) / 20, rn(D, D) / 20, rn(D, D) / 20, rn(2 * D, D) / 20, rn(D, 2 * D) / 28
    bq, b1, b2 = rn(D) / 20, rn(2 * D) / 20, rn(D) / 20
    alpha = torch.tensor(1.0, device=dev)
    o768, o384, q384 = torch.empty(B, N, 2 * D, device=dev), torch.empty(B, N, D, device=dev), torch.empty(B, N, D, device=dev)
    res = x.clone()
    S = B * N * D * 4  # bytes of one (B, N, d) fp32 stream
    if precision == "fp16":
        # fp16 mode: AdaGN(x), K|V, q, the attention output and the MLP hidden layer are fp16 TENSORS
        att16, big16 = rn(B, N, D).half(), rn(B, N, 2 * D).half()
        # K | V and q head-major, as the network stores them: (B, 2H, N, hd) and (B, H, N, hd)
        kv16 = torch.empty(B, 2 * H, N, D // H, device=dev, dtype=torch.float16)
        q16 = torch.empty(B, H, N, D // H, device=dev, dtype=torch.float16)
        h16 = torch.empty(B, N, 2 * D, device=dev, dtype=torch.float16)
        H2 = S // 2   # bytes of one (B, N, d) fp16 stream
        # the three point-stream launches of a layer as the network runs them: AdaGN + kv|q
        # (A-stationary), unpool attention + out_proj + residual + statistics, AdaGN + mlp.0 + activation + mlp.2 +
        # residual + statistics.  Bytes: what has to cross HBM once (x in / x out counted once each).
        # The three closures work on shared buffers (kv|q reads the x the MLP wrote and writes the q the unpool launch
        # reads, which writes the x the MLP reads); no allocation inside, so they can be captured (time_graphed).
        kvh = rn(B, I, 2 * D)
        xw = x.clone()   # updated in place by the fused launches (bounded: the coefficients and weights are fixed)
        # weight images made once (as the network does once per forward): the timed calls launch the kernels alone
        ws = [torch.empty(n, dtype=torch.uint8, device=dev) for n in (3 * D * D * 4, 2 * D * D, 4 * D * 2 * D)]
        st1, st2 = (torch.empty(B, N // 128, 2, D, device=dev) for _ in range(2))
        ops.linear_astat_f16(xw, (pa, po), Wkv, None, Wq, bq, out=(kv16, q16), head_dim=D // H, wsplit=ws[0])
        ops.unpool_outproj_f16(xw, q16, kvh, Wo, bq, H, wsplit=ws[1], stats=st1)
        ops.mlp_fused_f16(xw, (pa, po), W1, b1, W2, b2, act_alpha=alpha, wsplit=ws[2], stats=st2)
        return [
            ("norm+kv_proj|q_proj", 2 * B * N * D * 3 * D, S + 3 * H2,
             lambda: ops.linear_astat_f16(xw, (pa, po), Wkv, None, Wq, bq, out=(kv16, q16), head_dim=D // H, wsplit=ws[0], image_ready=True)),
            ("unpool_attn+out_proj+res+stats", 2 * B * N * D * D + 4 * B * N * I * D, H2 + 2 * S,
             lambda: ops.unpool_outproj_f16(xw, q16, kvh, Wo, bq, H, wsplit=ws[1], image_ready=True, stats=st1)),
            ("norm+mlp.0+act+mlp.2+res+stats", 4 * B * N * D * 2 * D, 2 * S,
             lambda: ops.mlp_fused_f16(xw, (pa, po), W1, b1, W2, b2, act_alpha=alpha, wsplit=ws[2], image_ready=True, stats=st2)),
        ]
    pr = dict(precision=precision)
    sites = [
        ("kv_proj|q_proj", 2 * B * N * D * 3 * D, S + 3 * S, lambda: ops.linear_pair(x, Wkv, None, Wq, bq, (pa, po), out=(o768, q384), **pr)),
        ("out_proj+res+stats", 2 * B * N * D * D, 3 * S, lambda: ops.linear(x, Wo, bq, residual=res, want_stats=True, out=o384, **pr)),
        ("mlp.0+act", 2 * B * N * D * 2 * D, S + 2 * S, lambda: ops.linear(x, W1, b1, (pa, po), act_alpha=alpha, out=o768, **pr)),
        ("mlp.2+res+stats", 2 * B * N * 2 * D * D, 2 * S + 2 * S, lambda: ops.linear(big, W2, b2, residual=res, want_stats=True, out=o384, **pr)),
    ]
    return sites


def split_bf16_round(ops, dev):
    """out_proj -> mlp.0 -> mlp.2 of one layer as the split-bf16 unit operator on SHARED buffers (x updated in place, the
    hidden layer handed from mlp.0 to mlp.2), so that inside a replayed round each launch meets the cache state its
    predecessor leaves, as in the forward.  Entries as gemm_call_sites: (name, FLOPs, algorithmic HBM bytes, closure).
    Launches of the whole batch on ONE stream — the kernel on its own, as `GECCO_FWD_STREAMS=1 bench.py` runs it and as
    profiles/r02zc_bench_kernel_stats.csv shows it; the default evaluation issues the same kernels for two half batches on two
    streams (hip_ops._two_stream_halves), where their durations overlap and no per-kernel time exists."""
    g = torch.Generator(device="cpu").manual_seed(2)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    xw, att, hid = rn(B, N, D), rn(B, N, D), torch.empty(B, N, 2 * D, device=dev)
    pa, po = 1 + 0.1 * rn(B, D), 0.1 * rn(B, D)
    Wo, W1, W2 = rn(D, D) / 40, rn(2 * D, D) / 20, rn(D, 2 * D) / 56
    bo, b1, b2 = rn(D) / 20, rn(2 * D) / 20, rn(D) / 20
    alpha = torch.tensor(1.0, device=dev)
    S = B * N * D * 4
    pr = dict(precision="bf16x3")
    return [
        ("out_proj+res", 2 * B * N * D * D, 3 * S, lambda: ops.linear(att, Wo, bo, residual=xw, out=xw, **pr)),
        ("mlp.0+act", 2 * B * N * D * 2 * D, S + 2 * S, lambda: ops.linear(xw, W1, b1, (pa, po), act_alpha=alpha, out=hid, **pr)),
        ("mlp.2+res", 2 * B * N * 2 * D * D, 2 * S + 2 * S, lambda: ops.linear(hid, W2, b2, residual=xw, out=xw, **pr)),
    ]


def h8_round(ops, dev):
    """out_proj -> mlp.0 -> mlp.2 of one layer as the mixed mode runs them since round 3 (h8 arithmetic: fp16 main product + two fp8
    cross terms): out_proj and mlp.2 on the register-fed kernel (gemm_h8_areg.hip) reading h8 activation images, mlp.0 on the
    A-stationary kernel (gemm_h8_astat.hip) writing one — on SHARED buffers (x updated in place, the hidden image handed from
    mlp.0 to mlp.2), image-ready calls = kernel launches only, so that inside a replayed round each launch meets the cache
    state its predecessor leaves.  Entries: (name, 2MNK FLOPs, algorithmic HBM bytes, closure)."""
    g = torch.Generator(device="cpu").manual_seed(2)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    xw = rn(B, N, D)
    pa, po = 1 + 0.1 * rn(B, D), 0.1 * rn(B, D)
    Wo, W1, W2 = rn(D, D) / 40, rn(2 * D, D) / 20, rn(D, 2 * D) / 56
    bo, b1, b2 = rn(D) / 20, rn(2 * D) / 20, rn(D) / 20
    alpha = torch.tensor(1.0, device=dev)
    ws0, ws1, ws2 = (torch.empty(n, dtype=torch.uint8, device=dev) for n in (D * D * 4, 2 * D * D * 4, D * 2 * D * 4))
    hid = ops.linear_h8_img(xw, (pa, po), W1, b1, act_alpha=alpha, wsplit=ws1, kind=2)          # (B, N / 128, 2D / 64, 24576) bytes
    att = ops.linear_h8_img(xw, None, Wo, None, wsplit=ws0, kind=2)                              # stands in for the attention output image
    st = torch.empty(B, N // 128, 2, D, device=dev)
    ops.linear_h8_areg(att, Wo, bo, residual=xw, out=xw, wsplit=ws0)
    ops.linear_h8_areg(hid, W2, b2, residual=xw, out=xw, wsplit=ws2)
    S, S16 = B * N * D * 4, B * N * D * 3    # bytes of an fp32 (B, N, d) stream / of a d-wide h8 image
    return [
        ("out_proj+res (h8)", 2 * B * N * D * D, S16 + 2 * S, lambda: ops.linear_h8_areg(att, Wo, bo, residual=xw, out=xw, wsplit=ws0, image_ready=True)),
        ("mlp.0+act (h8)", 2 * B * N * D * 2 * D, S + 2 * S16,
         lambda: ops.linear_h8_img(xw, (pa, po), W1, b1, act_alpha=alpha, wsplit=ws1, image_ready=True, out=hid, kind=2)),
        ("mlp.2+res (h8)", 2 * B * N * 2 * D * D, 2 * S16 + 2 * S, lambda: ops.linear_h8_areg(hid, W2, b2, residual=xw, out=xw, wsplit=ws2, image_ready=True)),
    ]


def w2_round(ops, dev):
    """The large launches of one layer as the "w2" mode runs them, kernel launches only (weight images / streams ready), on shared buffers:
    out_proj on the register-fed h8 kernel (what the mixed mode shares with it) and the point MLP as ONE launch (mlp_fused_w.hip).
    Entries: (name, 2MNK FLOPs, algorithmic HBM bytes, closure)."""
    g = torch.Generator(device="cpu").manual_seed(2)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    xw = rn(B, N, D)
    pa, po = 1 + 0.1 * rn(B, D), 0.1 * rn(B, D)
    Wo, W1, W2 = rn(D, D) / 40, rn(2 * D, D) / 20, rn(D, 2 * D) / 56
    bo, b1, b2 = rn(D) / 20, rn(2 * D) / 20, rn(D) / 20
    alpha = torch.tensor(1.0, device=dev)
    lib = ops._lib.load()
    ws0 = torch.empty(D * D * 4, dtype=torch.uint8, device=dev)
    wsm = torch.empty(lib.gecco_mlp_fused_w_wsplit_bytes(D, 2 * D), dtype=torch.uint8, device=dev)
    att = ops.linear_h8_img(xw, None, Wo, None, wsplit=ws0, kind=2)
    st = torch.empty(B, N // 128, 2, D, device=dev)
    ops.linear_h8_areg(att, Wo, bo, residual=xw, out=xw, wsplit=ws0)
    ops.mlp_fused_w(xw, (pa, po), W1, b1, W2, b2, act_alpha=alpha, wsplit=wsm, stats=st)
    S, S16 = B * N * D * 4, B * N * D * 3
    return [
        ("out_proj+res (h8)", 2 * B * N * D * D, S16 + 2 * S, lambda: ops.linear_h8_areg(att, Wo, bo, residual=xw, out=xw, wsplit=ws0, image_ready=True)),
        ("norm+mlp.0+act+mlp.2+res+stats (w2, one launch)", 2 * B * N * D * 2 * D * 2, 2 * S,
         lambda: ops.mlp_fused_w(xw, (pa, po), W1, b1, W2, b2, act_alpha=alpha, wsplit=wsm, image_ready=True, stats=st)),
    ]


def time_events(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters  # ms


def time_graph_of(fns, reps=8, iters=4):
    """ms per replayed round of the launch sequence `fns` inside a hipGraph of `reps` rounds (HIP events around the
    replays, on the stream the graph is replayed on): no host launch gap inside the timed region — an event pair around
    eager launches adds ~20 us per launch here — i.e. the condition the kernels have in the captured forward."""
    for fn in fns:
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for fn in fns:
                fn()
    g.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        g.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / (iters * reps)


def time_in_sequence(fns):
    """Duration of each launch of the sequence `fns` IN the sequence (so that it meets the cache state its predecessor
    leaves, as in the forward): the plain round minus the round without launch k."""
    base = time_graph_of(fns)
    return [base - time_graph_of(fns[:k] + fns[k + 1:]) for k in range(len(fns))], base


def build_info():
    """What the loaded library was built from (__graft_entry__.build() records a content hash of sources + flags beside the .so)."""
    import __graft_entry__ as ge
    info = {"sources_sha": ge.built_sources_sha(), "tree_sources_sha": ge.sources_sha()}
    info["matches_tree_sources"] = info["sources_sha"] == info["tree_sources_sha"]
    try:
        info["tree_commit"] = open(os.path.join(ROOT, "TREE_COMMIT")).read().strip()
    except OSError:
        info["tree_commit"] = None
    return info


def load_traffic(mode):
    """The committed counter constants of `mode` (profiles/gemm_hbm_traffic.json, written by tools/pmc_collect.sh with PMC_WRITE=1) —
    ONLY when they were collected on the kernel sources this library was built from (their `sources_sha` = the build's): otherwise
    ({}, stale-record) — the line then carries `"traffic_stale": true` and no counter figure instead of a number from another tree."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "gemm_hbm_traffic.json"))).get(mode, {})
    except Exception:
        return {}, {"traffic_stale": True, "why": "profiles/gemm_hbm_traffic.json unreadable"}
    built = build_info()["sources_sha"]
    meta = {"traffic_tree": {"sources_sha": d.get("sources_sha"), "commit": d.get("tree"), "source": d.get("source")},
            "traffic_stale": not (built and d.get("sources_sha") == built)}
    if meta["traffic_stale"]:
        meta["why"] = f"counters collected on kernel sources {d.get('sources_sha')}, this library is built from {built}"
        return {}, meta
    return d, meta


def cpu_baseline(p, x, sigma):
    """The oracle (plain PyTorch CPU restatement of the reference, verified equal to it) on the host cores.  Two figures: the
    FULL C2 batch with one torch thread per physical core (BASELINE.md section 4's definition; on a 128-core box this
    oversubscribes the 64-cloud batch's small GEMMs and is SLOWER than fewer threads), and beside it `best_of_threads`: the best
    of a {16, 32, 64} thread sweep on 8 of the clouds — what a user of the reference would tune to.  Bounded: ~10-30 s in all."""
    from oracle import cpu_ref
    cores = os.cpu_count() or torch.get_num_threads()
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    xs, ss = x.cpu(), sigma.cpu()
    Dn = cpu_ref.uncond_denoiser({k: v.cpu() for k, v in p.items()}, "", H)
    sweep = {}
    with torch.no_grad():
        for th in sorted({t for t in (16, 32, 64) if t <= cores} or {cores}):
            torch.set_num_threads(th)
            Dn(xs[:2], ss[:2])                      # warm-up of the thread pool at this size
            t0 = time.perf_counter()
            Dn(xs[:8], ss[:8])
            sweep[th] = 8 * N / (time.perf_counter() - t0)
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        Dn(xs, ss)  # warm-up
        warm = time.perf_counter() - t0
        iters = 2 if warm < 12 else 1
        t0 = time.perf_counter()
        for _ in range(iters):
            Dn(xs, ss)
        dt = time.perf_counter() - t0
    best_th = max(sweep, key=sweep.get)
    return {"value": B * N * iters / dt, "unit": "points/s", "cores": cores, "kind": "port",
            "sample": f"oracle/cpu_ref.py fp32 forward on all {B} clouds (N={N}, d={D}, L={L}), torch threads = {cores} "
                      f"physical cores, 1 warm-up + {iters} timed iterations, {dt:.1f} s",
            "best_of_threads": {"value": sweep[best_th], "threads": best_th, "sample": "8 of the clouds, one pass per thread count",
                                "sweep_points_per_sec": {str(k): round(v, 1) for k, v in sweep.items()}}}


def spawn_ranks(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (the parent has not
    touched the GPU and never does), one per GPU, with the torch.distributed.run environment; rank 0's stdout (the one
    JSON line) is the parent's.  Returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    try:
        for pr in procs:
            rc = max(rc, abs(pr.wait()))
    finally:
        for pr in procs:   # a failed rank must not leave the others blocked in a collective
            if pr.poll() is None:
                pr.kill()
    return rc


def distributed_report(dev, local_points_per_sec=None, sizes=(53_900_000, 8_388_608)):
    """What the FIRST multi-GPU run should say about itself (the build container has one GPU; the driver's 8-GPU node runs this
    unattended): the ranks the process group really has, the collective library, the bus bandwidth of the gradient all-reduce at
    the sizes the training step issues (the whole 53.9 MB flat gradient buffer; one ~8 MB bucket — xGMI is point-to-point, a ring
    all-reduce is per-link bound: bus bandwidth = 2 (n - 1) / n x bytes / t), and this rank's own un-barriered rate, so that
    the N = 1 line of a scaling table can be checked against the single-GPU bench.  Collective call on every rank; the dict is
    meaningful on rank 0.  Reference: gecco-jax models/diffusion.py:571-573 (pmean), Lightning's implicit DDP."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rep = {"backend": dist.get_backend() if dist.is_initialized() else None, "ranks_in_group": world,
           "collective_library": None, "allreduce": {}, "rank0_points_per_sec_unbarriered": local_points_per_sec,
           "per_rank_points_per_sec": None}
    on_gpu = dev is not None and torch.device(dev).type == "cuda"
    if local_points_per_sec is not None:
        # every rank's own un-barriered rate (rank order): the N = 1 entry of a scaling table is directly comparable with the single-GPU
        # bench line, and a straggler GPU shows up by name instead of inside a max-over-ranks time
        if dist.is_initialized() and world > 1:
            mine = torch.tensor([float(local_points_per_sec)], dtype=torch.float64, device=dev if on_gpu else "cpu")
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            rep["per_rank_points_per_sec"] = [float(t.item()) for t in allr]
        else:
            rep["per_rank_points_per_sec"] = [float(local_points_per_sec)]
    if on_gpu:
        try:
            rep["collective_library"] = "RCCL/NCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:   # noqa: BLE001
            rep["collective_library"] = f"unknown ({e!r})"[:80]
    if dist.is_initialized():
        for nbytes in sizes:
            n = max(nbytes // 4, 1)
            buf = torch.ones(n, dtype=torch.float32, device=dev if on_gpu else "cpu")
            for _ in range(2):
                dist.all_reduce(buf)
            if on_gpu:
                torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            iters = 5
            for _ in range(iters):
                dist.all_reduce(buf)
            if on_gpu:
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters
            rep["allreduce"][f"{nbytes} B"] = {"ms": dt * 1e3,
                                               "busbw_gbs": 2 * (world - 1) / world * n * 4 / dt / 1e9 if world > 1 else None}
    return rep


def launcher_selftest(args):
    """`--selftest-launcher`: the N > 1 plumbing alone (spawn, rendezvous, barrier, max-over-ranks timing, one JSON line
    from rank 0) on the gloo backend with a stand-in step — no GPU, no compute path; covered by tests/."""
    from gecco_amd import distributed as gd
    rank, world = gd.init("gloo")
    gd.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))
    local = time.perf_counter() - t0          # this rank's own time: before the barrier, as in the compute paths
    gd.barrier()
    dt = gd.max_over_ranks(time.perf_counter() - t0)
    rep = distributed_report(None, local_points_per_sec=args.steps / local, sizes=(4096, 1024)) if world > 1 else None
    if rank == 0:
        print(json.dumps({"metric": "launcher_selftest", "n_gpus": world, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3,
                          "distributed": rep}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def train_bench(args, rank, world, dev):
    """Data-parallel training step (reference: Lightning DDP over training_step + Adam + EMACallback,
    example_configs/shapenet_airplane_unconditional.py:59-77, diffusion.py:210-222, ema.py:273-325)."""
    import torch.distributed as dist
    from gecco_amd import distributed as gd
    from gecco_amd import hip_ops as ops
    from gecco_amd.optim import FusedAdamEMA
    from gecco_amd.structs import Example
    ops.set_default_precision(args.precision)
    Bt = args.train_batch
    g = torch.Generator().manual_seed(100 + rank)      # every rank: its own shard of the global batch
    Nt, Dt, Lt, cond = N, D, L, args.config in ("C3", "C4")
    if cond:
        # image-conditional training (BASELINE config C4: Taskonomy 256 x 256, N = 4096, d = 512, DDP; C3: 224 x 224, N = 2048,
        # d = 384): channels-last ConvNeXt conditioner + projective lookup + RayNetwork, ALL trained through the HIP autograd
        # path like the reference's training_step (the conditioner is a sub-module of Diffusion: diffusion.py:210-211
        # optimises self.parameters()); --freeze-conditioner evaluates it without gradients instead
        from gecco_amd.diffusion import Diffusion, EDMLoss, EDMPrecond, LogUniformSchedule
        from gecco_amd.models.activation import GaussianActivation
        from gecco_amd.models.feature_pyramid import ConvNeXtExtractor
        from gecco_amd.models.ray import RayNetwork
        from gecco_amd.models.set_transformer import SetTransformer
        from gecco_amd.reparam import UVLReparam
        from gecco_amd.structs import Context3d
        Nt, Dt, hw = (2048, 384, 224) if args.config == "C3" else (4096, 512, 256)
        torch.manual_seed(3)
        rp = UVLReparam(torch.tensor([0.0, 0.0, 1.38]), torch.tensor([0.56, 0.60, 0.49]))
        net = RayNetwork(backbone=SetTransformer(n_layers=Lt, num_inducers=I, feature_dim=Dt, t_embed_dim=1, num_heads=H,
                                                 activation=GaussianActivation), reparam=rp, context_dims=(96, 192, 384))
        model = Diffusion(backbone=EDMPrecond(model=net), conditioner=ConvNeXtExtractor(pretrained=False), reparam=rp,
                          loss=EDMLoss(schedule=LogUniformSchedule(max=180.0))).to(dev).train()
        if args.freeze_conditioner:
            model.conditioner.requires_grad_(False)
        K = torch.zeros(Bt, 3, 3)
        K[:, 0, 0] = K[:, 1, 1] = 1.1
        K[:, 0, 2] = K[:, 1, 2] = 0.5
        K[:, 2, 2] = 1.0
        ctx = Context3d(image=torch.rand(Bt, 3, hw, hw, generator=g).to(dev), K=K.to(dev))
        data = model.reparam.diffusion_to_data(torch.randn(Bt, Nt, 3, generator=g).to(dev), ctx)   # clouds in front of the camera
        example = Example(data, ctx)
        params = [q for q in model.parameters() if q.requires_grad]
    else:
        model = build_model(random_state_dict(seed=3)).to(dev).train()
        data = (torch.randn(Bt, N, 3, generator=g) * model.reparam.sigma.cpu() + model.reparam.mean.cpu()).to(dev)
        example = Example(data, None)
        params = list(model.parameters())
    gd.broadcast_parameters(model)
    opt = FusedAdamEMA(params, lr=1e-4, ema_decay=0.99, amp_on_device=bool(args.amp))   # --amp: scale / found_inf stay on the device
    red = gd.BucketedGradAllReducer(opt, bucket_bytes=args.bucket_mb << 20, force_collective=args.force_collective)
    # --amp: what Lightning's precision="16-mixed" does around the reference's training_step (torch default scaler settings);
    # FusedAdamEMA takes the scaler's scale / found_inf tensors on the device (no host read-back of found_inf per step)
    scaler = torch.amp.GradScaler("cuda") if args.amp else None

    def step(i):
        opt.zero_grad(set_to_none=not args.grad_views)   # autograd hands the gradients over; gathered per bucket / at step()
        if scaler is None:
            loss = model.training_step(example, i)
            loss.backward()
            red.finish()
            opt.step()
            return loss
        with torch.autocast("cuda", dtype=torch.float16):
            loss = model.training_step(example, i)
        scaler.scale(loss).backward()
        red.finish()
        scaler.step(opt)
        scaler.update()
        return loss

    for i in range(max(args.warmup, 1)):
        loss = step(i)
    torch.cuda.synchronize()
    gd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    gd.barrier()
    torch.cuda.synchronize()
    dt = gd.max_over_ranks(time.perf_counter() - t0, dev)
    assert torch.isfinite(loss.detach()).all()
    ms = dt / args.steps * 1e3
    per_layer = 16 * Nt * Dt * Dt + 8 * Nt * I * Dt + 14 * I * Dt * Dt
    fwd_flops = Lt * per_layer + (2 * Nt * 672 * Dt + 12 * Nt * Dt if cond else 12 * Nt * Dt)   # per sample
    rec = {"metric": "train_points_per_sec", "value": world * Bt * Nt * args.steps / dt, "unit": "points/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None,
           "dtype": "f16 operands, f32 accumulate (torch.autocast(float16) + GradScaler = the reference's precision='16-mixed')"
                    if args.amp else
                    {"bf16x3": "bf16 (split hi+lo operands, 3 MFMAs per product, fp32 accumulate)", "fp32": "f32",
                     "fp16": "bf16 (split; the fp16 mode is not used for gradients)",
                     "mixed": "bf16 (split; the mixed mode trains in split-bf16)",
                     "w2": "bf16 (split; the w2 mode trains in split-bf16)"}[args.precision],
           "data": "synthetic",
           "config": {"workload": (f"{args.config} image-conditional training step ({'frozen' if args.freeze_conditioner else 'trained'} channels-last ConvNeXt-T conditioner inside the step, "
                                   f"projective lookup, RayNetwork): batch {Bt}/GPU, N={Nt}, d={Dt}, L={Lt}" if cond else
                                   f"C2 unconditional training step: batch {Bt}/GPU, N={N}, d={D}, L={L}") +
                                  ": EDMLoss forward + backward (HIP autograd Functions), bucketed gradient all-reduce overlapped with "
                                  "backward, fused Adam+EMA",
                      "parallelism": f"dp{world}"},
           "loss": float(loss.detach()), "peak_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30,
           "train_tflops_algorithmic": 3 * fwd_flops * Bt / (ms * 1e-3) / 1e12}
    # the optimizer step alone, and the collective alone (bus bandwidth = 2 (n-1)/n bytes / t for an all-reduce)
    flat = opt.flat_grad()
    rec["adam_ema_ms"] = time_events(lambda: opt.launch(opt._adam_step, True), 10)   # the kernel alone (state kept: same step)
    if scaler is not None:
        rec["amp"] = {"loss_scale": scaler.get_scale(), "steps_skipped": opt._adam_step - opt.adam_steps_taken}
    rec["grad_bytes"] = flat.numel() * 4
    rec["buckets"] = len(red.buckets)
    if rank == 0 and not cond and args.precision in ("w2", "mixed", "bf16x3", "fp16"):
        # dominant kernel of the step: the weight-gradient product dW = dY^T X (gemm_tn_x3_kernel, split-bf16: 3 MFMAs per product)
        # at its largest call site — mlp.2: dY (Bt N, d), X = the hidden layer (Bt N, 2d) — timed with HIP events on the stream it
        # is launched on (+ the ~5 us fixed-order reduction of its per-group partials)
        from gecco_amd import autograd as ga
        gg = torch.Generator().manual_seed(5)
        # (round 6) the form the STEP runs at its most expensive call site — rocprofv3 of a step (profiles/r06*_train_amp_kernel_stats.csv):
        # gemm_tn_f16_kernel<2, 2, true, false>, 18 launches per step, 12 of them at this shape — the weight gradient of a linear whose input
        # was AdaGN(x): dY (Bt N, 2d) = the K | V (or hidden-layer) gradient, X = x (Bt N, d) with the AdaGN apply while it is staged
        # (kv_proj, mlp.0).  Rounds 2 - 5 timed mlp.2's shape on fp32 operands, a form the step no longer runs (its hidden layer is fp16).
        dy = torch.randn(Bt, N, 2 * D, generator=gg).to(dev)
        xh = torch.randn(Bt, N, D, generator=gg).to(dev)
        pro = (1.0 + 0.1 * torch.randn(Bt, D, generator=gg)).to(dev), (0.1 * torch.randn(Bt, D, generator=gg)).to(dev)
        with torch.no_grad():
            t_dw_pro = time_events(lambda: ga._linear_dw_main(dy, xh, pro=pro, prec="fp16" if args.amp else None), 10)
            t_dw = t_dw_pro
            dma_form = bool(args.amp and os.environ.get("GECCO_TRAIN_Y16", "0") == "1" and os.environ.get("GECCO_TRAIN_IO16", "1") != "0")
            if dma_form:   # (opt-in GECCO_TRAIN_Y16=1) dY and fp16(AdaGN(x)) as fp16 tensors, slabs global -> LDS by DMA
                dy16, y16 = dy.half(), (xh * pro[0][:, None] + pro[1][:, None]).half()
                t_dw = time_events(lambda: ga._linear_dw_main(dy16, y16, prec="fp16"), 10)
            h16 = torch.randn(Bt, N, 2 * D, generator=gg).to(dev).half()
            dyo = torch.randn(Bt, N, D, generator=gg).to(dev)
            t_dw2 = time_events(lambda: ga._linear_dw_main(dyo, h16, want_db=True, prec="fp16"), 10) if args.amp else None
        fl = 2.0 * Bt * N * D * 2 * D
        units = 1 if args.amp else 3
        rec["dominant_kernel"] = {"kernel": (("gemm_tn_f16_dma_kernel<2, 2> (both operands fp16 tensors, slabs by LDS-DMA)" if dma_form else "gemm_tn_f16_kernel<2, 2, true, false>")
                                             if args.amp else "gemm_tn_x3_kernel (AdaGN form)") +
                                            " (dW = dY^T AdaGN(x) of kv_proj / mlp.0: 2 M N K with M = Bt N rows contracted, " +
                                            ("fp16 operands = 1 MFMA per product" if args.amp else "split-bf16 = 3 MFMAs per product") +
                                            "; incl. the fixed-order reduction of the per-group partials)",
                                  "launches_per_step_at_this_shape": 2 * L,
                                  "register_staged_adagn_form_ms": t_dw_pro,
                                  "mlp2_dw_fp16_hidden_ms": t_dw2,
                                  "ms": t_dw, "achieved_tflops": fl / (t_dw * 1e-3) / 1e12, "peak_tflops": PEAK_BF16_MFMA_TFLOPS / units,
                                  "frac": fl / (t_dw * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                                  "frac_per_matrix_unit": units * fl / (t_dw * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, "bound": "mfma"}
        if args.amp:
            # with one MFMA per product the kernel is bound by its operand stream, not the matrix pipe: dY and X once from HBM
            # (algorithmic), each re-read by the other operand's 3 / 6 column tiles from L2, + the per-sample partials
            es = 2.0 if dma_form else 4.0   # bytes per operand element
            by = es * Bt * N * (D + 2 * D) + 4.0 * Bt * D * 2 * D
            l2 = es * Bt * N * (D * (2 * D // 128) + 2 * D * (D // 128))   # each operand re-read by the other's 128-column tiles
            rec["dominant_kernel"].update({"bound": "hbm", "algorithmic_bytes": by, "achieved_gbs": by / (t_dw * 1e-3) / 1e9,
                                           "peak_gbs": PEAK_HBM_GBS, "frac": by / (t_dw * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                           "frac_mfma": fl / (t_dw * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                                           "l2_operand_bytes": l2, "l2_gbs": l2 / (t_dw * 1e-3) / 1e9})
    if world > 1 or args.force_collective:
        # --force-collective at one rank: the RCCL all-reduce of every bucket really executes (a group of one: no bytes
        # cross a link, so no bus bandwidth is claimed) — the collective path of the step runs before an 8-GPU box has to
        ar = time_events(lambda: dist.all_reduce(flat), 10)
        rec["allreduce_ms_standalone"] = ar
        rec["allreduce_busbw_gbs"] = 2 * (world - 1) / world * flat.numel() * 4 / (ar * 1e-3) / 1e9 if world > 1 else None
        rec["collective"] = {"backend": dist.get_backend(), "world": world, "forced": bool(args.force_collective),
                             "all_reduces_issued": red.collectives_issued}
        red.enabled = False      # same step with NO collective anywhere (hooks and finish()): the difference is what the overlap leaves exposed
        for i in range(2):
            step(i)
        torch.cuda.synchronize()
        gd.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        gd.barrier()
        ms0 = gd.max_over_ranks(time.perf_counter() - t0, dev) / args.steps * 1e3
        rec["ms_per_step_without_allreduce"] = ms0
        rec["allreduce_ms_exposed"] = ms - ms0
        rec["distributed"] = distributed_report(dev, sizes=(flat.numel() * 4, 8 << 20))
    if rank == 0:
        print(json.dumps(rec))
    if dist.is_initialized():
        dist.destroy_process_group()


# matrix-pipe units (16-bit-instruction equivalents) per product of mlp.0 in the mixed mode: fp16 main product + two cross terms, fp6 x fp6
# with block scales (option "h6", default: a quarter of the 16-bit cycles each) or fp8 x fp8 (GECCO_H6=0: half each)
MLP0_UNITS = 2.0 if os.environ.get("GECCO_H6", "1") == "0" else 1.5


def executed_mfma_flops(mode, Bc=B, Nc=N, d=D, Ll=L):
    """MFMA FLOPs the evaluation EXECUTES in 16-bit-equivalent matrix-pipe units (an fp8 64-k instruction counts half the
    cycles per FLOP of a 16-bit one, so its FLOPs count half; split-bf16 executes 3 instructions per product): what
    `roofline.whole` prices against the 2500 TFLOP/s dense 16-bit peak.  Per sample and layer, SURVEY.md Appendix B terms."""
    kv, q, outp, m0, m2 = 4 * Nc * d * d, 2 * Nc * d * d, 2 * Nc * d * d, 4 * Nc * d * d, 4 * Nc * d * d
    attn = 2 * (4 * Nc * I * d)                       # pool + unpool: two products each
    chain = 2 * I * d * d + 4 * I * d * d + 4 * I * d * d + 4 * I * d * d   # pool.out_proj, broadcast.mlp.0 / .2, unpool k|v
    if mode == "fp32":
        return None
    if mode == "fp16":
        u = kv + q + outp + m0 + m2 + attn + chain
    elif mode == "bf16x3":
        u = 3 * (kv + q + outp + m0 + m2 + attn + chain)
    elif mode == "w2":   # the mixed mode's sites, the point MLP on the one-launch kernel: mlp.0 fp16 + two fp6 terms (1.5), mlp.2 fp16 + one (1.25)
        u = 0.5 * kv * (1 + 1.5) + q + attn + 2 * chain + 2 * outp + 1.5 * m0 + 1.25 * m2
    else:   # mixed: K, q one fp16 term; V fp16 + fp8 lo term (1.5); attention fp16; chain two-term fp16 weights (2); out_proj, mlp.2 h8 (2);
        # mlp.0 h6 (fp16 + two fp6 cross terms at a quarter of the 16-bit cycles each: 1.5) unless GECCO_H6=0 (h8: 2)
        u = 0.5 * kv * (1 + 1.5) + q + attn + 2 * chain + 2 * outp + MLP0_UNITS * m0 + 2 * m2
    return Bc * Ll * u


def set_metrics_bench(dev, S=256, Np=2048):
    """The evaluation protocol's set-vs-set Chamfer matrix (gecco-jax benchmark.py:21-39: every generated cloud against every reference
    cloud; S = T = 256 clouds of 2048 points = 2 x 2.7e11 point pairs) + 1-NNA / MMD / COV on it (benchmark.py:128-156), HIP events.  The inner
    dimension is 3 (no matrix-core shape): priced against the fp32 vector-ALU peak at 3 FMA + 1 min = 7 flops per point pair and direction."""
    from gecco_amd import metrics
    g = torch.Generator().manual_seed(0)
    a = torch.randn(S, Np, 3, generator=g).to(dev)
    b = torch.randn(S, Np, 3, generator=g).to(dev)
    ms = time_events(lambda: metrics.pairwise_set_distance(a, b), 3, warmup=1)
    ms_all = time_events(lambda: metrics.evaluate_sets(a[:64], b[:64]), 2, warmup=1)
    flops = 7.0 * 2.0 * S * S * Np * Np
    return {"workload": f"set-vs-set Chamfer, S = T = {S} clouds x {Np} points (gecco_set_chamfer_f32: no N x M matrix per pair)", "ms": ms,
            "point_pairs_per_s": 2.0 * S * S * Np * Np / ms * 1e3,
            "roofline": {"bound": "valu", "achieved": flops / ms / 1e9, "peak": 157.3, "unit": "TFLOP/s (fp32 vector)", "frac": flops / ms / 1e9 / 157.3,
                         "flops_per_point_pair": 7},
            "evaluate_sets_64_ms": ms_all}


def run_child(extra, timeout=420):
    """One more measurement in a fresh process (own GPU context; started as a child — never exec'd over this one)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), *extra, "--no-extras"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"rc {r.returncode}: {r.stderr[-300:]}"}
        return json.loads(lines[-1])
    except Exception as e:   # a failed extra must not cost the headline line
        return {"error": repr(e)[:300]}


def other_config_bench(args, rank, world, dev):
    """`--config C3 | C4 | C5`: the other BASELINE.json shapes, same metric (denoiser forward points/s/GPU), same timing
    contract; random-init weights (tests/test_hip_fullsize.py checks these very shapes against the oracle).  The conditioner is outside the timed step (it runs once per batch); its own time
    on the channels-last HIP path is reported beside it for C3 / C4."""
    from gecco_amd import distributed as gd
    from gecco_amd import hip_ops as ops
    ops.set_default_precision(args.precision)
    cfg = args.config
    Hh, Ii, Ll = H, I, 6
    g0 = torch.Generator().manual_seed(1000 + rank)

    def cloud(Bc, Nc):
        import math
        data = torch.randn(Bc, Nc, 3, generator=g0)
        u = (torch.arange(Bc) + torch.rand(Bc, generator=g0)) / Bc
        sig = torch.exp(math.log(0.002) + u * (math.log(165.0) - math.log(0.002)))
        return (data + sig[:, None, None] * torch.randn(Bc, Nc, 3, generator=g0)).contiguous(), sig.float().contiguous()

    def ray_weights(dc):
        p_ll = random_state_dict(31, dc, Ll)
        uu = lambda o, i_: (torch.rand(o, i_, generator=g0) * 2 - 1) / i_ ** 0.5
        pr = {k.replace("inner.", "backbone."): v for k, v in p_ll.items() if k.startswith("inner.")}
        pr["xyz_embed.weight"], pr["xyz_embed.bias"] = uu(dc, 3), uu(1, dc)[0]
        pr["img_feature_proj.1.weight"], pr["img_feature_proj.1.bias"] = uu(dc, 672), uu(1, dc)[0]
        pr["output_proj.1.weight"], pr["output_proj.1.bias"] = uu(3, dc), uu(1, 3)[0]
        pr["reparam.uvl_mean"], pr["reparam.uvl_std"] = torch.tensor([0.0, 0.0, 1.38]), torch.tensor([0.56, 0.60, 0.49])
        return {k: v.float().contiguous() for k, v in pr.items()}
    if cfg == "C1":
        # BASELINE.json configs[0]: unconditional ShapeNet-PointFlow airplane, N = 2048, d = 128, 4 layers (the reference's CPU-runnable case)
        Bc, Nc, dc, hw, Ll = 64, 2048, 128, 0, 4
    elif cfg == "C3":
        Bc, Nc, dc, hw = 64, 2048, 384, 224
    elif cfg == "C4":
        Bc, Nc, dc, hw = 32, 4096, 512, 256
    else:
        Bc, Nc, dc, hw = 8, 2048, 384, 0
        n_new = 16384
    rec_extra = {}
    per_layer = lambda n, d_: 16 * n * d_ * d_ + 8 * n * Ii * d_ + 14 * Ii * d_ * d_
    if cfg in ("C3", "C4"):
        p = {k: v.to(dev) for k, v in ray_weights(dc).items()}
        feats = [torch.randn(Bc, c, hw // st_, hw // st_, generator=g0) for c, st_ in ((96, 4), (192, 8), (384, 16))]
        K = torch.zeros(Bc, 3, 3)
        K[:, 0, 0] = K[:, 1, 1] = 1.1
        K[:, 0, 2] = K[:, 1, 2] = 0.5
        K[:, 2, 2] = 1.0
        net = ops.RayNetworkPlan(p, Hh, Ii)
        levels = ops.to_channels_last_levels([f.to(dev) for f in feats])
        x, sigma = cloud(Bc, Nc)
        x, sigma, K = x.to(dev), sigma.to(dev), K.to(dev)
        out = torch.empty_like(x)
        step = lambda: net.forward(x, sigma, K, levels, out=out)
        points = Bc * Nc
        flops = Bc * (Ll * per_layer(Nc, dc) + 2 * Nc * 672 * dc + 2 * 2 * Nc * 3 * dc)
        what = (f"{cfg} image-conditional denoiser forward: B={Bc}/GPU, N={Nc}, d={dc}, L={Ll}, {hw}x{hw} image -> pyramids "
                f"{hw // 4}/{hw // 8}/{hw // 16} (96/192/384 ch), projective lookup + RayNetwork; conditioner outside the step")
        # the conditioner itself, once per batch
        from gecco_amd.models.feature_pyramid import ConvNeXtExtractor
        from gecco_amd.structs import Context3d
        cn = ConvNeXtExtractor(pretrained=False).to(dev).eval()
        img = torch.rand(Bc, 3, hw, hw, device=dev)
        ctx = Context3d(image=img, K=K)
        cn(ctx)
        rec_extra["conditioner_ms"] = time_events(lambda: cn(ctx), 3, warmup=1)
        # the projective lookup alone (models/ray.py:64-87 -> csrc/lookup.hip: reparam^-1, projection, 4 bilinear taps per level,
        # channels-last texels, GroupNorm partials): SURVEY 8(d) calls it gather-bound — 4 taps x 672 channels x 4 B = 10.75 KB
        # gathered + 2.69 KB written per point — timed with HIP events on the stream it is launched on.  The taps of neighbouring
        # points share texels, so most of the gathered bytes are L2 / Infinity-Cache hits, not HBM: the bound is labelled
        # "l2-gather", `gather_gbs` is the algorithmic gather+write rate (it may exceed the HBM peak and is NOT an HBM fraction),
        # and the HBM figure is `hbm_floor_ms` / `hbm_frac` = the bytes that MUST cross HBM once (pyramids + output) at 8 TB/s.
        coef = ops.edm_coeffs(sigma)
        lk_levels = net._lookup_levels(levels)      # what the plan's evaluation gathers: in the w2 mode the fp16 texel image (round 6)
        tex = 2 if lk_levels[0].dtype == torch.float16 else 4
        lk = lambda: ops.ray_lookup(x, K, lk_levels, net.table.reparam, coef=coef, want_stats=True)
        lk_ms = time_events(lk, 10)
        lk32_ms = time_events(lambda: ops.ray_lookup(x, K, levels, net.table.reparam, coef=coef, want_stats=True), 10) if tex == 2 else lk_ms
        ct = 96 + 192 + 384
        gathered, written = Bc * Nc * 4 * ct * tex, Bc * Nc * ct * 4
        pyr_bytes = sum(f.numel() * tex for f in levels)
        rec_extra["lookup"] = {"kernel": f"ray_lookup_kernel<{'true' if tex == 2 else 'false'}> (one wave per point, channel chunks of channels-last "
                                         f"{'fp16' if tex == 2 else 'fp32'} texels; interpolation in fp32)",
                               "texel_bytes": tex, "ms_fp32_texels": lk32_ms,
                               "ms": lk_ms, "bound": "l2-gather", "algorithmic_bytes": gathered + written,
                               "gather_gbs": (gathered + written) / (lk_ms * 1e-3) / 1e9,
                               "hbm_bytes_once": pyr_bytes + written, "hbm_peak_gbs": PEAK_HBM_GBS,
                               "hbm_floor_ms": (pyr_bytes + written) / PEAK_HBM_GBS / 1e9 * 1e3,
                               "hbm_frac": (pyr_bytes + written) / PEAK_HBM_GBS / 1e9 * 1e3 / lk_ms,
                               "bytes_per_point": {"gathered": 4 * ct * tex, "written": ct * 4}}
    elif cfg == "C1":
        p = {k: v.to(dev) for k, v in random_state_dict(9, dc, Ll).items()}
        net = ops.LinearLiftPlan(p, Hh, Ii)
        x, sigma = cloud(Bc, Nc)
        x, sigma = x.to(dev), sigma.to(dev)
        out = torch.empty_like(x)
        step = lambda: net.forward(x, sigma, out=out)
        points = Bc * Nc
        flops = Bc * Ll * per_layer(Nc, dc)
        what = (f"C1 unconditional denoiser forward at the reference's CPU-runnable shape: B={Bc}/GPU, N={Nc}, d={dc}, L={Ll}, I={Ii}, H={Hh}, "
                "EDMPrecond(LinearLift(SetTransformer))")
    else:
        p = {k: v.to(dev) for k, v in random_state_dict(9, dc, Ll).items()}
        net = ops.LinearLiftPlan(p, Hh, Ii)
        xk, sk = cloud(Bc, Nc)
        xk, sk = xk.to(dev), sk.to(dev)
        _, cache = net.forward(xk, sk, do_cache=True)
        g = torch.Generator().manual_seed(7 + rank)
        x = torch.randn(Bc, n_new, 3, generator=g).to(dev)
        out = torch.empty_like(x)
        step = lambda: net.forward(x, sk, cache=cache, out=out)
        points = Bc * n_new
        flops = Bc * Ll * (12 * n_new * dc * dc + 4 * n_new * Ii * dc)
        what = (f"C5 cached-inducer (upsampling) evaluation: B={Bc}/GPU, n_new={n_new} points against the inducer states of "
                f"N={Nc} known points, d={dc}, L={Ll} (diffusion.py:433-447: every sub-step of `upsample`)")
        if rank == 0 and world == 1 and not args.no_sampler and (dc, Ll) == (D, L):
            # Diffusion.upsample end to end (reference diffusion.py:354-470) on a SHORT schedule (8 of the 128 steps: the
            # per-step cost is what scales): per outer step one full evaluation of the known clouds + 5 x 2 cached
            # evaluations of the new points + the sampler kernels, one captured hipGraph per step; and the same eagerly
            ups = build_model(random_state_dict(9, dc, Ll)).to(dev).eval()
            known = (torch.randn(Bc, Nc, 3, generator=g) * ups.reparam.sigma.cpu() + ups.reparam.mean.cpu()).to(dev)
            nst, nsub = 8, 5
            res = {}
            for name, ug in (("hipgraph", True), ("eager", False)):
                ups.upsample(known, n_new=n_new, num_steps=3, num_substeps=nsub, use_graph=ug)   # warm-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                o = ups.upsample(known, n_new=n_new, num_steps=nst, num_substeps=nsub, use_graph=ug)
                torch.cuda.synchronize()
                res[name] = time.perf_counter() - t0
                assert torch.isfinite(o).all()
            best = min(res.values())   # Diffusion.upsample's default is the eager loop (the faster of the two here)
            rec_extra["upsample"] = {"outer_steps": nst, "num_substeps": nsub, "evaluations_per_step": "1 full + 10 cached",
                                     "seconds": res["eager"], "ms_per_outer_step": res["eager"] / nst * 1e3,
                                     "eager_ms_per_outer_step": res["eager"] / nst * 1e3,
                                     "hipgraph_ms_per_outer_step": res["hipgraph"] / nst * 1e3,
                                     "default": "eager (use_graph=False): the captured outer step of ~1100 small kernel nodes replays slower",
                                     "projected_128_steps_s": best / nst * 128,
                                     "new_points_per_sec_128_steps": Bc * n_new / (best / nst * 128)}
    step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step()
    run = gr.replay
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    gd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    gd.barrier()
    torch.cuda.synchronize()
    dt = gd.max_over_ranks(time.perf_counter() - t0, dev)
    assert torch.isfinite(out).all()
    ms = dt / args.steps * 1e3
    tf = flops / (ms * 1e-3) / 1e12
    rec = {"metric": "denoiser_fwd_points_per_sec", "value": world * points * args.steps / dt, "unit": "points/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
           "config": {"workload": what + f"; arithmetic mode {args.precision}, seeded random weights",
                      "parallelism": "replicas (batch-sharded, no data-path collective)" if world > 1 else "single GPU"},
           "forward_tflops": tf, "launch": "hipgraph replay of one captured evaluation",
           "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_BF16_MFMA_TFLOPS,
                        "traffic": None,
                        "kernel": "whole evaluation: algorithmic FLOPs (SURVEY.md Appendix B) / step time against the dense 16-bit MFMA "
                                  "peak; the split-bf16 products of the mixed / bf16x3 modes execute 3 MFMAs per algorithmic product"},
           **rec_extra}
    if rank == 0:
        print(json.dumps(rec))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra blocks of the default run (train / configs / upsample: child processes, ~6 s of timed work each)")
    ap.add_argument("--eager", action="store_true", help="time eager Diffusion.forward calls instead of the hipGraph replay")
    ap.add_argument("--precision", default=os.environ.get("GECCO_PRECISION", "w2"), choices=["fp32", "bf16x3", "mixed", "w2", "fp16"],
                    help="arithmetic of the linears and attention products: w2 (default: the mixed mode with the point MLP of each layer as "
                         "ONE launch, fp16 + fp6 block-scaled correction terms, the hidden layer never in HBM: D and F_x <= 4.1e-4, asserted "
                         "against 5e-4 = half the north-star bar), mixed (the strict mode: fp16 where operand rounding does not reach the "
                         "output, split-bf16 elsewhere: D and F_x ~6e-5), split-bf16 (3 MFMAs per product, D and F_x "
                         "~2e-5 .. 5e-5 from the fp32 reference), fp16 operands with fp32 accumulation (faster; D ~4e-4, F_x ~1e-3: "
                         "at the 1e-3 bar) or exact fp32 MFMA (~1e-6)")
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C3", "C4", "C5"],
                    help="BASELINE.json configuration: C2 (default, the headline), C3 / C4 image-conditional, C5 cached upsampling evaluation")
    ap.add_argument("--train", action="store_true", help="time the data-parallel training step instead of the forward")
    ap.add_argument("--amp", action="store_true",
                    help="--train: the reference's shipped trainer setting, precision='16-mixed' (example_configs/*.py): training_step under "
                         "torch.autocast(float16), loss scaled by a torch.amp.GradScaler — the HIP path then runs its linears with fp16 operands")
    ap.add_argument("--train-batch", type=int, default=48, help="per-GPU batch of --train (shipped config: 48)")
    ap.add_argument("--bucket-mb", type=int, default=8, help="gradient all-reduce bucket size of --train")
    ap.add_argument("--grad-views", action="store_true",
                    help="--train: keep p.grad as views of the flat buffer (autograd adds into zeros: one small kernel per parameter)")
    ap.add_argument("--freeze-conditioner", action="store_true",
                    help="--train --config C3|C4: evaluate the ConvNeXt conditioner without gradients (the reference trains it)")
    ap.add_argument("--force-collective", action="store_true",
                    help="--train: create the (RCCL) process group even for one rank and issue the per-bucket all-reduces in it")
    ap.add_argument("--selftest-launcher", action="store_true", help="N-rank plumbing on gloo with a stand-in step (no GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one.  Nothing above touched the GPU (importing torch does not), the ranks are
        # fresh processes; the library is built once here so that N ranks do not race on it.
        if not args.selftest_launcher:
            import __graft_entry__ as ge
            ge.build()
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE)")
    if args.selftest_launcher:
        return launcher_selftest(args)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from gecco_amd import distributed as gd   # rendezvous / barrier / max-over-ranks (covered on gloo in tests/)
    if args.force_collective and "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local))
    gd.init("nccl", dev, force=args.force_collective)

    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(ge.LIB):
        ge.build()
    gd.barrier()
    from gecco_amd import hip_ops as ops
    if args.train:
        return train_bench(args, rank, world, dev)
    if args.config != "C2":
        return other_config_bench(args, rank, world, dev)
    ops.set_default_precision(args.precision)

    p_cpu = random_state_dict(seed=3)
    x_cpu, sigma_cpu = synthetic_cloud(seed=rank)  # each rank: its own batch (weak scaling)
    x, sigma = x_cpu.to(dev), sigma_cpu.to(dev)
    model = build_model(p_cpu).to(dev).eval()      # gecco_amd.Diffusion, the drop-in module API
    out = torch.empty_like(x)

    @torch.no_grad()
    def eager_step():
        model(x, sigma, None, out=out)             # Diffusion.forward (reference diffusion.py:233-247)

    if args.eager:
        step = eager_step
    else:
        # the same evaluation captured once as a hipGraph and replayed: every kernel still runs every step, the ~100
        # launches cost one host call (the eager loop is host-launch-bound on a busy box: it is timed beside it below)
        run = model.graphed_forward(x, sigma, None)
        out = run()
        step = run

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    gd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    local_dt = time.perf_counter() - t0
    gd.barrier()
    torch.cuda.synchronize()
    dt = gd.max_over_ranks(time.perf_counter() - t0, dev)
    assert torch.isfinite(out).all(), "non-finite denoiser output"
    dist_rep = distributed_report(dev, local_points_per_sec=B * N * args.steps / local_dt) if world > 1 else None

    ms = dt / args.steps * 1e3
    value = world * B * N * args.steps / dt
    rec = {
        "metric": "denoiser_fwd_points_per_sec", "value": value, "unit": "points/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C2 unconditional denoiser forward: B={B}/GPU, N={N}, d={D}, L={L}, I={I}, H={H}, "
                               "EDMPrecond(LinearLift(SetTransformer)), fp32 MFMA, random-init weights",
                   "parallelism": "replicas (batch-sharded, no data-path collective)" if world > 1 else "single GPU"},
        "forward_tflops": flops_per_sample() * B / (ms * 1e-3) / 1e12,
        "launch": "eager" if args.eager else "hipgraph replay of one captured Diffusion.forward",
        "target_points_per_sec_per_gpu": 2.0e6,
        "per_rank_points_per_sec": (dist_rep or {}).get("per_rank_points_per_sec") or [B * N * args.steps / local_dt],
        "build": build_info(),
    }
    if dist_rep is not None:
        rec["distributed"] = dist_rep
    mode = args.precision
    rec["dtype"] = {"mixed": "f16/fp8/bf16 mixed (kv_proj|q_proj: fp16 activations x fp16 weights, the V columns + an fp8 second weight term; fp16 K|V, q "
                             "and attention products; out_proj and the point MLP: fp16 main product + two fp8 cross terms (h8, split-bf16 accuracy); "
                             "inducer chain: fp16 activations x two-term fp16 weights; fp32 accumulate, residual stream and statistics)",
                    "w2": "f16/fp6/fp8 mixed (the mixed mode with the point MLP of every layer as ONE launch: fp16 main products + fp6 block-scaled second "
                          "terms for AdaGN(x) and both weights, the 768-wide hidden layer kept in registers as fp16; the other sites as in the mixed mode; "
                          "fp32 accumulate, residual stream and statistics)",
                    "fp16": "f16 (fp16 operands, fp32 accumulate; fp16-stored intermediates, fp32 residual stream and statistics)",
                    "bf16x3": "bf16 (split hi+lo operands, 3 MFMAs per product, fp32 accumulate; fp32 activations in HBM)",
                    "fp32": "f32"}[mode]
    rec["config"]["workload"] = rec["config"]["workload"].replace(
        "fp32 MFMA", {"fp16": "fp16 MFMA", "bf16x3": "split-bf16 MFMA", "fp32": "fp32 MFMA", "mixed": "mixed fp16 / fp8-cross-term / split-bf16 MFMA",
                     "w2": "mixed fp16 / fp6- and fp8-cross-term MFMA, one-launch point MLP"}[mode])
    if rank == 0 and not args.no_roofline:
        site_mode = "bf16x3" if mode in ("mixed", "w2") else mode
        sites = gemm_call_sites(ops, dev, site_mode) if mode not in ("mixed", "w2") else []   # mixed / w2: their own rounds (h8_round, w2_round)
        tot_f, tot_b, tot_ms, per = 0.0, 0.0, 0.0, {}
        # fp16 mode: kernel-only launches (images prepared) timed inside hipGraphs of the three-launch round
        seq_ms = None
        if site_mode == "fp16" and mode != "mixed":
            times, seq_ms = time_in_sequence([fn for _, _, _, fn in sites])
        else:
            times = [time_events(fn, 10) for _, _, _, fn in sites]
        for (name, fl, by, fn), t in zip(sites, times):
            per[name] = {"ms": round(t, 4), "tflops": round(fl / (t * 1e-3) / 1e12, 2), "gbs": round(by / (t * 1e-3) / 1e9, 1)}
            tot_f += fl
            tot_b += by
            tot_ms += t
        tf = tot_f / (tot_ms * 1e-3) / 1e12 if tot_ms else 0.0
        gbs = tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms else 0.0
        tj_site, tmeta = load_traffic(site_mode)
        traffic = tj_site.get("bytes_per_launch")
        if site_mode == "fp16":
            # The dominant kernel is the fused point MLP (35 % of device time): 155 GFLOP over 402 MB that must cross HBM
            # = 385 FLOP/B, above the ridge of 2500 TF / 8 TB/s = 312 -> bound: mfma.  The other two launches of a layer
            # (kv|q 231 FLOP/B, unpool + out_proj 103 FLOP/B) sit on the HBM side; they are priced in "hbm_side".
            mk = "norm+mlp.0+act+mlp.2+res+stats"
            mlp_tf = per[mk]["tflops"]
            others = [k for k in per if k != mk]
            o_b = sum(by for name, fl, by, fn in sites if name != mk)
            o_ms = sum(per[k]["ms"] for k in others)
            rec["roofline"] = {"bound": "mfma", "achieved": mlp_tf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": mlp_tf / PEAK_BF16_MFMA_TFLOPS, "traffic": traffic,
                               "kernel": "mlp_fused_f16_kernel<3> (AdaGN + mlp.0 + GaussianActivation + mlp.2 + residual + GroupNorm "
                                         "partials in one launch, v_mfma_f32_32x32x16_f16; the hidden layer never leaves the CU); achieved = "
                                         "4 B N d 2d FLOP / event-timed duration; traffic = FETCH_SIZE x 2 + WRITE_SIZE of that kernel "
                                         "(x is read twice: operand build and residual) vs 402 MB algorithmic",
                               "hbm_side": {"kernels": "gemm_f16_astat_kernel<12,6> (AdaGN + kv|q, head-major fp16 out) and "
                                                       "unpool_outproj_f16_kernel<3,48> (unpool attention + out_proj + residual + partials)",
                                            "achieved_gbs_algorithmic": o_b / (o_ms * 1e-3) / 1e9, "peak_gbs": PEAK_HBM_GBS,
                                            "frac": o_b / (o_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
                               "all_three": {"achieved_tflops": tot_f / (seq_ms * 1e-3) / 1e12,
                                             "achieved_gbs_algorithmic": tot_b / (seq_ms * 1e-3) / 1e9, "ms_per_round": seq_ms},
                               "timing": "HIP events around hipGraph replays of the three-launch round (kernel launches only, weight "
                                         "images prepared); a launch's duration = plain round - round without that launch",
                               "per_site": per}
        elif mode == "w2":
            # The dominant kernel = the launch with the largest TOTAL device time per evaluation: the one-launch point MLP (6 launches
            # of ~0.34 ms of a 4.4 ms evaluation; profiles/r05*_fwd_kernel_stats_one_stream.csv).  Priced against the PLAIN dense 16-bit
            # peak on its algorithmic 2MNK (both products); the figure per executed matrix-pipe unit is beside it.  Timed live: HIP events
            # around hipGraph replays of the round out_proj -> MLP on shared buffers, on the stream the graph replays on (round - round
            # without the launch), at the one-stream launch shape (B = 64 clouds per launch: GECCO_FWD_STREAMS=1 profiles).
            rsites = w2_round(ops, dev)
            rtimes, round_ms = time_in_sequence([fn for _, _, _, fn in rsites])
            per = {name: {"ms": round(t, 4), "tflops": round(fl / (t * 1e-3) / 1e12, 2), "gbs": round(by / (t * 1e-3) / 1e9, 1), "launches_per_evaluation": L}
                   for (name, fl, by, fn), t in zip(rsites, rtimes)}
            per["round_ms"] = round(round_ms, 4)
            mk = max((k for k in per if k != "round_ms"), key=lambda k: per[k]["ms"] * per[k]["launches_per_evaluation"])
            mtf = per[mk]["tflops"]
            mb = [by for name, fl, by, fn in rsites if name == mk][0]
            units = 1.375   # mlp.0: fp16 + two fp6 terms (1.5); mlp.2: fp16 + one (1.25)
            tjd, tmeta = load_traffic("w2")
            kk = next((v for k, v in tjd.get("per_kernel", {}).items() if k.startswith("mlp_fused_w_kernel")), {})
            rec["roofline"] = {"bound": "mfma", "achieved": mtf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": mtf / PEAK_BF16_MFMA_TFLOPS, "traffic": kk.get("bytes_per_launch"),
                               "dominant_kernel": mk, "dominant_by": "largest total device time per evaluation (ms x launches) among per_site",
                               "matrix_units_per_product": units, "frac_per_matrix_unit": units * mtf / PEAK_BF16_MFMA_TFLOPS,
                               "mfma_busy_pmc": kk.get("mfma_busy"),
                               "reproduce_from": "profiles/r06v_fwd_kernel_stats_one_stream.csv (tools/prof_fwd.sh: GECCO_FWD_STREAMS=1 = the B = 64 launch shape "
                                                 "timed here): 154.6e9 FLOP / AverageNs of mlp_fused_w_kernel / 2.5e15",
                               "kernel": "mlp_fused_w_kernel<1> = the point MLP of a layer in one launch (mlp_fused_w.hip: 4 waves of 512 registers per 128-row "
                                         "tile, one per SIMD; AdaGN apply, mlp.0 as v_mfma_f32_32x32x16_f16 + two v_mfma_scale_f32_32x32x64_f8f6f4 (fp6 x fp6, "
                                         "block scales) per 64 k, GaussianActivation, the hidden layer kept as register fragments, mlp.2 as fp16 + one fp6 term, "
                                         "residual, GroupNorm partials); achieved = 4 B N d 2d FLOP (both products' 2MNK) / its duration inside hipGraph replays of "
                                         "the round out_proj -> MLP (HIP events; round - round without it); peak = the dense 16-bit MFMA peak; traffic / "
                                         "mfma_busy_pmc = FETCH_SIZE x 2 + WRITE_SIZE and SQ_VALU_MFMA_BUSY_CYCLES of that kernel in the forward ("
                                         + str(tjd.get("source")) + ": committed constants, not measured in this run)",
                               "hbm": {"achieved_gbs_algorithmic": mb / (per[mk]["ms"] * 1e-3) / 1e9, "peak_gbs": PEAK_HBM_GBS,
                                       "frac": mb / (per[mk]["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": mb},
                               "per_site": per}
        elif mode == "mixed":
            # The dominant kernel of the mixed mode since round 3 is mlp.0 on the A-stationary h8 kernel (gemm_h8_astat_kernel: AdaGN
            # apply + fp16 main product + two fp8 cross terms + GaussianActivation, writes the h8 activation image; 20 % of the device
            # time, profiles/r03p_fwd_kernel_stats_one_stream.csv).  Its product executes 2 matrix-pipe units (one 16-bit instruction
            # stream + two fp8 streams at twice the rate; with option "h6", the default, two fp6 streams at four times the rate: 1.5 units),
            # so its matrix roof is 2500 / 2 (1.5) TFLOP/s of 2MNK; 77.3 GFLOP x 2 over 503 MB
            # = 307 unit-FLOP/B, at the ridge of 2500 TF / 8 TB/s = 312: both roofs are reported, `bound` names the matrix side the
            # counters show busier (0.36 of the cycles against 0.33 of 8 TB/s).  Timed live with HIP events inside hipGraph replays of
            # the round out_proj -> mlp.0 -> mlp.2 (all three on their h8 kernels, shared buffers): round - round without it.
            rsites = h8_round(ops, dev)
            rtimes, round_ms = time_in_sequence([fn for _, _, _, fn in rsites])
            per = {name: {"ms": round(t, 4), "tflops": round(fl / (t * 1e-3) / 1e12, 2), "gbs": round(by / (t * 1e-3) / 1e9, 1), "launches_per_evaluation": L}
                   for (name, fl, by, fn), t in zip(rsites, rtimes)}
            per["round_ms"] = round(round_ms, 4)
            # dominant = the launch with the largest total device time per evaluation among per_site (every site launches L times)
            mk = max((k for k in per if k != "round_ms"), key=lambda k: per[k]["ms"] * per[k]["launches_per_evaluation"])
            mtf = per[mk]["tflops"]
            mb = [by for name, fl, by, fn in rsites if name == mk][0]
            tjd, tmeta = load_traffic("mixed")
            pk = tjd.get("per_kernel", {})
            is0 = mk.startswith("mlp.0")
            munits = MLP0_UNITS if is0 else 1.25   # mlp.2 / out_proj on the register-fed kernel: fp16 + one fp6 term
            kk = next((v for k, v in pk.items() if k.startswith("gemm_h8_astat_kernel" if is0 else "gemm_h8_areg_kernel")), {})
            rec["roofline"] = {"bound": "mfma", "achieved": mtf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": mtf / PEAK_BF16_MFMA_TFLOPS, "traffic": kk.get("bytes_per_launch"),
                               "dominant_kernel": mk, "dominant_by": "largest total device time per evaluation (ms x launches) among per_site",
                               "frac_per_matrix_unit": munits * mtf / PEAK_BF16_MFMA_TFLOPS,
                               "matrix_units_per_product": munits,
                               # cycle-based view of the same kernel (committed PMC constant, not measured in this run)
                               "mfma_busy_pmc": kk.get("mfma_busy"),
                               "kernel": ("the register-fed gemm_h8_areg_kernel of that site (fp16 + one fp6 term); the mode's mlp.0 kernel for reference: " if not is0 else "") +
                                         "gemm_h8_astat_kernel<6,4,6,1,true,0," + ("false" if MLP0_UNITS == 2.0 else "true") + "> = mlp.0 of the mixed "
                                         "mode (A-stationary over 128-row blocks, AdaGN apply, 4 x v_mfma_f32_32x32x16_f16 + 2 x "
                                         "v_mfma_scale_f32_32x32x64_f8f6f4 (" + ("fp8 x fp8" if MLP0_UNITS == 2.0 else "fp6 x fp6, block scales") + ") per 64 k of a "
                                         "32 x 32 tile, GaussianActivation, h8 activation image "
                                         "out), one launch over the whole batch on one stream; achieved = 2MNK / its duration inside hipGraph replays of "
                                         "the round out_proj -> mlp.0 -> mlp.2 on shared buffers (HIP events; round - round without it); peak = the plain dense "
                                         "16-bit MFMA peak (2500 TFLOP/s) on the algorithmic 2MNK; frac_per_matrix_unit counts the " + str(munits) + " matrix-pipe "
                                         "units the arithmetic executes per product; traffic / mfma_busy_pmc = FETCH_SIZE x 2 + "
                                         "WRITE_SIZE and SQ_VALU_MFMA_BUSY_CYCLES of that kernel in the forward (" + str(tjd.get("source")) + ": committed "
                                         "constants, not measured in this run)",
                               "hbm": {"achieved_gbs_algorithmic": mb / (per[mk]["ms"] * 1e-3) / 1e9, "peak_gbs": PEAK_HBM_GBS,
                                       "frac": mb / (per[mk]["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS},
                               "per_site": per}
        elif mode == "bf16x3":
            # The split-bf16 algorithm issues 3 MFMAs per product, so its matrix roof is the dense bf16 peak / 3 =
            # 833 TFLOP/s of 2MNK work.  The launches run at 96..192 FLOP/B (2MNK over fp32 A/residual/C bytes): at or
            # above the ridge of that roof (833 TF / 8 TB/s = 104 FLOP/B), and the PMC counters agree — the matrix
            # pipe is the busiest unit (44 % busy, HBM at 30 % of 8 TB/s; profiles/README.md).  Bound: mfma.
            rec["roofline"] = {"bound": "mfma", "achieved": tf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / PEAK_BF16_MFMA_TFLOPS, "frac_per_matrix_unit": 3 * tf / PEAK_BF16_MFMA_TFLOPS, "traffic": traffic,
                               "kernel": "gemm_dma_kernel<3,*,true,128> (LDS-DMA ring, 3 x v_mfma_f32_32x32x16_bf16 per product), mean over "
                                         "its 4 per-layer launch shapes; achieved = 2MNK / event-timed duration, peak = the plain dense bf16 MFMA "
                                         "peak (2500 TFLOP/s); frac_per_matrix_unit counts the 3 MFMAs per product",
                               "hbm": {"achieved_gbs_algorithmic": gbs, "peak_gbs": PEAK_HBM_GBS, "frac": gbs / PEAK_HBM_GBS},
                               "per_site": per}
        else:
            rec["roofline"] = {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                               "kernel": "gemm_dma_kernel<3,*,false,128> (LDS-DMA ring, v_mfma_f32_32x32x2_f32), mean over its 4 per-layer launch shapes",
                               "per_site": per}
        # the other arithmetic modes beside it, for the record (same model, same inputs)
        for other in ("fp16", "w2", "mixed", "bf16x3", "fp32"):
            if other == mode:
                continue
            ops.set_default_precision(other)
            for _ in range(2):
                eager_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                eager_step()
            torch.cuda.synchronize()
            ms_o = (time.perf_counter() - t0) / 10 * 1e3
            rec[{"fp16": "fp16_mode", "w2": "w2_mode", "mixed": "mixed_mode", "bf16x3": "split_bf16_mode", "fp32": "exact_fp32_mode"}[other]] = {
                "ms_per_step": ms_o, "points_per_sec": B * N / (ms_o * 1e-3), "launch": "eager",
                "parity_vs_fp32_reference": {"fp16": "D ~4e-4, F_x ~1e-3 (at the bar; tests/test_hip_fullsize.py)",
                                             "w2": "D <= 3.4e-4, F_x 1.8e-4 .. 4.1e-4 (bar of the mode: 5e-4)",
                                             "mixed": "D ~2e-5, F_x ~6e-5", "bf16x3": "D ~2e-5, F_x ~5e-5", "fp32": "~1e-6"}[other]}
        rec["parity_vs_fp32_reference"] = {"w2": "F_x 3.2e-4 .. 3.8e-4 on C2 (every sigma, and the B = 64 headline batch), 2.0e-4 on C3, 3.6e-4 .. 4.1e-4 on C5, 3.0e-4 at L = 8 "
                                                 "(tests/test_hip_fullsize.py: asserted against 5e-4 on BOTH outputs; north-star bar 1e-3, max-norm)",
                                           "mixed": "D ~3e-5, F_x 5e-5 .. 8e-5 on C2 - C5, <= 1.6e-4 on the L = 8 / 10 / 14 networks (tests/test_hip_fullsize.py; bar 1e-3)",
                                           "fp16": "D ~4e-4, F_x ~1e-3", "bf16x3": "D ~2e-5, F_x ~5e-5", "fp32": "~1e-6"}[mode]
        ops.set_default_precision(mode)
        if not args.eager:   # the eager loop of the headline mode, for the record
            for _ in range(2):
                eager_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                eager_step()
            torch.cuda.synchronize()
            rec["eager_ms_per_step"] = (time.perf_counter() - t0) / 10 * 1e3
    if rank == 0 and world == 1 and not args.eager:
        # beside the headline (whose every replay rebuilds the weight images, as a forward between two weight updates must): the same
        # graph captured inside hip_ops.frozen_weights() — images built once by the warm-up call, as every sampler call does
        try:   # (an extra: whatever happens here must not cost the headline line)
            out_head = run().clone()            # the headline graph's own output, whatever the buffers held meanwhile
            runf = model.graphed_forward(x, sigma, None, frozen_weights=True)
            same = bool(torch.equal(runf(), out_head))
            rec["frozen_weights_ms_per_step"] = time_events(runf, args.steps, warmup=3)
            rec["frozen_weights_bit_identical"] = same
            if not same:
                rec["frozen_weights_error"] = "frozen-weights evaluation differs from the headline's"
        except Exception as e:
            rec["frozen_weights_ms_per_step"] = None
            rec["frozen_weights_error"] = repr(e)[:300]
    if rank == 0 and world == 1 and not args.no_sampler:
        # Metric 2 (BASELINE.json): 128-step sample_stochastic wall-clock = 255 evaluations + fp64 sampler kernels,
        # one hipGraph per step replayed 127 times
        model.sample_stochastic((B, N, 3), None, num_steps=4)  # warm-up (graph capture path)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        smp = model.sample_stochastic((B, N, 3), None, num_steps=128)
        torch.cuda.synchronize()
        ts = time.perf_counter() - t0
        assert torch.isfinite(smp).all()
        rec["sample_128_steps"] = {"seconds": ts, "evaluations": 255, "points_per_sec": B * N / ts,
                                   "ms_per_evaluation": ts / 255 * 1e3, "hipgraph": True}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        rec["cpu_baseline"] = cpu_baseline(p_cpu, x_cpu, sigma_cpu)
    if rank == 0 and "roofline" in rec:
        # whole evaluation against both roofs: executed matrix-pipe work / time / 2500 TFLOP/s, and HBM bytes from the counters
        # (profiles/gemm_hbm_traffic.json: FETCH_SIZE / WRITE_SIZE passes over whole evaluations of this tree) / time / 8 TB/s
        ex = executed_mfma_flops(mode)
        tjw, tmeta_w = load_traffic(mode)
        rec["roofline"].update(tmeta_w)     # "traffic_tree" (where the counter constants come from) and "traffic_stale"
        cb = tjw.get("bytes_per_evaluation")
        rec["roofline"]["whole"] = {
            "executed_mfma_tflops": ex / (ms * 1e-3) / 1e12 if ex else None,
            "mfma_frac": ex / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS if ex else None,
            "counter_bytes_per_evaluation": cb, "counter_source": tjw.get("source"),
            "hbm_gbs": cb / (ms * 1e-3) / 1e9 if cb else None, "hbm_frac": cb / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS if cb else None,
            "algorithmic_bytes_fp32_stream": 6.04e9}
    if rank == 0 and world == 1 and not args.no_extras and not args.eager:
        # what else the tree does, in front of the driver (compact; each a child process with ~10 timed steps).  The headline is
        # complete at this point: it goes to stderr now (a harness that loses patience with the extras still has it), this
        # process gives its device memory back before the children start, and the extras share ONE time budget (a slow or
        # hung child costs at most its own cap; what does not fit is reported as skipped, never waited for)
        print("[bench headline, extras pending] " + json.dumps({k: rec[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")}),
              file=sys.stderr, flush=True)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        t_extras = time.time()
        budget, cap = 360.0, 150.0

        def run_extra(argv):
            left = budget - (time.time() - t_extras)
            if left < 20.0:
                return {"error": "skipped: the extras' time budget (360 s) is spent"}
            return run_child(argv, timeout=min(cap, left))
        # the training step in the reference's shipped trainer setting (precision="16-mixed": both example configs) ...
        tr = run_extra(["--train", "--amp", "--steps", "20", "--warmup", "3", "--precision", args.precision])
        rec["train"] = tr if "error" in tr else {
            "config": "C2 training step, batch 48/GPU (forward + backward + fused Adam/EMA, HIP autograd path) under torch.autocast(float16) + "
                      "GradScaler — the reference's precision='16-mixed': linears with fp16 operands / fp32 accumulation, fp32 tensors",
            "ms_per_step": tr["ms_per_step"], "points_per_sec": tr["value"], "tflops_algorithmic": tr["train_tflops_algorithmic"],
            "adam_ema_ms": tr.get("adam_ema_ms"), "dominant_kernel": tr.get("dominant_kernel"), "amp": tr.get("amp")}
        # ... and without autocast: every product in split-bf16 (gradients within 1e-4 of the fp32 oracle's)
        tr = run_extra(["--train", "--steps", "10", "--warmup", "3", "--precision", args.precision])
        rec["train_split_bf16"] = tr if "error" in tr else {
            "config": "the same step without autocast: split-bf16 (3 MFMAs per product) everywhere",
            "ms_per_step": tr["ms_per_step"], "points_per_sec": tr["value"], "tflops_algorithmic": tr["train_tflops_algorithmic"],
            "dominant_kernel": tr.get("dominant_kernel")}
        cfgs = {}
        for c in ("C1", "C3", "C4", "C5"):
            rc_ = run_extra(["--config", c, "--steps", "10", "--warmup", "3", "--precision", args.precision] + (["--no-sampler"] if c != "C5" else []))
            if "error" in rc_:
                cfgs[c] = rc_
                continue
            cfgs[c] = {"ms_per_step": rc_["ms_per_step"], "points_per_sec": rc_["value"], "forward_tflops": rc_["forward_tflops"],
                       "workload": rc_["config"]["workload"][:160]}
            if "conditioner_ms" in rc_:
                cfgs[c]["conditioner_ms_per_batch"] = rc_["conditioner_ms"]
            if "lookup" in rc_:
                cfgs[c]["lookup"] = rc_["lookup"]
            if "upsample" in rc_:
                rec["upsample"] = rc_["upsample"]
        rec["configs"] = cfgs
        try:
            rec["set_metrics"] = set_metrics_bench(dev)
        except Exception as e:   # a failed extra must not cost the headline line
            rec["set_metrics"] = {"error": repr(e)[:300]}
    if rank == 0:
        print(json.dumps(rec))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
