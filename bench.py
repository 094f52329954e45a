#!/usr/bin/env python3
"""Benchmark of the GECCO denoiser forward on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A "step" is one Diffusion.forward (EDMPrecond + LinearLift + 6-layer set transformer) over one batch
of synthetic clouds resident in HBM: config C2 = B=64, N=2048, d=384, L=6, I=64, H=8.  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank evaluates its own batch — samples
are independent, there is no data-path collective ("replicas only", weak scaling) — and the time is
the max over ranks.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     — the dominant kernel of the measured mode.  fp16 (default): the fused point MLP
                 (mlp_fused_f16_kernel, 35 % of device time), bound "mfma": 4 B N d 2d FLOP / its
                 duration, where the duration is timed with HIP events around hipGraph replays of the
                 layer's three point-stream launches (kernel launches only) as "round - round without
                 that launch", so the kernel meets the cache state of the forward and no host launch
                 gap is inside the timed region; the other two launches are priced against HBM in
                 "hbm_side"; "traffic" = FETCH_SIZE x 2 + WRITE_SIZE of that kernel from the committed
                 --pmc passes (profiles/).  bf16x3 / fp32: the LDS-DMA GEMM at its four call-site shapes.
                 The rocprofv3 --kernel-trace --stats summary of this same command lives in profiles/.
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


def random_state_dict(seed):
    """Random-init weights of the C2 architecture, keyed like the reference state dict
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


def cpu_baseline(p, x, sigma, budget_s=15.0):
    from oracle import cpu_ref
    nb = 8
    xs, ss = x[:nb].cpu(), sigma[:nb].cpu()
    Dn = cpu_ref.uncond_denoiser({k: v.cpu() for k, v in p.items()}, "", H)
    with torch.no_grad():
        # the host may have far more cores than this bounded sample can feed: try a few thread counts, keep the best
        best, cores = None, torch.get_num_threads()
        for nt in sorted({min(cores, c) for c in (16, 32, 64, cores)}):
            torch.set_num_threads(nt)
            Dn(xs[:2], ss[:2])
            t0 = time.perf_counter()
            Dn(xs[:2], ss[:2])
            dt1 = time.perf_counter() - t0
            if best is None or dt1 < best[0]:
                best = (dt1, nt)
        cores = best[1]
        torch.set_num_threads(cores)
        Dn(xs, ss)  # warm-up
        t0 = time.perf_counter()
        it = 0
        while True:
            Dn(xs, ss)
            it += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or it >= 20:
                break
    return {"value": nb * N * it / dt, "unit": "points/s", "cores": cores, "kind": "port",
            "sample": f"oracle/cpu_ref.py fp32 forward on {nb} of the {B} clouds (N={N}, d={D}, L={L}), {it} iterations, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true")
    ap.add_argument("--eager", action="store_true", help="time eager Diffusion.forward calls instead of the hipGraph replay")
    ap.add_argument("--precision", default=os.environ.get("GECCO_PRECISION", "fp16"), choices=["fp32", "bf16x3", "fp16"],
                    help="arithmetic of the linears and attention products: fp16 operands with fp32 accumulation (default; "
                         "~3e-4 from the fp32 reference, bar 1e-3), split-bf16 (3 MFMAs per product, ~1.5e-5) or exact fp32 MFMA (~1e-6)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from gecco_amd import distributed as gd   # rendezvous / barrier / max-over-ranks (covered on gloo in tests/)
    gd.init("nccl", dev)

    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(ge.LIB):
        ge.build()
    gd.barrier()
    from gecco_amd import hip_ops as ops
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
    gd.barrier()
    torch.cuda.synchronize()
    dt = gd.max_over_ranks(time.perf_counter() - t0, dev)
    assert torch.isfinite(out).all(), "non-finite denoiser output"

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
    }
    mode = args.precision
    rec["dtype"] = {"fp16": "f16 (fp16 operands, fp32 accumulate; fp16-stored intermediates, fp32 residual stream and statistics)",
                    "bf16x3": "bf16 (split hi+lo operands, 3 MFMAs per product, fp32 accumulate; fp32 activations in HBM)",
                    "fp32": "f32"}[mode]
    rec["config"]["workload"] = rec["config"]["workload"].replace(
        "fp32 MFMA", {"fp16": "fp16 MFMA", "bf16x3": "split-bf16 MFMA", "fp32": "fp32 MFMA"}[mode])
    if rank == 0 and not args.no_roofline:
        sites = gemm_call_sites(ops, dev, mode)
        tot_f, tot_b, tot_ms, per = 0.0, 0.0, 0.0, {}
        # fp16 mode: kernel-only launches (images prepared) timed inside hipGraphs of the three-launch round
        seq_ms = None
        if mode == "fp16":
            times, seq_ms = time_in_sequence([fn for _, _, _, fn in sites])
        else:
            times = [time_events(fn, 10) for _, _, _, fn in sites]
        for (name, fl, by, fn), t in zip(sites, times):
            per[name] = {"ms": round(t, 4), "tflops": round(fl / (t * 1e-3) / 1e12, 2), "gbs": round(by / (t * 1e-3) / 1e9, 1)}
            tot_f += fl
            tot_b += by
            tot_ms += t
        tf = tot_f / (tot_ms * 1e-3) / 1e12
        gbs = tot_b / (tot_ms * 1e-3) / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "gemm_hbm_traffic.json")
        if os.path.exists(tj):
            traffic = json.load(open(tj)).get(mode, {}).get("bytes_per_launch")
        if mode == "fp16":
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
        elif mode == "bf16x3":
            # The split-bf16 algorithm issues 3 MFMAs per product, so its matrix roof is the dense bf16 peak / 3 =
            # 833 TFLOP/s of 2MNK work.  The launches run at 96..192 FLOP/B (2MNK over fp32 A/residual/C bytes): at or
            # above the ridge of that roof (833 TF / 8 TB/s = 104 FLOP/B), and the PMC counters agree — the matrix
            # pipe is the busiest unit (44 % busy, HBM at 30 % of 8 TB/s; profiles/README.md).  Bound: mfma.
            rec["roofline"] = {"bound": "mfma", "achieved": tf, "peak": PEAK_BF16_MFMA_TFLOPS / 3, "unit": "TFLOP/s",
                               "frac": 3 * tf / PEAK_BF16_MFMA_TFLOPS, "traffic": traffic,
                               "kernel": "gemm_dma_kernel<3,*,true,128> (LDS-DMA ring, 3 x v_mfma_f32_32x32x16_bf16 per product), mean over "
                                         "its 4 per-layer launch shapes; achieved = 2MNK / event-timed duration, peak = dense bf16 MFMA "
                                         "peak (2500 TFLOP/s) / 3 MFMAs per product",
                               "hbm": {"achieved_gbs_algorithmic": gbs, "peak_gbs": PEAK_HBM_GBS, "frac": gbs / PEAK_HBM_GBS},
                               "per_site": per}
        else:
            rec["roofline"] = {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                               "kernel": "gemm_dma_kernel<3,*,false,128> (LDS-DMA ring, v_mfma_f32_32x32x2_f32), mean over its 4 per-layer launch shapes",
                               "per_site": per}
        # the other arithmetic modes beside it, for the record (same model, same inputs)
        for other in ("bf16x3", "fp32"):
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
            rec[{"bf16x3": "split_bf16_mode", "fp32": "exact_fp32_mode"}[other]] = {
                "ms_per_step": ms_o, "points_per_sec": B * N / (ms_o * 1e-3),
                "parity_vs_fp32_reference": {"bf16x3": "~2e-5", "fp32": "~1e-6"}[other]}
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
    if rank == 0:
        print(json.dumps(rec))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
