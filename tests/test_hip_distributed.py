"""The data-parallel training step on the REAL HIP path with two ranks: both processes use cuda:0 and the
gloo backend for the collective (one GPU per box here; on the 8-GPU node the backend is RCCL, the code path above it the
same) — FusedAdamEMA's flat gradient buffer, the bucketed reducer's post-accumulate hooks firing during a backward made of
HIP autograd Functions, 1 / world folded into the fused Adam + EMA kernel — against one process stepping on the full batch.
Reference: Lightning DDP over training_step + Adam + EMACallback (example_configs/shapenet_airplane_unconditional.py:59-77,
diffusion.py:210-222, ema.py:273-325)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
D, L, N, B = 128, 2, 256, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    from oracle import weights as W   # test infrastructure: seeded weights only
    from tests.test_modules_cpu import build_uncond, uncond_state_dict
    m = build_uncond(D, L)
    m.load_state_dict(uncond_state_dict(W.linear_lift_state_dict(11, D, L, 64, 8)), strict=True)
    return m.cuda().train()


def _batch():
    g = torch.Generator().manual_seed(5)
    data = torch.randn(B, N, 3, generator=g)
    noise = torch.randn(B, N, 3, generator=g)
    sigma = torch.tensor([0.05, 0.4, 2.0, 30.0])
    return data, noise, sigma


def _loss(model, data, noise, sigma):
    """EDMLoss with the draws injected (diffusion.py:136-143): mean over the shard."""
    s = sigma.reshape(-1, 1, 1)
    weight = (s ** 2 + 1.0) / (s ** 2)
    den = model(data + noise * s, sigma, None)
    return (100.0 * weight * (den - data) ** 2).mean()


def _run(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      GECCO_PRECISION="bf16x3")
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import distributed as gd
    from gecco_amd import hip_ops
    from gecco_amd.optim import FusedAdamEMA
    hip_ops.set_default_precision("bf16x3")
    torch.cuda.set_device(0)
    gd.init("gloo")
    model = _model()
    gd.broadcast_parameters(model)
    opt = FusedAdamEMA(model.parameters(), lr=1e-3, ema_decay=0.9)
    red = gd.BucketedGradAllReducer(opt, bucket_bytes=64 << 10) if world > 1 else None   # several buckets
    data, noise, sigma = (t.cuda() for t in _batch())
    lo, hi = gd.shard_range(B, rank, world)
    grads, losses = [], []
    for it in range(2):   # two steps: bucket bookkeeping and optimizer state carry over
        # step 0: p.grad are views of the flat buffer (autograd adds in place); step 1: p.grad = None, autograd hands the
        # gradient tensors over and the reducer gathers them bucket by bucket as the backward completes them
        opt.zero_grad(set_to_none=it == 1)
        loss = _loss(model, data[lo:hi], noise[lo:hi], sigma[lo:hi])
        loss.backward()
        if red is not None:
            red.finish()
        grads.append((opt.flat_grad() * opt.grad_scale).cpu().numpy().copy())
        losses.append(float(loss.detach()))
        opt.step()
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
    ema = torch.cat([e.reshape(-1) for e in opt.ema_params]).cpu().numpy()
    nb = len(red.buckets) if red is not None else 0
    q.put((rank, grads, losses, params, ema, nb))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def _spawn(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_two_rank_hip_training_step_matches_the_full_batch():
    single = _spawn(1)[0]
    two = _spawn(2)
    assert two[0][5] >= 3, "the test wants several gradient buckets"
    for rank, grads, losses, params, ema, _ in two:
        # step 0: identical weights on both sides — the averaged shard gradients are the full-batch gradient
        assert _rel(grads[0], single[1][0]) < 2e-4, _rel(grads[0], single[1][0])
        # the two ranks hold the same averaged gradient, parameters and EMA weights bit for bit
        np.testing.assert_array_equal(grads[0], two[0][1][0])
        np.testing.assert_array_equal(params, two[0][3])
        np.testing.assert_array_equal(ema, two[0][4])
        # after two Adam + EMA steps the replicas sit where the single process does: Adam normalises the gradient, so a
        # parameter whose gradient is rounding noise may move by up to lr in either direction on each side — bounded by
        # 2 lr per step, and rare
        dp, de = np.abs(params - single[3]), np.abs(ema - single[4])
        assert dp.max() <= 2 * 2 * 1e-3 + 1e-6 and de.max() <= 2 * 2 * 1e-3 + 1e-6, (dp.max(), de.max())
        assert np.mean(dp > 2e-5) < 0.02 and np.mean(de > 2e-5) < 0.02, (np.mean(dp > 2e-5), np.mean(de > 2e-5))
    # mean of the shard losses = the full-batch loss
    assert abs(0.5 * (two[0][2][0] + two[1][2][0]) - single[2][0]) / abs(single[2][0]) < 1e-5


def _run_sampler(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import functools
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import distributed as gd
    from gecco_amd import hip_ops
    hip_ops.set_default_precision("mixed")
    torch.cuda.set_device(0)
    gd.init("gloo")
    model = _model().eval()
    fn = functools.partial(model.sample_stochastic, context=None)
    out = gd.sample_stochastic_sharded(fn, (5, N, 3), num_steps=4, seed=7, device="cuda", gather=True)   # 5 clouds: 3 + 2
    q.put((rank, out.cpu().numpy()))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def test_two_rank_sharded_sampling_on_the_hip_path_equals_one_process():
    """Replicas: the sampler sharded by batch over two ranks — per-sample noise from (seed, global index), the
    HIP forward bit-identical for a sample whatever batch it sits in, the captured step graph per shard size — returns on every
    rank exactly the clouds one process draws."""
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_run_sampler, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    ref = res[1][0][1]
    assert ref.shape == (5, N, 3) and np.isfinite(ref).all()
    for rank, out in res[2]:
        np.testing.assert_array_equal(out, ref)
