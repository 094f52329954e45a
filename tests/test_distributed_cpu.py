"""The N > 1 path on CPU: two gloo processes exercise the batch-sharding helpers that bench.py --gpus N and a
sharded sampler use on RCCL (no data-path collective: replicas only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gecco_amd import distributed as gd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _toy_sampler(shape, noise, num_steps):
    """Stands in for Diffusion.sample_stochastic on CPU: a per-sample function of that sample's noise only."""
    assert noise.shape == (num_steps + 1, *shape)
    x = noise[0] * 3.0
    for i in range(num_steps):
        x = torch.tanh(x) + 0.1 * noise[i + 1]
    return x.double()


def _worker(rank, world, port, B, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w = gd.init("gloo")
    assert (r, w) == (rank, world)
    full = gd.sample_stochastic_sharded(_toy_sampler, (B, 16, 3), num_steps=4, seed=7, device="cpu")
    local = gd.sample_stochastic_sharded(_toy_sampler, (B, 16, 3), num_steps=4, seed=7, device="cpu", gather=False)
    lo, hi = gd.shard_range(B, rank, world)
    assert torch.equal(full[lo:hi], local)
    t = gd.max_over_ranks(1.0 + rank)
    gd.barrier()
    q.put((rank, full.numpy(), t))   # by value: a tensor travels as a shared-memory handle the exiting child may unlink
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [6, 5])
def test_two_rank_sharded_sampling_equals_single_process(B):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _toy_sampler((B, 16, 3), gd.sample_noise((16, 3), 5, 7, 0, B, "cpu"), 4)
    for rank, full, t in got:
        assert torch.equal(torch.from_numpy(full), single)      # union of the shards == the single-process batch, bit for bit
        assert t == 2.0                       # max over ranks


def test_shard_range_partitions():
    for total in (1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            rs = [gd.shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


def test_sample_noise_is_shard_invariant():
    a = gd.sample_noise((8, 3), 3, 42, 0, 6, "cpu")
    b = torch.cat([gd.sample_noise((8, 3), 3, 42, 0, 2, "cpu"), gd.sample_noise((8, 3), 3, 42, 2, 6, "cpu")], dim=1)
    assert torch.equal(a, b)
    assert not torch.equal(a[:, 0], a[:, 1])


def _dp_worker(rank, world, port, q):
    """Two ranks, each with its shard of a batch: flat-buffer gradient all-reduce == single-process full-batch grads."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    gd.init("gloo")
    torch.manual_seed(rank + 5)                     # ranks start from different weights ...
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    net[2].bias.requires_grad_(False)
    gd.broadcast_parameters(net)                    # ... and agree after the broadcast
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    lo, hi = gd.shard_range(8, rank, world)
    buf = gd.FlatGradBuffer(net.parameters())        # the frozen bias has no span: it can never report a gradient
    red = gd.BucketedGradAllReducer(buf, bucket_bytes=1 << 30)   # one bucket = one message
    assert len(red.buckets) == 1
    loss = ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean()
    loss.backward()
    red.finish()
    assert red.collectives_issued == 1
    grads = [(p.grad * buf.grad_scale).numpy().copy() for p in net.parameters() if p.requires_grad]
    # the same step with the reducer disabled: NO collective anywhere (hooks, finish), local gradients, scale 1
    buf.zero_grad()
    red.enabled = False
    ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
    red.finish()
    assert red.collectives_issued == 1 and buf.grad_scale == 1.0
    local = [p.grad.numpy().copy() for p in net.parameters() if p.requires_grad]
    q.put((rank, [p.detach().numpy().copy() for p in net.parameters()], grads, local))   # by value (see _worker)
    dist.destroy_process_group()


def test_two_rank_gradient_all_reduce_matches_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = [(r, [torch.from_numpy(a) for a in w], [torch.from_numpy(a) for a in g], [torch.from_numpy(a) for a in l])
           for r, w, g, l in got]
    (_, w0, g0, l0), (_, w1, g1, l1) = got
    assert all(torch.equal(a, b) for a, b in zip(w0, w1))
    assert all(torch.equal(a, b) for a, b in zip(g0, g1))
    assert not all(torch.equal(a, b) for a, b in zip(l0, l1))       # disabled: each rank kept its own shard's gradient
    assert all(torch.allclose(0.5 * (a + b), g, atol=1e-6) for a, b, g in zip(l0, l1, g0))
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    net[2].bias.requires_grad_(False)
    with torch.no_grad():
        for p, w in zip(net.parameters(), w0):
            p.copy_(w)
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    ((net(x) - y) ** 2).mean().backward()           # equal shards: mean of shard means == full-batch mean
    ref = [p.grad for p in net.parameters() if p.requires_grad]
    assert all(torch.allclose(a, b, atol=1e-6) for a, b in zip(g0, ref))


def _bucket_worker(rank, world, port, q):
    """The overlapped, bucketed reducer (hooks fire during backward, async all-reduce per bucket, scale folded into the
    optimizer's read of g) against the single-process full-batch gradient; one parameter never receives a gradient."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    gd.init("gloo")
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                              torch.nn.Linear(32, 3))
    unused = torch.nn.Parameter(torch.ones(5))
    params = list(net.parameters()) + [unused]
    buf = gd.FlatGradBuffer(params)
    red = gd.BucketedGradAllReducer(buf, bucket_bytes=512)   # several buckets
    assert len(red.buckets) >= 3
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    lo, hi = gd.shard_range(8, rank, world)
    outs = []
    for step in range(3):                                      # several steps: the bucket bookkeeping resets
        buf.zero_grad(set_to_none=step == 1)                   # step 1: gradients handed over by autograd, gathered per bucket
        ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
        red.finish()
        outs.append([(p.grad * buf.grad_scale).numpy().copy() for p in params])
        # `unused` has no gradient on ANY rank: the per-parameter mask keeps it "missing" (step 1: gradients handed over)
        assert buf.take_missing() == (1 if step == 1 else 0)
    q.put((rank, outs))
    dist.destroy_process_group()


def test_two_rank_bucketed_overlapped_all_reduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                              torch.nn.Linear(32, 3))
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    ref = [p.grad for p in net.parameters()] + [torch.zeros(5)]
    for rank, outs in got:
        for grads in outs:
            assert all(torch.allclose(torch.from_numpy(a), b, atol=1e-6) for a, b in zip(grads, ref))


def _order_worker(rank, world, port, q):
    """Ranks whose hooks fire in DIFFERENT orders (rank 1 differentiates the branches in the opposite order; its last
    parameter gets no gradient at all) must still issue the same sequence of same-sized messages: fixed bucket order."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    gd.init("gloo")
    torch.manual_seed(1)
    ws = [torch.nn.Parameter(torch.randn(40 + 8 * i)) for i in range(5)]   # different sizes: a mispaired message would fail
    buf = gd.FlatGradBuffer(ws)
    red = gd.BucketedGradAllReducer(buf, bucket_bytes=64)
    assert len(red.buckets) == 5
    order = []
    launch = red._launch
    red._launch = lambda b: (order.append(red.buckets.index(b)), launch(b))[1]
    # independent scalar losses, backward one at a time in a rank-dependent order; all five backward passes belong to ONE
    # step, so the test drives the hooks directly through .backward() on detached graphs before finish()
    idx = [0, 1, 2, 3, 4] if rank == 0 else [3, 1, 4, 0]       # rank 1: parameter 2 never receives a gradient
    buf.zero_grad(set_to_none=True)
    torch.autograd.backward([(w * (i + 1 + rank)).sum() for i, w in enumerate(ws) if i in idx])
    red.finish()
    assert buf.take_missing() == 0   # parameter 2 had no gradient on rank 1, but rank 0 supplied one: not missing after the reduction
    q.put((rank, order, [(w.grad * buf.grad_scale).numpy().copy() for w in ws]))
    dist.destroy_process_group()


def test_bucket_collectives_are_issued_in_one_fixed_order():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_order_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, order, grads in got:
        assert order == [4, 3, 2, 1, 0]
        for i, g in enumerate(grads):
            want = ((i + 1) + (i + 2 if i != 2 else 0)) / 2.0      # mean over the two ranks of d/dw sum(w * c) = c
            assert np.allclose(g, want), (rank, i, g[:3], want)


def test_second_backward_before_finish_is_refused():
    """Gradient accumulation across backward passes would add local gradients onto an already reduced slice."""
    w = [torch.nn.Parameter(torch.ones(8)), torch.nn.Parameter(torch.ones(8))]
    buf = gd.FlatGradBuffer(w)
    red = gd.BucketedGradAllReducer(buf, bucket_bytes=16)
    (w[0].sum() + w[1].sum()).backward()
    (w[0].sum() + w[1].sum()).backward()          # world 1: nothing was reduced, accumulation is harmless and allowed
    red.finish()
    assert float(w[0].grad[0]) == 2.0
    red._collective = lambda: True                # as if in a group: the buckets of the first pass count as reduced
    red._launch = lambda b: b.__setitem__("launched", True)
    buf.zero_grad()
    (w[0].sum() + w[1].sum()).backward()
    with pytest.raises(RuntimeError, match="already all-reduced"):
        (w[0].sum() + w[1].sum()).backward()


def test_bench_self_launcher_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the script spawns its own ranks (fresh processes, env
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), they rendezvous, and rank 0 prints the one JSON line.  Exercised on gloo
    with the stand-in step (`--selftest-launcher`): the compute path needs the GPU, the plumbing does not."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launcher", "--steps", "3"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3
    assert rec["ms_per_step"] >= 2.0                      # max over ranks: rank 1 sleeps 2 ms per step
    # the self-description a first multi-GPU run carries (bench.distributed_report): ranks the group really has, backend,
    # all-reduce time / bus bandwidth at two sizes, rank 0's own un-barriered rate
    d = rec["distributed"]
    assert d["ranks_in_group"] == 2 and d["backend"] == "gloo" and "collective_library" in d
    assert len(d["allreduce"]) == 2 and all(v["ms"] > 0 and v["busbw_gbs"] > 0 for v in d["allreduce"].values())
    assert d["rank0_points_per_sec_unbarriered"] > 0
    # every rank's own rate, in rank order (round 6): rank 1 sleeps twice as long per step as rank 0
    pr = d["per_rank_points_per_sec"]
    assert len(pr) == 2 and pr[0] == d["rank0_points_per_sec_unbarriered"] and 0 < pr[1] < pr[0]
