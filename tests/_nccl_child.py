"""Child process of tests/test_hip_nccl.py: a fresh interpreter (nothing has touched the GPU before the process group
exists) that runs two data-parallel training steps of a small unconditional model on the HIP path in a ONE-RANK `nccl`
(= RCCL) process group and writes the gradients / parameters / EMA weights it ends with.

  python tests/_nccl_child.py <out.npz> <force_collective 0|1> [amp]   (amp: the reference's 16-mixed trainer setting around the step)

With WORLD_SIZE > 1 in the environment (tests/test_hip_nccl.py::test_all_visible_gpus_*: one process per visible GPU, RANK / LOCAL_RANK set
by the test; GECCO_CHILD_BACKEND=gloo + GECCO_CHILD_ONE_GPU=1: the same code with every rank on cuda:0, which a 1-GPU box can run) each
rank steps on its OWN shard of a global batch, writes `<out>.rank<r>.npz`, and rank 0 also records the group's size and the bus
bandwidth of the 53.9 MB gradient all-reduce.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
D, L, N, B = 128, 2, 256, 4


def main():
    out, force = sys.argv[1], sys.argv[2] == "1"
    amp = len(sys.argv) > 3 and sys.argv[3] == "amp"
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import distributed as gd
    from gecco_amd import hip_ops
    from gecco_amd.optim import FusedAdamEMA
    from oracle import weights as W   # test infrastructure: seeded weights only
    from tests.test_modules_cpu import build_uncond, uncond_state_dict
    hip_ops.set_default_precision("bf16x3")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("GECCO_CHILD_BACKEND", "nccl")
    local = 0 if os.environ.get("GECCO_CHILD_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if force or world > 1:
        gd.init(backend, dev, force=True)      # RANK / WORLD_SIZE from the environment (a one-rank group when WORLD_SIZE=1)
        import torch.distributed as dist
        assert dist.is_initialized() and dist.get_backend() == backend and dist.get_world_size() == world
        force = True
    m = build_uncond(D, L)
    m.load_state_dict(uncond_state_dict(W.linear_lift_state_dict(11, D, L, 64, 8)), strict=True)
    model = m.cuda().train()
    gd.broadcast_parameters(model)
    opt = FusedAdamEMA(model.parameters(), lr=1e-3, ema_decay=0.9, amp_on_device=True)
    red = gd.BucketedGradAllReducer(opt, bucket_bytes=64 << 10, force_collective=force)   # several buckets
    g = torch.Generator().manual_seed(5 + 1000 * rank)          # (rank 0 of any world = the one-rank run's batch)
    data = torch.randn(B, N, 3, generator=g).cuda()
    noise = torch.randn(B, N, 3, generator=g).cuda()
    sigma = torch.tensor([0.05, 0.4, 2.0, 30.0]).cuda()
    s = sigma.reshape(-1, 1, 1)
    weight = (s ** 2 + 1.0) / (s ** 2)
    grads = []
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10) if amp else None
    for it in range(2):
        # step 1 with p.grad = None: the reducer gathers per bucket after sync_side_stream() — the side-stream weight
        # gradients (autograd._linear_dw) are in play exactly as in bench.py --train
        opt.zero_grad(set_to_none=it == 1)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            den = model(data + noise * s, sigma, None)
            loss = (100.0 * weight * (den.float() - data) ** 2).mean()
        (scaler.scale(loss) if amp else loss).backward()
        red.finish()
        grads.append((opt.flat_grad() * opt.grad_scale).cpu().numpy().copy())
        if amp:
            scaler.step(opt)      # FusedAdamEMA inside the scaler's protocol: unscale + found_inf on the device, 1 / world folded in
            scaler.update()
        else:
            opt.step()
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
    ema = torch.cat([e.reshape(-1) for e in opt.ema_params]).cpu().numpy()
    extra = {}
    if world > 1:
        import time
        import torch.distributed as dist
        out = f"{out}.rank{rank}.npz"
        n = 53_900_000 // 4                                      # the shipped model's flat gradient buffer
        buf = torch.ones(n, dtype=torch.float32, device=dev)
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        extra = dict(ranks_in_group=dist.get_world_size(), allreduce_ms=dt * 1e3, busbw_gbs=2 * (world - 1) / world * n * 4 / dt / 1e9,
                     backend=backend)
        print(f"[rank {rank}] ranks_in_group={world} backend={backend} all-reduce 53.9 MB: {dt * 1e3:.3f} ms, bus {extra['busbw_gbs']:.1f} GB/s", flush=True)
    np.savez(out, g0=grads[0], g1=grads[1], params=params, ema=ema, issued=red.collectives_issued, buckets=len(red.buckets), **extra)
    if force:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
